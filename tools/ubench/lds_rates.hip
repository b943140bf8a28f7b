// lds_rates.hip -- measured rates of the primitives the scatter/gather designs choose between:
// random-index LDS float atomics, LDS int atomics, plain LDS read / read-modify-write, and
// L2 (global) float atomics.  Standalone: hipcc --offload-arch=gfx950 -O3 lds_rates.hip -o lds_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int T = 256, ROW = 4096, OPS = 512;

__device__ __forceinline__ unsigned lcg(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ __launch_bounds__(T) void k(float *out, float *gslab, int row) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < row; i += T) lds[i] = 0.f;
  __syncthreads();
  unsigned s = blockIdx.x * 7919u + threadIdx.x * 104729u + 1u;
  float acc = 0.f;
  int *ilds = reinterpret_cast<int *>(lds);
#pragma unroll 8
  for (int i = 0; i < OPS; ++i) {
    const int idx = lcg(s) % row;
    if (MODE == 0) atomicAdd(&lds[idx], 1.0f);                  // ds_add_f32
    else if (MODE == 1) atomicAdd(&ilds[idx], 1);               // ds_add_u32
    else if (MODE == 2) acc += lds[idx];                        // ds_read_b32
    else if (MODE == 3) lds[idx] = acc + i;                     // ds_write_b32
    else if (MODE == 4) atomicAdd(&gslab[(size_t)(blockIdx.x % 1024) * row + idx], 1.0f);   // global_atomic_add_f32
    else if (MODE == 5) { float v = lds[idx]; lds[idx] = v + 1.0f; }   // non-atomic RMW (racy; rate only)
    else if (MODE == 6) acc += atomicAdd(&lds[idx], 1.0f);      // ds_add_rtn_f32
    else if (MODE == 7) atomicAdd(&lds[(idx & ~63) | (threadIdx.x & 63)], 1.0f);   // conflict-free banks, f32 atomic
    else if (MODE == 8) atomicAdd(&ilds[(idx & ~63) | (threadIdx.x & 63)], 1);    // conflict-free banks, int atomic
  }
  __syncthreads();
  if (MODE == 2 || MODE == 6) out[blockIdx.x * T + threadIdx.x] = acc;
  else out[blockIdx.x * T + threadIdx.x] = lds[threadIdx.x];
}

template <int MODE>
void run(const char *name, float *out, float *gslab, int row) {
  const int blocks = 2048;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(T), row * 4, 0, out, gslab, row);
  hipEventRecord(a);
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(T), row * 4, 0, out, gslab, row);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double ops = (double)blocks * T * OPS * reps;
  printf("%-28s row=%6d  %8.1f us/launch  %7.2f Gop/s  %6.3f lane-ops/clk/CU (2.4GHz,256CU)\n", name, row,
         ms * 1e3 / reps, ops / (ms * 1e-3) / 1e9, ops / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  float *out, *gslab;
  hipMalloc(&out, 2048 * T * 4);
  hipMalloc(&gslab, (size_t)1024 * 32768 * 4);
  hipMemset(gslab, 0, (size_t)1024 * 32768 * 4);
  for (int row : {4096, 32768}) {
    if (row * 4 > 64 * 1024) {
      hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, row * 4);
      hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, row * 4);
      hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, row * 4);
      hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, row * 4);
      hipFuncSetAttribute((const void *)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, row * 4);
      hipFuncSetAttribute((const void *)k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, row * 4);
      hipFuncSetAttribute((const void *)k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, row * 4);
      hipFuncSetAttribute((const void *)k<7>, hipFuncAttributeMaxDynamicSharedMemorySize, row * 4);
      hipFuncSetAttribute((const void *)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, row * 4);
    }
    run<0>("ds_add_f32 random", out, gslab, row);
    run<1>("ds_add_u32 random", out, gslab, row);
    run<6>("ds_add_rtn_f32 random", out, gslab, row);
    run<7>("ds_add_f32 conflict-free", out, gslab, row);
    run<8>("ds_add_u32 conflict-free", out, gslab, row);
    run<2>("ds_read_b32 random", out, gslab, row);
    run<3>("ds_write_b32 random", out, gslab, row);
    run<5>("ds read+write (non-atomic)", out, gslab, row);
    run<4>("global_atomic_add_f32 rand", out, gslab, row);
  }
  return 0;
}
