// cuda_on_cpu.h -- a minimal CUDA execution model on the host CPU, used ONLY to run the
// reference's own .cu kernels (read in place from /root/reference by oracle/build_ref.py) so the
// CPU oracle can be checked against the real reference code.  Test infrastructure.
//
//   * __global__ kernels become plain functions; kernel<<<grid, block, ...>>>(args) is rewritten
//     by build_ref.py to PVREF_LAUNCH(kernel, grid, block, ...)(args);
//   * every CUDA thread of a block is a ucontext fiber; __syncthreads() yields to the block
//     scheduler, which resumes the fibers round-robin (so barriers and __shared__ work);
//   * blocks run one after another, threads in ascending threadIdx order between barriers, so
//     atomicAdd is a plain read-modify-write: a legal, deterministic CUDA schedule;
//   * atomicAdd is noinline: its argument is rounded BEFORE the add, as on the GPU (the
//     compiler must not contract `*p + a*b` into an fma across the call).
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };
extern uint3_ threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

// CUDA's overloaded min/max (math_functions.hpp): ints, floats (fminf/fmaxf), mixed -> double
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }
inline double min(double a, double b) { return fmin(a, b); }
inline double max(double a, double b) { return fmax(a, b); }
inline double min(float a, double b) { return fmin((double)a, b); }
inline double min(double a, float b) { return fmin(a, (double)b); }
inline double max(float a, double b) { return fmax((double)a, b); }
inline double max(double a, float b) { return fmax(a, (double)b); }

__attribute__((noinline)) float atomicAdd(float *p, float v);
__attribute__((noinline)) int atomicAdd(int *p, int v);

// --- the bits of the CUDA runtime API the reference's launch code touches --------------------
typedef int cudaError_t;
enum { cudaSuccess = 0 };
inline cudaError_t cudaGetLastError() { return cudaSuccess; }
inline const char *cudaGetErrorString(cudaError_t) { return "no error"; }
namespace at { namespace cuda { inline int getCurrentCUDAStream() { return 0; } } }

namespace pvref {
void yield_to_scheduler();                              // __syncthreads()
void run_grid(dim3 grid, dim3 block, const std::function<void()> &thread_body);
// Threads of a block run in ascending threadIdx order between barriers by default; `true` runs
// them in descending order (needed by the reference's FPS kernel, see ref_api.cpp).
void set_descending_schedule(bool on);

template <class K>
struct Launcher {
  K kernel;
  dim3 grid, block;
  template <class... A>
  void operator()(A... args) const {
    K k = kernel;
    run_grid(grid, block, [=]() { k(args...); });
  }
};
template <class K>
Launcher<K> make_launcher(K k, dim3 grid, dim3 block, size_t = 0, int = 0) { return Launcher<K>{k, grid, block}; }
}  // namespace pvref

inline void __syncthreads() { pvref::yield_to_scheduler(); }
#define PVREF_LAUNCH(kernel, ...) pvref::make_launcher(kernel, __VA_ARGS__)
