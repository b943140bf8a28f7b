// fps.hip -- iterative furthest point sampling for gfx950.
//
// Reference: sampling/sampling.cu:86-167 -- 512 threads per cloud, distances in global memory,
// per step a 9-level LDS tree with a barrier per level.  FPS is a chain of M-1 dependent steps,
// so the design goal is the shortest possible step:
//   - coordinates and running point-to-set distances live in REGISTERS (PPT points per thread);
//   - the per-step arg-max is one 64-bit max-reduction: key = (float bits of d) << 32 | ~tie,
//     6 cross-lane steps inside the wave, then ONE barrier and a <=16-entry LDS read across
//     waves (double-buffered slots, so a single barrier per step suffices);
//   - the winner's coordinates come from an LDS copy of the cloud (broadcast read).
// Tie rule: the reference's result among equidistant candidates is an artefact of its launch
// shape (512 strided slots with strict '>', then a tree that keeps the LEFT slot): the winner
// is the candidate with the smallest (k mod 512, k).  `tie` encodes exactly that, so the
// result does not depend on THIS kernel's block size.
#include "common.h"

namespace pvcnn {

__device__ __forceinline__ unsigned tie_key(int k) { return ((unsigned)(k & 511) << 20) | (unsigned)(k >> 9); }
__device__ __forceinline__ int tie_decode(unsigned key) { return (int)(((key & 0xFFFFFu) << 9) | (key >> 20)); }

// Maximum of a 64-bit key over the wave, returned wave-uniform.  The M-1 steps of FPS are a dependent chain,
// so this reduction IS the kernel's critical path: four DPP steps (cross-lane moves inside the VALU, a few
// cycles each: quad xor 1, quad xor 2, half-row mirror, row mirror) leave every 16-lane row with its maximum,
// then four v_readlane pairs + scalar compares combine the rows.  (The generic __shfl_xor lowers to
// ds_bpermute_b32 -- an LDS round trip of ~100 cycles -- twice per 64-bit step: 12 of them per reduction.)
#define PVCNN_DPP_MAX_STEP(CTRL)                                                                  \
  do {                                                                                            \
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xF, 0xF, false); \
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xF, 0xF, false); \
    const bool take = (ohi > hi) || (ohi == hi && olo > lo);                                      \
    hi = take ? ohi : hi;                                                                         \
    lo = take ? olo : lo;                                                                         \
  } while (0)

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  PVCNN_DPP_MAX_STEP(0xB1);    // quad_perm [1,0,3,2]
  PVCNN_DPP_MAX_STEP(0x4E);    // quad_perm [2,3,0,1]
  PVCNN_DPP_MAX_STEP(0x141);   // row_half_mirror
  PVCNN_DPP_MAX_STEP(0x140);   // row_mirror
  unsigned long long best = 0ull;
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    const unsigned long long r = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, row * 16) << 32) |
                                 (unsigned)__builtin_amdgcn_readlane((int)lo, row * 16);
    best = (r > best) ? r : best;
  }
  return best;
}
#undef PVCNN_DPP_MAX_STEP

template <int THREADS, int PPT>
__global__ __launch_bounds__(THREADS) void fps_kernel(const float *__restrict__ coords, int N, int M,
                                                      int lds_coords, float *__restrict__ distances,
                                                      int32_t *__restrict__ indices) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int W = THREADS / kWave;
  unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem);   // [2][W]
  float *lc = reinterpret_cast<float *>(smem + 2 * W * sizeof(unsigned long long));   // [3][N] when lds_coords
  const int b = blockIdx.x, tid = threadIdx.x;
  coords += (size_t)b * 3 * N;
  indices += (size_t)b * M;

  float x[PPT], y[PPT], z[PPT], dist[PPT];
#pragma unroll
  for (int q = 0; q < PPT; ++q) {
    const int k = tid + q * THREADS;
    const bool valid = k < N;
    x[q] = valid ? coords[k] : 0.f;
    y[q] = valid ? coords[k + N] : 0.f;
    z[q] = valid ? coords[k + 2 * N] : 0.f;
    dist[q] = valid ? 1e38f : -1.0f;   // sampling.cpp:53-54; -1 marks "no such point"
    if (lds_coords && valid) { lc[k] = x[q]; lc[k + N] = y[q]; lc[k + 2 * N] = z[q]; }
  }
  if (tid == 0) indices[0] = 0;
  __syncthreads();

  int old = 0;
  for (int j = 1; j < M; ++j) {
    float x1, y1, z1;
    if (lds_coords) { x1 = lc[old]; y1 = lc[old + N]; z1 = lc[old + 2 * N]; }
    else            { x1 = coords[old]; y1 = coords[old + N]; z1 = coords[old + 2 * N]; }
    // With 8 points per thread and 16 waves on one CU the step is VALU-bound, so the per-point work is kept to
    // ten instructions: the thread's own winner is tracked as (distance, q) with a strict '>' -- its points
    // k = tid + q*THREADS share k mod 512 when THREADS is a multiple of 512, so among equal distances the
    // smallest q IS the tie rule's winner -- and the 64-bit key is built once per thread, not once per point.
    float bd = -1.0f;
    int bq = 0;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
      const float ex = x[q] - x1, ey = y[q] - y1, ez = z[q] - z1;
      const float d = fmaf(ez, ez, fmaf(ex, ex, ey * ey));
      const float d2 = fminf(d, dist[q]);
      dist[q] = d2;
      if (THREADS % 512 == 0) {
        const bool better = d2 > bd;
        bd = better ? d2 : bd;
        bq = better ? q : bq;
      }
    }
    unsigned long long best = 0ull;
    if (THREADS % 512 == 0) {
      best = (bd >= 0.0f) ? (((unsigned long long)__float_as_uint(bd) << 32) | (0xFFFFFFFFu - tie_key(tid + bq * THREADS))) : 0ull;
    } else {
#pragma unroll
      for (int q = 0; q < PPT; ++q) {
        const int k = tid + q * THREADS;
        const unsigned long long cand =
            (dist[q] >= 0.0f) ? (((unsigned long long)__float_as_uint(dist[q]) << 32) | (0xFFFFFFFFu - tie_key(k))) : 0ull;
        best = (cand > best) ? cand : best;
      }
    }
    best = wave_max_u64(best);
    if (W > 1) {
      // cross-wave, 16 waves: lane w of every wave fetches wave w's key (one LDS read per lane, not 16 per thread)
      // and the same DPP reduction makes the result uniform again
      unsigned long long *sl = slots + (j & 1) * W;
      if ((tid & 63) == 0) sl[tid >> 6] = best;
      __syncthreads();
      if (W >= 16) {
        const int lane = tid & 63;
        best = wave_max_u64(lane < W ? sl[lane] : 0ull);
      } else {   // few waves: W broadcast reads + compares per thread are cheaper than a second reduction (measured)
#pragma unroll
        for (int w = 0; w < W; ++w) { const unsigned long long o = sl[w]; best = (o > best) ? o : best; }
      }
    }
    old = tie_decode(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
    if (tid == 0) indices[j] = old;
  }
  if (distances) {
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
      const int k = tid + q * THREADS;
      if (k < N) distances[(size_t)b * N + k] = dist[q];
    }
  }
}

// Unbounded-N fallback: distances live in the caller's (B,N) scratch (must be given).
__global__ __launch_bounds__(1024) void fps_global_kernel(const float *__restrict__ coords, int N, int M,
                                                          float *__restrict__ distances,
                                                          int32_t *__restrict__ indices) {
  __shared__ unsigned long long slots[2][16];
  const int b = blockIdx.x, tid = threadIdx.x;
  coords += (size_t)b * 3 * N;
  distances += (size_t)b * N;
  indices += (size_t)b * M;
  for (int k = tid; k < N; k += 1024) distances[k] = 1e38f;
  if (tid == 0) indices[0] = 0;
  int old = 0;
  for (int j = 1; j < M; ++j) {
    const float x1 = coords[old], y1 = coords[old + N], z1 = coords[old + 2 * N];
    unsigned long long best = 0ull;
    for (int k = tid; k < N; k += 1024) {
      const float ex = coords[k] - x1, ey = coords[k + N] - y1, ez = coords[k + 2 * N] - z1;
      const float d = fmaf(ez, ez, fmaf(ex, ex, ey * ey));
      const float d2 = fminf(d, distances[k]);
      distances[k] = d2;
      const unsigned long long cand =
          (d2 >= 0.0f) ? (((unsigned long long)__float_as_uint(d2) << 32) | (0xFFFFFFFFu - tie_key(k))) : 0ull;
      best = (cand > best) ? cand : best;
    }
    best = wave_max_u64(best);
    if ((tid & 63) == 0) slots[j & 1][tid >> 6] = best;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; ++w) { const unsigned long long o = slots[j & 1][w]; best = (o > best) ? o : best; }
    old = tie_decode(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
    if (tid == 0) indices[j] = old;
  }
}

template <int THREADS, int PPT>
static int launch_fps(const float *coords, int B, int N, int M, float *distances, int32_t *indices, hipStream_t s) {
  constexpr int W = THREADS / kWave;
  size_t lds = 2 * W * sizeof(unsigned long long);
  int lds_coords = 0;
  if ((size_t)3 * N * sizeof(float) + lds <= 144 * 1024) { lds_coords = 1; lds += (size_t)3 * N * sizeof(float); }
  auto k = fps_kernel<THREADS, PPT>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("fps: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(k, dim3(B), dim3(THREADS), lds, s, coords, N, M, lds_coords, distances, indices);
  return check_launch("fps");
}

template <int THREADS>
static int launch_fps_ppt(const float *coords, int B, int N, int M, float *distances, int32_t *indices, hipStream_t s) {
  const int need = ceil_div(N, THREADS);
  if (need <= 1) return launch_fps<THREADS, 1>(coords, B, N, M, distances, indices, s);
  if (need <= 2) return launch_fps<THREADS, 2>(coords, B, N, M, distances, indices, s);
  if (need <= 4) return launch_fps<THREADS, 4>(coords, B, N, M, distances, indices, s);
  if (need <= 8) return launch_fps<THREADS, 8>(coords, B, N, M, distances, indices, s);
  return launch_fps<THREADS, 16>(coords, B, N, M, distances, indices, s);
}

}  // namespace pvcnn

using namespace pvcnn;

extern "C" int pvcnn_fps(const float *coords, int B, int N, int M, float *distances, int32_t *indices,
                         void *stream) {
  PVCNN_REQUIRE(B >= 0 && N >= 0 && M >= 0, "negative size");
  if (B == 0 || M == 0) return 0;
  PVCNN_REQUIRE(N > 0, "cannot sample from an empty cloud");
  PVCNN_REQUIRE(coords && indices, "null pointer");
  PVCNN_REQUIRE(N < (1 << 29), "N too large for the tie key");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (N > PVCNN_FPS_MAX_RESIDENT_POINTS) {
    PVCNN_REQUIRE(distances, "N > PVCNN_FPS_MAX_RESIDENT_POINTS needs the distances scratch");
    hipLaunchKernelGGL(fps_global_kernel, dim3(B), dim3(1024), 0, s, coords, N, M, distances, indices);
    return check_launch("fps_global");
  }
  if (N <= 1024) return launch_fps_ppt<64>(coords, B, N, M, distances, indices, s);
  if (N <= 4096) return launch_fps_ppt<256>(coords, B, N, M, distances, indices, s);
  return launch_fps_ppt<1024>(coords, B, N, M, distances, indices, s);
}
