// fps.hip -- iterative furthest point sampling for gfx950.
//
// Reference: sampling/sampling.cu:86-167 -- 512 threads per cloud, distances in global memory,
// per step a 9-level LDS tree with a barrier per level.  FPS is a chain of M-1 dependent steps,
// so the design goal is the shortest possible step:
//   - coordinates and running point-to-set distances live in REGISTERS (PPT points per thread);
//   - the per-step arg-max is a FLOAT maximum (v_max3 trees per thread, 6 DPP-fused v_max_f32 per wave) followed by
//     "who holds it" from compare masks on the scalar unit; then ONE barrier and a <=16-entry LDS read across waves
//     (double-buffered slots of 64-bit keys (float bits of d) << 32 | ~tie, so a single barrier per step suffices);
//     steps with exactly equidistant candidates fall back to a 64-bit key reduction that applies the tie rule in full;
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

typedef float fps_f2 __attribute__((ext_vector_type(2)));

// Maximum of a float over the wave (any signs, no NaNs), wave-uniform: six v_max_f32 with the cross-lane move folded into the
// instruction (DPP: quad xor 1, quad xor 2, half-row mirror, row mirror, then row_bcast 15 / 31 carry the row maxima up to lane 63).
// 6 VALU instructions where the 64-bit key reduction above takes ~50: with 16 waves sharing one CU's VALUs that difference was a
// third of the FPS step.  (s_nop: a DPP read of a VGPR needs two wait states after the VALU write; the assembler does not see
// into an asm block.)
__device__ __forceinline__ float wave_max_f32(float v) {
  asm("s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
      : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Maximum of an unsigned over each 16-lane row, left in every lane of the row.
__device__ __forceinline__ unsigned row_max_u32(unsigned v) {
  asm("s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
      : "+v"(v));
  return v;
}

// The i-th of a thread's points k = tid + q * THREADS in the order of the tie rule (k mod 512, k): a thread that scans its points
// in this order with a strict '>' keeps the rule's winner among its own equidistant candidates.  THREADS a multiple of 512: every
// q has the same k mod 512, the order is q.  THREADS < 512 (64, 256): R = 512 / THREADS residues, then they repeat.
template <int THREADS, int PPT>
__device__ __forceinline__ constexpr int fps_tie_order(int i) {
  constexpr int R = THREADS >= 512 ? PPT : 512 / THREADS;   // (THREADS >= 512: one residue class, G = 1 below is the identity)
  constexpr int G = PPT <= R ? 1 : PPT / R;
  return G == 1 ? i : (i / G) + R * (i % G);
}

// v_min_f32 / v_max3_f32 as written (fminf / fmaxf on a register the compiler cannot prove canonical cost an extra
// v_max_f32 x, x, x each; a NaN distance -- a NaN coordinate -- loses against the number in both, like the reference's '>' test)
__device__ __forceinline__ float fps_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float fps_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// One step = distances of the thread's PPT points to the newest sample (packed fp32: two points per v_pk_add / v_pk_mul / v_pk_fma),
// the maximum of the thread's distances (v_max3 trees over groups of eight in tie order), a FLOAT maximum over the wave (6 DPP
// instructions), then WHO holds it -- lane, group, point -- from v_cmp masks on the scalar unit; one barrier; the same over the waves.
// FPS is a chain of M-1 dependent steps on one CU per cloud, and every VALU instruction of a step is on that chain once per wave of
// the SIMD: hence few waves with many points each (256 threads x 32 points for N = 8192: the per-wave part of a step is paid once per
// SIMD, not four times), and no per-point index tracking.  Exactly equidistant candidates in different lanes (THREADS < 512) or in
// different waves are rare (duplicated points at distance 0, lattices): those steps take the 64-bit key reduction, which implements
// the tie rule in full.
template <int THREADS, int PPT, bool LDSC>
__global__ __launch_bounds__(THREADS) void fps_kernel(const float *__restrict__ coords, int N, int M, float *__restrict__ distances,
                                                      int32_t *__restrict__ indices) {
  static_assert(PPT % 2 == 0 && (THREADS % 512 == 0 || 512 % THREADS == 0), "launch shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int W = THREADS / kWave, H = PPT / 2, GS = PPT < 8 ? PPT : 8, NG = PPT / GS;
  unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem);   // [2][W]
  float *lc = reinterpret_cast<float *>(smem + 2 * W * sizeof(unsigned long long));   // [3][N] when LDSC
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  coords += (size_t)b * 3 * N;
  indices += (size_t)b * M;

  fps_f2 x[H], y[H], z[H];
  float dist[PPT];
#pragma unroll
  for (int q = 0; q < PPT; ++q) {          // straight-line loads (clamped index): all 3 * PPT in flight at once
    const int k = tid + q * THREADS, kc = k < N ? k : N - 1;
    x[q >> 1][q & 1] = coords[kc];
    y[q >> 1][q & 1] = coords[kc + N];
    z[q >> 1][q & 1] = coords[kc + 2 * N];
    dist[q] = k < N ? 1e38f : -1.0f;       // sampling.cpp:53-54; -1 marks "no such point"
  }
  if (LDSC) {
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
      const int k = tid + q * THREADS;
      if (k < N) { lc[k] = x[q >> 1][q & 1]; lc[k + N] = y[q >> 1][q & 1]; lc[k + 2 * N] = z[q >> 1][q & 1]; }
    }
  }
  if (tid == 0) indices[0] = 0;
  __syncthreads();

  int old = 0;
  for (int j = 1; j < M; ++j) {
    float x1, y1, z1;
    if (LDSC) { x1 = lc[old]; y1 = lc[old + N]; z1 = lc[old + 2 * N]; }
    else      { x1 = coords[old]; y1 = coords[old + N]; z1 = coords[old + 2 * N]; }
    const fps_f2 X1 = {x1, x1}, Y1 = {y1, y1}, Z1 = {z1, z1};
#pragma unroll
    for (int p = 0; p < H; ++p) {
      const fps_f2 ex = x[p] - X1, ey = y[p] - Y1, ez = z[p] - Z1;
      const fps_f2 d = __builtin_elementwise_fma(ez, ez, __builtin_elementwise_fma(ex, ex, ey * ey));   // = fmaf(ez,ez,fmaf(ex,ex,ey*ey))
      dist[2 * p] = fps_min(d[0], dist[2 * p]);
      dist[2 * p + 1] = fps_min(d[1], dist[2 * p + 1]);
    }
    // group g = the points at positions g*GS .. g*GS+GS-1 of the tie order
    float gm[NG], bd = -1.0f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float m = dist[fps_tie_order<THREADS, PPT>(g * GS)];
#pragma unroll
      for (int i = 1; i + 1 < GS; i += 2)
        m = fps_max3(m, dist[fps_tie_order<THREADS, PPT>(g * GS + i)], dist[fps_tie_order<THREADS, PPT>(g * GS + i + 1)]);
      gm[g] = fps_max3(m, dist[fps_tie_order<THREADS, PPT>(g * GS + GS - 1)], -1.0f);       // GS is even: one value left
      bd = fps_max3(bd, gm[g], -1.0f);
    }
    // the wave's candidate, as the 64-bit key (float bits of the distance) << 32 | ~tie
    const float wmax = wave_max_f32(bd);
    const unsigned long long holders = __ballot(bd == wmax);
    unsigned long long best;
    if (wmax < 0.0f) {
      best = 0ull;                                                    // no point in this wave
    } else if (THREADS % 512 == 0 || __popcll(holders) == 1) {        // THREADS % 512 == 0: k mod 512 grows with the lane -> the first holder wins
      const int wl = __ffsll((long long)holders) - 1;
      int wg = 0;
#pragma unroll
      for (int g = NG - 1; g > 0; --g)
        if ((__ballot(gm[g] == wmax) >> wl) & 1) wg = g;
      if ((__ballot(gm[0] == wmax) >> wl) & 1) wg = 0;
      int wi = 0;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (g == wg) {
#pragma unroll
          for (int i = GS - 1; i >= 0; --i)
            if ((__ballot(dist[fps_tie_order<THREADS, PPT>(g * GS + i)] == wmax) >> wl) & 1) wi = g * GS + i;
        }
      }
      constexpr int R = THREADS >= 512 ? PPT : 512 / THREADS, G = PPT <= R ? 1 : PPT / R;      // fps_tie_order at run time
      const int wq = G == 1 ? wi : (wi / G) + R * (wi % G);
      best = ((unsigned long long)__float_as_uint(wmax) << 32) | (0xFFFFFFFFu - tie_key((tid & ~63) + wl + wq * THREADS));
    } else {                                                          // equidistant candidates in several lanes: the tie rule on every point
      best = 0ull;
#pragma unroll
      for (int q = 0; q < PPT; ++q) {
        const unsigned long long cand =
            (dist[q] == wmax) ? (((unsigned long long)__float_as_uint(wmax) << 32) | (0xFFFFFFFFu - tie_key(tid + q * THREADS))) : 0ull;
        best = (cand > best) ? cand : best;
      }
      best = wave_max_u64(best);
    }
    if (W > 1) {
      // lane w of every wave fetches wave w's key: one LDS read per lane, a 4-step row maximum of the distance words, a ballot
      unsigned long long *sl = slots + (j & 1) * W;
      if (lane == 0) sl[tid >> 6] = best;
      __syncthreads();
      const unsigned long long mine = lane < W ? sl[lane] : 0ull;
      const unsigned hi = (unsigned)(mine >> 32), lo = (unsigned)mine;
      const unsigned top = (unsigned)__builtin_amdgcn_readfirstlane((int)row_max_u32(hi));
      const unsigned long long tops = __ballot(hi == top && lane < W);
      if (__popcll(tops) == 1) {
        best = ((unsigned long long)top << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, __ffsll((long long)tops) - 1);
      } else {
        best = wave_max_u64(mine);
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
  const size_t slots = 2 * W * sizeof(unsigned long long), cloud = (size_t)3 * N * sizeof(float);
  if (cloud + slots <= 144 * 1024) {     // the cloud in LDS: the newest sample's coordinates are one broadcast read away
    auto k = fps_kernel<THREADS, PPT, true>;
    if (cloud + slots > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(cloud + slots));
      if (e != hipSuccess) { set_error("fps: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
    }
    hipLaunchKernelGGL(k, dim3(B), dim3(THREADS), cloud + slots, s, coords, N, M, distances, indices);
  } else {
    hipLaunchKernelGGL((fps_kernel<THREADS, PPT, false>), dim3(B), dim3(THREADS), slots, s, coords, N, M, distances, indices);
  }
  return check_launch("fps");
}

template <int THREADS, int MAXPPT>   // MAXPPT: 16 or 32 points per thread at most (N <= THREADS * MAXPPT is the caller's business)
static int launch_fps_ppt(const float *coords, int B, int N, int M, float *distances, int32_t *indices, hipStream_t s) {
  const int need = ceil_div(N, THREADS);
  if (need <= 2) return launch_fps<THREADS, 2>(coords, B, N, M, distances, indices, s);
  if (need <= 4) return launch_fps<THREADS, 4>(coords, B, N, M, distances, indices, s);
  if (need <= 8) return launch_fps<THREADS, 8>(coords, B, N, M, distances, indices, s);
  if constexpr (MAXPPT >= 32) {
    if (need <= 16) return launch_fps<THREADS, 16>(coords, B, N, M, distances, indices, s);
    return launch_fps<THREADS, 32>(coords, B, N, M, distances, indices, s);
  } else {
    return launch_fps<THREADS, 16>(coords, B, N, M, distances, indices, s);
  }
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
  // Launch shapes by measurement (tools/fpsbench.py, us per step at B = 8): N = 8192: 1024 x 8: 0.91, 512 x 16: 0.86, 256 x 32: 0.88;
  // N = 1024: 64 x 16 and 256 x 4: 0.50.  With the reductions cheap, a step is bound by its serial part (LDS round trips, the barrier,
  // ~100 scalar / cross-lane instructions at one issue per 4 cycles and wave), not by VALU throughput: the shapes differ by < 8 %.
  if (N <= 1024) return launch_fps_ppt<64, 16>(coords, B, N, M, distances, indices, s);
  if (N <= 8192) return launch_fps_ppt<512, 16>(coords, B, N, M, distances, indices, s);
  return launch_fps_ppt<1024, 16>(coords, B, N, M, distances, indices, s);
}
