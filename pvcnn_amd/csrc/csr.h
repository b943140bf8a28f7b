// csr.h -- deterministic scatter-add on gfx950: counting sort (integer LDS atomics) + owned sums.
//
// Every "scatter" of the hot path -- avg_voxelize fwd, trilinear_devoxelize bwd, grouping /
// gather bwd, 3-NN interpolate bwd -- is   dst[b,c,t] = sum_{e : key(e) = t} w(e) * src[b,c,j(e)].
// The reference implements all of them with fp32 atomicAdd on global memory (vox.cu:68,
// trilinear_devox.cu:150-157, grouping.cu:76, sampling.cu:65, neighbor_interpolate.cu:166-168):
// run-to-run non-deterministic, and on MI355X hopeless -- measured (tools/ubench/lds_rates.hip):
//     global_atomic_add_f32, random addresses : 0.035 lane-ops/clk/CU  (~21 Gop/s per chip)
//     ds_add_f32 (LDS float atomic)           : 0.33  lane-ops/clk/CU  (even conflict-free)
//     ds_add_u32 / ds_read_b32 / ds_write_b32 : >= 4.2 lane-ops/clk/CU
// So float atomics are avoided altogether.  The entries (e -> key, j, w) do not depend on the
// channel, hence ONE pass per cloud sorts them by key with INTEGER LDS atomics:
//   csr_prep_kernel (1 workgroup per cloud, histogram of the L targets in LDS)
//       start[t]  exclusive prefix of the per-target counts (L+1 entries)
//       ent[...]  (j, w) pairs grouped by target, in ASCENDING entry id inside each target
//                 (a stable rank pass; its work is per entry, so a degenerate key
//                 distribution costs O(max_cnt * E / threads), never a serial lane)
// and every (channel, target) sum is then OWNED by one lane:
//   segsum_kernel  (workgroup = G channel rows of one cloud; source rows staged in LDS)
//       acc = acc + w * src[j]  over the target's entries in entry-id order, each target
//       (empty or not) written exactly once with coalesced 16-byte stores.
// Entry ids are chosen so that "ascending entry id" is the oracle's serial loop order
// (point-major, then corner), which makes all of these ops bit-identical to
// oracle/pvcnn_oracle.c, run-to-run deterministic, memset-free and atomic-free on fp32.
#pragma once
#include "slab.h"

namespace pvcnn {

constexpr int kCsrThreads = 1024;
constexpr int kCsrMaxTargets = 38000;   // (L + L/32 + 34) * 4 bytes of histogram must fit 160 KiB

__host__ __device__ __forceinline__ int pad32(int v) { return v + (v >> 5); }          // conflict-free scan layout
__host__ __device__ __forceinline__ int start_stride(int L) { return (L + 1 + 3) & ~3; }   // per-cloud stride of start[]

struct CsrWorkspace {
  int32_t *start;   // (B, start_stride(L))
  int32_t *tmp;     // (B, E)   unordered placement (global; LDS copy used when it fits)
  int2 *ent;        // (B, E)   {source index j, float bits of w}
  static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
  static size_t bytes(int B, int L, long E) {
    return align16((size_t)B * start_stride(L) * 4) + align16((size_t)B * E * 4) + align16((size_t)B * E * 8) + 16;
  }
  void carve(void *ws, int B, int L, long E) {
    char *p = static_cast<char *>(ws);
    start = reinterpret_cast<int32_t *>(p); p += align16((size_t)B * start_stride(L) * 4);
    tmp = reinterpret_cast<int32_t *>(p);   p += align16((size_t)B * E * 4);
    ent = reinterpret_cast<int2 *>(p);
  }
};

// ---------------------------------------------------------------------------------------------
// Entry providers: entry id e in [0,E) of cloud b -> (key, source index, weight).
// ---------------------------------------------------------------------------------------------

// avg_voxelize: one entry per point; key = voxel id (vox.cu:31), weight = 1/cnt[key] (vox.cu:66).
struct VoxelEntries {
  static constexpr bool kInvCountWeight = true;
  const int32_t *coords;   // (B,3,N)
  int32_t *ind;            // (B,N) out
  int N, R, S;
  __device__ __forceinline__ int key_first(int b, int e) const {
    const int32_t *c = coords + (size_t)b * 3 * N;
    int v = c[e] * R * R + c[e + N] * R + c[e + 2 * N];
    v = min(max(v, 0), S - 1);   // reference: unchecked (undefined behaviour when out of range)
    ind[(size_t)b * N + e] = v;
    return v;
  }
  __device__ __forceinline__ int key(int b, int e) const { return ind[(size_t)b * N + e]; }
  __device__ __forceinline__ int src(int b, int e) const { return e; }
  __device__ __forceinline__ float weight(int b, int e) const { return 1.0f; }
};

// (B,NC,J) index / weight planes; entry id = j*NC + k is the reference's (point, corner) loop
// order (trilinear_devox.cu:133-157, neighbor_interpolate.cu:157-169).
template <int NC>
struct TapEntries {
  static constexpr bool kInvCountWeight = false;
  const int32_t *inds;
  const float *wgts;
  int J, L;
  __device__ __forceinline__ int key_first(int b, int e) const { return key(b, e); }
  __device__ __forceinline__ int key(int b, int e) const {
    const int j = e / NC, k = e - j * NC;
    return min(max(inds[((size_t)b * NC + k) * J + j], 0), L - 1);
  }
  __device__ __forceinline__ int src(int b, int e) const { return e / NC; }
  __device__ __forceinline__ float weight(int b, int e) const {
    const int j = e / NC, k = e - j * NC;
    return wgts[((size_t)b * NC + k) * J + j];
  }
};

// plain index list: grouping bwd (E = M*U), gather bwd (E = M); weight 1.
struct IndexEntries {
  static constexpr bool kInvCountWeight = false;
  const int32_t *idx;   // (B,E)
  long E;
  int L;
  __device__ __forceinline__ int key_first(int b, int e) const { return key(b, e); }
  __device__ __forceinline__ int key(int b, int e) const { return min(max(idx[(size_t)b * E + e], 0), L - 1); }
  __device__ __forceinline__ int src(int b, int e) const { return e; }
  __device__ __forceinline__ float weight(int b, int e) const { return 1.0f; }
};

// ---------------------------------------------------------------------------------------------
// csr_prep_kernel: grid = B, block = 1024, dynamic LDS = histogram (+ tmp copy when TMP_LDS).
// ---------------------------------------------------------------------------------------------
template <class EP, bool TMP_LDS>
__global__ __launch_bounds__(kCsrThreads) void csr_prep_kernel(EP ep, int E, int L, int32_t *__restrict__ cnt_out,
                                                              int32_t *__restrict__ start,
                                                              int32_t *__restrict__ tmp_g, int2 *__restrict__ ent) {
  extern __shared__ __attribute__((aligned(16))) int hist[];   // pad32(L)+1 bins, 32 wave totals, [E tmp]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int HP = pad32(L) + 1;
  int *wave_tot = hist + HP;
  int *tmp = TMP_LDS ? (hist + HP + 32) : (tmp_g + (size_t)b * E);
  start += (size_t)b * start_stride(L);
  ent += (size_t)b * E;

  for (int i = tid; i < HP; i += kCsrThreads) hist[i] = 0;
  __syncthreads();
  for (int e = tid; e < E; e += kCsrThreads) atomicAdd(&hist[pad32(ep.key_first(b, e))], 1);   // ds_add_u32
  __syncthreads();
  if (cnt_out)
    for (int t = tid; t < L; t += kCsrThreads) cnt_out[(size_t)b * L + t] = hist[pad32(t)];
  // exclusive scan: thread t owns bins [t*per, t*per + per)
  const int per = (L + kCsrThreads - 1) / kCsrThreads;
  const int t0 = tid * per;
  int local = 0;
  for (int k = 0; k < per; ++k)
    if (t0 + k < L) local += hist[pad32(t0 + k)];
  int incl = local;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if ((tid & (kWave - 1)) >= d) incl += up;
  }
  __syncthreads();   // cnt_out reads of hist are done before bins are rewritten
  if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
  __syncthreads();
  int run = incl - local;
  for (int w = 0; w < (tid >> 6); ++w) run += wave_tot[w];
  for (int k = 0; k < per; ++k)
    if (t0 + k < L) {
      const int c = hist[pad32(t0 + k)];
      hist[pad32(t0 + k)] = run;   // becomes the placement cursor
      run += c;
    }
  __syncthreads();
  for (int t = tid; t < L; t += kCsrThreads) start[t] = hist[pad32(t)];
  if (tid == 0) start[L] = E;
  __syncthreads();
  // unordered placement of entry ids inside each target's segment
  for (int e = tid; e < E; e += kCsrThreads) tmp[atomicAdd(&hist[pad32(ep.key(b, e))], 1)] = e;
  __syncthreads();   // after this hist[pad32(t)] = end of segment t = begin of segment t+1
  // stable rank: position of e among the ids of its segment
  for (int e = tid; e < E; e += kCsrThreads) {
    const int t = ep.key(b, e);
    const int end = hist[pad32(t)];
    const int beg = (t == 0) ? 0 : hist[pad32(t - 1)];
    int rank = 0;
    for (int q = beg; q < end; ++q) rank += (tmp[q] < e) ? 1 : 0;
    const float w = EP::kInvCountWeight ? (float)(1.0 / (double)(float)(end - beg)) : ep.weight(b, e);
    ent[beg + rank] = make_int2(ep.src(b, e), __float_as_int(w));
  }
}

// ---------------------------------------------------------------------------------------------
// segsum_kernel: grid = (ceil(C/G), B).  acc = acc + (w * src): product rounded, then added --
// the reference's atomicAdd(dst, w * g) (no contraction; the library is built with
// -ffp-contract=off).  STAGE: source rows (J floats each) are streamed into LDS first.
// ---------------------------------------------------------------------------------------------
template <int G, int VEC, int THREADS, bool STAGE>
__global__ __launch_bounds__(THREADS) void segsum_kernel(const float *__restrict__ src,
                                                         const int32_t *__restrict__ start,
                                                         const int2 *__restrict__ ent, float *__restrict__ dst,
                                                         int C, int L, int J, int E) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * G;
  const int g = min(G, C - c0);
  const float *rows = src + ((size_t)b * C + c0) * J;
  if (STAGE) {
    slab_copy<THREADS>(lds, rows, g * J);
    __syncthreads();
    rows = lds;
  }
  const int32_t *st = start + (size_t)b * start_stride(L);
  const int2 *en = ent + (size_t)b * E;
  float *out = dst + ((size_t)b * C + c0) * L;
  int roff[G];
#pragma unroll
  for (int c = 0; c < G; ++c) roff[c] = min(c, g - 1) * J;   // rows past g alias a valid row (discarded)
  for (int v0 = threadIdx.x * VEC; v0 < L; v0 += THREADS * VEC) {
    int s[VEC + 1];
    if constexpr (VEC == 4) {
      const int4 s4 = ld4(st + v0);
      s[0] = s4.x; s[1] = s4.y; s[2] = s4.z; s[3] = s4.w; s[4] = st[v0 + 4];
    } else {
      s[0] = st[v0]; s[1] = st[v0 + 1];
    }
    float acc[VEC][G];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
#pragma unroll
      for (int c = 0; c < G; ++c) acc[q][c] = 0.0f;
      for (int e = s[q]; e < s[q + 1]; ++e) {
        const int2 t = en[e];
        const float w = __int_as_float(t.y);
#pragma unroll
        for (int c = 0; c < G; ++c) acc[q][c] = acc[q][c] + w * rows[roff[c] + t.x];
      }
    }
#pragma unroll
    for (int c = 0; c < G; ++c) {
      if (c < g) {
        if constexpr (VEC == 4) st4(out + (size_t)c * L + v0, acc[0][c], acc[1][c], acc[2][c], acc[3][c]);
        else out[(size_t)c * L + v0] = acc[0][c];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Host-side launch.
// ---------------------------------------------------------------------------------------------
inline bool csr_supported(int L, long E) { return L <= kCsrMaxTargets && E <= 0x7fffffffL / 8; }

template <int G, bool STAGE>
int launch_segsum_g(const float *src, const CsrWorkspace &ws, float *dst, int B, int C, int L, int J, int E,
                    bool vec, int threads, hipStream_t s, const char *what) {
  const size_t lds = STAGE ? (size_t)G * J * sizeof(float) : 0;
  const dim3 grid(ceil_div(C, G), B);
#define PVCNN_SEGSUM(VEC, T)                                                                          \
  do {                                                                                                \
    auto k = segsum_kernel<G, VEC, T, STAGE>;                                                         \
    if (int e = enable_big_lds(k, lds)) { set_error("%s: LDS attribute: %d", what, e); return e; }    \
    hipLaunchKernelGGL(k, grid, dim3(T), lds, s, src, ws.start, ws.ent, dst, C, L, J, E);             \
  } while (0)
  if (threads == 1024) { if (vec) PVCNN_SEGSUM(4, 1024); else PVCNN_SEGSUM(1, 1024); }
  else                 { if (vec) PVCNN_SEGSUM(4, 256);  else PVCNN_SEGSUM(1, 256); }
#undef PVCNN_SEGSUM
  return check_launch(what);
}

// dst (B,C,L) = segmented sums of src (B,C,J) over the entries of `ep` (E per cloud).
// cnt_out: optional (B,L) per-target counts (avg_voxelize's `cnt`).
template <class EP>
int launch_csr_scatter(const EP &ep, const float *src, float *dst, int B, int C, int L, int J, long E_,
                       int32_t *cnt_out, void *workspace, size_t workspace_bytes, hipStream_t s, const char *what) {
  const int E = (int)E_;
  if (B == 0) return 0;
  if (!workspace || workspace_bytes < CsrWorkspace::bytes(B, L, E) || !aligned16(workspace)) {
    set_error("%s: workspace missing, misaligned or too small (%zu bytes needed)", what, CsrWorkspace::bytes(B, L, E));
    return PVCNN_ERR_INVALID_ARGUMENT;
  }
  CsrWorkspace ws;
  ws.carve(workspace, B, L, E);
  // 1. per-cloud counting sort of the entries
  const size_t hist_bytes = ((size_t)pad32(L) + 1 + 32) * sizeof(int);
  const bool tmp_lds = hist_bytes + (size_t)E * 4 <= (size_t)kLdsBytesPerCU;
  const size_t prep_lds = hist_bytes + (tmp_lds ? (size_t)E * 4 : 0);
  if (tmp_lds) {
    auto k = csr_prep_kernel<EP, true>;
    if (int e = enable_big_lds(k, prep_lds)) { set_error("%s: LDS attribute: %d", what, e); return e; }
    hipLaunchKernelGGL(k, dim3(B), dim3(kCsrThreads), prep_lds, s, ep, E, L, cnt_out, ws.start, ws.tmp, ws.ent);
  } else {
    auto k = csr_prep_kernel<EP, false>;
    if (int e = enable_big_lds(k, prep_lds)) { set_error("%s: LDS attribute: %d", what, e); return e; }
    hipLaunchKernelGGL(k, dim3(B), dim3(kCsrThreads), prep_lds, s, ep, E, L, cnt_out, ws.start, ws.tmp, ws.ent);
  }
  if (int e = check_launch(what)) return e;
  if (C == 0) return 0;
  // 2. owned segmented sums.  G rows per workgroup: as many as keep the slab <= 64 KiB and the
  //    grid >= 2 workgroups per CU (the entry list is re-read once per workgroup, from L2).
  const size_t row = (size_t)J * sizeof(float);
  const bool stage = row > 0 && row <= (size_t)kLdsBytesPerCU;
  int G = 1;
  if (stage && row <= 64 * 1024) {
    G = (int)((64 * 1024) / row);
    if (G > 8) G = 8;
    while (G > 1 && (long)B * ceil_div(C, G) < 2L * kNumCU) G >>= 1;
    if (G >= 8) G = 8; else if (G >= 4) G = 4; else if (G >= 2) G = 2; else G = 1;
  }
  const bool vec = (L % 4 == 0) && aligned16(dst);
  const int threads = (L >= 8192 || (size_t)G * row > 48 * 1024) ? 1024 : 256;
  if (!stage) return launch_segsum_g<1, false>(src, ws, dst, B, C, L, J, E, vec, threads, s, what);
  switch (G) {
    case 8: return launch_segsum_g<8, true>(src, ws, dst, B, C, L, J, E, vec, threads, s, what);
    case 4: return launch_segsum_g<4, true>(src, ws, dst, B, C, L, J, E, vec, threads, s, what);
    case 2: return launch_segsum_g<2, true>(src, ws, dst, B, C, L, J, E, vec, threads, s, what);
    default: return launch_segsum_g<1, true>(src, ws, dst, B, C, L, J, E, vec, threads, s, what);
  }
}

}  // namespace pvcnn
