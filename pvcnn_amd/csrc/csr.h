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
//   csr_prep_kernel (up to 16 workgroups per cloud, each owning an entry-balanced range of the L targets)
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
#include <algorithm>

#include "slab.h"

namespace pvcnn {

constexpr int kCsrThreads = 1024;
constexpr int kCsrMaxTargets = 1 << 20;   // targets per cloud (split into ranges of <= kCsrMaxRange per workgroup)
constexpr int kCsrMaxRange = 14336;      // (range + range/32 + 69 + 3*8192) * 4 bytes must fit the 160 KiB LDS

__host__ __device__ __forceinline__ int pad32(int v) { return v + (v >> 5); }          // conflict-free scan layout
__host__ __device__ __forceinline__ int start_stride(int L) { return (L + 1 + 3) & ~3; }   // per-cloud stride of start[]

constexpr int kTileTargets = 4096;          // targets per LDS output tile of segsum_tile_kernel (= one `order` range)
constexpr int kTileThreads = 1024;
__host__ __device__ inline int order_stride(int L) { return ceil_div(L, kTileTargets) * kTileTargets; }   // per-cloud stride of order[]

// The PLAN of a scatter: everything the owned sums need that depends only on the entries (keys, sources, weights) --
// not on the feature tensor being scattered, its channel count, or the layer.  A plan built once per (coords, R) is
// applied by every PVConv layer that shares them (PVCNN: three layers at R = 16), forward and backward.
//   start (B, start_stride(L))  exclusive prefix of the per-target entry counts
//   ent   (B, E)                {source index j, float bits of w} grouped by target, ascending entry id inside a target
//   order (B, order_stride(L))  per range of kTileTargets targets: their LOCAL ids sorted by descending entry count
//                               (0xFFFF pads the last range): lanes of a wave take neighbours of this list, so they
//                               walk segments of (nearly) equal length whatever the key distribution
//   seg   (B, order_stride(L))  the [first, end) entry range of the target at the same list position: a lane learns its
//                               target and its segment from two coalesced loads, no dependent start[] lookups
constexpr int kGroupSlack = 256;            // padding a wave group of the interleaved entry copy may cost beyond 2x its entries
__host__ __device__ inline long entw_range_base(int start_t0, int r) { return 2L * start_t0 + (long)r * (kTileTargets / kWave) * kGroupSlack; }
__host__ __device__ inline long entw_stride(int L, long E) { return 2L * E + (long)ceil_div(L, kTileTargets) * (kTileTargets / kWave) * kGroupSlack; }

struct CsrPlan {
  int32_t *start;
  int2 *ent;
  uint16_t *order;
  int2 *seg;        // (B, order_stride(L)): {first entry, end entry} of the target at the same position of `order`
  int32_t *gofs;    // (B, order_stride(L) / 64): offset of a wave group's interleaved entries in entw, or -1 (walk `ent`)
  int2 *entw;       // (B, entw_stride(L, E)): WAVE-INTERLEAVED copy of the entries: the i-th entry of the target at list
                    // position p sits at gofs[p / 64] + i * 64 + p % 64, so the 64 lanes of a wave, each walking its own
                    // segment, read 512 contiguous bytes per step (a per-lane walk of `ent` touches 64 cache lines per
                    // step and was L1-miss bound: measured 3.1 M L2 requests per launch for 8.4 M entry reads)
  static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
  static size_t bytes(int B, int L, long E) {
    return align16((size_t)B * start_stride(L) * 4) + align16((size_t)B * E * 8) + align16((size_t)B * order_stride(L) * 2) +
           align16((size_t)B * order_stride(L) * 8) + align16((size_t)B * (order_stride(L) / kWave) * 4) +
           align16((size_t)B * entw_stride(L, E) * 8) + 16;
  }
  void carve(void *p_, int B, int L, long E) {
    char *p = static_cast<char *>(p_);
    start = reinterpret_cast<int32_t *>(p); p += align16((size_t)B * start_stride(L) * 4);
    ent = reinterpret_cast<int2 *>(p);      p += align16((size_t)B * E * 8);
    order = reinterpret_cast<uint16_t *>(p); p += align16((size_t)B * order_stride(L) * 2);
    seg = reinterpret_cast<int2 *>(p);      p += align16((size_t)B * order_stride(L) * 8);
    gofs = reinterpret_cast<int32_t *>(p);  p += align16((size_t)B * (order_stride(L) / kWave) * 4);
    entw = reinterpret_cast<int2 *>(p);
  }
};
// scratch of the prep step only (overflow path of csr_prep_kernel): tmp (B, E) int32
inline size_t csr_prep_scratch_bytes(int B, long E) { return CsrPlan::align16((size_t)B * E * 4) + 16; }

// combined-call workspace (plan followed by the prep scratch): what the one-shot entry points ask the caller for
struct CsrWorkspace {
  static size_t bytes(int B, int /*C*/, int L, int /*J*/, long E) { return CsrPlan::bytes(B, L, E) + csr_prep_scratch_bytes(B, E); }
};

// ---------------------------------------------------------------------------------------------
// Entry providers: entry id e in [0,E) of cloud b -> (key, source index, weight).
// ---------------------------------------------------------------------------------------------

// Entries are addressed as (plane k, index j) so that sweeps need no integer division and read every
// plane with consecutive lanes on consecutive j (coalesced); the entry id  eid(k,j)  defines the
// stable order inside a target and is chosen to be the reference's serial loop order.

// avg_voxelize: one entry per point; key = voxel id (vox.cu:31), weight = 1/cnt[key] (vox.cu:66).
struct VoxelEntries {
  static constexpr bool kInvCountWeight = true;
  static constexpr int kPlanes = 1;
  const int32_t *coords;   // (B,3,N)
  int32_t *ind;            // (B,N) out
  int N, R, S;
  __device__ __forceinline__ int plane_len() const { return N; }
  __device__ __forceinline__ int eid(int k, int j) const { return j; }
  __device__ __forceinline__ int key(int b, int k, int j) const {
    const int32_t *c = coords + (size_t)b * 3 * N;
    const int v = c[j] * R * R + c[j + N] * R + c[j + 2 * N];
    return min(max(v, 0), S - 1);   // reference: unchecked (undefined behaviour when out of range)
  }
  __device__ __forceinline__ int key_first(int b, int k, int j, bool writer) const {
    const int v = key(b, k, j);
    if (writer) ind[(size_t)b * N + j] = v;
    return v;
  }
  __device__ __forceinline__ int src(int k, int j) const { return j; }
  __device__ __forceinline__ float weight(int b, int k, int j) const { return 1.0f; }
};

// (B,NC,J) index / weight planes; entry id = j*NC + k is the reference's (point, corner) loop
// order (trilinear_devox.cu:133-157, neighbor_interpolate.cu:157-169).
template <int NC>
struct TapEntries {
  static constexpr bool kInvCountWeight = false;
  static constexpr int kPlanes = NC;
  const int32_t *inds;
  const float *wgts;
  int J, L;
  __device__ __forceinline__ int plane_len() const { return J; }
  __device__ __forceinline__ int eid(int k, int j) const { return j * NC + k; }
  __device__ __forceinline__ int key(int b, int k, int j) const { return min(max(inds[((size_t)b * NC + k) * J + j], 0), L - 1); }
  __device__ __forceinline__ int key_first(int b, int k, int j, bool) const { return key(b, k, j); }
  __device__ __forceinline__ int src(int k, int j) const { return j; }
  __device__ __forceinline__ float weight(int b, int k, int j) const { return wgts[((size_t)b * NC + k) * J + j]; }
};

// The devoxelize-backward entries straight from the float grid coordinates (no saved (inds, wgts) planes yet: the plan can be built
// together with the voxelize plan, before any layer has devoxelized).  Keys and weights are the expressions of
// TrilinearFromCoords::pack / unpack (slab.h; trilinear_devox.cu:41-75) -- the bits the forward emits as inds / wgts -- so the plan is
// the one pvcnn_trilinear_devox_bwd_plan builds from those planes.
struct CoordTapEntries {
  static constexpr bool kInvCountWeight = false;
  static constexpr int kPlanes = 8;
  const float *coords;   // (B,3,N) in [0, R-1]
  int N, R, R2, L;
  __device__ __forceinline__ int plane_len() const { return N; }
  __device__ __forceinline__ int eid(int k, int j) const { return j * 8 + k; }
  __device__ __forceinline__ int key(int b, int k, int j) const {
    const float *c = coords + (size_t)b * 3 * N;
    const float x = c[j], y = c[j + N], z = c[j + 2 * N];
    const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
    const float xd1 = x - xl, yd1 = y - yl, zd1 = z - zl;
    int v = (int)xl * R2 + (int)yl * R + (int)zl;
    if ((k & 4) && xd1 > 0) v += R2;
    if ((k & 2) && yd1 > 0) v += R;
    if ((k & 1) && zd1 > 0) v += 1;
    return min(max(v, 0), L - 1);
  }
  __device__ __forceinline__ int key_first(int b, int k, int j, bool) const { return key(b, k, j); }
  __device__ __forceinline__ int src(int k, int j) const { return j; }
  __device__ __forceinline__ float weight(int b, int k, int j) const {
    const float *c = coords + (size_t)b * 3 * N;
    const float x = c[j], y = c[j + N], z = c[j + 2 * N];
    const float xd1 = x - floorf(x), yd1 = y - floorf(y), zd1 = z - floorf(z);
    const float xd = (k & 4) ? xd1 : 1.0f - xd1, yd = (k & 2) ? yd1 : 1.0f - yd1, zd = (k & 1) ? zd1 : 1.0f - zd1;
    return (xd * yd) * zd;                                   // w_ab * zd_c with w_ab = xd_a * yd_b, as unpack() rounds it
  }
};

// plain index list: grouping bwd (E = M*U), gather bwd (E = M); weight 1.
struct IndexEntries {
  static constexpr bool kInvCountWeight = false;
  static constexpr int kPlanes = 1;
  const int32_t *idx;   // (B,E)
  long E;
  int L;
  __device__ __forceinline__ int plane_len() const { return (int)E; }
  __device__ __forceinline__ int eid(int k, int j) const { return j; }
  __device__ __forceinline__ int key(int b, int k, int j) const { return min(max(idx[(size_t)b * E + j], 0), L - 1); }
  __device__ __forceinline__ int key_first(int b, int k, int j, bool) const { return key(b, k, j); }
  __device__ __forceinline__ int src(int k, int j) const { return j; }
  __device__ __forceinline__ float weight(int b, int k, int j) const { return 1.0f; }
};

// ---------------------------------------------------------------------------------------------
// csr_prep_kernel: grid = (P, B), block = 1024.
// A cloud's sort is instruction-bound on one CU (~250 instructions per entry), so the TARGETS of each
// cloud are split into up to P contiguous ranges, one workgroup each.  Real clouds are anything but
// uniform (walls and floors put a third of the points into one slice of the grid), so the ranges are
// balanced by ENTRIES, not by targets: every workgroup first builds the same coarse histogram (256
// buckets of 2^BS targets, LDS integer atomics), prefix-sums it, and derives the same range table --
// a new range starts where floor(entries_before / Q) changes, and at every BPM-th bucket so that a
// range's fine histogram fits LDS.  Workgroup p takes range p (or exits if there are fewer ranges);
// the global offset of its range (entries with a smaller key) falls out of the same prefix sum, so the
// P workgroups never communicate.  Every workgroup sweeps all E keys twice (cheap, L2-resident) but
// histograms / places / ranks only the entries of its own range.
// ---------------------------------------------------------------------------------------------
constexpr int kCsrList = 8192;     // in-range entries a workgroup can hold in LDS (list + placement buffer)
constexpr int kCsrCoarse = 256;    // coarse buckets

struct CsrSplit {
  int BS;    // log2 of the bucket width (targets)
  int BPM;   // max buckets per range (fine histogram must fit LDS)
  int Q;     // entries per range aimed at
  int P;     // workgroups per cloud (>= number of ranges that can arise)
  int HP;    // LDS ints reserved for the fine histogram
};

inline CsrSplit csr_split(int B, int L, int E, int wg_cap = 16) {
  CsrSplit sp{};
  sp.BS = 0;
  while (((long)kCsrCoarse << sp.BS) < L) ++sp.BS;
  const int BW = 1 << sp.BS;
  sp.BPM = std::max(1, std::min(kCsrCoarse, kCsrMaxRange / BW));
  const int forced = ceil_div(ceil_div(L, BW), sp.BPM);          // ranges forced by the LDS bound alone
  // workgroups per cloud: the launch should fit the chip in one round (a workgroup takes most of a CU's LDS)
  int PT = std::max(1, std::min(wg_cap, kNumCU / std::max(B, 1)));
  PT = std::min(PT, std::max(1, L / 64));
  PT = std::max(PT, forced);
  const int P0 = PT - forced + 1;                                // entry-balanced ranges on top of the forced cuts
  sp.Q = std::max(1, ceil_div(std::max(E, 1), P0));
  sp.P = PT;                                                      // ranges <= P0 + forced - 1 (bucket 0 is both)
  sp.HP = pad32(std::min(L, sp.BPM * BW)) + 1;
  return sp;
}

template <class EP>
__device__ __forceinline__ void csr_prep_body(const EP &ep, int E, int L, const CsrSplit &sp, int32_t *__restrict__ cnt_out,
                                              int32_t *__restrict__ start, int32_t *__restrict__ tmp_g, int2 *__restrict__ ent,
                                              int part, int b, int *csr_lds) {
  // LDS: coarse[256] | cpre[257] (+3 pad) | 32 wave totals | 4 range words | list_kj[kCsrList] |
  //      list_key[kCsrList] | tmp[kCsrList] | fine histogram (pad32 layout, sp.HP ints)
  const int tid = threadIdx.x;
  int *coarse = csr_lds;
  int *cpre = coarse + kCsrCoarse;               // exclusive prefix, cpre[256] = E
  int *wave_tot = cpre + kCsrCoarse + 4;
  int *rng = wave_tot + 32;                      // [0] first bucket, [1] end bucket, [2] list counter
  int *list_n = rng + 2;
  int *list_kj = rng + 4;                        // packed (k, j): k * PL + j  (fits: E < 2^31)
  int *list_key = list_kj + kCsrList;
  int *tmp_l = list_key + kCsrList;
  int *hist = tmp_l + kCsrList;
  start += (size_t)b * start_stride(L);
  ent += (size_t)b * E;
  const bool writer = (part == 0);   // side outputs of key_first (avg_voxelize's `ind`) are written once
  constexpr int kJB = (16 / EP::kPlanes) > 0 ? (16 / EP::kPlanes) : 1;   // elements per thread and sweep step
  const int PL = ep.plane_len();

  // ---- sweep 1: coarse histogram of ALL entries ----
  if (tid < kCsrCoarse) coarse[tid] = 0;
  if (tid == 0) { rng[0] = -1; rng[1] = kCsrCoarse; *list_n = 0; }
  __syncthreads();
  // (both sweeps keep kJB * kPlanes independent key loads in flight per thread: a sweep is a handful of
  //  L2 round trips, not one per plane)
  for (int j0 = tid; j0 < PL; j0 += kCsrThreads * kJB) {
    int key[kJB][EP::kPlanes];
#pragma unroll
    for (int u = 0; u < kJB; ++u)
#pragma unroll
      for (int k = 0; k < EP::kPlanes; ++k) {
        const int j = j0 + u * kCsrThreads;
        key[u][k] = (j < PL) ? ep.key_first(b, k, j, writer) : -1;
      }
#pragma unroll
    for (int u = 0; u < kJB; ++u)
#pragma unroll
      for (int k = 0; k < EP::kPlanes; ++k)
        if (key[u][k] >= 0) atomicAdd(&coarse[key[u][k] >> sp.BS], 1);   // ds_add_u32
  }
  __syncthreads();
  // ---- range table (wave 0): prefix sum of the buckets, range id per bucket ----
  if (tid < kWave) {
    int c[4], run = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { c[u] = coarse[tid * 4 + u]; run += c[u]; }
    int incl = run;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (tid >= d) incl += up;
    }
    int ex = incl - run;                          // entries before bucket 4*tid
    const int prev_last = __shfl_up(ex + run - c[3], 1);   // entries before bucket 4*tid - 1
    const int emax = max(E - 1, 0);
    int flag[4], nflag = 0, before_prev = prev_last;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid * 4 + u;
      cpre[i] = ex;
      // entries-before is clamped to E-1 so that the trailing (entry-free) buckets stay in the last range
      flag[u] = (i == 0) || (min(ex, emax) / sp.Q != min(before_prev, emax) / sp.Q) || (i % sp.BPM == 0);
      nflag += flag[u];
      before_prev = ex;
      ex += c[u];
    }
    if (tid == kWave - 1) cpre[kCsrCoarse] = ex;
    int fincl = nflag;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int up = __shfl_up(fincl, d);
      if (tid >= d) fincl += up;
    }
    int rid = fincl - nflag - 1;                  // range id of the bucket before 4*tid
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      rid += flag[u];
      if (flag[u] && rid == part) rng[0] = tid * 4 + u;
      if (flag[u] && rid == part + 1) rng[1] = tid * 4 + u;
    }
  }
  __syncthreads();
  const int lo_b = rng[0];
  if (lo_b < 0) return;                           // fewer ranges than workgroups
  const int lo = lo_b << sp.BS;
  if (lo >= L) return;                            // buckets beyond the last target
  const int hi = min(L, rng[1] << sp.BS), nb = hi - lo;
  const int base = cpre[lo_b];                    // entries with a key below this range
  const int HP = pad32(nb) + 1;

  for (int i = tid; i < HP; i += kCsrThreads) hist[i] = 0;
  __syncthreads();
  // ---- sweep 2: fine histogram + compaction of the in-range entries ----
  for (int j0 = tid; j0 < PL; j0 += kCsrThreads * kJB) {
    int key[kJB][EP::kPlanes];
#pragma unroll
    for (int u = 0; u < kJB; ++u)
#pragma unroll
      for (int k = 0; k < EP::kPlanes; ++k) {
        const int j = j0 + u * kCsrThreads;
        key[u][k] = (j < PL) ? ep.key(b, k, j) : 0x7fffffff;
      }
#pragma unroll
    for (int u = 0; u < kJB; ++u)
#pragma unroll
      for (int k = 0; k < EP::kPlanes; ++k) {
        const int kv = key[u][k];
        if (kv >= lo && kv < hi) {
          atomicAdd(&hist[pad32(kv - lo)], 1);   // ds_add_u32
          const int slot = atomicAdd(list_n, 1);
          if (slot < kCsrList) { list_kj[slot] = k * PL + (j0 + u * kCsrThreads); list_key[slot] = kv; }
        }
      }
  }
  __syncthreads();
  const int n_in = *list_n;
  if (cnt_out)
    for (int t = tid; t < nb; t += kCsrThreads) cnt_out[(size_t)b * L + lo + t] = hist[pad32(t)];
  // ---- exclusive scan of this range's bins: thread t owns bins [t*per, t*per + per) ----
  const int per = (nb + kCsrThreads - 1) / kCsrThreads;
  const int t0 = tid * per;
  int local = 0;
  for (int k = 0; k < per; ++k)
    if (t0 + k < nb) local += hist[pad32(t0 + k)];
  int incl = local;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if ((tid & (kWave - 1)) >= d) incl += up;
  }
  __syncthreads();   // cnt_out reads of hist are done before bins are rewritten
  if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
  __syncthreads();
  int run = base + incl - local;
  for (int w = 0; w < (tid >> 6); ++w) run += wave_tot[w];
  for (int k = 0; k < per; ++k)
    if (t0 + k < nb) {
      const int c = hist[pad32(t0 + k)];
      hist[pad32(t0 + k)] = run;   // becomes the placement cursor (global positions)
      run += c;
    }
  __syncthreads();
  for (int t = tid; t < nb; t += kCsrThreads) start[lo + t] = hist[pad32(t)];
  if (hi == L && tid == 0) start[L] = E;
  __syncthreads();

  if (n_in <= kCsrList) {
    // ---- dense heavy passes over the compacted list: every lane has an in-range entry ----
    for (int i = tid; i < n_in; i += kCsrThreads) {
      const int kj = list_kj[i];
      const int k = (EP::kPlanes == 1) ? 0 : kj / PL, j = (EP::kPlanes == 1) ? kj : kj - (kj / PL) * PL;
      tmp_l[atomicAdd(&hist[pad32(list_key[i] - lo)], 1) - base] = ep.eid(k, j);
    }
    __syncthreads();   // hist[pad32(t)] = end of segment lo+t = begin of segment lo+t+1
    for (int i = tid; i < n_in; i += kCsrThreads) {
      const int kj = list_kj[i];
      const int k = (EP::kPlanes == 1) ? 0 : kj / PL, j = (EP::kPlanes == 1) ? kj : kj - (kj / PL) * PL;
      const int e = ep.eid(k, j);
      const int t = list_key[i] - lo;
      const int end = hist[pad32(t)];
      const int beg = (t == 0) ? base : hist[pad32(t - 1)];
      int rank = 0;
      for (int q = beg; q < end; ++q) rank += (tmp_l[q - base] < e) ? 1 : 0;
      const float w = EP::kInvCountWeight ? (float)(1.0 / (double)(float)(end - beg)) : ep.weight(b, k, j);
      ent[beg + rank] = make_int2(ep.src(k, j), __float_as_int(w));
    }
    return;
  }
  // ---- overflow path (a range holding more than kCsrList entries: degenerate key distribution):
  //      same algorithm by re-sweeping all entries, placement buffer in global memory ----
  int *tmp = tmp_g + (size_t)b * E + base;
  for (int k = 0; k < EP::kPlanes; ++k)
    for (int j = tid; j < PL; j += kCsrThreads) {
      const int key = ep.key(b, k, j);
      if (key >= lo && key < hi) tmp[atomicAdd(&hist[pad32(key - lo)], 1) - base] = ep.eid(k, j);
    }
  __syncthreads();
  for (int k = 0; k < EP::kPlanes; ++k)
    for (int j = tid; j < PL; j += kCsrThreads) {
      const int key = ep.key(b, k, j);
      if (key < lo || key >= hi) continue;
      const int e = ep.eid(k, j);
      const int t = key - lo;
      const int end = hist[pad32(t)];
      const int beg = (t == 0) ? base : hist[pad32(t - 1)];
      int rank = 0;
      for (int q = beg; q < end; ++q) rank += (tmp[q - base] < e) ? 1 : 0;
      const float w = EP::kInvCountWeight ? (float)(1.0 / (double)(float)(end - beg)) : ep.weight(b, k, j);
      ent[beg + rank] = make_int2(ep.src(k, j), __float_as_int(w));
    }
}

template <class EP>
__global__ __launch_bounds__(kCsrThreads) void csr_prep_kernel(EP ep, int E, int L, CsrSplit sp, int32_t *__restrict__ cnt_out,
                                                              int32_t *__restrict__ start,
                                                              int32_t *__restrict__ tmp_g, int2 *__restrict__ ent) {
  extern __shared__ __attribute__((aligned(16))) int csr_lds[];
  csr_prep_body(ep, E, L, sp, cnt_out, start, tmp_g, ent, (int)blockIdx.x, (int)blockIdx.y, csr_lds);
}

// TWO sorts of the same clouds in one launch (a PVConv geometry: the voxelize plan and the devoxelize-backward plan of one
// (coords, R)): workgroups [0, P1) of a cloud run the first, [P1, P1 + P2) the second.  The two chains are latency chains of ~8
// barrier phases each on L2-resident data and never fill the chip alone; side by side they cost the longer of the two.
template <class EP1, class EP2>
__global__ __launch_bounds__(kCsrThreads) void csr_prep_pair_kernel(EP1 ep1, int E1, CsrSplit sp1, int32_t *__restrict__ cnt1,
                                                                   int32_t *__restrict__ start1, int32_t *__restrict__ tmp1,
                                                                   int2 *__restrict__ ent1, EP2 ep2, int E2, CsrSplit sp2,
                                                                   int32_t *__restrict__ start2, int32_t *__restrict__ tmp2,
                                                                   int2 *__restrict__ ent2, int L) {
  extern __shared__ __attribute__((aligned(16))) int csr_lds[];
  const int p = (int)blockIdx.x;                             // (uniform per workgroup)
  if (p < sp1.P) csr_prep_body(ep1, E1, L, sp1, cnt1, start1, tmp1, ent1, p, (int)blockIdx.y, csr_lds);
  else           csr_prep_body(ep2, E2, L, sp2, static_cast<int32_t *>(nullptr), start2, tmp2, ent2, p - sp1.P, (int)blockIdx.y, csr_lds);
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
                                                         int C, int L, int J, int E, long src_bstride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * G;
  const int g = min(G, C - c0);
  const float *rows = src + (size_t)b * src_bstride + (size_t)c0 * J;   // src: rows of one cloud contiguous, clouds src_bstride apart
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
      int e = s[q];
      for (; e + 4 <= s[q + 1]; e += 4) {   // 4 entries' loads in flight; the adds stay in entry order
        int2 t[4];
        float xv[4][G];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = en[e + u];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int c = 0; c < G; ++c) xv[u][c] = rows[roff[c] + t[u].x];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float w = __int_as_float(t[u].y);
#pragma unroll
          for (int c = 0; c < G; ++c) acc[q][c] = acc[q][c] + w * xv[u][c];
        }
      }
      for (; e < s[q + 1]; ++e) {
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
// csr_order_kernel: grid = (ranges, B).  (1) Counting sort of one range of kTileTargets targets by min(count, 255),
// descending -> order / seg.  (2) Per wave group (64 consecutive list positions = the 64 lanes of one wave of
// segsum_tile_kernel): if padding every lane's segment to the group's longest costs at most 2x the group's entries
// + kGroupSlack, the group's entries are copied into the interleaved layout entw (see CsrPlan); else gofs = -1 and the
// lanes walk `ent` directly (the handful of groups at the head of a degenerate distribution).
// A few microseconds; part of the plan (built once, reused by every apply and every channel slab).
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ void csr_order_body(const int32_t *__restrict__ start, const int2 *__restrict__ ent, int L, int E,
                                                      uint16_t *__restrict__ order, int2 *__restrict__ seg,
                                                      int32_t *__restrict__ gofs, int2 *__restrict__ entw, int r, int b) {
  constexpr int kSlots = kTileTargets / kTileThreads, kGroups = kTileTargets / kWave;
  __shared__ int hist[256];
  __shared__ int wtot[4];
  __shared__ uint16_t l_lt[kTileTargets];
  __shared__ int2 l_seg[kTileTargets];
  __shared__ int gsz[kGroups];
  const int tid = threadIdx.x, lane = tid & 63;
  const int t0 = r * kTileTargets, nt = min(kTileTargets, L - t0);
  const int32_t *st = start + (size_t)b * start_stride(L) + t0;
  uint16_t *out = order + (size_t)b * order_stride(L) + t0;
  int2 *sout = seg + (size_t)b * order_stride(L) + t0;
  if (tid < 256) hist[tid] = 0;
  __syncthreads();
  int bucket[kSlots], e0[kSlots], e1[kSlots];
#pragma unroll
  for (int k = 0; k < kSlots; ++k) {
    const int t = tid + k * kTileThreads;
    bucket[k] = -1;
    if (t < nt) {
      e0[k] = st[t]; e1[k] = st[t + 1];
      bucket[k] = 255 - min(e1[k] - e0[k], 255);          // bucket 0 = the longest segments
      atomicAdd(&hist[bucket[k]], 1);
    }
  }
  __syncthreads();
  int v = 0, incl = 0;
  if (tid < 256) {                                        // exclusive prefix over the 256 buckets (waves 0..3)
    v = hist[tid];
    incl = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    if (lane == 63) wtot[tid >> 6] = incl;
  }
  __syncthreads();
  if (tid < 256) {
    int base = incl - v;
    for (int w = 0; w < (tid >> 6); ++w) base += wtot[w];
    hist[tid] = base;
  }
  for (int t = nt + tid; t < kTileTargets; t += kTileThreads) { l_lt[t] = 0xFFFF; l_seg[t] = make_int2(0, 0); }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kSlots; ++k)
    if (bucket[k] >= 0) {
      const int pos = atomicAdd(&hist[bucket[k]], 1);
      l_lt[pos] = (uint16_t)(tid + k * kTileThreads);
      l_seg[pos] = make_int2(e0[k], e1[k]);
    }
  __syncthreads();
  // ---- the sorted list leaves coalesced; per wave group: longest segment and total ----
  int2 sg[kSlots];
#pragma unroll
  for (int k = 0; k < kSlots; ++k) {
    const int pos = k * kTileThreads + tid;
    sg[k] = l_seg[pos];
    out[pos] = l_lt[pos];
    sout[pos] = sg[k];
    int m = sg[k].y - sg[k].x, sum = m;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { m = max(m, __shfl_xor(m, d)); sum += __shfl_xor(sum, d); }
    if (lane == 0) gsz[pos >> 6] = (m > 0 && 64L * m <= 2L * sum + kGroupSlack) ? 64 * m : 0;
  }
  __syncthreads();
  if (tid < kGroups) {                                    // exclusive prefix over the 64 groups (wave 0)
    const int sz = gsz[tid];
    int inc = sz;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int up = __shfl_up(inc, d);
      if (lane >= d) inc += up;
    }
    const long base = entw_range_base(st[0], r) + (inc - sz);
    const int ofs = sz > 0 ? (int)base : -1;
    gsz[tid] = ofs;
    gofs[(size_t)b * (order_stride(L) / kWave) + (size_t)r * kGroups + tid] = ofs;
  }
}

static __global__ __launch_bounds__(kTileThreads) void csr_order_kernel(const int32_t *__restrict__ start, const int2 *__restrict__ ent,
                                                                        int L, int E, uint16_t *__restrict__ order,
                                                                        int2 *__restrict__ seg, int32_t *__restrict__ gofs,
                                                                        int2 *__restrict__ entw) {
  csr_order_body(start, ent, L, E, order, seg, gofs, entw, (int)blockIdx.x, (int)blockIdx.y);
}

// the order / interleave steps of TWO plans over the same L targets in one launch each (see csr_prep_pair_kernel)
struct CsrPlanRef {
  const int32_t *start;
  const int2 *ent;
  uint16_t *order;
  int2 *seg;
  int32_t *gofs;
  int2 *entw;
  int E;
};

[[maybe_unused]] static __global__ __launch_bounds__(kTileThreads) void csr_order_pair_kernel(CsrPlanRef a, CsrPlanRef c, int L, int nr) {   // grid (2 nr, B)
  const int r = (int)blockIdx.x;
  const bool f = r < nr;                                     // (uniform; ONE inlined body: its LDS arrays exist once)
  csr_order_body(f ? a.start : c.start, f ? a.ent : c.ent, L, f ? a.E : c.E, f ? a.order : c.order, f ? a.seg : c.seg,
                 f ? a.gofs : c.gofs, f ? a.entw : c.entw, f ? r : r - nr, (int)blockIdx.y);
}

// one wave per wave group copies the group's entries into the interleaved layout (reads: each lane its own segment, once per
// plan; writes: 512 contiguous bytes per step).
static __device__ __forceinline__ void csr_interleave_body(const int2 *__restrict__ ent, const int2 *__restrict__ seg,
                                                           const int32_t *__restrict__ gofs, int L, int E, int2 *__restrict__ entw,
                                                           int blk, int b) {
  const int pos = blk * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const int ofs = gofs[(size_t)b * (order_stride(L) / kWave) + (pos >> 6)];
  if (ofs < 0) return;
  const int2 sg = seg[(size_t)b * order_stride(L) + pos];
  const int2 *en = ent + (size_t)b * E + sg.x;
  int2 *ew = entw + (size_t)b * entw_stride(L, E) + ofs + lane;
  const int n = sg.y - sg.x;
  int i = 0;
  for (; i + 4 <= n; i += 4) {
    const int2 a = en[i], c = en[i + 1], d = en[i + 2], f = en[i + 3];
    ew[(size_t)i * 64] = a; ew[(size_t)(i + 1) * 64] = c; ew[(size_t)(i + 2) * 64] = d; ew[(size_t)(i + 3) * 64] = f;
  }
  for (; i < n; ++i) ew[(size_t)i * 64] = en[i];
}

// grid = (order_stride(L) / 256, B), 256 threads
static __global__ __launch_bounds__(256) void csr_interleave_kernel(const int2 *__restrict__ ent, const int2 *__restrict__ seg,
                                                                    const int32_t *__restrict__ gofs, int L, int E,
                                                                    int2 *__restrict__ entw) {
  csr_interleave_body(ent, seg, gofs, L, E, entw, (int)blockIdx.x, (int)blockIdx.y);
}

[[maybe_unused]] static __global__ __launch_bounds__(256) void csr_interleave_pair_kernel(CsrPlanRef a, CsrPlanRef c, int L, int nblk) {   // grid (2 nblk, B)
  const int k = (int)blockIdx.x;
  const bool f = k < nblk;
  csr_interleave_body(f ? a.ent : c.ent, f ? a.seg : c.seg, f ? a.gofs : c.gofs, L, f ? a.E : c.E, f ? a.entw : c.entw, f ? k : k - nblk,
                      (int)blockIdx.y);
}

// ---------------------------------------------------------------------------------------------
// segsum_tile_kernel: the owned sums, for any key distribution.  grid = (ceil(C/G), nsplit, B), 1024 threads.
//   * the workgroup's G source rows (J floats each) are staged ONCE into LDS, channel-interleaved
//     (srcI[j*G + c]): one ds_read of G floats serves all G channels of an entry;
//   * targets are processed in ranges of kTileTargets; inside a range, thread t takes the targets at ranks
//     t, t+1024, ... of the range's count-sorted `order` list, so the 64 lanes of a wave walk segments of nearly equal
//     length (a lane-per-consecutive-target walk idles most lanes: real clouds fill a fifth of the grid, and the
//     occupied voxels own ~45 corner entries each at R = 16);
//   * each target's sum is acc = acc + w * src[j] over its entries in ascending entry id (the oracle's serial order:
//     bit-exact), and lands in an LDS tile [G][range]; the tile leaves as coalesced 16-byte stores -- every target,
//     empty or not, is written exactly once: no memset, no float atomics, no transposed copy of the source.
//   nsplit > 1 spreads the ranges of one (cloud, channel slab) over several workgroups (few channels: voxelize C = 9).
// ---------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(kTileThreads) void segsum_tile_kernel(const float *__restrict__ src, const int2 *__restrict__ seg,
                                                                   const int2 *__restrict__ ent, const uint16_t *__restrict__ order,
                                                                   const int32_t *__restrict__ gofs, const int2 *__restrict__ entw,
                                                                   float *__restrict__ dst, int C, int L, int J, int E,
                                                                   long src_bstride, int nsplit, int JP) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using vecG = __attribute__((ext_vector_type(G))) float;
  const int b = blockIdx.z, split = blockIdx.y, c0 = blockIdx.x * G, tid = threadIdx.x;
  const int g = min(G, C - c0);
  float *srcI = lds;                    // JP * G floats (JP = J rounded up to 4)
  float *tile = lds + (size_t)JP * G;   // G rows of kTileTargets floats
  const float *rows = src + (size_t)b * src_bstride + (size_t)c0 * J;
  // ---- stage the source rows, transposing to channel-interleaved ----
  if ((J & 3) == 0 && aligned16(rows) && ((src_bstride & 3) == 0)) {
    for (int j0 = tid * 4; j0 < J; j0 += kTileThreads * 4) {
      float4 v[G];
#pragma unroll
      for (int c = 0; c < G; ++c) v[c] = ld4(rows + (size_t)min(c, g - 1) * J + j0);   // rows past g alias a valid row (discarded)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        vecG o;
#pragma unroll
        for (int c = 0; c < G; ++c) o[c] = (u == 0) ? v[c].x : (u == 1) ? v[c].y : (u == 2) ? v[c].z : v[c].w;
        *reinterpret_cast<vecG *>(srcI + (size_t)(j0 + u) * G) = o;
      }
    }
  } else {
    for (int j = tid; j < J; j += kTileThreads) {
      vecG o;
#pragma unroll
      for (int c = 0; c < G; ++c) o[c] = rows[(size_t)min(c, g - 1) * J + j];
      *reinterpret_cast<vecG *>(srcI + (size_t)j * G) = o;
    }
  }
  __syncthreads();
  const int2 *en = ent + (size_t)b * E;
  const int2 *ew = entw + (size_t)b * entw_stride(L, E) + (tid & 63);
  const uint16_t *ord = order + (size_t)b * order_stride(L);
  const int2 *sg = seg + (size_t)b * order_stride(L);
  const int32_t *go = gofs + (size_t)b * (order_stride(L) / kWave);
  const int nr = ceil_div(L, kTileTargets);
  constexpr int kSlots = kTileTargets / kTileThreads;
  // this thread's list positions of a range: target id, segment, and where its wave group's entries are (prefetched one
  // range ahead)
  int lt[kSlots], gb[kSlots];
  int2 sq[kSlots];
  auto fetch = [&](int r) {
#pragma unroll
    for (int k = 0; k < kSlots; ++k) {
      const int pos = r * kTileTargets + k * kTileThreads + tid;
      lt[k] = ord[pos];
      sq[k] = sg[pos];
      gb[k] = go[pos >> 6];
    }
  };
  if (split < nr) fetch(split);
  for (int r = split; r < nr; r += nsplit) {
    const int t0 = r * kTileTargets, nt = min(kTileTargets, L - t0);
    int clt[kSlots], cgb[kSlots];
    int2 csq[kSlots];
#pragma unroll
    for (int k = 0; k < kSlots; ++k) { clt[k] = lt[k]; csq[k] = sq[k]; cgb[k] = gb[k]; }
    if (r + nsplit < nr) fetch(r + nsplit);          // next range's list entries are in flight while this one is summed
#pragma unroll
    for (int k = 0; k < kSlots; ++k) {
      if (clt[k] == 0xFFFF) continue;
      const int n = csq[k].y - csq[k].x;
      // entry i of this lane: interleaved copy (coalesced across the wave) or, for a group that was not copied, `ent`
      const int2 *ep = cgb[k] >= 0 ? ew + cgb[k] : en + csq[k].x;
      const int estep = cgb[k] >= 0 ? 64 : 1;
      vecG acc;
#pragma unroll
      for (int c = 0; c < G; ++c) acc[c] = 0.0f;
      int i = 0;
      for (; i + 8 <= n; i += 8) {       // 8 entries' loads in flight; the adds stay in entry order
        int2 t[8];
        vecG x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = ep[(size_t)(i + u) * estep];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const vecG *>(srcI + (size_t)t[u].x * G);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float w = __int_as_float(t[u].y);
#pragma unroll
          for (int c = 0; c < G; ++c) acc[c] = acc[c] + w * x[u][c];
        }
      }
      // tail: 4, 2, 1 -- exactly n loads in total (most targets of a sparse scatter own one or two entries)
#define PVCNN_TAIL(U)                                                                              \
      if ((n - i) & U) {                                                                           \
        int2 t[U];                                                                                 \
        vecG x[U];                                                                                 \
        _Pragma("unroll") for (int u = 0; u < U; ++u) t[u] = ep[(size_t)(i + u) * estep];          \
        _Pragma("unroll") for (int u = 0; u < U; ++u) x[u] = *reinterpret_cast<const vecG *>(srcI + (size_t)t[u].x * G); \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                            \
          const float w = __int_as_float(t[u].y);                                                  \
          _Pragma("unroll") for (int c = 0; c < G; ++c) acc[c] = acc[c] + w * x[u][c];             \
        }                                                                                          \
        i += U;                                                                                    \
      }
      PVCNN_TAIL(4)
      PVCNN_TAIL(2)
      PVCNN_TAIL(1)
#undef PVCNN_TAIL
#pragma unroll
      for (int c = 0; c < G; ++c) tile[c * kTileTargets + clt[k]] = acc[c];
    }
    lds_barrier();
    float *out = dst + ((size_t)b * C + c0) * L + t0;
    if ((L & 3) == 0 && aligned16(dst)) {
      const int q_per_row = nt >> 2;      // nt % 4 == 0 since L % 4 == 0 and kTileTargets % 4 == 0
      for (int q = tid; q < g * q_per_row; q += kTileThreads) {
        const int c = q / q_per_row, i = (q - c * q_per_row) * 4;
        const float4 v = *reinterpret_cast<const float4 *>(tile + c * kTileTargets + i);
        using v4f = __attribute__((ext_vector_type(4))) float;
        v4f o = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(out + (size_t)c * L + i));
      }
    } else {
      for (int q = tid; q < g * nt; q += kTileThreads) {
        const int c = q / nt, i = q - c * nt;
        out[(size_t)c * L + i] = tile[c * kTileTargets + i];
      }
    }
    lds_barrier();     // the tile is rewritten by the next range (LDS-only barrier: the stores above keep draining)
  }
}

// ---------------------------------------------------------------------------------------------
// Host-side launch: plan (prep) and apply are separate steps so that a plan can be reused.
// ---------------------------------------------------------------------------------------------
inline bool csr_supported(int L, long E) { return L <= kCsrMaxTargets && E <= 0x7fffffffL / 8; }

template <int G, bool STAGE>
int launch_segsum_g(const float *src, const CsrPlan &ws, float *dst, int B, int C, int L, int J, int E,
                    bool vec, int threads, hipStream_t s, const char *what, long src_bstride) {
  const size_t lds = STAGE ? (size_t)G * J * sizeof(float) : 0;
  const dim3 grid(ceil_div(C, G), B);
#define PVCNN_SEGSUM(VEC, T)                                                                          \
  do {                                                                                                \
    auto k = segsum_kernel<G, VEC, T, STAGE>;                                                         \
    if (int e = enable_big_lds(k, lds)) { set_error("%s: LDS attribute: %d", what, e); return e; }    \
    hipLaunchKernelGGL(k, grid, dim3(T), lds, s, src, ws.start, ws.ent, dst, C, L, J, E, src_bstride); \
  } while (0)
  if (threads == 1024) { if (vec) PVCNN_SEGSUM(4, 1024); else PVCNN_SEGSUM(1, 1024); }
  else                 { if (vec) PVCNN_SEGSUM(4, 256);  else PVCNN_SEGSUM(1, 256); }
#undef PVCNN_SEGSUM
  return check_launch(what);
}

// Step 1: build the plan of the scatter described by `ep` (E entries per cloud onto L targets).
// cnt_out: optional (B,L) per-target counts (avg_voxelize's `cnt`).  scratch: csr_prep_scratch_bytes(B, E).
template <class EP>
int launch_csr_prep(const EP &ep, int B, int L, long E_, int32_t *cnt_out, void *plan, size_t plan_bytes, void *scratch,
                    size_t scratch_bytes, hipStream_t s, const char *what) {
  const int E = (int)E_;
  if (B == 0) return 0;
  if (!plan || plan_bytes < CsrPlan::bytes(B, L, E) || !aligned16(plan) || !scratch || !aligned16(scratch) ||
      scratch_bytes < csr_prep_scratch_bytes(B, E)) {
    set_error("%s: plan / scratch missing, misaligned or too small (%zu + %zu bytes needed)", what, CsrPlan::bytes(B, L, E),
              csr_prep_scratch_bytes(B, E));
    return PVCNN_ERR_INVALID_ARGUMENT;
  }
  CsrPlan pl;
  pl.carve(plan, B, L, E);
  // per-cloud counting sort of the entries, targets split into entry-balanced ranges (one workgroup each)
  const CsrSplit sp = csr_split(B, L, E);
  const size_t prep_lds = ((size_t)kCsrCoarse * 2 + 4 + 32 + 4 + 3 * kCsrList + sp.HP) * sizeof(int);
  auto k = csr_prep_kernel<EP>;
  if (int e = enable_big_lds(k, prep_lds)) { set_error("%s: LDS attribute: %d", what, e); return e; }
  hipLaunchKernelGGL(k, dim3(sp.P, B), dim3(kCsrThreads), prep_lds, s, ep, E, L, sp, cnt_out, pl.start,
                     static_cast<int32_t *>(scratch), pl.ent);
  if (int e = check_launch(what)) return e;
  hipLaunchKernelGGL(csr_order_kernel, dim3(ceil_div(L, kTileTargets), B), dim3(kTileThreads), 0, s, pl.start, pl.ent, L, E, pl.order, pl.seg,
                     pl.gofs, pl.entw);
  if (int e = check_launch(what)) return e;
  hipLaunchKernelGGL(csr_interleave_kernel, dim3(order_stride(L) / 256, B), dim3(256), 0, s, pl.ent, pl.seg, pl.gofs, L, E, pl.entw);
  return check_launch(what);
}

// Step 1 for TWO scatters over the same L targets of the same clouds (a PVConv geometry: voxelize + devoxelize-backward of one
// (coords, R)) in THREE launches instead of six: each step runs both plans side by side (see csr_prep_pair_kernel).
template <class EP1, class EP2>
int launch_csr_prep_pair(const EP1 &ep1, long E1_, int32_t *cnt1, void *plan1, size_t plan1_bytes, const EP2 &ep2, long E2_, void *plan2,
                         size_t plan2_bytes, int B, int L, void *scratch, size_t scratch_bytes, hipStream_t s, const char *what) {
  const int E1 = (int)E1_, E2 = (int)E2_;
  if (B == 0) return 0;
  const size_t s1 = csr_prep_scratch_bytes(B, E1), s2 = csr_prep_scratch_bytes(B, E2);
  if (!plan1 || !plan2 || plan1_bytes < CsrPlan::bytes(B, L, E1) || plan2_bytes < CsrPlan::bytes(B, L, E2) || !aligned16(plan1) ||
      !aligned16(plan2) || !scratch || !aligned16(scratch) || scratch_bytes < s1 + s2) {
    set_error("%s: plans / scratch missing, misaligned or too small (%zu, %zu, %zu bytes needed)", what, CsrPlan::bytes(B, L, E1),
              CsrPlan::bytes(B, L, E2), s1 + s2);
    return PVCNN_ERR_INVALID_ARGUMENT;
  }
  CsrPlan p1, p2;
  p1.carve(plan1, B, L, E1);
  p2.carve(plan2, B, L, E2);
  // the chip holds kNumCU / B prep workgroups per cloud in one round: a quarter for the light sort, the rest for the heavy one
  const int total = std::max(2, std::min(16, kNumCU / std::max(B, 1)));
  const int cap1 = std::max(1, total / 4);
  const CsrSplit sp1 = csr_split(B, L, E1, cap1), sp2 = csr_split(B, L, E2, std::max(1, total - cap1));
  const size_t prep_lds = ((size_t)kCsrCoarse * 2 + 4 + 32 + 4 + 3 * kCsrList + std::max(sp1.HP, sp2.HP)) * sizeof(int);
  auto k = csr_prep_pair_kernel<EP1, EP2>;
  if (int e = enable_big_lds(k, prep_lds)) { set_error("%s: LDS attribute: %d", what, e); return e; }
  char *sc = static_cast<char *>(scratch);
  hipLaunchKernelGGL(k, dim3(sp1.P + sp2.P, B), dim3(kCsrThreads), prep_lds, s, ep1, E1, sp1, cnt1, p1.start, reinterpret_cast<int32_t *>(sc),
                     p1.ent, ep2, E2, sp2, p2.start, reinterpret_cast<int32_t *>(sc + s1), p2.ent, L);
  if (int e = check_launch(what)) return e;
  const CsrPlanRef r1{p1.start, p1.ent, p1.order, p1.seg, p1.gofs, p1.entw, E1}, r2{p2.start, p2.ent, p2.order, p2.seg, p2.gofs, p2.entw, E2};
  const int nr = ceil_div(L, kTileTargets), nblk = order_stride(L) / 256;
  hipLaunchKernelGGL(csr_order_pair_kernel, dim3(2 * nr, B), dim3(kTileThreads), 0, s, r1, r2, L, nr);
  if (int e = check_launch(what)) return e;
  hipLaunchKernelGGL(csr_interleave_pair_kernel, dim3(2 * nblk, B), dim3(256), 0, s, r1, r2, L, nblk);
  return check_launch(what);
}

// Step 2: dst (B,C,L) = segmented sums of src (B,C,J) over the plan's entries.  src: rows of a cloud contiguous, clouds
// src_bstride floats apart.
inline int launch_csr_apply(const float *src, const void *plan, size_t plan_bytes, float *dst, int B, int C, int L, int J, long E_,
                            hipStream_t s, const char *what, long src_bstride = 0) {
  if (src_bstride <= 0) src_bstride = (long)C * J;
  const int E = (int)E_;
  if (B == 0 || C == 0) return 0;
  if (!plan || plan_bytes < CsrPlan::bytes(B, L, E) || !aligned16(plan)) {
    set_error("%s: plan missing, misaligned or too small (%zu bytes needed)", what, CsrPlan::bytes(B, L, E));
    return PVCNN_ERR_INVALID_ARGUMENT;
  }
  CsrPlan pl;
  pl.carve(const_cast<void *>(plan), B, L, E);
  // tile kernel: G source rows channel-interleaved + a G x kTileTargets output tile in LDS
  const int JP = (J + 3) & ~3;
  int G = 0;
  for (int cand = 4; cand >= 1; cand >>= 1)
    if ((size_t)cand * (JP + kTileTargets) * sizeof(float) <= (size_t)kLdsBytesPerCU) { G = cand; break; }
  // one entry per >= 4 targets (voxelize at R = 32) AND enough channel slabs to fill the chip: the lane-per-4-targets kernel
  // below is then just the coalesced write of a mostly empty grid (measured 31 vs 36 us at (16,64,4096,32))
  const size_t row = (size_t)J * sizeof(float);
  int GS = 1;
  if (row > 0 && row <= 64 * 1024) {
    GS = (int)std::min<size_t>(8, (64 * 1024) / row);
    while (GS > 1 && (long)B * ceil_div(C, GS) < 2L * kNumCU) GS >>= 1;
    if (GS >= 8) GS = 8; else if (GS >= 4) GS = 4; else if (GS >= 2) GS = 2; else GS = 1;
  }
  const bool very_sparse = (long)E * 4 <= (long)L && row > 0 && row <= (size_t)kLdsBytesPerCU && (long)B * ceil_div(C, GS) >= kNumCU;
  if (G > 0 && J > 0 && !very_sparse) {
    while (G > 1 && G / 2 >= C) G >>= 1;                                   // C = 1, 2: no wider than needed
    const int nr = ceil_div(L, kTileTargets), slabs = ceil_div(C, G);
    int nsplit = 1;
    while (nsplit < nr && (long)B * slabs * nsplit < kNumCU) nsplit <<= 1;   // few channels: spread the ranges too
    nsplit = std::min(nsplit, nr);
    const size_t lds = (size_t)G * (JP + kTileTargets) * sizeof(float);
    const dim3 grid(slabs, nsplit, B);
#define PVCNN_TILE(GV)                                                                                                   \
  do {                                                                                                                   \
    auto k = segsum_tile_kernel<GV>;                                                                                     \
    if (int e = enable_big_lds(k, lds)) { set_error("%s: LDS attribute: %d", what, e); return e; }                       \
    hipLaunchKernelGGL(k, grid, dim3(kTileThreads), lds, s, src, pl.seg, pl.ent, pl.order, pl.gofs, pl.entw, dst, C, L, J, E,  \
                       src_bstride, nsplit, JP);                                                                                      \
  } while (0)
    if (G == 4) PVCNN_TILE(4); else if (G == 2) PVCNN_TILE(2); else PVCNN_TILE(1);
#undef PVCNN_TILE
    return check_launch(what);
  }
  // very sparse targets (voxelize at R = 32: one entry per 8 voxels) or source rows too long for LDS next to a tile:
  // lane-per-4-consecutive-targets sums, source rows in LDS.  Adjacent lanes own adjacent targets, whose entries are adjacent
  // in `ent`: reads stay coalesced, and with hardly any entries the kernel is the coalesced write of the (mostly zero) grid.
  const bool stage = row > 0 && row <= (size_t)kLdsBytesPerCU;
  const bool vec = (L % 4 == 0) && aligned16(dst);
  const int threads = (L >= 8192 || (size_t)GS * row > 48 * 1024) ? 1024 : 256;
  if (!stage) return launch_segsum_g<1, false>(src, pl, dst, B, C, L, J, E, vec, threads, s, what, src_bstride);
  switch (GS) {
    case 8: return launch_segsum_g<8, true>(src, pl, dst, B, C, L, J, E, vec, threads, s, what, src_bstride);
    case 4: return launch_segsum_g<4, true>(src, pl, dst, B, C, L, J, E, vec, threads, s, what, src_bstride);
    case 2: return launch_segsum_g<2, true>(src, pl, dst, B, C, L, J, E, vec, threads, s, what, src_bstride);
    default: return launch_segsum_g<1, true>(src, pl, dst, B, C, L, J, E, vec, threads, s, what, src_bstride);
  }
}

// one-shot: plan + apply in caller-owned workspace (CsrWorkspace::bytes)
template <class EP>
int launch_csr_scatter(const EP &ep, const float *src, float *dst, int B, int C, int L, int J, long E,
                       int32_t *cnt_out, void *workspace, size_t workspace_bytes, hipStream_t s, const char *what,
                       long src_bstride = 0) {
  if (B == 0) return 0;
  const size_t pb = CsrPlan::bytes(B, L, E);
  if (!workspace || workspace_bytes < pb + csr_prep_scratch_bytes(B, E) || !aligned16(workspace)) {
    set_error("%s: workspace missing, misaligned or too small (%zu bytes needed)", what, pb + csr_prep_scratch_bytes(B, E));
    return PVCNN_ERR_INVALID_ARGUMENT;
  }
  char *w = static_cast<char *>(workspace);
  if (int e = launch_csr_prep(ep, B, L, E, cnt_out, w, pb, w + pb, workspace_bytes - pb, s, what)) return e;
  return launch_csr_apply(src, w, pb, dst, B, C, L, J, E, s, what, src_bstride);
}

}  // namespace pvcnn
