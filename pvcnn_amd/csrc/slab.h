// slab.h -- the two workhorse kernels of the PVConv hot path on gfx950.
//
// Almost every native op of the reference is "random 4-byte access into one channel row":
//   out[b,c,j]   = sum_k w_k(b,j) * row[b,c][idx_k(b,j)]          (gather : devoxelize fwd,
//                  voxelize bwd, grouping/gather fwd, 3-NN interpolate fwd)
//   row[b,c][idx_k(b,j)] += w_k(b,j) * g[b,c,j]                    (scatter: devoxelize bwd,
//                  grouping/gather bwd, 3-NN interpolate bwd)
// where a "row" is one channel of one cloud: a voxel grid of S = R^3 floats or a point
// feature row of N floats.  The reference walks these with one 512-thread block per cloud
// and uncoalesced global loads / global float atomics (e.g. trilinear_devox.cu:96-103,
// :145-157).  Here a workgroup owns a SLAB = G consecutive channel rows of one cloud:
//   gather : the slab is streamed HBM -> LDS once with 16-byte coalesced loads, all random
//            reads are served by LDS (ds_read_b32), outputs leave as coalesced 16-byte stores;
//   scatter: see csr.h -- float atomics (LDS or global) are far too slow on gfx950, so scatters
//            are a per-cloud counting sort + lane-owned segmented sums instead.
// HBM traffic is therefore the compulsory minimum (slab once + per-element streams once).
// An R=32 grid row is 128 KiB: it fits the 160 KiB LDS of a gfx950 CU as a single-row slab.
// Refinements (all below): a workgroup walks several slabs with its elements' taps packed in registers (and, when a slab is
// a whole LDS, prefetches the next one into registers while it gathers from the current one: gather_lds_pipe_kernel);
// voxel-grid rows are staged with a padded z-row stride so planar clouds do not serialise on one LDS bank;
// a per-row transform (BatchNorm + LeakyReLU) can be applied while a slab is staged.
// Rows that do not fit LDS (R > 34, N > 40960) take the *_direct kernels (global gathers /
// global atomics after a memset).
//
// The index/weight source ("provider") is a template parameter; see the structs below.
#pragma once
#include <algorithm>
#include <stdlib.h>

#include "common.h"

namespace pvcnn {

template <int NC>
struct Taps {
  int32_t idx[NC];
  float w[NC];
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ int4 ld4(const int32_t *p) { return *reinterpret_cast<const int4 *>(p); }
__device__ __forceinline__ void st4(float *p, float a, float b, float c, float d) {
  *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void st4(int32_t *p, int a, int b, int c, int d) {
  *reinterpret_cast<int4 *>(p) = make_int4(a, b, c, d);
}

// sum_k w_k * row[idx_k], left to right, with the floating-point contraction the reference's
// expressions (trilinear_devox.cu:98-102, neighbor_interpolate.cu:112-114) receive from
// LLVM/NVVM and GCC alike: the first addition fuses its LEFT product, w0*f0 + w1*f1 ->
// fma(w0, f0, w1*f1), and every further term is an fma onto the running sum (verified against the
// reference's own sources, oracle/_ref).  With NC == 1 it is a single product (voxelize bwd) and
// with w == 1.0f an exact copy (grouping / gather).
// PAD: the row sits in LDS with one pad float after every 2^pshift elements (position i + (i >> pshift)).
// A voxel grid of resolution R = 2^pshift then has a z-row stride of R + 1, so the LDS bank of a voxel is
// (x + y + z) mod 32-ish instead of z mod 32: points on a floor or a wall -- same z, or same y -- no longer
// pile onto one bank (measured on planar clouds: the unpadded gather is 1.4x slower than on uniform ones).
template <int NC, bool MAY_SKIP, bool PAD = false>
__device__ __forceinline__ float combine(const Taps<NC> &t, const float *row, int pshift = 0) {
  if (MAY_SKIP && t.idx[0] < 0) return 0.0f;
  auto at = [&](int i) { return PAD ? row[i + (i >> pshift)] : row[i]; };
  if constexpr (NC == 1) {
    return t.w[0] * at(t.idx[0]);
  } else {
    float acc = fmaf(t.w[0], at(t.idx[0]), t.w[1] * at(t.idx[1]));
#pragma unroll
    for (int k = 2; k < NC; ++k) acc = fmaf(t.w[k], at(t.idx[k]), acc);
    return acc;
  }
}

// ---------------------------------------------------------------------------------------------
// Providers.  load1: taps of element j of cloud b.  load4: taps of elements j0..j0+3 using
// 16-byte loads (only launched when J % 4 == 0 and all arrays are 16-byte aligned).
// post1/post4: side outputs, executed by the workgroups of channel-slab 0 only.
// ---------------------------------------------------------------------------------------------

// Trilinear corner indices/weights from point coordinates: trilinear_devox.cu:41-75.
struct TrilinearFromCoords {
  static constexpr int NC = 8;
  static constexpr bool kMaySkip = false;
  static constexpr bool kGridRows = true;    // rows are R^3 voxel grids: the padded LDS layout applies
  const float *coords;   // (B,3,N)
  int32_t *inds;         // (B,8,N) or nullptr
  float *wgts;           // (B,8,N) or nullptr
  int N, R, R2;

  // Packed form of a point's taps (4 registers instead of 16): base corner + the three fractions.
  // unpack() evaluates exactly the expressions of setup(), so keeping points packed across channel
  // slabs and expanding them per slab changes no bit of the result.
  struct Packed {
    int32_t i000;
    float xd1, yd1, zd1;
  };
  __device__ __forceinline__ Packed pack(float x, float y, float z) const {
    const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
    Packed pk;
    pk.xd1 = x - xl; pk.yd1 = y - yl; pk.zd1 = z - zl;
    pk.i000 = (int)xl * R2 + (int)yl * R + (int)zl;
    return pk;
  }
  __device__ __forceinline__ void unpack(const Packed &pk, Taps<8> &t) const {
    const float xd1 = pk.xd1, yd1 = pk.yd1, zd1 = pk.zd1;
    const float xd0 = 1.0f - xd1, yd0 = 1.0f - yd1, zd0 = 1.0f - zd1;
    const float w00 = xd0 * yd0, w01 = xd0 * yd1, w10 = xd1 * yd0, w11 = xd1 * yd1;
    t.w[0] = w00 * zd0; t.w[1] = w00 * zd1; t.w[2] = w01 * zd0; t.w[3] = w01 * zd1;
    t.w[4] = w10 * zd0; t.w[5] = w10 * zd1; t.w[6] = w11 * zd0; t.w[7] = w11 * zd1;
    const int i000 = pk.i000;
    const int zo = (zd1 > 0) ? 1 : 0;
    const int yo = (yd1 > 0) ? R : 0;
    const int xo = (xd1 > 0) ? R2 : 0;
    t.idx[0] = i000;           t.idx[1] = i000 + zo;
    t.idx[2] = i000 + yo;      t.idx[3] = i000 + yo + zo;
    t.idx[4] = i000 + xo;      t.idx[5] = i000 + xo + zo;
    t.idx[6] = i000 + xo + yo; t.idx[7] = i000 + xo + yo + zo;
  }
  __device__ __forceinline__ void setup(float x, float y, float z, Taps<8> &t) const { unpack(pack(x, y, z), t); }
  __device__ __forceinline__ void pack1(int b, int j, Packed &pk) const {
    const float *c = coords + (size_t)b * 3 * N;
    pk = pack(c[j], c[j + N], c[j + 2 * N]);
  }
  __device__ __forceinline__ void pack4(int b, int j0, Packed (&pk)[4]) const {
    const float *c = coords + (size_t)b * 3 * N;
    const float4 x = ld4(c + j0), y = ld4(c + N + j0), z = ld4(c + 2 * N + j0);
    pk[0] = pack(x.x, y.x, z.x); pk[1] = pack(x.y, y.y, z.y);
    pk[2] = pack(x.z, y.z, z.z); pk[3] = pack(x.w, y.w, z.w);
  }
  __device__ __forceinline__ void load1(int b, int j, Taps<8> &t) const {
    const float *c = coords + (size_t)b * 3 * N;
    setup(c[j], c[j + N], c[j + 2 * N], t);
  }
  __device__ __forceinline__ void load4(int b, int j0, Taps<8> (&t)[4]) const {
    const float *c = coords + (size_t)b * 3 * N;
    const float4 x = ld4(c + j0), y = ld4(c + N + j0), z = ld4(c + 2 * N + j0);
    setup(x.x, y.x, z.x, t[0]); setup(x.y, y.y, z.y, t[1]);
    setup(x.z, y.z, z.z, t[2]); setup(x.w, y.w, z.w, t[3]);
  }
  __device__ __forceinline__ void post1(int b, int j, const Taps<8> &t) const {
    if (!inds) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      inds[((size_t)b * 8 + k) * N + j] = t.idx[k];
      wgts[((size_t)b * 8 + k) * N + j] = t.w[k];
    }
  }
  __device__ __forceinline__ void post4(int b, int j0, const Taps<8> (&t)[4]) const {
    if (!inds) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      st4(inds + ((size_t)b * 8 + k) * N + j0, t[0].idx[k], t[1].idx[k], t[2].idx[k], t[3].idx[k]);
      st4(wgts + ((size_t)b * 8 + k) * N + j0, t[0].w[k], t[1].w[k], t[2].w[k], t[3].w[k]);
    }
  }
};

// Saved (B,NC,J) index / weight planes: devoxelize bwd (NC=8), 3-NN interpolate (NC=3).
template <int NC_>
struct SavedTaps {
  static constexpr int NC = NC_;
  static constexpr bool kMaySkip = false;
  static constexpr bool kGridRows = false;
  const int32_t *inds;   // (B,NC,J)
  const float *wgts;     // (B,NC,J)
  int J;
  __device__ __forceinline__ void load1(int b, int j, Taps<NC> &t) const {
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      t.idx[k] = inds[((size_t)b * NC + k) * J + j];
      t.w[k] = wgts[((size_t)b * NC + k) * J + j];
    }
  }
  __device__ __forceinline__ void load4(int b, int j0, Taps<NC> (&t)[4]) const {
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int4 i = ld4(inds + ((size_t)b * NC + k) * J + j0);
      const float4 w = ld4(wgts + ((size_t)b * NC + k) * J + j0);
      t[0].idx[k] = i.x; t[1].idx[k] = i.y; t[2].idx[k] = i.z; t[3].idx[k] = i.w;
      t[0].w[k] = w.x; t[1].w[k] = w.y; t[2].w[k] = w.z; t[3].w[k] = w.w;
    }
  }
  using Packed = Taps<NC>;   // nothing to compress: the taps are what is kept across channel slabs
  __device__ __forceinline__ void pack1(int b, int j, Packed &pk) const { load1(b, j, pk); }
  __device__ __forceinline__ void pack4(int b, int j0, Packed (&pk)[4]) const { load4(b, j0, pk); }
  __device__ __forceinline__ void unpack(const Packed &pk, Taps<NC> &t) const { t = pk; }
  __device__ __forceinline__ void post1(int, int, const Taps<NC> &) const {}
  __device__ __forceinline__ void post4(int, int, const Taps<NC> (&)[4]) const {}
};

// One index per element, weight 1: grouping (J = M*U) and gather (J = M).
struct IndexOnly {
  static constexpr int NC = 1;
  static constexpr bool kMaySkip = false;
  static constexpr bool kGridRows = false;
  const int32_t *idx;   // (B,J)
  int J;
  __device__ __forceinline__ void load1(int b, int j, Taps<1> &t) const {
    t.idx[0] = idx[(size_t)b * J + j];
    t.w[0] = 1.0f;
  }
  __device__ __forceinline__ void load4(int b, int j0, Taps<1> (&t)[4]) const {
    const int4 i = ld4(idx + (size_t)b * J + j0);
    t[0].idx[0] = i.x; t[1].idx[0] = i.y; t[2].idx[0] = i.z; t[3].idx[0] = i.w;
    t[0].w[0] = t[1].w[0] = t[2].w[0] = t[3].w[0] = 1.0f;
  }
  using Packed = Taps<NC>;   // nothing to compress: the taps are what is kept across channel slabs
  __device__ __forceinline__ void pack1(int b, int j, Packed &pk) const { load1(b, j, pk); }
  __device__ __forceinline__ void pack4(int b, int j0, Packed (&pk)[4]) const { load4(b, j0, pk); }
  __device__ __forceinline__ void unpack(const Packed &pk, Taps<NC> &t) const { t = pk; }
  __device__ __forceinline__ void post1(int, int, const Taps<1> &) const {}
  __device__ __forceinline__ void post4(int, int, const Taps<1> (&)[4]) const {}
};

// Voxelize backward: idx = ind[b,j], w = 1/cnt[b,idx] (vox.cu:99-106); cnt == 0 -> output 0.
struct VoxelMean {
  static constexpr int NC = 1;
  static constexpr bool kMaySkip = true;
  static constexpr bool kGridRows = false;   // one LDS read per point: the padded staging costs more than conflicts do
  const int32_t *ind;   // (B,N)
  const int32_t *cnt;   // (B,S)
  int N, S;
  __device__ __forceinline__ void one(int b, int pos, Taps<1> &t) const {
    const int c = cnt[(size_t)b * S + pos];
    t.idx[0] = (c > 0) ? pos : -1;
    t.w[0] = (c > 0) ? (float)(1.0 / (double)(float)c) : 0.0f;
  }
  __device__ __forceinline__ void load1(int b, int j, Taps<1> &t) const { one(b, ind[(size_t)b * N + j], t); }
  __device__ __forceinline__ void load4(int b, int j0, Taps<1> (&t)[4]) const {
    const int4 p = ld4(ind + (size_t)b * N + j0);
    one(b, p.x, t[0]); one(b, p.y, t[1]); one(b, p.z, t[2]); one(b, p.w, t[3]);
  }
  using Packed = Taps<NC>;   // nothing to compress: the taps are what is kept across channel slabs
  __device__ __forceinline__ void pack1(int b, int j, Packed &pk) const { load1(b, j, pk); }
  __device__ __forceinline__ void pack4(int b, int j0, Packed (&pk)[4]) const { load4(b, j0, pk); }
  __device__ __forceinline__ void unpack(const Packed &pk, Taps<NC> &t) const { t = pk; }
  __device__ __forceinline__ void post1(int, int, const Taps<1> &) const {}
  __device__ __forceinline__ void post4(int, int, const Taps<1> (&)[4]) const {}
};

// ---------------------------------------------------------------------------------------------
// LDS slab kernels.  grid = (ceil(C/G), B); blockIdx.x = channel slab, blockIdx.y = cloud.
// ---------------------------------------------------------------------------------------------
// Streaming copy (HBM/L2 -> LDS or LDS -> HBM).  Latency-bound unless many loads are in flight, so a
// thread issues 8 independent 16-byte loads before the first store (8 KiB in flight per wave).
template <int THREADS>
__device__ __forceinline__ void slab_copy(float *dst, const float *src, int total) {
  // wave-uniform branch: 16-byte path when both ends are aligned and the size is a multiple of 4 floats
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 && (total & 3) == 0) {
    constexpr int kB = 8;
    const int nq = total >> 2;
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    for (int q0 = threadIdx.x; q0 < nq; q0 += THREADS * kB) {
      float4 v[kB];
#pragma unroll
      for (int u = 0; u < kB; ++u) v[u] = s4[min(q0 + u * THREADS, nq - 1)];
#pragma unroll
      for (int u = 0; u < kB; ++u)
        if (q0 + u * THREADS < nq) d4[q0 + u * THREADS] = v[u];
    }
  } else {
    for (int i = threadIdx.x; i < total; i += THREADS) dst[i] = src[i];
  }
}

// Optional per-row transform applied while a slab is staged into LDS ("gather of f(row)" without ever writing
// f(row) to memory).  XfNone: plain copy.  XfBnAct: BatchNorm (per-channel scale/shift) + LeakyReLU with the
// very expressions of bnact_apply_kernel (bnact.hip), so gathering through it is bit-identical to gathering
// the tensor that kernel would have written.
struct XfNone {
  static constexpr bool kIdentity = true;
  static constexpr int kRaw = 1;
  __device__ __forceinline__ XfNone at(int) const { return *this; }
};
// `se` (optional, (B, C) row-major): the squeeze-and-excitation factor of SE3d (modules/se.py:17), y = act(bn(x)) * se[b][c] --
// the multiplication is a SECOND rounded operation on the activated value, exactly like the reference's separate `inputs * fc(...)`.
struct XfBnAct {
  static constexpr bool kIdentity = false;
  static constexpr int kRaw = 5;
  const float *gamma, *beta, *mean, *rstd;   // per channel; gamma / beta may be null
  float slope;
  const float *se = nullptr;                 // per (cloud, channel) excitation, or null; kernels take `xf.at(b)`: the cloud's row
  __device__ __forceinline__ XfBnAct at(int b_times_C) const {
    XfBnAct x = *this;
    if (x.se) x.se += b_times_C;
    return x;
  }
  __device__ __forceinline__ void params(int c, float &scale, float &shift) const {
    scale = (gamma ? gamma[c] : 1.0f) * rstd[c];
    shift = (beta ? beta[c] : 0.0f) - mean[c] * scale;
  }
  __device__ __forceinline__ float mul(int c) const { return se ? se[c] : 1.0f; }
  __device__ __forceinline__ float apply(float v, float scale, float shift, float m = 1.0f) const {
    v = fmaf(v, scale, shift);
    v = v > 0.f ? v : v * slope;
    return se ? v * m : v;
  }
  // params() in two halves, for kernels that fetch a row's raw parameters one slab ahead and combine them when the row arrives
  __device__ __forceinline__ void fetch(int c, float (&raw)[5]) const {
    raw[0] = gamma ? gamma[c] : 1.0f; raw[1] = rstd[c]; raw[2] = beta ? beta[c] : 0.0f; raw[3] = mean[c]; raw[4] = mul(c);
  }
  __device__ __forceinline__ void combine(const float (&raw)[5], float &scale, float &shift) const {
    scale = raw[0] * raw[1];
    shift = raw[2] - raw[3] * scale;
  }
};

// one row of a slab through a transform (row = channel c of the cloud)
template <int THREADS, class XF>
__device__ __forceinline__ void slab_copy_row_xf(float *dst, const float *src, int len, const XF &xf, int c) {
  float scale, shift;
  xf.params(c, scale, shift);
  const float m = xf.mul(c);
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 && (len & 3) == 0) {
    constexpr int kB = 8;
    const int nq = len >> 2;
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    for (int q0 = threadIdx.x; q0 < nq; q0 += THREADS * kB) {
      float4 v[kB];
#pragma unroll
      for (int u = 0; u < kB; ++u) v[u] = s4[min(q0 + u * THREADS, nq - 1)];
#pragma unroll
      for (int u = 0; u < kB; ++u)
        if (q0 + u * THREADS < nq)
          d4[q0 + u * THREADS] = make_float4(xf.apply(v[u].x, scale, shift, m), xf.apply(v[u].y, scale, shift, m),
                                             xf.apply(v[u].z, scale, shift, m), xf.apply(v[u].w, scale, shift, m));
    }
  } else {
    for (int i = threadIdx.x; i < len; i += THREADS) dst[i] = xf.apply(src[i], scale, shift, m);
  }
}

// one row into the padded layout: 16-byte global loads, four 4-byte LDS stores per quad (a quad never
// straddles a pad when 2^pshift >= 4; 4-byte LDS stores move as many bytes per cycle as 16-byte ones)
template <int THREADS, class XF>
__device__ __forceinline__ void slab_copy_row_padded(float *dst, const float *src, int len, int pshift, const XF &xf, int c) {
  float scale = 1.0f, shift = 0.0f;
  [[maybe_unused]] float m = 1.0f;
  if constexpr (!XF::kIdentity) { xf.params(c, scale, shift); m = xf.mul(c); }
  auto f = [&](float v) {
    if constexpr (XF::kIdentity) return v; else return xf.apply(v, scale, shift, m);
  };
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (len & 3) == 0) {
    constexpr int kB = 8;
    const int nq = len >> 2;
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    for (int q0 = threadIdx.x; q0 < nq; q0 += THREADS * kB) {
      float4 v[kB];
#pragma unroll
      for (int u = 0; u < kB; ++u) v[u] = s4[min(q0 + u * THREADS, nq - 1)];
#pragma unroll
      for (int u = 0; u < kB; ++u)
        if (q0 + u * THREADS < nq) {
          const int i = (q0 + u * THREADS) * 4;
          float *d = dst + i + (i >> pshift);
          d[0] = f(v[u].x); d[1] = f(v[u].y); d[2] = f(v[u].z); d[3] = f(v[u].w);
        }
    }
  } else {
    for (int i = threadIdx.x; i < len; i += THREADS) dst[i + (i >> pshift)] = f(src[i]);
  }
}

template <int THREADS, bool PAD, class XF>
__device__ __forceinline__ void slab_stage(float *lds, const float *src, int g, int L, int Lp, int pshift, const XF &xf, int c0) {
  if constexpr (PAD) {
    for (int c = 0; c < g; ++c) slab_copy_row_padded<THREADS>(lds + c * Lp, src + (size_t)c * L, L, pshift, xf, c0 + c);
  } else if constexpr (XF::kIdentity) {
    slab_copy<THREADS>(lds, src, g * L);
  } else {
    for (int c = 0; c < g; ++c) slab_copy_row_xf<THREADS>(lds + c * L, src + (size_t)c * L, L, xf, c0 + c);
  }
}

// A workgroup walks SEQ consecutive channel slabs of one cloud through the same LDS buffer.  When a
// thread owns all of its elements in one pass (J <= THREADS*VEC, the usual case) their taps are fetched and
// packed ONCE and stay in registers for all slabs: coordinates / indices are read from memory once per
// SEQ*G channels instead of once per slab, and only "stream slab, barrier, LDS gather, store" repeats.
template <class P, int VEC, int THREADS, bool RESIDENT, class XF, bool PAD>
__global__ __launch_bounds__(THREADS) void gather_lds_kernel(P p, XF xf, const float *__restrict__ src,
                                                             float *__restrict__ dst, int C, int L,
                                                             int J, int G, int SEQ, int pshift,
                                                             const float *__restrict__ addend = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NC = P::NC;
  const int b = blockIdx.y;
  const XF xfb = xf.at(b * C);                          // (the transform's per-cloud part: the SE factors of cloud b)
  const bool side = (blockIdx.x == 0);
  constexpr bool resident = RESIDENT;                  // J <= THREADS * VEC
  const int jf = threadIdx.x * VEC;
  typename P::Packed pk[VEC];
  if (resident && jf < J) {
    if constexpr (VEC == 4) p.pack4(b, jf, pk); else p.pack1(b, jf, pk[0]);
    if (side) {   // side outputs (devoxelize: inds / wgts) once per cloud
      Taps<NC> t[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) p.unpack(pk[v], t[v]);
      if constexpr (VEC == 4) p.post4(b, jf, t); else p.post1(b, jf, t[0]);
    }
  }
  for (int sq = 0; sq < SEQ; ++sq) {
    const int c0 = (blockIdx.x * SEQ + sq) * G;
    if (c0 >= C) break;
    const int g = min(G, C - c0);
    if (sq > 0) lds_barrier();                         // all LDS reads of the previous slab are done (its output stores keep draining)
    const int Lp = PAD ? L + (L >> pshift) : L;            // LDS row length
    slab_stage<THREADS, PAD>(lds, src + ((size_t)b * C + c0) * L, g, L, Lp, pshift, xfb, c0);
    __syncthreads();
    float *out = dst + ((size_t)b * C + c0) * J;
    // optional epilogue: out = gather + addend (PVConv's "devoxelized voxel branch + point branch", modules/pvconv.py:38,
    // one rounded fp32 addition like the reference's separate kernel) -- saves a full read-modify-write pass over (B,C,J)
    const float *add = addend ? addend + ((size_t)b * C + c0) * J : nullptr;
    for (int j0 = jf; j0 < J; j0 += THREADS * VEC) {
      Taps<NC> t[VEC];
      if constexpr (resident) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) p.unpack(pk[v], t[v]);
      } else if constexpr (VEC == 4) {
        p.load4(b, j0, t);
        if (side && sq == 0) p.post4(b, j0, t);
      } else {
        p.load1(b, j0, t[0]);
        if (side && sq == 0) p.post1(b, j0, t[0]);
      }
      for (int c = 0; c < g; ++c) {
        const float *row = lds + c * Lp;
        if constexpr (VEC == 4) {
          const float r0 = combine<NC, P::kMaySkip, PAD>(t[0], row, pshift), r1 = combine<NC, P::kMaySkip, PAD>(t[1], row, pshift);
          float r2 = combine<NC, P::kMaySkip, PAD>(t[2], row, pshift), r3 = combine<NC, P::kMaySkip, PAD>(t[3], row, pshift);
          float q0 = r0, q1 = r1;
          if (add) { const float4 a = ld4(add + (size_t)c * J + j0); q0 = q0 + a.x; q1 = q1 + a.y; r2 = r2 + a.z; r3 = r3 + a.w; }
          st4(out + (size_t)c * J + j0, q0, q1, r2, r3);
        } else {
          float r = combine<NC, P::kMaySkip, PAD>(t[0], row, pshift);
          if (add) r = r + add[(size_t)c * J + j0];
          out[(size_t)c * J + j0] = r;
        }
      }
    }
  }
}

// The single-row-slab case, software-pipelined (R = 32: one 128 KiB grid fills the CU's LDS, ONE 1024-thread workgroup per CU, so
// in gather_lds_kernel nothing overlaps the HBM -> LDS stream with the gather phase -- per slab ~5 us of streaming and ~3 us of
// tap expansion + LDS gather run back to back: 41.6 us, 0.51 of the HBM peak at (16,64,4096,32)).  Here the NEXT slab's 8 x 16-byte
// loads per thread are issued right after the barrier that publishes the current slab and stay in flight (32 VGPRs) while the
// current slab is gathered; to make room the four points of a thread are expanded one at a time.  Same expressions, same order:
// bit-identical to gather_lds_kernel<P, 4, 1024, true, XF, true> (tools/pipecheck*.py: 7 ragged shapes x training / eval x plain /
// BatchNorm / addend) and 36.3 us on the same box (profiles/ab/r02_pipecheck_*).  PVCNN_GATHER_PIPE=0 selects the classic kernel.
// (Round 3: a 512-thread variant with TWO grids in flight -- 2 x 16 x 16-byte loads per thread, the slab loop unrolled so that the
// compiler's vmcnt waits leave the younger grid in flight -- measured 35.1 vs 35.2 us at (16,64,4096,32): with half the waves the LDS
// write and gather phases take twice as long, and the fixed part of the launch (one slab per CU alone costs 9 us; eight slabs per
// workgroup reach 0.69 of the HBM peak where four reach 0.55) is untouched.  Removed.)
template <class P, class XF>
__global__ __launch_bounds__(1024) void gather_lds_pipe_kernel(P p, XF xf, const float *__restrict__ src, float *__restrict__ dst,
                                                               int C, int L, int J, int SEQ, int pshift,
                                                               const float *__restrict__ addend = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NC = P::NC, THREADS = 1024, kB = 8;
  const int b = blockIdx.y;
  const XF xfb = xf.at(b * C);
  const int jf = threadIdx.x * 4;
  const bool has = jf < J;
  const int nq = L >> 2;                                    // <= THREADS * kB quads (checked by the launcher)
  float4 v[kB];
  // loads through a buffer descriptor: ONE per-lane offset register (tid * 16) for all eight loads, the rest of the address
  // (row base, u * 16 KiB) is scalar; the hardware bounds check returns zeros past the row (no clamp)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int voff = (int)threadIdx.x * 16;
  // STRAIGHT-LINE on purpose: every vector-memory operation of the slab loop is issued unconditionally -- a slab that does not exist
  // / a null addend / a thread past J is a buffer descriptor of ZERO bytes (the bounds check answers zeros and drops the store) --
  // because the compiler can only count outstanding loads along one path: with a branch around any of them it waits with vmcnt(0)
  // in front of every committed quad, i.e. for the load it has just issued (seen in the ISA of the first rolling version).
  auto row_descriptor = [&](const float *base, size_t off, int bytes) {
    // (the row base is wave-uniform; readfirstlane says so to the compiler, which otherwise wraps every load in a waterfall loop)
    const uintptr_t rowp = reinterpret_cast<uintptr_t>(base + off);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)rowp), hi = __builtin_amdgcn_readfirstlane((uint32_t)(rowp >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), /*stride*/ 0,
                                             /*bytes*/ __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
  };
  auto load = [&](const __amdgpu_buffer_rsrc_t &rsrc, int u) {
    const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, u * THREADS * 16, 0);
    v[u] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w));
  };
  const int first = blockIdx.x * SEQ;
  // the row transform's raw parameters are fetched one slab ahead, BEFORE the row's own loads (so that waiting for them
  // never waits for the row), and combined into (scale, shift) when the row is committed
  [[maybe_unused]] float raw[XF::kRaw];
  // ROLLING commit (round 4): load u of the NEXT grid is issued into the registers load u of this grid has just left for the LDS,
  // so the HBM stream never pauses while a grid is written to the LDS (32 ds_write_b32 per thread: ~1 us per grid, during which
  // the round-3 kernel had nothing in flight -- the next grid's loads went out only after the commit and the barrier behind it)
  auto commit = [&](bool refill, int cnext) {
    float scale = 1.0f, shift = 0.0f;
    [[maybe_unused]] float m = 1.0f;
    if constexpr (!XF::kIdentity) { xfb.combine(raw, scale, shift); m = raw[XF::kRaw - 1]; }
    auto f = [&](float x) {
      if constexpr (XF::kIdentity) return x; else return xfb.apply(x, scale, shift, m);
    };
    const int cn = refill ? cnext : first;                  // (a valid channel for the parameter fetch; its grid loads are 0 bytes)
    const __amdgpu_buffer_rsrc_t next = row_descriptor(src, ((size_t)b * C + cn) * L, refill ? L * 4 : 0);
    if constexpr (!XF::kIdentity) xfb.fetch(cn, raw);
#pragma unroll
    for (int u = 0; u < kB; ++u) {
      const int q = (int)threadIdx.x + u * THREADS;
      if (q < nq) {
        const int i = q * 4;
        float *d = lds + i + (i >> pshift);
        d[0] = f(v[u].x); d[1] = f(v[u].y); d[2] = f(v[u].z); d[3] = f(v[u].w);
      }
      load(next, u);
      __builtin_amdgcn_sched_barrier(0);                    // one load per committed quad, in this order
    }
  };
  {                                                         // the first grid's loads go out BEFORE the taps are derived: the
    const bool any = first < C;                             // coordinate loads and the tap arithmetic run under them
    if constexpr (!XF::kIdentity) xfb.fetch(any ? first : 0, raw);
    const __amdgpu_buffer_rsrc_t rsrc = row_descriptor(src, ((size_t)b * C + (any ? first : 0)) * L, any ? L * 4 : 0);
#pragma unroll
    for (int u = 0; u < kB; ++u) load(rsrc, u);
  }
  typename P::Packed pk[4] = {};
  if (has) p.pack4(b, jf, pk);
  for (int sq = 0; sq < SEQ; ++sq) {
    const int c = first + sq;
    if (c >= C) break;
    // this slab's addend quad is requested BEFORE the next grid's loads: loads return in order -- behind a grid's loads the
    // slab's store would wait for that whole grid
    const size_t rowoff = ((size_t)b * C + c) * J;
    const __amdgpu_buffer_rsrc_t arsrc = row_descriptor(addend ? addend : dst, rowoff, addend ? J * 4 : 0);
    const u32x4 add = __builtin_amdgcn_raw_buffer_load_b128(arsrc, voff, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (sq > 0) lds_barrier();                              // every wave is done reading the previous slab
    commit(sq + 1 < SEQ && c + 1 < C, c + 1);               // waits for this slab's loads only; refills the registers as it goes
    lds_barrier();                                          // slab visible (LDS-only: the previous outputs keep draining)
    float r[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      // the taps are re-expanded per slab ON PURPOSE (4 registers per point instead of 16 held across the loop): hide the
      // loop invariance from the optimiser, or it hoists the expansion out of the slab loop and spills it
      if constexpr (sizeof(typename P::Packed) == 16) {
        uint4 &raw4 = reinterpret_cast<uint4 &>(pk[h]);
        asm volatile("" : "+v"(raw4.x), "+v"(raw4.y), "+v"(raw4.z), "+v"(raw4.w));
      }
      Taps<NC> t;
      p.unpack(pk[h], t);
      r[h] = combine<NC, P::kMaySkip, true>(t, lds, pshift);
    }
    if (addend) {                                           // (no memory operation in here)
      r[0] = r[0] + __uint_as_float(add.x); r[1] = r[1] + __uint_as_float(add.y);
      r[2] = r[2] + __uint_as_float(add.z); r[3] = r[3] + __uint_as_float(add.w);
    }
    // uniform row descriptor + the thread's 32-bit byte offset: threads past J are dropped by the bounds check
    const __amdgpu_buffer_rsrc_t drsrc = row_descriptor(dst, rowoff, J * 4);
    u32x4 o;
    o.x = __float_as_uint(r[0]); o.y = __float_as_uint(r[1]); o.z = __float_as_uint(r[2]); o.w = __float_as_uint(r[3]);
    __builtin_amdgcn_raw_buffer_store_b128(o, drsrc, voff, 0, 0);
  }
  // side outputs (devoxelize: inds / wgts) once per cloud -- last: four expanded tap sets (64 registers) have no room while a grid's
  // loads are in flight
  if (has && blockIdx.x == 0) {
    Taps<NC> t[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) p.unpack(pk[v], t[v]);
    p.post4(b, jf, t);
  }
}

// The same pipeline for SMALL grids (R = 16: 16 KiB rows): G channel rows per slab.  With one row per slab a CU has 16 KiB in
// flight per HBM round trip (the pipelined kernel above at R = 16: 23 us for 42 MB, no better than the 256-thread slabs it
// replaced); with G rows it has G times that, the corner taps of a point are expanded once per G channels, and the first slab's
// grid loads are in flight together with the coordinate loads.  G = 2 (G = 4 spills ~70 registers at 1024 threads).  Requires L / 4 == 1024:
// load u of a thread is quad `tid` of row u, so the row transform of load u is that of channel c0 + u.  Same expressions per
// output element as gather_lds_kernel / gather_lds_pipe_kernel: bit-identical.
template <class P, class XF, int G>
__global__ __launch_bounds__(1024) void gather_lds_pipe_rows_kernel(P p, XF xf, const float *__restrict__ src, float *__restrict__ dst,
                                                                int C, int L, int J, int SEQ, int pshift,
                                                                const float *__restrict__ addend = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NC = P::NC, THREADS = 1024;
  const int b = blockIdx.y;
  const XF xfb = xf.at(b * C);
  const int jf = threadIdx.x * 4;
  const bool has = jf < J;
  const int Lp = L + (L >> pshift);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int voff = (int)threadIdx.x * 16;
  float4 v[G];
  auto issue_sized = [&](int c0, int bytes) {
    const uintptr_t rowp = reinterpret_cast<uintptr_t>(src + ((size_t)b * C + c0) * L);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)rowp), hi = __builtin_amdgcn_readfirstlane((uint32_t)(rowp >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), /*stride*/ 0, /*bytes*/ __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
#pragma unroll
    for (int u = 0; u < G; ++u) {                           // rows past C: the bounds check returns zeros
      const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, u * THREADS * 16, 0);
      v[u] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w));
    }
  };
  auto issue = [&](int c0) { issue_sized(c0, min(G, C - c0) * L * 4); };
  auto issue_none = [&]() { issue_sized(0, 0); };           // the same G loads, zero bytes: the loop's memory operations never branch
  [[maybe_unused]] float raw[G][XF::kRaw];
#pragma unroll
  for (int u = 0; u < G; ++u)
#pragma unroll
    for (int i = 0; i < XF::kRaw; ++i) raw[u][i] = (i < 2 || i == 4) ? 1.0f : 0.0f;
  auto fetch = [&](int c0) {
    if constexpr (!XF::kIdentity) {
#pragma unroll
      for (int u = 0; u < G; ++u)
        if (c0 + u < C) xfb.fetch(c0 + u, raw[u]);
    }
  };
  auto commit = [&]() {
    const int i = (int)threadIdx.x * 4;
#pragma unroll
    for (int u = 0; u < G; ++u) {
      float scale = 1.0f, shift = 0.0f;
      if constexpr (!XF::kIdentity) xfb.combine(raw[u], scale, shift);
      auto f = [&](float x) {
        if constexpr (XF::kIdentity) return x; else return xfb.apply(x, scale, shift, raw[u][XF::kRaw - 1]);
      };
      float *d = lds + u * Lp + i + (i >> pshift);
      d[0] = f(v[u].x); d[1] = f(v[u].y); d[2] = f(v[u].z); d[3] = f(v[u].w);
    }
  };
  const int first = blockIdx.x * SEQ * G;
  // straight-line slab loop, as in gather_lds_pipe_kernel: loads and stores through wave-uniform buffer descriptors whose SIZE says
  // what exists (rows past C, a null addend, threads past J: zero bytes) -- no branch around a memory operation, no 64-bit per-lane
  // address, and (round 4) no scratch: the round-3 form of this loop spilled 10 registers at the 128-register cap of a 1024-thread
  // workgroup, 44 bytes of scratch per lane stored and reloaded per launch (1.44x the algorithmic HBM traffic by the counters)
  auto row_descriptor = [&](const float *base, size_t off, int bytes) {
    const uintptr_t rowp = reinterpret_cast<uintptr_t>(base + off);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)rowp), hi = __builtin_amdgcn_readfirstlane((uint32_t)(rowp >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
  };
  {                                                         // the grid rows first: they are in flight while the taps are derived
    fetch(min(first, max(C - 1, 0)));
    if (first < C) issue(first); else issue_none();
  }
  typename P::Packed pk[4] = {};
  if (has) p.pack4(b, jf, pk);
  for (int sq = 0; sq < SEQ; ++sq) {
    const int c0 = first + sq * G;
    if (c0 >= C) break;
    const int rows = min(G, C - c0);
    const size_t rowoff = ((size_t)b * C + c0) * J;
    // the slab's addend quads (one per row) BEFORE the next slab's grid loads: loads return in order
    const __amdgpu_buffer_rsrc_t arsrc = row_descriptor(addend ? addend : dst, rowoff, addend ? rows * J * 4 : 0);
    u32x4 add[G];
#pragma unroll
    for (int u = 0; u < G; ++u) add[u] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, voff, u * J * 4, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (sq > 0) lds_barrier();                              // every wave is done reading the previous slab
    commit();                                               // waits for this slab's loads only
    lds_barrier();
    {
      const bool more = sq + 1 < SEQ && c0 + G < C;
      fetch(more ? c0 + G : c0);
      if (more) issue(c0 + G); else issue_none();
    }
    __builtin_amdgcn_sched_barrier(0);                      // keep the loads HERE
    float r[G][4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if constexpr (sizeof(typename P::Packed) == 16) {     // (see gather_lds_pipe_kernel: keep the expansion inside the slab loop)
        uint4 &rw = reinterpret_cast<uint4 &>(pk[h]);
        asm volatile("" : "+v"(rw.x), "+v"(rw.y), "+v"(rw.z), "+v"(rw.w));
      }
      Taps<NC> t;
      p.unpack(pk[h], t);
#pragma unroll
      for (int u = 0; u < G; ++u) r[u][h] = combine<NC, P::kMaySkip, true>(t, lds + u * Lp, pshift);
    }
    const __amdgpu_buffer_rsrc_t drsrc = row_descriptor(dst, rowoff, rows * J * 4);   // rows past C / threads past J: dropped
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (addend) {                                         // (no memory operation in here)
        r[u][0] = r[u][0] + __uint_as_float(add[u].x); r[u][1] = r[u][1] + __uint_as_float(add[u].y);
        r[u][2] = r[u][2] + __uint_as_float(add[u].z); r[u][3] = r[u][3] + __uint_as_float(add[u].w);
      }
      u32x4 o;
      o.x = __float_as_uint(r[u][0]); o.y = __float_as_uint(r[u][1]); o.z = __float_as_uint(r[u][2]); o.w = __float_as_uint(r[u][3]);
      // (row u of the slab starts u * J floats behind row 0; a thread past J must not land in the next row: its offset is pushed
      // past the descriptor's end instead)
      __builtin_amdgcn_raw_buffer_store_b128(o, drsrc, has ? voff : 0x7ffffff0, u * J * 4, 0);
    }
  }
  // side outputs (devoxelize: inds / wgts) once per cloud -- LAST: four expanded tap sets are 64 registers, which the loop above
  // (grid rows in flight + four rows of results) has no room for at 1024 threads
  if (has && blockIdx.x == 0) {
    Taps<NC> t[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) p.unpack(pk[h], t[h]);
    p.post4(b, jf, t);
  }
}

// ---------------------------------------------------------------------------------------------
// Direct (no-LDS) fallbacks for rows larger than LDS.  grid = (ceil(J/256), ceil(C/CT), B).
// scatter_direct needs dst zeroed first (the launcher enqueues a hipMemsetAsync).
// ---------------------------------------------------------------------------------------------
template <class P>
__global__ __launch_bounds__(256) void gather_direct_kernel(P p, const float *__restrict__ src,
                                                            float *__restrict__ dst, int C, int L,
                                                            int J, int CT) {
  constexpr int NC = P::NC;
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= J) return;
  Taps<NC> t;
  p.load1(b, j, t);
  if (blockIdx.y == 0) p.post1(b, j, t);
  const int c0 = blockIdx.y * CT, c1 = min(C, c0 + CT);
  for (int c = c0; c < c1; ++c)
    dst[((size_t)b * C + c) * J + j] = combine<NC, P::kMaySkip>(t, src + ((size_t)b * C + c) * L);
}

template <class P>
__global__ __launch_bounds__(256) void scatter_direct_kernel(P p, const float *__restrict__ src,
                                                             float *__restrict__ dst, int C, int L,
                                                             int J, int CT) {
  constexpr int NC = P::NC;
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= J) return;
  Taps<NC> t;
  p.load1(b, j, t);
  const int c0 = blockIdx.y * CT, c1 = min(C, c0 + CT);
  for (int c = c0; c < c1; ++c) {
    const float gv = src[((size_t)b * C + c) * J + j];
    float *row = dst + ((size_t)b * C + c) * L;
#pragma unroll
    for (int k = 0; k < NC; ++k) atomicAdd(row + t.idx[k], t.w[k] * gv);   // global_atomic_add_f32
  }
}

// ---------------------------------------------------------------------------------------------
// Host-side launch policy.
// ---------------------------------------------------------------------------------------------
struct SlabPlan {
  bool lds;       // LDS slab path possible
  int G;          // channel rows per slab
  int seq;        // consecutive slabs one workgroup walks (gather only)
  int threads;    // 256 or 1024
  size_t bytes;   // dynamic LDS bytes
};

// Rows per slab: keep a slab <= 64 KiB when a row allows it (>= 2 workgroups per CU so one
// workgroup's HBM->LDS stream overlaps another's LDS phase), and keep >= ~3 workgroups per CU
// in the grid; a row > 64 KiB (R = 32: 128 KiB) is a slab of its own with 1024 threads.
inline SlabPlan plan_slab(int B, int C, int L, int pshift = 0) {
  SlabPlan pl{};
  const size_t row = (size_t)(pshift ? L + (L >> pshift) : L) * sizeof(float);   // LDS bytes per row
  pl.lds = row <= (size_t)kLdsBytesPerCU;
  if (!pl.lds) return pl;
  int G = 1;
  if (row <= 64 * 1024) {
    G = (int)((64 * 1024) / row);
    if (G > C) G = C;
    while (G > 1 && (long)B * ceil_div(C, G) < 3L * kNumCU) G = (G + 1) / 2;
  }
  pl.G = G;
  pl.bytes = (size_t)G * row;
  pl.threads = (pl.bytes > 48 * 1024) ? 1024 : 256;
  // More slabs than the chip holds at once (R = 32: 1024 single-row slabs, one per CU at a time):
  // let a workgroup walk several in sequence so the per-cloud tap work is paid once per workgroup.
  const long resident_wgs = (long)kNumCU * std::max<long>(1, (long)(kLdsBytesPerCU / pl.bytes));
  const long slabs = (long)B * ceil_div(C, G);
  pl.seq = (int)std::min<long>(8, std::max<long>(1, slabs / resident_wgs));
  return pl;
}

template <class K>
inline int enable_big_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024)
    return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return 0;
}

// vec_ok: J % 4 == 0 and every per-element array (incl. src/dst rows of length J) 16-byte aligned.
// pshift > 0: rows are voxel grids of resolution 2^pshift; stage them padded (see combine()).
template <class P, class XF = XfNone>
int launch_gather(const P &p, const float *src, float *dst, int B, int C, int L, int J, bool vec_ok,
                  hipStream_t s, const char *what, const XF &xf = XF{}, int pshift = 0, const float *addend = nullptr) {
  if (B == 0 || C == 0 || J == 0) return 0;
  if (pshift > 0 && (size_t)(L + (L >> pshift)) * sizeof(float) > (size_t)kLdsBytesPerCU) pshift = 0;   // padding must not cost the LDS path
  SlabPlan pl = plan_slab(B, C, L, pshift);
  if constexpr (P::kGridRows) {
    // Small grids whose rows are exactly 1024 quads (R = 16) with every point of the cloud resident in ONE 1024-thread workgroup: the
    // generic plan above picks 256-thread single-row slabs, whose 16 points per thread re-derive their corner taps from the
    // coordinates for EVERY channel (a (16,64,4096,16) devoxelization: 25 us for 42 MB, 0.27 of the HBM peak -- latency, not
    // bandwidth).  gather_lds_pipe_rows_kernel: G rows per slab, taps packed once per workgroup, SEQ slabs with the next one in flight.
    static const bool pipe4 = [] { const char *e = getenv("PVCNN_GATHER_PIPE"); return !(e && e[0] == '0'); }();
    if (pipe4 && pl.lds && pshift > 0 && vec_ok && J <= 1024 * 4 && J > 1024 && (L >> 2) == 1024 && aligned16(src) && C >= 4) {
      constexpr int GR = 2;                                   // rows per slab (4 rows spill 70 registers at 1024 threads / 128 VGPRs)
      const int slabs = ceil_div(C, GR);
      const int seq = (int)std::min<long>(8, std::max<long>(1, (long)B * slabs / kNumCU));
      const size_t bytes = (size_t)GR * (L + (L >> pshift)) * sizeof(float);
      auto k = gather_lds_pipe_rows_kernel<P, XF, GR>;
      if (int e = enable_big_lds(k, bytes)) { set_error("%s: LDS attribute: %d", what, e); return e; }
      hipLaunchKernelGGL(k, dim3(ceil_div(slabs, seq), B), dim3(1024), bytes, s, p, xf, src, dst, C, L, J, seq, pshift, addend);
      return check_launch(what);
    }
  }
  if (!pl.lds) {
    if (!XF::kIdentity || addend) {
      set_error("%s: row does not fit LDS; the fused transform / addend needs the LDS path", what);
      return PVCNN_ERR_INVALID_ARGUMENT;
    } else if constexpr (XF::kIdentity) {
      const int CT = 16;
      hipLaunchKernelGGL((gather_direct_kernel<P>), dim3(ceil_div(J, 256), ceil_div(C, CT), B), dim3(256), 0, s,
                         p, src, dst, C, L, J, CT);
      return check_launch(what);
    }
  }
  const dim3 grid(ceil_div(ceil_div(C, pl.G), pl.seq), B);
  if constexpr (P::kGridRows) {
    // software-pipelined single-row-slab variant (see gather_lds_pipe_kernel); PVCNN_GATHER_PIPE=0 opts out (read once per process)
    static const bool pipe = [] { const char *e = getenv("PVCNN_GATHER_PIPE"); return !(e && e[0] == '0'); }();
    if (pipe && pl.threads == 1024 && pl.G == 1 && pshift > 0 && vec_ok && J <= 1024 * 4 && (L & 3) == 0 && (L >> 2) <= 1024 * 8 &&
        aligned16(src) && (((size_t)L * sizeof(float)) & 15) == 0) {
      auto k = gather_lds_pipe_kernel<P, XF>;
      if (int e = enable_big_lds(k, pl.bytes)) { set_error("%s: LDS attribute: %d", what, e); return e; }
      hipLaunchKernelGGL(k, grid, dim3(1024), pl.bytes, s, p, xf, src, dst, C, L, J, pl.seq, pshift, addend);
      return check_launch(what);
    }
  }
#define PVCNN_LAUNCH_GATHER_P(VEC, T, PADV)                                                      \
  do {                                                                                           \
    auto k = (J <= T * VEC) ? gather_lds_kernel<P, VEC, T, true, XF, PADV> : gather_lds_kernel<P, VEC, T, false, XF, PADV>; \
    if (int e = enable_big_lds(k, pl.bytes)) { set_error("%s: LDS attribute: %d", what, e); return e; } \
    hipLaunchKernelGGL(k, grid, dim3(T), pl.bytes, s, p, xf, src, dst, C, L, J, pl.G, pl.seq, pshift, addend); \
  } while (0)
#define PVCNN_LAUNCH_GATHER(VEC, T)                                                              \
  do {                                                                                           \
    if constexpr (P::kGridRows) { if (pshift > 0) PVCNN_LAUNCH_GATHER_P(VEC, T, true); else PVCNN_LAUNCH_GATHER_P(VEC, T, false); } \
    else PVCNN_LAUNCH_GATHER_P(VEC, T, false);                                                   \
  } while (0)
  if (pl.threads == 1024) { if (vec_ok) PVCNN_LAUNCH_GATHER(4, 1024); else PVCNN_LAUNCH_GATHER(1, 1024); }
  else                    { if (vec_ok) PVCNN_LAUNCH_GATHER(4, 256);  else PVCNN_LAUNCH_GATHER(1, 256); }
#undef PVCNN_LAUNCH_GATHER
#undef PVCNN_LAUNCH_GATHER_P
  return check_launch(what);
}

// Atomic fallback for scatters whose target row does not fit the CSR histogram in LDS
// (csr.h, L > kCsrMaxTargets): memset + global_atomic_add_f32, like the reference.
template <class P>
int launch_scatter_direct(const P &p, const float *src, float *dst, int B, int C, int L, int J, hipStream_t s,
                          const char *what) {
  if (B == 0 || C == 0 || L == 0) return 0;
  hipError_t e = hipMemsetAsync(dst, 0, (size_t)B * C * L * sizeof(float), s);
  if (e != hipSuccess) { set_error("%s: memset: %s", what, hipGetErrorString(e)); return (int)e; }
  if (J == 0) return 0;
  const int CT = 16;
  hipLaunchKernelGGL((scatter_direct_kernel<P>), dim3(ceil_div(J, 256), ceil_div(C, CT), B), dim3(256), 0, s,
                     p, src, dst, C, L, J, CT);
  return check_launch(what);
}

}  // namespace pvcnn
