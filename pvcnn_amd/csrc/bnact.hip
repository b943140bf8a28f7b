// bnact.hip -- BatchNorm (training or eval) fused with the ReLU / LeakyReLU that follows it.
//
// Reference: every conv of the hot path is followed by BatchNorm + activation as separate modules
// (modules/pvconv.py:20-27: BatchNorm3d(eps=1e-4) + LeakyReLU(0.1); modules/shared_mlp.py:20-25:
// BatchNorm + ReLU), i.e. cuDNN BN plus an extra elementwise pass forward and backward.  On a
// (16, 64, 32^3) grid every pass is 134 MB each way, so the activation is folded into the BN passes:
//   forward : bn_stats (1 read)  + bnact_apply  (1 read, 1 write)          y = act(g*(x-m)*rstd + b)
//   backward: bnact_bwd_reduce (2 reads) + bnact_bwd_apply (2 reads, 1 write)
// Tensors are (B, C, S) channel-major (S = R^3 voxels or N points); statistics per channel over B*S.
// Sums are accumulated in fp32 per workgroup slice (<= 8192 elements) and combined in fp64.
#include <algorithm>

#include "common.h"

namespace pvcnn {

constexpr int kBnThreads = 256;
constexpr int kBnSlice = 8192;   // elements of one (b, c) row handled by one workgroup

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// block-wide sum of two values; result valid in thread 0
__device__ __forceinline__ void block_sum2(float &a, float &b, float *sm) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sm[wave] = a; sm[8 + wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = sm[0] + sm[1] + sm[2] + sm[3];
    b = sm[8] + sm[9] + sm[10] + sm[11];
  }
}

// Variance as E[(x-K)^2] - E[x-K]^2 with a per-channel shift K close to the mean: the plain E[x^2] - E[x]^2 loses
// all digits in fp32 partial sums once |mean| / std reaches ~1e3 (sum of squares rounded at 6e-8 * n * mean^2).
// K = the channel's first element here; = the convolution's bias for the epilogue statistics (conv3d.hip, pointwise.hip).
// grid = (slices, B, C): partial (sum, sum of squares) of (x - K) over one slice of row (b, c)
__global__ __launch_bounds__(kBnThreads) void bn_stats_kernel(const float *__restrict__ x, int C, int S, int slices,
                                                             float2 *__restrict__ part, float *__restrict__ shift) {
  __shared__ float sm[16];
  const int sl = blockIdx.x, b = blockIdx.y, c = blockIdx.z;
  const float *row = x + ((size_t)b * C + c) * S;
  const float K = x[(size_t)c * S];                      // first element of the channel (cloud 0): same K in every workgroup
  if (sl == 0 && b == 0 && threadIdx.x == 0) shift[c] = K;
  const int lo = sl * kBnSlice, hi = min(S, lo + kBnSlice);
  float s = 0.f, q = 0.f;
  if ((S & 3) == 0 && aligned16(row)) {
    for (int i = lo + threadIdx.x * 4; i < hi; i += kBnThreads * 4) {
      float4 v = *reinterpret_cast<const float4 *>(row + i);
      v.x -= K; v.y -= K; v.z -= K; v.w -= K;
      s += (v.x + v.y) + (v.z + v.w);
      q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += kBnThreads) { const float v = row[i] - K; s += v; q += v * v; }
  }
  block_sum2(s, q, sm);
  if (threadIdx.x == 0) part[((size_t)c * gridDim.y + b) * slices + sl] = make_float2(s, q);
}

// grid = C: combine the partials (of x - shift[c]; shift may be null = 0) in fp64 -> mean, rstd; update the running
// statistics (unbiased var)
__global__ __launch_bounds__(64) void bn_finalize_kernel(const float2 *__restrict__ part, int nparts, double count, float eps,
                                                        float momentum, const float *__restrict__ shift, float *__restrict__ mean,
                                                        float *__restrict__ rstd, float *__restrict__ running_mean,
                                                        float *__restrict__ running_var, uint32_t *__restrict__ zero_word,
                                                        long zero_count, long long *__restrict__ counter) {
  const int c = blockIdx.x;
  // arms the amax buffer the apply pass will fill by atomic maxima (word [0] and, when that pass splits the channels over several
  // workgroups, the table behind it)
  if (zero_word != nullptr)
    for (long i = (long)c * 64 + threadIdx.x; i < zero_count; i += (long)gridDim.x * 64) zero_word[i] = 0u;
  if (counter != nullptr && c == 0 && threadIdx.x == 0) *counter += 1;       // BatchNorm's num_batches_tracked (one launch less per layer)
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) { const float2 p = part[(size_t)c * nparts + i]; s += p.x; q += p.y; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { s += __shfl_xor(s, d); q += __shfl_xor(q, d); }
  if (threadIdx.x == 0) {
    const double ms = s / count;                          // mean of (x - shift)
    const double m = ms + (shift ? (double)shift[c] : 0.0);
    double var = q / count - ms * ms;
    var = var < 0.0 ? 0.0 : var;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
  }
}

// grid = (slices, B, C): y = act(scale * x + shift), scale = gamma*rstd, shift = beta - mean*scale
__global__ __launch_bounds__(kBnThreads) void bnact_apply_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                                const float *__restrict__ rstd,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, float slope, int C, int S,
                                                                float *__restrict__ y) {
  const int sl = blockIdx.x, b = blockIdx.y, c = blockIdx.z;
  const float scale = (gamma ? gamma[c] : 1.0f) * rstd[c];
  const float shift = (beta ? beta[c] : 0.0f) - mean[c] * scale;
  const size_t off = ((size_t)b * C + c) * S;
  const int lo = sl * kBnSlice, hi = min(S, lo + kBnSlice);
  if ((S & 3) == 0 && aligned16(x + off) && aligned16(y + off)) {
    for (int i = lo + threadIdx.x * 4; i < hi; i += kBnThreads * 4) {
      float4 v = *reinterpret_cast<const float4 *>(x + off + i);
      v.x = fmaf(v.x, scale, shift); v.y = fmaf(v.y, scale, shift); v.z = fmaf(v.z, scale, shift); v.w = fmaf(v.w, scale, shift);
      v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
      v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
      // streaming store: the activated tensor is consumed by the NEXT kernel (conv staging or the devoxelize
      // gather); left dirty in L2 it makes that kernel's reads wait on the write-back (gather: 1.5x slower)
      using v4f = __attribute__((ext_vector_type(4))) float;
      v4f o = {v.x, v.y, v.z, v.w};
      __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(y + off + i));
    }
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += kBnThreads) {
      const float v = fmaf(x[off + i], scale, shift);
      y[off + i] = v > 0.f ? v : v * slope;
    }
  }
}

// ---- nn.Dropout behind a BatchNorm + ReLU pair (the classifier heads: models/utils.py:15-36, `[SharedMLP, Dropout(p), ...]`) fused
// into the pair's passes: y = keep ? act(bn(x)) / (1 - p) : 0 in the apply pass, grad' = keep ? grad / (1 - p) : 0 on load in the two
// backward passes.  keep(e) of element e of the (B, C, S) tensor is a pure function of (e, key): 16 bits of a 32-bit integer mixer
// (two multiply-xorshift rounds) over (e >> 1) ^ key -- recomputed wherever it is needed, never stored; the key is one int64 in
// device memory drawn by the caller per forward call (torch's generator: seeded by torch.manual_seed, graph-capture safe).
// As torch ops the dropout is a read + write of the activated tensor forward and again backward (+ a mask): 0.15 ms per PVCNN step.
struct Drop { const unsigned long long *key; uint32_t thr; float scale; };   // keep iff random16 >= thr; thr = round(p * 65536)

__device__ __forceinline__ uint32_t drop_mix(uint32_t pair, uint32_t k0, uint32_t k1) {
  uint32_t x = pair ^ k0;
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x += k1; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// factors (scale or 0) of the four consecutive elements e0 .. e0 + 3, e0 % 4 == 0
__device__ __forceinline__ void drop_factors4(size_t e0, uint32_t k0, uint32_t k1, uint32_t thr, float scale, float (&f)[4]) {
  const uint32_t h0 = drop_mix((uint32_t)(e0 >> 1), k0, k1), h1 = drop_mix((uint32_t)(e0 >> 1) + 1u, k0, k1);
  f[0] = (h0 & 0xffffu) >= thr ? scale : 0.0f; f[1] = (h0 >> 16) >= thr ? scale : 0.0f;
  f[2] = (h1 & 0xffffu) >= thr ? scale : 0.0f; f[3] = (h1 >> 16) >= thr ? scale : 0.0f;
}
__device__ __forceinline__ float drop_factor1(size_t e, uint32_t k0, uint32_t k1, uint32_t thr, float scale) {
  const uint32_t h = drop_mix((uint32_t)(e >> 1), k0, k1);
  return ((e & 1) ? (h >> 16) : (h & 0xffffu)) >= thr ? scale : 0.0f;
}

__global__ __launch_bounds__(256) void dropout_keep_mask_kernel(Drop d, long numel, unsigned char *__restrict__ keep) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long key = *d.key;
  if (e < numel) keep[e] = drop_factor1((size_t)e, (uint32_t)key, (uint32_t)(key >> 32), d.thr, 1.0f) != 0.0f;
}

// What bnact_bwd_finalize_kernel does, as the tail of the reduce kernel (tickets != NULL): the workgroup that writes the LAST partial of
// channel c (one ticket word per channel; common.h: ticket_take) combines that channel's partials -- the same fp64 loop and
// shuffle tree over 64 lanes, so dgamma / dbeta are the same bits -- and every workgroup zeroes its share of grad_x's amax buffer.
// Fifteen launches of ~5 us less per PVCNN step; the channels finish spread over the last sample's pass (sample-major traversal).
struct BwdFold {
  unsigned *tickets;     // C words, zero before and after the launch; NULL: the separate finalize kernel follows
  float *dgamma, *dbeta;
  uint32_t *zero_word;
  long zero_count;
};

// grid = (slices, B, C): partial (sum g', sum g' * xhat),  g' = gy * act'(z),  z = scale*x + shift
template <bool DROP = false>
__global__ __launch_bounds__(kBnThreads) void bnact_bwd_reduce_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                     const float *__restrict__ mean,
                                                                     const float *__restrict__ rstd,
                                                                     const float *__restrict__ gamma,
                                                                     const float *__restrict__ beta, float slope, int C,
                                                                     int S, int slices, float2 *__restrict__ part,
                                                                     long gy_bstride, Drop drop = Drop{nullptr, 0u, 1.0f},
                                                                     BwdFold fold = BwdFold{nullptr, nullptr, nullptr, nullptr, 0L}) {
  __shared__ float sm[16];
  uint32_t k0 = 0u, k1 = 0u;
  if constexpr (DROP) { const unsigned long long key = *drop.key; k0 = (uint32_t)key; k1 = (uint32_t)(key >> 32); }
  // SAMPLE-MAJOR traversal (the grid says (slices, B, C), dispatch runs x fastest): consecutive workgroups take the slices of one
  // channel, then the next channel of the SAME sample, so the chip reads one sample's C * S contiguous floats of x and grad_y at a
  // time.  In grid order (every sample of channel c, then channel c + 1) the workgroups in flight read B short runs C * S floats
  // apart: 0.43 -> 0.37 ms over the 15 launches of a PVCNN step, the 1024-channel tensor 104 -> 95 us (profiles/ab/r04m).
  int lin = blockIdx.x + (int)gridDim.x * (blockIdx.y + (int)gridDim.y * blockIdx.z);
  const int sl = lin % (int)gridDim.x; lin /= (int)gridDim.x;
  const int c = lin % (int)gridDim.z, b = lin / (int)gridDim.z;
  const float m = mean[c], r = rstd[c];
  const float scale = (gamma ? gamma[c] : 1.0f) * r;
  const float shift = (beta ? beta[c] : 0.0f) - m * scale;
  const size_t off = ((size_t)b * C + c) * S;
  const size_t goff = (size_t)b * gy_bstride + (size_t)c * S;   // grad_y: channels of a cloud contiguous, clouds gy_bstride apart
  const int lo = sl * kBnSlice, hi = min(S, lo + kBnSlice);
  float s = 0.f, q = 0.f;
  if ((S & 3) == 0 && aligned16(x + off) && (!gy || aligned16(gy + goff))) {
    for (int i = lo + threadIdx.x * 4; i < hi; i += kBnThreads * 4) {
      const float4 xv = *reinterpret_cast<const float4 *>(x + off + i);
      const float4 gv = gy ? *reinterpret_cast<const float4 *>(gy + goff + i) : make_float4(1.0f, 1.0f, 1.0f, 1.0f);
      const float xs_[4] = {xv.x, xv.y, xv.z, xv.w};
      float gs_[4] = {gv.x, gv.y, gv.z, gv.w};
      if constexpr (DROP) {
        float f[4];
        drop_factors4(off + i, k0, k1, drop.thr, drop.scale, f);
#pragma unroll
        for (int u = 0; u < 4; ++u) gs_[u] *= f[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float z = fmaf(xs_[u], scale, shift);
        const float g = gs_[u] * (z > 0.f ? 1.0f : slope);
        s += g;
        q += g * ((xs_[u] - m) * r);
      }
    }
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += kBnThreads) {
      const float xv = x[off + i];
      const float z = fmaf(xv, scale, shift);
      float gin = gy ? gy[goff + i] : 1.0f;
      if constexpr (DROP) gin *= drop_factor1(off + i, k0, k1, drop.thr, drop.scale);
      const float g = gin * (z > 0.f ? 1.0f : slope);
      s += g;
      q += g * ((xv - m) * r);
    }
  }
  block_sum2(s, q, sm);
  float2 *mine = part + ((size_t)c * gridDim.y + b) * slices + sl;
  if (fold.tickets == nullptr) {                               // (uniform over the launch)
    if (threadIdx.x == 0) *mine = make_float2(s, q);
    return;
  }
  // (published below by wave 0; the other waves are done)
  if (fold.zero_word != nullptr) {                             // arms grad_x's amax buffer for the apply pass (a later launch)
    const long wgs = (long)gridDim.x * gridDim.y * gridDim.z;
    const long me = blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z);
    for (long i = me * kBnThreads + threadIdx.x; i < fold.zero_count; i += wgs * kBnThreads) fold.zero_word[i] = 0u;
  }
  if (threadIdx.x >= 64) return;                               // wave 0 alone: publish, ticket, and -- for the last one -- the combine
  if (threadIdx.x == 0)
    publish64(reinterpret_cast<unsigned long long *>(mine), ((unsigned long long)__float_as_uint(q) << 32) | __float_as_uint(s));
  if (!ticket_take_wave(fold.tickets + c, (unsigned)(gridDim.x * gridDim.y))) return;
  {                                                            // bnact_bwd_finalize_kernel's combine, operation for operation
    const int nparts = (int)(gridDim.x * gridDim.y);
    double ds = 0.0, dq = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) {
      const unsigned long long w = peek64(reinterpret_cast<const unsigned long long *>(part + (size_t)c * nparts + i));
      ds += __uint_as_float((uint32_t)w); dq += __uint_as_float((uint32_t)(w >> 32));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { ds += __shfl_xor(ds, d); dq += __shfl_xor(dq, d); }
    if (threadIdx.x == 0) { fold.dbeta[c] = (float)ds; fold.dgamma[c] = (float)dq; }
  }
}

// grid = C: dbeta = sum g', dgamma = sum g' xhat (fp64 combine)
__global__ __launch_bounds__(64) void bnact_bwd_finalize_kernel(const float2 *__restrict__ part, int nparts,
                                                               float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                               uint32_t *__restrict__ zero_word, long zero_count) {
  const int c = blockIdx.x;
  if (zero_word != nullptr)                                     // arms grad_x's amax buffer (apply pass), see bn_finalize_kernel
    for (long i = (long)c * 64 + threadIdx.x; i < zero_count; i += (long)gridDim.x * 64) zero_word[i] = 0u;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) { const float2 p = part[(size_t)c * nparts + i]; s += p.x; q += p.y; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { s += __shfl_xor(s, d); q += __shfl_xor(q, d); }
  if (threadIdx.x == 0) { dbeta[c] = (float)s; dgamma[c] = (float)q; }
}

// grid = (slices, B, C): training: gx = gamma*rstd*(g' - dbeta/M - xhat*dgamma/M); eval: gx = gamma*rstd*g'
__global__ __launch_bounds__(kBnThreads) void bnact_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                    const float *__restrict__ mean,
                                                                    const float *__restrict__ rstd,
                                                                    const float *__restrict__ gamma,
                                                                    const float *__restrict__ beta,
                                                                    const float *__restrict__ dgamma,
                                                                    const float *__restrict__ dbeta, float slope,
                                                                    float inv_count, int training, int C, int S,
                                                                    float *__restrict__ gx, long gy_bstride) {
  const int sl = blockIdx.x, b = blockIdx.y, c = blockIdx.z;
  const float m = mean[c], r = rstd[c];
  const float gmm = gamma ? gamma[c] : 1.0f;
  const float scale = gmm * r;
  const float shift = (beta ? beta[c] : 0.0f) - m * scale;
  const float db = training ? dbeta[c] * inv_count : 0.0f, dg = training ? dgamma[c] * inv_count : 0.0f;
  const size_t off = ((size_t)b * C + c) * S;
  const size_t goff = (size_t)b * gy_bstride + (size_t)c * S;
  const int lo = sl * kBnSlice, hi = min(S, lo + kBnSlice);
  if ((S & 3) == 0 && aligned16(x + off) && aligned16(gy + goff) && aligned16(gx + off)) {
    for (int i = lo + threadIdx.x * 4; i < hi; i += kBnThreads * 4) {
      const float4 xv = *reinterpret_cast<const float4 *>(x + off + i);
      const float4 gv = *reinterpret_cast<const float4 *>(gy + goff + i);
      const float xs_[4] = {xv.x, xv.y, xv.z, xv.w}, gs_[4] = {gv.x, gv.y, gv.z, gv.w};
      float o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float z = fmaf(xs_[u], scale, shift);
        const float g = gs_[u] * (z > 0.f ? 1.0f : slope);
        o[u] = scale * (g - db - ((xs_[u] - m) * r) * dg);
      }
      using v4f = __attribute__((ext_vector_type(4))) float;
      v4f ov = {o[0], o[1], o[2], o[3]};
      __builtin_nontemporal_store(ov, reinterpret_cast<v4f *>(gx + off + i));   // streaming: read next by another kernel
    }
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += kBnThreads) {
      const float xv = x[off + i];
      const float z = fmaf(xv, scale, shift);
      const float g = gy[goff + i] * (z > 0.f ? 1.0f : slope);
      const float xhat = (xv - m) * r;
      gx[off + i] = scale * (g - db - xhat * dg);
    }
  }
}

// float -> uint32 whose unsigned order is torch.max's order: -0 == +0, numbers by value, every NaN on top (all NaNs equal)
__device__ __forceinline__ uint32_t max_order_bits(float v) {
  uint32_t u = __float_as_uint(v);
  if (u == 0x80000000u) u = 0u;
  return v != v ? 0xffffffffu : (u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u));
}

__global__ __launch_bounds__(256) void row_keys_decode_kernel(const unsigned long long *__restrict__ keys, const float *__restrict__ y, long rows,
                                                              int S, long long *__restrict__ winners, float *__restrict__ values, int C,
                                                              long y_bstride) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const uint32_t k = ~(uint32_t)keys[row];
  winners[row] = (long long)k;
  // (the element itself: a -0 stays a -0); y's samples may be y_bstride elements apart (a channel slice of a wider tensor)
  if (values) values[row] = y[(size_t)(row / C) * y_bstride + (size_t)(row % C) * S + k];
}

// ---- the two apply passes in POSITION-BLOCK-MAJOR form: a workgroup owns <= 256 consecutive positions of one sample and walks ALL
// channels (its four waves take every fourth channel, four rows in flight each), so the maximum over the channels of every position
// segment -- the "amax buffer" of the tensor being written (include/pvcnn_hip.h: the f16x2 scale table of the convolution that
// consumes it) -- falls out of the pass: per-lane maxima of four positions, combined across waves and positions with ds_max_u32,
// every table entry written by exactly one workgroup.  The arithmetic is the expressions of bnact_apply_kernel /
// bnact_bwd_apply_kernel above, element for element (same bits).
// Channel groups (gridDim.z > 1, `cgroup` channels each): at S = 4096 (the R = 16 grids, every per-point tensor) there are only
// B * S / 256 = 256 position blocks, one workgroup per CU with four rows in flight per wave -- 16 KiB in flight per CU, a latency-bound
// 2 TB/s on the 1024-channel tensor.  With the channels split over gridDim.z workgroups (and eight rows in flight per wave) the chip is
// full; the table entries are then maxima over the groups: atomicMax into a table zeroed by the finalize kernel (`table_by_atomic`;
// order-independent, deterministic).
// ROWMAX (forward only): the pass ALSO emits the arg-max of every (sample, channel) row of what it writes -- the global max-pool over the
// points that follows the last point stage (models/s3dis/pvcnn.py:41-43) costs no read of its own.  A wave holds 256 positions of
// one channel: each lane's first maximum of its four values, a butterfly maximum of the VALUE over the wave, and the lowest lane that
// holds it (= the smallest position) sends one 64-bit atomic maximum of (orderable value bits << 32 | ~position) to row_keys[b * C + c]
// (zeroed by the caller).  Largest value, then smallest position: torch.max's winners (-0 == +0, every NaN beats every number --
// pool.hip: pool_better); pvcnn_row_keys_decode turns the keys into indices and reads the values back.
template <bool BWD, bool DROP = false, bool ROWMAX = false>
__global__ __launch_bounds__(256) void bnact_apply_pb_kernel(const float *__restrict__ x, const float *__restrict__ gy, long gy_bstride,
                                                             const float *__restrict__ mean, const float *__restrict__ rstd,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             const float *__restrict__ dgamma, const float *__restrict__ dbeta,
                                                             float slope, float inv_count, int training, int C, int S, int seg,
                                                             int nseg, int vec, float *__restrict__ out, uint32_t *__restrict__ amax,
                                                             int global_by_atomic, const float *__restrict__ bc_mul = nullptr,
                                                             const float *__restrict__ bc_add = nullptr, int cgroup = 0x7fffffff,
                                                             int table_by_atomic = 0, Drop drop = Drop{nullptr, 0u, 1.0f},
                                                             unsigned long long *__restrict__ row_keys = nullptr, long out_bstride = 0) {
  static_assert(!ROWMAX || (!BWD && !DROP), "row maxima: plain forward pass only");
  __shared__ uint32_t seg_max[256];
  // SAMPLE-MAJOR traversal: workgroups are dispatched in linear block order, and consecutive ones take the channel groups of one
  // position block, then the next position block of the same sample -- the chip streams through one sample (C * S contiguous floats)
  // at a time instead of touching every sample once per channel group (measured in the step: -0.02 ms, profiles/ab/r04m)
  int lin = blockIdx.x + (int)gridDim.x * (blockIdx.y + (int)gridDim.y * blockIdx.z);
  const int bz = lin % (int)gridDim.z; lin /= (int)gridDim.z;
  const int bx = lin % (int)gridDim.x, by = lin / (int)gridDim.x;
  uint32_t k0 = 0u, k1 = 0u;
  if constexpr (DROP) { const unsigned long long key = *drop.key; k0 = (uint32_t)key; k1 = (uint32_t)(key >> 32); }
  const int spb = seg >= 256 ? 1 : 256 / seg;                  // whole segments per workgroup (seg <= 256 enforced by the host)
  const int b = by, s0 = bx * spb;
  const int p0 = s0 * seg, span = min(spb * seg, S - p0);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, pos = 4 * lane;
  const int cbeg = bz * min(cgroup, C), cend = min(C, cbeg + min(cgroup, C));
  if (tid < spb) seg_max[tid] = 0u;
  __syncthreads();
  const float *xb = x + (size_t)b * C * S + p0;
  const float *gb = BWD ? gy + (size_t)b * gy_bstride + p0 : nullptr;
  // (out_bstride: the output's samples that many elements apart -- the pass writes INTO a channel slice of a wider tensor, the
  //  concatenation in front of the classifier: pvcnn_bnact_apply_rowmax; 0 = C * S)
  float *ob = out + (size_t)b * (out_bstride > 0 ? (size_t)out_bstride : (size_t)C * S) + p0;
  uint32_t mx[4] = {0u, 0u, 0u, 0u};
  struct Par { float scale, shift, m, r, db, dg, gmul, gadd; };
  auto par_of = [&](int c) {
    Par p;
    p.m = mean[c]; p.r = rstd[c];
    p.gmul = (BWD && bc_mul) ? bc_mul[(size_t)b * C + c] : 1.0f;          // SE tail: g' = (grad_y * mul + add) * act'(z)
    p.gadd = (BWD && bc_add) ? bc_add[(size_t)b * C + c] : 0.0f;
    p.scale = (gamma ? gamma[c] : 1.0f) * p.r;
    p.shift = (beta ? beta[c] : 0.0f) - p.m * p.scale;
    p.db = (BWD && training) ? dbeta[c] * inv_count : 0.0f;
    p.dg = (BWD && training) ? dgamma[c] * inv_count : 0.0f;
    return p;
  };
  // kf: the dropout factor of the element (1 / (1 - p) or 0; DROP only): on the OUTPUT forward, on the incoming gradient backward
  auto one = [&](float xv, float gv, const Par &p, float kf) {
    if constexpr (BWD) {
      const float z = fmaf(xv, p.scale, p.shift);
      float gin = (bc_mul || bc_add) ? fmaf(gv, p.gmul, p.gadd) : gv;
      if constexpr (DROP) gin *= kf;
      const float g = gin * (z > 0.f ? 1.0f : slope);
      return p.scale * (g - p.db - ((xv - p.m) * p.r) * p.dg);
    } else {
      const float v = fmaf(xv, p.scale, p.shift);
      const float a = v > 0.f ? v : v * slope;
      if constexpr (DROP) return a * kf; else return a;
    }
  };
  const size_t ebase = (size_t)b * C * S + p0;                  // linear index of (b, channel 0, position p0)
  if (pos < span) {
    if (vec) {
      using v4f = __attribute__((ext_vector_type(4))) float;
      constexpr int RF = 8;                                    // rows in flight per wave: channels c0, c0 + 4, ..., c0 + 4 (RF - 1)
      for (int c0 = cbeg + wave; c0 < cend; c0 += 4 * RF) {
        float4 xv[RF], gv[RF];
#pragma unroll
        for (int u = 0; u < RF; ++u) {
          const int c = c0 + 4 * u;
          xv[u] = gv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (c < cend) {
            xv[u] = *reinterpret_cast<const float4 *>(xb + (size_t)c * S + pos);
            if constexpr (BWD) gv[u] = *reinterpret_cast<const float4 *>(gb + (size_t)c * S + pos);
          }
        }
#pragma unroll
        for (int u = 0; u < RF; ++u) {
          const int c = c0 + 4 * u;
          if (c < cend) {
            const Par p = par_of(c);
            float kf[4] = {1.0f, 1.0f, 1.0f, 1.0f};
            if constexpr (DROP) drop_factors4(ebase + (size_t)c * S + pos, k0, k1, drop.thr, drop.scale, kf);
            const float o0 = one(xv[u].x, gv[u].x, p, kf[0]), o1 = one(xv[u].y, gv[u].y, p, kf[1]), o2 = one(xv[u].z, gv[u].z, p, kf[2]),
                        o3 = one(xv[u].w, gv[u].w, p, kf[3]);
            mx[0] = max(mx[0], __float_as_uint(fabsf(o0))); mx[1] = max(mx[1], __float_as_uint(fabsf(o1)));
            mx[2] = max(mx[2], __float_as_uint(fabsf(o2))); mx[3] = max(mx[3], __float_as_uint(fabsf(o3)));
            v4f ov = {o0, o1, o2, o3};
            __builtin_nontemporal_store(ov, reinterpret_cast<v4f *>(ob + (size_t)c * S + pos));   // streaming: read next by another kernel
            if constexpr (ROWMAX) {                              // (host: span == 256, every lane is here)
              const uint32_t e0 = max_order_bits(o0), e1 = max_order_bits(o1), e2 = max_order_bits(o2), e3 = max_order_bits(o3);
              uint32_t bv = e0;
              int bi = 0;
              if (e1 > bv) { bv = e1; bi = 1; }
              if (e2 > bv) { bv = e2; bi = 2; }
              if (e3 > bv) { bv = e3; bi = 3; }
              uint32_t wv = bv;
#pragma unroll
              for (int o = 32; o > 0; o >>= 1) wv = max(wv, (uint32_t)__shfl_xor((int)wv, o));
              const unsigned long long holders = __ballot(bv == wv);
              if (lane == __ffsll((long long)holders) - 1)
                atomicMax(&row_keys[(size_t)b * C + c], ((unsigned long long)wv << 32) | (unsigned long long)(~(uint32_t)(p0 + pos + bi)));
            }
          }
        }
      }
    } else {
      for (int c = cbeg + wave; c < cend; c += 4) {
        const Par p = par_of(c);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (pos + i < span) {
            const float kf = DROP ? drop_factor1(ebase + (size_t)c * S + pos + i, k0, k1, drop.thr, drop.scale) : 1.0f;
            const float o = one(xb[(size_t)c * S + pos + i], BWD ? gb[(size_t)c * S + pos + i] : 0.0f, p, kf);
            ob[(size_t)c * S + pos + i] = o;
            mx[i] = max(mx[i], __float_as_uint(fabsf(o)));
          }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (pos + i < span) atomicMax(&seg_max[(pos + i) / seg], mx[i]);
  }
  lds_barrier();                                                // LDS only: the streaming stores above need not have drained
  if (tid < spb && s0 + tid < nseg) {
    if (table_by_atomic) { if (seg_max[tid] != 0u) atomicMax(&amax[1 + (size_t)b * nseg + s0 + tid], seg_max[tid]); }
    else amax[1 + (size_t)b * nseg + s0 + tid] = seg_max[tid];
  }
  // word [0], the tensor's maximum: zeroed by the finalize kernel that ran before this pass -> one fire-and-forget atomic per
  // workgroup (order-independent: deterministic); otherwise the host launches the table reduction behind this kernel
  if (global_by_atomic && tid == 0) {
    uint32_t m = 0;
    for (int i = 0; i < spb; ++i) m = max(m, seg_max[i]);
    if (m != 0) atomicMax(amax, m);
  }
}

}  // namespace pvcnn

using namespace pvcnn;

// channel groups of the position-block-major apply pass (see the kernel): enough workgroups for ~4 per CU, >= 32 channels each
// The one statement of the segment-length contract of every entry that emits an amax buffer (include/pvcnn_hip.h, "amax_seg").
static const char kAmaxSegRange[] = "amax_seg must be in 1..256";
static bool amax_seg_in_range(int seg) { return seg > 0 && seg <= 256; }

static int pb_channel_groups(long position_blocks, int C) {
  int g = 1;
  while (position_blocks * g < 4 * kNumCU && C / (2 * g) >= 32) g *= 2;
  return g;
}

// host side of the fused dropout: p in [0, 1) -> threshold on 16 random bits + the survivors' scale; p == 0 or no seed: off
static bool make_drop(const void *drop_seed, float drop_p, Drop *d) {
  if (drop_seed == nullptr || !(drop_p > 0.0f)) return false;
  long t = lrintf(drop_p * 65536.0f);
  d->key = static_cast<const unsigned long long *>(drop_seed);
  d->thr = (uint32_t)(t < 0 ? 0 : t > 65536 ? 65536 : t);
  d->scale = 1.0f / (1.0f - drop_p);
  return true;
}

extern "C" int pvcnn_dropout_keep_mask(const void *drop_seed, float drop_p, long numel, unsigned char *keep, void *stream) {
  PVCNN_REQUIRE(drop_seed && keep && numel >= 0 && drop_p >= 0.0f && drop_p < 1.0f, "bad argument");
  if (numel == 0) return 0;
  Drop d{static_cast<const unsigned long long *>(drop_seed), 0u, 1.0f};
  make_drop(drop_seed, drop_p, &d);
  hipLaunchKernelGGL(dropout_keep_mask_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), d, numel, keep);
  return check_launch("dropout_keep_mask");
}

extern "C" size_t pvcnn_bnact_workspace_bytes(int B, int C, int S) {
  if (B <= 0 || C <= 0 || S <= 0) return 16;
  return (size_t)C * B * ceil_div(S, kBnSlice) * sizeof(float2) + (size_t)C * sizeof(float) + 16;   // partials + per-channel shift
}

extern "C" int pvcnn_bnact_fwd(const float *x, const float *gamma, const float *beta, float *running_mean,
                               float *running_var, int B, int C, int S, float eps, float momentum, float slope, int training,
                               float *mean, float *rstd, float *y, void *y_amax, int amax_seg, int amax_zeroed, void *workspace,
                               size_t workspace_bytes, const void *drop_seed, float drop_p, void *stream) {
  PVCNN_REQUIRE(B > 0 && C > 0 && S > 0 && x && y && mean && rstd, "bad argument");
  PVCNN_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "drop_p must be in [0, 1)");
  Drop drop{nullptr, 0u, 1.0f};
  const bool dropping = make_drop(drop_seed, drop_p, &drop);
  PVCNN_REQUIRE(!dropping || (y_amax && (size_t)B * C * S < ((size_t)1 << 33)), "fused dropout needs the amax-emitting pass (y_amax) and < 2^33 elements");
  PVCNN_REQUIRE(B <= 65535 && C <= 65535, "batch or channel count > 65535");
  PVCNN_REQUIRE(!y_amax || amax_seg_in_range(amax_seg), kAmaxSegRange);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int slices = ceil_div(S, kBnSlice);
  const dim3 grid(slices, B, C);
  if (training) {
    PVCNN_REQUIRE(workspace && workspace_bytes >= pvcnn_bnact_workspace_bytes(B, C, S), "workspace too small");
    float2 *part = static_cast<float2 *>(workspace);
    float *shift = reinterpret_cast<float *>(part + (size_t)C * B * slices);
    hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(kBnThreads), 0, s, x, C, S, slices, part, shift);
    if (int e = check_launch("bn_stats")) return e;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, s, part, B * slices, (double)B * S, eps, momentum, shift, mean, rstd,
                       running_mean, running_var, static_cast<uint32_t *>(y_amax), y_amax ? 1 + (long)B * ceil_div(S, amax_seg) : 0L,
                       static_cast<long long *>(nullptr));
    if (int e = check_launch("bn_finalize")) return e;
    amax_zeroed = 1;
  }
  // eval: the caller passes mean = running_mean and rstd = 1/sqrt(running_var + eps)
  if (y_amax != nullptr) {                                      // position-block-major pass that also emits y's amax buffer
    const int nseg = ceil_div(S, amax_seg), spb = amax_seg >= 256 ? 1 : 256 / amax_seg;
    const int vec = (S % 4 == 0) && (amax_seg % 4 == 0) && aligned16(x) && aligned16(y);
    uint32_t *am = static_cast<uint32_t *>(y_amax);
    const int groups = amax_zeroed ? pb_channel_groups((long)ceil_div(nseg, spb) * B, C) : 1;
    if (dropping)
      hipLaunchKernelGGL((bnact_apply_pb_kernel<false, true>), dim3(ceil_div(nseg, spb), B, groups), dim3(256), 0, s, x, nullptr, 0L, mean, rstd,
                         gamma, beta, nullptr, nullptr, slope, 0.0f, 0, C, S, amax_seg, nseg, vec, y, am, amax_zeroed ? 1 : 0, nullptr, nullptr,
                         ceil_div(C, groups), groups > 1 ? 1 : 0, drop);
    else
      hipLaunchKernelGGL(bnact_apply_pb_kernel<false>, dim3(ceil_div(nseg, spb), B, groups), dim3(256), 0, s, x, nullptr, 0L, mean, rstd, gamma,
                         beta, nullptr, nullptr, slope, 0.0f, 0, C, S, amax_seg, nseg, vec, y, am, amax_zeroed ? 1 : 0, nullptr, nullptr,
                         ceil_div(C, groups), groups > 1 ? 1 : 0);
    if (int e = check_launch("bnact_apply_pb")) return e;
    return amax_zeroed ? 0 : launch_amax_reduce(am, (long)B * nseg, s);
  }
  hipLaunchKernelGGL(bnact_apply_kernel, grid, dim3(kBnThreads), 0, s, x, mean, rstd, gamma, beta, slope, C, S, y);
  return check_launch("bnact_apply");
}

// The apply pass alone (statistics known: pvcnn_bn_finalize, which also ZEROED y_amax and row_keys -- its zero_words argument), with the
// row maxima of y: row_keys[b * C + c] (uint64, see the kernel) for pvcnn_row_keys_decode.  S % 256 == 0, amax_seg % 4 == 0, 256 % amax_seg == 0.
extern "C" int pvcnn_bnact_apply_rowmax(const float *x, const float *gamma, const float *beta, const float *mean, const float *rstd, int B,
                                        int C, int S, float slope, float *y, long y_batch_stride, void *y_amax, int amax_seg, void *row_keys,
                                        void *stream) {
  PVCNN_REQUIRE(B > 0 && C > 0 && S > 0 && x && y && mean && rstd && y_amax && row_keys, "bad argument");
  PVCNN_REQUIRE(y_batch_stride == 0 || (y_batch_stride >= (long)C * S && y_batch_stride % 4 == 0), "y_batch_stride: 0, or >= C * S and a multiple of 4");
  PVCNN_REQUIRE(B <= 65535 && C <= 65535, "batch or channel count > 65535");
  // (256 % amax_seg == 0: the row-maximum butterfly assumes every lane of the workgroup's 256 positions is active -- a segment length
  //  that does not divide 256, e.g. 12 -> span 252, would leave lanes out of the __shfl_xor / __ballot: undefined winners)
  PVCNN_REQUIRE(amax_seg_in_range(amax_seg), kAmaxSegRange);
  PVCNN_REQUIRE(amax_seg % 4 == 0 && 256 % amax_seg == 0 && S % 256 == 0,
                "the row-maximum pass additionally needs S % 256 == 0 and amax_seg a multiple of 4 that divides 256");
  PVCNN_REQUIRE(aligned16(x) && aligned16(y) && ((uintptr_t)row_keys & 7) == 0, "x / y must be 16-byte aligned, row_keys 8-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nseg = ceil_div(S, amax_seg), spb = amax_seg >= 256 ? 1 : 256 / amax_seg;
  const int groups = pb_channel_groups((long)ceil_div(nseg, spb) * B, C);
  hipLaunchKernelGGL((bnact_apply_pb_kernel<false, false, true>), dim3(ceil_div(nseg, spb), B, groups), dim3(256), 0, s, x, nullptr, 0L, mean, rstd,
                     gamma, beta, nullptr, nullptr, slope, 0.0f, 0, C, S, amax_seg, nseg, 1, y, static_cast<uint32_t *>(y_amax), 1, nullptr, nullptr,
                     ceil_div(C, groups), groups > 1 ? 1 : 0, Drop{nullptr, 0u, 1.0f}, static_cast<unsigned long long *>(row_keys), y_batch_stride);
  return check_launch("bnact_apply_rowmax");
}

// row_keys (rows uint64) of a (rows, S) tensor y -> winners (int64 position of the row maximum: torch.max's) and, unless NULL, values
extern "C" int pvcnn_row_keys_decode(const void *row_keys, const float *y, long rows, int S, long long *winners, float *values, int C,
                                     long y_batch_stride, void *stream) {
  PVCNN_REQUIRE(rows >= 0 && S > 0, "negative size");
  if (rows == 0) return 0;
  PVCNN_REQUIRE(row_keys && y && winners, "null pointer");
  if (C <= 0 || y_batch_stride <= 0) { C = 1; y_batch_stride = S; }          // rows of a contiguous (rows, S) tensor
  hipLaunchKernelGGL(row_keys_decode_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned long long *>(row_keys), y, rows, S, winners, values, C, y_batch_stride);
  return check_launch("row_keys_decode");
}

// mean / rstd (+ running statistics) from per-workgroup partials produced by a convolution epilogue
// (pvcnn_conv3d_fwd_stats, pvcnn_pwconv_fwd_stats): part is (C, nparts) float2 {sum, sum of squares}.
extern "C" int pvcnn_bn_finalize(const float *part, int C, long nparts, double count, float eps, float momentum, const float *shift,
                                 float *running_mean, float *running_var, float *mean, float *rstd, void *zero_words, long zero_count,
                                 void *num_batches_tracked, void *stream) {
  PVCNN_REQUIRE(C > 0 && nparts > 0 && nparts <= 0x7fffffffL && count > 0 && part && mean && rstd, "bad argument");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float2 *>(part), (int)nparts, count, eps, momentum, shift, mean, rstd, running_mean,
                     running_var, static_cast<uint32_t *>(zero_words), zero_words ? zero_count : 0L,
                     static_cast<long long *>(num_batches_tracked));
  return check_launch("bn_finalize");
}

extern "C" int pvcnn_bn_stats(const float *x, float *running_mean, float *running_var, int B, int C, int S, float eps,
                              float momentum, float *mean, float *rstd, void *workspace, size_t workspace_bytes, void *stream) {
  PVCNN_REQUIRE(B > 0 && C > 0 && S > 0 && x && mean && rstd, "bad argument");
  PVCNN_REQUIRE(B <= 65535 && C <= 65535, "batch or channel count > 65535");
  PVCNN_REQUIRE(workspace && workspace_bytes >= pvcnn_bnact_workspace_bytes(B, C, S), "workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int slices = ceil_div(S, kBnSlice);
  float2 *part = static_cast<float2 *>(workspace);
  float *shift = reinterpret_cast<float *>(part + (size_t)C * B * slices);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(slices, B, C), dim3(kBnThreads), 0, s, x, C, S, slices, part, shift);
  if (int e = check_launch("bn_stats")) return e;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, s, part, B * slices, (double)B * S, eps, momentum, shift, mean, rstd,
                     running_mean, running_var, static_cast<uint32_t *>(nullptr), 0L, static_cast<long long *>(nullptr));
  return check_launch("bn_finalize");
}

static int bnact_bwd_impl(const float *x, const float *grad_y, long gy_bstride, const float *gamma, const float *beta,
                          const float *mean, const float *rstd, int B, int C, int S, float slope, int training, float *grad_x,
                          float *grad_gamma, float *grad_beta, void *workspace, size_t workspace_bytes, void *stream,
                          void *gx_amax = nullptr, int amax_seg = 0, const void *drop_seed = nullptr, float drop_p = 0.0f,
                          void *tickets = nullptr) {
  PVCNN_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "drop_p must be in [0, 1)");
  Drop drop{nullptr, 0u, 1.0f};
  const bool dropping = make_drop(drop_seed, drop_p, &drop);
  PVCNN_REQUIRE(!dropping || (gx_amax && (size_t)B * C * S < ((size_t)1 << 33)), "fused dropout needs the amax-emitting pass (gx_amax) and < 2^33 elements");
  PVCNN_REQUIRE(B > 0 && C > 0 && S > 0 && x && grad_y && mean && rstd && grad_x && grad_gamma && grad_beta, "bad argument");
  PVCNN_REQUIRE(B <= 65535 && C <= 65535, "batch or channel count > 65535");
  PVCNN_REQUIRE(gy_bstride >= (long)C * S, "grad_y batch stride smaller than one sample");
  PVCNN_REQUIRE(workspace && workspace_bytes >= pvcnn_bnact_workspace_bytes(B, C, S), "workspace too small");
  PVCNN_REQUIRE(!gx_amax || amax_seg_in_range(amax_seg), kAmaxSegRange);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int slices = ceil_div(S, kBnSlice);
  const dim3 grid(slices, B, C);
  float2 *part = static_cast<float2 *>(workspace);
  const long zero_count = gx_amax ? 1 + (long)B * ceil_div(S, amax_seg) : 0L;
  // tickets (C zeroed words, left zeroed): the finalize step is the tail of the reduce launch; NULL: a launch of its own
  PVCNN_REQUIRE(!tickets || (reinterpret_cast<uintptr_t>(tickets) & 3) == 0, "tickets must be 4-byte aligned");
  const BwdFold fold{static_cast<unsigned *>(tickets), grad_gamma, grad_beta, static_cast<uint32_t *>(gx_amax), zero_count};
  if (dropping)
    hipLaunchKernelGGL(bnact_bwd_reduce_kernel<true>, grid, dim3(kBnThreads), 0, s, x, grad_y, mean, rstd, gamma, beta, slope, C, S, slices, part,
                       gy_bstride, drop, fold);
  else
    hipLaunchKernelGGL(bnact_bwd_reduce_kernel<false>, grid, dim3(kBnThreads), 0, s, x, grad_y, mean, rstd, gamma, beta, slope, C, S, slices, part,
                       gy_bstride, Drop{nullptr, 0u, 1.0f}, fold);
  if (int e = check_launch("bnact_bwd_reduce")) return e;
  if (tickets == nullptr) {
    hipLaunchKernelGGL(bnact_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, part, B * slices, grad_gamma, grad_beta,
                       static_cast<uint32_t *>(gx_amax), zero_count);
    if (int e = check_launch("bnact_bwd_finalize")) return e;
  }
  const float inv_count = (float)(1.0 / ((double)B * S));
  if (gx_amax != nullptr) {                                     // position-block-major pass that also emits grad_x's amax buffer
    const int nseg = ceil_div(S, amax_seg), spb = amax_seg >= 256 ? 1 : 256 / amax_seg;
    const int vec = (S % 4 == 0) && (amax_seg % 4 == 0) && (gy_bstride % 4 == 0) && aligned16(x) && aligned16(grad_y) && aligned16(grad_x);
    uint32_t *am = static_cast<uint32_t *>(gx_amax);
    const int groups = pb_channel_groups((long)ceil_div(nseg, spb) * B, C);
    if (dropping)
      hipLaunchKernelGGL((bnact_apply_pb_kernel<true, true>), dim3(ceil_div(nseg, spb), B, groups), dim3(256), 0, s, x, grad_y, gy_bstride, mean,
                         rstd, gamma, beta, grad_gamma, grad_beta, slope, inv_count, training, C, S, amax_seg, nseg, vec, grad_x, am, 1, nullptr,
                         nullptr, ceil_div(C, groups), groups > 1 ? 1 : 0, drop);
    else
      hipLaunchKernelGGL(bnact_apply_pb_kernel<true>, dim3(ceil_div(nseg, spb), B, groups), dim3(256), 0, s, x, grad_y, gy_bstride, mean, rstd,
                         gamma, beta, grad_gamma, grad_beta, slope, inv_count, training, C, S, amax_seg, nseg, vec, grad_x, am, 1, nullptr,
                         nullptr, ceil_div(C, groups), groups > 1 ? 1 : 0);
    return check_launch("bnact_bwd_apply_pb");
  }
  hipLaunchKernelGGL(bnact_bwd_apply_kernel, grid, dim3(kBnThreads), 0, s, x, grad_y, mean, rstd, gamma, beta, grad_gamma,
                     grad_beta, slope, inv_count, training, C, S, grad_x, gy_bstride);
  return check_launch("bnact_bwd_apply");
}

extern "C" int pvcnn_bnact_bwd(const float *x, const float *grad_y, const float *gamma, const float *beta, const float *mean,
                               const float *rstd, int B, int C, int S, float slope, int training, float *grad_x,
                               float *grad_gamma, float *grad_beta, void *workspace, size_t workspace_bytes, void *stream) {
  return bnact_bwd_impl(x, grad_y, (long)C * S, gamma, beta, mean, rstd, B, C, S, slope, training, grad_x, grad_gamma, grad_beta,
                        workspace, workspace_bytes, stream);
}

// grad_y may be a channel-slice view of a wider (B, C_total, S) tensor (what torch.cat's backward hands out):
// channels of one sample contiguous, samples grad_y_batch_stride elements apart -- no .contiguous() copy needed.
extern "C" int pvcnn_bnact_bwd_strided(const float *x, const float *grad_y, long grad_y_batch_stride, const float *gamma,
                                       const float *beta, const float *mean, const float *rstd, int B, int C, int S, float slope,
                                       int training, float *grad_x, float *grad_gamma, float *grad_beta, void *gx_amax, int amax_seg,
                                       void *workspace, size_t workspace_bytes, const void *drop_seed, float drop_p, void *tickets,
                                       void *stream) {
  return bnact_bwd_impl(x, grad_y, grad_y_batch_stride, gamma, beta, mean, rstd, B, C, S, slope, training, grad_x, grad_gamma,
                        grad_beta, workspace, workspace_bytes, stream, gx_amax, amax_seg, drop_seed, drop_p, tickets);
}

extern "C" int pvcnn_bnact_slices(int S) { return S > 0 ? ceil_div(S, kBnSlice) : 0; }

extern "C" int pvcnn_bnact_partial_sums(const float *x, const float *grad_y, long grad_y_batch_stride, const float *gamma, const float *beta,
                                        const float *mean, const float *rstd, int B, int C, int S, float slope, float *part, void *stream) {
  PVCNN_REQUIRE(B > 0 && C > 0 && S > 0 && x && mean && rstd && part, "bad argument");
  PVCNN_REQUIRE(B <= 65535 && C <= 65535, "batch or channel count > 65535");
  PVCNN_REQUIRE(!grad_y || grad_y_batch_stride >= (long)C * S, "grad_y batch stride smaller than one sample");
  PVCNN_REQUIRE((reinterpret_cast<uintptr_t>(part) & 7) == 0, "part must be 8-byte aligned");
  const int slices = ceil_div(S, kBnSlice);
  hipLaunchKernelGGL(bnact_bwd_reduce_kernel<false>, dim3(slices, B, C), dim3(kBnThreads), 0, static_cast<hipStream_t>(stream), x, grad_y, mean,
                     rstd, gamma, beta, slope, C, S, slices, reinterpret_cast<float2 *>(part), grad_y ? grad_y_batch_stride : (long)C * S);
  return check_launch("bnact_partial_sums");
}

extern "C" int pvcnn_bnact_bwd_apply(const float *x, const float *grad_y, long grad_y_batch_stride, const float *gamma, const float *beta,
                                     const float *mean, const float *rstd, const float *sum_gamma, const float *sum_beta, const float *bc_mul,
                                     const float *bc_add, int B, int C, int S, float slope, int training, float *grad_x, void *gx_amax,
                                     int amax_seg, void *stream) {
  PVCNN_REQUIRE(B > 0 && C > 0 && S > 0 && x && grad_y && mean && rstd && grad_x, "bad argument");
  PVCNN_REQUIRE(!training || (sum_gamma && sum_beta), "training mode needs the two per-channel sums");
  PVCNN_REQUIRE(B <= 65535 && C <= 65535, "batch or channel count > 65535");
  PVCNN_REQUIRE(grad_y_batch_stride >= (long)C * S, "grad_y batch stride smaller than one sample");
  PVCNN_REQUIRE(!gx_amax || amax_seg_in_range(amax_seg), kAmaxSegRange);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the position-block-major pass (it carries the per-(cloud, channel) factors); without an amax request the table goes to a dummy
  // segmentation of 256 positions that is simply not written out (amax == NULL is not supported by the kernel: give it scratch)
  PVCNN_REQUIRE(gx_amax, "pvcnn_bnact_bwd_apply needs an amax buffer (pvcnn_absmax_tiles_count words)");
  const int nseg = ceil_div(S, amax_seg), spb = amax_seg >= 256 ? 1 : 256 / amax_seg;
  const int vec = (S % 4 == 0) && (amax_seg % 4 == 0) && (grad_y_batch_stride % 4 == 0) && aligned16(x) && aligned16(grad_y) && aligned16(grad_x);
  uint32_t *am = static_cast<uint32_t *>(gx_amax);
  hipLaunchKernelGGL(bnact_apply_pb_kernel<true>, dim3(ceil_div(nseg, spb), B), dim3(256), 0, s, x, grad_y, grad_y_batch_stride, mean, rstd, gamma,
                     beta, sum_gamma, sum_beta, slope, (float)(1.0 / ((double)B * S)), training, C, S, amax_seg, nseg, vec, grad_x, am, 0, bc_mul,
                     bc_add);
  if (int e = check_launch("bnact_bwd_apply_pb")) return e;
  return launch_amax_reduce(am, (long)B * nseg, s);
}

// ---- concatenation of per-point feature maps along the channels (torch.cat(taps, dim=1) of models/s3dis/pvcnn.py:45) -------------
// Position-block-major like the BatchNorm apply passes: a workgroup owns 256 points of one cloud and walks all channels of all
// sources, so the amax buffer of the OUTPUT (the f16x2 scale table of the classifier GEMM that consumes it: one maximum per 256
// points) falls out of the copy -- torch's cat (0.20 ms for PVCNN's 386 MB) plus a separate amax pass (0.07 ms) become one kernel.
// A source may be a broadcast over the points (point stride 0: the cloud descriptor of models/s3dis/pvcnn.py:44).
namespace pvcnn {
constexpr int kCatMaxSrc = 8;
struct CatSources {
  const float *p[kCatMaxSrc];
  long bstride[kCatMaxSrc];     // elements between clouds
  int c0[kCatMaxSrc + 1];       // first output channel of each source (c0[n] = total)
  int pstride[kCatMaxSrc];      // 1: (B, C, N) rows; 0: one value per (cloud, channel), broadcast over the points
  const uint32_t *pre[kCatMaxSrc];   // NULL, or: this source already IS its channel slice of `out` (the pass that produced it wrote it
                                     // there); nothing to copy, and this is its amax buffer (256-point segments) for the output's table
  int n;
};

__global__ __launch_bounds__(256) void concat_points_kernel(CatSources src, int N, int vec, float *__restrict__ out,
                                                            uint32_t *__restrict__ amax, unsigned *__restrict__ ticket) {
  __shared__ uint32_t wave_max[4];
  const int b = blockIdx.y, p0 = blockIdx.x * 256;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, pos = p0 + 4 * lane;
  const int Ctot = src.c0[src.n];
  // channel groups (gridDim.z > 1): with few position blocks (ShapeNet: 8 clouds x 2048 points = 64 workgroups walking ~2700
  // channels each: 1.7 TB/s) the concatenated channel range is split over gridDim.z workgroups; their table entries meet by atomicMax
  // in a table the host zeroed
  const int cgroup = ceil_div(Ctot, (int)gridDim.z), gbeg = (int)blockIdx.z * cgroup, gend = min(Ctot, gbeg + cgroup);
  float *ob = out + (size_t)b * Ctot * N;
  uint32_t m = 0;
  using v4f = __attribute__((ext_vector_type(4))) float;
  for (int s = 0; s < src.n; ++s) {
    const int cb = src.c0[s], lo = max(gbeg, cb) - cb, cn = min(gend, src.c0[s + 1]) - cb;      // this group's rows [lo, cn) of source s
    if (lo >= cn) continue;
    if (src.pre[s] != nullptr) {                             // in place: only its share of the table (a maximum: adding it twice is harmless)
      if (tid == 0 && amax != nullptr) m = max(m, src.pre[s][1 + (size_t)b * gridDim.x + blockIdx.x]);
      continue;
    }
    const float *sp = src.p[s] + (size_t)b * src.bstride[s];
    if (src.pstride[s] == 0) {                               // broadcast rows
      for (int c = lo + wave; c < cn; c += 4) {
        const float v = sp[c];
        m = max(m, __float_as_uint(fabsf(v)));
        if (vec && pos < N) { v4f o = {v, v, v, v}; __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(ob + (size_t)(cb + c) * N + pos)); }
        else for (int i = 0; i < 4; ++i) if (pos + i < N) ob[(size_t)(cb + c) * N + pos + i] = v;
      }
    } else if (vec) {
      if (pos < N)
        for (int c0 = lo + wave; c0 < cn; c0 += 32) {        // eight rows in flight per wave
          float4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (c0 + 4 * u < cn) v[u] = *reinterpret_cast<const float4 *>(sp + (size_t)(c0 + 4 * u) * N + pos);
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (c0 + 4 * u < cn) {
              m = max(max(m, __float_as_uint(fabsf(v[u].x))), max(__float_as_uint(fabsf(v[u].y)), max(__float_as_uint(fabsf(v[u].z)), __float_as_uint(fabsf(v[u].w)))));
              v4f o = {v[u].x, v[u].y, v[u].z, v[u].w};
              __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(ob + (size_t)(cb + c0 + 4 * u) * N + pos));
            }
        }
    } else {
      for (int c = lo + wave; c < cn; c += 4)
        for (int i = 0; i < 4; ++i)
          if (pos + i < N) {
            const float v = sp[(size_t)c * N + pos + i];
            m = max(m, __float_as_uint(fabsf(v)));
            ob[(size_t)(cb + c) * N + pos + i] = v;
          }
    }
  }
  if (amax != nullptr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
    if (lane == 0) wave_max[wave] = m;
    lds_barrier();
    if (tid == 0) {
      const uint32_t t = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
      uint32_t *slot = &amax[1 + (size_t)b * gridDim.x + blockIdx.x];
      if (gridDim.z > 1) {                                   // (a returning atomic when a ticket follows: it has been performed)
        if (ticket != nullptr) { const uint32_t old = atomicMax(slot, t); asm volatile("" ::"v"(old) : "memory"); }
        else if (t != 0u) atomicMax(slot, t);
      } else if (ticket != nullptr) publish32(slot, t);
      else *slot = t;
    }
    // amax[0] by the wave that takes the last ticket (ticket != NULL) instead of a launch of its own (common.h: ticket_take_wave;
    // the host passes a ticket only for tables of <= kFoldTableMax words)
    if (ticket != nullptr && tid < 64 && ticket_take_wave(ticket, gridDim.x * gridDim.y * gridDim.z))
      amax_table_max_wave(amax, (long)gridDim.x * gridDim.y);
  }
}
}  // namespace pvcnn

// out (B, sum C_i, N) = concatenation of nsrc <= 8 sources along the channels; source i: srcs[i] with channels[i] channels, clouds
// bstrides[i] elements apart, pstrides[i] = 1 ((C_i, N) rows contiguous within a cloud) or 0 (one value per (cloud, channel), broadcast).
// out_amax: NULL, or pvcnn_absmax_tiles_count(B, N, 256) words = the amax buffer of `out` with 256-point segments.
extern "C" int pvcnn_concat_points(const float *const *srcs, const long *bstrides, const int *channels, const int *pstrides,
                                   const void *const *src_amax, int nsrc, int B, int N, float *out, void *out_amax, void *ticket,
                                   void *stream) {
  PVCNN_REQUIRE(srcs && bstrides && channels && pstrides && out && nsrc > 0 && nsrc <= kCatMaxSrc && B > 0 && N > 0, "bad argument");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  CatSources cs{};
  cs.n = nsrc;
  bool vec = (N % 4 == 0) && aligned16(out);
  int c = 0;
  for (int i = 0; i < nsrc; ++i) {
    PVCNN_REQUIRE(srcs[i] && channels[i] > 0 && (pstrides[i] == 0 || pstrides[i] == 1), "bad source");
    cs.p[i] = srcs[i]; cs.bstride[i] = bstrides[i]; cs.pstride[i] = pstrides[i]; cs.c0[i] = c;
    cs.pre[i] = (src_amax != nullptr) ? static_cast<const uint32_t *>(src_amax[i]) : nullptr;
    c += channels[i];
    if (pstrides[i] == 1 && cs.pre[i] == nullptr) vec = vec && aligned16(srcs[i]) && (bstrides[i] % 4 == 0);
  }
  cs.c0[nsrc] = c;
  for (int i = 0; i < nsrc; ++i)                               // an in-place source must BE its slice of out
    PVCNN_REQUIRE(cs.pre[i] == nullptr || (srcs[i] == out + (size_t)cs.c0[i] * N && bstrides[i] == (long)c * N && pstrides[i] == 1),
                  "a source with src_amax must already be its channel slice of out (srcs[i] == out + c0 * N, bstride == C_total * N)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int blocks = ceil_div(N, 256);
  int groups = 1;                                            // ~4 workgroups per CU, >= 64 channels each
  while ((long)blocks * B * groups < 4 * kNumCU && c / (2 * groups) >= 64) groups *= 2;
  if (groups > 1 && out_amax != nullptr) {
    hipError_t e = hipMemsetAsync(out_amax, 0, (1 + (size_t)B * blocks) * sizeof(uint32_t), s);
    if (e != hipSuccess) { set_error("concat_points: memset: %s", hipGetErrorString(e)); return (int)e; }
  }
  const bool table_only = ticket == PVCNN_TABLE_ONLY;       // (ABI v12) see pvcnn_absmax_tiles
  if (table_only) ticket = nullptr;
  PVCNN_REQUIRE(!ticket || (reinterpret_cast<uintptr_t>(ticket) & 3) == 0, "ticket must be 4-byte aligned");
  if ((long)B * blocks > kFoldTableMax) ticket = nullptr;    // a long table is read faster by the 1024 threads of the reduce launch
  hipLaunchKernelGGL(concat_points_kernel, dim3(blocks, B, groups), dim3(256), 0, s, cs, N, vec ? 1 : 0, out, static_cast<uint32_t *>(out_amax),
                     static_cast<unsigned *>(out_amax ? ticket : nullptr));
  if (int e = check_launch("concat_points")) return e;
  if (out_amax != nullptr && ticket == nullptr && !table_only) return launch_amax_reduce(static_cast<uint32_t *>(out_amax), (long)B * blocks, s);
  return 0;
}
