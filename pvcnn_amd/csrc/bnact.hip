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
#include <stdlib.h>

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
                                                        float *__restrict__ running_var) {
  const int c = blockIdx.x;
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

// Block-wide max of the bit patterns of non-negative floats, folded into *out (order-independent: deterministic).  Thousands of
// workgroups target ONE word: same-address atomics serialise at the memory side, so a workgroup first looks at the current
// value (a stale read only costs a redundant atomic) and most of them find their maximum already covered.
__device__ __forceinline__ void block_atomic_max_bits(uint32_t m, uint32_t *__restrict__ out) {
  __shared__ uint32_t red[kBnThreads / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < kBnThreads / 64; ++w) m = max(m, red[w]);
    if (m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
  }
}

// EXPERIMENTAL second form (PVCNN_AMAX_REDUCE=2; built, not yet measured): measured on the chip, the form above makes
// bnact_bwd_apply_kernel 27 % slower (34.6 -> 44.1 us) -- __syncthreads() waits for the kernel's outstanding streaming STORES, and
// the filter load is a dependent global round trip at the very end of a ~10 us workgroup.  Here the barrier orders LDS only
// (lds_barrier), the filter value `seen` was loaded when the kernel STARTED (thread 0; stale by design), and the atomic has no
// return value (fire and forget).  Same result: the maximum is order-independent.
__device__ __forceinline__ void block_atomic_max_bits_v2(uint32_t m, uint32_t seen, uint32_t *__restrict__ out) {
  __shared__ uint32_t red2[kBnThreads / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) red2[threadIdx.x >> 6] = m;
  lds_barrier();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < kBnThreads / 64; ++w) m = max(m, red2[w]);
    if (m > seen) atomicMax(out, m);
  }
}

// grid = (slices, B, C): bits of max |act(scale * x + shift)| -- pvcnn_absmax_bits of the tensor bnact_apply_kernel WOULD write
// (same expressions, so the same bits), for consumers that apply the transform while staging (conv3d_bf16.hip, XF)
__global__ __launch_bounds__(kBnThreads) void bnact_absmax_kernel(const float *__restrict__ x, BnActXf xf, int C, int S,
                                                                 uint32_t *__restrict__ out) {
  const int sl = blockIdx.x, b = blockIdx.y, c = blockIdx.z;
  const float2 p = xf.params(c);
  const size_t off = ((size_t)b * C + c) * S;
  const int lo = sl * kBnSlice, hi = min(S, lo + kBnSlice);
  uint32_t m = 0;
  if ((S & 3) == 0 && aligned16(x + off)) {
    for (int i = lo + threadIdx.x * 4; i < hi; i += kBnThreads * 4) {
      const float4 v = *reinterpret_cast<const float4 *>(x + off + i);
      m = max(max(m, __float_as_uint(fabsf(xf.apply(v.x, p)))), __float_as_uint(fabsf(xf.apply(v.y, p))));
      m = max(max(m, __float_as_uint(fabsf(xf.apply(v.z, p)))), __float_as_uint(fabsf(xf.apply(v.w, p))));
    }
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += kBnThreads) m = max(m, __float_as_uint(fabsf(xf.apply(x[off + i], p))));
  }
  block_atomic_max_bits(m, out);
}

// grid = (slices, B, C): partial (sum g', sum g' * xhat),  g' = gy * act'(z),  z = scale*x + shift
__global__ __launch_bounds__(kBnThreads) void bnact_bwd_reduce_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                     const float *__restrict__ mean,
                                                                     const float *__restrict__ rstd,
                                                                     const float *__restrict__ gamma,
                                                                     const float *__restrict__ beta, float slope, int C,
                                                                     int S, int slices, float2 *__restrict__ part,
                                                                     long gy_bstride) {
  __shared__ float sm[16];
  const int sl = blockIdx.x, b = blockIdx.y, c = blockIdx.z;
  const float m = mean[c], r = rstd[c];
  const float scale = (gamma ? gamma[c] : 1.0f) * r;
  const float shift = (beta ? beta[c] : 0.0f) - m * scale;
  const size_t off = ((size_t)b * C + c) * S;
  const size_t goff = (size_t)b * gy_bstride + (size_t)c * S;   // grad_y: channels of a cloud contiguous, clouds gy_bstride apart
  const int lo = sl * kBnSlice, hi = min(S, lo + kBnSlice);
  float s = 0.f, q = 0.f;
  if ((S & 3) == 0 && aligned16(x + off) && aligned16(gy + goff)) {
    for (int i = lo + threadIdx.x * 4; i < hi; i += kBnThreads * 4) {
      const float4 xv = *reinterpret_cast<const float4 *>(x + off + i);
      const float4 gv = *reinterpret_cast<const float4 *>(gy + goff + i);
      const float xs_[4] = {xv.x, xv.y, xv.z, xv.w}, gs_[4] = {gv.x, gv.y, gv.z, gv.w};
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
      const float g = gy[goff + i] * (z > 0.f ? 1.0f : slope);
      s += g;
      q += g * ((xv - m) * r);
    }
  }
  block_sum2(s, q, sm);
  if (threadIdx.x == 0) part[((size_t)c * gridDim.y + b) * slices + sl] = make_float2(s, q);
}

// grid = C: dbeta = sum g', dgamma = sum g' xhat (fp64 combine)
__global__ __launch_bounds__(64) void bnact_bwd_finalize_kernel(const float2 *__restrict__ part, int nparts,
                                                               float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                               uint32_t *__restrict__ gx_absmax) {
  const int c = blockIdx.x;
  if (gx_absmax != nullptr && c == 0 && threadIdx.x == 0) *gx_absmax = 0u;   // re-armed for bnact_bwd_apply_kernel (next launch)
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
                                                                    float *__restrict__ gx, long gy_bstride,
                                                                    uint32_t *__restrict__ gx_absmax, int amax_form) {
  const int sl = blockIdx.x, b = blockIdx.y, c = blockIdx.z;
  uint32_t seen = 0;                                           // form 2: the maximum other workgroups have posted so far
  if (gx_absmax != nullptr && amax_form == 2 && threadIdx.x == 0) seen = __hip_atomic_load(gx_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const float m = mean[c], r = rstd[c];
  const float gmm = gamma ? gamma[c] : 1.0f;
  const float scale = gmm * r;
  const float shift = (beta ? beta[c] : 0.0f) - m * scale;
  const float db = training ? dbeta[c] * inv_count : 0.0f, dg = training ? dgamma[c] * inv_count : 0.0f;
  const size_t off = ((size_t)b * C + c) * S;
  const size_t goff = (size_t)b * gy_bstride + (size_t)c * S;
  const int lo = sl * kBnSlice, hi = min(S, lo + kBnSlice);
  uint32_t amax = 0;                                           // bits of max |gx| over this thread's elements
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
        amax = max(amax, __float_as_uint(fabsf(o[u])));
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
      const float o = scale * (g - db - xhat * dg);
      gx[off + i] = o;
      amax = max(amax, __float_as_uint(fabsf(o)));
    }
  }
  // the gradient's max |.| rides along: the f16x2 products that consume grad_x (Conv3d / 1x1 backward-data and backward-weight)
  // derive their power-of-two scale from it, and a separate absmax pass would re-read the whole tensor
  if (gx_absmax != nullptr) {
    if (amax_form == 2) block_atomic_max_bits_v2(amax, seen, gx_absmax);
    else block_atomic_max_bits(amax, gx_absmax);
  }
}

}  // namespace pvcnn

using namespace pvcnn;

extern "C" size_t pvcnn_bnact_workspace_bytes(int B, int C, int S) {
  if (B <= 0 || C <= 0 || S <= 0) return 16;
  return (size_t)C * B * ceil_div(S, kBnSlice) * sizeof(float2) + (size_t)C * sizeof(float) + 16;   // partials + per-channel shift
}

extern "C" int pvcnn_bnact_fwd(const float *x, const float *gamma, const float *beta, float *running_mean,
                               float *running_var, int B, int C, int S, float eps, float momentum, float slope, int training,
                               float *mean, float *rstd, float *y, void *workspace, size_t workspace_bytes, void *stream) {
  PVCNN_REQUIRE(B > 0 && C > 0 && S > 0 && x && y && mean && rstd, "bad argument");
  PVCNN_REQUIRE(B <= 65535 && C <= 65535, "batch or channel count > 65535");
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
                       running_mean, running_var);
    if (int e = check_launch("bn_finalize")) return e;
  }
  // eval: the caller passes mean = running_mean and rstd = 1/sqrt(running_var + eps)
  hipLaunchKernelGGL(bnact_apply_kernel, grid, dim3(kBnThreads), 0, s, x, mean, rstd, gamma, beta, slope, C, S, y);
  return check_launch("bnact_apply");
}

// mean / rstd (+ running statistics) from per-workgroup partials produced by a convolution epilogue
// (pvcnn_conv3d_fwd_stats, pvcnn_pwconv_fwd_stats): part is (C, nparts) float2 {sum, sum of squares}.
extern "C" int pvcnn_bn_finalize(const float *part, int C, long nparts, double count, float eps, float momentum, const float *shift,
                                 float *running_mean, float *running_var, float *mean, float *rstd, void *stream) {
  PVCNN_REQUIRE(C > 0 && nparts > 0 && nparts <= 0x7fffffffL && count > 0 && part && mean && rstd, "bad argument");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float2 *>(part), (int)nparts, count, eps, momentum, shift, mean, rstd, running_mean,
                     running_var);
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
                     running_mean, running_var);
  return check_launch("bn_finalize");
}

static int bnact_bwd_impl(const float *x, const float *grad_y, long gy_bstride, const float *gamma, const float *beta,
                          const float *mean, const float *rstd, int B, int C, int S, float slope, int training, float *grad_x,
                          float *grad_gamma, float *grad_beta, void *workspace, size_t workspace_bytes, void *stream,
                          void *gx_absmax = nullptr) {
  PVCNN_REQUIRE(B > 0 && C > 0 && S > 0 && x && grad_y && mean && rstd && grad_x && grad_gamma && grad_beta, "bad argument");
  PVCNN_REQUIRE(B <= 65535 && C <= 65535, "batch or channel count > 65535");
  PVCNN_REQUIRE(gy_bstride >= (long)C * S, "grad_y batch stride smaller than one sample");
  PVCNN_REQUIRE(workspace && workspace_bytes >= pvcnn_bnact_workspace_bytes(B, C, S), "workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int slices = ceil_div(S, kBnSlice);
  const dim3 grid(slices, B, C);
  float2 *part = static_cast<float2 *>(workspace);
  hipLaunchKernelGGL(bnact_bwd_reduce_kernel, grid, dim3(kBnThreads), 0, s, x, grad_y, mean, rstd, gamma, beta, slope, C, S,
                     slices, part, gy_bstride);
  if (int e = check_launch("bnact_bwd_reduce")) return e;
  uint32_t *am = static_cast<uint32_t *>(gx_absmax);
  // which form of the in-kernel maximum (see block_atomic_max_bits_v2); read once per process
  static const int amax_form = [] { const char *e = getenv("PVCNN_AMAX_REDUCE"); return (e && e[0] == '2') ? 2 : 1; }();
  hipLaunchKernelGGL(bnact_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, part, B * slices, grad_gamma, grad_beta, am);
  if (int e = check_launch("bnact_bwd_finalize")) return e;
  hipLaunchKernelGGL(bnact_bwd_apply_kernel, grid, dim3(kBnThreads), 0, s, x, grad_y, mean, rstd, gamma, beta, grad_gamma,
                     grad_beta, slope, (float)(1.0 / ((double)B * S)), training, C, S, grad_x, gy_bstride, am, amax_form);
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
                                       int training, float *grad_x, float *grad_gamma, float *grad_beta, void *workspace,
                                       size_t workspace_bytes, void *stream) {
  return bnact_bwd_impl(x, grad_y, grad_y_batch_stride, gamma, beta, mean, rstd, B, C, S, slope, training, grad_x, grad_gamma,
                        grad_beta, workspace, workspace_bytes, stream);
}

// pvcnn_bnact_bwd_strided that also leaves pvcnn_absmax_bits(grad_x) in gx_absmax (one uint32): the f16x2 products consuming
// grad_x need it, and here it costs nothing (no extra pass over the tensor, no memset launch).
extern "C" int pvcnn_bnact_bwd_absmax(const float *x, const float *grad_y, long grad_y_batch_stride, const float *gamma,
                                      const float *beta, const float *mean, const float *rstd, int B, int C, int S, float slope,
                                      int training, float *grad_x, float *grad_gamma, float *grad_beta, void *gx_absmax,
                                      void *workspace, size_t workspace_bytes, void *stream) {
  PVCNN_REQUIRE(gx_absmax != nullptr, "null gx_absmax");
  return bnact_bwd_impl(x, grad_y, grad_y_batch_stride, gamma, beta, mean, rstd, B, C, S, slope, training, grad_x, grad_gamma,
                        grad_beta, workspace, workspace_bytes, stream, gx_absmax);
}

// out[0] = pvcnn_absmax_bits of act(bn(x)) without materialising it: x (B,C,S), per-channel mean / rstd (+ gamma / beta or NULL).
extern "C" int pvcnn_bnact_absmax_bits(const float *x, const float *gamma, const float *beta, const float *mean, const float *rstd,
                                       int B, int C, int S, float slope, void *out, void *stream) {
  PVCNN_REQUIRE(B > 0 && C > 0 && S > 0 && x && mean && rstd && out, "bad argument");
  PVCNN_REQUIRE(B <= 65535 && C <= 65535, "batch or channel count > 65535");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(out, 0, sizeof(uint32_t), s);
  if (e != hipSuccess) { set_error("bnact_absmax: memset: %s", hipGetErrorString(e)); return (int)e; }
  const BnActXf xf{mean, rstd, gamma, beta, slope};
  hipLaunchKernelGGL(bnact_absmax_kernel, dim3(ceil_div(S, kBnSlice), B, C), dim3(kBnThreads), 0, s, x, xf, C, S,
                     static_cast<uint32_t *>(out));
  return check_launch("bnact_absmax");
}
