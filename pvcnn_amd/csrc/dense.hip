// dense.hip -- Linear + BatchNorm1d + ReLU on a handful of rows.
//
// The cloud-descriptor heads of the reference are stacks of `nn.Sequential(nn.Linear, nn.BatchNorm1d, nn.ReLU)` (models/utils.py:11-12,
// `_linear_bn_relu`) applied to ONE row per cloud: (B, 1024) -> 256 -> 128 in models/s3dis/pvcnn.py:22-25, (B, 259) -> 256 -> 128 and
// (B, 515) -> 512 -> 256 in the Frustum nets (models/kitti/frustum/center_regression_net.py, box_estimation/pointnet.py).  As torch
// modules one such layer is 5 launches forward (addmm, two BatchNorm kernels, the counter, the ReLU) and 6 backward, each ~5 us of
// launch latency for microseconds of work on B = 16 ... 32 rows: ~22 launches per PVCNN step, ~50 per Frustum-PVCNN step.
//
// With so few rows a workgroup that owns an output channel owns the WHOLE batch of that channel, so the BatchNorm statistics are
// local to it: the layer is one launch forward (dot products, batch statistics, running statistics, counter, normalise, ReLU) and
// one launch backward for everything but grad_x (ReLU mask, the BatchNorm backward's two sums, grad_z, the weight / bias / gamma /
// beta gradients); grad_x = grad_z . W stays a library GEMM (the caller's: one launch).
//   forward   grid = ceil(Cout / 4), 256 threads: wave w owns channel 4 * blockIdx.x + w; x (rows x Cin) staged in LDS once per
//             workgroup, the weight row streamed coalesced, rows accumulators per lane, one wave reduction per row.
//   backward  same shape; the channel's grad_z row is broadcast to the lanes, which then own input channels: grad_W[co][ci] =
//             sum_r grad_z[r] * x[r][ci] with x from LDS, stored coalesced.
// fp32 throughout, every sum in a fixed order (deterministic); parity: <= 1e-6 of torch's modules (tests/test_gpu_dense.py).
#include <algorithm>

#include "common.h"

namespace pvcnn {

constexpr int kDenseThreads = 256;
constexpr int kDenseRowsMax = 64;
constexpr size_t kDenseLdsMax = 144 * 1024;

__device__ __forceinline__ float dense_wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// x (n floats) -> LDS with 16-byte loads, four in flight per thread (the staging IS most of these launches: 64 KiB per workgroup)
__device__ __forceinline__ void dense_stage(float *__restrict__ lds, const float *__restrict__ x, int n) {
  const int tid = threadIdx.x;
  if ((n & 3) == 0 && aligned16(x)) {
    const int nq = n >> 2;
    const float4 *xq = reinterpret_cast<const float4 *>(x);
    float4 *lq = reinterpret_cast<float4 *>(lds);
    int q = tid;
    for (; q + 3 * kDenseThreads < nq; q += 4 * kDenseThreads) {
      const float4 a = xq[q], b = xq[q + kDenseThreads], c = xq[q + 2 * kDenseThreads], d = xq[q + 3 * kDenseThreads];
      lq[q] = a; lq[q + kDenseThreads] = b; lq[q + 2 * kDenseThreads] = c; lq[q + 3 * kDenseThreads] = d;
    }
    for (; q < nq; q += kDenseThreads) lq[q] = xq[q];
  } else {
    for (int i = tid; i < n; i += kDenseThreads) lds[i] = x[i];
  }
}

// rows <= RT (RT = 16 | 32 | 64: the accumulators are registers).  z: (rows, Cout) = x W^T + bias, kept for backward.
template <int RT>
__global__ __launch_bounds__(kDenseThreads) void dense_bn_relu_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                         const float *__restrict__ bias, const float *__restrict__ gamma,
                                                                         const float *__restrict__ beta, float *__restrict__ running_mean,
                                                                         float *__restrict__ running_var, long long *__restrict__ counter,
                                                                         int rows, int Cin, int Cout, float eps, float momentum,
                                                                         float *__restrict__ z, float *__restrict__ y,
                                                                         float *__restrict__ mean, float *__restrict__ rstd) {
  extern __shared__ __attribute__((aligned(16))) float dense_lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  dense_stage(dense_lds, x, rows * Cin);
  if (counter != nullptr && blockIdx.x == 0 && tid == 0) *counter += 1;      // BatchNorm's num_batches_tracked
  __syncthreads();
  const int co = blockIdx.x * 4 + wave;
  if (co >= Cout) return;
  float acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = 0.0f;
  const float *wr = w + (size_t)co * Cin;
  // (eight weight loads in flight per lane: a loop of load -> 16 FMAs -> load is a chain of Cin / 64 memory round trips)
  for (int ci0 = lane; ci0 < Cin; ci0 += 8 * 64) {
    float wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) wv[u] = (ci0 + 64 * u < Cin) ? wr[ci0 + 64 * u] : 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int ci = ci0 + 64 * u;
      if (ci < Cin) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
          if (r < rows) acc[r] = fmaf(wv[u], dense_lds[r * Cin + ci], acc[r]);
      }
    }
  }
  // every lane ends up with the totals of all rows; lane r keeps row r
  float mine = 0.0f;
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const float t = dense_wave_sum(acc[r]);
    if (lane == r) mine = t;
  }
  const bool live = lane < rows;
  const float zv = live ? mine + (bias ? bias[co] : 0.0f) : 0.0f;
  const float inv = 1.0f / (float)rows;
  const float m = dense_wave_sum(zv) * inv;
  const float d = live ? zv - m : 0.0f;
  const float var = dense_wave_sum(d * d) * inv;            // biased, two-pass
  const float rs = 1.0f / sqrtf(var + eps);
  if (live) {
    const float pre = (gamma ? gamma[co] : 1.0f) * (d * rs) + (beta ? beta[co] : 0.0f);
    z[(size_t)lane * Cout + co] = zv;
    y[(size_t)lane * Cout + co] = pre > 0.0f ? pre : 0.0f;
  }
  if (lane == 0) {
    mean[co] = m;
    rstd[co] = rs;
    if (running_mean != nullptr) {
      const float unbiased = rows > 1 ? var * (float)rows / (float)(rows - 1) : var;
      running_mean[co] = (1.0f - momentum) * running_mean[co] + momentum * m;
      running_var[co] = (1.0f - momentum) * running_var[co] + momentum * unbiased;
    }
  }
}

// grad_z (rows, Cout): the gradient of z = x W^T + bias; grad_W (Cout, Cin), grad_bias / grad_gamma / grad_beta (Cout).
template <int RT>
__global__ __launch_bounds__(kDenseThreads) void dense_bn_relu_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                         const float *__restrict__ z, const float *__restrict__ mean,
                                                                         const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                                         const float *__restrict__ beta, int rows, int Cin, int Cout,
                                                                         float *__restrict__ gz, float *__restrict__ gw,
                                                                         float *__restrict__ gbias, float *__restrict__ ggamma,
                                                                         float *__restrict__ gbeta) {
  extern __shared__ __attribute__((aligned(16))) float dense_lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  dense_stage(dense_lds, x, rows * Cin);
  __syncthreads();
  const int co = blockIdx.x * 4 + wave;
  if (co >= Cout) return;
  const bool live = lane < rows;
  const float m = mean[co], rs = rstd[co], gm = gamma ? gamma[co] : 1.0f, bt = beta ? beta[co] : 0.0f;
  const float xhat = live ? (z[(size_t)lane * Cout + co] - m) * rs : 0.0f;
  const float g = (live && gm * xhat + bt > 0.0f) ? gy[(size_t)lane * Cout + co] : 0.0f;     // ReLU'(pre) * grad_y
  const float dbeta = dense_wave_sum(g), dgamma = dense_wave_sum(g * xhat);
  const float inv = 1.0f / (float)rows;
  const float gzv = live ? gm * rs * (g - dbeta * inv - xhat * (dgamma * inv)) : 0.0f;
  const float gb = dense_wave_sum(gzv);
  if (live) gz[(size_t)lane * Cout + co] = gzv;
  if (lane == 0) {
    if (gbias) gbias[co] = gb;
    if (ggamma) ggamma[co] = dgamma;
    if (gbeta) gbeta[co] = dbeta;
  }
  float row[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) row[r] = __shfl(gzv, r);
  float *out = gw + (size_t)co * Cin;
  for (int ci = lane; ci < Cin; ci += 64) {
    float a = 0.0f;
#pragma unroll
    for (int r = 0; r < RT; ++r)
      if (r < rows) a = fmaf(row[r], dense_lds[r * Cin + ci], a);
    out[ci] = a;
  }
}

template <class K>
static int dense_big_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace pvcnn

using namespace pvcnn;

// 1 when the pair of launches below serves (rows, Cin, Cout): 2 <= rows <= 64 (train-mode BatchNorm1d needs two rows) and x fits the LDS
// (kDenseLdsMax is gfx950's 160 KiB minus headroom: this library is built for gfx950 only -- common.h: kLdsBytesPerCU -- and the
//  Makefile's ARCH is not a supported knob for this file)
extern "C" int pvcnn_dense_bn_relu_supported(int rows, int Cin, int Cout) {
  return rows >= 2 && rows <= kDenseRowsMax && Cin > 0 && Cout > 0 && (size_t)rows * Cin * sizeof(float) <= kDenseLdsMax;
}

extern "C" int pvcnn_dense_bn_relu_fwd(const float *x, const float *weight, const float *bias, const float *gamma, const float *beta,
                                       float *running_mean, float *running_var, void *num_batches_tracked, int rows, int Cin, int Cout,
                                       float eps, float momentum, float *z, float *y, float *mean, float *rstd, void *stream) {
  PVCNN_REQUIRE(pvcnn_dense_bn_relu_supported(rows, Cin, Cout), "unsupported size: ask pvcnn_dense_bn_relu_supported");
  PVCNN_REQUIRE(x && weight && z && y && mean && rstd, "null pointer");
  PVCNN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running_mean and running_var come as a pair");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t lds = (size_t)rows * Cin * sizeof(float);
  const dim3 grid(ceil_div(Cout, 4));
  long long *ctr = static_cast<long long *>(num_batches_tracked);
#define PVCNN_DENSE_FWD(RT)                                                                                                             \
  do {                                                                                                                                  \
    auto k = dense_bn_relu_fwd_kernel<RT>;                                                                                              \
    if (int e = dense_big_lds(k, lds)) { set_error("dense_bn_relu_fwd: LDS attribute: %d", e); return e; }                              \
    hipLaunchKernelGGL(k, grid, dim3(kDenseThreads), lds, s, x, weight, bias, gamma, beta, running_mean, running_var, ctr, rows, Cin,   \
                       Cout, eps, momentum, z, y, mean, rstd);                                                                          \
  } while (0)
  if (rows <= 16) PVCNN_DENSE_FWD(16); else if (rows <= 32) PVCNN_DENSE_FWD(32); else PVCNN_DENSE_FWD(64);
#undef PVCNN_DENSE_FWD
  return check_launch("dense_bn_relu_fwd");
}

extern "C" int pvcnn_dense_bn_relu_bwd(const float *x, const float *grad_y, const float *z, const float *mean, const float *rstd,
                                       const float *gamma, const float *beta, int rows, int Cin, int Cout, float *grad_z,
                                       float *grad_weight, float *grad_bias, float *grad_gamma, float *grad_beta, void *stream) {
  PVCNN_REQUIRE(pvcnn_dense_bn_relu_supported(rows, Cin, Cout), "unsupported size: ask pvcnn_dense_bn_relu_supported");
  PVCNN_REQUIRE(x && grad_y && z && mean && rstd && grad_z && grad_weight, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t lds = (size_t)rows * Cin * sizeof(float);
  const dim3 grid(ceil_div(Cout, 4));
#define PVCNN_DENSE_BWD(RT)                                                                                                             \
  do {                                                                                                                                  \
    auto k = dense_bn_relu_bwd_kernel<RT>;                                                                                              \
    if (int e = dense_big_lds(k, lds)) { set_error("dense_bn_relu_bwd: LDS attribute: %d", e); return e; }                              \
    hipLaunchKernelGGL(k, grid, dim3(kDenseThreads), lds, s, x, grad_y, z, mean, rstd, gamma, beta, rows, Cin, Cout, grad_z, grad_weight, \
                       grad_bias, grad_gamma, grad_beta);                                                                               \
  } while (0)
  if (rows <= 16) PVCNN_DENSE_BWD(16); else if (rows <= 32) PVCNN_DENSE_BWD(32); else PVCNN_DENSE_BWD(64);
#undef PVCNN_DENSE_BWD
  return check_launch("dense_bn_relu_bwd");
}
