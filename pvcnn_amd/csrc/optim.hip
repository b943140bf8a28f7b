// optim.hip -- the optimizer update of the data-parallel step on FLAT buffers.
//
// Reference: train.py:96-119 steps torch.optim.Adam over ~100 separate parameter tensors; torch's fused multi-tensor Adam needs
// 2 launches of 72 us for PVCNN's 9.8 MiB of parameters (70 MB of traffic that streams in ~15 us): the per-tensor chunking, not the
// arithmetic, is the cost.  pvcnn_amd/dp.py already keeps the gradients in a few flat buckets (what RCCL all-reduces); with the
// parameters laid out the same way (GradBucketReducer.flatten_parameters) the update is ONE elementwise pass per bucket.
// Arithmetic = torch.optim.Adam (no amsgrad, L2 weight decay folded into the gradient), fp32, step counter on the device
// (graph-capturable: nothing is read back by the host):
//     g   = grad + wd * p
//     m   = m + (g - m) * (1 - b1)                  (torch: exp_avg.lerp_)
//     v   = b2 * v + (1 - b2) * g * g
//     p  -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps),        t = *step + 1
// The hyper-parameters (lr, beta1, beta2, eps, weight_decay) are read from DEVICE memory too: a learning-rate schedule changes five
// floats there, and a hipGraph that captured this launch follows it (a scalar kernel argument would be frozen at capture time).
#include <algorithm>

#include "common.h"

namespace pvcnn {

__global__ __launch_bounds__(256) void adam_flat_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                        float *__restrict__ v, size_t n, const float *__restrict__ step,
                                                        const float *__restrict__ hyper) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
  const float t = *step + 1.0f;
  const float bc1 = 1.0f - powf(b1, t), bc2_sqrt = sqrtf(1.0f - powf(b2, t));
  const float step_size = lr / bc1;
  const size_t n4 = n >> 2, stride = (size_t)gridDim.x * 256;
  auto upd = [&](float &pp, float gg, float &mm, float &vv) {
    gg = gg + wd * pp;
    mm = mm + (gg - mm) * (1.0f - b1);
    vv = b2 * vv + (1.0f - b2) * gg * gg;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp = pp - step_size * (mm / denom);
  };
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pv = reinterpret_cast<float4 *>(p)[i], mv = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
    const float4 gv = reinterpret_cast<const float4 *>(g)[i];
    upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
    reinterpret_cast<float4 *>(p)[i] = pv; reinterpret_cast<float4 *>(m)[i] = mv; reinterpret_cast<float4 *>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = (n4 << 2) + threadIdx.x;
    upd(p[i], g[i], m[i], v[i]);
  }
}

__global__ void adam_step_inc_kernel(float *step) { *step += 1.0f; }

}  // namespace pvcnn

using namespace pvcnn;

// One Adam update of n contiguous fp32 parameters (p, m = exp_avg, v = exp_avg_sq updated in place; g read).  `step` = one float in
// device memory, the number of updates done so far; inc_step != 0 increments it behind the update (pass it with the LAST buffer of an
// optimizer step when the parameters live in several buffers).  hyper: five floats in device memory (lr, beta1, beta2, eps, weight_decay).
extern "C" int pvcnn_adam_step(float *p, const float *g, float *m, float *v, size_t n, float *step, const float *hyper, int inc_step,
                               void *stream) {
  PVCNN_REQUIRE(step && hyper, "null step counter / hyper-parameter block");
  PVCNN_REQUIRE(n == 0 || (p && g && m && v), "null pointer");
  PVCNN_REQUIRE(n == 0 || (aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v)), "buffers must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n > 0) {
    const unsigned grid = (unsigned)std::min<size_t>(2048, (n / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(adam_flat_kernel, dim3(grid), dim3(256), 0, s, p, g, m, v, n, step, hyper);
    if (int rc = check_launch("adam_flat")) return rc;
  }
  if (inc_step) {
    hipLaunchKernelGGL(adam_step_inc_kernel, dim3(1), dim3(1), 0, s, step);
    return check_launch("adam_step_inc");
  }
  return 0;
}
