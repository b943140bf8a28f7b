// frustum.hip -- the box part of Frustum-PointNet's multi-task loss (reference: modules/frustum.py:43-124, FrustumPointNetLoss) as
// ONE launch that returns the loss AND its gradient with respect to the eight network outputs it reads.
//
//   box = huber(|c_t - c|, 2) + huber(|c_t - c_reg|, 1) + CE(heading_scores, h) + CE(size_scores, s)
//       + w_h * huber(hrn[h] - h_res_t / (pi / NH), 1) + w_s * huber(|s_res_t / T[s] - srn[s]|, 1)
//       + w_c * huber(min(|corners - corners_t|, |corners - corners_t turned by pi|), 1)          (every term a mean over the batch)
//
// The tensors are (B, <= 8 x 3): as torch ops the term is ~120 launches forward and ~140 backward of 2.5-4 us each -- a third of the
// launches of a Frustum-PVCNN step and, with the dependency gaps between such short kernels inside the step graph, ~0.8 of its
// 6.9 ms.  Here a thread owns a sample: it evaluates the seven terms and their analytic derivatives (the sub-gradient conventions
// are autograd's: d|x|/dx = 0 at 0, d||v||/dv = 0 at v = 0, clamp passes the gradient at the bound, a tie of the two corner
// distances splits it evenly), writes its rows of the eight gradients (every element: no memset) and its share of the seven sums;
// the sums meet in LDS in thread order (deterministic).  The foreground mask's cross entropy (B x N points) stays a torch op.
#include "common.h"

namespace pvcnn {

struct FrustumLossArgs {
  // network outputs
  const float *center, *center_reg, *heading_scores, *size_scores, *hrn, *srn, *hr, *sr;
  // targets
  const long long *heading_bin_id, *size_template_id;
  const float *heading_residual, *size_residual, *center_t;
  // constants of the loss
  const float *templates, *bin_centers;
  float w_heading, w_size, w_corners, inv_bin_width;      // inv_bin_width: h_res_t is DIVIDED by (pi / NH) in the reference -> pass that divisor
  int B, NH, NS;
  // outputs: loss[0] = box; grads = [center 3B | center_reg 3B | heading_scores B NH | size_scores B NS | hrn B NH | srn B NS 3 | hr B NH | sr B NS 3]
  float *loss, *grads;
};

__device__ __forceinline__ float huber_val(float mag, float delta) {
  const float inner = fminf(mag, delta);
  return 0.5f * inner * inner + delta * (mag - inner);
}
// d huber / d mag for mag >= 0 (clamp(max = delta) passes the gradient where mag <= delta)
__device__ __forceinline__ float huber_dmag(float mag, float delta) { return mag <= delta ? mag : delta; }

// cross entropy of one row of `n` scores against class k; writes (softmax - onehot) * scale to g
__device__ __forceinline__ float ce_row(const float *__restrict__ x, int n, int k, float scale, float *__restrict__ g) {
  float m = x[0];
  for (int j = 1; j < n; ++j) m = fmaxf(m, x[j]);
  float s = 0.0f;
  for (int j = 0; j < n; ++j) s += expf(x[j] - m);
  const float lse = m + logf(s);
  for (int j = 0; j < n; ++j) g[j] = (expf(x[j] - lse) - (j == k ? 1.0f : 0.0f)) * scale;
  return lse - x[k];
}

__global__ __launch_bounds__(256) void frustum_box_loss_kernel(FrustumLossArgs a) {
  constexpr int SX[8] = {1, 1, -1, -1, 1, 1, -1, -1}, SY[8] = {1, 1, 1, 1, -1, -1, -1, -1}, SZ[8] = {1, -1, -1, 1, 1, -1, -1, 1};
  __shared__ float red[7][256];
  const int B = a.B, NH = a.NH, NS = a.NS, tid = threadIdx.x;
  float *g_center = a.grads, *g_creg = g_center + 3 * (size_t)B, *g_hs = g_creg + 3 * (size_t)B, *g_ss = g_hs + (size_t)B * NH,
        *g_hrn = g_ss + (size_t)B * NS, *g_srn = g_hrn + (size_t)B * NH, *g_hr = g_srn + (size_t)B * NS * 3, *g_sr = g_hr + (size_t)B * NH;
  const float invB = 1.0f / (float)B;
  float sum[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // center, center_reg, CE heading, CE size, heading residual, size residual, corners
  for (int b = tid; b < B; b += 256) {
    const int h = min(max((int)a.heading_bin_id[b], 0), NH - 1), s = min(max((int)a.size_template_id[b], 0), NS - 1);
    const float *T = a.templates + 3 * s;
    // ---- classification
    sum[2] += ce_row(a.heading_scores + (size_t)b * NH, NH, h, invB, g_hs + (size_t)b * NH);
    sum[3] += ce_row(a.size_scores + (size_t)b * NS, NS, s, invB, g_ss + (size_t)b * NS);
    // ---- the two centre terms: huber(|c_t - c|, delta)
    const float ct[3] = {a.center_t[3 * b], a.center_t[3 * b + 1], a.center_t[3 * b + 2]};
    float gc[3];
    {
      const float c[3] = {a.center[3 * b], a.center[3 * b + 1], a.center[3 * b + 2]};
      const float d[3] = {ct[0] - c[0], ct[1] - c[1], ct[2] - c[2]};
      const float n = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      sum[0] += huber_val(n, 2.0f);
      const float k = n > 0.0f ? -huber_dmag(n, 2.0f) / n * invB : 0.0f;
      gc[0] = k * d[0]; gc[1] = k * d[1]; gc[2] = k * d[2];
    }
    {
      const float c[3] = {a.center_reg[3 * b], a.center_reg[3 * b + 1], a.center_reg[3 * b + 2]};
      const float d[3] = {ct[0] - c[0], ct[1] - c[1], ct[2] - c[2]};
      const float n = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      sum[1] += huber_val(n, 1.0f);
      const float k = n > 0.0f ? -huber_dmag(n, 1.0f) / n * invB : 0.0f;
      g_creg[3 * b] = k * d[0]; g_creg[3 * b + 1] = k * d[1]; g_creg[3 * b + 2] = k * d[2];
    }
    // ---- normalised residuals of the target bin / template
    const float hres_t = a.heading_residual[b];
    {
      const float e = a.hrn[(size_t)b * NH + h] - hres_t / a.inv_bin_width;
      const float mag = fabsf(e);
      sum[4] += huber_val(mag, 1.0f);
      const float ge = huber_dmag(mag, 1.0f) * (e > 0.0f ? 1.0f : e < 0.0f ? -1.0f : 0.0f) * a.w_heading * invB;
      for (int j = 0; j < NH; ++j) g_hrn[(size_t)b * NH + j] = j == h ? ge : 0.0f;
    }
    const float srt[3] = {a.size_residual[3 * b], a.size_residual[3 * b + 1], a.size_residual[3 * b + 2]};
    {
      const float *p = a.srn + ((size_t)b * NS + s) * 3;
      const float v[3] = {srt[0] / T[0] - p[0], srt[1] / T[1] - p[1], srt[2] / T[2] - p[2]};
      const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      sum[5] += huber_val(n, 1.0f);
      const float k = n > 0.0f ? -huber_dmag(n, 1.0f) / n * a.w_size * invB : 0.0f;
      for (int j = 0; j < NS; ++j)
        for (int i = 0; i < 3; ++i) g_srn[((size_t)b * NS + j) * 3 + i] = j == s ? k * v[i] : 0.0f;
    }
    // ---- corner loss against the target box and the same box turned by pi
    {
      const float bin = a.bin_centers[h];
      const float heading = a.hr[(size_t)b * NH + h] + bin;
      const float *q = a.sr + ((size_t)b * NS + s) * 3;
      const float hl = (q[0] + T[0]) / 2, hw = (q[1] + T[1]) / 2, hh = (q[2] + T[2]) / 2;        // half length, width, height
      const float cs = cosf(heading), sn = sinf(heading);
      const float heading_t = bin + hres_t;
      const float tl = (T[0] + srt[0]) / 2, tw = (T[1] + srt[1]) / 2, th = (T[2] + srt[2]) / 2;
      const float ct_ = cosf(heading_t), st_ = sinf(heading_t);
      const float c0 = a.center[3 * b], c1 = a.center[3 * b + 1], c2 = a.center[3 * b + 2];
      const float gscale = a.w_corners * invB * 0.125f;
      float g_heading = 0.0f, g_hl = 0.0f, g_hw = 0.0f, g_hh = 0.0f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float lx = hl * SX[k], ly = hh * SY[k], lz = hw * SZ[k];
        const float X = cs * lx + sn * lz + c0, Y = ly + c1, Z = -sn * lx + cs * lz + c2;
        const float tx = tl * SX[k], ty = th * SY[k], tz = tw * SZ[k];
        const float u1[3] = {X - (ct_ * tx + st_ * tz + ct[0]), Y - (ty + ct[1]), Z - (-st_ * tx + ct_ * tz + ct[2])};
        const float u2[3] = {X - (-ct_ * tx - st_ * tz + ct[0]), Y - (ty + ct[1]), Z - (st_ * tx - ct_ * tz + ct[2])};
        const float d1 = sqrtf(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]), d2 = sqrtf(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
        const float m = fminf(d1, d2);
        sum[6] += huber_val(m, 1.0f);
        const float gm = huber_dmag(m, 1.0f) * gscale;
        const float w1 = d1 < d2 ? 1.0f : d1 > d2 ? 0.0f : 0.5f, w2 = 1.0f - w1;
        const float k1 = d1 > 0.0f ? gm * w1 / d1 : 0.0f, k2 = d2 > 0.0f ? gm * w2 / d2 : 0.0f;
        const float gX = k1 * u1[0] + k2 * u2[0], gY = k1 * u1[1] + k2 * u2[1], gZ = k1 * u1[2] + k2 * u2[2];
        gc[0] += gX; gc[1] += gY; gc[2] += gZ;
        g_heading += gX * (-sn * lx + cs * lz) + gZ * (-cs * lx - sn * lz);
        g_hl += (gX * cs - gZ * sn) * SX[k];
        g_hh += gY * SY[k];
        g_hw += (gX * sn + gZ * cs) * SZ[k];
      }
      for (int j = 0; j < NH; ++j) g_hr[(size_t)b * NH + j] = j == h ? g_heading : 0.0f;
      for (int j = 0; j < NS; ++j) {
        float *o = g_sr + ((size_t)b * NS + j) * 3;
        o[0] = j == s ? 0.5f * g_hl : 0.0f; o[1] = j == s ? 0.5f * g_hw : 0.0f; o[2] = j == s ? 0.5f * g_hh : 0.0f;
      }
    }
    g_center[3 * b] = gc[0]; g_center[3 * b + 1] = gc[1]; g_center[3 * b + 2] = gc[2];
  }
#pragma unroll
  for (int t = 0; t < 7; ++t) red[t][tid] = sum[t];
  __syncthreads();
  if (tid == 0) {
    float tot[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      float s = 0.0f;
      for (int i = 0; i < 256; ++i) s += red[t][i];
      tot[t] = s;
    }
    a.loss[0] = tot[0] * invB + tot[1] * invB + (tot[2] * invB + tot[3] * invB) + a.w_heading * (tot[4] * invB) + a.w_size * (tot[5] * invB) +
                a.w_corners * (tot[6] * invB * 0.125f);
  }
}

}  // namespace pvcnn

using namespace pvcnn;

extern "C" size_t pvcnn_frustum_box_loss_grad_floats(int B, int NH, int NS) {
  if (B <= 0 || NH <= 0 || NS <= 0) return 0;
  return (size_t)B * (6 + 3 * (size_t)NH + 7 * (size_t)NS);
}

extern "C" int pvcnn_frustum_box_loss(const float *center, const float *center_reg, const float *heading_scores, const float *size_scores,
                                      const float *heading_residuals_normalized, const float *size_residuals_normalized,
                                      const float *heading_residuals, const float *size_residuals, const long long *heading_bin_id,
                                      const long long *size_template_id, const float *heading_residual_t, const float *size_residual_t,
                                      const float *center_t, const float *size_templates, const float *heading_bin_centers, int B, int NH,
                                      int NS, float heading_bin_width, float w_heading_residual, float w_size_residual, float w_corners,
                                      float *loss, float *grads, void *stream) {
  PVCNN_REQUIRE(B > 0 && NH > 0 && NS > 0, "bad size");
  PVCNN_REQUIRE(center && center_reg && heading_scores && size_scores && heading_residuals_normalized && size_residuals_normalized &&
                    heading_residuals && size_residuals && heading_bin_id && size_template_id && heading_residual_t && size_residual_t &&
                    center_t && size_templates && heading_bin_centers && loss && grads,
                "null pointer");
  PVCNN_REQUIRE(heading_bin_width > 0.0f, "heading_bin_width must be positive (pi / NH in the reference)");
  FrustumLossArgs a{center, center_reg, heading_scores, size_scores, heading_residuals_normalized, size_residuals_normalized,
                    heading_residuals, size_residuals, heading_bin_id, size_template_id, heading_residual_t, size_residual_t, center_t,
                    size_templates, heading_bin_centers, w_heading_residual, w_size_residual, w_corners, heading_bin_width, B, NH, NS,
                    loss, grads};
  hipLaunchKernelGGL(frustum_box_loss_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch("frustum_box_loss");
}
