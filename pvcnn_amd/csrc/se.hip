// se.hip -- the excitation of SE3d (reference: modules/se.py:6-17) between the two reduction passes of PVConv's fused tail.
//
//   squeezed[b][c] = (gamma[c] * Ax[b][c] + beta[c] * A[b][c]) / S          (= mean over the grid of leaky_relu(bn(x)), from two sums)
//   hidden         = relu(squeezed W1^T)             W1 (H, C), H = C / reduction
//   excite         = sigmoid(hidden W2^T)            W2 (C, H)
//
// and its backward.  These are (B, C) x (C, H) products with B = 8..64, C <= 1024: as torch ops they were ~9 launches forward and
// ~28 backward per squeeze-and-excitation block, 4.8 us each -- 2.3 of PVCNN++'s 16 ms step, 0.8 of ShapeNet-PVCNN's 5 ms.  Here:
// one launch forward (a workgroup per cloud), two backward (per cloud: the chain through the two layers; per 64 channels: the sums
// over the clouds -- weight gradients and the two BatchNorm sums -- in cloud order: deterministic).  The inputs are the reduction
// pass's partial sums as it writes them, `part` (C, B, slices, 2) (pvcnn_bnact_partial_sums): the sum over the slices happens here,
// in slice order (as torch ops: a sum + two transposing copies per pass, 78 more launches per PVCNN++ step).
#include "common.h"

namespace pvcnn {

constexpr int kSeMaxC = 2048, kSeMaxH = 256;

__device__ __forceinline__ float se_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// (first, second) sums of channel c, cloud b over the slices of the reduction pass
__device__ __forceinline__ float2 se_slice_sum(const float2 *__restrict__ part, int B, int slices, int b, int c) {
  const float2 *p = part + ((size_t)c * B + b) * slices;
  float2 t = p[0];
  for (int s = 1; s < slices; ++s) { t.x += p[s].x; t.y += p[s].y; }
  return t;
}

// grid = B, block = 256
__global__ __launch_bounds__(256) void se_excite_fwd_kernel(const float2 *__restrict__ part, int B, int slices,
                                                            float *__restrict__ a_sum, float *__restrict__ ax_sum,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ w1, const float *__restrict__ w2, int C, int H,
                                                            float inv_s, float *__restrict__ squeezed, float *__restrict__ hidden,
                                                            float *__restrict__ excite) {
  __shared__ float sq[kSeMaxC], hd[kSeMaxH];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int c = tid; c < C; c += 256) {
    const float g = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
    const float2 t = se_slice_sum(part, B, slices, b, c);
    a_sum[(size_t)b * C + c] = t.x;
    ax_sum[(size_t)b * C + c] = t.y;
    const float v = (g * t.y + bt * t.x) * inv_s;
    sq[c] = v;
    squeezed[(size_t)b * C + c] = v;
  }
  __syncthreads();
  for (int h = wave; h < H; h += 4) {
    float s = 0.0f;
    for (int c = lane; c < C; c += 64) s = fmaf(sq[c], w1[(size_t)h * C + c], s);
    s = se_wave_sum(s);
    s = s > 0.0f ? s : 0.0f;
    if (lane == 0) { hd[h] = s; hidden[(size_t)b * H + h] = s; }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float s = 0.0f;
    for (int h = 0; h < H; ++h) s = fmaf(hd[h], w2[(size_t)c * H + h], s);
    excite[(size_t)b * C + c] = 1.0f / (1.0f + expf(-s));
  }
}

// backward, per cloud (grid = B): g_excite = gamma * Q + beta * P;  g_pre2 = g_excite * e (1 - e);  g_pre1 = (g_pre2 W2) * [hidden > 0];
// g_mean = g_pre1 W1 / S  (the gradient of the squeeze, spread over the S voxels).  g_pre2 / g_pre1 go to `ws` for the second launch.
__global__ __launch_bounds__(256) void se_excite_bwd_chain_kernel(const float2 *__restrict__ part, int B, int slices,
                                                                  float *__restrict__ p_sum, float *__restrict__ q_sum,
                                                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                  const float *__restrict__ hidden, const float *__restrict__ excite,
                                                                  const float *__restrict__ w1, const float *__restrict__ w2, int C, int H,
                                                                  float inv_s, float *__restrict__ g_pre2, float *__restrict__ g_pre1,
                                                                  float *__restrict__ g_mean) {
  __shared__ float gp2[kSeMaxC], gp1[kSeMaxH];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int c = tid; c < C; c += 256) {
    const float g = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f, e = excite[(size_t)b * C + c];
    const float2 t = se_slice_sum(part, B, slices, b, c);                 // P = sum g_y act', Q = sum g_y act' xhat
    p_sum[(size_t)b * C + c] = t.x;
    q_sum[(size_t)b * C + c] = t.y;
    const float v = (g * t.y + bt * t.x) * e * (1.0f - e);
    gp2[c] = v;
    g_pre2[(size_t)b * C + c] = v;
  }
  __syncthreads();
  for (int h = wave; h < H; h += 4) {
    float s = 0.0f;
    for (int c = lane; c < C; c += 64) s = fmaf(gp2[c], w2[(size_t)c * H + h], s);
    s = se_wave_sum(s);
    s = hidden[(size_t)b * H + h] > 0.0f ? s : 0.0f;
    if (lane == 0) { gp1[h] = s; g_pre1[(size_t)b * H + h] = s; }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float s = 0.0f;
    for (int h = 0; h < H; ++h) s = fmaf(gp1[h], w1[(size_t)h * C + c], s);
    g_mean[(size_t)b * C + c] = s * inv_s;
  }
}

// backward, sums over the clouds (grid = ceil(C / 64), block = 256 = 64 channels x 4 hidden-unit lanes):
//   g_w2[c][h] = sum_b g_pre2[b][c] hidden[b][h];  g_w1[h][c] = sum_b g_pre1[b][h] squeezed[b][c];
//   sum_beta[c] = sum_b (e P + g_mean A);  sum_gamma[c] = sum_b (e Q + g_mean Ax)      (the BatchNorm backward's two sums, bnact.hip)
__global__ __launch_bounds__(256) void se_excite_bwd_sums_kernel(const float *__restrict__ p_sum, const float *__restrict__ q_sum,
                                                                 const float *__restrict__ a_sum, const float *__restrict__ ax_sum,
                                                                 const float *__restrict__ squeezed, const float *__restrict__ hidden,
                                                                 const float *__restrict__ excite, const float *__restrict__ g_pre2,
                                                                 const float *__restrict__ g_pre1, const float *__restrict__ g_mean, int B,
                                                                 int C, int H, float *__restrict__ g_w1, float *__restrict__ g_w2,
                                                                 float *__restrict__ sum_beta, float *__restrict__ sum_gamma) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), hl = threadIdx.x >> 6;
  if (c >= C) return;
  for (int h = hl; h < H; h += 4) {
    float s2 = 0.0f, s1 = 0.0f;
    for (int b = 0; b < B; ++b) {
      s2 = fmaf(g_pre2[(size_t)b * C + c], hidden[(size_t)b * H + h], s2);
      s1 = fmaf(g_pre1[(size_t)b * H + h], squeezed[(size_t)b * C + c], s1);
    }
    g_w2[(size_t)c * H + h] = s2;
    g_w1[(size_t)h * C + c] = s1;
  }
  if (hl == 0) {
    float sb = 0.0f, sg = 0.0f;
    for (int b = 0; b < B; ++b) {
      const size_t i = (size_t)b * C + c;
      const float e = excite[i], gm = g_mean[i];
      sb += e * p_sum[i] + gm * a_sum[i];
      sg += e * q_sum[i] + gm * ax_sum[i];
    }
    sum_beta[c] = sb;
    sum_gamma[c] = sg;
  }
}

}  // namespace pvcnn

using namespace pvcnn;

extern "C" int pvcnn_se_excite_fwd(const float *part, int slices, const float *gamma, const float *beta, const float *w1, const float *w2,
                                   int B, int C, int H, float inv_s, float *a_sum, float *ax_sum, float *squeezed, float *hidden,
                                   float *excite, void *stream) {
  PVCNN_REQUIRE(B > 0 && C > 0 && H > 0 && slices > 0 && C <= kSeMaxC && H <= kSeMaxH, "bad size (C <= 2048, H <= 256)");
  PVCNN_REQUIRE(part && a_sum && ax_sum && w1 && w2 && squeezed && hidden && excite, "null pointer");
  PVCNN_REQUIRE((reinterpret_cast<uintptr_t>(part) & 7) == 0, "part must be 8-byte aligned");
  hipLaunchKernelGGL(se_excite_fwd_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const float2 *>(part), B,
                     slices, a_sum, ax_sum, gamma, beta, w1, w2, C, H, inv_s, squeezed, hidden, excite);
  return check_launch("se_excite_fwd");
}

extern "C" int pvcnn_se_excite_bwd(const float *part, int slices, const float *a_sum, const float *ax_sum, const float *gamma,
                                   const float *beta, const float *squeezed, const float *hidden, const float *excite, const float *w1,
                                   const float *w2, int B, int C, int H, float inv_s, float *g_w1, float *g_w2, float *g_mean,
                                   float *sum_beta, float *sum_gamma, float *workspace /* B * (3 C + H) floats */, void *stream) {
  PVCNN_REQUIRE(B > 0 && C > 0 && H > 0 && slices > 0 && C <= kSeMaxC && H <= kSeMaxH, "bad size (C <= 2048, H <= 256)");
  PVCNN_REQUIRE(part && a_sum && ax_sum && squeezed && hidden && excite && w1 && w2 && g_w1 && g_w2 && g_mean && sum_beta && sum_gamma &&
                    workspace, "null pointer");
  PVCNN_REQUIRE((reinterpret_cast<uintptr_t>(part) & 7) == 0, "part must be 8-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float *g_pre2 = workspace, *p_sum = workspace + (size_t)B * C, *q_sum = workspace + (size_t)2 * B * C, *g_pre1 = workspace + (size_t)3 * B * C;
  hipLaunchKernelGGL(se_excite_bwd_chain_kernel, dim3(B), dim3(256), 0, s, reinterpret_cast<const float2 *>(part), B, slices, p_sum, q_sum, gamma,
                     beta, hidden, excite, w1, w2, C, H, inv_s, g_pre2, g_pre1, g_mean);
  if (int e = check_launch("se_excite_bwd_chain")) return e;
  hipLaunchKernelGGL(se_excite_bwd_sums_kernel, dim3(ceil_div(C, 64)), dim3(256), 0, s, p_sum, q_sum, a_sum, ax_sum, squeezed, hidden, excite,
                     g_pre2, g_pre1, g_mean, B, C, H, g_w1, g_w2, sum_beta, sum_gamma);
  return check_launch("se_excite_bwd_sums");
}
