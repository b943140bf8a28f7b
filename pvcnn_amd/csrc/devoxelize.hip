// devoxelize.hip -- trilinear_devoxelize forward / backward for gfx950.
//
// Reference: interpolate/trilinear_devox.cu:21-105 (forward: 8 uncoalesced global gathers per
// point per channel from B workgroups) and :119-162 (backward: 8*C global float atomics per
// point onto a memset grid).  Both are instances of the LDS slab kernels in slab.h:
//   forward : gather_lds_kernel<TrilinearFromCoords>  -- a workgroup streams G channel grids
//             (one 128 KiB grid at R = 32) into LDS, recomputes the 8 corner indices/weights
//             of each point from its 12-byte coordinate (cheaper than re-reading 64 bytes of
//             saved inds/wgts), and serves the 8*C random reads per point from LDS.  The
//             workgroups of channel-slab 0 also emit inds/wgts (B,8,N) in training mode.
//             At R = 32 (one grid = the whole LDS of a CU) the software-pipelined
//             gather_lds_pipe_kernel keeps the next grid's loads in flight during the gather.
//   backward: the deterministic CSR scatter of csr.h with 8 entries per point (entry id =
//             point*8 + corner, the reference's loop order): one counting sort per cloud, then
//             lane-owned sums over grad_y rows staged in LDS; every grid element written once;
//             no memset, no float atomics, bit-identical to the serial oracle.
#include "csr.h"

using namespace pvcnn;

// C == 0 in training mode: no grid to read, but inds/wgts are still produced (edge case only)
__global__ __launch_bounds__(256) void trilinear_taps_only_kernel(TrilinearFromCoords p, int N) {
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  Taps<8> t;
  p.load1(b, j, t);
  p.post1(b, j, t);
}

extern "C" int pvcnn_trilinear_devox_fwd(const float *coords, const float *feat, int B, int C, int N, int R,
                                         int is_training, int32_t *inds, float *wgts, float *outs,
                                         void *stream) {
  PVCNN_REQUIRE(B >= 0 && C >= 0 && N >= 0 && R > 0, "negative size");
  PVCNN_REQUIRE((long)R * R * R <= 0x7fffffffL / 4, "resolution too large");
  if (B == 0 || N == 0) return 0;
  PVCNN_REQUIRE(coords && (outs || C == 0) && (feat || C == 0), "null pointer");
  PVCNN_REQUIRE(!is_training || (inds && wgts), "training mode needs inds and wgts");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TrilinearFromCoords p{coords, is_training ? inds : nullptr, is_training ? wgts : nullptr, N, R, R * R};
  bool vec = (N % 4 == 0) && aligned16(coords) && aligned16(outs);
  if (is_training) vec = vec && aligned16(inds) && aligned16(wgts);
  if (C == 0) {
    // nothing to interpolate, but the training side outputs are still part of the contract
    if (!is_training) return 0;
    hipLaunchKernelGGL(trilinear_taps_only_kernel, dim3(ceil_div(N, 256), B), dim3(256), 0, s, p, N);
    return check_launch("trilinear_taps_only");
  }
  return launch_gather(p, feat, outs, B, C, /*L=*/R * R * R, /*J=*/N, vec, s, "trilinear_devox_fwd", XfNone{}, grid_pad_shift(R));
}

extern "C" int pvcnn_trilinear_devox_bnact_fwd(const float *coords, const float *feat, const float *gamma,
                                               const float *beta, const float *mean, const float *rstd, float slope, int B,
                                               int C, int N, int R, int is_training, int32_t *inds, float *wgts,
                                               const float *addend, const float *se_scale, float *outs, void *stream) {
  PVCNN_REQUIRE(B >= 0 && C > 0 && N >= 0 && R > 0, "bad size");
  PVCNN_REQUIRE((long)R * R * R * sizeof(float) <= (size_t)kLdsBytesPerCU, "grid row does not fit LDS: use bnact_fwd + trilinear_devox_fwd");
  if (B == 0 || N == 0) return 0;
  PVCNN_REQUIRE(coords && outs && feat && mean && rstd, "null pointer");
  PVCNN_REQUIRE(!is_training || (inds && wgts), "training mode needs inds and wgts");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TrilinearFromCoords p{coords, is_training ? inds : nullptr, is_training ? wgts : nullptr, N, R, R * R};
  bool vec = (N % 4 == 0) && aligned16(coords) && aligned16(outs) && (!addend || aligned16(addend));
  if (is_training) vec = vec && aligned16(inds) && aligned16(wgts);
  const XfBnAct xf{gamma, beta, mean, rstd, slope, se_scale};
  return launch_gather(p, feat, outs, B, C, /*L=*/R * R * R, /*J=*/N, vec, s, "trilinear_devox_bnact_fwd", xf, grid_pad_shift(R), addend);
}

extern "C" size_t pvcnn_trilinear_devox_bwd_workspace_bytes(int B, int C, int N, int R) {
  if (B <= 0 || C < 0 || N < 0 || R <= 0) return 0;
  const long S = (long)R * R * R;
  if (S > 0x7fffffffL || !csr_supported((int)S, 8L * N)) return 16;   // atomic fallback: no scratch
  return CsrWorkspace::bytes(B, C, (int)S, N, 8L * N);
}

static int devox_bwd_impl(const float *grad_y, long gy_bstride, const int32_t *inds, const float *wgts, int B, int C, int N,
                          int R, float *grad_x, void *workspace, size_t workspace_bytes, void *stream) {
  PVCNN_REQUIRE(B >= 0 && C >= 0 && N >= 0 && R > 0, "negative size");
  PVCNN_REQUIRE((long)R * R * R <= 0x7fffffffL / 4, "resolution too large");
  if (B == 0 || C == 0) return 0;
  PVCNN_REQUIRE(grad_x && (N == 0 || (grad_y && inds && wgts)), "null pointer");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  PVCNN_REQUIRE(gy_bstride >= (long)C * N, "grad_y batch stride smaller than one cloud");
  const int S = R * R * R;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!csr_supported(S, 8L * N)) {
    PVCNN_REQUIRE(gy_bstride == (long)C * N, "the atomic fallback needs a contiguous grad_y");
    SavedTaps<8> p{inds, wgts, N};
    return launch_scatter_direct(p, grad_y, grad_x, B, C, S, N, s, "trilinear_devox_bwd(atomic)");
  }
  TapEntries<8> ep{inds, wgts, N, S};
  return launch_csr_scatter(ep, grad_y, grad_x, B, C, /*L=*/S, /*J=*/N, /*E=*/8L * N, nullptr, workspace,
                            workspace_bytes, s, "trilinear_devox_bwd", gy_bstride);
}

extern "C" int pvcnn_trilinear_devox_bwd(const float *grad_y, const int32_t *inds, const float *wgts, int B,
                                         int C, int N, int R, float *grad_x, void *workspace,
                                         size_t workspace_bytes, void *stream) {
  return devox_bwd_impl(grad_y, (long)C * N, inds, wgts, B, C, N, R, grad_x, workspace, workspace_bytes, stream);
}

extern "C" int pvcnn_trilinear_devox_bwd_strided(const float *grad_y, long grad_y_batch_stride, const int32_t *inds,
                                                 const float *wgts, int B, int C, int N, int R, float *grad_x,
                                                 void *workspace, size_t workspace_bytes, void *stream) {
  return devox_bwd_impl(grad_y, grad_y_batch_stride, inds, wgts, B, C, N, R, grad_x, workspace, workspace_bytes, stream);
}

// ---- plan / apply split: the corner entries (inds, wgts) depend on (coords, R) only -- one counting sort serves every
// layer that devoxelizes at these coordinates ----
extern "C" size_t pvcnn_trilinear_devox_bwd_plan_bytes(int B, int N, int R) {
  if (B <= 0 || N < 0 || R <= 0) return 0;
  const long S = (long)R * R * R;
  if (S > 0x7fffffffL / 4 || !csr_supported((int)S, 8L * N)) return 0;
  return CsrPlan::bytes(B, (int)S, 8L * N);
}

extern "C" size_t pvcnn_trilinear_devox_bwd_plan_scratch_bytes(int B, int N, int R) {
  if (B <= 0 || N < 0 || R <= 0) return 0;
  return csr_prep_scratch_bytes(B, 8L * N);
}

extern "C" int pvcnn_trilinear_devox_bwd_plan(const int32_t *inds, const float *wgts, int B, int N, int R, void *plan,
                                              size_t plan_bytes, void *scratch, size_t scratch_bytes, void *stream) {
  PVCNN_REQUIRE(B >= 0 && N >= 0 && R > 0, "negative size");
  PVCNN_REQUIRE((long)R * R * R <= 0x7fffffffL / 4, "resolution too large");
  if (B == 0) return 0;
  const int S = R * R * R;
  PVCNN_REQUIRE(csr_supported(S, 8L * N), "grid too large for a plan: use pvcnn_trilinear_devox_bwd");
  PVCNN_REQUIRE(N == 0 || (inds && wgts), "null pointer");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  TapEntries<8> ep{inds, wgts, N, S};
  return launch_csr_prep(ep, B, /*L=*/S, /*E=*/8L * N, nullptr, plan, plan_bytes, scratch, scratch_bytes,
                         static_cast<hipStream_t>(stream), "trilinear_devox_bwd_plan");
}

extern "C" int pvcnn_trilinear_devox_bwd_apply(const float *grad_y, long grad_y_batch_stride, const void *plan, size_t plan_bytes,
                                               int B, int C, int N, int R, float *grad_x, void *stream) {
  PVCNN_REQUIRE(B >= 0 && C >= 0 && N >= 0 && R > 0, "negative size");
  PVCNN_REQUIRE((long)R * R * R <= 0x7fffffffL / 4, "resolution too large");
  if (B == 0 || C == 0) return 0;
  PVCNN_REQUIRE(grad_x && (N == 0 || grad_y), "null pointer");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  PVCNN_REQUIRE(grad_y_batch_stride >= (long)C * N, "grad_y batch stride smaller than one cloud");
  const int S = R * R * R;
  return launch_csr_apply(grad_y, plan, plan_bytes, grad_x, B, C, /*L=*/S, /*J=*/N, /*E=*/8L * N, static_cast<hipStream_t>(stream),
                          "trilinear_devox_bwd_apply", grad_y_batch_stride);
}
