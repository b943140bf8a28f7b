// voxelize.hip -- avg_voxelize forward / backward for gfx950.
//
// Reference: voxelization/vox.cu:18-72 (grid_stats + per-point float atomicAdd scatter from B
// workgroups, run-to-run non-deterministic) and vox.cu:86-110 (backward).
//
// Forward here is the deterministic CSR scatter of csr.h with one entry per point:
//   csr_prep_kernel<VoxelEntries>  (voxel range of a cloud split over workgroups, histogram in LDS)
//       ind[b,i], cnt[b,v], and the points grouped by voxel in ASCENDING point index;
//   segsum_kernel  (workgroup = G feature rows staged in LDS; lane = 4 consecutive voxels)
//       out[b,c,v] = sum over the voxel's points, in ascending point index, of
//                    feat[b,c,p] * (1/cnt) -- each addend pre-multiplied like vox.cu:66-68 --
//       and every voxel (occupied or not) is written exactly once with 16-byte stores.
// Result: no memset pass, no float atomics, bit-reproducible, and bit-identical to a serial
// point-order evaluation (what oracle/pvcnn_oracle.c computes).  HBM traffic = compulsory.
// Grids with R^3 > kCsrMaxTargets (2^20: R > 101) use the atomic fallback at the bottom.
#include <algorithm>

#include "csr.h"

namespace pvcnn {

// ---- fallback for R^3 > kCsrMaxTargets: global int atomics + float atomics (reference-like) ----
__global__ __launch_bounds__(256) void vox_stats_atomic_kernel(const int32_t *__restrict__ coords, int N,
                                                               int R, int S, int32_t *__restrict__ ind,
                                                               int32_t *__restrict__ cnt) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int32_t *co = coords + (size_t)b * 3 * N;
  int v = co[i] * R * R + co[i + N] * R + co[i + 2 * N];
  v = min(max(v, 0), S - 1);
  ind[(size_t)b * N + i] = v;
  atomicAdd(cnt + (size_t)b * S + v, 1);
}

__global__ __launch_bounds__(256) void vox_scatter_atomic_kernel(
    const float *__restrict__ feat, const int32_t *__restrict__ ind, const int32_t *__restrict__ cnt, int C,
    int N, int S, int CT, float *__restrict__ out) {
  const int b = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int pos = ind[(size_t)b * N + i];
  const int cur = cnt[(size_t)b * S + pos];
  if (cur <= 0) return;
  const float rcp = (float)(1.0 / (double)(float)cur);
  const int c0 = blockIdx.y * CT, c1 = min(C, c0 + CT);
  for (int c = c0; c < c1; ++c)
    atomicAdd(out + ((size_t)b * C + c) * S + pos, feat[((size_t)b * C + c) * N + i] * rcp);
}

}  // namespace pvcnn

using namespace pvcnn;

extern "C" size_t pvcnn_avg_voxelize_fwd_workspace_bytes(int B, int C, int N, int R) {
  if (B <= 0 || C < 0 || N < 0 || R <= 0) return 0;
  const long S = (long)R * R * R;
  if (!csr_supported((int)std::min<long>(S, 0x7fffffffL), N)) return 16;   // atomic fallback: no scratch
  return CsrWorkspace::bytes(B, C, (int)S, N, N);
}

extern "C" int pvcnn_avg_voxelize_fwd(const float *feat, const int32_t *coords, int B, int C, int N, int R,
                                      float *out, int32_t *ind, int32_t *cnt, void *workspace,
                                      size_t workspace_bytes, void *stream) {
  PVCNN_REQUIRE(B >= 0 && C >= 0 && N >= 0 && R > 0, "negative size");
  PVCNN_REQUIRE((long)R * R * R <= 0x7fffffffL / 4, "resolution too large");
  const int S = R * R * R;
  if (B == 0) return 0;
  PVCNN_REQUIRE((out || C == 0) && cnt && (ind || N == 0) && (feat || C == 0 || N == 0) && (coords || N == 0), "null pointer");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  hipStream_t s = static_cast<hipStream_t>(stream);

  if (!csr_supported(S, N)) {
    PVCNN_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)B * S * 4, s));
    PVCNN_HIP_TRY(hipMemsetAsync(out, 0, (size_t)B * C * S * 4, s));
    if (N == 0) return 0;
    hipLaunchKernelGGL(vox_stats_atomic_kernel, dim3(ceil_div(N, 256), B), dim3(256), 0, s, coords, N, R, S, ind, cnt);
    if (int e = check_launch("vox_stats_atomic")) return e;
    if (C == 0) return 0;
    const int CT = 16;
    hipLaunchKernelGGL(vox_scatter_atomic_kernel, dim3(ceil_div(N, 256), ceil_div(C, CT), B), dim3(256), 0, s,
                       feat, ind, cnt, C, N, S, CT, out);
    return check_launch("vox_scatter_atomic");
  }

  VoxelEntries ep{coords, ind, N, R, S};
  return launch_csr_scatter(ep, feat, out, B, C, /*L=*/S, /*J=*/N, /*E=*/N, cnt, workspace, workspace_bytes, s,
                            "avg_voxelize_fwd");
}

extern "C" int pvcnn_avg_voxelize_bwd(const float *grad_y, const int32_t *ind, const int32_t *cnt, int B, int C,
                                      int N, int S, float *grad_x, void *stream) {
  PVCNN_REQUIRE(B >= 0 && C >= 0 && N >= 0 && S > 0, "negative size");
  if (B == 0 || C == 0 || N == 0) return 0;
  PVCNN_REQUIRE(grad_y && ind && cnt && grad_x, "null pointer");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  VoxelMean p{ind, cnt, N, S};
  const bool vec = (N % 4 == 0) && aligned16(ind) && aligned16(grad_x);
  int pshift = 0;                                          // S = R^3 with R = 2^k >= 4: padded LDS rows
  for (int k = 2; k <= 10; ++k)
    if ((1L << (3 * k)) == (long)S) pshift = k;
  return launch_gather(p, grad_y, grad_x, B, C, /*L=*/S, /*J=*/N, vec, static_cast<hipStream_t>(stream),
                       "avg_voxelize_bwd", XfNone{}, pshift);
}
