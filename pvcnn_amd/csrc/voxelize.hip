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

// ---------------------------------------------------------------------------------------------
// Coordinate pre-pass of Voxelization.forward (modules/voxelization.py:16-25) in ONE launch instead of a
// dozen tiny library kernels:   c = coords - mean_N(coords);
//   normalize: c / (max_N ||c||_2 * 2 + eps) + 0.5      else: (c + 1) / 2
//   norm = clamp(c * R, 0, R - 1)  (float, kept for the devoxelization);  vox = rint(norm) (half to even)
// One workgroup per cloud; the 12 N bytes of a cloud stay in L2 over the three sweeps.  Every per-element
// expression is the reference's fp32 expression; the two reductions are order-free here (mean from an fp64
// sum = the correctly rounded mean; max is exact), where the reference's depend on the library's reduction tree.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void voxel_coords_kernel(const float *__restrict__ coords, int N, int R, int normalize,
                                                            float eps, float *__restrict__ norm_out,
                                                            int32_t *__restrict__ vox_out) {
  __shared__ double red[3][16];
  __shared__ float fred[16];
  __shared__ float stat[4];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float *c = coords + (size_t)b * 3 * N;
  double s[3] = {0.0, 0.0, 0.0};
  for (int i = tid; i < N; i += 1024) { s[0] += c[i]; s[1] += c[i + N]; s[2] += c[i + 2 * N]; }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s[a] += __shfl_xor(s[a], d);
    if (lane == 0) red[a][wave] = s[a];
  }
  __syncthreads();
  if (tid < 3) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[tid][w];
    stat[tid] = (float)(t / (double)N);
  }
  __syncthreads();
  const float mx = stat[0], my = stat[1], mz = stat[2];
  float denom = 1.0f;
  if (normalize) {
    float m2 = 0.0f;
    for (int i = tid; i < N; i += 1024) {
      const float x = c[i] - mx, y = c[i + N] - my, z = c[i + 2 * N] - mz;
      m2 = fmaxf(m2, (x * x + y * y) + z * z);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, d));
    if (lane == 0) fred[wave] = m2;
    __syncthreads();
    if (tid == 0) {
      float t = 0.0f;
      for (int w = 0; w < 16; ++w) t = fmaxf(t, fred[w]);
      stat[3] = sqrtf(t) * 2.0f + eps;
    }
    __syncthreads();
    denom = stat[3];
  }
  const float rf = (float)R, hi = (float)(R - 1);
  float *no = norm_out + (size_t)b * 3 * N;
  int32_t *vo = vox_out + (size_t)b * 3 * N;
  for (int i = tid; i < N; i += 1024) {
    float v[3] = {c[i] - mx, c[i + N] - my, c[i + 2 * N] - mz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float unit = normalize ? v[a] / denom + 0.5f : (v[a] + 1.0f) / 2.0f;
      const float g = fminf(fmaxf(unit * rf, 0.0f), hi);
      no[i + a * N] = g;
      vo[i + a * N] = (int32_t)rintf(g);
    }
  }
}


// ---------------------------------------------------------------------------------------------
// DEFAULT coordinate pre-pass: the two reductions of modules/voxelization.py:17-20 (mean over the points,
// max over the points of the centred norm) stay the caller's torch ops -- the reference's own calls on the same
// device, hence the same reduction trees -- and this kernel is the fused elementwise TAIL: centre, divide,
// + 0.5, * R, clamp, round-half-even.  Every expression is one separately rounded fp32 operation in the
// reference (each is its own torch kernel), and so it is here (-ffp-contract=off): norm / vox are bit-identical
// to Voxelization.forward's on the same device, for any reduction tree the library picks.
//   mean   (B,3)  = coords.mean(2)
//   radius (B)    = (coords - mean).norm(dim=1).max(dim=2)      (normalize only; else NULL)
// coords rows of one cloud are contiguous, clouds cstride floats apart (a channel slice of the input tensor).
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void voxel_coords_tail_kernel(const float *__restrict__ coords, long cstride,
                                                               const float *__restrict__ mean,
                                                               const float *__restrict__ radius, int N, int R, float eps,
                                                               float *__restrict__ norm_out, int32_t *__restrict__ vox_out) {
  const int b = blockIdx.z, a = blockIdx.y, i0 = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (i0 >= N) return;
  const float m = mean[b * 3 + a];
  const bool normalize = radius != nullptr;
  float denom = 1.0f;
  if (normalize) { denom = radius[b] * 2.0f; denom = denom + eps; }
  const float rf = (float)R, hi = (float)(R - 1);
  const float *src = coords + (size_t)b * cstride + (size_t)a * N + i0;
  float x[VEC];
  if constexpr (VEC == 4) { const float4 v = *reinterpret_cast<const float4 *>(src); x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w; }
  else x[0] = src[0];
  float g[VEC];
  int32_t q[VEC];
#pragma unroll
  for (int u = 0; u < VEC; ++u) {
    const float c = x[u] - m;
    float unit;
    if (normalize) { unit = c / denom; unit = unit + 0.5f; }
    else { unit = c + 1.0f; unit = unit * 0.5f; }      // torch: division by the host scalar 2.0 = multiplication by 0.5
    const float sc = unit * rf;
    g[u] = (sc != sc) ? sc : fminf(fmaxf(sc, 0.0f), hi);   // torch.clamp propagates NaN (degenerate cloud, eps = 0)
    q[u] = (int32_t)rintf(g[u]);
  }
  const size_t o = ((size_t)b * 3 + a) * N + i0;
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4 *>(norm_out + o) = make_float4(g[0], g[1], g[2], g[3]);
    *reinterpret_cast<int4 *>(vox_out + o) = make_int4(q[0], q[1], q[2], q[3]);
  } else { norm_out[o] = g[0]; vox_out[o] = q[0]; }
}

}  // namespace pvcnn

using namespace pvcnn;

extern "C" int pvcnn_voxel_coords(const float *coords, int B, int N, int R, int normalize, float eps, float *norm_coords,
                                  int32_t *vox_coords, void *stream) {
  PVCNN_REQUIRE(B >= 0 && N >= 0 && R > 0, "negative size");
  if (B == 0 || N == 0) return 0;
  PVCNN_REQUIRE(coords && norm_coords && vox_coords, "null pointer");
  hipLaunchKernelGGL(voxel_coords_kernel, dim3(B), dim3(1024), 0, static_cast<hipStream_t>(stream), coords, N, R, normalize, eps,
                     norm_coords, vox_coords);
  return check_launch("voxel_coords");
}

extern "C" int pvcnn_voxel_coords_tail(const float *coords, long coords_batch_stride, const float *mean, const float *radius,
                                       int B, int N, int R, float eps, float *norm_coords, int32_t *vox_coords, void *stream) {
  PVCNN_REQUIRE(B >= 0 && N >= 0 && R > 0, "negative size");
  if (B == 0 || N == 0) return 0;
  PVCNN_REQUIRE(coords && mean && norm_coords && vox_coords, "null pointer");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  PVCNN_REQUIRE(coords_batch_stride >= 3L * N, "coords batch stride smaller than one cloud");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = (N % 4 == 0) && (coords_batch_stride % 4 == 0) && aligned16(coords) && aligned16(norm_coords) && aligned16(vox_coords);
  if (vec)
    hipLaunchKernelGGL(voxel_coords_tail_kernel<4>, dim3(ceil_div(N, 1024), 3, B), dim3(256), 0, s, coords, coords_batch_stride,
                       mean, radius, N, R, eps, norm_coords, vox_coords);
  else
    hipLaunchKernelGGL(voxel_coords_tail_kernel<1>, dim3(ceil_div(N, 256), 3, B), dim3(256), 0, s, coords, coords_batch_stride,
                       mean, radius, N, R, eps, norm_coords, vox_coords);
  return check_launch("voxel_coords_tail");
}

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

// ---- plan / apply split: one counting sort per (voxel coordinates, R), shared by every layer that voxelizes with them ----
extern "C" size_t pvcnn_avg_voxelize_plan_bytes(int B, int N, int R) {
  if (B <= 0 || N < 0 || R <= 0) return 0;
  const long S = (long)R * R * R;
  if (S > 0x7fffffffL / 4 || !csr_supported((int)S, N)) return 0;      // 0: no plan for this size (use the one-shot call)
  return CsrPlan::bytes(B, (int)S, N);
}

extern "C" size_t pvcnn_avg_voxelize_plan_scratch_bytes(int B, int N, int R) {
  if (B <= 0 || N < 0 || R <= 0) return 0;
  return csr_prep_scratch_bytes(B, N);
}

extern "C" int pvcnn_avg_voxelize_plan(const int32_t *coords, int B, int N, int R, int32_t *ind, int32_t *cnt, void *plan,
                                       size_t plan_bytes, void *scratch, size_t scratch_bytes, void *stream) {
  PVCNN_REQUIRE(B >= 0 && N >= 0 && R > 0, "negative size");
  PVCNN_REQUIRE((long)R * R * R <= 0x7fffffffL / 4, "resolution too large");
  const int S = R * R * R;
  if (B == 0) return 0;
  PVCNN_REQUIRE(csr_supported(S, N), "grid too large for a plan: use pvcnn_avg_voxelize_fwd");
  PVCNN_REQUIRE(cnt && (ind || N == 0) && (coords || N == 0), "null pointer");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  VoxelEntries ep{coords, ind, N, R, S};
  return launch_csr_prep(ep, B, /*L=*/S, /*E=*/N, cnt, plan, plan_bytes, scratch, scratch_bytes, static_cast<hipStream_t>(stream),
                         "avg_voxelize_plan");
}

// (ABI v11) BOTH scatter plans of one PVConv geometry -- avg_voxelize's (from the integer voxel coordinates) and
// trilinear_devoxelize backward's (from the float grid coordinates: the corner entries the forward will emit as inds / wgts) -- in one
// chain of three launches instead of two chains of three.  Outputs exactly what pvcnn_avg_voxelize_plan and
// pvcnn_trilinear_devox_bwd_plan write (the same plan bytes where a plan defines them, the same ind / cnt).
extern "C" size_t pvcnn_pvconv_plans_scratch_bytes(int B, int N, int R) {
  if (B <= 0 || N < 0 || R <= 0) return 0;
  return csr_prep_scratch_bytes(B, N) + csr_prep_scratch_bytes(B, 8L * N);
}

extern "C" int pvcnn_pvconv_plans(const int32_t *vox_coords, const float *norm_coords, int B, int N, int R, int32_t *ind, int32_t *cnt,
                                  void *vox_plan, size_t vox_plan_bytes, void *devox_bwd_plan, size_t devox_bwd_plan_bytes, void *scratch,
                                  size_t scratch_bytes, void *stream) {
  PVCNN_REQUIRE(B >= 0 && N > 0 && R > 0, "bad size");
  PVCNN_REQUIRE((long)R * R * R <= 0x7fffffffL / 4, "resolution too large");
  const int S = R * R * R;
  if (B == 0) return 0;
  PVCNN_REQUIRE(csr_supported(S, 8L * N), "grid too large for a plan: use the one-shot calls");
  PVCNN_REQUIRE(cnt && ind && vox_coords && norm_coords, "null pointer");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  VoxelEntries e1{vox_coords, ind, N, R, S};
  CoordTapEntries e2{norm_coords, N, R, R * R, S};
  return launch_csr_prep_pair(e1, /*E1=*/N, cnt, vox_plan, vox_plan_bytes, e2, /*E2=*/8L * N, devox_bwd_plan, devox_bwd_plan_bytes, B, /*L=*/S,
                              scratch, scratch_bytes, static_cast<hipStream_t>(stream), "pvconv_plans");
}

extern "C" int pvcnn_avg_voxelize_apply(const float *feat, const void *plan, size_t plan_bytes, int B, int C, int N, int R,
                                        float *out, void *stream) {
  PVCNN_REQUIRE(B >= 0 && C >= 0 && N >= 0 && R > 0, "negative size");
  PVCNN_REQUIRE((long)R * R * R <= 0x7fffffffL / 4, "resolution too large");
  if (B == 0 || C == 0) return 0;
  PVCNN_REQUIRE(out && (feat || N == 0), "null pointer");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  const int S = R * R * R;
  return launch_csr_apply(feat, plan, plan_bytes, out, B, C, /*L=*/S, /*J=*/N, /*E=*/N, static_cast<hipStream_t>(stream),
                          "avg_voxelize_apply");
}
