// neighbors.hip -- ball_query, grouping, gather, 3-NN interpolation for gfx950.
//
// Reference kernels: ball_query/ball_query.cu:19-50, grouping/grouping.cu:18-77,
// sampling/sampling.cu:17-66, interpolate/neighbor_interpolate.cu:20-170 -- all launched with
// one workgroup per cloud (B workgroups on a 256-CU chip).
//
// ball_query : one WAVE per group of kCPW centres; the wave's 64 lanes test 64 consecutive
//              points per step (coalesced 256-byte coordinate rows), __ballot + popcount of the
//              lower lanes gives each hit its slot in ascending point order -- exactly the
//              reference's sequential "first U hits" semantics, with coalesced index writes and
//              a wave-uniform early exit.  d^2 uses the reference's contraction fma(dz,dz,fma(dx,dx,dy*dy)).
// grouping / gather / 3-NN interpolate forward: LDS slab gathers (slab.h); a feature row of
//              N <= 40960 floats lives in LDS, so the U-fold re-reads of grouping never touch L2.
// their backwards: deterministic CSR scatters (csr.h), entry id = the reference's loop order.
// 3-NN search: one lane per query point, centres read through wave-uniform (scalar) loads.
#include "csr.h"

namespace pvcnn {

constexpr int kCPW = 4;   // centres per wave in ball_query (re-uses each loaded point 4x)

__global__ __launch_bounds__(256) void ball_query_kernel(const float *__restrict__ centers,
                                                         const float *__restrict__ points, int N, int M,
                                                         float r2, int U, int32_t *__restrict__ out) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256 + threadIdx.x) >> 6));
  const int m0 = wave * kCPW;
  if (m0 >= M) return;
  const float *pc = points + (size_t)b * 3 * N;
  const float *cc = centers + (size_t)b * 3 * M;
  int32_t *ni = out + ((size_t)b * M + m0) * U;

  float cx[kCPW], cy[kCPW], cz[kCPW];
  int cnt[kCPW], first[kCPW];
#pragma unroll
  for (int q = 0; q < kCPW; ++q) {
    const int m = min(m0 + q, M - 1);
    cx[q] = cc[m]; cy[q] = cc[m + M]; cz[q] = cc[m + 2 * M];
    cnt[q] = (m0 + q < M) ? 0 : U;   // out-of-range centres are born "full"
    first[q] = 0;
  }
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int k0 = 0; k0 < N; k0 += 64) {
    const int k = k0 + lane;
    const bool valid = k < N;
    const float px = valid ? pc[k] : 0.f, py = valid ? pc[k + N] : 0.f, pz = valid ? pc[k + 2 * N] : 0.f;
    bool all_full = true;
#pragma unroll
    for (int q = 0; q < kCPW; ++q) {
      if (cnt[q] < U) {   // wave-uniform
        const float dx = cx[q] - px, dy = cy[q] - py, dz = cz[q] - pz;
        const float d2 = fmaf(dz, dz, fmaf(dx, dx, dy * dy));   // = dx*dx + dy*dy + dz*dz as contracted
        const bool hit = valid && (d2 < r2);
        const unsigned long long mask = __ballot(hit);
        if (mask) {
          if (cnt[q] == 0) first[q] = k0 + __builtin_ctzll(mask);
          const int slot = cnt[q] + __popcll(mask & lt_mask);
          if (hit && slot < U) ni[(size_t)q * U + slot] = k;
          cnt[q] += __popcll(mask);
        }
        all_full = all_full && (cnt[q] >= U);
      }
    }
    if (all_full) break;
  }
  // padding: slots [cnt, U) repeat the first hit; rows without any hit are 0 (ball_query.cu:39-47)
#pragma unroll
  for (int q = 0; q < kCPW; ++q) {
    if (m0 + q >= M) continue;
    const int filled = min(cnt[q], U);
    for (int u = filled + lane; u < U; u += 64) ni[(size_t)q * U + u] = first[q];
  }
}

// 3-NN search: neighbor_interpolate.cu:20-75.  The reference keeps the running minima in double
// (init 1e40) but only ever compares them with float distances, so float minima with init +inf
// order identically; the clamp to [1e-10,1e10] and the double products rounded to float are
// reproduced exactly (a product of two floats is exact in double, so its float rounding equals
// the float product).
__global__ __launch_bounds__(256) void three_nn_kernel(const float *__restrict__ points_coords,
                                                       const float *__restrict__ centers_coords, int N, int M,
                                                       int32_t *__restrict__ indices, float *__restrict__ weights) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const float *pc = points_coords + (size_t)b * 3 * N;
  const float *cc = centers_coords + (size_t)b * 3 * M;
  const int jj = min(j, N - 1);
  const float ux = pc[jj], uy = pc[jj + N], uz = pc[jj + 2 * N];
  float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
  int i0 = 0, i1 = 0, i2 = 0;
  for (int k = 0; k < M; ++k) {
    const float x = cc[k], y = cc[k + M], z = cc[k + 2 * M];   // wave-uniform -> scalar loads
    const float ex = ux - x, ey = uy - y, ez = uz - z;
    const float d = fmaf(ez, ez, fmaf(ex, ex, ey * ey));
    if (d < b2) {
      if (d < b1) {
        b2 = b1; i2 = i1;
        if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = k; }
        else        { b1 = d;  i1 = k; }
      } else { b2 = d; i2 = k; }
    }
  }
  if (j >= N) return;
  b0 = fmaxf(fminf(1e10f, b0), 1e-10f);
  b1 = fmaxf(fminf(1e10f, b1), 1e-10f);
  b2 = fmaxf(fminf(1e10f, b2), 1e-10f);
  const float d0d1 = b0 * b1, d0d2 = b0 * b2, d1d2 = b1 * b2;
  const float inv = (float)(1.0 / (double)(d0d1 + d0d2 + d1d2));   // correctly rounded 1.0f / x
  float *w = weights + (size_t)b * 3 * N;
  int32_t *id = indices + (size_t)b * 3 * N;
  w[j] = d1d2 * inv;         id[j] = i0;
  w[j + N] = d0d2 * inv;     id[j + N] = i1;
  w[j + 2 * N] = d0d1 * inv; id[j + 2 * N] = i2;
}

}  // namespace pvcnn

using namespace pvcnn;

extern "C" int pvcnn_ball_query(const float *centers, const float *points, int B, int N, int M, float radius,
                                int U, int32_t *out, void *stream) {
  PVCNN_REQUIRE(B >= 0 && N >= 0 && M >= 0 && U >= 0, "negative size");
  if (B == 0 || M == 0 || U == 0) return 0;
  PVCNN_REQUIRE(centers && out && (points || N == 0), "null pointer");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  const float r2 = radius * radius;   // float, like ball_query.cpp:24
  const int waves = ceil_div(M, kCPW);
  hipLaunchKernelGGL(ball_query_kernel, dim3(ceil_div(waves, 4), B), dim3(256), 0, static_cast<hipStream_t>(stream),
                     centers, points, N, M, r2, U, out);
  return check_launch("ball_query");
}

extern "C" int pvcnn_grouping_fwd(const float *features, const int32_t *indices, int B, int C, int N, int M,
                                  int U, float *out, void *stream) {
  PVCNN_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0 && U >= 0, "negative size");
  PVCNN_REQUIRE((long)M * U <= 0x7fffffffL / 4, "M*U too large");
  const int J = M * U;
  if (B == 0 || C == 0 || J == 0) return 0;
  PVCNN_REQUIRE(features && indices && out && N > 0, "null pointer / empty feature row");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  IndexOnly p{indices, J};
  const bool vec = (J % 4 == 0) && aligned16(indices) && aligned16(out);
  return launch_gather(p, features, out, B, C, N, J, vec, static_cast<hipStream_t>(stream), "grouping_fwd");
}

extern "C" size_t pvcnn_grouping_bwd_workspace_bytes(int B, int C, int N, int M, int U) {
  if (B <= 0 || C < 0 || N <= 0 || M < 0 || U < 0) return 0;
  const long E = (long)M * U;
  if (!csr_supported(N, E)) return 16;
  return CsrWorkspace::bytes(B, C, N, (int)E, E);
}

extern "C" int pvcnn_grouping_bwd(const float *grad_y, const int32_t *indices, int B, int C, int N, int M, int U,
                                  float *grad_x, void *workspace, size_t workspace_bytes, void *stream) {
  PVCNN_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0 && U >= 0, "negative size");
  PVCNN_REQUIRE((long)M * U <= 0x7fffffffL / 8, "M*U too large");
  const int J = M * U;
  if (B == 0 || C == 0 || N == 0) return 0;
  PVCNN_REQUIRE(grad_x && (J == 0 || (grad_y && indices)), "null pointer");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!csr_supported(N, J)) {
    IndexOnly p{indices, J};
    return launch_scatter_direct(p, grad_y, grad_x, B, C, N, J, s, "grouping_bwd(atomic)");
  }
  IndexEntries ep{indices, J, N};
  return launch_csr_scatter(ep, grad_y, grad_x, B, C, /*L=*/N, /*J=*/J, /*E=*/J, nullptr, workspace, workspace_bytes, s,
                            "grouping_bwd");
}

extern "C" int pvcnn_gather_fwd(const float *features, const int32_t *indices, int B, int C, int N, int M,
                                float *out, void *stream) {
  return pvcnn_grouping_fwd(features, indices, B, C, N, M, 1, out, stream);
}

extern "C" size_t pvcnn_gather_bwd_workspace_bytes(int B, int C, int N, int M) {
  return pvcnn_grouping_bwd_workspace_bytes(B, C, N, M, 1);
}

extern "C" int pvcnn_gather_bwd(const float *grad_y, const int32_t *indices, int B, int C, int N, int M,
                                float *grad_x, void *workspace, size_t workspace_bytes, void *stream) {
  return pvcnn_grouping_bwd(grad_y, indices, B, C, N, M, 1, grad_x, workspace, workspace_bytes, stream);
}

extern "C" int pvcnn_three_nn_interp_fwd(const float *points_coords, const float *centers_coords,
                                         const float *centers_features, int B, int C, int M, int N,
                                         int32_t *indices, float *weights, float *out, void *stream) {
  PVCNN_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0, "negative size");
  if (B == 0 || N == 0) return 0;
  PVCNN_REQUIRE(points_coords && indices && weights && (centers_coords || M == 0), "null pointer");
  PVCNN_REQUIRE(M > 0, "no centres to interpolate from");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(three_nn_kernel, dim3(ceil_div(N, 256), B), dim3(256), 0, s, points_coords, centers_coords, N,
                     M, indices, weights);
  if (int e = check_launch("three_nn")) return e;
  if (C == 0) return 0;
  PVCNN_REQUIRE(centers_features && out, "null pointer");
  SavedTaps<3> p{indices, weights, N};
  const bool vec = (N % 4 == 0) && aligned16(indices) && aligned16(weights) && aligned16(out);
  return launch_gather(p, centers_features, out, B, C, /*L=*/M, /*J=*/N, vec, s, "three_nn_interp_fwd");
}

extern "C" size_t pvcnn_three_nn_interp_bwd_workspace_bytes(int B, int C, int N, int M) {
  if (B <= 0 || C < 0 || N < 0 || M <= 0) return 0;
  if (!csr_supported(M, 3L * N)) return 16;
  return CsrWorkspace::bytes(B, C, M, N, 3L * N);
}

extern "C" int pvcnn_three_nn_interp_bwd(const float *grad_y, const int32_t *indices, const float *weights, int B,
                                         int C, int N, int M, float *grad_x, void *workspace,
                                         size_t workspace_bytes, void *stream) {
  PVCNN_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0, "negative size");
  if (B == 0 || C == 0 || M == 0) return 0;
  PVCNN_REQUIRE(grad_x && (N == 0 || (grad_y && indices && weights)), "null pointer");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!csr_supported(M, 3L * N)) {
    SavedTaps<3> p{indices, weights, N};
    return launch_scatter_direct(p, grad_y, grad_x, B, C, M, N, s, "three_nn_interp_bwd(atomic)");
  }
  TapEntries<3> ep{indices, weights, N, M};
  return launch_csr_scatter(ep, grad_y, grad_x, B, C, /*L=*/M, /*J=*/N, /*E=*/3L * N, nullptr, workspace,
                            workspace_bytes, s, "three_nn_interp_bwd");
}
