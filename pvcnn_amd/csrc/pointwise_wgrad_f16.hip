// pointwise_wgrad_f16.hip -- backward-weight of the SharedMLP 1x1 convolutions on the fp16 matrix cores, "f16x2" arithmetic
// (split16.h: both fp32 operands scaled by a power of two, split into fp16 hi + lo, three exact partial products, fp32 accumulate).
//
//   grad_w[m][k] = sum over (b, n) of grad_y[b][m][n] * x[b][k][n]          (m = output channels, k = input channels, n = points)
//
// Both operands are channel-major with the reduction index (points) contiguous, which is exactly what v_mfma_f32_32x32x16_f16 wants
// (8 consecutive reduction elements per lane for a row of A and for a column of B): no transposition anywhere.
//   * a workgroup (4 waves, 2 x 2, 64 x 64 each) owns a 128 x 128 tile of grad_w for one partition of the points (split-K);
//   * per chunk of 32 points it converts a 128 x 32 slab of grad_y and of x (whole 128-byte lines in, fp16 hi / lo planes out, rows
//     padded to 80 bytes: the 16-byte fragment reads of 16 rows hit 64 distinct banks); the next chunk's slabs are in flight in
//     registers while this one is multiplied;
//   * partial tiles go to part[p] ([m][k]: 128-byte rows), pw_wgrad_f16_reduce_kernel sums the partitions in a fixed order
//     (deterministic, no atomics) and scales back by 2^-(sx + sgy); grad_bias falls out of the grad_y slabs a thread converts.
// N % 4 == 0 (16-byte loads); other shapes stay on the fp32-MFMA kernel of pointwise.hip.
// Measured at (16, 1472 -> 512, 4096): 0.61 ms (fp32 MFMA: 0.87).  Tried and slower: 128 x 256 tiles (64 x 128 per wave: 256 VGPRs and
// spills, 0.95 ms); two chunks in flight with CONDITIONAL loads (the compiler then waits on vmcnt(0) at every conversion: 1.09 ms).
#include <algorithm>

#include "common.h"
#include "split16.h"

namespace pvcnn {

constexpr int kGwM = 128, kGwK = 128, kGwPc = 32;
constexpr int kGwRowB = (kGwPc + 8) * 2;            // 80 bytes = 20 dwords
constexpr int kGwPlane = 128 * kGwRowB;             // one fp16 plane of one operand slab

__global__ __launch_bounds__(256, 2) void pw_wgrad_f16_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                              const uint32_t *__restrict__ x_absmax, const uint32_t *__restrict__ gy_absmax,
                                                              int K, int M, int N, int P, int ktiles, int cps, int total_chunks,
                                                              float *__restrict__ part, float *__restrict__ gb_part) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kGwPlane];     // [grad_y hi, lo][x hi, lo]
  unsigned char *gl = lds, *xl = lds + 2 * kGwPlane;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int wm = wave >> 1, wk = wave & 1;
  int bid = blockIdx.x;
  const int p = bid % P; bid /= P;
  const int kt = bid % ktiles, mt = bid / ktiles;
  const int m0 = mt * kGwM, k0 = kt * kGwK;
  const float x_scale = exp2_int(scale_shift(*x_absmax)), gy_scale = exp2_int(scale_shift(*gy_absmax));

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b2][r] = 0.0f;

  // staging: item u of a thread = row r0 + 32 u of the 256-row slab (rows 0..127 grad_y, 128..255 x), point quad q.  The loads are
  // branch-free (a row outside the tensor reads row 0 and is zeroed afterwards) so that the compiler can count them: TWO chunks
  // are in flight in registers -- a chunk's MFMAs take ~0.6 us, less than a miss to HBM.
  const int q = tid & 7, r0 = tid >> 3;
  const float *rowp[8];
  bool rowok[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int row = r0 + 32 * (u & 3);
    rowok[u] = u < 4 ? (m0 + row < M) : (k0 + row < K);
    rowp[u] = u < 4 ? gy + (size_t)(rowok[u] ? m0 + row : 0) * N : x + (size_t)(rowok[u] ? k0 + row : 0) * N;
  }
  float4 va[8], vb[8];
  float gsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const int c_begin = (int)((long)total_chunks * p / P), c_end = (int)((long)total_chunks * (p + 1) / P);
  auto load = [&](int c, float4 (&v)[8]) {                      // c may run past c_end: it then re-reads the last chunk (discarded)
    c = min(c, c_end - 1);
    const int b = c / cps, n = (c - b * cps) * kGwPc + 4 * q;
    const size_t og = (size_t)b * M * N + (n < N ? n : 0), ox = (size_t)b * K * N + (n < N ? n : 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(rowp[u] + (u < 4 ? og : ox));
  };
  auto convert = [&](int c, float4 (&v)[8]) {
    const int b = c / cps, n = (c - b * cps) * kGwPc + 4 * q;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (!(rowok[u] && n < N)) v[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      const int row = r0 + 32 * (u & 3);
      const float sc = u < 4 ? gy_scale : x_scale;
      uint32_t w0[2], w1[2];
      split_pair<2>(v[u].x * sc, v[u].y * sc, w0);
      split_pair<2>(v[u].z * sc, v[u].w * sc, w1);
      unsigned char *dst = (u < 4 ? gl : xl) + row * kGwRowB + 8 * q;
      *reinterpret_cast<uint2 *>(dst) = make_uint2(w0[0], w1[0]);
      *reinterpret_cast<uint2 *>(dst + kGwPlane) = make_uint2(w0[1], w1[1]);
      if (u < 4) gsum[u] += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
  };
  auto multiply = [&]() {
#pragma unroll
    for (int ks = 0; ks < kGwPc / 16; ++ks) {
      const int off = (ks * 16 + kh * 8) * 2;
      uint4 a[2][2], bq[2][2];                                  // [block][hi, lo]
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          a[blk][pl] = *reinterpret_cast<const uint4 *>(gl + pl * kGwPlane + (wm * 64 + blk * 32 + j) * kGwRowB + off);
          bq[blk][pl] = *reinterpret_cast<const uint4 *>(xl + pl * kGwPlane + (wk * 64 + blk * 32 + j) * kGwRowB + off);
        }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma16<2>(a[mb][1], bq[nb][0], acc[mb][nb]);      // lo x hi
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma16<2>(a[mb][0], bq[nb][1], acc[mb][nb]);      // hi x lo
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma16<2>(a[mb][0], bq[nb][0], acc[mb][nb]);      // hi x hi
    }
  };
  if (c_begin < c_end) {
    load(c_begin, va);
    load(c_begin + 1, vb);
  }
  for (int c = c_begin; c < c_end; c += 2) {
    __syncthreads();                                            // the previous chunk's fragment reads are done
    convert(c, va);
    __syncthreads();
    load(c + 2, va);                                            // lands two chunks of MFMAs later
    __builtin_amdgcn_sched_barrier(0);                          // keep the loads ahead of the MFMAs
    multiply();
    if (c + 1 < c_end) {
      __syncthreads();
      convert(c + 1, vb);
      __syncthreads();
      load(c + 3, vb);
      __builtin_amdgcn_sched_barrier(0);
      multiply();
    }
  }

  // ---- epilogue: part[p][m][k] over the padded (MP x KP) grid, lanes along k ----
  const int MP = (int)gridDim.x / (P * ktiles) * kGwM, KP = ktiles * kGwK;
  float *pp = part + (size_t)p * MP * KP;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        pp[(size_t)m * KP + k0 + wk * 64 + nb * 32 + j] = acc[mb][nb][r];
      }
  if (gb_part != nullptr && kt == 0) {                          // grad_bias partial: the 8 quads of a row, fixed order
    __syncthreads();
    float *red = reinterpret_cast<float *>(lds);                // [128 rows][8 quads]
#pragma unroll
    for (int u = 0; u < 4; ++u) red[(r0 + 32 * u) * 8 + q] = gsum[u];
    __syncthreads();
    if (tid < kGwM) {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += red[tid * 8 + i];
      gb_part[(size_t)p * MP + m0 + tid] = s;
    }
  }
}

__global__ __launch_bounds__(256) void pw_wgrad_f16_reduce_kernel(const float *__restrict__ part, const float *__restrict__ gb_part,
                                                                  const uint32_t *__restrict__ x_absmax, const uint32_t *__restrict__ gy_absmax,
                                                                  int P, int MP, int KP, int M, int K, float *__restrict__ gw,
                                                                  float *__restrict__ gb) {
  const size_t block = (size_t)MP * KP, e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e < block) {
    const int k = (int)(e % KP), m = (int)(e / KP);
    if (m < M && k < K) {
      float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
      int i = 0;
      for (; i + 3 < P; i += 4) {
        s0 += part[(size_t)i * block + e];
        s1 += part[(size_t)(i + 1) * block + e];
        s2 += part[(size_t)(i + 2) * block + e];
        s3 += part[(size_t)(i + 3) * block + e];
      }
      for (; i < P; ++i) s0 += part[(size_t)i * block + e];
      gw[(size_t)m * K + k] = ((s0 + s1) + (s2 + s3)) * exp2_int(-scale_shift(*x_absmax)) * exp2_int(-scale_shift(*gy_absmax));
    }
  }
  if (gb != nullptr && e < (size_t)M) {
    float s = 0.0f;
    for (int i = 0; i < P; ++i) s += gb_part[(size_t)i * MP + e];
    gb[e] = s;
  }
}

struct PwWgradPlan { int mtiles, ktiles, cps, total_chunks, P; size_t part_floats, gb_floats; };

static PwWgradPlan pw_wgrad_f16_plan(int B, int K, int M, int N) {
  PwWgradPlan w;
  w.mtiles = ceil_div(M, kGwM);
  w.ktiles = ceil_div(K, kGwK);
  w.cps = ceil_div(N, kGwPc);
  w.total_chunks = B * w.cps;
  w.P = std::max(1, std::min(w.total_chunks, ceil_div(512, w.mtiles * w.ktiles)));        // two workgroups per CU
  w.part_floats = (size_t)w.P * w.mtiles * kGwM * w.ktiles * kGwK;
  w.gb_floats = (size_t)w.P * w.mtiles * kGwM;
  return w;
}

}  // namespace pvcnn

using namespace pvcnn;

// 0 when the f16x2 kernel does not serve this shape (N % 4 != 0): use pvcnn_pwconv_bwd_weight
extern "C" size_t pvcnn_pwconv_bwd_weight_f16_workspace_bytes(int B, int K, int M, int N) {
  if (B <= 0 || K <= 0 || M <= 0 || N <= 0 || N % 4 != 0) return 0;
  const PwWgradPlan w = pw_wgrad_f16_plan(B, K, M, N);
  return (w.part_floats + w.gb_floats) * sizeof(float);
}

// grad_w (M, K) [, grad_bias (M)]:  x (B, K, N), grad_y (B, M, N);  *_absmax = pvcnn_absmax_bits of the two tensors
extern "C" int pvcnn_pwconv_bwd_weight_f16(const float *x, const float *grad_y, const void *x_absmax, const void *gy_absmax, int B, int K,
                                           int M, int N, float *grad_w, float *grad_bias, void *workspace, size_t workspace_bytes,
                                           void *stream) {
  PVCNN_REQUIRE(B > 0 && K > 0 && M > 0 && N > 0 && N % 4 == 0, "bad size (N must be a multiple of 4)");
  PVCNN_REQUIRE(x && grad_y && grad_w && x_absmax && gy_absmax, "null pointer");
  PVCNN_REQUIRE(aligned16(x) && aligned16(grad_y), "x and grad_y must be 16-byte aligned");
  PVCNN_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pvcnn_pwconv_bwd_weight_f16_workspace_bytes(B, K, M, N),
                "workspace missing, misaligned or too small (see pvcnn_pwconv_bwd_weight_f16_workspace_bytes)");
  const PwWgradPlan w = pw_wgrad_f16_plan(B, K, M, N);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint32_t *xa = static_cast<const uint32_t *>(x_absmax), *ga = static_cast<const uint32_t *>(gy_absmax);
  float *part = static_cast<float *>(workspace), *gb_part = part + w.part_floats;
  hipLaunchKernelGGL(pw_wgrad_f16_kernel, dim3((unsigned)(w.P * w.mtiles * w.ktiles)), dim3(256), 0, s, x, grad_y, xa, ga, K, M, N, w.P, w.ktiles,
                     w.cps, w.total_chunks, part, grad_bias ? gb_part : nullptr);
  if (int rc = check_launch("pwconv_wgrad_f16")) return rc;
  const int MP = w.mtiles * kGwM, KP = w.ktiles * kGwK;
  hipLaunchKernelGGL(pw_wgrad_f16_reduce_kernel, dim3((unsigned)ceil_div(MP * KP, 256)), dim3(256), 0, s, part, gb_part, xa, ga, w.P, MP, KP, M, K,
                     grad_w, grad_bias);
  return check_launch("pwconv_wgrad_f16_reduce");
}
