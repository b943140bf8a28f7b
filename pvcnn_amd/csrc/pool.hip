// pool.hip -- the max over the K neighbours of a centre behind the SharedMLP of a set-abstraction level
// (reference: modules/pointnet.py:85, `mlp(grouper(...)).max(dim=-1).values` on a (B, C, M, K) tensor) and its backward.
//
// torch's reduction over a short innermost dimension reads the 67 MB tensor of PVCNN++'s first level at 0.6 TB/s (111 us), and
// its backward is a fill plus a scatter.  Here a row of K = 4 * LPR floats is one 16-byte load per lane of an LPR-lane group
// (a wave reads 1 KiB contiguous per instruction), the group's (value, k) maximum is an LPR-wide butterfly, and the backward
// writes the whole gradient in one pass (grad at the winner, zeros elsewhere): both directions are one streaming pass.
// Ties: the smallest k (torch's rule for max with indices); a NaN wins against numbers, like torch.max.
#include "common.h"

namespace pvcnn {

__device__ __forceinline__ bool pool_better(float vo, int ko, float v, int k) {
  const bool on = vo != vo, n = v != v;
  return on ? (!n || ko < k) : (!n && (vo > v || (vo == v && ko < k)));
}

template <int LPR>
__global__ __launch_bounds__(256) void neighbor_max_fwd_kernel(const float4 *__restrict__ x, long rows, float *__restrict__ out,
                                                               unsigned char *__restrict__ winners) {
  constexpr int U = 4;
  const long lanes = rows * LPR, first = (long)blockIdx.x * (256 * U) + threadIdx.x;
  float4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {                       // straight-line loads from clamped addresses: all four in flight
    const long i = first + u * 256;
    v[u] = x[i < lanes ? i : lanes - 1];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long i = first + u * 256;
    const int k0 = (int)(i & (LPR - 1)) * 4;
    float b = v[u].x;
    int k = k0;
    if (pool_better(v[u].y, k0 + 1, b, k)) { b = v[u].y; k = k0 + 1; }
    if (pool_better(v[u].z, k0 + 2, b, k)) { b = v[u].z; k = k0 + 2; }
    if (pool_better(v[u].w, k0 + 3, b, k)) { b = v[u].w; k = k0 + 3; }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) {
      const float bo = __shfl_xor(b, o);
      const int ko = __shfl_xor(k, o);
      if (pool_better(bo, ko, b, k)) { b = bo; k = ko; }
    }
    if (i < lanes && (i & (LPR - 1)) == 0) {
      const long row = i / LPR;
      out[row] = b;
      winners[row] = (unsigned char)k;
    }
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void neighbor_max_bwd_kernel(const float *__restrict__ grad_out, const unsigned char *__restrict__ winners,
                                                               long rows, float4 *__restrict__ grad_x) {
  constexpr int U = 4;
  const long lanes = rows * LPR, first = (long)blockIdx.x * (256 * U) + threadIdx.x;
  float g[U];
  int w[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long i = first + u * 256, row = (i < lanes ? i : lanes - 1) / LPR;
    g[u] = grad_out[row];
    w[u] = winners[row];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long i = first + u * 256;
    const int k0 = (int)(i & (LPR - 1)) * 4;
    if (i < lanes)
      grad_x[i] = make_float4(w[u] == k0 ? g[u] : 0.0f, w[u] == k0 + 1 ? g[u] : 0.0f, w[u] == k0 + 2 ? g[u] : 0.0f, w[u] == k0 + 3 ? g[u] : 0.0f);
  }
}

// arg-max over a LONG row (the global max-pool over the N points of a cloud, models/s3dis/pvcnn.py:41-43: `features.max(dim=-1)` on
// (B, C, N)): a wave per row, 1 KiB contiguous per load instruction, eight loads in flight, then a 6-step butterfly.  Same tie / NaN
// rule as above.  K % 4 == 0 (16-byte rows).
__global__ __launch_bounds__(256) void row_argmax_kernel(const float *__restrict__ x, long rows, int K, long long *__restrict__ winners,
                                                         float *__restrict__ values) {
  constexpr int U = 8;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;                                // (wave-uniform)
  const float4 *r4 = reinterpret_cast<const float4 *>(x + row * K);
  const int quads = K >> 2;
  // Rows are K * 4 bytes apart and every wave of the chip walks its row front to back at about the same pace: with all of them at
  // the same offset modulo the row length only a fraction of the HBM channels is busy at any moment (N = 4096: 16 KiB rows, the
  // un-rotated kernel -- and torch's reduction -- ran at 2.9 TB/s).  Row r therefore starts 1 KiB * r into the row and wraps around;
  // the maximum does not care about the order, the tie rule is explicit in pool_better.
  const int start = (int)((row * 64) % quads);
  float b = 0.0f;
  int k = -1;                                             // -1: nothing yet (loses against everything)
  for (int q0 = 0; q0 < quads; q0 += 64 * U) {
    float4 v[U];
    int qq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = min(q0 + u * 64 + lane, quads - 1) + start;
      qq[u] = q >= quads ? q - quads : q;
      v[u] = r4[qq[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (q0 + u * 64 + lane < quads) {
        const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (k < 0 || pool_better(e[i], 4 * qq[u] + i, b, k)) { b = e[i]; k = 4 * qq[u] + i; }
      }
    }
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float bo = __shfl_xor(b, o);
    const int ko = __shfl_xor(k, o);
    if (ko >= 0 && (k < 0 || pool_better(bo, ko, b, k))) { b = bo; k = ko; }
  }
  if (lane == 0) {
    winners[row] = k;
    if (values) values[row] = b;
  }
}

static int lanes_per_row(int K) { return (K >= 4 && K <= 64 && K % 4 == 0 && ((K / 4) & (K / 4 - 1)) == 0) ? K / 4 : 0; }

}  // namespace pvcnn

using namespace pvcnn;

extern "C" int pvcnn_neighbor_max_supported(int K) { return lanes_per_row(K) != 0; }

extern "C" int pvcnn_neighbor_max_fwd(const float *x, long rows, int K, float *out, unsigned char *winners, void *stream) {
  PVCNN_REQUIRE(rows >= 0 && K > 0, "negative size");
  if (rows == 0) return 0;
  const int lpr = lanes_per_row(K);
  PVCNN_REQUIRE(lpr != 0, "K must be 4, 8, 16, 32 or 64 (pvcnn_neighbor_max_supported)");
  PVCNN_REQUIRE(x && out && winners && aligned16(x), "null or misaligned pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long blocks = (rows * lpr + 1023) / 1024;
  PVCNN_REQUIRE(blocks <= 0x7fffffffL, "tensor too large");
  const float4 *x4 = reinterpret_cast<const float4 *>(x);
  switch (lpr) {
    case 1: hipLaunchKernelGGL(neighbor_max_fwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, x4, rows, out, winners); break;
    case 2: hipLaunchKernelGGL(neighbor_max_fwd_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, x4, rows, out, winners); break;
    case 4: hipLaunchKernelGGL(neighbor_max_fwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, x4, rows, out, winners); break;
    case 8: hipLaunchKernelGGL(neighbor_max_fwd_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, s, x4, rows, out, winners); break;
    default: hipLaunchKernelGGL(neighbor_max_fwd_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, x4, rows, out, winners); break;
  }
  return check_launch("neighbor_max_fwd");
}

extern "C" int pvcnn_neighbor_max_bwd(const float *grad_out, const unsigned char *winners, long rows, int K, float *grad_x, void *stream) {
  PVCNN_REQUIRE(rows >= 0 && K > 0, "negative size");
  if (rows == 0) return 0;
  const int lpr = lanes_per_row(K);
  PVCNN_REQUIRE(lpr != 0, "K must be 4, 8, 16, 32 or 64 (pvcnn_neighbor_max_supported)");
  PVCNN_REQUIRE(grad_out && winners && grad_x && aligned16(grad_x), "null or misaligned pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long blocks = (rows * lpr + 1023) / 1024;
  PVCNN_REQUIRE(blocks <= 0x7fffffffL, "tensor too large");
  float4 *g4 = reinterpret_cast<float4 *>(grad_x);
  switch (lpr) {
    case 1: hipLaunchKernelGGL(neighbor_max_bwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, grad_out, winners, rows, g4); break;
    case 2: hipLaunchKernelGGL(neighbor_max_bwd_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, grad_out, winners, rows, g4); break;
    case 4: hipLaunchKernelGGL(neighbor_max_bwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, grad_out, winners, rows, g4); break;
    case 8: hipLaunchKernelGGL(neighbor_max_bwd_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, s, grad_out, winners, rows, g4); break;
    default: hipLaunchKernelGGL(neighbor_max_bwd_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, grad_out, winners, rows, g4); break;
  }
  return check_launch("neighbor_max_bwd");
}

extern "C" int pvcnn_row_argmax(const float *x, long rows, int K, long long *winners, float *values, void *stream) {
  PVCNN_REQUIRE(rows >= 0 && K > 0, "negative size");
  if (rows == 0) return 0;
  PVCNN_REQUIRE(K % 4 == 0, "K must be a multiple of 4 (16-byte rows)");
  PVCNN_REQUIRE(x && winners && aligned16(x), "null or misaligned pointer");
  const long blocks = (rows + 3) / 4;
  PVCNN_REQUIRE(blocks <= 0x7fffffffL, "tensor too large");
  hipLaunchKernelGGL(row_argmax_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, rows, K, winners, values);
  return check_launch("row_argmax");
}
