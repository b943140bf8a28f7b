// pointwise.hip -- the SharedMLP 1x1 convolutions (modules/shared_mlp.py:9-25) as fp32-MFMA GEMMs for gfx950.
//
//   forward        y[b,m,n]  = sum_k W[m,k] x[b,k,n] + bias[m]        (m = Co, k = Ci, n = points)
//   backward-data  gx[b,k,n] = sum_m W[m,k] gy[b,m,n]                 (the same kernel, K and M exchanged)
//   backward-w     gW[m,k]   = sum_{b,n} gy[b,m,n] x[b,k,n],  gb[m] = sum_{b,n} gy[b,m,n]
//
// The tensors stay in the reference's channel-major (B,C,N) layout -- no NHWC round trips (the library path
// spends ~0.45 ms/step of PVCNN in layout transposes alone) -- and bias / bias-gradient ride on the GEMMs.
// All three run on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate).  Rules learned on the 3-D convolution
// kernels apply: staging is "16-byte load -> 16-byte LDS store" with nothing in between (so the loads of the next
// chunk can stay in flight across the MFMA loop), every LDS offset inside the K loop is an immediate, one LDS
// instruction feeds at least two MFMAs (ds_read2_b32 / ds_read_b64), the next step's operands are fetched
// under this step's MFMAs, and accumulators are read from their AGPRs only at the point of use.
#include <algorithm>

#include "common.h"

namespace pvcnn {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kPwN = 256;     // points per workgroup: 4 waves x 64 (2 MFMA column blocks each)
constexpr int kPwK = 32;      // reduction channels per LDS chunk

__device__ __forceinline__ float4 pw_ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// (M,K) -> (K32,M), K32 = K rounded up to 32 with zero rows: the forward kernel wants the weights k-major, and
// the zero rows let its unguarded fast path run a K that is not a multiple of the 32-channel chunk
__global__ __launch_bounds__(256) void pw_transpose_kernel(const float *__restrict__ w, int M, int K, float *__restrict__ wt) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8)
    tile[r][tx] = (m0 + r < M && k0 + tx < K) ? w[(size_t)(m0 + r) * K + k0 + tx] : 0.0f;
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (m0 + tx < M) wt[(size_t)(k0 + r) * M + m0 + tx] = tile[tx][r];   // rows K..K32-1 receive the zeros
}

// ---------------------------------------------------------------------------------------------
// y (B,M,N) = wt^T (K,M) . x (B,K,N) + bias.   1-D grid, XCD-aware tile order (below), block = 256.
// A workgroup computes 32*MB output channels x 256 points; wave w owns points [64w, 64w+64) as 2 MFMA column
// blocks and ALL MB row blocks (MB = 4: 128 channels, 8 accumulator tiles -- x is the big operand of this
// GEMM, so every staged x element should feed as many output channels as the register file allows;
// MB = 2 for layers with <= 64 output channels).
// LDS per chunk of 32 reduction channels: xs[32][256] and ws[32][32*MB], rows exactly as in global memory,
// so staging is "16-byte load, 16-byte LDS store" with nothing in between.  A K-step (2 channels, one per
// lane half) reads A at {0,32,..} and B at {0,32} floats from the lane's base with immediate offsets.
//   FAST (K % 32 == 0, N % 256 == 0, M % 32 == 0, 16-byte aligned): no per-lane bounds anywhere, and the K
//   chunks are software-pipelined -- chunk k+1's global loads are issued before chunk k's MFMA loop and
//   land in LDS after it.  Otherwise: bounds-checked scalar staging, same MFMA loop.
// ---------------------------------------------------------------------------------------------
template <int MB, bool FAST, bool BIAS>
__global__ __launch_bounds__(256) void pw_gemm_kernel(const float *__restrict__ x, const float *__restrict__ wt,
                                                      const float *__restrict__ bias, float *__restrict__ y, int K, int M,
                                                      int N, int tiles_n, int tiles_total, float2 *__restrict__ stats_part) {
  constexpr int TM = 32 * MB;
  __shared__ __attribute__((aligned(16))) float xs[kPwK * kPwN];
  __shared__ __attribute__((aligned(16))) float ws[kPwK * TM];
  // XCD-aware tile order.  Workgroup ids go round-robin over the 8 XCDs, each with its own L2; the
  // workgroups that read the SAME x tile (one per block of output channels) are given consecutive slots on
  // ONE XCD, so the tile comes from HBM once and from that L2 afterwards.
  const int mt_count = (M + TM - 1) / TM;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int m_idx = slot % mt_count, tile = (slot / mt_count) * 8 + xcd;
  if (tile >= tiles_total) return;
  const int b = tile / tiles_n, n0 = (tile - b * tiles_n) * kPwN;
  const int m0 = m_idx * TM;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const float *xb = x + (size_t)b * K * N;

  f32x16 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

  const float *a_base = ws + kh * TM + j;
  const float *b_base = xs + kh * kPwN + wave * 64 + j;

  // one K-chunk of MFMAs on what is in LDS; the next step's operands are fetched under this step's MFMAs
  auto mfma_chunk = [&]() {
    float a_cur[MB], b_cur[2];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) a_cur[mb] = a_base[mb * 32];
    b_cur[0] = b_base[0];
    b_cur[1] = b_base[32];
#pragma unroll
    for (int cc = 0; cc < kPwK; cc += 2) {
      const int cn = (cc + 2) & (kPwK - 1);               // the wrap-around prefetch of the last step is unused
      float a_nxt[MB], b_nxt[2];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) a_nxt[mb] = a_base[cn * TM + mb * 32];
      b_nxt[0] = b_base[cn * kPwN];
      b_nxt[1] = b_base[cn * kPwN + 32];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mb], b_cur[0], acc[mb][0], 0, 0, 0);
        acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mb], b_cur[1], acc[mb][1], 0, 0, 0);
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) a_cur[mb] = a_nxt[mb];
      b_cur[0] = b_nxt[0];
      b_cur[1] = b_nxt[1];
      // issue order: MFMA, LDS read, MFMA, LDS read, ...: the next step's reads complete under this step's MFMAs
#pragma unroll
      for (int g = 0; g < 1 + MB / 2; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * MB - (1 + MB / 2), 0);
    }
  };

  if constexpr (FAST) {
    // x: 32 rows x 64 quads = 2048 quads, 8 per thread; w: 32 rows x 8*MB quads, MB per thread
    float4 xq[8], wq[MB];
    auto load_chunk = [&](int k0, int t) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int q = t + it * 256;
        // K tail (K % 32 != 0): the row index is clamped; the matching weight rows are zero (wt is zero-padded)
        xq[it] = pw_ld4(xb + (size_t)min(k0 + (q >> 6), K - 1) * N + n0 + (q & 63) * 4);
      }
#pragma unroll
      for (int it = 0; it < MB; ++it) {
        const int q = t + it * 256;
        // rows beyond M (last tile of an M that is a multiple of 32 but not of 32*MB): the address is clamped
        // and the values are never used -- an output row depends only on its own A row and is not stored
        wq[it] = pw_ld4(wt + (size_t)(k0 + q / (8 * MB)) * M + min(m0 + (q % (8 * MB)) * 4, M - 4));
      }
    };
    auto store_chunk = [&](int t) {
#pragma unroll
      for (int it = 0; it < 8; ++it) *reinterpret_cast<float4 *>(xs + (t + it * 256) * 4) = xq[it];
#pragma unroll
      for (int it = 0; it < MB; ++it) *reinterpret_cast<float4 *>(ws + (t + it * 256) * 4) = wq[it];
    };
    load_chunk(0, tid);
    for (int k0 = 0; k0 < K; k0 += kPwK) {
      __syncthreads();                                     // everyone is done reading the previous chunk
      int t = tid;
      asm volatile("" : "+v"(t));                          // staging addresses recomputed per chunk, not kept live
      store_chunk(t);
      __syncthreads();
      if (k0 + kPwK < K) load_chunk(k0 + kPwK, t);
      mfma_chunk();
    }
  } else {
    for (int k0 = 0; k0 < K; k0 += kPwK) {
      __syncthreads();
      for (int e = tid; e < kPwK * kPwN; e += 256) {
        const int c = e / kPwN, n = n0 + (e - c * kPwN);
        xs[e] = (k0 + c < K && n < N) ? xb[(size_t)(k0 + c) * N + n] : 0.0f;
      }
      for (int e = tid; e < kPwK * TM; e += 256) {
        const int c = e / TM, m = m0 + (e - c * TM);
        ws[e] = (k0 + c < K && m < M) ? wt[(size_t)(k0 + c) * M + m] : 0.0f;
      }
      __syncthreads();
      mfma_chunk();
    }
  }

  // ---- epilogue: D[i = m][j = point]; lanes = consecutive points (128-byte rows) ----
  // stats_part != nullptr: per-channel (sum, sum of squares) of this workgroup's outputs ride on the epilogue, so
  // the BatchNorm that follows needs no statistics pass over y (partials (M, tiles_total), combined by bn_finalize).
  const bool want_stats = stats_part != nullptr;
  float2 *stat_lds = reinterpret_cast<float2 *>(xs);        // [4 waves][TM]
  if (want_stats) __syncthreads();                          // all waves are done reading xs / ws
  float *yb = y + (size_t)b * M * N;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    if (FAST && m0 + mb * 32 >= M) break;                  // FAST: M % 32 == 0, whole row blocks in or out
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      bv[r] = (BIAS && (FAST || m < M)) ? bias[m] : 0.0f;
    }
    float ss[16], qq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ss[r] = qq[r] = 0.0f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int n = n0 + wave * 64 + nb * 32 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        // the accumulator is fetched from its AGPR right here: left to the compiler, all of them are copied to
        // VGPRs in one block at the loop exit (+128 VGPRs for MB = 4: one wave per SIMD instead of two)
        float v;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[mb][nb][r]));
        if (want_stats) {                                   // statistics of (y - bias), see bn_finalize_kernel
          const float mv = (FAST || n < N) ? v : 0.0f;
          ss[r] += mv;
          qq[r] += mv * mv;
        }
        v += bv[r];
        if (FAST || (n < N && m < M)) yb[(size_t)m * N + n] = v;
      }
    }
    if (want_stats) {
      // lane j ends up with the totals of register (j >> 1) & 15 over its 32 voxels / points
      const float st = half_wave_sum16(ss, j), qt = half_wave_sum16(qq, j);
      const int rr = (j >> 1) & 15;
      if ((j & 1) == 0) stat_lds[wave * TM + mb * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * kh] = make_float2(st, qt);
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < TM && m0 + tid < M) {
      float2 t = stat_lds[tid];
#pragma unroll
      for (int w = 1; w < 4; ++w) { t.x += stat_lds[w * TM + tid].x; t.y += stat_lds[w * TM + tid].y; }
      stats_part[(size_t)(m0 + tid) * tiles_total + tile] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward-weight: gW[m][k] = sum over (b, n) of gy[b,m,n] * x[b,k,n]; the GEMM's reduction runs over points.
//   grid = (ceil(K/128), P, ceil(M/128)), block = 256: a workgroup owns a 128 x 128 tile of gW (each wave a
//   64 x 64 quarter = 2 x 2 MFMA blocks) and walks every P-th 32-point chunk of the B*N points, accumulating
//   in registers; partial tiles go to workspace[p], pw_reduce_kernel adds them up (no float atomics).
//   LDS per chunk: gys[128][36], xs[128][36] (row stride 36: 16-byte stores, 2-way conflicts at most on the
//   8-byte reads).  MFMA step (pair p, sub c) contracts points 4p + c (lanes 0-31) and 4p + 2 + c (lanes
//   32-63): a lane reads two consecutive floats per operand row and pair of steps.
//   The k-tile-0 workgroups also sum their gy rows: the bias gradient.
// ---------------------------------------------------------------------------------------------
constexpr int kPwWgT = 128;     // gW tile edge
constexpr int kPwWgC = 32;      // points per chunk
constexpr int kPwWgS = 36;      // LDS row stride (floats)

template <bool VEC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void pw_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                       float *__restrict__ part, float *__restrict__ bias_part, int B,
                                                       int K, int M, int N, int chunks_per_cloud, int P) {
  __shared__ __attribute__((aligned(16))) float gys[kPwWgT * kPwWgS];
  __shared__ __attribute__((aligned(16))) float xs[kPwWgT * kPwWgS];
  const int k0 = blockIdx.x * kPwWgT, p = blockIdx.y, m0 = blockIdx.z * kPwWgT;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int mh = wave & 1, nh = wave >> 1;                // this wave's 64 x 64 quarter

  f32x16 acc[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

  const float *a0 = gys + (mh * 64 + j) * kPwWgS + 2 * kh, *a1 = a0 + 32 * kPwWgS;
  const float *b0 = xs + (nh * 64 + j) * kPwWgS + 2 * kh, *b1 = b0 + 32 * kPwWgS;
  const bool do_bias = bias_part != nullptr && blockIdx.x == 0;
  float bsum = 0.0f;

  const int chunks_total = B * chunks_per_cloud;
  // one chunk of MFMAs on what is in LDS; the next pair's operands are fetched under this pair's MFMAs
  auto mfma_chunk = [&]() {
    float2 av0 = *reinterpret_cast<const float2 *>(a0), av1 = *reinterpret_cast<const float2 *>(a1);
    float2 bw0 = *reinterpret_cast<const float2 *>(b0), bw1 = *reinterpret_cast<const float2 *>(b1);
#pragma unroll
    for (int pr = 0; pr < kPwWgC / 4; ++pr) {
      const int pn = (pr + 1) & (kPwWgC / 4 - 1);          // the wrap-around prefetch of the last pair is unused
      const float2 an0 = *reinterpret_cast<const float2 *>(a0 + 4 * pn), an1 = *reinterpret_cast<const float2 *>(a1 + 4 * pn);
      const float2 bn0 = *reinterpret_cast<const float2 *>(b0 + 4 * pn), bn1 = *reinterpret_cast<const float2 *>(b1 + 4 * pn);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.x, bw0.x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.x, bw1.x, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.x, bw0.x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.x, bw1.x, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.y, bw0.y, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.y, bw1.y, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.y, bw0.y, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.y, bw1.y, acc[1][1], 0, 0, 0);
      av0 = an0; av1 = an1; bw0 = bn0; bw1 = bn1;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
  };
  auto bias_acc = [&]() {
    if (do_bias) {                                         // 2 threads per gy row, 16 points each
      const float *rowp = gys + (tid >> 1) * kPwWgS + (tid & 1) * 16;
#pragma unroll
      for (int i = 0; i < 16; ++i) bsum += rowp[i];
    }
  };

  if constexpr (VEC) {
    // software pipeline over the chunks: chunk ch+P's global loads are issued before chunk ch's MFMA loop and
    // land in LDS after it.  128 rows x 8 quads = 1024 quads per operand, 4 + 4 per thread.  Loads are
    // unconditional (row indices clamped into the tensors).
    float4 g[4], v[4];
    auto load_chunk = [&](int ch, int t) {
      const int b = ch / chunks_per_cloud, n0 = (ch - b * chunks_per_cloud) * kPwWgC;
      const float *gyb = gy + (size_t)b * M * N, *xb = x + (size_t)b * K * N;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = t + it * 256;
        const int row = q >> 3, n = n0 + (q & 7) * 4;
        g[it] = pw_ld4(gyb + (size_t)min(m0 + row, M - 1) * N + n);
        v[it] = pw_ld4(xb + (size_t)min(k0 + row, K - 1) * N + n);
      }
    };
    // (no zero-fill: N % 32 == 0 on this path, and rows beyond M / K only feed outputs that are never stored)
    auto store_chunk = [&](int t) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = t + it * 256;
        const int row = q >> 3, qi = q & 7;
        *reinterpret_cast<float4 *>(gys + row * kPwWgS + qi * 4) = g[it];
        *reinterpret_cast<float4 *>(xs + row * kPwWgS + qi * 4) = v[it];
      }
    };
    load_chunk(p, tid);                                    // P <= chunks_total: every workgroup has a first chunk
    for (int ch = p; ch < chunks_total; ch += P) {
      __syncthreads();
      int t = tid;
      asm volatile("" : "+v"(t));
      store_chunk(t);
      __syncthreads();
      if (ch + P < chunks_total) load_chunk(ch + P, t);
      mfma_chunk();
      bias_acc();
    }
  } else {
    for (int ch = p; ch < chunks_total; ch += P) {
      const int b = ch / chunks_per_cloud, n0 = (ch - b * chunks_per_cloud) * kPwWgC;
      const float *gyb = gy + (size_t)b * M * N, *xb = x + (size_t)b * K * N;
      __syncthreads();
      for (int e = tid; e < kPwWgT * kPwWgC; e += 256) {
        const int row = e / kPwWgC, i = e - row * kPwWgC;
        const int n = n0 + i;
        gys[row * kPwWgS + i] = (m0 + row < M && n < N) ? gyb[(size_t)(m0 + row) * N + n] : 0.0f;
        xs[row * kPwWgS + i] = (k0 + row < K && n < N) ? xb[(size_t)(k0 + row) * N + n] : 0.0f;
      }
      __syncthreads();
      mfma_chunk();
      bias_acc();
    }
  }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 1);
    const int m = m0 + (tid >> 1);
    if ((tid & 1) == 0 && m < M) bias_part[(size_t)p * M + m] = bsum;
  }
  float *out = part + (size_t)p * M * K;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int k = k0 + nh * 64 + nb * 32 + j;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mh * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        float val;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(val) : "a"(acc[mb][nb][r]));
        if (m < M && k < K) out[(size_t)m * K + k] = val;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward-weight for SMALL layers (M*K <= 128*128, e.g. the 9->64 / 64->64 / 64->128 point branches of PVConv):
// a 128 x 128 tile would be mostly padding there.  grid = (ceil(K/64), P, ceil(M/64)); a workgroup owns a
// 64 x 64 tile of gW and stages 64-point chunks; its 4 waves split the chunk's POINTS (16 each) and each
// keeps its own 2 x 2 accumulator tiles -- they leave as 4 separate partials (index 4p + wave), so the waves
// never have to add their tiles together; pw_reduce_kernel sums 4P partials.
// ---------------------------------------------------------------------------------------------
constexpr int kPwSmT = 64;      // gW tile edge
constexpr int kPwSmC = 64;      // points per chunk (16 per wave)
constexpr int kPwSmS = 68;      // LDS row stride (floats)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void pw_wgrad_small_kernel(
    const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ part, float *__restrict__ bias_part, int B,
    int K, int M, int N, int chunks_per_cloud, int P) {
  __shared__ __attribute__((aligned(16))) float gys[kPwSmT * kPwSmS];
  __shared__ __attribute__((aligned(16))) float xs[kPwSmT * kPwSmS];
  const int k0 = blockIdx.x * kPwSmT, p = blockIdx.y, m0 = blockIdx.z * kPwSmT;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

  const float *a0 = gys + j * kPwSmS + wave * 16 + 2 * kh, *a1 = a0 + 32 * kPwSmS;
  const float *b0 = xs + j * kPwSmS + wave * 16 + 2 * kh, *b1 = b0 + 32 * kPwSmS;
  const bool do_bias = bias_part != nullptr && blockIdx.x == 0;
  float bsum = 0.0f;
  const int chunks_total = B * chunks_per_cloud;

  // 64 rows x 16 quads = 1024 quads per operand, 4 + 4 per thread; N % 64 == 0 on this kernel (launcher)
  float4 g[4], v[4];
  auto load_chunk = [&](int ch, int t) {
    const int b = ch / chunks_per_cloud, n0 = (ch - b * chunks_per_cloud) * kPwSmC;
    const float *gyb = gy + (size_t)b * M * N, *xb = x + (size_t)b * K * N;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = t + it * 256;
      const int row = q >> 4, n = n0 + (q & 15) * 4;
      g[it] = pw_ld4(gyb + (size_t)min(m0 + row, M - 1) * N + n);
      v[it] = pw_ld4(xb + (size_t)min(k0 + row, K - 1) * N + n);
    }
  };
  load_chunk(p, tid);
  for (int ch = p; ch < chunks_total; ch += P) {
    __syncthreads();
    int t = tid;
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = t + it * 256;
      *reinterpret_cast<float4 *>(gys + (q >> 4) * kPwSmS + (q & 15) * 4) = g[it];
      *reinterpret_cast<float4 *>(xs + (q >> 4) * kPwSmS + (q & 15) * 4) = v[it];
    }
    __syncthreads();
    if (ch + P < chunks_total) load_chunk(ch + P, t);
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      const float2 av0 = *reinterpret_cast<const float2 *>(a0 + 4 * pr), av1 = *reinterpret_cast<const float2 *>(a1 + 4 * pr);
      const float2 bw0 = *reinterpret_cast<const float2 *>(b0 + 4 * pr), bw1 = *reinterpret_cast<const float2 *>(b1 + 4 * pr);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.x, bw0.x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.x, bw1.x, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.x, bw0.x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.x, bw1.x, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.y, bw0.y, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.y, bw1.y, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.y, bw0.y, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.y, bw1.y, acc[1][1], 0, 0, 0);
    }
    if (do_bias) {                                         // 4 threads per gy row, 16 points each
      const float *rowp = gys + (tid >> 2) * kPwSmS + (tid & 3) * 16;
#pragma unroll
      for (int i = 0; i < 16; ++i) bsum += rowp[i];
    }
  }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 1);
    bsum += __shfl_xor(bsum, 2);
    const int m = m0 + (tid >> 2);
    if ((tid & 3) == 0 && m < M) bias_part[(size_t)p * M + m] = bsum;
  }
  float *out = part + (size_t)(p * 4 + wave) * M * K;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int k = k0 + nb * 32 + j;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        float val;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(val) : "a"(acc[mb][nb][r]));
        if (m < M && k < K) out[(size_t)m * K + k] = val;
      }
  }
}

// out[e] = sum over the P partial tiles: a workgroup owns 32 consecutive elements, its eight 32-lane groups take every eighth partial
// (eight loads in flight each) and meet in LDS in a fixed order -- deterministic.  (First version: one thread per element walking all
// P partials: 16 workgroups for a 64 x 64 weight and 512 partials, 14 us per launch, 8 launches per PVCNN step.)
// A second array (the bias gradient's partials: n2 elements, P2 partials) rides in the same launch: the workgroups behind the first
// ceil(n / 32) take it -- one launch per backward-weight call instead of two (64 -> 32 launches per PVCNN++ step).
__global__ __launch_bounds__(256) void pw_reduce_kernel(const float *__restrict__ part, int n, int P, float *__restrict__ out,
                                                        const float *__restrict__ part2 = nullptr, int n2 = 0, int P2 = 0,
                                                        float *__restrict__ out2 = nullptr) {
  __shared__ float red[8][32];
  const int first = (n + 31) / 32;
  int blk = blockIdx.x;
  if (blk >= first) { blk -= first; part = part2; n = n2; P = P2; out = out2; }      // (uniform per workgroup)
  const int l = threadIdx.x & 31, g = threadIdx.x >> 5, e = blk * 32 + l;
  float s = 0.0f;
  if (e < n) {
    int p = g;
    for (; p + 56 < P; p += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(p + 8 * u) * n + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < P; p += 8) s += part[(size_t)p * n + e];
  }
  red[g][l] = s;
  __syncthreads();
  if (g == 0 && e < n) {
    float t = red[0][l];
#pragma unroll
    for (int i = 1; i < 8; ++i) t += red[i][l];
    out[e] = t;
  }
}

inline bool pw_wgrad_small(int K, int M, int N) { return (long)M * K <= 128L * 128L && N % kPwSmC == 0; }

// partitions of the point range; the small-layer kernel produces 4 partial tiles per partition (one per wave)
inline int pw_wgrad_partitions(int B, int K, int M, int N) {
  if (pw_wgrad_small(K, M, N)) {
    const int tiles = ceil_div(K, kPwSmT) * ceil_div(M, kPwSmT);
    const long chunks = (long)B * (N / kPwSmC);
    long P = std::max<long>(1, (2L * kNumCU) / tiles);
    P = std::min<long>(P, std::max<long>(1, chunks / 2));   // >= 2 chunks per workgroup: the pipeline needs a next one
    P = std::min<long>(P, 128);
    return (int)P;
  }
  const int tiles = ceil_div(K, kPwWgT) * ceil_div(M, kPwWgT);
  const long chunks = (long)B * ceil_div(N, kPwWgC);
  long P = std::max<long>(1, (2L * kNumCU) / tiles);      // ~2 workgroups per CU in the launch
  P = std::min<long>(P, std::max<long>(1, (32L << 20) / ((long)M * K * 4)));   // <= 32 MiB of partial tiles
  P = std::min<long>(P, chunks);
  return (int)P;
}

}  // namespace pvcnn

using namespace pvcnn;

extern "C" int pvcnn_pwconv_transpose(const float *w, int M, int K, float *wt, void *stream) {
  PVCNN_REQUIRE(M > 0 && K > 0 && w && wt, "bad argument");
  hipLaunchKernelGGL(pw_transpose_kernel, dim3(ceil_div(K, 32), ceil_div(M, 32)), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                     M, K, wt);
  return check_launch("pwconv_transpose");
}

static int pwconv_fwd_impl(const float *x, const float *wt, int wt_rows, const float *bias, int B, int K, int M, int N, float *y,
                           float2 *stats_part, hipStream_t s) {
  const int tiles_n = ceil_div(N, kPwN);
  const int MB = M > 64 ? 4 : 2;                         // output channels per workgroup: 128 or 64
  const long tiles_total = (long)B * tiles_n;
  const long wgs = ((tiles_total + 7) / 8) * 8 * ceil_div(M, 32 * MB);   // tiles padded to the 8 XCDs
  PVCNN_REQUIRE(wgs <= 0x7fffffffL, "grid too large");
  const dim3 grid((unsigned)wgs);
  // the fast path reads whole 32-row chunks of wt: K must be a multiple of 32 or wt zero-padded to one (wt_rows)
  const bool fast = (wt_rows >= ceil_div(K, kPwK) * kPwK) && (N % kPwN == 0) && (M % 32 == 0) && aligned16(x) && aligned16(wt);
#define PVCNN_PW_LAUNCH(MBV, FASTV)                                                                                   \
  do {                                                                                                                \
    if (bias) hipLaunchKernelGGL((pw_gemm_kernel<MBV, FASTV, true>), grid, dim3(256), 0, s, x, wt, bias, y, K, M, N, tiles_n, (int)tiles_total, stats_part);  \
    else      hipLaunchKernelGGL((pw_gemm_kernel<MBV, FASTV, false>), grid, dim3(256), 0, s, x, wt, bias, y, K, M, N, tiles_n, (int)tiles_total, stats_part); \
  } while (0)
  if (MB == 4) { if (fast) PVCNN_PW_LAUNCH(4, true); else PVCNN_PW_LAUNCH(4, false); }
  else         { if (fast) PVCNN_PW_LAUNCH(2, true); else PVCNN_PW_LAUNCH(2, false); }
#undef PVCNN_PW_LAUNCH
  return check_launch("pwconv_fwd");
}

extern "C" int pvcnn_pwconv_fwd(const float *x, const float *wt, int wt_rows, const float *bias, int B, int K, int M, int N,
                                float *y, void *stream) {
  PVCNN_REQUIRE(B >= 0 && K > 0 && M > 0 && N >= 0, "bad size");
  if (B == 0 || N == 0) return 0;
  PVCNN_REQUIRE(x && wt && y, "null pointer");
  PVCNN_REQUIRE((long)N * std::max(K, M) <= 0x7fffffffL, "cloud too large");
  PVCNN_REQUIRE(wt_rows >= K, "wt has fewer rows than K");
  return pwconv_fwd_impl(x, wt, wt_rows, bias, B, K, M, N, y, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" size_t pvcnn_pwconv_fwd_stats_parts(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return (size_t)B * ceil_div(N, kPwN);
}

extern "C" int pvcnn_pwconv_fwd_stats(const float *x, const float *wt, int wt_rows, const float *bias, int B, int K, int M,
                                      int N, float *y, float *stats_part, void *stream) {
  PVCNN_REQUIRE(B > 0 && K > 0 && M > 0 && N > 0, "bad size");
  PVCNN_REQUIRE(x && wt && y && stats_part, "null pointer");
  PVCNN_REQUIRE((reinterpret_cast<uintptr_t>(stats_part) & 7) == 0, "stats_part must be 8-byte aligned");
  PVCNN_REQUIRE((long)N * std::max(K, M) <= 0x7fffffffL, "cloud too large");
  PVCNN_REQUIRE(wt_rows >= K, "wt has fewer rows than K");
  return pwconv_fwd_impl(x, wt, wt_rows, bias, B, K, M, N, y, reinterpret_cast<float2 *>(stats_part), static_cast<hipStream_t>(stream));
}

extern "C" size_t pvcnn_pwconv_bwd_weight_workspace_bytes(int B, int K, int M, int N) {
  if (B <= 0 || K <= 0 || M <= 0 || N <= 0) return 16;
  const size_t P = (size_t)pw_wgrad_partitions(B, K, M, N);
  const size_t tiles_per_p = pw_wgrad_small(K, M, N) ? 4 : 1;
  return P * tiles_per_p * M * K * sizeof(float) + P * M * sizeof(float) + 16;
}

extern "C" int pvcnn_pwconv_bwd_weight(const float *x, const float *grad_y, int B, int K, int M, int N, float *grad_w,
                                       float *grad_bias, void *workspace, size_t workspace_bytes, void *stream) {
  PVCNN_REQUIRE(B >= 0 && K > 0 && M > 0 && N >= 0, "bad size");
  PVCNN_REQUIRE(grad_w, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B == 0 || N == 0) {
    PVCNN_HIP_TRY(hipMemsetAsync(grad_w, 0, (size_t)M * K * sizeof(float), s));
    if (grad_bias) PVCNN_HIP_TRY(hipMemsetAsync(grad_bias, 0, (size_t)M * sizeof(float), s));
    return 0;
  }
  PVCNN_REQUIRE(x && grad_y, "null pointer");
  PVCNN_REQUIRE((long)N * std::max(K, M) <= 0x7fffffffL, "cloud too large");
  PVCNN_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pvcnn_pwconv_bwd_weight_workspace_bytes(B, K, M, N),
                "workspace missing, misaligned or too small (see pvcnn_pwconv_bwd_weight_workspace_bytes)");
  const int P = pw_wgrad_partitions(B, K, M, N);
  const bool small = pw_wgrad_small(K, M, N) && aligned16(x) && aligned16(grad_y);
  const int PT = small ? 4 * P : P;                         // partial tiles to reduce
  float *part = static_cast<float *>(workspace);
  float *bias_part = grad_bias ? part + (size_t)(pw_wgrad_small(K, M, N) ? 4 : 1) * P * M * K : nullptr;
  if (small) {
    const dim3 grid(ceil_div(K, kPwSmT), P, ceil_div(M, kPwSmT));
    hipLaunchKernelGGL(pw_wgrad_small_kernel, grid, dim3(256), 0, s, x, grad_y, part, bias_part, B, K, M, N, N / kPwSmC, P);
  } else {
    const dim3 grid(ceil_div(K, kPwWgT), P, ceil_div(M, kPwWgT));
    const bool vec = (N % kPwWgC == 0) && aligned16(x) && aligned16(grad_y);   // whole chunks, 16-byte rows
    const int cpc = ceil_div(N, kPwWgC);
    if (vec) hipLaunchKernelGGL(pw_wgrad_kernel<true>, grid, dim3(256), 0, s, x, grad_y, part, bias_part, B, K, M, N, cpc, P);
    else     hipLaunchKernelGGL(pw_wgrad_kernel<false>, grid, dim3(256), 0, s, x, grad_y, part, bias_part, B, K, M, N, cpc, P);
  }
  if (int rc = check_launch("pwconv_wgrad")) return rc;
  const int n = M * K;
  if (grad_bias)
    hipLaunchKernelGGL(pw_reduce_kernel, dim3(ceil_div(n, 32) + ceil_div(M, 32)), dim3(256), 0, s, part, n, PT, grad_w, bias_part, M, P, grad_bias);
  else
    hipLaunchKernelGGL(pw_reduce_kernel, dim3(ceil_div(n, 32)), dim3(256), 0, s, part, n, PT, grad_w);
  return check_launch("pwconv_wgrad_reduce");
}
