// pointwise_bf16.hip -- the SharedMLP 1x1 convolutions (forward / backward-data) on the bf16 matrix cores of gfx950.
//
//   y[b,m,n] = sum_k W[m,k] x[b,k,n] + bias[m]          (m = output channels, k = reduction channels, n = points)
//
// Same arithmetic as conv3d_bf16.hip: NS = 3 "bf16x3" -- both fp32 operands split exactly into three bf16 pieces, the six
// partial products of weight >= 2^-16 accumulated in fp32 (fp32-class accuracy, <= 1e-5 vs fp64 like the fp32-MFMA kernels
// of pointwise.hip) at up to 2.7x the fp32-MFMA rate -- or NS = 1, plain bf16 operands.  It is the implicit GEMM of
// conv3d_bf16.hip with a single tap and no halo:
//   * a workgroup owns 256 consecutive points x 32*MB output channels (MB = 4: x is the big operand, every staged and
//     converted element should feed as many output channels as the register file allows);
//   * per chunk of 16 reduction channels the 256 x 16 input tile is staged once, fp32 -> NS bf16 planes, as packed channel
//     PAIRS with points contiguous: xs[plane][channel pair][point] (coalesced 16-byte loads in, 16-byte LDS writes, and the
//     operand reads are conflict-free);
//   * the weights come from a pre-split, pre-swizzled image in global memory (pw_weight_split_kernel; L2-resident), each
//     lane fetching its own 16-byte A fragments, requested BEFORE the tile is staged so they land behind the staging;
//   * C/D rows are 32 consecutive points of one channel = 128-byte rows of the channel-major (B, M, N) output; bias and the
//     optional BatchNorm partial sums of (y - bias) ride on the epilogue.
// Backward-data is the same kernel on the transposed weights (for_bwd_data image), x = grad_y.
//
// History of the "convert + LDS + barrier" phase that does not overlap the MFMA phase (0.29 of 0.69 ms at the classifier shape in the
// bf16x3 form, round 2): one-chunk-ahead register prefetch, 32-channel stages, XCD-aware tile order (kept: x is fetched from HBM
// once) and issue priorities did not move it; round 3 found the cause in the ISA -- the compiler drained every load (vmcnt(0)) in front
// of each chunk's first MFMA because the prefetch sat behind bounds checks -- and moved the conversion between the MFMAs
// (pw_gemm_f16_pipe_kernel below).  Arithmetic is selected with `backend.pw_math` (f16x2 default, bf16x3, fp32).
#include <algorithm>
#include <type_traits>
#include <stdlib.h>

#include "common.h"
#include <string.h>
#include "split16.h"

namespace pvcnn {

constexpr int kPbN = 256;          // points per workgroup
constexpr int kPbK = 16;           // reduction channels per chunk = MFMA K

__device__ __forceinline__ uint32_t pb_bf16_bits(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
template <int NS>
__device__ __forceinline__ void pb_split(float v, uint32_t (&p)[NS]) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    p[s] = pb_bf16_bits(v);
    if (s + 1 < NS) v = v - __uint_as_float(p[s] << 16);      // exact residual
  }
}

// W (Mo, Ko) fp32 [forward: Mo = Co, Ko = Ci;  for_bwd_data: the GEMM's output channels are Ci and it reduces over Co, W'[ci][co] =
// W[co][ci]] -> image [chunk][mtile][plane][TM rows][16 k (halves swizzled by bit 3 of the row)] bf16, TM = 32 * MB
template <int NS>
__device__ __forceinline__ void pw_weight_split_elem(const float *__restrict__ w, int Co, int Ci, int for_bwd_data, int TM,
                                                     uint16_t *__restrict__ wts, long e) {
  const int KE = for_bwd_data ? Co : Ci, ME = for_bwd_data ? Ci : Co;
  const int chunks = ceil_div(KE, kPbK), mtiles = ceil_div(ME, TM);
  const long total = (long)chunks * mtiles * TM * kPbK;
  if (e >= total) return;
  const int k_l = (int)(e % kPbK), row = (int)((e / kPbK) % TM);
  const long rest = e / ((long)kPbK * TM);
  const int mt = (int)(rest % mtiles), chunk = (int)(rest / mtiles);
  const int k = chunk * kPbK + k_l, m = mt * TM + row;
  float v = 0.0f;
  if (k < KE && m < ME) v = for_bwd_data ? w[(size_t)k * Ci + m] : w[(size_t)m * Ci + k];
  uint32_t p[NS];
  pb_split<NS>(v, p);
  const int pos = ((k_l >> 3) ^ ((row >> 3) & 1)) * 8 + (k_l & 7);
  const size_t blk = ((size_t)chunk * mtiles + mt) * ((size_t)NS * TM * kPbK);
#pragma unroll
  for (int s = 0; s < NS; ++s) wts[blk + ((size_t)s * TM + row) * kPbK + pos] = (uint16_t)p[s];
}

template <int NS>
__global__ __launch_bounds__(256) void pw_weight_split_kernel(const float *__restrict__ w, int Co, int Ci, int for_bwd_data, int TM,
                                                              uint16_t *__restrict__ wts) {
  pw_weight_split_elem<NS>(w, Co, Ci, for_bwd_data, TM, wts, (long)blockIdx.x * 256 + threadIdx.x);
}

// plain-bf16 (torch.autocast) images of every registered 1x1 weight in one launch (see conv3d_weight_split_bf16_batch_kernel): an
// entry's rows are 256-element blocks, the forward image's first
__global__ __launch_bounds__(256) void pw_weight_split_bf16_batch_kernel(const SplitEntry *__restrict__ tab, int n) {
  const long long blk = blockIdx.x;
  int i = 0;
  while (i + 1 < n && tab[i + 1].row_begin <= blk) ++i;
  const SplitEntry e = tab[i];
  const long local = (long)(blk - e.row_begin);
  if (local < (long)e.rows_f) pw_weight_split_elem<1>(e.w, (int)e.Co, (int)e.Ci, 0, (int)(e.tm & 0xffffffffLL), e.wts_f, local * 256 + threadIdx.x);
  else pw_weight_split_elem<1>(e.w, (int)e.Co, (int)e.Ci, 1, (int)(e.tm >> 32), e.wts_b, (local - (long)e.rows_f) * 256 + threadIdx.x);
}

// f16x2 image: one workgroup per (padded) output row finds the row's max |w|, scales by the power of two of scale_shift and writes the
// fp16 hi / lo planes; wexp[row] = the shift (the epilogue scales back per output channel).
__device__ __forceinline__ void pw_weight_split_f16_row(const float *__restrict__ w, int Co, int Ci, int for_bwd_data, int TM,
                                                        uint16_t *__restrict__ wts, int *__restrict__ wexp, int m) {
  const int KE = for_bwd_data ? Co : Ci, ME = for_bwd_data ? Ci : Co;
  const int chunks = ceil_div(KE, kPbK), mtiles = ceil_div(ME, TM);
  const int mt = m / TM, row = m - mt * TM, tid = threadIdx.x;
  auto load = [&](int k) { return for_bwd_data ? w[(size_t)k * Ci + m] : w[(size_t)m * Ci + k]; };
  __shared__ uint32_t red[4];
  uint32_t mx = 0;
  if (m < ME)
    for (int k = tid; k < KE; k += 256) mx = max(mx, __float_as_uint(fabsf(load(k))));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  const int shift = scale_shift(max(max(red[0], red[1]), max(red[2], red[3])));
  if (tid == 0) wexp[m] = shift;
  const float scale = exp2_int(shift);
  for (int i = tid; i < chunks * 8; i += 256) {                 // item = (chunk, channel pair)
    const int cp = i & 7, chunk = i >> 3, k = chunk * kPbK + 2 * cp;
    const float a = (m < ME && k < KE) ? load(k) * scale : 0.0f, b = (m < ME && k + 1 < KE) ? load(k + 1) * scale : 0.0f;
    uint32_t p[2];
    split_pair<2>(a, b, p);
    const int word = ((cp >> 2) ^ ((row >> 3) & 1)) * 4 + (cp & 3);
    uint32_t *img = reinterpret_cast<uint32_t *>(wts + ((size_t)chunk * mtiles + mt) * ((size_t)2 * TM * kPbK));
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) img[((size_t)s2 * TM + row) * 8 + word] = p[s2];
  }
}

__global__ __launch_bounds__(256) void pw_weight_split_f16_kernel(const float *__restrict__ w, int Co, int Ci, int for_bwd_data, int TM,
                                                                  uint16_t *__restrict__ wts, int *__restrict__ wexp) {
  pw_weight_split_f16_row(w, Co, Ci, for_bwd_data, TM, wts, wexp, blockIdx.x);
}

// forward AND backward-data image of one weight in one launch
__global__ __launch_bounds__(256) void pw_weight_split_f16_pair_kernel(const float *__restrict__ w, int Co, int Ci, int rows_fwd, int TM_f,
                                                                       int TM_b, uint16_t *__restrict__ wts_f, int *__restrict__ wexp_f,
                                                                       uint16_t *__restrict__ wts_b, int *__restrict__ wexp_b) {
  if ((int)blockIdx.x < rows_fwd) pw_weight_split_f16_row(w, Co, Ci, 0, TM_f, wts_f, wexp_f, blockIdx.x);
  else pw_weight_split_f16_row(w, Co, Ci, 1, TM_b, wts_b, wexp_b, blockIdx.x - rows_fwd);
}

// ... of every registered 1x1 weight of a model in one launch (see conv3d_weight_split_f16_batch_kernel)
__global__ __launch_bounds__(256) void pw_weight_split_f16_batch_kernel(const SplitEntry *__restrict__ tab, int n) {
  const long long blk = blockIdx.x;
  int i = 0;
  while (i + 1 < n && tab[i + 1].row_begin <= blk) ++i;
  const SplitEntry e = tab[i];
  const int row = (int)(blk - e.row_begin);
  if (row < (int)e.rows_f) pw_weight_split_f16_row(e.w, (int)e.Co, (int)e.Ci, 0, (int)(e.tm & 0xffffffffLL), e.wts_f, e.wexp_f, row);
  else pw_weight_split_f16_row(e.w, (int)e.Co, (int)e.Ci, 1, (int)(e.tm >> 32), e.wts_b, e.wexp_b, row - (int)e.rows_f);
}

// ---- epilogue shared by the 1x1 GEMM kernels: D[i = m][j = point]; lanes = consecutive points (128-byte rows); bias; the optional
// BatchNorm partial sums of (y - bias).  `lds` = the workgroup's staging buffer (>= 4 / WM * TM float pairs), free by now.
template <int NS, int MB, int WM>
__device__ __forceinline__ void pb_epilogue(f32x16 (&acc)[MB / WM][2 * WM], uint32_t *lds, const float *__restrict__ bias,
                                            float *__restrict__ y, int M, int N, int b, int n0, int m0, int tile, int tiles_total,
                                            float2 *__restrict__ stats_part, const int *__restrict__ wexp, int x_shift) {
  constexpr int TM = 32 * MB, MBW = MB / WM, NBW = 2 * WM;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int wm = wave % WM, wn = wave / WM;
  uint32_t *xs = lds;
  const bool want_stats = stats_part != nullptr;
  float2 *stat_lds = reinterpret_cast<float2 *>(xs);            // [4 / WM point groups][TM]
  if (want_stats) __syncthreads();
  float *yb = y + (size_t)b * M * N;
#pragma unroll
  for (int mbl = 0; mbl < MBW; ++mbl) {
    const int mb = wm * MBW + mbl;                              // row block inside the workgroup tile
    float bv[16], unscale[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      bv[r] = (bias != nullptr && m < M) ? bias[m] : 0.0f;
      if constexpr (NS == 2) unscale[r] = exp2_int(-wexp[m]);   // wexp covers the padded rows of the tile
    }
    const float x_unscale = exp2_int(-x_shift);
    float ss[16], qq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ss[r] = qq[r] = 0.0f;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      const int n = n0 + wn * (32 * NBW) + nb * 32 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        float v = acc[mbl][nb][r];
        if constexpr (NS == 2) v = v * unscale[r] * x_unscale;  // powers of two: exact
        if (want_stats) {                                       // statistics of (y - bias), see bn_finalize_kernel
          const float mv = n < N ? v : 0.0f;
          ss[r] += mv;
          qq[r] += mv * mv;
        }
        v += bv[r];
        if (n < N && m < M) yb[(size_t)m * N + n] = v;
      }
    }
    if (want_stats) {
      const float st = half_wave_sum16(ss, j), qt = half_wave_sum16(qq, j);
      const int rr = (j >> 1) & 15;
      if ((j & 1) == 0) stat_lds[wn * TM + mb * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * kh] = make_float2(st, qt);
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < TM && m0 + tid < M) {
      float2 t = stat_lds[tid];
#pragma unroll
      for (int w = 1; w < 4 / WM; ++w) { t.x += stat_lds[w * TM + tid].x; t.y += stat_lds[w * TM + tid].y; }
      stats_part[(size_t)(m0 + tid) * tiles_total + tile] = t;
    }
  }
}

// f16x2 operand scale: amax_seg = 0 -> one scale for the whole tensor (x_absmax[0]); amax_seg = 256 -> x_absmax is an "amax buffer"
// (include/pvcnn_hip.h) with one maximum per 256-point tile behind the global one: every workgroup scales by ITS tile's maximum.
// (A 256-channel tile, MB = 8 with the accumulators in AGPRs, was built and measured in round 3: 0.196 vs 0.154 ms at 128 -> 1024,
// 1730 vs 1760 clouds/s in the step -- removed.)
// VEC (N % 4 == 0, N >= 4, x 16-byte aligned): the chunk loop is STRAIGHT-LINE code -- every row load is issued unconditionally from a
// clamped address (rows >= K read row K - 1: their weights in the image are zero; the points of a ragged last tile read the last
// quad: those columns are never stored), the next weight fragments from a clamped chunk.  With branches around the loads the
// compiler cannot count the loads in flight and waits with vmcnt(0) in front of the first MFMA of every chunk -- i.e. for the
// prefetch it has just issued (ISA of round 3: the whole HBM latency was exposed once per chunk).
template <int NS, int MB, int PF = 1, bool VEC = false>
__global__ __launch_bounds__(256, 2) void pw_gemm_bf16_kernel(const float *__restrict__ x, const uint16_t *__restrict__ wts,
                                                              const float *__restrict__ bias, float *__restrict__ y, int K, int M,
                                                              int N, int tiles_n, int tiles_total, float2 *__restrict__ stats_part,
                                                              const uint32_t *__restrict__ x_absmax, const int *__restrict__ wexp,
                                                              int amax_seg) {
  // Wave arrangement inside the 32*MB x 256 workgroup tile.  Each lane fetches its own A (weight) fragments from global memory:
  // with the four waves side by side along the points (WM = 1) every wave pulls ALL 32*MB rows through the CU's vector-memory
  // path -- at MB = 4 that is 48 KiB per chunk and workgroup (A four times + the x rows) = as many L1 cycles (64 B/clk) as the
  // chunk has MFMA cycles: the kernel sat at 0.23 of the MFMA peak whatever was prefetched.  WM = 2 arranges the waves 2 x 2: a
  // wave owns 32*MB/2 rows x 128 points, the A traffic halves (32 KiB: 2/3 of the MFMA time) and the B fragments -- LDS reads,
  // which had headroom -- double.  Same products in the same order per output element: results are bit-identical.
  constexpr int TM = 32 * MB, WM = (MB >= 4) ? 2 : 1, MBW = MB / WM, NBW = 2 * WM;
  constexpr int WBLK = NS * TM * kPbK;                          // bf16 elements of one (chunk, mtile) weight block
  __shared__ __attribute__((aligned(16))) uint32_t xs[NS * 8 * kPbN];       // [NS][8 channel pairs][256 points] words

  // XCD-aware tile order: workgroup ids go round-robin over the 8 XCDs (each with its own L2).  The mtiles workgroups that
  // share one 256-point tile of x get ids that are congruent mod 8 and adjacent in an XCD's queue, so x is fetched from HBM
  // once and re-read from that XCD's L2 (otherwise the big GEMM moves x mtiles = 4 times: 1.5 GB, and is bandwidth-bound).
  const int mtiles = ceil_div(M, TM);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile = (slot / mtiles) * 8 + xcd, mt = slot - (slot / mtiles) * mtiles;
  if (tile >= tiles_total) return;
  const int x_shift = NS == 2 ? scale_shift(amax_seg > 0 ? x_absmax[1 + tile] : *x_absmax) : 0;    // tile = b * tiles_n + point tile
  const float x_scale = exp2_int(x_shift);
  const int b = tile / tiles_n, n0 = (tile - b * tiles_n) * kPbN, m0 = mt * TM;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int wm = wave % WM, wn = wave / WM;                     // this wave's row group / point group
  const float *xb = x + (size_t)b * K * N;
  const int chunks = ceil_div(K, kPbK);

  int a_off[MBW];                                               // A fragment uint4 offsets inside one plane slab
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) {
    const int row = (wm * MBW + mb) * 32 + j;
    a_off[mb] = (row * 8 + ((kh ^ ((row >> 3) & 1)) * 4)) >> 2;
  }
  int b_pt[NBW];                                                // this lane's point inside the tile, per column block
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) b_pt[nb] = wn * (32 * NBW) + nb * 32 + j;
  f32x16 acc[MBW][NBW];
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

  // Staging: an item = (channel pair kp, point quad q): two fully coalesced float4 loads (4 consecutive points of channels
  // 2kp and 2kp+1: a wave reads one whole 1 KiB channel row per instruction), split into NS bf16 pieces, packed pairwise and
  // stored with ONE 16-byte LDS write per plane into xs[plane][kp][point] (points contiguous: conflict-free).  A lane's B
  // fragment (8 consecutive channels of its point) is then 4 ds_read_b32 at stride 256 words (bank = point: conflict-free).
  // (First version: scalar loads with (channel pair, point) items -- 16 cache lines per wave instruction; the kernel spent its
  // time in the texture addresser and was no faster than fp32 MFMA.)
  // Pipeline: the next chunk's rows are requested right after the barrier that publishes this chunk's tile (they land during
  // its MFMAs); the next chunk's weight fragments as soon as this chunk's MFMAs have been issued.
  constexpr int ITEMS = (kPbK / 2) * (kPbN / 4) / 256;          // 2 items per thread and chunk
  // PF = how many chunks ahead the rows are requested (a register ring of PF stages; the chunk loop is unrolled PF times so
  // that the ring indices are compile-time).  With PF = 1 every workgroup on the chip alternates "burst of loads" / "multiply":
  // the loads of chunk i + 1 have only chunk i's MFMAs (~0.3 us) to land in.
  float4 va_[PF][ITEMS], vb_[PF][ITEMS];
  const bool vec = !VEC && (N % 4 == 0) && aligned16(xb) && (n0 + kPbN <= N);
  auto load_x = [&](int chunk, float4 (&va)[ITEMS], float4 (&vb)[ITEMS]) {
    const int c0 = chunk * kPbK;
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
      const int e = u * 256 + tid;
      const int q = e & 63, kp = e >> 6;                        // 64 point quads x 8 channel pairs
      const int c = c0 + 2 * kp, n = n0 + 4 * q;
      if constexpr (VEC) {
        const float *col = xb + min(n, N - 4);
        va[u] = *reinterpret_cast<const float4 *>(col + (size_t)min(c, K - 1) * N);
        vb[u] = *reinterpret_cast<const float4 *>(col + (size_t)min(c + 1, K - 1) * N);
        continue;
      }
      va[u] = vb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (vec) {
        if (c < K) va[u] = *reinterpret_cast<const float4 *>(xb + (size_t)c * N + n);
        if (c + 1 < K) vb[u] = *reinterpret_cast<const float4 *>(xb + (size_t)(c + 1) * N + n);
      } else {
        float a[4] = {0.f, 0.f, 0.f, 0.f}, bq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (n + t < N) {
            if (c < K) a[t] = xb[(size_t)c * N + n + t];
            if (c + 1 < K) bq[t] = xb[(size_t)(c + 1) * N + n + t];
          }
        va[u] = make_float4(a[0], a[1], a[2], a[3]);
        vb[u] = make_float4(bq[0], bq[1], bq[2], bq[3]);
      }
    }
  };
  uint4 af[MBW][NS];
  auto load_a = [&](int chunk) {
    const uint4 *wq = reinterpret_cast<const uint4 *>(wts + ((size_t)chunk * mtiles + mt) * WBLK);
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int s = 0; s < NS; ++s) af[mb][s] = wq[s * (TM * kPbK / 8) + a_off[mb]];
  };
#pragma unroll
  for (int d = 0; d < PF; ++d) load_x(d, va_[d], vb_[d]);       // (a chunk beyond K loads zeros: c >= K)
  load_a(0);
  // the chunk count is padded to a multiple of PF: a padding chunk stages zeros and multiplies them with the previous chunk's
  // (finite) weight fragments -- nothing is added, and the unrolled body stays straight-line code
  for (int chunk0 = 0; chunk0 < chunks; chunk0 += PF) {
#pragma unroll
  for (int d = 0; d < PF; ++d) {
    const int chunk = chunk0 + d;
    float4 (&va)[ITEMS] = va_[d];
    float4 (&vb)[ITEMS] = vb_[d];
    __syncthreads();                                            // previous chunk's fragment reads are done
    // VEC: a padding chunk (chunk >= chunks) holds clamped rows, not zeros, and meets the previous chunk's weights: scaled to zero
    const float cs = (VEC && chunk >= chunks) ? 0.0f : x_scale;
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
      const int e = u * 256 + tid;
      const int q = e & 63, kp = e >> 6;
      const float a[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, bq[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
      uint32_t w[NS][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        uint32_t pw[NS];
        if constexpr (NS == 2 || VEC) split_pair<NS>(a[t] * cs, bq[t] * cs, pw);
        else split_pair<NS>(a[t], bq[t], pw);
#pragma unroll
        for (int s = 0; s < NS; ++s) w[s][t] = pw[s];
      }
#pragma unroll
      for (int s = 0; s < NS; ++s)
        *reinterpret_cast<uint4 *>(xs + (s * 8 + kp) * kPbN + 4 * q) = make_uint4(w[s][0], w[s][1], w[s][2], w[s][3]);
    }
    __syncthreads();
    load_x(chunk + PF, va, vb);                                 // in flight during the next PF chunks' MFMAs
    uint4 bf[NBW][NS];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const uint32_t *col = xs + (s * 8 + 4 * kh) * kPbN + b_pt[nb];
        bf[nb][s] = make_uint4(col[0], col[kPbN], col[2 * kPbN], col[3 * kPbN]);
      }
    __builtin_amdgcn_sched_barrier(0);
#define PVCNN_PB_MFMA(SA, SB)                                                                                            \
    _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb)                                                                   \
    _Pragma("unroll") for (int mb = 0; mb < MBW; ++mb)                                                                   \
      acc[mb][nb] = mfma16<NS>(af[mb][SA], bf[nb][SB], acc[mb][nb])
    if constexpr (NS == 1) {
      PVCNN_PB_MFMA(0, 0);
    } else if constexpr (NS == 2) {
      PVCNN_PB_MFMA(1, 0); PVCNN_PB_MFMA(0, 1);                         // lo x hi, hi x lo, then hi x hi
      PVCNN_PB_MFMA(0, 0);
    } else {
      PVCNN_PB_MFMA(2, 0); PVCNN_PB_MFMA(1, 1); PVCNN_PB_MFMA(0, 2);    // smallest partial products first
      PVCNN_PB_MFMA(1, 0); PVCNN_PB_MFMA(0, 1);
      PVCNN_PB_MFMA(0, 0);
    }
#undef PVCNN_PB_MFMA
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (VEC) load_a(min(chunk + 1, chunks - 1));      // overwrites af once this chunk's MFMAs have been issued
    else if (chunk + 1 < chunks) load_a(chunk + 1);
  }
  }

  pb_epilogue<NS, MB, WM>(acc, xs, bias, y, M, N, b, n0, m0, tile, tiles_total, stats_part, wexp, x_shift);
}

// The wide f16x2 tile (NS = 2, MB = 4, vector loads) with the conversion INSIDE the multiply phase: the staging buffer is double-
// buffered (2 x 16 KiB) and a chunk costs ONE barrier.  In iteration c a wave reads its B fragments of tile c, then issues the 24
// MFMAs of chunk c with the fp32 -> fp16 hi / lo conversion of the rows of chunk c + 1 (in registers since the previous iteration)
// scheduled BETWEEN them (two vector-ALU instructions per MFMA: the matrix pipe is busy 32 cycles per instruction, the conversions
// are free), stores the converted tile into the other buffer, requests the rows of chunk c + 3 and meets the other waves.  Every
// load has a whole chunk to land: the weight fragments of chunk c + 1 are requested in front of chunk c's MFMAs (second register set),
// and the loop is straight-line code, so the compiler's vmcnt waits leave the younger loads in flight.  (pw_gemm_bf16_kernel converts between two barriers and relies on the co-resident workgroup to fill the
// matrix pipe meanwhile.)  Same products in the same order per output element: bit-identical to pw_gemm_bf16_kernel<2, 4>.
template <int NS>   // 2 = f16x2; 1 = plain bf16 operands (torch.autocast): one plane, one product -- the conversion is then the longer phase
__global__ __launch_bounds__(256, 2) void pw_gemm_f16_pipe_kernel(const float *__restrict__ x, const uint16_t *__restrict__ wts,
                                                                  const float *__restrict__ bias, float *__restrict__ y, int K, int M,
                                                                  int N, int tiles_n, int tiles_total, float2 *__restrict__ stats_part,
                                                                  const uint32_t *__restrict__ x_absmax, const int *__restrict__ wexp,
                                                                  int amax_seg) {
  static_assert(NS == 1 || NS == 2, "f16x2 or bf16");
  constexpr int MB = 4, TM = 32 * MB, WM = 2, MBW = MB / WM, NBW = 2 * WM, WBLK = NS * TM * kPbK;
  constexpr int ITEMS = (kPbK / 2) * (kPbN / 4) / 256, TILE = NS * 8 * kPbN;
  // two tiles [plane][kh][128-point block][h][128 points][2 words], channel pair = 4 kh + 2 h + word: a staging thread owns 4 channels
  // x 4 points = two 16-byte stores of 8 consecutive words per plane; a lane's B fragment (the 8 channels 8 kh .. 8 kh + 7 of its
  // point) = its h = 0 and h = 1 word pairs, 1 KiB apart = ONE ds_read2_b64 that lands in the four registers of the MFMA operand
  __shared__ __attribute__((aligned(16))) uint32_t xs[2 * TILE];
  const int mtiles = ceil_div(M, TM);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;                    // XCD-aware tile order, as in pw_gemm_bf16_kernel
  const int tile = (slot / mtiles) * 8 + xcd, mt = slot - (slot / mtiles) * mtiles;
  if (tile >= tiles_total) return;
  const int x_shift = NS == 2 ? scale_shift(amax_seg > 0 ? x_absmax[1 + tile] : *x_absmax) : 0;
  const float x_scale = exp2_int(x_shift);
  const int b = tile / tiles_n, n0 = (tile - b * tiles_n) * kPbN, m0 = mt * TM;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int wm = wave % WM, wn = wave / WM;
  const float *xb = x + (size_t)b * K * N;
  const int chunks = ceil_div(K, kPbK);

  int a_off[MBW];
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) {
    const int row = (wm * MBW + mb) * 32 + j;
    a_off[mb] = (row * 8 + ((kh ^ ((row >> 3) & 1)) * 4)) >> 2;
  }
  f32x16 acc[MBW][NBW];
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

  // a thread's items: channel pairs 2 * wave and 2 * wave + 1 of point quad `lane`, the same positions in every chunk (a wave
  // instruction reads one whole 1 KiB channel row)
  static_assert(ITEMS == 2, "two channel pairs per thread");
  const float *col = xb + min(n0 + 4 * lane, N - 4);
  const int kp0 = 2 * wave, st = ((((wave >> 1) * 2 + (lane >> 5)) * 2 + (wave & 1)) * 128 + ((4 * lane) & 127)) * 2;
  auto load_x = [&](int chunk, float4 (&va)[ITEMS], float4 (&vb)[ITEMS]) {   // rows >= K: row K - 1 (zero weights / zero scale)
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
      const int c = chunk * kPbK + 2 * (kp0 + u);
      va[u] = *reinterpret_cast<const float4 *>(col + (size_t)min(c, K - 1) * N);
      vb[u] = *reinterpret_cast<const float4 *>(col + (size_t)min(c + 1, K - 1) * N);
    }
  };
  uint4 af_[2][MBW][NS];                                        // two sets: the next chunk's fragments are requested a whole chunk ahead
  auto load_a = [&](int chunk, uint4 (&af)[MBW][NS]) {
    const uint4 *wq = reinterpret_cast<const uint4 *>(wts + ((size_t)chunk * mtiles + mt) * WBLK);
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
      for (int s = 0; s < NS; ++s) af[mb][s] = wq[s * (TM * kPbK / 8) + a_off[mb]];
  };
  // w[s][h] = words (point 2h, pair 0), (point 2h, pair 1), (point 2h + 1, pair 0), (point 2h + 1, pair 1) of plane s
  auto convert = [&](const float4 (&va)[ITEMS], const float4 (&vb)[ITEMS], float cs, uint4 (&w)[NS][2]) {
    uint32_t t4[NS][ITEMS][4];
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
      const float a[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, bq[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        uint32_t pw[NS];
        split_pair<NS>(a[t] * cs, bq[t] * cs, pw);
#pragma unroll
        for (int s = 0; s < NS; ++s) t4[s][u][t] = pw[s];
      }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) w[s][h] = make_uint4(t4[s][0][2 * h], t4[s][1][2 * h], t4[s][0][2 * h + 1], t4[s][1][2 * h + 1]);
  };
  auto store = [&](int buf, const uint4 (&w)[NS][2]) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) *reinterpret_cast<uint4 *>(xs + buf * TILE + s * 8 * kPbN + st + 4 * h) = w[s][h];
  };

  float4 va_[2][ITEMS], vb_[2][ITEMS];
  uint4 w[NS][2];
  load_x(0, va_[0], vb_[0]);
  load_x(1, va_[1], vb_[1]);
  load_a(0, af_[0]);
  convert(va_[0], vb_[0], x_scale, w);
  store(0, w);
  load_x(2, va_[0], vb_[0]);
  __syncthreads();
  // the chunk count is padded to even (ring / buffer indices are compile-time): the padding chunk's tile was converted with scale 0
  for (int chunk0 = 0; chunk0 < chunks; chunk0 += 2) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int chunk = chunk0 + d;
      // tile `chunk` is published in buffer d; ring slot d ^ 1 holds the rows of chunk + 1 (landed), slot d those of chunk + 2
      uint4 bf[NBW][NS];
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const uint32_t *c2 = xs + d * TILE + ((((s * 2 + kh) * 2 + wn) * 2) * 128 + nb * 32 + j) * 2;
          const uint2 lo = *reinterpret_cast<const uint2 *>(c2), hi = *reinterpret_cast<const uint2 *>(c2 + 256);
          bf[nb][s] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      load_a(min(chunk + 1, chunks - 1), af_[d ^ 1]);
      uint4 (&af)[MBW][NS] = af_[d];
      __builtin_amdgcn_sched_barrier(0);
      convert(va_[d ^ 1], vb_[d ^ 1], chunk + 1 < chunks ? x_scale : 0.0f, w);
#define PVCNN_PB_MFMA(SA, SB)                                                                                            \
      _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb)                                                                 \
      _Pragma("unroll") for (int mb = 0; mb < MBW; ++mb)                                                                 \
        acc[mb][nb] = mfma16<NS>(af[mb][SA], bf[nb][SB], acc[mb][nb])
      if constexpr (NS == 2) {
        PVCNN_PB_MFMA(1, 0); PVCNN_PB_MFMA(0, 1);                       // lo x hi, hi x lo, then hi x hi
        PVCNN_PB_MFMA(0, 0);
      } else {
        PVCNN_PB_MFMA(0, 0);
      }
#undef PVCNN_PB_MFMA
#pragma unroll
      for (int i = 0; i < (NS == 2 ? 3 : 1) * MBW * NBW; ++i) {         // 1 MFMA, 2 (bf16: 4) vector-ALU; 24 (8) times
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NS == 2 ? 2 : 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      store(d ^ 1, w);
      load_x(chunk + 3, va_[d ^ 1], vb_[d ^ 1]);
      __syncthreads();
    }
  }
  pb_epilogue<NS, MB, WM>(acc, xs, bias, y, M, N, b, n0, m0, tile, tiles_total, stats_part, wexp, x_shift);
}

// ---- the 256 x 256 f16x2 tile, ONE workgroup per CU, persistent (round 6) ---------------------------------------------------------
// pw_gemm_f16_pipe_kernel owns 128 output channels x 256 points and meets at a barrier every 24 MFMAs per wave; by the counters its
// matrix pipes are busy in 0.40 of the cycles (issue-stalled 0.52): every chunk starts with the LDS reads of its B fragments exposed,
// the conversion of a staged element is amortised over 128 output channels only (4 m-tiles re-convert x at M = 512), and the epilogue
// of a workgroup overlaps nothing but the other workgroup of its CU.  Here:
//   * a workgroup owns 256 output channels (two 128-row blocks of the SAME weight image) x 256 points: the four waves sit 2 x 2, each
//     with 128 channels x 128 points = 16 accumulator tiles (256 registers: the kernel takes the whole register file of its SIMD,
//     one wave per SIMD), 48 MFMAs per 16-channel step and wave against 8 A and 8 B fragment loads and ~70 conversion instructions;
//   * everything a step needs was requested earlier: the fragments of step s + 1 (A from the image, B from LDS: the staged tiles live
//     in a ring of THREE buffers, a tile is published two steps before it is multiplied) during the first 16 MFMAs of step s into a
//     second register set, the fp32 rows of step s + 4 when the registers of step s + 2's rows have been converted; the conversion
//     of step s + 2 sits between the MFMAs of step s; ONE LDS-only barrier per step;
//   * NO memory instruction stands between two groups of MFMAs: a step is ONE basic block -- the request streams are branch-free (a
//     stream is at most one item ahead of the multiply: its scalar offset is a select between this item's and the next item's base) --
//     of 48 PINNED slots, each one MFMA + at most one request + a few vector-ALU instructions (see `group`).  What that bought, by
//     the ablation builds of tools/probe (1472 -> 512 over 65 536 points, normal fill, us per launch): requests in blocks between
//     the MFMA groups 350 (first version; the 128-row kernel on the same box: 396) -> scheduler-placed 347 -> pinned slots 334 ->
//     conversion packed along the points 330; of those 330, the conversion costs ~70 (leaving it out: 227), the row requests ~30,
//     the step barrier ~25, and the MFMAs alone (no request, no conversion, no barrier) take 204: the matrix pipes at the clock the
//     chip sustains under a dense fp16 MFMA stream on random operands;
//   * the workgroup is PERSISTENT: it walks its (point tile, channel block) items in the XCD-aware order of the kernels above (the
//     M / 256 workgroups that stream the same 256 points run side by side behind one L2); when an item's last MFMA has been issued
//     the first two tiles of the next one are already published and its first fragments are in registers.  An item's first MFMAs
//     start from C = 0 (the accumulators are not carried across items), the epilogue's stores drain behind the next item's MFMAs.
// Needs K % 64 == 0 (the step loop is unrolled four times: register-ring indices are compile-time, an item is a whole number of
// groups), an even number of 128-row blocks in the image, N % 256 == 0 and tensors below 4 GiB (no ragged point tile: loads and
// stores go through wave-uniform buffer descriptors with 32-bit offsets; the bounds check drops the padded rows of the last channel
// block).  Per output element the same products in the same order as pw_gemm_f16_pipe_kernel: bit-identical results, BatchNorm partial
// sums included (tests/test_gpu_pw_wide.py).  PVCNN_PW_WIDE=0 keeps the 128-row kernel.
constexpr int kWideBufs = 3;
// WMW = waves along the output channels.  2: the 256 x 256 item above.  4 (an image with a multiple of FOUR 128-row blocks: M = 512,
// 1024, 1472): an item is 512 output channels x 128 points -- the four waves own a 128-row block each and share the B fragments --,
// so that a staged and converted element feeds 512 channels: half the conversion, half the row requests per MFMA (the ablation builds
// price the conversion at a quarter of the 256 x 256 kernel), and a layer with M = 512 streams x ONCE, without counting on a second
// workgroup next to it.  The two 128-point halves of a 256-point tile are consecutive items of one workgroup: they take the tile's
// scale, and their BatchNorm partial sums meet in LDS in the order of the 128-row kernel's two point groups (same bits).
template <int WMW> struct WideGeom {
  static constexpr int WNW = 4 / WMW, TP = 128 * WNW, ROWS = 128 * WMW;            // point groups; points / rows of an item
  static constexpr int TILE = 2 * 2 * TP * 4;                                      // words of one staged 16-channel tile (two planes)
  static constexpr size_t LDS = (size_t)kWideBufs * TILE * sizeof(uint32_t) + (size_t)(2 + 2) * ROWS * sizeof(float2);
};

// AB (ablation bits, tools/probe builds only; the library instantiates 0): 1 = no A requests in the step loop, 2 = no row requests,
// 4 = no conversion / tile store, 8 = no B reads, 16 = no step barrier -- the step then multiplies stale fragments: wrong results, the
// MFMAs and everything else stay, and the difference in time is what the removed part costs (phase clocks perturb too much here).
template <int WMW, int AB = 0>
__global__ __launch_bounds__(256, 1) void pw_gemm_f16_wide_kernel(const float *__restrict__ x, const uint16_t *__restrict__ wts,
                                                                  const float *__restrict__ bias, float *__restrict__ y, int K, int M,
                                                                  int N, int tiles_n, int tiles_total, float2 *__restrict__ stats_part,
                                                                  const uint32_t *__restrict__ x_absmax, const int *__restrict__ wexp,
                                                                  int amax_seg, unsigned x_bytes, unsigned w_bytes) {
  static_assert(WMW == 2 || WMW == 4, "2 x 2 or 4 x 1 waves");
  using G = WideGeom<WMW>;
  constexpr int NS = 2, MBW = 4, NBW = 4, TMI = 128, WBLK = NS * TMI * kPbK, TILE = G::TILE, TP = G::TP, ROWS = G::ROWS, XR = 2;
  constexpr int HALVES = WMW == 4 ? 2 : 1;                      // 128-point halves of a 256-point tile = consecutive items
  constexpr int NROW = WMW == 4 ? 2 : 4, NPAIR = NROW / 2;      // rows / channel pairs a thread stages per step
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) uint32_t wide_lds[];
  // staged tiles: [buffer][plane][kh][TP points][4 words]; word w of (kh, point) = channel pair 4 kh + w: a lane's B fragment (the 8
  // channels 8 kh .. 8 kh + 7 of its point) is ONE 16-byte read, 32 consecutive points = 512 contiguous bytes (conflict-free)
  uint32_t *xs = wide_lds;
  float2 *stat_lds = reinterpret_cast<float2 *>(wide_lds + kWideBufs * TILE);     // [2 point groups / halves][ROWS]
  float2 *row_lds = stat_lds + 2 * ROWS;                        // [item parity][ROWS] (bias, 2^-wexp) of the item's rows
  const int mtiles = ceil_div(M, TMI), mgroups = mtiles / WMW;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int items_local = ((tiles_total + 7) >> 3) * mgroups * HALVES;   // items of this XCD: (its point tiles) x (row groups) x halves
  if (slot * HALVES >= items_local) return;
  // a workgroup takes HALVES consecutive items per turn (the two halves of one tile), its turns are nslots apart
  const int turns = (items_local / HALVES - slot + nslots - 1) / nslots, rounds = turns * HALVES, chunks = K / kPbK;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = WMW == 4 ? wave : (wave & 1), wn = WMW == 4 ? 0 : (wave >> 1);
  const uint32_t row_bytes = (uint32_t)N * 4u;

  // ---- the items of this workgroup and what the request streams need of an item: three scalars ----
  struct Item { int tile, mg, half; uint32_t x_base, a_base; float scale; };
  auto item_at = [&](int r) {                                   // (clamped: past the last item the streams re-read it; nothing of that is used)
    Item it;
    const int rc = min(r, rounds - 1), turn = rc / HALVES;
    it.half = rc - turn * HALVES;
    const int jdx = slot + turn * nslots, tl = jdx / mgroups;
    it.mg = jdx - tl * mgroups;
    it.tile = tl * 8 + xcd;                                     // may be >= tiles_total in the padded tail: never stored
    const int tc = min(it.tile, tiles_total - 1), b = tc / tiles_n, n0 = (tc - b * tiles_n) * kPbN + it.half * 128;
    it.x_base = (uint32_t)b * (uint32_t)K * row_bytes + (uint32_t)n0 * 4u;              // bytes from x to (cloud, row 0, point n0)
    it.a_base = (uint32_t)(WMW * it.mg + wm) * (uint32_t)(WBLK * 2);                    // bytes from the image to this wave's row block, chunk 0
    it.scale = exp2_int(scale_shift(amax_seg > 0 ? x_absmax[1 + tc] : *x_absmax));     // (the 256-point tile's, also for a half)
    // (wave-uniform, all of it: say so -- a scale that the compiler keeps in a vector register is one more register across the step loop)
    it.x_base = __builtin_amdgcn_readfirstlane(it.x_base);
    it.a_base = __builtin_amdgcn_readfirstlane(it.a_base);
    it.scale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(it.scale)));
    return it;
  };
  auto descriptor = [](const void *base, uint32_t bytes) {
    const uintptr_t p = reinterpret_cast<uintptr_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t xrsrc = descriptor(x, x_bytes), wrsrc = descriptor(wts, w_bytes);
  const uint32_t a_chunk = (uint32_t)mtiles * (uint32_t)(WBLK * 2), x_chunk = (uint32_t)kPbK * row_bytes;   // bytes per step

  // per-lane offsets (bytes), the same in every step
  // A fragment: row mb * 32 + j of the 128-row block (the swizzle bit is bit 3 of the row, which a multiple of 32 does not touch)
  const uint32_t a_off = (uint32_t)(j * 8 + ((kh ^ ((j >> 3) & 1)) * 4)) * 4u;
  // staging item of a thread.  WMW = 2: channel pairs 2 wave, 2 wave + 1 (words 2 (wave & 1) + {0, 1} of half kh' = wave >> 1) of
  // point quad `lane` (64 quads).  WMW = 4: channel pair p = lane & 7 (word p & 3 of half p >> 2) of point quad 8 wave + (lane >> 3)
  // (32 quads): a wave instruction reads 128-byte runs of 8 rows
  const int sp = lane & 7, sq = WMW == 4 ? 8 * wave + (lane >> 3) : lane;
  const uint32_t x_off = WMW == 4 ? (uint32_t)sq * 16u + (uint32_t)(2 * sp) * row_bytes : (uint32_t)lane * 16u + (uint32_t)(4 * wave) * row_bytes;
  const uint32_t st_off = WMW == 4 ? (uint32_t)((((sp >> 2) * TP + 4 * sq) * 4 + (sp & 3)) * 4)
                                   : (uint32_t)((((wave >> 1) * TP + 4 * lane) * 4 + 2 * (wave & 1)) * 4);
  const uint32_t b_off = (uint32_t)((kh * TP + wn * 128 + j) * 16);
  unsigned char *xs8 = reinterpret_cast<unsigned char *>(xs);

  // conversion of channel pair u (rows 2u, 2u + 1 of the thread's), points 2 h2 and 2 h2 + 1, in two halves that are issued
  // between different MFMAs: (1) scale + round to the hi fp16 pairs, (2) the residuals' fp16 pairs.  split_pair's arithmetic with the
  // packed vector ALU running ALONG THE POINTS (two neighbouring points of one row sit in neighbouring registers of the 16-byte load;
  // the two rows of a pair do not): packed multiplies + packs, then conversions back + packed fused multiply-subtracts
  // (a * scale - hi in ONE rounding = the exact residual, like the product minus hi) + packs
  f16x2 th[NPAIR][4];                                           // hi pairs, kept from half 1 to half 2
  uint32_t tw[NS][NPAIR][4];                                    // [plane][pair u][point q]: the converted tile of this thread
  auto conv_hi = [&](const float4 (&v)[NROW], float scale, int u, int h2) {
    const f32x2 a = h2 == 0 ? f32x2{v[2 * u].x, v[2 * u].y} : f32x2{v[2 * u].z, v[2 * u].w};
    const f32x2 c = h2 == 0 ? f32x2{v[2 * u + 1].x, v[2 * u + 1].y} : f32x2{v[2 * u + 1].z, v[2 * u + 1].w};
    const f32x2 sa = a * scale, sc = c * scale;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      th[u][2 * h2 + e] = __builtin_convertvector(f32x2{sa[e], sc[e]}, f16x2);
      tw[0][u][2 * h2 + e] = __builtin_bit_cast(uint32_t, th[u][2 * h2 + e]);
    }
  };
  auto conv_lo = [&](const float4 (&v)[NROW], float scale, int u, int h2) {
    const f32x2 a = h2 == 0 ? f32x2{v[2 * u].x, v[2 * u].y} : f32x2{v[2 * u].z, v[2 * u].w};
    const f32x2 c = h2 == 0 ? f32x2{v[2 * u + 1].x, v[2 * u + 1].y} : f32x2{v[2 * u + 1].z, v[2 * u + 1].w};
    const f32x2 ha = {(float)th[u][2 * h2][0], (float)th[u][2 * h2 + 1][0]}, hc = {(float)th[u][2 * h2][1], (float)th[u][2 * h2 + 1][1]};
    const f32x2 sv = {scale, scale};
    const f32x2 ra = __builtin_elementwise_fma(a, sv, -ha), rc = __builtin_elementwise_fma(c, sv, -hc);     // exact (see split_pair)
#pragma unroll
    for (int e = 0; e < 2; ++e) tw[1][u][2 * h2 + e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{ra[e], rc[e]}, f16x2));
  };
  auto store_pt = [&](int buf, int s2, int q) {                 // the thread's word(s) of (plane s2, point q): 8 / 4 bytes
    unsigned char *dst = xs8 + (buf * TILE + s2 * 2 * TP * 4) * 4 + st_off + q * 16;
    if constexpr (WMW == 4) *reinterpret_cast<uint32_t *>(dst) = tw[s2][0][q];
    else *reinterpret_cast<u32x2 *>(dst) = u32x2{tw[s2][0][q], tw[s2][NPAIR - 1][q]};
  };
  // plane 0 = hi, plane 1 = lo (split_pair)
  auto load_a1 = [&](uint32_t soff, int plane, int mb) {
    return __builtin_amdgcn_raw_buffer_load_b128(wrsrc, a_off + (uint32_t)(plane * (TMI * kPbK * 2) + mb * 1024), soff, 0);
  };
  auto load_b1 = [&](int buf, int plane, int nb) {
    return *reinterpret_cast<const u32x4 *>(xs8 + (buf * TILE + plane * 2 * TP * 4) * 4 + b_off + nb * 512);
  };
  auto load_x1 = [&](uint32_t soff, int k) {
    const u32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, x_off + (uint32_t)k * row_bytes, soff, 0);
    return make_float4(__uint_as_float(r4.x), __uint_as_float(r4.y), __uint_as_float(r4.z), __uint_as_float(r4.w));
  };
  auto mma = [](const u32x4 &a, const u32x4 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  };

  // ---- prologue (item 0): tiles 0 and 1 published, rows of tiles 2 and 3 in flight, fragments of step 0 in registers ----
  PVCNN_PROBE_BEGIN();
  Item cur = item_at(0), nxt = item_at(1);
  float4 xv[XR][NROW];
  u32x4 a_hi[2][MBW], a_lo[2][MBW], b_hi[2][NBW], b_lo[2][NBW];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int k = 0; k < NROW; ++k) xv[t][k] = load_x1(cur.x_base + (uint32_t)t * x_chunk, k);
  }
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) { a_hi[0][mb] = load_a1(cur.a_base, 0, mb); a_lo[0][mb] = load_a1(cur.a_base, 1, mb); }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int u = 0; u < NPAIR; ++u)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) { conv_hi(xv[t], cur.scale, u, h2); conv_lo(xv[t], cur.scale, u, h2); }
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
      for (int q = 0; q < 4; ++q) store_pt(t, s2, q);
#pragma unroll
    for (int k = 0; k < NROW; ++k) xv[t][k] = load_x1(cur.x_base + (uint32_t)(2 + t) * x_chunk, k);
  }
  lds_barrier();
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) { b_hi[0][nb] = load_b1(0, 0, nb); b_lo[0][nb] = load_b1(0, 1, nb); }
  PVCNN_PROBE(6);                                               // slot 6: prologue
  int b_cur = 0;                                                // ring position of the tile being multiplied (step % 3)

  f32x16 acc[MBW][NBW];                                         // (every element is defined by the first group of an item)
  // One group of four steps, compute chunks c0 .. c0 + 3 of item `cur`.  FIRST: the item's first group -- its very first MFMAs start
  // from C = 0 (an inline constant), so the accumulators are not carried from item to item (no zeroing pass, no register shuffle at
  // the loop boundary: seen in the ISA of the first version).
  // A step is 48 SLOTS, each one MFMA followed by at most one request and a few vector-ALU instructions, PINNED in this order
  // (sched_barrier after every slot): left to itself -- also under sched_group_barrier -- the scheduler bunches the MFMAs (12 back
  // to back, then 40 instructions with the matrix pipe idle: the second version of this kernel, phase clocks 2300 cycles per step
  // for 1536 of MFMA).  In-order issue lets ~7 other instructions go between two MFMAs for free; no slot has more.
  //     slots  0 .. 15  lo x hi   + the fragments of step s + 1: B hi, B lo (LDS), A hi, A lo (image) -- second register sets
  //     slots 16 .. 31  hi x lo   + the conversion of the rows of step s + 2 (8 / 4 half-conversions)
  //     slots 32 .. 47  hi x hi   + the converted tile's 8 stores, the row requests of step s + 2 + XR
  auto group = [&](auto first_tag, int c0) {
    constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int c = c0 + d, cs = d & 1, ns = cs ^ 1;
      const int b_nxt = b_cur == 2 ? 0 : b_cur + 1, b_st = b_nxt == 2 ? 0 : b_nxt + 1;
      // the three request streams: chunk c + 1 (fragments), c + 2 (conversion), c + 2 + XR (rows) -- of this item or of the next
      const bool na = c + 1 >= chunks, nv = c + 2 >= chunks, nx = c + 2 + XR >= chunks;
      const uint32_t a_soff = (na ? nxt.a_base : cur.a_base) + (uint32_t)(c + 1 - (na ? chunks : 0)) * a_chunk;
      const uint32_t x_soff = (nx ? nxt.x_base : cur.x_base) + (uint32_t)(c + 2 + XR - (nx ? chunks : 0)) * x_chunk;
      const float vscale = nv ? nxt.scale : cur.scale;
      float4 (&v)[NROW] = xv[d % XR];                           // rows of step s + 2
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {                            // ---- lo x hi
        const int nb = i >> 2, mb = i & 3;
        if constexpr (FIRST) {
          if (d == 0) acc[mb][nb] = mma(a_lo[cs][mb], b_hi[cs][nb], f32x16{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f});
          else acc[mb][nb] = mma(a_lo[cs][mb], b_hi[cs][nb], acc[mb][nb]);
        } else {
          acc[mb][nb] = mma(a_lo[cs][mb], b_hi[cs][nb], acc[mb][nb]);
        }
        if (i < 4) { if constexpr (!(AB & 8)) b_hi[ns][i] = load_b1(b_nxt, 0, i); else b_hi[ns][i] = b_hi[cs][i]; }     // (the tile of step s + 1 was published at the last barrier)
        else if (i < 8) { if constexpr (!(AB & 1)) a_lo[ns][i - 4] = load_a1(a_soff, 1, i - 4); else a_lo[ns][i - 4] = a_lo[cs][i - 4]; }
        else if (i < 12) { if constexpr (!(AB & 1)) a_hi[ns][i - 8] = load_a1(a_soff, 0, i - 8); else a_hi[ns][i - 8] = a_hi[cs][i - 8]; }
        else { if constexpr (!(AB & 8)) b_lo[ns][i - 12] = load_b1(b_nxt, 1, i - 12); else b_lo[ns][i - 12] = b_lo[cs][i - 12]; }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {                            // ---- hi x lo
        const int nb = i >> 2, mb = i & 3;
        acc[mb][nb] = mma(a_hi[cs][mb], b_lo[cs][nb], acc[mb][nb]);
        if constexpr (!(AB & 4)) {
          // NPAIR x 2 half-conversions of two points each (hi), then the same (lo): every other slot / every fourth
          constexpr int NH = NPAIR * 2, STRIDE = 8 / NH;        // 4 -> every 2nd slot of a half; 2 -> every 4th
          if (i % STRIDE == 0) {
            const int l = (i & 7) / STRIDE;
            if (i < 8) conv_hi(v, vscale, l >> 1, l & 1); else conv_lo(v, vscale, l >> 1, l & 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {                            // ---- hi x hi
        const int nb = i >> 2, mb = i & 3;
        acc[mb][nb] = mma(a_hi[cs][mb], b_hi[cs][nb], acc[mb][nb]);
        if (i < 8) { if constexpr (!(AB & 4)) store_pt(b_st, i >> 2, i & 3); }
        else if (i < 8 + NROW) { if constexpr (!(AB & 2)) v[i - 8] = load_x1(x_soff, i - 8); }
        __builtin_amdgcn_sched_barrier(0);
      }
      PVCNN_PROBE(3);                                           // slot 3: the step's 48 MFMAs and everything between them
      if constexpr (!(AB & 16)) lds_barrier();
      PVCNN_PROBE(4);                                           // slot 4: the step barrier
      b_cur = b_nxt;
    }
  };
  for (int r = 0; r < rounds; ++r) {
    // the epilogue's per-row constants: requested now, parked in LDS behind the first group of steps (one read per row there instead
    // of 128 predicated loads that the compiler hoists to the top of the epilogue: seen in the ISA, 90 spilled registers)
    float2 row_const[ROWS / 256];
#pragma unroll
    for (int e = 0; e < ROWS / 256; ++e) {
      const int m = cur.mg * ROWS + e * 256 + tid;
      row_const[e] = make_float2((bias != nullptr && m < M) ? bias[m] : 0.0f, exp2_int(-wexp[m]));   // wexp covers the padded rows of the image
    }
    group(std::true_type{}, 0);
#pragma unroll
    for (int e = 0; e < ROWS / 256; ++e) row_lds[(r & 1) * ROWS + e * 256 + tid] = row_const[e];    // (published by the barriers that follow)
    for (int c0 = 4; c0 < chunks; c0 += 4) group(std::false_type{}, c0);
    // ---- the item's epilogue: D[i = m][j = point]; lanes = consecutive points (128-byte rows); bias; BatchNorm partial sums ----
    if (chunks == 4) lds_barrier();                             // (K = 64: no step barrier between the row constants' store and their readers)
    if (cur.tile < tiles_total) {
      // (the lane's coordinates are re-derived from the thread index HERE: kept across the step loop they cost registers that the
      //  loop does not have -- the compiler parked them in scratch)
      int tid_e = tid;
      asm volatile("" : "+v"(tid_e));
      const int j = tid_e & 31, kh = (tid_e >> 5) & 1;
      const int b = cur.tile / tiles_n, n0 = (cur.tile - b * tiles_n) * kPbN + cur.half * 128, m0 = cur.mg * ROWS + wm * TMI;
      const float x_unscale = 1.0f / cur.scale;                 // (a power of two: exact)
      const bool want_stats = stats_part != nullptr;
      // stores through a descriptor of the cloud's M x N outputs: rows >= M (the padded rows of the last channel block) are dropped by
      // the bounds check; the lane's offset is ONE register, the row a scalar multiple of the row pitch, the column block an immediate
      const __amdgpu_buffer_rsrc_t yrsrc = descriptor(y + (size_t)b * M * N, (uint32_t)M * row_bytes);
      const uint32_t yoff = (uint32_t)(m0 + 4 * kh) * row_bytes + (uint32_t)(n0 + wn * 128 + j) * 4u;
      // the point group of this wave's partial sums: its 128-point column group (WMW = 2) / the item's half (WMW = 4)
      const int sg = WMW == 4 ? cur.half : wn;
      // (eight rows at a time: the request streams of the NEXT item hold ~130 registers across this epilogue)
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
          float bv[8], unscale[8], ss[8], qq[8];
#pragma unroll
          for (int qi = 0; qi < 8; ++qi) {
            const int q = h8 * 8 + qi;
            const float2 rc = row_lds[(r & 1) * ROWS + wm * TMI + mb * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh];
            bv[qi] = rc.x;
            unscale[qi] = rc.y;
            ss[qi] = qq[qi] = 0.0f;
          }
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int qi = 0; qi < 8; ++qi) {
              const int q = h8 * 8 + qi;
              float v = acc[mb][nb][q] * unscale[qi] * x_unscale;        // powers of two: exact
              if (want_stats) {                                 // statistics of (y - bias), see bn_finalize_kernel
                ss[qi] += v;
                qq[qi] += v * v;
              }
              v += bv[qi];
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrsrc,
                                                    yoff + (uint32_t)(mb * 32 + (q & 3) + 8 * (q >> 2)) * row_bytes + (uint32_t)(nb * 128), 0, 0);
            }
          if (want_stats) {
            const float st2 = half_wave_sum8(ss, j), qt = half_wave_sum8(qq, j);
            const int q = h8 * 8 + ((j >> 2) & 7);
            if ((j & 3) == 0) stat_lds[sg * ROWS + wm * TMI + mb * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh] = make_float2(st2, qt);
          }
          __builtin_amdgcn_sched_barrier(0);                    // eight rows at a time, really
        }
      if (want_stats && (WMW == 2 || cur.half == 1)) {          // (uniform over the workgroup: every wave has the same item)
        lds_barrier();
#pragma unroll
        for (int e = 0; e < ROWS / 256; ++e) {
          const int rr = e * 256 + tid_e, m = cur.mg * ROWS + rr;
          if (m < M) {
            const float2 s0 = stat_lds[rr], s1 = stat_lds[ROWS + rr];
            stats_part[(size_t)m * tiles_total + cur.tile] = make_float2(s0.x + s1.x, s0.y + s1.y);
          }
        }
      }
    }
    cur = nxt;
    nxt = item_at(r + 2);
    PVCNN_PROBE(5);                                             // slot 5: the item's epilogue
  }
  PVCNN_PROBE_END();
}

static int pb_mb(int M) { return M > 64 ? 4 : 2; }

}  // namespace pvcnn

using namespace pvcnn;

static size_t pb_image_bytes(int KE, int ME, int nsplit) {
  const int TM = 32 * pb_mb(ME);
  return (size_t)ceil_div(KE, kPbK) * ceil_div(ME, TM) * nsplit * TM * kPbK * sizeof(uint16_t);
}

// nsplit: 1 = bf16, 3 = bf16x3, 2 = f16x2 (image followed by one int32 shift per padded output channel)
extern "C" size_t pvcnn_pwconv_weight_split_bytes(int Co, int Ci, int for_bwd_data, int nsplit) {
  if (Co <= 0 || Ci <= 0 || nsplit < 1 || nsplit > 3) return 0;
  const int KE = for_bwd_data ? Co : Ci, ME = for_bwd_data ? Ci : Co;
  const int TM = 32 * pb_mb(ME);
  return pb_image_bytes(KE, ME, nsplit) + (nsplit == 2 ? (size_t)ceil_div(ME, TM) * TM * sizeof(int) : 0);
}

extern "C" int pvcnn_pwconv_weight_split(const float *w, int Co, int Ci, int for_bwd_data, int nsplit, void *wts, void *stream) {
  PVCNN_REQUIRE(w && wts && Co > 0 && Ci > 0, "bad argument");
  PVCNN_REQUIRE(nsplit >= 1 && nsplit <= 3, "nsplit must be 1 (bf16), 2 (f16x2) or 3 (bf16x3)");
  PVCNN_REQUIRE(aligned16(wts), "wts must be 16-byte aligned");
  const int KE = for_bwd_data ? Co : Ci, ME = for_bwd_data ? Ci : Co;
  const int TM = 32 * pb_mb(ME);
  if (nsplit == 2) {
    int *wexp = reinterpret_cast<int *>(static_cast<char *>(wts) + pb_image_bytes(KE, ME, 2));
    hipLaunchKernelGGL(pw_weight_split_f16_kernel, dim3(ceil_div(ME, TM) * TM), dim3(256), 0, static_cast<hipStream_t>(stream), w, Co, Ci,
                       for_bwd_data, TM, static_cast<uint16_t *>(wts), wexp);
    return check_launch("pwconv_weight_split_f16");
  }
  const long total = (long)ceil_div(KE, kPbK) * ceil_div(ME, TM) * TM * kPbK;
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (nsplit == 1) hipLaunchKernelGGL(pw_weight_split_kernel<1>, grid, dim3(256), 0, s, w, Co, Ci, for_bwd_data, TM, static_cast<uint16_t *>(wts));
  else             hipLaunchKernelGGL(pw_weight_split_kernel<3>, grid, dim3(256), 0, s, w, Co, Ci, for_bwd_data, TM, static_cast<uint16_t *>(wts));
  return check_launch("pwconv_weight_split");
}

// both f16x2 images of w (forward + backward-data) in one launch; buffers sized by pvcnn_pwconv_weight_split_bytes(.., 0 / 1, 2)
extern "C" int pvcnn_pwconv_weight_split_pair(const float *w, int Co, int Ci, void *wts_fwd, void *wts_bwd, void *stream) {
  PVCNN_REQUIRE(w && wts_fwd && wts_bwd && Co > 0 && Ci > 0, "bad argument");
  PVCNN_REQUIRE(aligned16(wts_fwd) && aligned16(wts_bwd), "images must be 16-byte aligned");
  const int TM_f = 32 * pb_mb(Co), TM_b = 32 * pb_mb(Ci);
  const int rows_f = ceil_div(Co, TM_f) * TM_f, rows_b = ceil_div(Ci, TM_b) * TM_b;
  int *wexp_f = reinterpret_cast<int *>(static_cast<char *>(wts_fwd) + pb_image_bytes(Ci, Co, 2));
  int *wexp_b = reinterpret_cast<int *>(static_cast<char *>(wts_bwd) + pb_image_bytes(Co, Ci, 2));
  hipLaunchKernelGGL(pw_weight_split_f16_pair_kernel, dim3(rows_f + rows_b), dim3(256), 0, static_cast<hipStream_t>(stream), w, Co, Ci, rows_f,
                     TM_f, TM_b, static_cast<uint16_t *>(wts_fwd), wexp_f, static_cast<uint16_t *>(wts_bwd), wexp_b);
  return check_launch("pwconv_weight_split_pair");
}

// batched form of pvcnn_pwconv_weight_split_pair (see pvcnn_conv3d_weight_split_pair_entry / _batch)
extern "C" long pvcnn_pwconv_weight_split_pair_entry(const float *w, int Co, int Ci, void *wts_fwd, void *wts_bwd, long long *entry) {
  if (!w || !wts_fwd || !wts_bwd || !entry || Co <= 0 || Ci <= 0 || !aligned16(wts_fwd) || !aligned16(wts_bwd)) return -1;
  const int TM_f = 32 * pb_mb(Co), TM_b = 32 * pb_mb(Ci);
  const int rows_f = ceil_div(Co, TM_f) * TM_f, rows_b = ceil_div(Ci, TM_b) * TM_b;
  SplitEntry e;
  e.w = w;
  e.wts_f = static_cast<uint16_t *>(wts_fwd);
  e.wexp_f = reinterpret_cast<int *>(static_cast<char *>(wts_fwd) + pb_image_bytes(Ci, Co, 2));
  e.wts_b = static_cast<uint16_t *>(wts_bwd);
  e.wexp_b = reinterpret_cast<int *>(static_cast<char *>(wts_bwd) + pb_image_bytes(Co, Ci, 2));
  e.Co = Co; e.Ci = Ci; e.rows_f = rows_f; e.tm = (long long)TM_f | ((long long)TM_b << 32); e.row_begin = 0;
  memcpy(entry, &e, sizeof(e));
  return rows_f + rows_b;
}

extern "C" int pvcnn_pwconv_weight_split_pair_batch(const void *table, int n, long total_rows, void *stream) {
  PVCNN_REQUIRE(n >= 0 && total_rows >= 0 && total_rows <= 0x7fffffffL, "bad size");
  if (n == 0 || total_rows == 0) return 0;
  PVCNN_REQUIRE(table && (reinterpret_cast<uintptr_t>(table) & 7) == 0, "null or misaligned table");
  hipLaunchKernelGGL(pw_weight_split_f16_batch_kernel, dim3((unsigned)total_rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const SplitEntry *>(table), n);
  return check_launch("pwconv_weight_split_pair_batch");
}

// ... and of the plain-bf16 images (nsplit = 1; buffers sized by pvcnn_pwconv_weight_split_bytes(.., 0 / 1, 1))
extern "C" long pvcnn_pwconv_weight_split_pair_entry_bf16(const float *w, int Co, int Ci, void *wts_fwd, void *wts_bwd, long long *entry) {
  if (!w || !wts_fwd || !wts_bwd || !entry || Co <= 0 || Ci <= 0 || !aligned16(wts_fwd) || !aligned16(wts_bwd)) return -1;
  const int TM_f = 32 * pb_mb(Co), TM_b = 32 * pb_mb(Ci);
  const long blocks_f = ((long)ceil_div(Ci, kPbK) * ceil_div(Co, TM_f) * TM_f * kPbK + 255) / 256;
  const long blocks_b = ((long)ceil_div(Co, kPbK) * ceil_div(Ci, TM_b) * TM_b * kPbK + 255) / 256;
  SplitEntry e;
  e.w = w;
  e.wts_f = static_cast<uint16_t *>(wts_fwd); e.wexp_f = nullptr;
  e.wts_b = static_cast<uint16_t *>(wts_bwd); e.wexp_b = nullptr;
  e.Co = Co; e.Ci = Ci; e.rows_f = blocks_f; e.tm = (long long)TM_f | ((long long)TM_b << 32); e.row_begin = 0;
  memcpy(entry, &e, sizeof(e));
  return blocks_f + blocks_b;
}

extern "C" int pvcnn_pwconv_weight_split_pair_batch_bf16(const void *table, int n, long total_rows, void *stream) {
  PVCNN_REQUIRE(n >= 0 && total_rows >= 0 && total_rows <= 0x7fffffffL, "bad size");
  if (n == 0 || total_rows == 0) return 0;
  PVCNN_REQUIRE(table && (reinterpret_cast<uintptr_t>(table) & 7) == 0, "null or misaligned table");
  hipLaunchKernelGGL(pw_weight_split_bf16_batch_kernel, dim3((unsigned)total_rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const SplitEntry *>(table), n);
  return check_launch("pwconv_weight_split_pair_batch_bf16");
}

extern "C" size_t pvcnn_pwconv_fwd_split_stats_parts(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return (size_t)B * ceil_div(N, kPbN);
}

// y (B,M,N) = W x + bias with the pre-split weights (forward: K = Ci, M = Co; backward-data: x = grad_y, K = Co, M = Ci, bias NULL,
// for_bwd_data = 1 image).  stats_part: NULL or (M, *_split_stats_parts) float pairs of (sum, sum of squares) of (y - bias).
extern "C" int pvcnn_pwconv_fwd_split(const float *x, const void *wts, const float *bias, int B, int K, int M, int N, int nsplit,
                                      const void *x_absmax, int amax_seg, float *y, float *stats_part, void *stream) {
  PVCNN_REQUIRE(B >= 0 && K > 0 && M > 0 && N >= 0, "bad size");
  PVCNN_REQUIRE(amax_seg == 0 || amax_seg == kPbN, "amax_seg must be 0 (scalar scale) or 256 (one maximum per point tile)");
  PVCNN_REQUIRE(nsplit >= 1 && nsplit <= 3, "nsplit must be 1 (bf16), 2 (f16x2) or 3 (bf16x3)");
  PVCNN_REQUIRE(nsplit != 2 || x_absmax, "f16x2 needs the input's pvcnn_absmax_bits / pvcnn_absmax_tiles");
  if (B == 0 || N == 0) return 0;
  PVCNN_REQUIRE(x && wts && y && aligned16(wts), "null or misaligned pointer");
  PVCNN_REQUIRE(!stats_part || (reinterpret_cast<uintptr_t>(stats_part) & 7) == 0, "stats_part must be 8-byte aligned");
  PVCNN_REQUIRE((long)N * std::max(K, M) <= 0x7fffffffL, "cloud too large");
  const int tiles_n = ceil_div(N, kPbN), MB = pb_mb(M);
  const long tiles_total = (long)B * tiles_n;
  const long wgs = ((tiles_total + 7) / 8) * 8 * ceil_div(M, 32 * MB);       // tiles padded to the 8 XCDs
  PVCNN_REQUIRE(wgs <= 0x7fffffffL, "grid too large");
  const dim3 grid((unsigned)wgs);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint16_t *w16 = static_cast<const uint16_t *>(wts);
  float2 *sp = reinterpret_cast<float2 *>(stats_part);
  const uint32_t *am = static_cast<const uint32_t *>(x_absmax);
  const int *wexp = nsplit == 2 ? reinterpret_cast<const int *>(static_cast<const char *>(wts) + pb_image_bytes(K, M, 2)) : nullptr;
  const bool vec = N % 4 == 0 && N >= 4 && aligned16(x);       // straight-line chunk loop (see the kernel)
#define PVCNN_PB_LAUNCH_PF(NSV, MBV, PFV)                                                                                            \
  do {                                                                                                                               \
    if (vec) hipLaunchKernelGGL((pw_gemm_bf16_kernel<NSV, MBV, PFV, true>), grid, dim3(256), 0, s, x, w16, bias, y, K, M, N, tiles_n, \
                                (int)tiles_total, sp, am, wexp, amax_seg);                                                           \
    else hipLaunchKernelGGL((pw_gemm_bf16_kernel<NSV, MBV, PFV, false>), grid, dim3(256), 0, s, x, w16, bias, y, K, M, N, tiles_n,    \
                            (int)tiles_total, sp, am, wexp, amax_seg);                                                               \
  } while (0)
#define PVCNN_PB_LAUNCH(NSV, MBV) PVCNN_PB_LAUNCH_PF(NSV, MBV, 1)
  // prefetch depth of the wide f16x2 tile, measured (profiles/ab/r03c_pwbench_pf*.jsonl, 1472 -> 512 over 65 536 points): PF = 1 / 2 / 3
  // = 0.576 / 0.538 / 0.523 ms forward, 1788 / 1820 / 1824 clouds/s in the step; PF = 2 is kept (232 VGPRs; PF = 3 needs 252 of 256)
  if (nsplit == 3)      { if (MB == 4) PVCNN_PB_LAUNCH(3, 4); else PVCNN_PB_LAUNCH(3, 2); }
  else if (nsplit == 2) {
    // measured (profiles/ab/r03k_*, 1472 -> 512 over 65 536 points, forward): round-3 start 0.483 ms; straight-line chunk loop
    // (VEC) 0.347 ms; + conversion between the MFMAs, one barrier per chunk (pipe kernel) 0.306 ms; the step 1925 -> 2103 -> 2108 clouds/s
    // round 6: 256 output channels per workgroup, one persistent workgroup per CU (pw_gemm_f16_wide_kernel)
    static const bool wide_on = [] { const char *e = getenv("PVCNN_PW_WIDE"); return !(e && e[0] == '0'); }();
    const int mtiles128 = ceil_div(M, 128);
    if (wide_on && MB == 4 && vec && K % 64 == 0 && N % kPbN == 0 && M >= 256 && mtiles128 % 2 == 0 &&
        (long)B * std::max(K, M) * N * 4 < 0xffffffffL) {        // (buffer descriptors: 32-bit byte offsets inside a tensor)
      // 512 x 128 items (4 x 1 waves) where the image has a multiple of four 128-row blocks (PVCNN_PW_WIDE=2: the 256 x 256 items only)
      static const bool wide4_on = [] { const char *e = getenv("PVCNN_PW_WIDE"); return !(e && e[0] == '2'); }();
      // (K >= 256: with a handful of steps per item -- 128 -> 1024: eight -- the launch is its epilogues and stores, and the 256 x 256
      //  items are faster: 79 vs 94 us, tools/calls_r06/r06_call13)
      const int wmw = (wide4_on && mtiles128 % 4 == 0 && K >= 256) ? 4 : 2;
      const long turns_local = ((tiles_total + 7) / 8) * (mtiles128 / wmw);      // (a turn = one 256-point tile x one row group)
      const unsigned wide_grid = 8u * (unsigned)std::min<long>(kNumCU / 8, turns_local);
#define PVCNN_WIDE_LAUNCH(ABV)                                                                                                      \
      do {                                                                                                                              \
        if (wmw == 4)                                                                                                                   \
          hipLaunchKernelGGL((pw_gemm_f16_wide_kernel<4, ABV>), dim3(wide_grid), dim3(256), WideGeom<4>::LDS, s, x, w16, bias, y, K, M, N,  \
                             tiles_n, (int)tiles_total, sp, am, wexp, amax_seg, (unsigned)((size_t)B * K * N * 4),                      \
                             (unsigned)pb_image_bytes(K, M, 2));                                                                        \
        else                                                                                                                            \
          hipLaunchKernelGGL((pw_gemm_f16_wide_kernel<2, ABV>), dim3(wide_grid), dim3(256), WideGeom<2>::LDS, s, x, w16, bias, y, K, M, N,  \
                             tiles_n, (int)tiles_total, sp, am, wexp, amax_seg, (unsigned)((size_t)B * K * N * 4),                      \
                             (unsigned)pb_image_bytes(K, M, 2));                                                                        \
      } while (0)
#ifdef PVCNN_ABLATE
      const char *ab_env = getenv("PVCNN_PW_ABLATE");
      switch (ab_env ? atoi(ab_env) : 0) {
        case 1: PVCNN_WIDE_LAUNCH(1); break;
        case 2: PVCNN_WIDE_LAUNCH(2); break;
        case 4: PVCNN_WIDE_LAUNCH(4); break;
        case 8: PVCNN_WIDE_LAUNCH(8); break;
        case 16: PVCNN_WIDE_LAUNCH(16); break;
        case 31: PVCNN_WIDE_LAUNCH(31); break;
        default: PVCNN_WIDE_LAUNCH(0);
      }
#else
      PVCNN_WIDE_LAUNCH(0);
#endif
#undef PVCNN_WIDE_LAUNCH
    } else if (MB == 4 && vec)
      hipLaunchKernelGGL(pw_gemm_f16_pipe_kernel<2>, grid, dim3(256), 0, s, x, w16, bias, y, K, M, N, tiles_n, (int)tiles_total, sp, am, wexp,
                         amax_seg);
    else if (MB == 4) PVCNN_PB_LAUNCH_PF(2, 4, 2);
    else PVCNN_PB_LAUNCH(2, 2);
  }
  else if (MB == 4 && vec)   // bf16 operands (autocast): the same pipelined structure with one plane
    hipLaunchKernelGGL(pw_gemm_f16_pipe_kernel<1>, grid, dim3(256), 0, s, x, w16, bias, y, K, M, N, tiles_n, (int)tiles_total, sp, am, wexp,
                       amax_seg);
  else                  { if (MB == 4) PVCNN_PB_LAUNCH(1, 4); else PVCNN_PB_LAUNCH(1, 2); }
#undef PVCNN_PB_LAUNCH
#undef PVCNN_PB_LAUNCH_PF
  return check_launch("pwconv_fwd_split");
}
