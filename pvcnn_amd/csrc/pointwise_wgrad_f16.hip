// pointwise_wgrad_f16.hip -- backward-weight of the SharedMLP 1x1 convolutions on the fp16 matrix cores, "f16x2" arithmetic
// (split16.h: both fp32 operands scaled by a power of two, split into fp16 hi + lo, three exact partial products, fp32 accumulate).
//
//   grad_w[m][k] = sum over (b, n) of grad_y[b][m][n] * x[b][k][n]          (m = output channels, k = input channels, n = points)
//
// Both operands are channel-major with the reduction index (points) contiguous, which is exactly what v_mfma_f32_32x32x16_f16 wants
// (8 consecutive reduction elements per lane for a row of A and for a column of B): no transposition anywhere.
//   * a workgroup (4 waves, 2 x 2, 64 x 64 each) owns a 128 x 128 tile of grad_w for one partition of the points (split-K);
//   * per chunk of 32 points it converts a 128 x 32 slab of grad_y and of x (whole 128-byte lines in, fp16 hi / lo planes out, 64-byte
//     rows whose four 16-byte slots are permuted by (row >> 2) & 3: the fragment reads of 16 rows hit 64 distinct banks); the next two
//     chunks' slabs are in registers / in flight while this one is multiplied, and the conversion runs between the MFMAs;
//   * partial tiles go to part[p] ([m][k]: 128-byte rows), pw_wgrad_f16_reduce_kernel sums the partitions in a fixed order
//     (deterministic, no atomics) and scales back by 2^-(sx + sgy); grad_bias falls out of the grad_y slabs a thread converts.
// N % 4 == 0 (16-byte loads); other shapes stay on the fp32-MFMA kernel of pointwise.hip.
// Round 2 (convert between two barriers, 80-byte rows): 0.56 ms at (16, 1472 -> 512, 4096) (fp32 MFMA: 0.87).  Tried and slower: 128 x 256
// tiles (64 x 128 per wave: 256 VGPRs and spills, 0.95 ms); two chunks in flight with CONDITIONAL loads (the compiler then waits on
// vmcnt(0) at every conversion: 1.09 ms).
#include <algorithm>

#include "common.h"
#include "split16.h"

namespace pvcnn {

constexpr int kGwM = 128, kGwK = 128, kGwPc = 32;
constexpr int kGwRowB = kGwPc * 2;                  // 64-byte rows; the four 16-byte slots of row r are permuted by (r >> 2) & 3
constexpr int kGwPlane = 128 * kGwRowB;             // one fp16 plane of one operand slab (8 KiB)
constexpr int kGwBuf = 4 * kGwPlane;                // [grad_y hi, lo][x hi, lo]

// Pipelined like pw_gemm_f16_pipe_kernel (pointwise_bf16.hip): the slab buffer is double-buffered (2 x 32 KiB) and a chunk costs ONE
// barrier.  In iteration i a wave reads its fragments of slab i, issues the 24 MFMAs with the conversion of slab i + 1 (in registers
// since iteration i - 2) scheduled between them -- ~5 vector-ALU instructions per MFMA, the matrix pipe is busy for 8 -- writes the
// converted rows into the other buffer, requests slab i + 3 into the registers just freed and meets the other waves.  The loop is
// straight-line code (the chunk count is padded to even; loads past the end re-read the last chunk and are converted with scale 0):
// with a branch around a load the compiler cannot count the loads in flight and drains them all (vmcnt(0)) once per iteration.
__global__ __launch_bounds__(256, 2) void pw_wgrad_f16_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                              const uint32_t *__restrict__ x_absmax, const uint32_t *__restrict__ gy_absmax,
                                                              int K, int M, int N, int P, int mtiles, int ktiles, int cps,
                                                              int total_chunks, float *__restrict__ part, float *__restrict__ gb_part,
                                                              long x_words, long gy_words, uint32_t *__restrict__ maxima) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kGwBuf];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int wm = wave >> 1, wk = wave & 1;
  // XCD-aware order: workgroup ids go round-robin over the 8 XCDs (each with its own L2).  ALL tiles of a partition of the points
  // run on one XCD, next to each other in its queue: a grad_y row is needed by the ktiles workgroups of its row block and an x row by
  // the mtiles workgroups of its column block -- with the partitions spread over the XCDs (round 2) every workgroup pulled its 32 KiB
  // per chunk through the fabric (1472 -> 512: 3.1 GB per call, 0.45 ms at ~7 TB/s); now the XCD fetches a chunk once.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, tiles = mtiles * ktiles;
  const int grp = slot / tiles, tile = slot - grp * tiles, p = xcd + 8 * grp;
  if (p >= P) return;
  const int kt = tile % ktiles, mt = tile / ktiles;
  const int m0 = mt * kGwM, k0 = kt * kGwK;
  // (ABI v12) the two global maxima from the tables when the buffers are amax buffers (word [0] may be unwritten), handed to the
  // reduce launch through `maxima`
  const uint32_t x_max = amax_table_value(x_absmax, x_words), gy_max = amax_table_value(gy_absmax, gy_words);
  if (blockIdx.x == 0 && threadIdx.x == 0) { maxima[0] = x_max; maxima[1] = gy_max; }
  const float x_scale = exp2_int(scale_shift(x_max)), gy_scale = exp2_int(scale_shift(gy_max));

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b2][r] = 0.0f;

  // staging: item u of a thread = row r0 + 32 u of the 256-row slab (rows 0..127 grad_y, 128..255 x), point quad q.  The loads are
  // branch-free: a row outside the tensor reads row 0, a quad outside the cloud reads quad 0, a chunk past the partition re-reads its
  // last chunk -- all of them are multiplied by a zero scale in the conversion (and land in rows / columns nobody reads).
  const int q = tid & 7, r0 = tid >> 3;
  const float *rowp[8];
  float rowsc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int row = r0 + 32 * (u & 3);
    const bool ok = u < 4 ? (m0 + row < M) : (k0 + row < K);
    rowsc[u] = ok ? (u < 4 ? gy_scale : x_scale) : 0.0f;
    rowp[u] = u < 4 ? gy + (size_t)(ok ? m0 + row : 0) * N : x + (size_t)(ok ? k0 + row : 0) * N;
  }
  // byte offset of this thread's 8-byte store inside a plane (row r0 + 32 u: the same swizzle for every u), and of a lane's fragments
  const int st = r0 * kGwRowB + (((q >> 1) ^ ((r0 >> 2) & 3)) << 4) + ((q & 1) << 3);
  const int fsw = (j >> 2) & 3;
  float gsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const int c_begin = (int)((long)total_chunks * p / P), c_end = (int)((long)total_chunks * (p + 1) / P);
  auto load = [&](int c, float4 (&v)[8]) {
    c = min(c, c_end - 1);
    const int b = c / cps, n = (c - b * cps) * kGwPc + 4 * q;
    const size_t og = (size_t)b * M * N + (n < N ? n : 0), ox = (size_t)b * K * N + (n < N ? n : 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(rowp[u] + (u < 4 ? og : ox));
  };
  // slab of chunk c (zeros when c >= c_end) -> buffer `buf`
  auto convert = [&](int c, const float4 (&v)[8], int buf) {
    const int cc = min(c, c_end - 1), b = cc / cps, n = (cc - b * cps) * kGwPc + 4 * q;
    const float live = ((int)(c < c_end) & (int)(n < N)) ? 1.0f : 0.0f;                // (no short circuit: the loop stays one block)
    unsigned char *base = lds + buf * kGwBuf + st;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float sc = rowsc[u] * live;
      uint32_t w0[2], w1[2];
      split_pair<2>(v[u].x * sc, v[u].y * sc, w0);
      split_pair<2>(v[u].z * sc, v[u].w * sc, w1);
      unsigned char *dst = base + (u < 4 ? 0 : 2 * kGwPlane) + 32 * (u & 3) * kGwRowB;
      *reinterpret_cast<uint2 *>(dst) = make_uint2(w0[0], w1[0]);
      *reinterpret_cast<uint2 *>(dst + kGwPlane) = make_uint2(w0[1], w1[1]);
      if (u < 4) gsum[u] += sc != 0.0f ? (v[u].x + v[u].y) + (v[u].z + v[u].w) : 0.0f;
    }
  };

  float4 ring[2][8];
  const int n_chunks = c_end - c_begin;
  if (n_chunks > 0) {
    load(c_begin, ring[0]);
    load(c_begin + 1, ring[1]);
    convert(c_begin, ring[0], 0);
    load(c_begin + 2, ring[0]);
    __syncthreads();
  }
  for (int i0 = 0; i0 < n_chunks; i0 += 2) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int c = c_begin + i0 + d;
      // slab c is published in buffer d; ring[d ^ 1] holds the rows of chunk c + 1 (landed), ring[d] those of chunk c + 2
      const unsigned char *gl = lds + d * kGwBuf, *xl = gl + 2 * kGwPlane;
      uint4 a[2][2][2], bq[2][2][2];                            // [k-step][block][hi, lo]
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            const int off = (((2 * ks + kh) ^ fsw) << 4);
            a[ks][blk][pl] = *reinterpret_cast<const uint4 *>(gl + pl * kGwPlane + (wm * 64 + blk * 32 + j) * kGwRowB + off);
            bq[ks][blk][pl] = *reinterpret_cast<const uint4 *>(xl + pl * kGwPlane + (wk * 64 + blk * 32 + j) * kGwRowB + off);
          }
      __builtin_amdgcn_sched_barrier(0);
      convert(c + 1, ring[d ^ 1], d ^ 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma16<2>(a[ks][mb][1], bq[ks][nb][0], acc[mb][nb]);    // lo x hi
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma16<2>(a[ks][mb][0], bq[ks][nb][1], acc[mb][nb]);    // hi x lo
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma16<2>(a[ks][mb][0], bq[ks][nb][0], acc[mb][nb]);    // hi x hi
      }
#pragma unroll
      for (int i = 0; i < 24; ++i) {                            // 1 MFMA, 5 vector-ALU, 24 times; the LDS stores follow their conversions
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        if (i % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      load(c + 3, ring[d ^ 1]);
      __syncthreads();
    }
  }

  // ---- epilogue: part[p][m][k] over the padded (MP x KP) grid, lanes along k ----
  const int MP = mtiles * kGwM, KP = ktiles * kGwK;
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

// ---- wide tile: 256 x 256 of grad_w per workgroup (8 waves, 2 x 4, 128 x 64 each), one workgroup per CU ----
// pw_wgrad_f16_kernel converts 32 elements per thread for 24 MFMAs and pulls 32 KiB per chunk and workgroup through the memory system:
// at 1472 -> 512 it is bound by both (issue slots: ~5.4 conversions' worth of vector ALU per MFMA gap on a SIMD shared by two waves;
// traffic: 3.1 GB per call at ~7 TB/s).  The 256 x 256 tile halves both per MFMA: a thread converts 32 elements for 48 MFMAs, grad_y is
// re-read by K / 256 and x by M / 256 workgroups.  Per chunk of 32 points a wave multiplies k-step 0 with the conversion of the next
// chunk's grad_y rows between the MFMAs, requests the grad_y rows of the chunk after next into the registers just freed, does the same
// with k-step 1 and the x rows, and meets the other waves: one barrier per chunk, slabs double-buffered (2 x 64 KiB).  Same XCD-aware
// order as pw_wgrad_f16_kernel.
// Two shapes of the 8-wave tile (a wave owns WMB x WNB MFMA tiles, the waves are arranged WR x WC): 256 x 256 = <4, 2, 2, 4> and
// 256 x 192 = <2, 3, 4, 2> -- picked per layer so that the tiles of the partitions fill the XCDs' 32 CUs (1472 -> 512: 2 x 6 tiles of
// 256 x 256 leave a quarter of the chip idle, 2 x 8 tiles of 256 x 192 use all of it).
template <int WMB, int WNB, int WR, int WC>
struct GwWide {
  static constexpr int TM = WR * WMB * 32, TK = WC * WNB * 32;
  static constexpr int GI = TM / 64, XI = TK / 64, NI = GI + XI;              // staging items per thread: grad_y rows, x rows
  static constexpr int GPL = TM * kGwRowB, XPL = TK * kGwRowB;                // one fp16 plane of the grad_y / x slab
  static constexpr int BUF = 2 * GPL + 2 * XPL;                               // [grad_y hi, lo][x hi, lo]
};

template <int WMB, int WNB, int WR, int WC>
__global__ __launch_bounds__(512) void pw_wgrad_f16_wide_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                const uint32_t *__restrict__ x_absmax, const uint32_t *__restrict__ gy_absmax,
                                                                int K, int M, int N, int P, int mtiles, int ktiles, int cps, int total_chunks,
                                                                float *__restrict__ part, float *__restrict__ gb_part, long x_words, long gy_words,
                                                                uint32_t *__restrict__ maxima) {
  using T = GwWide<WMB, WNB, WR, WC>;
  static_assert(WR * WC == 8 && T::TM % 64 == 0 && T::TK % 64 == 0, "8 waves; 64-row staging items");
  constexpr int GI = T::GI, NI = T::NI;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];          // 2 * T::BUF
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);     // (for the epilogue: survives the chunk loop in a scalar register)
  const int wm = wave / WC, wk = wave % WC;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, tiles = mtiles * ktiles;
  const int grp = slot / tiles, tile = slot - grp * tiles, p = xcd + 8 * grp;
  if (p >= P) return;
  const int kt = tile % ktiles, mt = tile / ktiles;
  const int m0 = mt * T::TM, k0 = kt * T::TK;
  // (ABI v12) the two global maxima from the tables when the buffers are amax buffers (word [0] may be unwritten), handed to the
  // reduce launch through `maxima`
  const uint32_t x_max = amax_table_value(x_absmax, x_words), gy_max = amax_table_value(gy_absmax, gy_words);
  if (blockIdx.x == 0 && threadIdx.x == 0) { maxima[0] = x_max; maxima[1] = gy_max; }
  const float x_scale = exp2_int(scale_shift(x_max)), gy_scale = exp2_int(scale_shift(gy_max));

  f32x16 acc[WMB][WNB];
#pragma unroll
  for (int a = 0; a < WMB; ++a)
#pragma unroll
    for (int b2 = 0; b2 < WNB; ++b2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b2][r] = 0.0f;

  // staging: item u of a thread = row r0 + 64 u of the slab (items 0 .. GI - 1 grad_y rows, GI .. NI - 1 x rows), point quad q; branch-free loads
  // (rows / quads / chunks outside read valid memory and are converted with scale 0, see pw_wgrad_f16_kernel)
  const int q = tid & 7, r0 = tid >> 3;
  uint32_t rowmask = 0;                                         // bit u: the row of item u exists (else: the tensor's last row, scale 0)
#pragma unroll
  for (int u = 0; u < NI; ++u) rowmask |= ((u < GI ? (m0 + r0 + 64 * u < M) : (k0 + r0 + 64 * (u - GI) < K)) ? 1u : 0u) << u;
  auto rowoff = [&](int u) {                                    // element offset of the row inside one cloud (recomputed: registers)
    return (uint32_t)(u < GI ? min(m0 + r0 + 64 * u, M - 1) : min(k0 + r0 + 64 * (u - GI), K - 1)) * (uint32_t)N;
  };
  const int st = r0 * kGwRowB + (((q >> 1) ^ ((r0 >> 2) & 3)) << 4) + ((q & 1) << 3);
  const int fsw = (j >> 2) & 3;
  float gsum[GI];
#pragma unroll
  for (int u = 0; u < GI; ++u) gsum[u] = 0.0f;
  const int c_begin = (int)((long)total_chunks * p / P), c_end = (int)((long)total_chunks * (p + 1) / P);
  // half h of a chunk's slab: h = 0 the grad_y rows, h = 1 the x rows
  auto load_half = [&](int c, int h, float4 (&v)[NI]) {
    c = min(c, c_end - 1);
    const int b = c / cps, n = (c - b * cps) * kGwPc + 4 * q;
    const float *pb = (h == 0 ? gy + (size_t)b * M * N : x + (size_t)b * K * N) + (n < N ? n : 0);
#pragma unroll
    for (int u = h ? GI : 0; u < (h ? NI : GI); ++u) v[u] = *reinterpret_cast<const float4 *>(pb + rowoff(u));
  };
  auto convert_half = [&](int c, int h, const float4 (&v)[NI], int buf) {    // (zeros when c >= c_end) -> buffer `buf`
    const int cc = min(c, c_end - 1), b = cc / cps, n = (cc - b * cps) * kGwPc + 4 * q;
    const bool live = (int)(c < c_end) & (int)(n < N);
    const float sh = live ? (h == 0 ? gy_scale : x_scale) : 0.0f;
    unsigned char *base = lds + buf * T::BUF + st + (h == 0 ? 0 : 2 * T::GPL);
    const int plane = h == 0 ? T::GPL : T::XPL;
#pragma unroll
    for (int u = h ? GI : 0; u < (h ? NI : GI); ++u) {
      const float sc = ((rowmask >> u) & 1u) ? sh : 0.0f;
      uint32_t w0[2], w1[2];
      split_pair<2>(v[u].x * sc, v[u].y * sc, w0);
      split_pair<2>(v[u].z * sc, v[u].w * sc, w1);
      unsigned char *dst = base + 64 * (h ? u - GI : u) * kGwRowB;
      *reinterpret_cast<uint2 *>(dst) = make_uint2(w0[0], w1[0]);
      *reinterpret_cast<uint2 *>(dst + plane) = make_uint2(w0[1], w1[1]);
      if (h == 0) gsum[h ? 0 : u] += sc != 0.0f ? (v[u].x + v[u].y) + (v[u].z + v[u].w) : 0.0f;
    }
  };
  auto multiply = [&](int buf, int ks) {
    const unsigned char *gl = lds + buf * T::BUF, *xl = gl + 2 * T::GPL;
    const int off = ((2 * ks + kh) ^ fsw) << 4;
    auto arow = [&](int mb, int pl) { return *reinterpret_cast<const uint4 *>(gl + pl * T::GPL + ((wm * WMB + mb) * 32 + j) * kGwRowB + off); };
    auto brow = [&](int nb, int pl) { return *reinterpret_cast<const uint4 *>(xl + pl * T::XPL + ((wk * WNB + nb) * 32 + j) * kGwRowB + off); };
    // fragments in two rounds (fewer registers live): grad_y lo + x hi, then grad_y hi (over the lo registers) + x lo
    uint4 a[WMB], bh[WNB], bl[WNB];
#pragma unroll
    for (int mb = 0; mb < WMB; ++mb) a[mb] = arow(mb, 1);
#pragma unroll
    for (int nb = 0; nb < WNB; ++nb) bh[nb] = brow(nb, 0);
#pragma unroll
    for (int mb = 0; mb < WMB; ++mb)
#pragma unroll
      for (int nb = 0; nb < WNB; ++nb) acc[mb][nb] = mfma16<2>(a[mb], bh[nb], acc[mb][nb]);      // lo x hi
#pragma unroll
    for (int mb = 0; mb < WMB; ++mb) a[mb] = arow(mb, 0);
#pragma unroll
    for (int nb = 0; nb < WNB; ++nb) bl[nb] = brow(nb, 1);
#pragma unroll
    for (int mb = 0; mb < WMB; ++mb)
#pragma unroll
      for (int nb = 0; nb < WNB; ++nb) acc[mb][nb] = mfma16<2>(a[mb], bl[nb], acc[mb][nb]);      // hi x lo
#pragma unroll
    for (int mb = 0; mb < WMB; ++mb)
#pragma unroll
      for (int nb = 0; nb < WNB; ++nb) acc[mb][nb] = mfma16<2>(a[mb], bh[nb], acc[mb][nb]);      // hi x hi
  };
  // one k-step of slab `buf` with the conversion of half h of the next chunk between its MFMAs (n x [1 MFMA, 3 - 4 vector-ALU], the LDS
  // stores behind their conversions), then the request for the same half of the chunk after next into the registers just freed: every
  // load has a whole chunk period to land, with ONE register set
  auto step = [&](int c, int d, int h, float4 (&ring)[NI]) {
    // the conversions depend on these (empty) statements: the instruction selector otherwise emits them -- and the wait for their
    // loads -- in front of the previous k-step's MFMAs (pure arithmetic: a sched_barrier does not order them)
#pragma unroll
    for (int u = h ? GI : 0; u < (h ? NI : GI); ++u) asm volatile("" : "+v"(ring[u].x), "+v"(ring[u].y), "+v"(ring[u].z), "+v"(ring[u].w));
    multiply(d, h);
    convert_half(c + 1, h, ring, d ^ 1);
#pragma unroll
    for (int i = 0; i < 3 * WMB * WNB; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, WMB * WNB >= 8 ? 3 : 4, 0);
      if (i % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    load_half(c + 2, h, ring);
    __builtin_amdgcn_sched_barrier(0);
  };

  float4 ring[NI];
  const int n_chunks = c_end - c_begin;
  if (n_chunks > 0) {
    load_half(c_begin, 0, ring);
    load_half(c_begin, 1, ring);
    convert_half(c_begin, 0, ring, 0);
    convert_half(c_begin, 1, ring, 0);
    load_half(c_begin + 1, 0, ring);
    load_half(c_begin + 1, 1, ring);
    __syncthreads();
  }
  for (int i0 = 0; i0 < n_chunks; i0 += 2) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int c = c_begin + i0 + d;                           // slab c is published in buffer d (zeros when c >= c_end)
      __builtin_amdgcn_sched_barrier(0);
      step(c, d, 0, ring);
      step(c, d, 1, ring);
      __syncthreads();
    }
  }

  // ---- epilogue: part[p][m][k] over the padded (MP x KP) grid, lanes along k ----
  // (the epilogue's lane-dependent offsets are derived from a laundered thread index: the compiler otherwise computes them in front of
  // the chunk loop and -- at the 256-register cap of this kernel -- spills them across it: 5 dwords of scratch per lane in round 3)
  int tid_e = wave_s * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // wave index (scalar register) + lane
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63, j_e = lane_e & 31, kh_e = lane_e >> 5, wave_e = tid_e >> 6, wm_e = wave_e / WC, wk_e = wave_e % WC;
  const int MP = mtiles * T::TM, KP = ktiles * T::TK;
  float *pp = part + (size_t)p * MP * KP;
#pragma unroll
  for (int mb = 0; mb < WMB; ++mb)
#pragma unroll
    for (int nb = 0; nb < WNB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm_e * WMB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh_e;
        pp[(size_t)m * KP + k0 + (wk_e * WNB + nb) * 32 + j_e] = acc[mb][nb][r];
      }
  if (gb_part != nullptr && kt == 0) {                          // grad_bias partial: the 8 quads of a row, fixed order
    __syncthreads();
    float *red = reinterpret_cast<float *>(lds);                // [TM rows][8 quads]
    const int q_e = tid_e & 7, r0_e = tid_e >> 3;
#pragma unroll
    for (int u = 0; u < GI; ++u) red[(r0_e + 64 * u) * 8 + q_e] = gsum[u];
    __syncthreads();
    if (tid_e < T::TM) {
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += red[tid_e * 8 + i];
      gb_part[(size_t)p * MP + m0 + tid_e] = s;
    }
  }
}

__global__ __launch_bounds__(256) void pw_wgrad_f16_reduce_kernel(const float *__restrict__ part, const float *__restrict__ gb_part,
                                                                  const uint32_t *__restrict__ maxima,
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
      gw[(size_t)m * K + k] = ((s0 + s1) + (s2 + s3)) * exp2_int(-scale_shift(maxima[0])) * exp2_int(-scale_shift(maxima[1]));
    }
  }
  if (gb != nullptr && e < (size_t)M) {
    float s = 0.0f;
    for (int i = 0; i < P; ++i) s += gb_part[(size_t)i * MP + e];
    gb[e] = s;
  }
}

struct PwWgradPlan { int mtiles, ktiles, cps, total_chunks, P, tm, tk, wide; size_t part_floats, gb_floats; };   // wide: 0 narrow, 1 = 256 x 256, 2 = 256 x 192

static PwWgradPlan pw_wgrad_f16_plan(int B, int K, int M, int N) {
  PwWgradPlan w;
  w.cps = ceil_div(N, kGwPc);
  w.total_chunks = B * w.cps;
  // partitions of the points: a multiple of 8 (one XCD runs all tiles of a partition, see the kernels), as many as fill the XCD's
  // workgroup slots (narrow tile: two per CU = 64; wide tiles: one per CU = 32) in ONE round -- rounded down (1472 -> 512 with 48 narrow
  // tiles: 11 partitions = 528 workgroups ran a second round of 16 on an otherwise idle chip)
  auto partitions = [&](int tiles, int slots) { return std::max(1, std::min(w.total_chunks, 8 * std::max(1, slots / tiles))); };
  // tile: the wide shape (256 x 256 or 256 x 192) that wastes least of the chip -- idle CUs of the XCD, padded rows -- when both extents
  // are wide enough and a workgroup keeps >= 64 chunks (fewer: the 256 KiB partial tile it writes, and the reduction over as many
  // partitions, cost more than the tile saves: 512 -> 256 over 65 536 points stays on the 128 x 128 kernel); else 128 x 128
  w.wide = 0;
  double best = 0.0;
  for (int cand = 1; cand <= 2 && M >= 192 && K >= 160; ++cand) {
    const int tm = 256, tk = cand == 1 ? 256 : 192, mt = ceil_div(M, tm), kt = ceil_div(K, tk), tiles = mt * kt, P = partitions(tiles, 32);
    if (w.total_chunks / P < 64) continue;
    const int per_xcd = ceil_div(P, 8) * tiles;
    const double fill = (double)per_xcd / (32.0 * ceil_div(per_xcd, 32));
    const double score = fill * ((double)M * K) / ((double)mt * tm * kt * tk);
    if (score > best && score >= 0.6) { best = score; w.wide = cand; }
  }
  w.tm = w.wide ? 256 : kGwM;
  w.tk = w.wide == 1 ? 256 : w.wide == 2 ? 192 : kGwK;
  w.mtiles = ceil_div(M, w.tm);
  w.ktiles = ceil_div(K, w.tk);
  w.P = partitions(w.mtiles * w.ktiles, w.wide ? 32 : 64);
  w.part_floats = (size_t)w.P * w.mtiles * w.tm * w.ktiles * w.tk;
  w.gb_floats = (size_t)w.P * w.mtiles * w.tm + 4;             // + the two global maxima the main launch hands to the reduce launch
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
extern "C" int pvcnn_pwconv_bwd_weight_f16(const float *x, const float *grad_y, const void *x_absmax, int x_amax_seg, const void *gy_absmax,
                                           int gy_amax_seg, int B, int K, int M, int N, float *grad_w, float *grad_bias, void *workspace,
                                           size_t workspace_bytes, void *stream) {
  PVCNN_REQUIRE(x_amax_seg >= 0 && gy_amax_seg >= 0, "amax_seg < 0");
  PVCNN_REQUIRE(B > 0 && K > 0 && M > 0 && N > 0 && N % 4 == 0, "bad size (N must be a multiple of 4)");
  PVCNN_REQUIRE(x && grad_y && grad_w && x_absmax && gy_absmax, "null pointer");
  PVCNN_REQUIRE(aligned16(x) && aligned16(grad_y), "x and grad_y must be 16-byte aligned");
  PVCNN_REQUIRE((long)N * std::max(K, M) <= 0x7fffffffL, "cloud too large");
  PVCNN_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pvcnn_pwconv_bwd_weight_f16_workspace_bytes(B, K, M, N),
                "workspace missing, misaligned or too small (see pvcnn_pwconv_bwd_weight_f16_workspace_bytes)");
  const PwWgradPlan w = pw_wgrad_f16_plan(B, K, M, N);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint32_t *xa = static_cast<const uint32_t *>(x_absmax), *ga = static_cast<const uint32_t *>(gy_absmax);
  float *part = static_cast<float *>(workspace), *gb_part = part + w.part_floats;
  uint32_t *maxima = reinterpret_cast<uint32_t *>(gb_part + w.gb_floats - 4);
  const long x_words = x_amax_seg > 0 ? (long)B * ceil_div(N, x_amax_seg) : 0L, gy_words = gy_amax_seg > 0 ? (long)B * ceil_div(N, gy_amax_seg) : 0L;
  const dim3 grid((unsigned)(8 * ceil_div(w.P, 8) * w.mtiles * w.ktiles));
  if (w.wide) {
    using WA = GwWide<4, 2, 2, 4>;
    using WB = GwWide<2, 3, 4, 2>;
    auto ka = pw_wgrad_f16_wide_kernel<4, 2, 2, 4>;
    auto kb = pw_wgrad_f16_wide_kernel<2, 3, 4, 2>;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ka), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WA::BUF);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WB::BUF);
      if (e != hipSuccess) { set_error("pwconv_wgrad_f16: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
      attr_set = true;
    }
    if (w.wide == 1)
      hipLaunchKernelGGL(ka, grid, dim3(512), 2 * WA::BUF, s, x, grad_y, xa, ga, K, M, N, w.P, w.mtiles, w.ktiles, w.cps, w.total_chunks, part,
                         grad_bias ? gb_part : nullptr, x_words, gy_words, maxima);
    else
      hipLaunchKernelGGL(kb, grid, dim3(512), 2 * WB::BUF, s, x, grad_y, xa, ga, K, M, N, w.P, w.mtiles, w.ktiles, w.cps, w.total_chunks, part,
                         grad_bias ? gb_part : nullptr, x_words, gy_words, maxima);
  } else {
    hipLaunchKernelGGL(pw_wgrad_f16_kernel, grid, dim3(256), 0, s, x, grad_y, xa, ga, K, M, N, w.P, w.mtiles, w.ktiles, w.cps, w.total_chunks,
                       part, grad_bias ? gb_part : nullptr, x_words, gy_words, maxima);
  }
  if (int rc = check_launch("pwconv_wgrad_f16")) return rc;
  const int MP = w.mtiles * w.tm, KP = w.ktiles * w.tk;
  hipLaunchKernelGGL(pw_wgrad_f16_reduce_kernel, dim3((unsigned)ceil_div(MP * KP, 256)), dim3(256), 0, s, part, gb_part, maxima, w.P, MP, KP, M, K,
                     grad_w, grad_bias);
  return check_launch("pwconv_wgrad_f16_reduce");
}
