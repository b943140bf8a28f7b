// conv3d_wgrad_f16.hip -- backward-weight of the 3x3x3 voxel convolution on the fp16 matrix cores, "f16x2" arithmetic
// (split16.h / top of conv3d_bf16.hip: both fp32 operands scaled by a power of two and split into fp16 hi + lo, three exact
// partial products per fp32 product, fp32 accumulation -- fp32-class accuracy at 3 MFMAs per 16-deep k-step).
//
//   grad_w[co][ci][tap] = sum over (b, voxel) of grad_y[b][co][voxel] * x[b][ci][voxel + offset(tap)]
//
// is a GEMM with M = co, N = (ci, tap) and K = every voxel of the batch: the output is tiny (Co x Ci x 27), K is huge, both
// operands are activations (nothing can be pre-split) and the 27 taps read the SAME x values at shifted positions.  Structure:
//   * D[co][ci] += A[co][k] * B[k][ci] on v_mfma_f32_32x32x16_f16 with k = 16 consecutive z voxels of one (x, y) output row:
//     A = grad_y row segments, B = x row segments of the (dx, dy) neighbour row, shifted by dz along z;
//   * a workgroup owns a 64 (co) x 32 (ci) x 27 (tap) block of grad_w -- 54 accumulator tiles -- as 18 units (dx, dy; 32-row co
//     block) of three dz tiles; its 8 waves take 3, 3, 2, 2, 2, 2, 2, 2 units, i.e. 5, 5, 4, 4 per SIMD;
//   * the three dz taps of a unit come from ONE LDS window of the x row (a 16-byte read plus the dword on either side): dz = 1 is
//     the aligned middle, dz = 0 / 2 are v_alignbit funnel shifts by one fp16 -- no shifted copies of x in LDS;
//   * K runs over "strips" (b, x plane): the workgroup walks y, keeping a ring of four y rows of the three x planes and a
//     double-buffered grad_y row in LDS (fp16 hi and lo planes); per step it loads ONE new row of each (whole 128-byte lines),
//     multiplies the current one, then converts and stores -- one barrier per output row;
//   * split-K over P partitions of the strips; every workgroup writes its block to part[p] ([tap][co][ci]: 128-byte rows) and
//     conv3d_wgrad_f16_reduce_kernel sums the partitions in a fixed order (deterministic, no atomics), scales back by
//     2^-(sx + sgy) and transposes to (Co, Ci, 27).  grad_bias falls out of the grad_y rows a thread stages.
// R = 8, 12, 16 and 32 are instantiated (a z row shorter than the 16-deep k-step is padded with zeros IN LDS: R = 12, the Frustum
// grids, wastes a quarter of the MFMA work, R = 8 half of it); anything else stays on the fp32-MFMA kernel of conv3d.hip.
#include <algorithm>
#include <stdlib.h>

#include "common.h"
#include "split16.h"

namespace pvcnn {

constexpr int kWgCo = 64, kWgCi = 32;

template <int R>
struct WgradLds {
  static constexpr int RP = (R + 15) / 16 * 16;     // z extent rounded up to whole 16-deep k-steps (the tail stays zero)
  static constexpr int RS = RP + 24;                // fp16 elements per LDS row: [8 pad | RP data | 16 pad]; z = i - 8.  RS * 2 bytes
  static constexpr int ROWB = RS * 2;               //   = 28 / 20 dwords (R = 32 / 16): 16-byte reads of 16 rows hit 64 distinct banks
  static constexpr int XPL = 4 * 3 * kWgCi * ROWB;  // one fp16 plane of the x ring  [slot 4][dx 3][ci 32][row]
  static constexpr int GPL = 2 * kWgCo * ROWB;      // one fp16 plane of grad_y      [buffer 2][co 64][row]
  static constexpr int BYTES = 2 * XPL + 2 * GPL;
};

// PACK (Ci <= 10, the first layer of a network: 9 input channels): the MFMA's 32 columns hold (dx, ci) -- 3 x Ci <= 30 of them --
// instead of 32 input channels of which Ci exist, so a workgroup has 6 units (dy; 32-row co block) instead of 18: a third of the
// MFMAs (the unpacked kernel took 0.18 ms at Ci = 9 against 0.35 ms at Ci = 64, for a seventh of the work).  Waves 0..5 take one
// unit each; all eight waves stage.
// AB (tools/probe, -DPVCNN_ABLATE only; the product instantiates AB = 0): 1 no global loads, 2 no conversion / LDS stores, 4 no MFMAs,
// 8 no partial store, 16 no row barrier -- wrong results, honest time.
template <int R, bool PACK = false, int AB = 0>
__global__ __launch_bounds__(512, 2) void conv3d_wgrad_f16_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                  const uint32_t *__restrict__ x_absmax,
                                                                  const uint32_t *__restrict__ gy_absmax, int B, int Ci, int Co, int P,
                                                                  int citiles, float *__restrict__ part, float *__restrict__ gb_part,
                                                                  int x_seg, long x_words, long gy_words, uint32_t *__restrict__ maxima) {
  using L = WgradLds<R>;
  constexpr int QZ = R / 4, KS = L::RP / 16, ROWB = L::ROWB;
  constexpr int XITEMS = 3 * kWgCi * QZ, GITEMS = kWgCo * QZ;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *xl = lds, *gl = lds + 2 * L::XPL;
  // ZERO ROWS of x (round 4; x_seg == R: x_absmax is an amax buffer with one maximum per z row behind the global one).  The input of a
  // PVConv's first convolution is a voxelised cloud -- exact zeros outside the ~14 % of the grid the block occupies -- and an output
  // row (b, x, y) whose nine neighbouring x rows (x - 1 .. x + 1, y - 1 .. y + 1) are all zero adds zeros to every tap: its MFMAs are
  // skipped (its grad_y row is still staged: grad_bias sums every row).  rowmax[dx][1 + y]: the strip's three x planes, zero halo.
  uint32_t *rowmax = reinterpret_cast<uint32_t *>(lds + L::BYTES);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  int bid = blockIdx.x;
  const int p = bid % P; bid /= P;
  const int cit = bid % citiles, cot = bid / citiles;
  const int ci0 = cit * kWgCi, co0 = cot * kWgCo;
  const size_t RR = (size_t)R * R, S = RR * R;
  // the two global maxima (ABI v12): from the tables when the caller says they are amax buffers -- word [0] may be unwritten --, and
  // handed to the reduce launch through `maxima` (the same two words for every workgroup)
  const uint32_t x_max = amax_table_value(x_absmax, x_words), gy_max = amax_table_value(gy_absmax, gy_words);
  if (blockIdx.x == 0 && threadIdx.x == 0) { maxima[0] = x_max; maxima[1] = gy_max; }
  const float x_scale = exp2_int(scale_shift(x_max)), gy_scale = exp2_int(scale_shift(gy_max));

  for (int e = tid; e < L::BYTES / 4; e += 512) reinterpret_cast<uint32_t *>(lds)[e] = 0u;    // z halos stay zero for good
  __syncthreads();

  // units of this wave: u = wave, wave + 8, wave + 16 (< 18);  u -> (dxy = u % 9, mb = u / 9)   [PACK: unit = wave < 6 -> (dy = u % 3, mb = u / 3)]
  const int nunits = PACK ? (wave < 6 ? 1 : 0) : (wave < 2 ? 3 : 2);
  // PACK: column j of the MFMA = (dx = j / Ci, ci = j % Ci); columns >= 3 Ci read a row of a channel that does not exist (zeros)
  const int pdx = PACK ? (j < 3 * Ci ? j / Ci : 0) : 0, pci = PACK ? (j < 3 * Ci ? j - (j / Ci) * Ci : kWgCi - 1) : 0;
  f32x16 acc[3][3];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][dz][r] = 0.0f;

  // staging roles: x item = (dx, ci, z quad), grad_y item = (co, z quad); a thread keeps its items for the whole kernel
  const int xq0 = tid % QZ, xci0 = (tid / QZ) % kWgCi, xdx0 = tid / (QZ * kWgCi);
  const int t1 = tid + 512;
  const int xq1 = t1 % QZ, xci1 = (t1 / QZ) % kWgCi, xdx1 = t1 / (QZ * kWgCi);
  const bool has_x0 = tid < XITEMS, has_x1 = t1 < XITEMS;
  const int gq = tid % QZ, gco = tid / QZ;
  const bool has_g = tid < GITEMS;
  float gsum = 0.0f;

  auto store_row = [&](unsigned char *plane0, int plane_bytes, int row_byte, int q, const float4 &v, float scale) {
    uint32_t w0[2], w1[2];
    split_pair<2>(v.x * scale, v.y * scale, w0);
    split_pair<2>(v.z * scale, v.w * scale, w1);
    const int off = row_byte + (8 + 4 * q) * 2;
    *reinterpret_cast<uint2 *>(plane0 + off) = make_uint2(w0[0], w1[0]);                 // hi
    *reinterpret_cast<uint2 *>(plane0 + plane_bytes + off) = make_uint2(w0[1], w1[1]);   // lo
  };

  // strips of partition p: pass i takes strip i * P + (p + i * kRot) % P -- rotated from pass to pass, because P is usually a multiple
  // of R, and `strip = p + i * P` then gives a partition the SAME x plane of every cloud: on a voxelised cloud (x planes ~9 .. 22 of
  // 32 occupied) a third of the partitions would own all the live planes and the rest none (measured: 78 % of the rows skipped, 15 %
  // of the time).  Any assignment that covers every strip once is the same sum; this one mixes the planes.
  constexpr int kRot = R / 4 + 1;
  const int nstrips = B * R;
  for (int i0 = 0; i0 * P < nstrips; ++i0) {
    const int strip = i0 * P + (p + i0 * kRot) % P;
    if (strip >= nstrips) continue;
    const int b = strip / R, xo = strip - b * R;
    const float *xb = x + (size_t)b * Ci * S, *gb_ = gy + (size_t)b * Co * S;
    if (tid < 3 * (R + 2)) {                                    // (read from t = 0 on: two barriers behind this store)
      const int dx = tid / (R + 2), yy = tid - dx * (R + 2) - 1, gx = xo + dx - 1;
      const bool in = (unsigned)gx < (unsigned)R && (unsigned)yy < (unsigned)R;
      rowmax[tid] = x_seg > 0 ? (in ? x_absmax[1 + ((size_t)b * R + gx) * R + yy] : 0u) : 1u;
    }
    for (int t = -2; t < R; ++t) {
      // ---- loads for the rows that enter the window: x rows y = t + 2 (three x planes), grad_y row y = t + 1 ----
      const int y2 = t + 2, y1 = t + 1;
      const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      float4 vx0 = zero4, vx1 = zero4, vg = zero4;
      if (!(AB & 1) && y2 < R) {
        const int gx0 = xo + xdx0 - 1, gx1 = xo + xdx1 - 1;
        if (has_x0 && (unsigned)gx0 < (unsigned)R && ci0 + xci0 < Ci)
          vx0 = *reinterpret_cast<const float4 *>(xb + (size_t)(ci0 + xci0) * S + (size_t)gx0 * RR + (size_t)y2 * R + 4 * xq0);
        if (has_x1 && (unsigned)gx1 < (unsigned)R && ci0 + xci1 < Ci)
          vx1 = *reinterpret_cast<const float4 *>(xb + (size_t)(ci0 + xci1) * S + (size_t)gx1 * RR + (size_t)y2 * R + 4 * xq1);
      }
      if (!(AB & 1) && y1 >= 0 && y1 < R && has_g && co0 + gco < Co)
        vg = *reinterpret_cast<const float4 *>(gb_ + (size_t)(co0 + gco) * S + (size_t)xo * RR + (size_t)y1 * R + 4 * gq);

      // ---- multiply output row y = t (unless every x row it reads is zero) ----
      uint32_t live = 0;
      if (t >= 0) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) live |= rowmax[dx * (R + 2) + t + dy];
        live = __builtin_amdgcn_readfirstlane(live);
      }
      if (!(AB & 4) && live != 0) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int z8 = ks * 16 + kh * 8;
          // (the A fragments are read per unit, not once for both 32-row blocks: 16 registers fewer across the unit loop -- the
          // kernel sat at its 256-register cap with 1-3 spilled -- for two more 16-byte LDS reads per k-step and wave)
          const unsigned char *arow = gl + (((t & 1) * kWgCo + j) * ROWB) + (z8 + 8) * 2;
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            if (u < nunits) {
              const int unit = wave + 8 * u, dxy = unit % 9, mb = PACK ? unit / 3 : unit / 9, dx = PACK ? pdx : dxy / 3,
                        dy = PACK ? unit % 3 : dxy - (dxy / 3) * 3;
              const int slot = (t + dy - 1) & 3;
              uint4 bw[2][3];                                    // [hi, lo][dz]
#pragma unroll
              for (int pl = 0; pl < 2; ++pl) {
                const unsigned char *row = xl + pl * L::XPL + ((slot * 3 + dx) * kWgCi + (PACK ? pci : j)) * ROWB + (z8 + 6) * 2;
                const uint32_t d0 = *reinterpret_cast<const uint32_t *>(row);
                const uint4 m = *reinterpret_cast<const uint4 *>(row + 4);
                const uint32_t d5 = *reinterpret_cast<const uint32_t *>(row + 20);
                bw[pl][1] = m;                                   // dz = 1: x[z]     = fp16 elements 8 .. 15 of the window
                bw[pl][0] = make_uint4(__builtin_amdgcn_alignbit(m.x, d0, 16), __builtin_amdgcn_alignbit(m.y, m.x, 16),
                                       __builtin_amdgcn_alignbit(m.z, m.y, 16), __builtin_amdgcn_alignbit(m.w, m.z, 16));   // x[z - 1]
                bw[pl][2] = make_uint4(__builtin_amdgcn_alignbit(m.y, m.x, 16), __builtin_amdgcn_alignbit(m.z, m.y, 16),
                                       __builtin_amdgcn_alignbit(m.w, m.z, 16), __builtin_amdgcn_alignbit(d5, m.w, 16));    // x[z + 1]
              }
              const uint4 ah = *reinterpret_cast<const uint4 *>(arow + mb * 32 * ROWB), al = *reinterpret_cast<const uint4 *>(arow + L::GPL + mb * 32 * ROWB);
#pragma unroll
              for (int dz = 0; dz < 3; ++dz) acc[u][dz] = mfma16<2>(al, bw[0][dz], acc[u][dz]);     // lo x hi
#pragma unroll
              for (int dz = 0; dz < 3; ++dz) acc[u][dz] = mfma16<2>(ah, bw[1][dz], acc[u][dz]);     // hi x lo
#pragma unroll
              for (int dz = 0; dz < 3; ++dz) acc[u][dz] = mfma16<2>(ah, bw[0][dz], acc[u][dz]);     // hi x hi
            }
          }
        }
      }

      // ---- convert and store the new rows (out-of-range rows are stored as zeros: they are the y / x halo) ----
      if (!(AB & 2)) {
      if (t == -2) {                                            // row y = -1 of the new strip lives in slot 3
        if (has_x0) store_row(xl, L::XPL, ((3 * 3 + xdx0) * kWgCi + xci0) * ROWB, xq0, zero4, 1.0f);
        if (has_x1) store_row(xl, L::XPL, ((3 * 3 + xdx1) * kWgCi + xci1) * ROWB, xq1, zero4, 1.0f);
      }
      if (has_x0) store_row(xl, L::XPL, (((y2 & 3) * 3 + xdx0) * kWgCi + xci0) * ROWB, xq0, vx0, x_scale);
      if (has_x1) store_row(xl, L::XPL, (((y2 & 3) * 3 + xdx1) * kWgCi + xci1) * ROWB, xq1, vx1, x_scale);
      if (has_g && y1 >= 0) {
        store_row(gl, L::GPL, ((y1 & 1) * kWgCo + gco) * ROWB, gq, vg, gy_scale);
        gsum += (vg.x + vg.y) + (vg.z + vg.w);
      }
      }
      if (!(AB & 16)) __syncthreads();
    }
  }

  // ---- epilogue: part[p][tap][co][ci] (CoP x CiP padded block grid), lanes along ci ----
  const int CoP = (int)gridDim.x / (P * citiles) * kWgCo, CiP = citiles * kWgCi;
  float *pp = part + (size_t)p * 27 * CoP * CiP;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    if (u < nunits) {
      const int unit = wave + 8 * u, dxy = PACK ? pdx * 3 + unit % 3 : unit % 9, mb = PACK ? unit / 3 : unit / 9;
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        const int tap = dxy * 3 + dz;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (!(AB & 8) && (!PACK || j < 3 * Ci)) pp[((size_t)tap * CoP + co) * CiP + ci0 + (PACK ? pci : j)] = acc[u][dz][r];
        }
      }
    }
  }
  if (gb_part != nullptr && cit == 0) {                         // grad_bias partial: the QZ quads of a channel, fixed order
    __syncthreads();
    float *red = reinterpret_cast<float *>(lds);
    if (has_g) red[tid] = gsum;
    __syncthreads();
    if (tid < kWgCo) {
      float s = 0.0f;
#pragma unroll
      for (int q = 0; q < QZ; ++q) s += red[tid * QZ + q];
      gb_part[(size_t)p * CoP + co0 + tid] = s;
    }
  }
}

// ---- round 6: the same split-K block with "pinned pieces" (pp) -- what was measured, what was kept -------------------------------
// The ablation builds of the kernel above (tools/wgrad_only.py, tools/calls_r06/README.md calls 16-24, profiles/r06_wgrad_*.jsonl):
// at 128 -> 128 @ 16^3 a launch takes 190 us, without its MFMAs 78 us, without its loads / conversion 126 us -- close to the SUM of
// the parts, and the matrix pipe is busy in 0.27-0.49 of the cycles.  This kernel keeps the block (64 co x 32 ci x 27 taps = 54
// accumulator tiles per workgroup, the LDS ring, the partial layout, split-K over 256 workgroups, the reduce launch) and changes:
//   * the rows of step s + 1 are requested during step s (a register set in flight, two steps per loop trip with the sets exchanged),
//     by buffer loads whose offset is out of range for a row that does not exist -- a request has no branch, the compiler can count
//     the loads in flight, and no vmcnt(0) stands in the matrix phase;
//   * every thread stages the same number of items (the surplus ones stage another thread's item again): no divergent branch;
//   * the zero-row test is ONE scalar bit mask per strip (a ballot over the amax table) instead of nine LDS reads per step;
//   * what a request needs of its strip (descriptors, row offsets) is computed once per strip, not once per step;
//   * the z shift of a tap is applied to the grad_y fragment -- once per k-step for all of a wave's tiles, which share one 32-row co
//     block -- instead of the x fragment of every (dx, dy) unit: a third of the funnel shifts and LDS reads (the same products
//     grouped into other k-steps: fp32-rounding-level differences to the kernel above, tests/test_gpu_wgrad_pp.py);
//   * the 27 tiles of a co block are dealt 7, 7, 7, 6 to four waves (14, 14, 14, 12 per SIMD; before: 15, 15, 12, 12);
//   * in a live step every MFMA is followed by one pinned piece of the step's other work (live_step below).
// What it bought: 4-7 % per launch, 0.01-0.035 ms of the 6.2 ms step -- and the finding that NONE of the schedules tried (opposite
// phases of a SIMD's two waves, hand-prefetched fragments, half the LDS reads, a third fewer vector-ALU instructions, the pinned
// interleave) moves the launch time by more than +-5 %: with both operands converted in the kernel (x is staged 3 x cotiles times,
// grad_y citiles times) the launch runs at ~2.0 GHz with the matrix pipe half busy -- the clock the chip grants for this mix of
// MFMA, vector-ALU and LDS work -- and what is saved in one unit is granted to no other.  Fewer conversions per MFMA (a larger
// block: 108 tiles need more registers than a workgroup has) or fewer MFMAs are what would help; neither is in this kernel.
template <int R, bool PACK = false, int AB = 0>
__global__ __launch_bounds__(512, 2) void conv3d_wgrad_f16_pp_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                     const uint32_t *__restrict__ x_absmax,
                                                                     const uint32_t *__restrict__ gy_absmax, int B, int Ci, int Co, int P,
                                                                     int citiles, float *__restrict__ part, float *__restrict__ gb_part,
                                                                     int x_seg, long x_words, long gy_words, uint32_t *__restrict__ maxima) {
  using L = WgradLds<R>;
  constexpr int QZ = R / 4, KS = L::RP / 16, ROWB = L::ROWB;
  constexpr int XITEMS = 3 * kWgCi * QZ, GITEMS = kWgCo * QZ;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char *xl = lds, *gl = lds + 2 * L::XPL;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  int bid = blockIdx.x;
  const int p = bid % P; bid /= P;
  const int cit = bid % citiles, cot = bid / citiles;
  const int ci0 = cit * kWgCi, co0 = cot * kWgCo;
  const size_t RR = (size_t)R * R, S = RR * R;
  // the two global maxima (ABI v12): from the tables when the caller says they are amax buffers -- word [0] may be unwritten --, and
  // handed to the reduce launch through `maxima` (the same two words for every workgroup)
  const uint32_t x_max = amax_table_value(x_absmax, x_words), gy_max = amax_table_value(gy_absmax, gy_words);
  if (blockIdx.x == 0 && threadIdx.x == 0) { maxima[0] = x_max; maxima[1] = gy_max; }
  const float x_scale = exp2_int(scale_shift(x_max)), gy_scale = exp2_int(scale_shift(gy_max));

  for (int e = tid; e < L::BYTES / 4; e += 512) reinterpret_cast<uint32_t *>(lds)[e] = 0u;    // z halos stay zero for good
  __syncthreads();

  // tiles of this wave: waves 0..3 hold the 27 tiles of the first 32 grad_y rows (co block mb = 0), waves 4..7 those of the second -- a
  // wave's tiles share ONE A fragment per k-step.  Within a block, wave q = w & 3 holds the whole units (three dz tiles of one (dx, dy))
  // 2 q and 2 q + 1 and, q < 3, tile dz = q of unit 8: 7, 7, 7, 6 tiles; a SIMD's two waves (w, w + 4) 14, 14, 14, 12.
  // PACK: unit = wave < 6 (dy = unit % 3, mb = unit / 3), as in the kernel above.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wq = wave_u & 3, wmb = PACK ? wave_u / 3 : wave_u >> 2;
  const int nfull = PACK ? (wave_u < 6 ? 1 : 0) : 2;
  const bool has_part = !PACK && wq < 3;
  const int fu0 = PACK ? wave_u % 3 : 2 * wq;                   // (unit numbers within the co block: (dx, dy) = unit / 3, unit % 3)
  const int pu = 8, pdz = wq;
  const bool convert_first = (AB & 64) && wave_u >= 4;
  const int pdx = PACK ? (j < 3 * Ci ? j / Ci : 0) : 0, pci = PACK ? (j < 3 * Ci ? j - (j / Ci) * Ci : kWgCi - 1) : 0;
  f32x16 acc[2][3], accp;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    accp[r] = 0.0f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) acc[u][dz][r] = 0.0f;
  }

  // staging roles: x item = (dx, ci, z quad), grad_y item = (co, z quad); a thread keeps its items for the whole kernel.  EVERY thread
  // has an x item, a grad_y item and (R = 32: 768 x items) a second x item: beyond the real ones a thread stages another thread's
  // item again -- the same bytes to the same place -- so that the step has no divergent branch (behind one, the compiler sinks the
  // conversion into the branch and the pinned pieces of the live step are gone); only grad_bias must count an item once (has_g).
  const int e0 = tid % XITEMS, e1 = (tid + 512) % XITEMS, eg = tid % GITEMS;
  const int xq0 = e0 % QZ, xci0 = (e0 / QZ) % kWgCi, xdx0 = e0 / (QZ * kWgCi);
  const int xq1 = e1 % QZ, xci1 = (e1 / QZ) % kWgCi, xdx1 = e1 / (QZ * kWgCi);
  constexpr bool has_x0 = true, has_x1 = XITEMS > 512;
  const int gq = eg % QZ, gco = eg / QZ;
  const bool has_g = tid < GITEMS;
  float gsum = 0.0f;

  auto store_row = [&](unsigned char *plane0, int plane_bytes, int row_byte, int q, const float4 &v, float scale) {
    uint32_t w0[2], w1[2];
    split_pair<2>(v.x * scale, v.y * scale, w0);
    split_pair<2>(v.z * scale, v.w * scale, w1);
    const int off = row_byte + (8 + 4 * q) * 2;
    *reinterpret_cast<uint2 *>(plane0 + off) = make_uint2(w0[0], w1[0]);                 // hi
    *reinterpret_cast<uint2 *>(plane0 + plane_bytes + off) = make_uint2(w0[1], w1[1]);   // lo
  };
  const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  // the rows that ENTER the window at step (strip, t): x rows y = t + 2 of the strip's three x planes, grad_y row y = t + 1.  Buffer
  // loads (one descriptor per cloud: Ci S floats) with the offset of a row that does not exist pushed out of range -- it arrives as
  // zeros -- so that a request is three loads and NO branch: behind divergent branches the compiler cannot count the loads in
  // flight and waits for all of them (vmcnt(0)) where it needs the previous step's.
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  auto descriptor = [](const void *base, uint32_t bytes) {
    const uintptr_t q = reinterpret_cast<uintptr_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)q), hi = __builtin_amdgcn_readfirstlane((uint32_t)(q >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
  };
  const uint32_t kOut = 0xfffffff0u;
  const uint32_t xoff0 = ci0 + xci0 < Ci ? (uint32_t)(((size_t)(ci0 + xci0) * S + 4 * xq0) * 4) : kOut;
  const uint32_t xoff1 = has_x1 && ci0 + xci1 < Ci ? (uint32_t)(((size_t)(ci0 + xci1) * S + 4 * xq1) * 4) : kOut;
  const uint32_t goff = co0 + gco < Co ? (uint32_t)(((size_t)(co0 + gco) * S + 4 * gq) * 4) : kOut;
  auto as_float4 = [](const u32x4 &r) { return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)); };
  // What a request needs of its strip -- the two descriptors and the three row offsets at y = 0 -- is computed ONCE per strip (for the
  // strip after the current one, at its first step) and kept: a step is ~1500 matrix-pipe cycles at R = 16, and the phase clocks put
  // a quarter of it into ~150 scalar instructions that recomputed strip numbers (a modulo by P), 64-bit bases and descriptors.
  struct StripRefs { __amdgpu_buffer_rsrc_t xr, gr; uint32_t o0, o1, og; };
  auto refs_of = [&](int strip) {
    StripRefs f;
    const int st = max(strip, 0), b = st / R, xo = st - b * R;
    f.xr = descriptor(x + (size_t)b * Ci * S, (uint32_t)((size_t)Ci * S * 4));
    f.gr = descriptor(gy + (size_t)b * Co * S, (uint32_t)((size_t)Co * S * 4));
    const int gx0 = xo + xdx0 - 1, gx1 = xo + xdx1 - 1;
    f.o0 = strip >= 0 && (unsigned)gx0 < (unsigned)R && xoff0 != kOut ? xoff0 + (uint32_t)(gx0 * R * R * 4) : kOut;
    f.o1 = strip >= 0 && (unsigned)gx1 < (unsigned)R && xoff1 != kOut ? xoff1 + (uint32_t)(gx1 * R * R * 4) : kOut;
    f.og = strip >= 0 && goff != kOut ? goff + (uint32_t)(xo * R * R * 4) : kOut;
    return f;
  };
  auto request = [&](const StripRefs &f, int t, float4 &vx0, float4 &vx1, float4 &vg) {       // the rows that enter at step t of f's strip
    if (AB & 1) { vx0 = zero4; vx1 = zero4; vg = zero4; return; }
    const int y2 = t + 2, y1 = t + 1;
    const uint32_t o0 = y2 < R && f.o0 != kOut ? f.o0 + (uint32_t)(y2 * R * 4) : kOut;
    const uint32_t o1 = y2 < R && f.o1 != kOut ? f.o1 + (uint32_t)(y2 * R * 4) : kOut;
    const uint32_t og = y1 >= 0 && y1 < R && f.og != kOut ? f.og + (uint32_t)(y1 * R * 4) : kOut;
    vx0 = as_float4(__builtin_amdgcn_raw_buffer_load_b128(f.xr, o0, 0, 0));
    vx1 = as_float4(__builtin_amdgcn_raw_buffer_load_b128(f.xr, o1, 0, 0));
    vg = as_float4(__builtin_amdgcn_raw_buffer_load_b128(f.gr, og, 0, 0));
  };
  // ... and their conversion into the ring (out-of-range rows arrive as zeros: they are the y / x halo)
  auto convert = [&](int t, const float4 &vx0, const float4 &vx1, const float4 &vg) {
    if (AB & 2) return;
    const int y2 = t + 2, y1 = t + 1;
    if (t == -2) {                                              // row y = -1 of the new strip lives in slot 3
      if (has_x0) store_row(xl, L::XPL, ((3 * 3 + xdx0) * kWgCi + xci0) * ROWB, xq0, zero4, 1.0f);
      if (has_x1) store_row(xl, L::XPL, ((3 * 3 + xdx1) * kWgCi + xci1) * ROWB, xq1, zero4, 1.0f);
    }
    if (has_x0) store_row(xl, L::XPL, (((y2 & 3) * 3 + xdx0) * kWgCi + xci0) * ROWB, xq0, vx0, x_scale);
    if (has_x1) store_row(xl, L::XPL, (((y2 & 3) * 3 + xdx1) * kWgCi + xci1) * ROWB, xq1, vx1, x_scale);
    if (y1 >= 0) {
      store_row(gl, L::GPL, ((y1 & 1) * kWgCo + gco) * ROWB, gq, vg, gy_scale);
      const float rs = (vg.x + vg.y) + (vg.z + vg.w);
      gsum += has_g ? rs : 0.0f;
    }
  };
  // The matrix phase of a step.  grad_w[tap dz] = sum_z grad_y[z] x[z + dz - 1] = sum_z' grad_y[z' - dz + 1] x[z']: the shift along z is
  // applied to the A operand (grad_y: ONE fragment per k-step for all of a wave's tiles -- a 16-byte LDS read plus the dword on
  // either side, two funnel shifts) instead of the B operand (x: one window per (dx, dy) unit, as the kernel above does it); a unit
  // is then two aligned 16-byte reads and nine MFMAs with no vector-ALU work.  Why it matters: the phase clocks and the variants of
  // tools/calls_r06 (fewer LDS reads, hand-prefetched fragments, opposite phases of a SIMD's two waves: all within +-5 %) say the
  // step is bound by the NUMBER of vector-ALU instructions -- ~8 per MFMA, and a SIMD issues 8 per 32-cycle MFMA, the MFMA itself
  // included.  The sums are the same products grouped into other 16-deep k-steps: same accuracy class, not the same bits as the
  // kernel above.
  auto a_shifts = [&](int t, int ks, uint4 (&as)[2][3]) {       // [hi, lo][dz]
    const int z8 = ks * 16 + kh * 8;
    const unsigned char *arow = gl + (((t & 1) * kWgCo + wmb * 32 + j) * ROWB) + (z8 + 6) * 2;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const unsigned char *row = arow + pl * L::GPL;
      const uint32_t d0 = *reinterpret_cast<const uint32_t *>(row);
      const uint4 m = *reinterpret_cast<const uint4 *>(row + 4);
      const uint32_t d5 = *reinterpret_cast<const uint32_t *>(row + 20);
      as[pl][1] = m;                                                                                                    // dz = 1: gy[z']
      as[pl][0] = make_uint4(__builtin_amdgcn_alignbit(m.y, m.x, 16), __builtin_amdgcn_alignbit(m.z, m.y, 16),
                             __builtin_amdgcn_alignbit(m.w, m.z, 16), __builtin_amdgcn_alignbit(d5, m.w, 16));          // dz = 0: gy[z' + 1]
      as[pl][2] = make_uint4(__builtin_amdgcn_alignbit(m.x, d0, 16), __builtin_amdgcn_alignbit(m.y, m.x, 16),
                             __builtin_amdgcn_alignbit(m.z, m.y, 16), __builtin_amdgcn_alignbit(m.w, m.z, 16));         // dz = 2: gy[z' - 1]
    }
  };
  auto b_fragment = [&](int t, int ks, int unit, uint4 &bh, uint4 &bl) {
    const int z8 = ks * 16 + kh * 8;
    const int dx = PACK ? pdx : unit / 3, dy = PACK ? unit : unit - (unit / 3) * 3;
    const unsigned char *row = xl + ((((t + dy - 1) & 3) * 3 + dx) * kWgCi + (PACK ? pci : j)) * ROWB + (z8 + 8) * 2;
    bh = *reinterpret_cast<const uint4 *>(row);
    bl = *reinterpret_cast<const uint4 *>(row + L::XPL);
  };
  auto multiply = [&](int t) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 as[2][3];
      a_shifts(t, ks, as);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u < nfull) {
          uint4 bh, bl;
          b_fragment(t, ks, fu0 + u, bh, bl);
#pragma unroll
          for (int dz = 0; dz < 3; ++dz) acc[u][dz] = mfma16<2>(as[1][dz], bh, acc[u][dz]);     // lo x hi
#pragma unroll
          for (int dz = 0; dz < 3; ++dz) acc[u][dz] = mfma16<2>(as[0][dz], bl, acc[u][dz]);     // hi x lo
#pragma unroll
          for (int dz = 0; dz < 3; ++dz) acc[u][dz] = mfma16<2>(as[0][dz], bh, acc[u][dz]);     // hi x hi
        }
      }
      if (has_part) {
        uint4 bh, bl;
        b_fragment(t, ks, pu, bh, bl);
        const uint4 ah = pdz == 0 ? as[0][0] : pdz == 1 ? as[0][1] : as[0][2], al = pdz == 0 ? as[1][0] : pdz == 1 ? as[1][1] : as[1][2];
        accp = mfma16<2>(al, bh, accp);
        accp = mfma16<2>(ah, bl, accp);
        accp = mfma16<2>(ah, bh, accp);
      }
    }
  };

  // ---- the live step, interleaved by hand (not PACK) -------------------------------------------------------------------------------
  // The phase clocks of the form above: matrix phase 0.61 us of a 1.24 us step at 128 -> 128 @ 16^3 (= the 42 MFMAs of the SIMD: the
  // pipe is full while it lasts), request + conversion + barrier the other half, NOTHING of it under the MFMAs -- the compiler issues
  // a unit's nine MFMAs back to back and the vector-ALU work around them; opposite phases of a SIMD's two waves do not help (a wave's
  // vector-ALU work issues slowly against the other wave's back-to-back MFMAs).  What does overlap is vector-ALU work issued by
  // the SAME wave right behind an MFMA (~7 issue slots before the pipe takes the next one): so every MFMA here is followed by one
  // pinned PIECE of the step's other work -- a quarter of an item's conversion (scale + hi | back-conversion | residual + lo |
  // address + two LDS stores), the request of the next step's rows, the LDS reads of the next fragments -- and a scheduling fence.
  struct Conv { f32x2 a, b; uint32_t h0, h1, l0, l1; };
  auto cpiece = [&](int k, Conv &c, const float4 &v, float scale, unsigned char *plane0, int plane_bytes, uint32_t off) {
    if (k == 0) {
      c.a = f32x2{v.x, v.y} * scale; c.b = f32x2{v.z, v.w} * scale;
      c.h0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(c.a, f16x2)); c.h1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(c.b, f16x2));
    } else if (k == 1) {
      c.a = c.a - __builtin_convertvector(__builtin_bit_cast(f16x2, c.h0), f32x2);        // exact
      c.b = c.b - __builtin_convertvector(__builtin_bit_cast(f16x2, c.h1), f32x2);
    } else if (k == 2) {
      c.l0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(c.a, f16x2)); c.l1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(c.b, f16x2));
    } else {
      *reinterpret_cast<uint2 *>(plane0 + off) = make_uint2(c.h0, c.h1);
      *reinterpret_cast<uint2 *>(plane0 + plane_bytes + off) = make_uint2(c.l0, c.l1);
    }
  };
  // per-thread LDS offsets of the staged items without the ring slot (added per step: one scalar)
  const uint32_t xst0 = (uint32_t)((xdx0 * kWgCi + xci0) * ROWB + (8 + 4 * xq0) * 2), xst1 = (uint32_t)((xdx1 * kWgCi + xci1) * ROWB + (8 + 4 * xq1) * 2);
  const uint32_t gst = (uint32_t)(gco * ROWB + (8 + 4 * gq) * 2);
  constexpr bool HAS_X1 = XITEMS > 512;
  auto live_step = [&](int t, const StripRefs &rf, int rt, const float4 &cx0, const float4 &cx1, const float4 &cg, float4 &nx0, float4 &nx1, float4 &ng) {
    const uint32_t xslot = (uint32_t)(((t + 2) & 3) * 3 * kWgCi * ROWB), gslot = (uint32_t)(((t + 1) & 1) * kWgCo * ROWB);
    Conv c0, c1, c2;
    // pieces of a k-step's 18 whole-unit slots (the shared tile's three slots carry none: not every wave has them)
    auto piece = [&](int ks, int slot) {
      if (ks == 0) {
        if (slot < 4) cpiece(slot, c0, cx0, x_scale, xl, L::XPL, xst0 + xslot);
        else if (slot < 8) cpiece(slot - 4, c2, cg, gy_scale, gl, L::GPL, gst + gslot);
        else if (slot == 8) { const float rs = (cg.x + cg.y) + (cg.z + cg.w); gsum += has_g ? rs : 0.0f; }
        else if (HAS_X1 && slot < 13) cpiece(slot - 9, c1, cx1, x_scale, xl, L::XPL, xst1 + xslot);
        else if (slot == 13) request(rf, rt, nx0, nx1, ng);
      }
    };
    uint4 as[2][3], b0h, b0l, b1h, b1l, bph = make_uint4(0, 0, 0, 0), bpl = bph;
    a_shifts(t, 0, as);
    b_fragment(t, 0, fu0, b0h, b0l);
    b_fragment(t, 0, fu0 + 1, b1h, b1l);
    if (has_part) b_fragment(t, 0, pu, bph, bpl);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks > 0) {                                             // (the second k-step of an R = 32 row: its fragments, not overlapped)
        a_shifts(t, ks, as);
        b_fragment(t, ks, fu0, b0h, b0l);
        b_fragment(t, ks, fu0 + 1, b1h, b1l);
        if (has_part) b_fragment(t, ks, pu, bph, bpl);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint4 &bh = u == 0 ? b0h : b1h, &bl = u == 0 ? b0l : b1l;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const int dz = q % 3, prod = q / 3;
          acc[u][dz] = mfma16<2>(prod == 0 ? as[1][dz] : as[0][dz], prod == 1 ? bl : bh, acc[u][dz]);      // lo x hi, hi x lo, hi x hi
          piece(ks, u * 9 + q);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (has_part) {
        const uint4 ah = pdz == 0 ? as[0][0] : pdz == 1 ? as[0][1] : as[0][2], al = pdz == 0 ? as[1][0] : pdz == 1 ? as[1][1] : as[1][2];
        accp = mfma16<2>(al, bph, accp);
        accp = mfma16<2>(ah, bpl, accp);
        accp = mfma16<2>(ah, bph, accp);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  constexpr int kRot = R / 4 + 1;
  const int nstrips = B * R;
  auto strip_of = [&](int i0) { const int st = i0 * P + (p + i0 * kRot) % P; return (i0 * P < nstrips && st < nstrips) ? st : -1; };
  // ONE flat loop over the steps (strip, t = -2 .. R - 1) of this partition's strips, two steps per trip with the two register sets
  // exchanged (no copies; R + 2 is even): a loop whose first trips are special invites the compiler to peel them, and the peeled
  // copies cost it the accumulators (spilled around them, reloaded -- with a vmcnt wait inside the matrix stream -- in the loop).
  int npass = 0;
  while (strip_of(npass) >= 0) ++npass;                        // (only the last pass can be without a strip)
  float4 ax0, ax1, ag, bx0, bx1, bg;                           // the rows of this step (in registers since the last one) / of the next
  int i0 = 0, t = -2, strip = strip_of(0);
  StripRefs cur = refs_of(strip), nxt = cur;
  request(cur, -2, ax0, ax1, ag);
  unsigned long long live_rows = ~0ull;
  PVCNN_PROBE_BEGIN();
  const int pslot = convert_first ? 8 : 0;
  (void)pslot;
  auto step = [&](const float4 &cx0, const float4 &cx1, const float4 &cg, float4 &nx0, float4 &nx1, float4 &ng) {
    if (t == -2) {
      // ZERO ROWS (see the kernel above): which output rows y of this strip have a non-zero x row among their nine neighbours -- one
      // bit per y, in a scalar register for the whole strip.  Lane y ORs its nine maxima straight from the amax buffer and the wave
      // votes (every wave for itself: 9 loads per lane and strip).  (The kernel above keeps the strip's maxima in LDS and reads nine
      // of them at the top of EVERY step: a handful of LDS reads that queue behind the other waves' fragment reads -- the phase
      // clocks put a quarter of a step there.)
      live_rows = ~0ull;
      if (x_seg > 0) {
        const int b = strip / R, xo = strip - b * R;
        uint32_t m = 0;
        if (lane < R) {
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
              const int gx = xo + dx, yy = lane + dy;
              if ((unsigned)gx < (unsigned)R && (unsigned)yy < (unsigned)R) m |= x_absmax[1 + ((size_t)b * R + gx) * R + yy];
            }
        }
        live_rows = __ballot(m != 0);
      }
      nxt = refs_of(strip_of(i0 + 1));
    }
    const bool last = t + 1 == R;
    const bool live = t >= 0 && ((live_rows >> t) & 1ull) != 0;
    if (!PACK && !(AB & 128) && t >= 0 && ((live_rows >> t) & 1ull) != 0) {      // (= live; spelled out: with the one flag the compiler keeps two copies of the accumulators)
      PVCNN_PROBE(0);
      StripRefs rf = cur;
      if (last) rf = nxt;
      live_step(t, rf, last ? -2 : t + 1, cx0, cx1, cg, nx0, nx1, ng);
      PVCNN_PROBE(2);
    } else {
      if (last) request(nxt, -2, nx0, nx1, ng);
      else request(cur, t + 1, nx0, nx1, ng);
      PVCNN_PROBE(0);
      if (convert_first) convert(t, cx0, cx1, cg);
      PVCNN_PROBE(1);
      if (!(AB & 4) && live) multiply(t);                       // (not PACK: never -- a live step took the branch above)
      PVCNN_PROBE(2);
      if (!convert_first) convert(t, cx0, cx1, cg);
      PVCNN_PROBE(3);
    }
    if (!(AB & 16)) __syncthreads();
    PVCNN_PROBE(4);
    if (last) { ++i0; t = -2; strip = strip_of(i0); cur = nxt; } else ++t;
  };
#pragma nounroll
  for (int q = 0; q < npass * (R + 2); q += 2) {
    step(ax0, ax1, ag, bx0, bx1, bg);
    step(bx0, bx1, bg, ax0, ax1, ag);
  }
#ifdef PVCNN_PHASE_PROBE
  if (lane == 0 && phase_probe_buf != nullptr) {                // slots 0..4: waves 0-3, 8..12: waves 4-7
    for (int k = 0; k < 5; ++k) atomicAdd(phase_probe_buf + pslot + k, (unsigned long long)probe_acc_[k]);
    atomicAdd(phase_probe_buf + 31, 1ull);
  }
#endif

  // ---- epilogue: part[p][tap][co][ci] (CoP x CiP padded block grid), lanes along ci ----
  const int CoP = (int)gridDim.x / (P * citiles) * kWgCo, CiP = citiles * kWgCi;
  float *pp = part + (size_t)p * 27 * CoP * CiP;
  auto put = [&](const f32x16 &a, int tap, int mb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (!PACK || j < 3 * Ci) pp[((size_t)tap * CoP + co) * CiP + ci0 + (PACK ? pci : j)] = a[r];
    }
  };
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    if (u < nfull) {
      const int unit = fu0 + u, dxy = PACK ? pdx * 3 + unit : unit;
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) put(acc[u][dz], dxy * 3 + dz, wmb);
    }
  }
  if (has_part) put(accp, pu * 3 + pdz, wmb);
  if (gb_part != nullptr && cit == 0) {                         // grad_bias partial: the QZ quads of a channel, fixed order
    __syncthreads();
    float *red = reinterpret_cast<float *>(lds);
    if (has_g) red[tid] = gsum;
    __syncthreads();
    if (tid < kWgCo) {
      float s = 0.0f;
#pragma unroll
      for (int q = 0; q < QZ; ++q) s += red[tid * QZ + q];
      gb_part[(size_t)p * CoP + co0 + tid] = s;
    }
  }
}

// gw[co][ci][tap] = 2^-(sx + sgy) * sum_p part[p][tap][co][ci];  gb[co] = sum_p gb_part[p][co]
// One workgroup per (co, 32-channel ci block): it owns the 27 x 32 = 864 CONTIGUOUS floats gw[co][ci0 .. ci0+31][0 .. 26].
// 864 of its 1024 threads = 216 float4 columns (tap, ci quad) x 4 partition slices; a thread sums its slice's partitions in a
// fixed order with eight 16-byte loads in flight (the first form gave one thread ALL partitions of one element behind four 4-byte
// loads: 1.8 MB in flight on the whole chip, 41.5 us for 57 MB), the four slices meet in LDS in a fixed order, and the block is
// written back transposed as one coalesced run.  Deterministic; no atomics.  Wave 14 sums the grad_bias partials of `co`.
constexpr int kRedSlices = 4, kRedCols = 27 * (kWgCi / 4);     // 216 float4 columns

__global__ __launch_bounds__(1024) void conv3d_wgrad_f16_reduce_kernel(const float *__restrict__ part, const float *__restrict__ gb_part,
                                                                       const uint32_t *__restrict__ maxima, int P, int CoP, int CiP,
                                                                       int Co, int Ci, float *__restrict__ gw, float *__restrict__ gb) {
  __shared__ __attribute__((aligned(16))) float red[kRedSlices][27 * kWgCi];
  const int citiles = CiP / kWgCi;
  const int co = blockIdx.x / citiles, ci0 = (blockIdx.x - co * citiles) * kWgCi;
  const int tid = threadIdx.x;
  const size_t block = (size_t)27 * CoP * CiP;
  if (tid < kRedSlices * kRedCols) {
    const int col = tid % kRedCols, sl = tid / kRedCols;
    const int tap = col / (kWgCi / 4), q = col - tap * (kWgCi / 4);
    const float *src = part + ((size_t)tap * CoP + co) * CiP + ci0 + 4 * q;
    const int per = ceil_div(P, kRedSlices), p0 = sl * per, p1 = min(P, p0 + per);
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [](float4 &a, const float4 &v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    int p = p0;
    for (; p + 7 < p1; p += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(src + (size_t)(p + u) * block);
#pragma unroll
      for (int u = 0; u < 8; ++u) add(acc[u & 3], v[u]);
    }
    for (; p < p1; ++p) add(acc[0], *reinterpret_cast<const float4 *>(src + (size_t)p * block));
    add(acc[0], acc[1]); add(acc[2], acc[3]); add(acc[0], acc[2]);
    *reinterpret_cast<float4 *>(&red[sl][tap * kWgCi + 4 * q]) = acc[0];
  } else if (gb != nullptr && ci0 == 0 && co < Co && tid >= 896 && tid < 960) {
    const int lane = tid - 896;
    float s = 0.0f;
    for (int q = lane; q < P; q += 64) s += gb_part[(size_t)q * CoP + co];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);    // fixed butterfly: deterministic
    if (lane == 0) gb[co] = s;
  }
  __syncthreads();
  if (tid < 27 * kWgCi) {
    const int ci_l = tid / 27, tap = tid - ci_l * 27;
    const int e = tap * kWgCi + ci_l;
    const float s = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    if (co < Co && ci0 + ci_l < Ci)
      gw[((size_t)co * Ci + ci0 + ci_l) * 27 + tap] = s * exp2_int(-scale_shift(maxima[0])) * exp2_int(-scale_shift(maxima[1]));
  }
}

struct WgradPlan { int cotiles, citiles, P; size_t part_floats, gb_floats; };

static WgradPlan wgrad_f16_plan(int B, int Ci, int Co, int R) {
  WgradPlan w;
  w.cotiles = ceil_div(Co, kWgCo);
  w.citiles = ceil_div(Ci, kWgCi);
  const int blocks = w.cotiles * w.citiles;
  w.P = std::max(1, std::min(B * R, 256 / blocks));             // one workgroup per CU (112 KiB of LDS each)
  w.part_floats = (size_t)w.P * 27 * w.cotiles * kWgCo * w.citiles * kWgCi;
  w.gb_floats = (size_t)w.P * w.cotiles * kWgCo + 4;            // + the two global maxima the main launch hands to the reduce launch
  return w;
}

template <int R>
static int launch_wgrad_f16(const float *x, const float *gy, const uint32_t *xa, const uint32_t *ga, int B, int Ci, int Co, float *gw,
                            float *gb, float *ws, hipStream_t s, int x_seg, int gy_seg) {
  const WgradPlan w = wgrad_f16_plan(B, Ci, Co, R);
  float *part = ws, *gb_part = ws + w.part_floats;
  uint32_t *maxima = reinterpret_cast<uint32_t *>(gb_part + w.gb_floats - 4);
  const long x_words = x_seg > 0 ? (long)B * R * R : 0L, gy_words = gy_seg > 0 ? (long)B * R * R : 0L;
  const bool pack = 3 * Ci <= kWgCi && (R == 32 || R == 16);     // (instantiated for the grids a network's first layer has)
  static const bool pingpong = [] { const char *e = getenv("PVCNN_WGRAD_PP"); return !(e && e[0] == '0'); }();   // 0: the kernel of rounds 3-5
  auto k = pack ? conv3d_wgrad_f16_kernel<R, (R == 32 || R == 16)> : conv3d_wgrad_f16_kernel<R, false>;
  const bool fits = (size_t)std::max(Ci, Co) * R * R * R * 4 < ((size_t)1 << 32) - 64;      // one buffer descriptor per cloud
  if (pingpong && fits) k = pack ? conv3d_wgrad_f16_pp_kernel<R, (R == 32 || R == 16)> : conv3d_wgrad_f16_pp_kernel<R, false>;
#ifdef PVCNN_ABLATE
  if (const char *e = getenv("PVCNN_WGRAD_ABLATE")) {
    switch (atoi(e)) {
      case 1: k = conv3d_wgrad_f16_kernel<R, false, 1>; break;
      case 2: k = conv3d_wgrad_f16_kernel<R, false, 2>; break;
      case 3: k = conv3d_wgrad_f16_kernel<R, false, 3>; break;
      case 4: k = conv3d_wgrad_f16_kernel<R, false, 4>; break;
      case 8: k = conv3d_wgrad_f16_kernel<R, false, 8>; break;
      case 16: k = conv3d_wgrad_f16_kernel<R, false, 16>; break;
      case 27: k = conv3d_wgrad_f16_kernel<R, false, 27>; break;
      case 31: k = conv3d_wgrad_f16_kernel<R, false, 31>; break;
      default: break;
    }
    if (pingpong && fits) switch (atoi(e)) {
      case 1: k = conv3d_wgrad_f16_pp_kernel<R, false, 1>; break;
      case 2: k = conv3d_wgrad_f16_pp_kernel<R, false, 2>; break;
      case 4: k = conv3d_wgrad_f16_pp_kernel<R, false, 4>; break;
      case 16: k = conv3d_wgrad_f16_pp_kernel<R, false, 16>; break;
      case 32: k = conv3d_wgrad_f16_pp_kernel<R, false, 32>; break;
      case 64: k = conv3d_wgrad_f16_pp_kernel<R, false, 64>; break;
      case 96: k = conv3d_wgrad_f16_pp_kernel<R, false, 96>; break;
      case 100: k = conv3d_wgrad_f16_pp_kernel<R, false, 100>; break;
      case 36: k = conv3d_wgrad_f16_pp_kernel<R, false, 36>; break;
      case 128: k = conv3d_wgrad_f16_pp_kernel<R, false, 128>; break;
      default: k = conv3d_wgrad_f16_pp_kernel<R, false, 0>; break;
    }
  }
#endif
  const int lds = WgradLds<R>::BYTES + (3 * (R + 2) * 4 + 15) / 16 * 16;     // + the strip's row maxima
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) { set_error("conv3d_wgrad_f16: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  hipLaunchKernelGGL(k, dim3((unsigned)(w.P * w.citiles * w.cotiles)), dim3(512), lds, s, x, gy, xa, ga, B, Ci, Co, w.P, w.citiles, part,
                     gb ? gb_part : nullptr, x_seg, x_words, gy_words, maxima);
  if (int rc = check_launch("conv3d_wgrad_f16")) return rc;
  const int CoP = w.cotiles * kWgCo, CiP = w.citiles * kWgCi;
  hipLaunchKernelGGL(conv3d_wgrad_f16_reduce_kernel, dim3((unsigned)(CoP * w.citiles)), dim3(1024), 0, s, part, gb_part, maxima, w.P, CoP, CiP,
                     Co, Ci, gw, gb);
  return check_launch("conv3d_wgrad_f16_reduce");
}

}  // namespace pvcnn

using namespace pvcnn;

static bool wgrad_f16_serves(int R) { return R == 8 || R == 12 || R == 16 || R == 32; }

// 0 when this shape is not served by the f16x2 kernel (R not 8, 12, 16 or 32): use pvcnn_conv3d_bwd_weight
extern "C" size_t pvcnn_conv3d_bwd_weight_f16_workspace_bytes(int B, int Ci, int Co, int R) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || !wgrad_f16_serves(R)) return 0;
  const WgradPlan w = wgrad_f16_plan(B, Ci, Co, R);
  return (w.part_floats + w.gb_floats) * sizeof(float);
}

extern "C" int pvcnn_conv3d_bwd_weight_f16(const float *x, const float *grad_y, const void *x_absmax, int x_amax_seg, const void *gy_absmax,
                                           int gy_amax_seg, int B, int Ci, int Co, int R, float *grad_w, float *grad_bias, void *workspace,
                                           size_t workspace_bytes, void *stream) {
  PVCNN_REQUIRE(B > 0 && Ci > 0 && Co > 0 && wgrad_f16_serves(R), "bad size (R must be 8, 12, 16 or 32)");
  PVCNN_REQUIRE(x_amax_seg == 0 || x_amax_seg == R, "x_amax_seg must be 0 (1-word buffer) or R (amax buffer with one maximum per z row)");
  PVCNN_REQUIRE(gy_amax_seg == 0 || gy_amax_seg == R, "gy_amax_seg must be 0 (word [0] holds the maximum) or R (amax buffer with one maximum per z row)");
  PVCNN_REQUIRE(x && grad_y && grad_w && x_absmax && gy_absmax, "null pointer");
  PVCNN_REQUIRE(aligned16(x) && aligned16(grad_y), "x and grad_y must be 16-byte aligned");
  PVCNN_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pvcnn_conv3d_bwd_weight_f16_workspace_bytes(B, Ci, Co, R),
                "workspace missing, misaligned or too small (see pvcnn_conv3d_bwd_weight_f16_workspace_bytes)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint32_t *xa = static_cast<const uint32_t *>(x_absmax), *ga = static_cast<const uint32_t *>(gy_absmax);
  float *ws = static_cast<float *>(workspace);
  switch (R) {
    case 32: return launch_wgrad_f16<32>(x, grad_y, xa, ga, B, Ci, Co, grad_w, grad_bias, ws, s, x_amax_seg, gy_amax_seg);
    case 16: return launch_wgrad_f16<16>(x, grad_y, xa, ga, B, Ci, Co, grad_w, grad_bias, ws, s, x_amax_seg, gy_amax_seg);
    case 12: return launch_wgrad_f16<12>(x, grad_y, xa, ga, B, Ci, Co, grad_w, grad_bias, ws, s, x_amax_seg, gy_amax_seg);
    default: return launch_wgrad_f16<8>(x, grad_y, xa, ga, B, Ci, Co, grad_w, grad_bias, ws, s, x_amax_seg, gy_amax_seg);
  }
}
