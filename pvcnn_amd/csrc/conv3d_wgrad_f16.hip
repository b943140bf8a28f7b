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
template <int R, bool PACK = false>
__global__ __launch_bounds__(512, 2) void conv3d_wgrad_f16_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                                  const uint32_t *__restrict__ x_absmax,
                                                                  const uint32_t *__restrict__ gy_absmax, int B, int Ci, int Co, int P,
                                                                  int citiles, float *__restrict__ part, float *__restrict__ gb_part,
                                                                  int x_seg) {
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
  const float x_scale = exp2_int(scale_shift(*x_absmax)), gy_scale = exp2_int(scale_shift(*gy_absmax));

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
      if (y2 < R) {
        const int gx0 = xo + xdx0 - 1, gx1 = xo + xdx1 - 1;
        if (has_x0 && (unsigned)gx0 < (unsigned)R && ci0 + xci0 < Ci)
          vx0 = *reinterpret_cast<const float4 *>(xb + (size_t)(ci0 + xci0) * S + (size_t)gx0 * RR + (size_t)y2 * R + 4 * xq0);
        if (has_x1 && (unsigned)gx1 < (unsigned)R && ci0 + xci1 < Ci)
          vx1 = *reinterpret_cast<const float4 *>(xb + (size_t)(ci0 + xci1) * S + (size_t)gx1 * RR + (size_t)y2 * R + 4 * xq1);
      }
      if (y1 >= 0 && y1 < R && has_g && co0 + gco < Co)
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
      if (live != 0) {
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
      __syncthreads();
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
          if (!PACK || j < 3 * Ci) pp[((size_t)tap * CoP + co) * CiP + ci0 + (PACK ? pci : j)] = acc[u][dz][r];
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

// gw[co][ci][tap] = 2^-(sx + sgy) * sum_p part[p][tap][co][ci];  gb[co] = sum_p gb_part[p][co]
// One workgroup per (co, 32-channel ci block): it owns the 27 x 32 = 864 CONTIGUOUS floats gw[co][ci0 .. ci0+31][0 .. 26].
// 864 of its 1024 threads = 216 float4 columns (tap, ci quad) x 4 partition slices; a thread sums its slice's partitions in a
// fixed order with eight 16-byte loads in flight (the first form gave one thread ALL partitions of one element behind four 4-byte
// loads: 1.8 MB in flight on the whole chip, 41.5 us for 57 MB), the four slices meet in LDS in a fixed order, and the block is
// written back transposed as one coalesced run.  Deterministic; no atomics.  Wave 14 sums the grad_bias partials of `co`.
constexpr int kRedSlices = 4, kRedCols = 27 * (kWgCi / 4);     // 216 float4 columns

__global__ __launch_bounds__(1024) void conv3d_wgrad_f16_reduce_kernel(const float *__restrict__ part, const float *__restrict__ gb_part,
                                                                       const uint32_t *__restrict__ x_absmax,
                                                                       const uint32_t *__restrict__ gy_absmax, int P, int CoP, int CiP,
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
      gw[((size_t)co * Ci + ci0 + ci_l) * 27 + tap] = s * exp2_int(-scale_shift(*x_absmax)) * exp2_int(-scale_shift(*gy_absmax));
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
  w.gb_floats = (size_t)w.P * w.cotiles * kWgCo;
  return w;
}

template <int R>
static int launch_wgrad_f16(const float *x, const float *gy, const uint32_t *xa, const uint32_t *ga, int B, int Ci, int Co, float *gw,
                            float *gb, float *ws, hipStream_t s, int x_seg) {
  const WgradPlan w = wgrad_f16_plan(B, Ci, Co, R);
  float *part = ws, *gb_part = ws + w.part_floats;
  const bool pack = 3 * Ci <= kWgCi && (R == 32 || R == 16);     // (instantiated for the grids a network's first layer has)
  auto k = pack ? conv3d_wgrad_f16_kernel<R, (R == 32 || R == 16)> : conv3d_wgrad_f16_kernel<R, false>;
  const int lds = WgradLds<R>::BYTES + (3 * (R + 2) * 4 + 15) / 16 * 16;     // + the strip's row maxima
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) { set_error("conv3d_wgrad_f16: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  hipLaunchKernelGGL(k, dim3((unsigned)(w.P * w.citiles * w.cotiles)), dim3(512), lds, s, x, gy, xa, ga, B, Ci, Co, w.P, w.citiles, part,
                     gb ? gb_part : nullptr, x_seg);
  if (int rc = check_launch("conv3d_wgrad_f16")) return rc;
  const int CoP = w.cotiles * kWgCo, CiP = w.citiles * kWgCi;
  hipLaunchKernelGGL(conv3d_wgrad_f16_reduce_kernel, dim3((unsigned)(CoP * w.citiles)), dim3(1024), 0, s, part, gb_part, xa, ga, w.P, CoP, CiP,
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
                                           int B, int Ci, int Co, int R, float *grad_w, float *grad_bias, void *workspace,
                                           size_t workspace_bytes, void *stream) {
  PVCNN_REQUIRE(B > 0 && Ci > 0 && Co > 0 && wgrad_f16_serves(R), "bad size (R must be 8, 12, 16 or 32)");
  PVCNN_REQUIRE(x_amax_seg == 0 || x_amax_seg == R, "x_amax_seg must be 0 (1-word buffer) or R (amax buffer with one maximum per z row)");
  PVCNN_REQUIRE(x && grad_y && grad_w && x_absmax && gy_absmax, "null pointer");
  PVCNN_REQUIRE(aligned16(x) && aligned16(grad_y), "x and grad_y must be 16-byte aligned");
  PVCNN_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pvcnn_conv3d_bwd_weight_f16_workspace_bytes(B, Ci, Co, R),
                "workspace missing, misaligned or too small (see pvcnn_conv3d_bwd_weight_f16_workspace_bytes)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint32_t *xa = static_cast<const uint32_t *>(x_absmax), *ga = static_cast<const uint32_t *>(gy_absmax);
  float *ws = static_cast<float *>(workspace);
  switch (R) {
    case 32: return launch_wgrad_f16<32>(x, grad_y, xa, ga, B, Ci, Co, grad_w, grad_bias, ws, s, x_amax_seg);
    case 16: return launch_wgrad_f16<16>(x, grad_y, xa, ga, B, Ci, Co, grad_w, grad_bias, ws, s, x_amax_seg);
    case 12: return launch_wgrad_f16<12>(x, grad_y, xa, ga, B, Ci, Co, grad_w, grad_bias, ws, s, x_amax_seg);
    default: return launch_wgrad_f16<8>(x, grad_y, xa, ga, B, Ci, Co, grad_w, grad_bias, ws, s, x_amax_seg);
  }
}
