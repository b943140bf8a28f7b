// conv3d.hip -- the dense 3x3x3 voxel convolutions of PVConv.voxel_layers on gfx950 matrix cores.
//
// Reference: modules/pvconv.py:20-27 -- nn.Conv3d(k=3, stride 1, pad 1) on (B, C, R, R, R) fp32
// grids, executed by cuDNN there.  This is the one genuinely dense contraction of the hot path
// (67-96 % of the model FLOPs), so it is the one place MFMA is used:
//     Y[b, co, v] = bias[co] + sum_{ci, tap} W[co, ci, tap] * X[b, ci, v + off(tap)]
// as an implicit GEMM  D[co][voxel] += A[co][k] * B[k][voxel],  k = (ci, tap), on
// v_mfma_f32_32x32x2_f32 (exact fp32 in / fp32 accumulate; 157 TFLOP/s peak; no TF32 on gfx950).
//
// Mapping (wave64, one workgroup = 4 waves = 256 output voxels x 64 output channels; a 128-voxel variant with one
// column block per wave is used when the 256-voxel grid would leave the chip below ~2 workgroups per CU):
//   * M = output channels, N = voxels: the MFMA C/D layout then puts 32 CONSECUTIVE-z voxels of
//     one channel in the 32 lanes of a half-wave, so results leave as full 128-byte rows of the
//     channel-major (B, C, R^3) tensor -- no transpose on the way out;
//   * per chunk of CIC input channels the workgroup stages in LDS
//       xs[CIC][(TX+2)(TY+2)(TZ+2)]  the input tile WITH its halo (zero-filled outside the grid)
//       ws[CIC][27][64]              the weights, pre-transposed on device to (Ci, 27, Co)
//     and all 27 taps are served from LDS (27-fold reuse of every staged input element);
//   * each wave owns 64 voxels x 64 channels = 2 x 2 MFMA tiles (64 accumulator VGPRs); per
//     (tap, channel pair) it issues 2 + 2 ds_read_b32 (conflict-free: consecutive z / consecutive
//     co) and 4 MFMAs;
//   * ~41 KiB of LDS per workgroup -> 3 workgroups per CU: one workgroup's global->LDS staging
//     overlaps the others' MFMA phases without explicit double buffering;
//   * the epilogue adds the bias, optionally emits per-workgroup BatchNorm partial sums of its outputs
//     (pvcnn_conv3d_fwd_stats), and reads each accumulator from its AGPR at the point of use.
// Backward-data is the same kernel on the flipped, channel-transposed weights (one tiny transform
// kernel).  Backward-weight is the transposed problem (K = voxels) -- conv3d_wgrad_kernel below.
#include <algorithm>

#include "common.h"

namespace pvcnn {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kCoTile = 64;

// (Co, Ci, 27) -> (Ci, 27, Co)                      [forward]
// (Co, Ci, 27) -> (Co, 27, Ci) with taps reversed   [backward-data: a conv with Ci' = Co, Co' = Ci]
__global__ __launch_bounds__(256) void conv3d_weight_transform_kernel(const float *__restrict__ w, int Co, int Ci,
                                                                      int for_bwd_data, float *__restrict__ wt) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= Co * Ci * 27) return;
  const int tap = e % 27, ci = (e / 27) % Ci, co = e / (27 * Ci);
  if (!for_bwd_data) wt[((size_t)ci * 27 + tap) * Co + co] = w[e];
  else               wt[((size_t)co * 27 + (26 - tap)) * Ci + ci] = w[e];
}


// ---------------------------------------------------------------------------------------------
// Tile staging (global -> registers -> LDS).  Staging is latency-bound unless many loads are in
// flight per thread, so both paths first issue ALL of a thread's loads into a register array and
// only then touch LDS.
//   VEC path (R == TZ: every z-row of the tile is one full, 16-byte aligned row of the grid and the
//   z halo is always outside the grid): one float4 per (row, quad);
//   scalar path (any R): element-wise with bounds checks, 8 loads per batch.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4g(const float *p) { return *reinterpret_cast<const float4 *>(p); }

template <int TX, int TY, int TZ, int NCH, bool VEC>
__device__ __forceinline__ void stage_halo_tile(float *xs, const float *xb, int c0, int Ci, int R, int x0, int y0, int z0,
                                                int tid) {
  constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2, HS = HX * HY * HZ;
  const size_t RR = (size_t)R * R, S = RR * R;
  if constexpr (VEC) {
    constexpr int QPR = TZ / 4, ROWS = NCH * HX * HY, NQ = ROWS * QPR, ITER = (NQ + 255) / 256;
    float4 v[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int q = tid + it * 256;
      const int row = q / QPR, qi = q - row * QPR;
      const int c = row / (HX * HY), hx = (row / HY) % HX, hy = row % HY;
      const int gx = x0 + hx - 1, gy = y0 + hy - 1;
      v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < NQ && c0 + c < Ci && (unsigned)gx < (unsigned)R && (unsigned)gy < (unsigned)R)
        v[it] = ld4g(xb + (size_t)(c0 + c) * S + (size_t)gx * RR + (size_t)gy * R + qi * 4);
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int q = tid + it * 256;
      if (q < NQ) {
        const int row = q / QPR, qi = q - row * QPR;
        float *d = xs + row * HZ + 1 + qi * 4;     // row = (c*HX + hx)*HY + hy  ->  c*HS + (hx*HY+hy)*HZ
        d[0] = v[it].x; d[1] = v[it].y; d[2] = v[it].z; d[3] = v[it].w;
      }
    }
    for (int r = tid; r < ROWS * 2; r += 256) xs[(r >> 1) * HZ + ((r & 1) ? HZ - 1 : 0)] = 0.0f;   // z halo
  } else {
    constexpr int N = NCH * HS;
    for (int e0 = 0; e0 < N; e0 += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * 256 + tid;
        const int c = e / HS, r = e - c * HS;
        const int hx = r / (HY * HZ), hy = (r / HZ) % HY, hz = r % HZ;
        const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
        v[u] = 0.0f;
        if (e < N && c0 + c < Ci && (unsigned)gx < (unsigned)R && (unsigned)gy < (unsigned)R && (unsigned)gz < (unsigned)R)
          v[u] = xb[(size_t)(c0 + c) * S + (size_t)gx * RR + (size_t)gy * R + gz];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * 256 + tid;
        if (e < N) xs[e] = v[u];
      }
    }
  }
}

// NBW = MFMA column blocks (32 voxels) per wave: 2 -> a 256-voxel workgroup tile, 1 -> a 128-voxel tile (twice
// the workgroups: used when a 256-voxel grid would leave the chip with about one workgroup per CU).
template <int TX, int TY, int TZ, int CIC, bool VEC, int NBW>
__global__ __launch_bounds__(256) void conv3d_igemm_kernel(const float *__restrict__ x, const float *__restrict__ wt,
                                                           const float *__restrict__ bias, float *__restrict__ y,
                                                           int Ci, int Co, int R, int tiles_x, int tiles_y, int tiles_z,
                                                           float2 *__restrict__ stats_part) {
  static_assert(TX * TY * TZ == 128 * NBW, "a workgroup tile is 4 waves x NBW x 32 voxels");
  static_assert(CIC % 2 == 0, "channels are consumed in pairs (MFMA K = 2)");
  constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2, HS = HX * HY * HZ;
  constexpr int WS = 27 * kCoTile;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *xs = lds;
  float *ws = lds + CIC * HS;

  int bid = blockIdx.x;
  const int tzi = bid % tiles_z; bid /= tiles_z;
  const int tyi = bid % tiles_y; bid /= tiles_y;
  const int txi = bid % tiles_x; bid /= tiles_x;
  const int b = bid;
  const int co0 = blockIdx.y * kCoTile;
  const int x0 = txi * TX, y0 = tyi * TY, z0 = tzi * TZ;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const size_t RR = (size_t)R * R;

  int hb[NBW];   // LDS offset of this lane's voxel (N-block 0/1) incl. the k-half channel offset
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int m = wave * 32 * NBW + nb * 32 + j;
    const int zt = m % TZ, yt = (m / TZ) % TY, xt = m / (TZ * TY);
    hb[nb] = (xt * HY + yt) * HZ + zt + kh * HS;
  }
  const int a_off = kh * WS + j;

  f32x16 acc[2][NBW];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

  const float *xb = x + (size_t)b * Ci * R * RR;
  for (int c0 = 0; c0 < Ci; c0 += CIC) {
    __syncthreads();
    stage_halo_tile<TX, TY, TZ, CIC, VEC>(xs, xb, c0, Ci, R, x0, y0, z0, tid);
    // ---- weights of this channel chunk: wt is (Ci, 27, Co) -> ws[c][tap][64] ----
    if (VEC) {   // Co % 4 == 0: whole 16-byte quads, those beyond Co are zero
      constexpr int NQ = CIC * 27 * (kCoTile / 4), ITER = (NQ + 255) / 256;
      float4 v[ITER];
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int q = tid + it * 256;
        const int rowi = q / (kCoTile / 4), qi = q - rowi * (kCoTile / 4);   // rowi = c*27 + tap
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < NQ && c0 + rowi / 27 < Ci && co0 + qi * 4 < Co) v[it] = ld4g(wt + ((size_t)c0 * 27 + rowi) * Co + co0 + qi * 4);
      }
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int q = tid + it * 256;
        if (q < NQ) *reinterpret_cast<float4 *>(ws + q * 4) = v[it];
      }
    } else {
      for (int e0 = 0; e0 < CIC * WS; e0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * 256 + tid;
          const int c = e / WS, r = e - c * WS;
          const int tap = r / kCoTile, co = r % kCoTile;
          v[u] = 0.0f;
          if (e < CIC * WS && c0 + c < Ci && co0 + co < Co) v[u] = wt[((size_t)(c0 + c) * 27 + tap) * Co + co0 + co];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * 256 + tid;
          if (e < CIC * WS) ws[e] = v[u];
        }
      }
    }
    __syncthreads();
    // ---- 27 taps x CIC/2 channel pairs x (2x2) MFMA tiles ----
#pragma unroll 1
    for (int dxy = 0; dxy < 9; ++dxy) {
      const int dx = dxy / 3, dy = dxy - dx * 3;
      const int toff_xy = (dx * HY + dy) * HZ;
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        const int tap = dxy * 3 + dz;
#pragma unroll
        for (int cc = 0; cc < CIC; cc += 2) {
          const float a0 = ws[cc * WS + tap * kCoTile + a_off];
          const float a1 = ws[cc * WS + tap * kCoTile + a_off + 32];
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb) {
            const float bv_ = xs[cc * HS + hb[nb] + toff_xy + dz];
            acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv_, acc[0][nb], 0, 0, 0);
            acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv_, acc[1][nb], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue: D[i = co][j = voxel]; lane -> voxel j, register r -> co row ----
  // One 32-row block at a time: its 16 bias values as one batch of loads, and every accumulator fetched from
  // its AGPR at the point of use (left to the compiler, all 64 are copied to VGPRs in one block at the loop
  // exit and the kernel drops from 3 to 2 waves per SIMD).
  float *yb = y + (size_t)b * Co * R * RR;
  size_t voff[NBW];
  bool vok[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int m = wave * 32 * NBW + nb * 32 + j;
    const int zt = m % TZ, yt = (m / TZ) % TY, xt = m / (TZ * TY);
    const int gx = x0 + xt, gy = y0 + yt, gz = z0 + zt;
    vok[nb] = gx < R && gy < R && gz < R;
    voff[nb] = (size_t)gx * RR + (size_t)gy * R + gz;
  }
  // stats_part != nullptr: per-channel (sum, sum of squares) of this workgroup's outputs ride on the epilogue --
  // the BatchNorm that follows the convolution then needs no statistics pass over y (bn_finalize combines the
  // per-workgroup partials in fp64 exactly like bn_stats_kernel's).
  const bool want_stats = stats_part != nullptr;
  float2 *stat_lds = reinterpret_cast<float2 *>(lds);       // [4 waves][64 channels]
  if (want_stats) __syncthreads();                          // all waves are done reading xs / ws
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      bv[r] = (bias != nullptr && co < Co) ? bias[co] : 0.0f;
    }
    float ss[16], qq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ss[r] = qq[r] = 0.0f;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        float v;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[mb][nb][r]));
        if (want_stats) {                                   // statistics of (y - bias): the shift keeps E[a^2] - E[a]^2 well
          const float m = vok[nb] ? v : 0.0f;               // conditioned when the bias dwarfs the spread (bn_finalize adds it back)
          ss[r] += m;
          qq[r] += m * m;
        }
        v += bv[r];
        if (vok[nb] && co < Co) yb[(size_t)co * R * RR + voff[nb]] = v;
      }
    if (want_stats) {
      // lane j ends up with the totals of register (j >> 1) & 15 over its 32 voxels / points
      const float st = half_wave_sum16(ss, j), qt = half_wave_sum16(qq, j);
      const int rr = (j >> 1) & 15;
      if ((j & 1) == 0) stat_lds[wave * kCoTile + mb * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * kh] = make_float2(st, qt);
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < kCoTile && co0 + tid < Co) {
      float2 t = stat_lds[tid];
#pragma unroll
      for (int w = 1; w < 4; ++w) { t.x += stat_lds[w * kCoTile + tid].x; t.y += stat_lds[w * kCoTile + tid].y; }
      stats_part[(size_t)(co0 + tid) * gridDim.x + blockIdx.x] = t;
    }
  }
}

template <int TX, int TY, int TZ, int CIC, bool VEC, int NBW>
static int launch_igemm_v(const float *x, const float *wt, const float *bias, float *y, int B, int Ci, int Co, int R,
                          hipStream_t s, float2 *stats_part) {
  constexpr int HS = (TX + 2) * (TY + 2) * (TZ + 2);
  const size_t lds = (size_t)(CIC * HS + CIC * 27 * kCoTile) * sizeof(float);
  const int tx = ceil_div(R, TX), ty = ceil_div(R, TY), tz = ceil_div(R, TZ);
  auto k = conv3d_igemm_kernel<TX, TY, TZ, CIC, VEC, NBW>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("conv3d: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(k, dim3((unsigned)((long)B * tx * ty * tz), ceil_div(Co, kCoTile)), dim3(256), lds, s, x, wt, bias,
                     y, Ci, Co, R, tx, ty, tz, stats_part);
  return check_launch("conv3d_igemm");
}

template <int TX, int TY, int TZ, int CIC, int NBW = 2>
static int launch_igemm(const float *x, const float *wt, const float *bias, float *y, int B, int Ci, int Co, int R,
                        hipStream_t s, float2 *stats_part) {
  // vector staging needs full aligned z-rows (R == TZ) and 16-byte aligned weight rows (whole quads of co)
  const bool vec = (R == TZ) && (Co % 4 == 0) && aligned16(x) && aligned16(wt);
  return vec ? launch_igemm_v<TX, TY, TZ, CIC, true, NBW>(x, wt, bias, y, B, Ci, Co, R, s, stats_part)
             : launch_igemm_v<TX, TY, TZ, CIC, false, NBW>(x, wt, bias, y, B, Ci, Co, R, s, stats_part);
}


// ---------------------------------------------------------------------------------------------
// Backward-weight:  gw[co][ci][tap] = sum_{b, v} gy[b,co,v] * x[b,ci,v + off(tap)]
// as the GEMM  D[co][n] += A[co][k] * B[k][n]  with  n = (ci, tap)  and  K = voxels.
//   * one workgroup owns a (64 co) x (16 input channels x 27 taps = 432 -> 448 columns) slab of gw
//     and walks a strided subset of the spatial tiles, accumulating in registers: wave w holds
//     row block (w & 1) x 7 of the 14 column blocks, i.e. 7 MFMA tiles fed by 1 A + 7 B LDS reads
//     per K-step, with the next step's operands fetched while this step's MFMAs run;
//   * per tile it stages gy[64][256 voxels] (row stride 260) and the input halo tile
//     xs[16][(TX+2)(TY+2)][TZ+8] with z = 0 at float 4 of a row, so both land with 16-byte LDS
//     stores; B reads gather xs at (channel, tap) offsets fixed per lane;
//   * on the fast path the next tile's global loads are issued before the MFMA loop and land in
//     LDS after it (one workgroup per CU: there is no other wave to hide them behind);
//   * the partial slab goes to workspace[p] with plain coalesced stores; conv3d_wgrad_reduce_kernel
//     sums the P partials (float atomics would cost more than the whole GEMM on this chip).
// ---------------------------------------------------------------------------------------------
constexpr int kWgCic = 16;
constexpr int kWgN = kWgCic * 27;          // 432 real columns
constexpr int kWgBlocksPerWave = 7;        // 14 column blocks of 32 over 2 wave pairs
constexpr int kGyStride = 260;
constexpr int kWgZOff = 4;                 // xs row: z = -1 at float 3, z = 0..TZ-1 at 4.., z = TZ at 4+TZ

// xs strides (floats): z-row HZP, x-plane PS, channel CS.  Rows start 16-byte aligned; the pads are chosen
// (brute force over multiples of 4) so that the 32 lanes of a B read -- 27 taps of one channel + 5 of the
// next -- spread over the 32 LDS banks with at most 2 addresses per bank (4 with the dense strides).
template <int TX, int TY, int TZ> struct WgradPad { static constexpr int HZP = TZ + 8, PSPAD = 0, CSPAD = 0; };
template <> struct WgradPad<2, 4, 32> { static constexpr int HZP = 40, PSPAD = 4, CSPAD = 4; };    // PS 244, CS 980
template <> struct WgradPad<4, 4, 16> { static constexpr int HZP = 24, PSPAD = 4, CSPAD = 4; };    // PS 148, CS 892
template <> struct WgradPad<4, 8, 8>  { static constexpr int HZP = 20, PSPAD = 8, CSPAD = 16; };   // PS 208, CS 1264

template <int TX, int TY, int TZ>
struct WgradGeom {
  static constexpr int HX = TX + 2, HY = TY + 2, ROWS = HX * HY;
  static constexpr int HZP = WgradPad<TX, TY, TZ>::HZP;
  static constexpr int PS = HY * HZP + WgradPad<TX, TY, TZ>::PSPAD;
  static constexpr int CS = HX * PS + WgradPad<TX, TY, TZ>::CSPAD;
  static constexpr size_t kLdsBytes = (size_t)(kCoTile * kGyStride + kWgCic * CS) * sizeof(float);
  static_assert(HZP >= TZ + kWgZOff + 1 && HZP % 4 == 0 && PS % 4 == 0 && CS % 4 == 0, "xs row layout");
  __host__ __device__ static constexpr int row_offset(int c, int hx, int hy) { return c * CS + hx * PS + hy * HZP; }
};

// register images of one tile (fast path): issued as global loads, landed in LDS one tile later
template <int TX, int TY, int TZ>
struct WgradTileRegs {
  using G = WgradGeom<TX, TY, TZ>;
  static constexpr int QPR = TZ / 4;
  static constexpr int GY_ITER = kCoTile * 256 / 4 / 256;                       // 16
  static constexpr int X_NQ = kWgCic * G::ROWS * QPR, X_ITER = (X_NQ + 255) / 256;
  float4 gy[GY_ITER];
  float4 x[X_ITER];

  __device__ __forceinline__ void load(const float *xg, const float *gyg, int b, int c0, int co0, int Ci, int Co, int R,
                                       int x0, int y0, int tid) {
    const size_t RR = (size_t)R * R, S = RR * R;
    const float *gyb = gyg + (size_t)b * Co * S;
    const float *xb = xg + (size_t)b * Ci * S;
#pragma unroll
    for (int it = 0; it < GY_ITER; ++it) {
      const int q = tid + it * 256;
      const int co = q / 64, mq = q - co * 64;
      const int zrow = mq / QPR, qi = mq - zrow * QPR;
      const int gx = x0 + zrow / TY, gyy = y0 + zrow % TY;
      gy[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (co0 + co < Co && gx < R && gyy < R) gy[it] = ld4g(gyb + (size_t)(co0 + co) * S + (size_t)gx * RR + (size_t)gyy * R + qi * 4);
    }
#pragma unroll
    for (int it = 0; it < X_ITER; ++it) {
      const int q = tid + it * 256;
      const int row = q / QPR, qi = q - row * QPR;
      const int c = row / G::ROWS, hx = (row / G::HY) % G::HX, hy = row % G::HY;
      const int gx = x0 + hx - 1, gyy = y0 + hy - 1;
      x[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < X_NQ && c0 + c < Ci && (unsigned)gx < (unsigned)R && (unsigned)gyy < (unsigned)R)
        x[it] = ld4g(xb + (size_t)(c0 + c) * S + (size_t)gx * RR + (size_t)gyy * R + qi * 4);
    }
  }
  __device__ __forceinline__ void store(float *gys, float *xs, int tid) const {
#pragma unroll
    for (int it = 0; it < GY_ITER; ++it) {
      const int q = tid + it * 256;
      const int co = q / 64, mq = q - co * 64;
      *reinterpret_cast<float4 *>(gys + co * kGyStride + mq * 4) = gy[it];
    }
#pragma unroll
    for (int it = 0; it < X_ITER; ++it) {
      const int q = tid + it * 256;
      if (q < X_NQ) {
        const int row = q / QPR, qi = q - row * QPR;     // row = c*ROWS + hx*HY + hy
        const int c = row / G::ROWS, hx = (row / G::HY) % G::HX, hy = row % G::HY;
        *reinterpret_cast<float4 *>(xs + G::row_offset(c, hx, hy) + kWgZOff + qi * 4) = x[it];
      }
    }
  }
};

template <int TX, int TY, int TZ, bool VEC>
__global__ __launch_bounds__(256) void conv3d_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                           float *__restrict__ part, float *__restrict__ bias_part,
                                                           int B, int Ci, int Co, int R, int tiles_x, int tiles_y,
                                                           int tiles_z, int P) {
  static_assert(TX * TY * TZ == 256, "a workgroup tile is 256 voxels");
  using G = WgradGeom<TX, TY, TZ>;
  constexpr int HY = G::HY, HZP = G::HZP, PS = G::PS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *gys = lds;                          // [64][260]
  float *xs = lds + kCoTile * kGyStride;     // [16][HX planes of PS][HY rows of HZP]

  const int chunk = blockIdx.x, p = blockIdx.y;
  const int co0 = blockIdx.z * kCoTile, c0 = chunk * kWgCic;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int mb = wave & 1, nb0 = (wave >> 1) * kWgBlocksPerWave;
  const size_t RR = (size_t)R * R, S = RR * R;

  // this lane's B-operand columns: n = (nb0 + q)*32 + j
  int boff[kWgBlocksPerWave];
  bool bval[kWgBlocksPerWave];
#pragma unroll
  for (int q = 0; q < kWgBlocksPerWave; ++q) {
    const int n = (nb0 + q) * 32 + j;
    const int cl = n / 27, tap = n - cl * 27;
    const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    bval[q] = n < kWgN && c0 + cl < Ci;
    boff[q] = bval[q] ? G::row_offset(cl, dx, dy) + dz + (kWgZOff - 1) : kWgZOff;
  }

  f32x16 acc[kWgBlocksPerWave];
#pragma unroll
  for (int q = 0; q < kWgBlocksPerWave; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;

  const int tiles_per_cloud = tiles_x * tiles_y * tiles_z;
  const int tiles_total = B * tiles_per_cloud;
  auto decode = [&](int t, int &b, int &x0, int &y0, int &z0) {
    const int tzi = t % tiles_z; t /= tiles_z;
    const int tyi = t % tiles_y; t /= tiles_y;
    const int txi = t % tiles_x; t /= tiles_x;
    b = t; x0 = txi * TX; y0 = tyi * TY; z0 = tzi * TZ;
  };
  // K order.  MFMA step (pair pi, sub c) contracts voxels 4*pi + c (lanes 0-31) and 4*pi + 2 + c (lanes
  // 32-63), so a lane needs two CONSECUTIVE floats of its A row and of each B column per pair of steps:
  // one ds_read_b64 + 7 ds_read2_b32 feed 14 MFMAs, with immediate offsets inside a z-row.  With a single
  // wave per SIMD every instruction issued between two MFMAs delays the second, so the loop is kept at
  // ~0.6 non-MFMA instructions per MFMA and the next pair's reads are issued under this pair's MFMAs.
  // Columns that do not exist (padding of 432 -> 448, channels beyond Ci) still take part in the MFMAs
  // with whatever xs holds: an output column depends only on its own B column, and those columns are
  // never stored -- this keeps the loop free of branches and selects.
  auto k_loop = [&]() {
    constexpr int PPR = TZ / 4, NROW = 256 / TZ;      // pairs per z-row, z-rows per tile
    const float *a_row = gys + (mb * 32 + j) * kGyStride + 2 * kh;
    const float *b_row[kWgBlocksPerWave];
#pragma unroll
    for (int q = 0; q < kWgBlocksPerWave; ++q) b_row[q] = xs + boff[q] + 2 * kh;
    float2 a_cur;
    float b_cur[kWgBlocksPerWave][2];
    a_cur = *reinterpret_cast<const float2 *>(a_row);
#pragma unroll
    for (int q = 0; q < kWgBlocksPerWave; ++q) { b_cur[q][0] = b_row[q][0]; b_cur[q][1] = b_row[q][1]; }
#pragma unroll 1
    for (int row = 0; row < NROW; ++row) {
      const int rn = (row + 1) & (NROW - 1);          // the wrap-around fetch of the last row is unused
      const int shift = (rn / TY) * PS + (rn % TY) * HZP - ((row / TY) * PS + (row % TY) * HZP);
      const float *a_next = a_row + (rn - row) * TZ;
      const float *b_next[kWgBlocksPerWave];
#pragma unroll
      for (int q = 0; q < kWgBlocksPerWave; ++q) b_next[q] = b_row[q] + shift;
#pragma unroll
      for (int pi = 0; pi < PPR; ++pi) {
        float2 a_nxt;
        float b_nxt[kWgBlocksPerWave][2];
        if (pi + 1 < PPR) {
          a_nxt = *reinterpret_cast<const float2 *>(a_row + 4 * (pi + 1));
#pragma unroll
          for (int q = 0; q < kWgBlocksPerWave; ++q) { b_nxt[q][0] = b_row[q][4 * (pi + 1)]; b_nxt[q][1] = b_row[q][4 * (pi + 1) + 1]; }
        } else {
          a_nxt = *reinterpret_cast<const float2 *>(a_next);
#pragma unroll
          for (int q = 0; q < kWgBlocksPerWave; ++q) { b_nxt[q][0] = b_next[q][0]; b_nxt[q][1] = b_next[q][1]; }
        }
#pragma unroll
        for (int q = 0; q < kWgBlocksPerWave; ++q)
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur[q][0], acc[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < kWgBlocksPerWave; ++q)
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur[q][1], acc[q], 0, 0, 0);
        a_cur = a_nxt;
#pragma unroll
        for (int q = 0; q < kWgBlocksPerWave; ++q) { b_cur[q][0] = b_nxt[q][0]; b_cur[q][1] = b_nxt[q][1]; }
        // issue order: MFMA, LDS read, MFMA, LDS read, ... (8 reads under the first 8 of the 14 MFMAs)
#pragma unroll
        for (int g = 0; g < kWgBlocksPerWave + 1; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, kWgBlocksPerWave - 1, 0);
      }
      a_row = a_next;
#pragma unroll
      for (int q = 0; q < kWgBlocksPerWave; ++q) b_row[q] = b_next[q];
    }
  };

  // grad_bias[co] = sum of grad_y over batch and voxels: the channel-chunk-0 workgroups add up the rows of
  // the gy tile they have in LDS anyway (4 threads per output channel, 64 voxels each).
  const bool do_bias = (bias_part != nullptr) && (chunk == 0);
  float bsum = 0.0f;
  auto bias_acc = [&]() {
    if (do_bias) {
      const float *rowp = gys + (tid >> 2) * kGyStride + (tid & 3) * 64;
#pragma unroll 16
      for (int i = 0; i < 64; ++i) bsum += rowp[i];
    }
  };

  if constexpr (VEC) {
    // software pipeline: tile t+P's global loads are in flight while tile t's MFMA loop runs
    constexpr int ROWS2 = kWgCic * G::ROWS * 2;
    for (int r = tid; r < ROWS2; r += 256) {   // z halo: always outside the grid (R == TZ)
      const int row = r >> 1;
      xs[G::row_offset(row / G::ROWS, (row / HY) % G::HX, row % HY) + ((r & 1) ? kWgZOff + TZ : kWgZOff - 1)] = 0.0f;
    }
    WgradTileRegs<TX, TY, TZ> regs;
    int t = p, b, x0, y0, z0;
    if (t < tiles_total) {
      decode(t, b, x0, y0, z0);
      regs.load(x, gy, b, c0, co0, Ci, Co, R, x0, y0, tid);
      regs.store(gys, xs, tid);
    }
    __syncthreads();
    while (t < tiles_total) {
      const int tn = t + P;
      // the thread index is made opaque per tile: otherwise the ~60 per-thread load/store addresses are
      // hoisted out of this loop, stay live across the MFMA loop and push it into a spill-bound schedule
      int tid_o = tid;
      asm volatile("" : "+v"(tid_o));
      if (tn < tiles_total) {
        decode(tn, b, x0, y0, z0);
        regs.load(x, gy, b, c0, co0, Ci, Co, R, x0, y0, tid_o);
      }
      k_loop();
      bias_acc();
      __syncthreads();
      if (tn < tiles_total) {
        asm volatile("" : "+v"(tid_o));
        regs.store(gys, xs, tid_o);
        __syncthreads();
      }
      t = tn;
    }
  } else {
    for (int t = p; t < tiles_total; t += P) {
      int b, x0, y0, z0;
      decode(t, b, x0, y0, z0);
      __syncthreads();
      const float *gyb = gy + (size_t)b * Co * S;
      for (int e0 = 0; e0 < kCoTile * 256; e0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * 256 + tid;
          const int co = e >> 8, m = e & 255;
          const int zt = m % TZ, yt = (m / TZ) % TY, xt = m / (TZ * TY);
          const int gx = x0 + xt, gyy = y0 + yt, gz = z0 + zt;
          v[u] = 0.0f;
          if (co0 + co < Co && gx < R && gyy < R && gz < R) v[u] = gyb[(size_t)(co0 + co) * S + (size_t)gx * RR + (size_t)gyy * R + gz];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * 256 + tid;
          gys[(e >> 8) * kGyStride + (e & 255)] = v[u];
        }
      }
      // input halo tile, scalar: element e = (c, hx, hy, hz) with hz = 0 <-> z0 - 1 stored at float kWgZOff - 1
      const float *xb = x + (size_t)b * Ci * S;
      constexpr int HZ = TZ + 2, NE = kWgCic * G::ROWS * HZ;
      for (int e0 = 0; e0 < NE; e0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * 256 + tid;
          const int row = e / HZ, hz = e - row * HZ;
          const int c = row / G::ROWS, hx = (row / HY) % G::HX, hy = row % HY;
          const int gx = x0 + hx - 1, gyy = y0 + hy - 1, gz = z0 + hz - 1;
          v[u] = 0.0f;
          if (e < NE && c0 + c < Ci && (unsigned)gx < (unsigned)R && (unsigned)gyy < (unsigned)R && (unsigned)gz < (unsigned)R)
            v[u] = xb[(size_t)(c0 + c) * S + (size_t)gx * RR + (size_t)gyy * R + gz];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * 256 + tid;
          if (e < NE) {
            const int row = e / HZ, hz = e - row * HZ;
            xs[G::row_offset(row / G::ROWS, (row / HY) % G::HX, row % HY) + (kWgZOff - 1) + hz] = v[u];
          }
        }
      }
      __syncthreads();
      k_loop();
      bias_acc();
    }
  }
  if (do_bias) {   // 4 partial sums per channel -> one value per (partition, channel)
    bsum += __shfl_xor(bsum, 1);
    bsum += __shfl_xor(bsum, 2);
    const int co = co0 + (tid >> 2);
    if ((tid & 3) == 0 && co < Co) bias_part[(size_t)p * Co + co] = bsum;
  }
  // ---- partial slab -> workspace[p][co][ci*27 + tap]: lanes = consecutive columns ----
  float *out = part + (size_t)p * Co * Ci * 27;
#pragma unroll
  for (int q = 0; q < kWgBlocksPerWave; ++q) {
    if (!bval[q]) continue;
    const int n = (nb0 + q) * 32 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (co < Co) out[(size_t)co * Ci * 27 + (size_t)c0 * 27 + n] = acc[q][r];
    }
  }
}

__global__ __launch_bounds__(256) void conv3d_wgrad_reduce_kernel(const float *__restrict__ part, int n, int P,
                                                                  float *__restrict__ gw) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float s = 0.0f;
  int p = 0;
  for (; p + 8 <= P; p += 8) {          // 8 loads in flight, summed in partition order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(p + u) * n + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; p < P; ++p) s += part[(size_t)p * n + e];
  gw[e] = s;
}

inline int wgrad_partitions(int B, int Ci, int Co, int tiles_per_cloud) {
  const int tiles_total = B * tiles_per_cloud;
  const int slabs = ceil_div(Ci, kWgCic) * ceil_div(Co, kCoTile);
  int P = std::max(1, kNumCU / slabs);                // one workgroup per CU (LDS) -> one full round
  P = std::min(P, tiles_total);
  while (P > 1 && tiles_total % P) --P;               // equal work per partition
  return P;
}

template <int TX, int TY, int TZ>
static int launch_wgrad(const float *x, const float *gy, float *gw, float *gb, float *part, int B, int Ci, int Co, int R,
                        int P, hipStream_t s) {
  const size_t lds = WgradGeom<TX, TY, TZ>::kLdsBytes;
  const int tx = ceil_div(R, TX), ty = ceil_div(R, TY), tz = ceil_div(R, TZ);
  const bool vec = (R == TZ) && aligned16(x) && aligned16(gy);
  auto k = vec ? conv3d_wgrad_kernel<TX, TY, TZ, true> : conv3d_wgrad_kernel<TX, TY, TZ, false>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("conv3d_wgrad: LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  float *bias_part = gb ? part + (size_t)P * Co * Ci * 27 : nullptr;   // (P, Co) behind the weight partials
  hipLaunchKernelGGL(k, dim3(ceil_div(Ci, kWgCic), P, ceil_div(Co, kCoTile)), dim3(256), lds, s, x, gy, part, bias_part, B,
                     Ci, Co, R, tx, ty, tz, P);
  if (int rc = check_launch("conv3d_wgrad")) return rc;
  const int n = Co * Ci * 27;
  hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, part, n, P, gw);
  if (int rc = check_launch("conv3d_wgrad_reduce")) return rc;
  if (gb) {
    hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(ceil_div(Co, 256)), dim3(256), 0, s, bias_part, Co, P, gb);
    return check_launch("conv3d_bias_reduce");
  }
  return 0;
}

inline void wgrad_tiles(int R, int &tpc) {
  if (R > 16) tpc = ceil_div(R, 2) * ceil_div(R, 4) * ceil_div(R, 32);
  else if (R > 8) tpc = ceil_div(R, 4) * ceil_div(R, 4) * ceil_div(R, 16);
  else tpc = ceil_div(R, 4) * ceil_div(R, 8) * ceil_div(R, 8);
}

}  // namespace pvcnn

using namespace pvcnn;

extern "C" int pvcnn_conv3d_weight_transform(const float *w, int Co, int Ci, int for_bwd_data, float *wt, void *stream) {
  PVCNN_REQUIRE(Co > 0 && Ci > 0 && w && wt, "bad argument");
  hipLaunchKernelGGL(conv3d_weight_transform_kernel, dim3(ceil_div(Co * Ci * 27, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), w, Co, Ci, for_bwd_data, wt);
  return check_launch("conv3d_weight_transform");
}

// tile variant of a forward launch: 0 -> (2,4,32), 1 -> (2,4,16) half tile, 2 -> (4,4,16), 3 -> (4,8,8), 4 -> (2,8,8) half tile;
// *nparts = spatial
// workgroups = statistics partials per output channel
static int igemm_variant(int B, int Co, int R, long *nparts) {
  int v, tx, ty, tz;
  if (R > 16) { v = 0; tx = 2; ty = 4; tz = 32; }
  else if (R > 8) {
    // 256-voxel tiles would give this launch fewer than ~2 workgroups per CU: halve the tile instead
    const long wgs256 = (long)B * ceil_div(R, 4) * ceil_div(R, 4) * ceil_div(R, 16) * ceil_div(Co, kCoTile);
    if (wgs256 < 2L * kNumCU) { v = 1; tx = 2; ty = 4; tz = 16; } else { v = 2; tx = 4; ty = 4; tz = 16; }
  } else {
    const long wgs256 = (long)B * ceil_div(R, 4) * ceil_div(R, 8) * ceil_div(R, 8) * ceil_div(Co, kCoTile);
    if (wgs256 < 2L * kNumCU) { v = 4; tx = 2; ty = 8; tz = 8; } else { v = 3; tx = 4; ty = 8; tz = 8; }
  }
  *nparts = (long)B * ceil_div(R, tx) * ceil_div(R, ty) * ceil_div(R, tz);
  return v;
}

static int conv3d_fwd_impl(const float *x, const float *wt, const float *bias, int B, int Ci, int Co, int R, float *y,
                           float2 *stats_part, hipStream_t s) {
  long nparts;
  switch (igemm_variant(B, Co, R, &nparts)) {
    case 0: return launch_igemm<2, 4, 32, 4>(x, wt, bias, y, B, Ci, Co, R, s, stats_part);
    case 1: return launch_igemm<2, 4, 16, 4, 1>(x, wt, bias, y, B, Ci, Co, R, s, stats_part);
    case 2: return launch_igemm<4, 4, 16, 4>(x, wt, bias, y, B, Ci, Co, R, s, stats_part);
    case 4: return launch_igemm<2, 8, 8, 4, 1>(x, wt, bias, y, B, Ci, Co, R, s, stats_part);
    default: return launch_igemm<4, 8, 8, 4>(x, wt, bias, y, B, Ci, Co, R, s, stats_part);
  }
}

extern "C" int pvcnn_conv3d_fwd(const float *x, const float *wt, const float *bias, int B, int Ci, int Co, int R,
                                float *y, void *stream) {
  PVCNN_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && R > 0, "bad size");
  if (B == 0) return 0;
  PVCNN_REQUIRE(x && wt && y, "null pointer");
  PVCNN_REQUIRE((long)R * R * R * (long)std::max(Ci, Co) <= 0x7fffffffL, "grid too large");
  return conv3d_fwd_impl(x, wt, bias, B, Ci, Co, R, y, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" size_t pvcnn_conv3d_fwd_stats_parts(int B, int Co, int R) {
  if (B <= 0 || Co <= 0 || R <= 0) return 0;
  long nparts;
  igemm_variant(B, Co, R, &nparts);
  return (size_t)nparts;
}

extern "C" int pvcnn_conv3d_fwd_stats(const float *x, const float *wt, const float *bias, int B, int Ci, int Co, int R,
                                      float *y, float *stats_part, void *stream) {
  PVCNN_REQUIRE(B > 0 && Ci > 0 && Co > 0 && R > 0, "bad size");
  PVCNN_REQUIRE(x && wt && y && stats_part, "null pointer");
  PVCNN_REQUIRE((reinterpret_cast<uintptr_t>(stats_part) & 7) == 0, "stats_part must be 8-byte aligned");
  PVCNN_REQUIRE((long)R * R * R * (long)std::max(Ci, Co) <= 0x7fffffffL, "grid too large");
  return conv3d_fwd_impl(x, wt, bias, B, Ci, Co, R, y, reinterpret_cast<float2 *>(stats_part), static_cast<hipStream_t>(stream));
}

extern "C" size_t pvcnn_conv3d_bwd_weight_workspace_bytes(int B, int Ci, int Co, int R) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || R <= 0) return 0;
  int tpc;
  wgrad_tiles(R, tpc);
  const size_t P = (size_t)wgrad_partitions(B, Ci, Co, tpc);
  return P * Co * Ci * 27 * sizeof(float) + P * Co * sizeof(float) + 16;
}

extern "C" int pvcnn_conv3d_bwd_weight(const float *x, const float *grad_y, int B, int Ci, int Co, int R, float *grad_w,
                                       float *grad_bias, void *workspace, size_t workspace_bytes, void *stream) {
  PVCNN_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && R > 0, "bad size");
  PVCNN_REQUIRE(grad_w, "null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B == 0) {
    PVCNN_HIP_TRY(hipMemsetAsync(grad_w, 0, (size_t)Co * Ci * 27 * sizeof(float), s));
    if (grad_bias) PVCNN_HIP_TRY(hipMemsetAsync(grad_bias, 0, (size_t)Co * sizeof(float), s));
    return 0;
  }
  PVCNN_REQUIRE(x && grad_y, "null pointer");
  PVCNN_REQUIRE((long)R * R * R * (long)std::max(Ci, Co) <= 0x7fffffffL, "grid too large");
  PVCNN_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= pvcnn_conv3d_bwd_weight_workspace_bytes(B, Ci, Co, R),
                "workspace missing, misaligned or too small (see pvcnn_conv3d_bwd_weight_workspace_bytes)");
  int tpc;
  wgrad_tiles(R, tpc);
  const int P = wgrad_partitions(B, Ci, Co, tpc);
  float *part = static_cast<float *>(workspace);
  if (R > 16) return launch_wgrad<2, 4, 32>(x, grad_y, grad_w, grad_bias, part, B, Ci, Co, R, P, s);
  if (R > 8)  return launch_wgrad<4, 4, 16>(x, grad_y, grad_w, grad_bias, part, B, Ci, Co, R, P, s);
  return launch_wgrad<4, 8, 8>(x, grad_y, grad_w, grad_bias, part, B, Ci, Co, R, P, s);
}
