// conv3d_bf16.hip -- the 3x3x3 voxel convolutions on the bf16 matrix cores of gfx950, in two precisions:
//
//   NS = 1  plain bf16 operands, fp32 accumulate            (BASELINE configs[4]: "bf16 with MFMA 3D conv")
//   NS = 3  "bf16x3": every fp32 operand is split EXACTLY into three bf16 pieces, x = x0 + x1 + x2 (8 + 8 + 8 mantissa
//           bits), and the product x * w is evaluated as the six partial products whose weight is >= 2^-16 of the
//           leading one:  x0w0 + x0w1 + x1w0 + x0w2 + x1w1 + x2w0  (dropped: x1w2, x2w1, x2w2 <= 2^-24 relative),
//           each exact in the MFMA (8 x 8 bit products), accumulated in fp32.  Result: fp32-class accuracy (measured
//           against fp64 in tests/test_gpu_conv3d.py: same 1e-5 bar, same error level as the exact-fp32 MFMA kernel of
//           conv3d.hip) at 6 bf16 MFMAs per 16-deep k-step -- v_mfma_f32_32x32x16_bf16 runs at 16x the rate of
//           v_mfma_f32_32x32x2_f32, so the ceiling is 16 / 6 = 2.7x the fp32-MFMA peak (157 -> 419 TFLOP/s effective).
//           (The reference's own fp32 convolution is cuDNN under PyTorch's default allow_tf32 = True: a 10-bit-mantissa
//           product on Ampere and later; this split keeps all 24 bits of both operands.)
//
// Implicit GEMM, D[co][voxel] += A[co][k] * B[k][voxel] on v_mfma_f32_32x32x16_bf16 with k = 16 INPUT CHANNELS of one tap:
//   * per chunk of 16 input channels a workgroup (256 threads, 256 output voxels x 64 output channels) stages its input
//     tile WITH halo once, converting fp32 -> NS bf16 planes on the way: xs[plane][halo voxel][16 ch] (32 B per voxel,
//     the two 8-channel halves XOR-swizzled by bit 3 of the voxel index: the 16-byte operand reads of 32 consecutive
//     voxels then hit 64 distinct banks);
//   * weights arrive pre-split and pre-swizzled in exactly the LDS layout (conv3d_weight_split_kernel, once per forward):
//     per (dx, dy) the 3 dz taps x NS planes x 64 co x 16 ci = NS * 6 KiB are one contiguous block -> straight 16-byte copies;
//   * a consumer wave owns 64 voxels x 64 channels (2 x 2 MFMA tiles); per tap it reads 2 * NS weight fragments (from the
//     pre-split image in global memory / L2) and 2 * NS input fragments (LDS) for 4 * (NS == 3 ? 6 : 1) MFMAs;
//   * 61 KiB of LDS at NS = 3 -> 2 workgroups per CU: one stages while the other multiplies;
//   * epilogue as in conv3d.hip: C/D rows are 32 consecutive-z voxels of one channel = 128-byte rows of (B, C, R^3); bias;
//     optional BatchNorm partial sums of (y - bias).
// Backward-data is the same kernel on the flipped, channel-transposed weights (the split kernel's for_bwd_data mode).
#include <algorithm>

#include "common.h"

namespace pvcnn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kCoTileB = 64;
constexpr int kKc = 16;            // input channels per chunk = MFMA K

// round-to-nearest-even fp32 -> bf16 (bits); inputs are finite in this path
__device__ __forceinline__ uint32_t bf16_bits(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_value(uint32_t bits) { return __uint_as_float(bits << 16); }

// v -> NS bf16 pieces with v = p0 + p1 + p2 (exactly, up to the last piece's rounding at 2^-24 |v|)
template <int NS>
__device__ __forceinline__ void split_bf16(float v, uint32_t (&p)[NS]) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    p[s] = bf16_bits(v);
    if (s + 1 < NS) v = v - bf16_value(p[s]);          // exact: the residual fits fp32
  }
}

// ---- weights: (Co, Ci, 27) fp32 -> [chunk][dxy][cotile][dz][plane][64 co][16 ci (halves swizzled)] bf16 ----
// for_bwd_data: the convolution computed is grad_x = conv(grad_y, w') with Ci' = Co, Co' = Ci, w'[ci][co][tap] = w[co][ci][26 - tap]
template <int NS>
__global__ __launch_bounds__(256) void conv3d_weight_split_kernel(const float *__restrict__ w, int Co, int Ci, int for_bwd_data,
                                                                  uint16_t *__restrict__ wts) {
  const int CiE = for_bwd_data ? Co : Ci, CoE = for_bwd_data ? Ci : Co;       // effective (reduction, output) channel counts
  const int chunks = ceil_div(CiE, kKc), cotiles = ceil_div(CoE, kCoTileB);
  const long total = (long)chunks * 9 * cotiles * 3 * kCoTileB * kKc;          // one thread per (.., dz, co, ci): writes NS planes
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int ci_l = (int)(e % kKc), co_l = (int)((e / kKc) % kCoTileB), dz = (int)((e / (kKc * kCoTileB)) % 3);
  long rest = e / (kKc * kCoTileB * 3);
  const int cot = (int)(rest % cotiles); rest /= cotiles;
  const int dxy = (int)(rest % 9);
  const int chunk = (int)(rest / 9);
  const int ci = chunk * kKc + ci_l, co = cot * kCoTileB + co_l, tap = dxy * 3 + dz;
  float v = 0.0f;
  if (ci < CiE && co < CoE)
    v = for_bwd_data ? w[((size_t)ci * Ci + co) * 27 + (26 - tap)] : w[((size_t)co * Ci + ci) * 27 + tap];
  uint32_t p[NS];
  split_bf16<NS>(v, p);
  const int pos = ((ci_l >> 3) ^ ((co_l >> 3) & 1)) * 8 + (ci_l & 7);
  const size_t blk = (((size_t)chunk * 9 + dxy) * cotiles + cot) * (3 * NS * kCoTileB * kKc);
#pragma unroll
  for (int s = 0; s < NS; ++s) wts[blk + ((size_t)(dz * NS + s) * kCoTileB + co_l) * kKc + pos] = (uint16_t)p[s];
}

// Variants measured on (16,64,64,32^3), bf16x3 (fp32 kernel: 0.87 ms; 6x the MFMAs at 16x the rate = 0.28 ms at 2.4 GHz):
//   this one (256 threads, 2 workgroups per CU, 27 taps unrolled, weights one tap ahead)   0.56 ms
//   explicit one-tap software pipeline of both operands behind sched_barriers              0.58-0.61 ms (1 / 2 waves per SIMD)
//   producer / consumer wave specialisation, double-buffered tile (1 workgroup per CU)     0.62 ms
//   the same MFMA stream with NO loads and NO staging at all                                0.43 ms
// i.e. the matrix pipe itself sustains ~1.6 PF on random data here (the chip clocks down under a dense bf16 MFMA stream),
// and what is left above it is prologue / epilogue exposure; the simplest structure is kept.
template <int NS, int TX, int TY, int TZ>
__global__ __launch_bounds__(256, 2) void conv3d_igemm_bf16_kernel(const float *__restrict__ x, const uint16_t *__restrict__ wts,
                                                                   const float *__restrict__ bias, float *__restrict__ y,
                                                                   int Ci, int Co, int R, int tiles_x, int tiles_y, int tiles_z,
                                                                   float2 *__restrict__ stats_part) {
  static_assert(TX * TY * TZ == 256, "a workgroup tile is 4 waves x 2 x 32 voxels");
  constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2, HS = HX * HY * HZ;
  constexpr int NBW = 2;
  constexpr int WBLK = 3 * NS * kCoTileB * kKc;                 // bf16 elements of one (chunk, dxy, cotile) weight block
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
  uint32_t *xs = lds_u;                                         // [NS][HS][8] words (16 bf16 per voxel)

  int bid = blockIdx.x;
  const int tzi = bid % tiles_z; bid /= tiles_z;
  const int tyi = bid % tiles_y; bid /= tiles_y;
  const int txi = bid % tiles_x; bid /= tiles_x;
  const int b = bid;
  const int cot = blockIdx.y, co0 = cot * kCoTileB, cotiles = gridDim.y;
  const int x0 = txi * TX, y0 = tyi * TY, z0 = tzi * TZ;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const size_t RR = (size_t)R * R, S = RR * R;
  const float *xb = x + (size_t)b * Ci * S;
  const int chunks = ceil_div(Ci, kKc);

  int hb[NBW];                                                  // halo index of this lane's output voxel (tap 0,0,0 corner)
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int m = wave * 64 + nb * 32 + j;
    const int zt = m % TZ, yt = (m / TZ) % TY, xt = m / (TZ * TY);
    hb[nb] = (xt * HY + yt) * HZ + zt;
  }
  int a_off[2];                                                 // A fragment word offsets inside one (dz, plane) slab
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int row = mb * 32 + j;
    a_off[mb] = row * 8 + ((kh ^ ((row >> 3) & 1)) * 4);
  }
  f32x16 acc[2][NBW];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

  for (int chunk = 0; chunk < chunks; ++chunk) {
    const int c0 = chunk * kKc;
    __syncthreads();                                            // previous chunk's fragment reads are done
    // ---- stage the halo tile: item = (halo voxel, channel pair); fp32 -> NS bf16 planes, channels-last, swizzled ----
    {
      constexpr int ITER_ALL = (HS * 8 + 255) / 256, ITER = (ITER_ALL + 1) / 2;   // two batches: ~22 loads in flight each
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        float va[ITER], vb[ITER];
#pragma unroll
        for (int u = 0; u < ITER; ++u) {
          const int e = (half * ITER + u) * 256 + tid;
          const int cp = e & 7, v = e >> 3;
          const int hx = v / (HY * HZ), hy = (v / HZ) % HY, hz = v % HZ;
          const int gx = x0 + hx - 1, gy = y0 + hy - 1, gz = z0 + hz - 1;
          const int c = c0 + 2 * cp;
          va[u] = vb[u] = 0.0f;
          if (e < HS * 8 && (unsigned)gx < (unsigned)R && (unsigned)gy < (unsigned)R && (unsigned)gz < (unsigned)R) {
            const size_t off = (size_t)gx * RR + (size_t)gy * R + gz;
            if (c < Ci) va[u] = xb[(size_t)c * S + off];
            if (c + 1 < Ci) vb[u] = xb[(size_t)(c + 1) * S + off];
          }
        }
#pragma unroll
        for (int u = 0; u < ITER; ++u) {
          const int e = (half * ITER + u) * 256 + tid;
          if (e < HS * 8) {
            const int cp = e & 7, v = e >> 3;
            uint32_t pa[NS], pb[NS];
            split_bf16<NS>(va[u], pa);
            split_bf16<NS>(vb[u], pb);
            const int word = v * 8 + (((cp >> 2) ^ ((v >> 3) & 1)) * 4) + (cp & 3);
#pragma unroll
            for (int s = 0; s < NS; ++s) xs[s * HS * 8 + word] = pa[s] | (pb[s] << 16);
          }
        }
      }
    }
    __syncthreads();                                            // x tile staged
    // ---- weights: each lane fetches its own 16-byte A fragments straight from the pre-split image in global memory (a
    // slab is 64 rows x 32 B: the 64 lanes of a wave read 2 KiB contiguous, L2-resident -- the whole image is a few hundred
    // KiB shared by every workgroup).  No LDS copy of the weights, no barriers inside the tap loop.
    const uint4 *wblk = reinterpret_cast<const uint4 *>(wts + (((size_t)chunk * 9) * cotiles + cot) * WBLK);
    auto load_a = [&](int tap, uint4 (&af)[2][NS]) {
      const int dxy = tap / 3, dz = tap - dxy * 3;
      const uint4 *wq = wblk + (size_t)dxy * cotiles * (WBLK / 8);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int s = 0; s < NS; ++s) af[mb][s] = wq[((dz * NS + s) * kCoTileB * 8 + a_off[mb]) >> 2];
    };
    uint4 afn[2][NS];
    load_a(0, afn);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      uint4 af[2][NS];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int s = 0; s < NS; ++s) af[mb][s] = afn[mb][s];
      if (tap + 1 < 27) load_a(tap + 1, afn);                   // next tap's weights are in flight during this tap's MFMAs
      const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
      const int toff = (dx * HY + dy) * HZ + dz;
      uint4 bf[NBW][NS];
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        const int vi = hb[nb] + toff;
        const int boff = vi * 8 + ((kh ^ ((vi >> 3) & 1)) * 4);
#pragma unroll
        for (int s = 0; s < NS; ++s) bf[nb][s] = *reinterpret_cast<const uint4 *>(xs + s * HS * 8 + boff);
      }
      // consecutive MFMAs go to different accumulators (4 independent tiles between two partial products of one tile)
#define PVCNN_MFMA4(SA, SB)                                                                                              \
      _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb)                                                                 \
      _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                                   \
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[mb][SA]),                    \
                                                              __builtin_bit_cast(bf16x8, bf[nb][SB]), acc[mb][nb], 0, 0, 0)
      if constexpr (NS == 1) {
        PVCNN_MFMA4(0, 0);
      } else {
        PVCNN_MFMA4(2, 0); PVCNN_MFMA4(1, 1); PVCNN_MFMA4(0, 2);      // smallest partial products first
        PVCNN_MFMA4(1, 0); PVCNN_MFMA4(0, 1);
        PVCNN_MFMA4(0, 0);
      }
#undef PVCNN_MFMA4
    }
  }
  if (stats_part != nullptr) __syncthreads();                   // all waves are done reading xs before it is reused below

  // ---- epilogue: D[i = co][j = voxel]; lane -> voxel j, register r -> co row (C/D map of the fp32 MFMA) ----
  float *yb = y + (size_t)b * Co * S;
  size_t voff[NBW];
  bool vok[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int m = wave * 64 + nb * 32 + j;
    const int zt = m % TZ, yt = (m / TZ) % TY, xt = m / (TZ * TY);
    const int gx = x0 + xt, gy = y0 + yt, gz = z0 + zt;
    vok[nb] = gx < R && gy < R && gz < R;
    voff[nb] = (size_t)gx * RR + (size_t)gy * R + gz;
  }
  const bool want_stats = stats_part != nullptr;
  float2 *stat_lds = reinterpret_cast<float2 *>(lds_u);        // [4 waves][64 channels]
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
        float v = acc[mb][nb][r];
        if (want_stats) {                                       // statistics of (y - bias), see bn_finalize_kernel
          const float m = vok[nb] ? v : 0.0f;
          ss[r] += m;
          qq[r] += m * m;
        }
        v += bv[r];
        if (vok[nb] && co < Co) yb[(size_t)co * S + voff[nb]] = v;
      }
    if (want_stats) {
      const float st = half_wave_sum16(ss, j), qt = half_wave_sum16(qq, j);
      const int rr = (j >> 1) & 15;
      if ((j & 1) == 0) stat_lds[wave * kCoTileB + mb * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * kh] = make_float2(st, qt);
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < kCoTileB && co0 + tid < Co) {
      float2 t = stat_lds[tid];
#pragma unroll
      for (int w = 1; w < 4; ++w) { t.x += stat_lds[w * kCoTileB + tid].x; t.y += stat_lds[w * kCoTileB + tid].y; }
      stats_part[(size_t)(co0 + tid) * gridDim.x + blockIdx.x] = t;
    }
  }
}

static void split_tiles(int R, int &tx, int &ty, int &tz) {
  if (R > 8) { tx = 4; ty = 4; tz = 16; } else { tx = 4; ty = 8; tz = 8; }
}

template <int NS, int TX, int TY, int TZ>
static int launch_igemm_bf16(const float *x, const uint16_t *wts, const float *bias, float *y, int B, int Ci, int Co, int R,
                             hipStream_t s, float2 *stats_part) {
  constexpr int HS = (TX + 2) * (TY + 2) * (TZ + 2);
  const size_t lds = std::max((size_t)NS * HS * 8 * sizeof(uint32_t), (size_t)4 * kCoTileB * sizeof(float2));
  const int tx = ceil_div(R, TX), ty = ceil_div(R, TY), tz = ceil_div(R, TZ);
  auto k = conv3d_igemm_bf16_kernel<NS, TX, TY, TZ>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("conv3d(bf16): LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(k, dim3((unsigned)((long)B * tx * ty * tz), ceil_div(Co, kCoTileB)), dim3(256), lds, s, x, wts, bias, y,
                     Ci, Co, R, tx, ty, tz, stats_part);
  return check_launch("conv3d_igemm_bf16");
}

}  // namespace pvcnn

using namespace pvcnn;

extern "C" size_t pvcnn_conv3d_weight_split_bytes(int Co, int Ci, int for_bwd_data, int nsplit) {
  if (Co <= 0 || Ci <= 0 || (nsplit != 1 && nsplit != 3)) return 0;
  const int CiE = for_bwd_data ? Co : Ci, CoE = for_bwd_data ? Ci : Co;
  return (size_t)ceil_div(CiE, kKc) * 9 * ceil_div(CoE, kCoTileB) * 3 * nsplit * kCoTileB * kKc * sizeof(uint16_t);
}

extern "C" int pvcnn_conv3d_weight_split(const float *w, int Co, int Ci, int for_bwd_data, int nsplit, void *wts, void *stream) {
  PVCNN_REQUIRE(w && wts && Co > 0 && Ci > 0, "bad argument");
  PVCNN_REQUIRE(nsplit == 1 || nsplit == 3, "nsplit must be 1 (bf16) or 3 (bf16x3)");
  PVCNN_REQUIRE(aligned16(wts), "wts must be 16-byte aligned");
  const int CiE = for_bwd_data ? Co : Ci, CoE = for_bwd_data ? Ci : Co;
  const long total = (long)ceil_div(CiE, kKc) * 9 * ceil_div(CoE, kCoTileB) * 3 * kCoTileB * kKc;
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (nsplit == 1) hipLaunchKernelGGL(conv3d_weight_split_kernel<1>, grid, dim3(256), 0, s, w, Co, Ci, for_bwd_data, static_cast<uint16_t *>(wts));
  else             hipLaunchKernelGGL(conv3d_weight_split_kernel<3>, grid, dim3(256), 0, s, w, Co, Ci, for_bwd_data, static_cast<uint16_t *>(wts));
  return check_launch("conv3d_weight_split");
}

extern "C" size_t pvcnn_conv3d_fwd_split_stats_parts(int B, int Co, int R) {
  if (B <= 0 || Co <= 0 || R <= 0) return 0;
  int tx, ty, tz;
  split_tiles(R, tx, ty, tz);
  return (size_t)B * ceil_div(R, tx) * ceil_div(R, ty) * ceil_div(R, tz);
}

// y = conv3d(x, w) + bias with the pre-split weights of pvcnn_conv3d_weight_split (forward layout: Ci, Co as given; backward-data:
// call with x = grad_y, Ci = the forward Co, Co = the forward Ci, bias = NULL and the for_bwd_data = 1 weights).
extern "C" int pvcnn_conv3d_fwd_split(const float *x, const void *wts, const float *bias, int B, int Ci, int Co, int R, int nsplit,
                                      float *y, float *stats_part, void *stream) {
  PVCNN_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && R > 0, "bad size");
  PVCNN_REQUIRE(nsplit == 1 || nsplit == 3, "nsplit must be 1 (bf16) or 3 (bf16x3)");
  if (B == 0) return 0;
  PVCNN_REQUIRE(x && wts && y && aligned16(wts), "null or misaligned pointer");
  PVCNN_REQUIRE(!stats_part || (reinterpret_cast<uintptr_t>(stats_part) & 7) == 0, "stats_part must be 8-byte aligned");
  PVCNN_REQUIRE((long)R * R * R * (long)std::max(Ci, Co) <= 0x7fffffffL, "grid too large");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint16_t *w16 = static_cast<const uint16_t *>(wts);
  float2 *sp = reinterpret_cast<float2 *>(stats_part);
  if (R > 8) return nsplit == 3 ? launch_igemm_bf16<3, 4, 4, 16>(x, w16, bias, y, B, Ci, Co, R, s, sp)
                                : launch_igemm_bf16<1, 4, 4, 16>(x, w16, bias, y, B, Ci, Co, R, s, sp);
  return nsplit == 3 ? launch_igemm_bf16<3, 4, 8, 8>(x, w16, bias, y, B, Ci, Co, R, s, sp)
                     : launch_igemm_bf16<1, 4, 8, 8>(x, w16, bias, y, B, Ci, Co, R, s, sp);
}
