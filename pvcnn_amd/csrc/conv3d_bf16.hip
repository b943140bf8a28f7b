// conv3d_bf16.hip -- the 3x3x3 voxel convolutions on the 16-bit matrix cores of gfx950 (v_mfma_f32_32x32x16_{bf16,f16}: 16x the
// rate of v_mfma_f32_32x32x2_f32), in three precisions:
//
//   NS = 1  plain bf16 operands, fp32 accumulate            (BASELINE configs[4]: "bf16 with MFMA 3D conv")
//   NS = 2  "f16x2" (the default for fp32 tensors): every fp32 operand is scaled by a power of two (per tensor for the
//           activations, per output channel for the weights -- fp16 has 5 exponent bits) and split into fp16 hi + lo,
//           x * 2^s = hi + lo (11 + 11 significant bits, |lo| <= 2^-11 |hi|); the product is hi*hi' + hi*lo' + lo*hi'
//           (dropped: lo*lo' <= 2^-22 relative), each exact in the MFMA (11 x 11 bit products), accumulated in fp32 and
//           scaled back by 2^-(s+s') in the epilogue (exact).  3 MFMAs per 16-deep k-step: ceiling 16 / 3 = 5.3x the
//           fp32-MFMA peak.  Measured against fp64: BELOW the error of the exact-fp32 MFMA kernel of conv3d.hip (its
//           accumulation order is the same; the operands keep 22 of 24 bits) -- tests/test_gpu_conv3d.py, also for tiny /
//           huge / outlier-dominated tensors and weight rows 14 decades apart.
//   NS = 3  "bf16x3": x = x0 + x1 + x2 EXACTLY in three bf16 pieces (8 + 8 + 8 bits, no scaling: bf16 has fp32's exponent),
//           product = the six partial products >= 2^-16 of the leading one (x0w0 + x0w1 + x1w0 + x0w2 + x1w1 + x2w0).
//           fp32-class as well, 6 MFMAs per k-step (ceiling 2.7x); kept as the scale-free alternative (PVCNN_CONV_MATH=bf16x3).
//   (The reference's own fp32 convolution is cuDNN under PyTorch's default allow_tf32 = True: a 10-bit-mantissa product on
//   Ampere and later; both splits keep >= 22 bits of both operands.)
//
// Implicit GEMM, D[co][voxel] += A[co][k] * B[k][voxel] on v_mfma_f32_32x32x16_bf16 with k = 16 INPUT CHANNELS of one tap:
//   * per chunk of 16 input channels a workgroup (256 threads, 256 or 512 output voxels x 64 output channels) stages its input
//     tile WITH halo once, converting fp32 -> NS 16-bit planes on the way (v_cvt_pk_{bf16,f16}_f32 on channel pairs): xs[plane][halo voxel][16 ch] (32 B per voxel,
//     the two 8-channel halves XOR-swizzled by bit 3 of the voxel index: the 16-byte operand reads of 32 consecutive
//     voxels then hit 64 distinct banks);
//   * weights arrive pre-split and pre-swizzled in exactly the LDS layout (conv3d_weight_split_kernel, once per forward):
//     per (dx, dy) the 3 dz taps x NS planes x 64 co x 16 ci = NS * 6 KiB are one contiguous block -> straight 16-byte copies;
//   * a consumer wave owns 64 voxels x 64 channels (2 x 2 MFMA tiles); per tap it reads 2 * NS weight fragments (from the
//     pre-split image in global memory / L2) and 2 * NS input fragments (LDS) for 4 * (NS == 3 ? 6 : 1) MFMAs;
//   * 41 / 61 KiB of LDS at NS = 2 / 3 for the 256-voxel tile -> 3 / 2 workgroups per CU: one stages while the others multiply;
//   * epilogue as in conv3d.hip: C/D rows are 32 consecutive-z voxels of one channel = 128-byte rows of (B, C, R^3); bias;
//     optional BatchNorm partial sums of (y - bias).
// Backward-data is the same kernel on the flipped, channel-transposed weights (the split kernel's for_bwd_data mode).
#include <algorithm>
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include <string.h>
#include "split16.h"

namespace pvcnn {

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

__global__ __launch_bounds__(512) void absmax_kernel(const float *__restrict__ x, size_t n, uint32_t *__restrict__ out) {
  uint32_t m = 0;
  const size_t n4 = n >> 2, stride = (size_t)gridDim.x * 512;
  const float4 *x4 = reinterpret_cast<const float4 *>(x);
  auto take = [&](const float4 &v) {
    m = max(max(m, __float_as_uint(fabsf(v.x))), max(__float_as_uint(fabsf(v.y)), max(__float_as_uint(fabsf(v.z)), __float_as_uint(fabsf(v.w)))));
  };
  size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {                // four independent 16-byte loads in flight
    const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
    take(a); take(b); take(c); take(d);
  }
  for (; i < n4; i += stride) take(x4[i]);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, __float_as_uint(fabsf(x[(n4 << 2) + threadIdx.x])));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  __shared__ uint32_t red[8];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < 8; ++w) m = max(m, red[w]);
    if (m != 0) atomicMax(out, m);                               // order-independent: deterministic
  }
}

// ---- "amax buffers": [0] = bits of max |x| over the tensor, [1 + t] = bits of max |x| over ALL CHANNELS of position segment t of
// x viewed as (B, C, L): segments of `seg` consecutive positions, t = b * nseg + l / seg (include/pvcnn_hip.h).
// A workgroup owns SPB whole segments (<= 256 positions) of one sample; its four waves split the channels, a lane keeps the
// maxima of four consecutive positions; the waves and the positions of a segment meet in LDS (ds_max_u32).  Every table entry
// is written by exactly one workgroup (no global atomics, no memset); the global maximum is a second, tiny kernel.
constexpr int kAmaxBlock = 256;    // positions per workgroup (at most)

__host__ __device__ inline int amax_segs_per_block(int seg) { return seg >= kAmaxBlock ? 1 : kAmaxBlock / seg; }

__device__ __forceinline__ uint32_t abs_bits(float v) { return __float_as_uint(fabsf(v)); }

__global__ __launch_bounds__(256) void absmax_tiles_kernel(const float *__restrict__ x, int C, long L, int seg, int nseg, int vec,
                                                           uint32_t *__restrict__ out, unsigned *__restrict__ ticket) {
  __shared__ uint32_t seg_max[kAmaxBlock];
  const int spb = amax_segs_per_block(seg);
  const int b = blockIdx.y, s0 = blockIdx.x * spb;             // first segment of this workgroup
  const long p0 = (long)s0 * seg;
  const int span = (int)min((long)spb * seg, L - p0);          // positions of this workgroup (<= 256 when seg <= 256)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  if (tid < spb) seg_max[tid] = 0u;
  __syncthreads();
  const float *xb = x + (size_t)b * C * L + p0;
  // seg > 256 (never used by the kernels of this library, kept general): the workgroup walks its one segment in 256-position steps
  for (int base = 0; base < span; base += kAmaxBlock) {
    const int pos = base + 4 * lane;
    uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    if (pos < span) {
      if (vec) {                                               // L % 4 == 0, seg % 4 == 0, x 16-byte aligned: whole quads in range
        int c = wave;
        for (; c + 12 < C; c += 16) {                          // four independent 16-byte loads in flight per lane
          const float4 a = *reinterpret_cast<const float4 *>(xb + (size_t)c * L + pos);
          const float4 d = *reinterpret_cast<const float4 *>(xb + (size_t)(c + 4) * L + pos);
          const float4 e = *reinterpret_cast<const float4 *>(xb + (size_t)(c + 8) * L + pos);
          const float4 f = *reinterpret_cast<const float4 *>(xb + (size_t)(c + 12) * L + pos);
          m0 = max(max(m0, abs_bits(a.x)), max(abs_bits(d.x), max(abs_bits(e.x), abs_bits(f.x))));
          m1 = max(max(m1, abs_bits(a.y)), max(abs_bits(d.y), max(abs_bits(e.y), abs_bits(f.y))));
          m2 = max(max(m2, abs_bits(a.z)), max(abs_bits(d.z), max(abs_bits(e.z), abs_bits(f.z))));
          m3 = max(max(m3, abs_bits(a.w)), max(abs_bits(d.w), max(abs_bits(e.w), abs_bits(f.w))));
        }
        for (; c < C; c += 4) {
          const float4 a = *reinterpret_cast<const float4 *>(xb + (size_t)c * L + pos);
          m0 = max(m0, abs_bits(a.x)); m1 = max(m1, abs_bits(a.y)); m2 = max(m2, abs_bits(a.z)); m3 = max(m3, abs_bits(a.w));
        }
      } else {
        for (int c = wave; c < C; c += 4) {
          const float *row = xb + (size_t)c * L + pos;
          m0 = max(m0, abs_bits(row[0]));
          if (pos + 1 < span) m1 = max(m1, abs_bits(row[1]));
          if (pos + 2 < span) m2 = max(m2, abs_bits(row[2]));
          if (pos + 3 < span) m3 = max(m3, abs_bits(row[3]));
        }
      }
      atomicMax(&seg_max[pos / seg], m0);
      if (pos + 1 < span) atomicMax(&seg_max[(pos + 1) / seg], m1);
      if (pos + 2 < span) atomicMax(&seg_max[(pos + 2) / seg], m2);
      if (pos + 3 < span) atomicMax(&seg_max[(pos + 3) / seg], m3);
    }
  }
  __syncthreads();
  // the global maximum out[0]: by the workgroup that takes the LAST ticket (ticket != NULL; common.h: ticket_take -- the table
  // entries are then published with returning atomics and peeked, no fences) instead of a one-workgroup launch behind this one
  // (absmax_tiles_reduce_kernel: ~5 us of launch latency, eleven times per PVCNN step)
  if (ticket == nullptr) {
    if (tid < spb && s0 + tid < nseg) out[1 + (size_t)b * nseg + s0 + tid] = seg_max[tid];
    return;
  }
  if (tid >= 64) return;                                     // wave 0 alone (spb <= 64: its lanes hold every entry of the workgroup)
  if (tid < spb && s0 + tid < nseg) publish32(out + 1 + (size_t)b * nseg + s0 + tid, seg_max[tid]);
  if (!ticket_take_wave(ticket, gridDim.x * gridDim.y)) return;
  amax_table_max_wave(out, (long)gridDim.y * nseg);
}

// out[0] = max over the table out[1 .. T] (one workgroup; T is a few thousand words)
__global__ __launch_bounds__(1024) void absmax_tiles_reduce_kernel(uint32_t *__restrict__ out, long T) {
  __shared__ uint32_t red[16];
  uint32_t m = 0;
  for (long i = threadIdx.x; i < T; i += 1024) m = max(m, out[1 + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < 16; ++w) m = max(m, red[w]);
    out[0] = m;
  }
}

// f16x2 weights: one workgroup per (padded) output channel finds the row maximum, then writes the row's hi / lo planes in the
// image layout below and the row's shift to wexp[co].
__device__ __forceinline__ void conv3d_weight_split_f16_row(const float *__restrict__ w, int Co, int Ci, int for_bwd_data,
                                                            uint16_t *__restrict__ wts, int *__restrict__ wexp, int co) {
  const int CiE = for_bwd_data ? Co : Ci, CoE = for_bwd_data ? Ci : Co;
  const int chunks = ceil_div(CiE, 16), cotiles = ceil_div(CoE, 64);
  const int cot = co >> 6, co_l = co & 63, tid = threadIdx.x;
  auto load = [&](int ci, int tap) { return for_bwd_data ? w[((size_t)ci * Ci + co) * 27 + (26 - tap)] : w[((size_t)co * Ci + ci) * 27 + tap]; };
  __shared__ uint32_t red[4];
  uint32_t m = 0;
  if (co < CoE)
    for (int i = tid; i < CiE * 27; i += 256) m = max(m, __float_as_uint(fabsf(load(i / 27, i % 27))));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  const int shift = scale_shift(max(max(red[0], red[1]), max(red[2], red[3])));
  if (tid == 0) wexp[co] = shift;
  const float scale = exp2_int(shift);
  for (int i = tid; i < chunks * 27 * 8; i += 256) {             // item = (chunk, tap, channel pair)
    const int cp = i & 7, tap = (i >> 3) % 27, chunk = (i >> 3) / 27;
    const int ci = chunk * 16 + 2 * cp, dxy = tap / 3, dz = tap - dxy * 3;
    const float a = (co < CoE && ci < CiE) ? load(ci, tap) * scale : 0.0f;
    const float b = (co < CoE && ci + 1 < CiE) ? load(ci + 1, tap) * scale : 0.0f;
    uint32_t p[2];
    split_pair<2>(a, b, p);
    const int pos = (((cp >> 2) ^ ((co_l >> 3) & 1)) * 8 + 2 * (cp & 3)) >> 1;     // word inside the 16-channel row
    const size_t blk = (((size_t)chunk * 9 + dxy) * cotiles + cot) * (3 * 2 * 64 * 16);
    uint32_t *img = reinterpret_cast<uint32_t *>(wts + blk);
#pragma unroll
    for (int s = 0; s < 2; ++s) img[((size_t)(dz * 2 + s) * 64 + co_l) * 8 + pos] = p[s];
  }
}

__global__ __launch_bounds__(256) void conv3d_weight_split_f16_kernel(const float *__restrict__ w, int Co, int Ci, int for_bwd_data,
                                                                      uint16_t *__restrict__ wts, int *__restrict__ wexp) {
  conv3d_weight_split_f16_row(w, Co, Ci, for_bwd_data, wts, wexp, blockIdx.x);
}

// forward AND backward-data image of one weight in one launch (a training step needs both; the weights do not change in between)
__global__ __launch_bounds__(256) void conv3d_weight_split_f16_pair_kernel(const float *__restrict__ w, int Co, int Ci, int rows_fwd,
                                                                           uint16_t *__restrict__ wts_f, int *__restrict__ wexp_f,
                                                                           uint16_t *__restrict__ wts_b, int *__restrict__ wexp_b) {
  if ((int)blockIdx.x < rows_fwd) conv3d_weight_split_f16_row(w, Co, Ci, 0, wts_f, wexp_f, blockIdx.x);
  else conv3d_weight_split_f16_row(w, Co, Ci, 1, wts_b, wexp_b, blockIdx.x - rows_fwd);
}

// ... of EVERY registered weight of a model in one launch (the weights change once per optimizer step: 7 launches of this kind per PVCNN
// step, 25 per PVCNN++ step become one).  The entry of a workgroup: a scan of the row_begin column (uniform scalar loads).
__global__ __launch_bounds__(256) void conv3d_weight_split_f16_batch_kernel(const SplitEntry *__restrict__ tab, int n) {
  const long long blk = blockIdx.x;
  int i = 0;
  while (i + 1 < n && tab[i + 1].row_begin <= blk) ++i;
  const SplitEntry e = tab[i];
  const int row = (int)(blk - e.row_begin);
  if (row < (int)e.rows_f) conv3d_weight_split_f16_row(e.w, (int)e.Co, (int)e.Ci, 0, e.wts_f, e.wexp_f, row);
  else conv3d_weight_split_f16_row(e.w, (int)e.Co, (int)e.Ci, 1, e.wts_b, e.wexp_b, row - (int)e.rows_f);
}

// ---- weights: (Co, Ci, 27) fp32 -> [chunk][dxy][cotile][dz][plane][64 co][16 ci (halves swizzled)] bf16 ----
// for_bwd_data: the convolution computed is grad_x = conv(grad_y, w') with Ci' = Co, Co' = Ci, w'[ci][co][tap] = w[co][ci][26 - tap]
template <int NS>
__device__ __forceinline__ void conv3d_weight_split_elem(const float *__restrict__ w, int Co, int Ci, int for_bwd_data,
                                                         uint16_t *__restrict__ wts, long e) {
  const int CiE = for_bwd_data ? Co : Ci, CoE = for_bwd_data ? Ci : Co;       // effective (reduction, output) channel counts
  const int chunks = ceil_div(CiE, kKc), cotiles = ceil_div(CoE, kCoTileB);
  const long total = (long)chunks * 9 * cotiles * 3 * kCoTileB * kKc;          // one thread per (.., dz, co, ci): writes NS planes
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

template <int NS>
__global__ __launch_bounds__(256) void conv3d_weight_split_kernel(const float *__restrict__ w, int Co, int Ci, int for_bwd_data,
                                                                  uint16_t *__restrict__ wts) {
  conv3d_weight_split_elem<NS>(w, Co, Ci, for_bwd_data, wts, (long)blockIdx.x * 256 + threadIdx.x);
}

// the plain-bf16 (torch.autocast) images of EVERY registered weight in one launch, forward and backward-data: the batched form of
// conv3d_weight_split_kernel<1> (the Frustum-PVCNN step issued 15 of those).  An entry's rows are 256-element blocks here: rows_f of
// the forward image first, then the backward-data image; wexp_* are unused (no per-row scale in this arithmetic).
__global__ __launch_bounds__(256) void conv3d_weight_split_bf16_batch_kernel(const SplitEntry *__restrict__ tab, int n) {
  const long long blk = blockIdx.x;
  int i = 0;
  while (i + 1 < n && tab[i + 1].row_begin <= blk) ++i;
  const SplitEntry e = tab[i];
  const long local = (long)(blk - e.row_begin);
  if (local < (long)e.rows_f) conv3d_weight_split_elem<1>(e.w, (int)e.Co, (int)e.Ci, 0, e.wts_f, local * 256 + threadIdx.x);
  else conv3d_weight_split_elem<1>(e.w, (int)e.Co, (int)e.Ci, 1, e.wts_b, (local - (long)e.rows_f) * 256 + threadIdx.x);
}

// Variants measured on (16,64,64,32^3), bf16x3 (fp32 kernel: 0.87 ms; 6x the MFMAs at 16x the rate = 0.28 ms at 2.4 GHz):
//   this one (256 threads, 2 workgroups per CU, 27 taps unrolled, weights one tap ahead)   0.56 ms
//   explicit one-tap software pipeline of both operands behind sched_barriers              0.58-0.61 ms (1 / 2 waves per SIMD)
//   producer / consumer wave specialisation, double-buffered tile (1 workgroup per CU)     0.62 ms
//   the same MFMA stream with NO loads and NO staging at all                                0.43 ms
// i.e. the matrix pipe itself sustains ~1.6 PF on random data here (the chip clocks down under a dense bf16 MFMA stream),
// and what is left above it is prologue / epilogue exposure; the simplest structure is kept.
// XCD-AWARE TILE ORDER (round 4).  The hardware hands consecutive workgroups of a grid to the eight XCDs in turn (block i -> XCD i % 8):
// with the plain order, the tiles of one y position of every x plane went to ONE XCD whenever tiles_y was a multiple of 8 -- and the
// tiles that have work on a voxelised cloud (the block of the room: x, y rows ~9 .. 22 of 32) are exactly a few y positions, so half
// the XCDs received nothing but zero-input tiles (measured: 75 % of the tiles skipped, launch time unchanged).  Here XCD x takes a
// CONTIGUOUS range of the tile list: the block's tiles are spread over all XCDs, and neighbouring tiles -- which share their halo
// rows and the weight image -- sit behind the same L2.  A bijection of [0, n): which tile a block computes, never whether.
__device__ __forceinline__ int xcd_tile_order(int bid, int n) {
  const int q = n >> 3, r = n & 7, x = bid & 7, k = bid >> 3;   // block bid is the k-th block of XCD x, which owns q (+1 if x < r) tiles
  return x * q + min(x, r) + k;
}

// ZERO-INPUT TILES (round 4).  The input of a PVConv's first convolution is a voxelised point cloud: a block of a room normalised into
// the unit ball fills ~14 % of the cube (S3DIS: 1.5 x 1.5 x 3 m), the rest of the grid is exact zeros -- and the amax buffer every
// f16x2 launch already reads says so: a workgroup whose halo tile has row maximum 0 multiplies zeros.  Its outputs are then bias
// (+0 + bias, as the full path would round it), its BatchNorm partial sums of (y - bias) are zero: written here without staging a
// row or issuing an MFMA.  Exact (not a tolerance): 0 * w accumulates to +0 in every fp32 accumulator.
template <int TX, int TY, int TZ>
__device__ __attribute__((noinline)) void write_zero_input_tile(float *__restrict__ yb, const float *__restrict__ bias, int Co, int co0, int R,
                                                      int x0, int y0, int z0, float2 *__restrict__ stats_part, int tid) {
  const size_t RR = (size_t)R * R, S = RR * R;
  const int rows = min(kCoTileB, Co - co0);
  if (R % 4 == 0 && z0 % 4 == 0) {
    constexpr int QZ = (TZ + 3) / 4;
    const int items = rows * TX * TY * QZ;
    for (int e = tid; e < items; e += 256) {
      const int q = e % QZ, yt = (e / QZ) % TY, xt = (e / (QZ * TY)) % TX, r = e / (QZ * TY * TX);
      const int gx = x0 + xt, gy = y0 + yt, gz = z0 + 4 * q;
      if (gx < R && gy < R && gz < R) {
        const float v = 0.0f + (bias != nullptr ? bias[co0 + r] : 0.0f);
        *reinterpret_cast<float4 *>(yb + (size_t)(co0 + r) * S + (size_t)gx * RR + (size_t)gy * R + gz) = make_float4(v, v, v, v);
      }
    }
  } else {
    const int items = rows * TX * TY * TZ;
    for (int e = tid; e < items; e += 256) {
      const int zt = e % TZ, yt = (e / TZ) % TY, xt = (e / (TZ * TY)) % TX, r = e / (TZ * TY * TX);
      const int gx = x0 + xt, gy = y0 + yt, gz = z0 + zt;
      if (gx < R && gy < R && gz < R) yb[(size_t)(co0 + r) * S + (size_t)gx * RR + (size_t)gy * R + gz] = 0.0f + (bias != nullptr ? bias[co0 + r] : 0.0f);
    }
  }
  if (stats_part != nullptr && tid < rows) stats_part[(size_t)(co0 + tid) * gridDim.x + blockIdx.x] = make_float2(0.0f, 0.0f);
}

// f16x2 operand scale: amax_seg = 0 -> one scale for the whole tensor (x_absmax[0]); amax_seg = R -> x_absmax is an "amax buffer"
// (include/pvcnn_hip.h) with one maximum per z row (b, gx, gy) behind the global one, and the workgroup scales ITS halo tile by the
// largest row it stages: an outlier somewhere in the grid costs precision only in the tiles that contain it.
template <int NS, int TX, int TY, int TZ, bool VEC, bool CO32 = false>
__global__ __launch_bounds__(256, (NS == 3 || TX * TY * TZ == 512 || TZ <= 16) ? 2 : 3) void conv3d_igemm_bf16_kernel(const float *__restrict__ x, const uint16_t *__restrict__ wts,
                                                                   const float *__restrict__ bias, float *__restrict__ y,
                                                                   int Ci, int Co, int R, int tiles_x, int tiles_y, int tiles_z,
                                                                   float2 *__restrict__ stats_part,
                                                                   const uint32_t *__restrict__ x_absmax, const int *__restrict__ wexp,
                                                                   int amax_seg) {
  static_assert(TX * TY * TZ == 64 || TX * TY * TZ == 128 || TX * TY * TZ == 256 || TX * TY * TZ == 512, "a workgroup tile is 4 waves x NBW x 32 voxels");
  constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2, HS = HX * HY * HZ;
  // Wave arrangement inside the 64-channel x (TX*TY*TZ)-voxel workgroup tile.  Every lane fetches its own A (weight) fragments from
  // global memory, once per tap.  With the four waves side by side along the voxels (WM = 1) each wave pulls all 64 rows: 4 KiB per
  // tap and wave through the CU's vector-memory path (64 B/clk) against 6 / 12 MFMAs at a 128- / 256-voxel tile -- at R = 16 the L1
  // path, not the matrix core, was the limit (0.33-0.40 of the MFMA peak whatever the prefetch depth).  WM = 2 arranges the waves
  // 2 x 2: a wave owns 32 channels x half the voxels, the A traffic halves and the B fragments (LDS reads, which have headroom)
  // double.  Only the 128-voxel tile is arranged so: at 256 voxels the wider B ring spills (measured by the compiler: 136-424 bytes
  // of scratch), and the 512-voxel tile has 24 MFMAs per A fetch anyway.  Per output element the products and their order are the same.
  // The 64-voxel tile (2 x 2 waves, ONE column block per wave) is for grids so small that larger tiles leave CUs idle (PVCNN++ at
  // R = 8, B = 8, 128 channels: 64 workgroups of 128 voxels on 256 CUs -- 57 us for 9 us of matrix work).
  // CO32 (Co <= 32: PVCNN++'s first stage, 32 channels at 32^3): only the first 32-row block of the 64-row weight tile exists -- the
  // second one would be MFMAs on padding, half of the kernel's matrix work.  Side-by-side waves (WM = 1) only.
  constexpr int VOX = TX * TY * TZ;
  constexpr int WM = VOX <= 128 ? 2 : 1, MBW = CO32 ? 1 : 2 / WM, NBW = VOX / (32 * (4 / WM));   // waves along the channels; row / column blocks per wave
  static_assert(!CO32 || WM == 1, "the 32-channel variant is for the tiles whose waves sit side by side");
  constexpr int WBLK = 3 * NS * kCoTileB * kKc;                 // bf16 elements of one (chunk, dxy, cotile) weight block
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
  uint32_t *xs = lds_u;                                         // [NS][HS][8] words (16 bf16 per voxel)

  int bid = __builtin_amdgcn_readfirstlane(xcd_tile_order(blockIdx.x, gridDim.x));
  const int tzi = bid % tiles_z; bid /= tiles_z;
  const int tyi = bid % tiles_y; bid /= tiles_y;
  const int txi = bid % tiles_x; bid /= tiles_x;
  const int b = bid;
  const int cot = blockIdx.y, co0 = cot * kCoTileB, cotiles = gridDim.y;
  const int x0 = txi * TX, y0 = tyi * TY, z0 = tzi * TZ;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int wm = wave % WM, wn = wave / WM;                     // this wave's channel group / voxel group
  const size_t RR = (size_t)R * R, S = RR * R;
  const float *xb = x + (size_t)b * Ci * S;
  const int chunks = ceil_div(Ci, kKc);
  int x_shift = 0;
  if constexpr (NS == 2) {
    uint32_t tm = 0;
    if (amax_seg > 0) {
      // max over the z rows (b, gx, gy) of this workgroup's halo tile.  All addresses are uniform (block index + kernel arguments):
      // scalar loads and s_max_u32 only -- the scale stays in a scalar register like the single-scale mode's one load.  (A
      // lane-parallel version with a wave reduction pushed the 256-register tiles into hundreds of spilled registers.)
      constexpr int LX = HX, LY = HY;
      const int lenx = min(LX, R), leny = min(LY, R);
      const int sx = min(max(x0 - 1, 0), R - lenx), sy = min(max(y0 - 1, 0), R - leny);   // clamped runs: a superset of the halo
      const uint32_t *tab = x_absmax + 1 + ((size_t)b * R + sx) * R + sy;
      for (int ix = 0; ix < lenx; ++ix)
        for (int iy = 0; iy < leny; ++iy) tm = max(tm, tab[(size_t)ix * R + iy]);
    } else {
      tm = *x_absmax;
    }
    if (tm == 0u) {                                             // (uniform: every thread of the workgroup leaves before the first barrier)
      write_zero_input_tile<TX, TY, TZ>(y + (size_t)b * Co * S, bias, Co, co0, R, x0, y0, z0, stats_part, tid);
      return;
    }
    x_shift = scale_shift(tm);
  }
  const float x_scale = exp2_int(x_shift);

  int hb[NBW];                                                  // halo index of this lane's output voxel (tap 0,0,0 corner)
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int m = wn * (32 * NBW) + nb * 32 + j;
    const int zt = m % TZ, yt = (m / TZ) % TY, xt = m / (TZ * TY);
    hb[nb] = (xt * HY + yt) * HZ + zt;
  }
  int a_off[MBW];                                               // A fragment word offsets inside one (dz, plane) slab
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) {
    const int row = (wm * MBW + mb) * 32 + j;
    a_off[mb] = row * 8 + ((kh ^ ((row >> 3) & 1)) * 4);
  }
  f32x16 acc[MBW][NBW];
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.0f;

  if constexpr (VEC) {                                          // the z halo (hz = 0 and hz = HZ - 1) is padding for every chunk
    for (int e = tid; e < NS * HX * HY * 2 * 8; e += 256) {
      const int w = e & 7, side = (e >> 3) & 1, row = (e >> 4) % (HX * HY), pl = (e >> 4) / (HX * HY);
      xs[pl * HS * 8 + (row * HZ + side * (HZ - 1)) * 8 + w] = 0u;
    }
  }
  for (int chunk = 0; chunk < chunks; ++chunk) {
    const int c0 = chunk * kKc;
    __syncthreads();                                            // previous chunk's fragment reads are done
    // ---- stage the halo tile, fp32 -> NS 16-bit planes, channels-last, swizzled ----
    if constexpr (VEC) {
      // R % 4 == 0 and the tile spans the whole z extent (z0 = 0, R <= TZ): a z row of one channel is R contiguous, 16-byte
      // aligned floats and its two halo voxels are padding (zeroed once, above).  item = (channel pair, hx, hy, z quad) with the
      // quad fastest across lanes: a wave reads whole 128-byte lines (the scalar path below touches 32 bytes of each line it
      // requests, and four times as many requests -- the L1's outstanding-request slots were what bounded this kernel).
      constexpr int QZ = TZ / 4, ITEMS = 8 * HX * HY * QZ, PER = (ITEMS + 255) / 256, ITER = PER > 4 ? 3 : PER, BATCHES = (PER + ITER - 1) / ITER;
#pragma unroll 1
      for (int batch = 0; batch < BATCHES; ++batch) {
        float4 va[ITER], vb[ITER];
#pragma unroll
        for (int u = 0; u < ITER; ++u) {
          const int e = (batch * ITER + u) * 256 + tid;
          const int q = e % QZ, hy = (e / QZ) % HY, hx = (e / (QZ * HY)) % HX, cp = e / (QZ * HY * HX);
          const int gx = x0 + hx - 1, gy = y0 + hy - 1, c = c0 + 2 * cp;
          va[u] = vb[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          const bool inside = e < ITEMS && (unsigned)gx < (unsigned)R && (unsigned)gy < (unsigned)R && 4 * q < R;
          if (inside) {
            const size_t off = (size_t)gx * RR + (size_t)gy * R + 4 * q;
            if (c < Ci) va[u] = *reinterpret_cast<const float4 *>(xb + (size_t)c * S + off);
            if (c + 1 < Ci) vb[u] = *reinterpret_cast<const float4 *>(xb + (size_t)(c + 1) * S + off);
          }
        }
#pragma unroll
        for (int u = 0; u < ITER; ++u) {
          const int e = (batch * ITER + u) * 256 + tid;
          if (e < ITEMS) {
            const int q = e % QZ, hy = (e / QZ) % HY, hx = (e / (QZ * HY)) % HX, cp = e / (QZ * HY * HX);
            const int v0 = (hx * HY + hy) * HZ + 1 + 4 * q;
            const float fa[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, fb[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int v = v0 + i;
              uint32_t pw[NS];
              if constexpr (NS == 2) split_pair<NS>(fa[i] * x_scale, fb[i] * x_scale, pw);
              else split_pair<NS>(fa[i], fb[i], pw);
              const int word = v * 8 + (((cp >> 2) ^ ((v >> 3) & 1)) * 4) + (cp & 3);
#pragma unroll
              for (int s = 0; s < NS; ++s) xs[s * HS * 8 + word] = pw[s];
            }
          }
        }
      }
    } else {   // any R: item = (halo voxel, channel pair), one float per channel
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
          const bool inside = e < HS * 8 && (unsigned)gx < (unsigned)R && (unsigned)gy < (unsigned)R && (unsigned)gz < (unsigned)R;
          if (inside) {
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
            uint32_t pw[NS];
            if constexpr (NS == 2) split_pair<NS>(va[u] * x_scale, vb[u] * x_scale, pw);
            else split_pair<NS>(va[u], vb[u], pw);
            const int word = v * 8 + (((cp >> 2) ^ ((v >> 3) & 1)) * 4) + (cp & 3);
#pragma unroll
            for (int s = 0; s < NS; ++s) xs[s * HS * 8 + word] = pw[s];
          }
        }
      }
    }
    __syncthreads();                                            // x tile staged
    // ---- weights: each lane fetches its own 16-byte A fragments straight from the pre-split image in global memory (a
    // slab is 64 rows x 32 B: the 64 lanes of a wave read 2 KiB contiguous, L2-resident -- the whole image is a few hundred
    // KiB shared by every workgroup).  No LDS copy of the weights, no barriers inside the tap loop.
    const uint4 *wblk = reinterpret_cast<const uint4 *>(wts + (((size_t)chunk * 9) * cotiles + cot) * WBLK);
    auto load_a = [&](int tap, uint4 (&af)[MBW][NS]) {
      const int dxy = tap / 3, dz = tap - dxy * 3;
      const uint4 *wq = wblk + (size_t)dxy * cotiles * (WBLK / 8);
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int s = 0; s < NS; ++s) af[mb][s] = wq[((dz * NS + s) * kCoTileB * 8 + a_off[mb]) >> 2];
    };
    // The weight fragments of the next AD taps are in flight during this tap's MFMAs.  AD = 1 hides them behind the other waves of
    // the SIMD; small grids (R <= 16) have only one or two waves per SIMD and a tap's MFMAs take 160-320 ns, far less than an L2
    // hit: there the ring is three taps deep (48 more VGPRs, which the 256-voxel tiles have).
    constexpr int AD = (TZ <= 16 && NS != 3) ? 3 : 1;
    constexpr bool PIN = AD > 1, BPRE = PIN;                    // (pinning the R = 32 tiles changes nothing: 3 waves per SIMD hide it)
    uint4 aq[AD + 1][MBW][NS], bq[2][NBW][NS];
    auto load_b = [&](int tap, uint4 (&bf)[NBW][NS]) {
      const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
      const int toff = (dx * HY + dy) * HZ + dz;
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        const int vi = hb[nb] + toff;
        const int boff = vi * 8 + ((kh ^ ((vi >> 3) & 1)) * 4);
#pragma unroll
        for (int s = 0; s < NS; ++s) bf[nb][s] = *reinterpret_cast<const uint4 *>(xs + s * HS * 8 + boff);
      }
    };
#pragma unroll
    for (int d = 0; d < AD; ++d) load_a(d, aq[d]);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      if (tap + AD < 27) load_a(tap + AD, aq[(tap + AD) % (AD + 1)]);
      if constexpr (BPRE) {                                     // ... and the input fragments (LDS) one tap ahead
        if (tap == 0) load_b(0, bq[0]);
        if (tap + 1 < 27) load_b(tap + 1, bq[(tap + 1) & 1]);
      } else {
        load_b(tap, bq[tap & 1]);
      }
      if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);     // or the scheduler sinks the loads to just before their use
      uint4 (&af)[MBW][NS] = aq[tap % (AD + 1)];
      uint4 (&bf)[NBW][NS] = bq[tap & 1];
      // consecutive MFMAs go to different accumulators (4 independent tiles between two partial products of one tile)
#define PVCNN_MFMA4(SA, SB)                                                                                              \
      _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb)                                                                 \
      _Pragma("unroll") for (int mb = 0; mb < MBW; ++mb)                                                                 \
        acc[mb][nb] = mfma16<NS>(af[mb][SA], bf[nb][SB], acc[mb][nb])
      if constexpr (NS == 1) {
        PVCNN_MFMA4(0, 0);
      } else if constexpr (NS == 2) {
        PVCNN_MFMA4(1, 0); PVCNN_MFMA4(0, 1);                         // lo x hi, hi x lo, then hi x hi
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
    const int m = wn * (32 * NBW) + nb * 32 + j;
    const int zt = m % TZ, yt = (m / TZ) % TY, xt = m / (TZ * TY);
    const int gx = x0 + xt, gy = y0 + yt, gz = z0 + zt;
    vok[nb] = gx < R && gy < R && gz < R;
    voff[nb] = (size_t)gx * RR + (size_t)gy * R + gz;
  }
  const bool want_stats = stats_part != nullptr;
  float2 *stat_lds = reinterpret_cast<float2 *>(lds_u);        // [4 / WM voxel groups][64 channels]
#pragma unroll
  for (int mbl = 0; mbl < MBW; ++mbl) {
    const int mb = wm * MBW + mbl;                              // 32-channel row block inside the 64-channel tile
    float bv[16], unscale[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      bv[r] = (bias != nullptr && co < Co) ? bias[co] : 0.0f;
      if constexpr (NS == 2) unscale[r] = exp2_int(-wexp[co]);   // wexp covers the padded rows of the tile
    }
    const float x_unscale = exp2_int(-x_shift);
    float ss[16], qq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ss[r] = qq[r] = 0.0f;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        float v = acc[mbl][nb][r];
        if constexpr (NS == 2) v = v * unscale[r] * x_unscale;  // powers of two: exact
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
      if ((j & 1) == 0) stat_lds[wn * kCoTileB + mb * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * kh] = make_float2(st, qt);
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < kCoTileB && co0 + tid < Co) {
      float2 t = stat_lds[tid];
#pragma unroll
      for (int w = 1; w < 4 / WM; ++w) { t.x += stat_lds[w * kCoTileB + tid].x; t.y += stat_lds[w * kCoTileB + tid].y; }
      stats_part[(size_t)(co0 + tid) * gridDim.x + blockIdx.x] = t;
    }
  }
}

// ---- the 128-voxel f16x2 tile (TX x TY x 16, R = 16 grids), pipelined ----------------------------------------------------------
// conv3d_igemm_bf16_kernel stages a chunk between two barriers -- request the rows, wait, convert, store, barrier -- and starts the 27
// taps with the weight fragments' latency exposed; with two workgroups of 162 MFMAs per wave and chunk on a CU that sequence was as
// long as the multiply phase (counters, round 3: MFMA pipe 49 % busy, 60 % of the LDS cycles bank conflicts of the 4-byte staging
// stores, whose lanes are 32 words apart).  Here:
//   * the tile is double-buffered (2 x 27 KiB) and a chunk costs ONE barrier: the rows of chunk c + 1 are converted and stored between
//     the MFMAs of taps 1..8 of chunk c (they were requested during the taps 24..26 of chunk c - 1 and sit in registers), the rows
//     of chunk c + 2 are requested at tap 24 -- behind the chunk's last weight-fragment request, so that the in-order wait for a
//     fragment never waits for a row;
//   * a staging thread owns 8 channels x 4 z voxels of one (x, y) row: 16-byte LDS stores of one voxel's four channel pairs, and the
//     z voxels of a quad are rotated by the quad index in LDS (position 4q + ((i + q) & 3)) so that the four quads of a row store
//     to four different bank groups;
//   * lanes are mapped to voxels by the hardware's ds_read_b128 service groups ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of a half
//     wave): each group reads one whole z row (16 consecutive voxels = 64 distinct banks);
//   * straight-line code: every load is issued unconditionally from a clamped address (rows outside the grid are scaled by zero;
//     Ci % 16 == 0 is required, other layers stay on conv3d_igemm_bf16_kernel), so the compiler's vmcnt waits leave the younger
//     loads in flight; weights and rows are addressed as uniform base + one 32-bit lane offset, LDS fragments as lane base + immediate.
// Same products in the same order per output element as conv3d_igemm_bf16_kernel<2, TX, TY, 16, true>.
template <int TX, int TY>
__global__ __launch_bounds__(256, 2) void conv3d_igemm_f16_pipe_kernel(const float *__restrict__ x, const uint16_t *__restrict__ wts,
                                                                       const float *__restrict__ bias, float *__restrict__ y, int Ci, int Co,
                                                                       int R, int tiles_x, int tiles_y, float2 *__restrict__ stats_part,
                                                                       const uint32_t *__restrict__ x_absmax, const int *__restrict__ wexp,
                                                                       int amax_seg) {
  constexpr int NS = 2, TZ = 16, HX = TX + 2, HY = TY + 2, HZ = TZ + 2, HS = HX * HY * HZ, TILE = NS * HS * 8;
  static_assert(TX * TY * TZ == 128, "4 waves, 2 x 2: 32 channels x 64 voxels each");
  constexpr int NBW = 2, WBLK = 3 * NS * kCoTileB * kKc;
  constexpr int ITEMS = 2 * HX * HY * (TZ / 4);                 // (channel octet, hx, hy, z quad)
  static_assert(ITEMS <= 256, "one staging item per thread");
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
  // two tiles [plane][channel half kh][halo voxel][4 words]: a lane's B fragment (8 channels of one voxel) is 16 bytes, 16 consecutive
  // voxels are 256 contiguous bytes (64 banks once), and a tap is a compile-time byte offset from the lane's base address
  unsigned char *xs = reinterpret_cast<unsigned char *>(lds_u);
  constexpr int TILEB = TILE * 4, HALFB = HS * 16;              // bytes of a tile / of one (plane, kh) slab

  int bid = xcd_tile_order(blockIdx.x, gridDim.x);
  const int tyi = bid % tiles_y; bid /= tiles_y;
  const int txi = bid % tiles_x; bid /= tiles_x;
  const int b = bid;
  const int cot = blockIdx.y, co0 = cot * kCoTileB, cotiles = gridDim.y;
  const int x0 = txi * TX, y0 = tyi * TY;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const size_t RR = (size_t)R * R, S = RR * R;
  const float *xb = x + (size_t)b * Ci * S;
  const int chunks = ceil_div(Ci, kKc);
  uint32_t tm = 0;
  if (amax_seg > 0) {                                           // max over the z rows of the halo tile: uniform scalar loads
    const int lenx = min(HX, R), leny = min(HY, R);
    const int sx = min(max(x0 - 1, 0), R - lenx), sy = min(max(y0 - 1, 0), R - leny);
    const uint32_t *tab = x_absmax + 1 + ((size_t)b * R + sx) * R + sy;
    for (int ix = 0; ix < lenx; ++ix)
      for (int iy = 0; iy < leny; ++iy) tm = max(tm, tab[(size_t)ix * R + iy]);
  } else {
    tm = *x_absmax;
  }
  if (tm == 0u) {                                               // zero-input tile: see write_zero_input_tile
    write_zero_input_tile<TX, TY, TZ>(y + (size_t)b * Co * S, bias, Co, co0, R, x0, y0, 0, stats_part, tid);
    return;
  }
  const int x_shift = scale_shift(tm);
  const float x_scale = exp2_int(x_shift);

  // lane -> voxel of a 32-voxel column block (two z rows): the two ds_read_b128 service groups of a half wave read one row each
  const int jj = j < 4 ? j : j < 12 ? 16 + (j - 4) : j < 16 ? j - 8 : j < 20 ? 24 + (j - 16) : j < 28 ? 8 + (j - 20) : 28 + (j - 28);
  // z position in LDS: the voxels of quad q rotated by q;  pos(hz) for hz = zt + dz, dz = 0, 1, 2
  auto zpos = [](int hz) { const int z = hz - 1; return hz == 0 ? 0 : 1 + ((z & ~3) | ((z + (z >> 2)) & 3)); };
  uint32_t bbase[NBW][3];                                       // byte offset of the lane's fragment for tap (0, 0, dz) inside plane 0
  const int zt = jj & 15;
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int m = wn * (32 * NBW) + nb * 32 + jj;
    const int yt = (m / TZ) % TY, xt = m / (TZ * TY);
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) bbase[nb][dz] = (uint32_t)(kh * HS + (xt * HY + yt) * HZ + zpos(zt + dz)) * 16u;
  }
  const int a_row = wm * 32 + j;
  const uint32_t a_off = (uint32_t)(a_row * 8 + ((kh ^ ((a_row >> 3) & 1)) * 4)) * 4u;      // bytes inside one (dz, plane) slab of the image
  f32x16 acc[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;

  // staging item of this thread: channel octet cg, halo row (hx, hy), z quad q
  const bool has_item = tid < ITEMS;
  const int e = has_item ? tid : 0;
  const int q = e & 3, hy = (e >> 2) % HY, hx = ((e >> 2) / HY) % HX, cg = (e >> 2) / (HY * HX);
  const int gx = x0 + hx - 1, gy = y0 + hy - 1;
  const bool inside = has_item && (unsigned)gx < (unsigned)R && (unsigned)gy < (unsigned)R && 4 * q < R;     // (R = 12: quad 3 is padding)
  const float item_scale = inside ? x_scale : 0.0f;
  // (uniform channel base + 32-bit per-thread offset: one address register for the eight rows)
  const uint32_t xoff = (uint32_t)(((size_t)cg * 8 * S + (size_t)min(max(gx, 0), R - 1) * RR + (size_t)min(max(gy, 0), R - 1) * R + min(4 * q, R - 4)) *
                                   sizeof(float));
  const uint32_t st0 = (uint32_t)(cg * HS + (hx * HY + hy) * HZ + 1 + 4 * q) * 16u;     // this thread's quad inside plane 0, bytes
  auto load_x = [&](int chunk, float4 (&v)[8]) {
    const int c0 = chunk * kKc;
#pragma unroll
    for (int k = 0; k < 8; ++k)                                   // Ci % 16 == 0 (host): no channel clamp
      v[k] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(xb + (size_t)(c0 + k) * S) + xoff);
  };
  auto convert_store = [&](const float4 (&v)[8], int i, float sc, int buf) {        // z voxel 4q + i of the item
    uint32_t w[NS][4];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
      const float a = i == 0 ? v[2 * k2].x : i == 1 ? v[2 * k2].y : i == 2 ? v[2 * k2].z : v[2 * k2].w;
      const float c = i == 0 ? v[2 * k2 + 1].x : i == 1 ? v[2 * k2 + 1].y : i == 2 ? v[2 * k2 + 1].z : v[2 * k2 + 1].w;
      uint32_t pw[NS];
      split_pair<NS>(a * sc, c * sc, pw);
      w[0][k2] = pw[0]; w[1][k2] = pw[1];
    }
    if (has_item) {
      unsigned char *dst = xs + buf * TILEB + st0 + (uint32_t)((i + q) & 3) * 16u;       // rotated inside the quad by q
#pragma unroll
      for (int s = 0; s < NS; ++s) *reinterpret_cast<uint4 *>(dst + s * 2 * HALFB) = make_uint4(w[s][0], w[s][1], w[s][2], w[s][3]);
    }
  };

  // the z halo (positions 0 and HZ - 1 of every row) is padding in both tiles
  for (int z = tid; z < 2 * NS * 2 * HX * HY * 2 * 4; z += 256) {
    const int w = z & 3, side = (z >> 2) & 1, row = (z >> 3) % (HX * HY), slab = (z >> 3) / (HX * HY);   // slab: (buffer, plane, kh)
    lds_u[(slab * HS + row * HZ + side * (HZ - 1)) * 4 + w] = 0u;
  }
  float4 xr[8];
  load_x(0, xr);
#pragma unroll
  for (int i = 0; i < 4; ++i) convert_store(xr, i, item_scale, 0);
  load_x(min(1, chunks - 1), xr);
  __syncthreads();

  constexpr int AD = 3;
  for (int chunk = 0; chunk < chunks; ++chunk) {
    const int d = chunk & 1;
    const unsigned char *xt_ = xs + d * TILEB;
    const float next_scale = chunk + 1 < chunks ? item_scale : 0.0f;
    const char *wblk = reinterpret_cast<const char *>(wts + (((size_t)chunk * 9) * cotiles + cot) * WBLK);
    auto load_a = [&](int tap, uint4 (&af)[NS]) {              // uniform tap base + the lane's 32-bit offset
      const int dxy = tap / 3, dz = tap - dxy * 3;
      const char *wq = wblk + ((size_t)dxy * cotiles * WBLK + (size_t)dz * NS * kCoTileB * kKc) * sizeof(uint16_t);
#pragma unroll
      for (int s = 0; s < NS; ++s) af[s] = *reinterpret_cast<const uint4 *>(wq + s * (kCoTileB * kKc * 2) + a_off);
    };
    auto load_b = [&](int tap, uint4 (&bf)[NBW][NS]) {
      const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int s = 0; s < NS; ++s)
          bf[nb][s] = *reinterpret_cast<const uint4 *>(xt_ + bbase[nb][dz] + (s * 2 * HALFB + (dx * HY + dy) * HZ * 16));
    };
    uint4 aq[AD + 1][NS], bq[2][NBW][NS];
#pragma unroll
    for (int t = 0; t < AD; ++t) load_a(t, aq[t]);
    load_b(0, bq[0]);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      if (tap + AD < 27) load_a(tap + AD, aq[(tap + AD) % (AD + 1)]);
      if (tap + 1 < 27) load_b(tap + 1, bq[(tap + 1) & 1]);
      if (tap == 24) load_x(min(chunk + 2, chunks - 1), xr);    // behind the chunk's last fragment request
      __builtin_amdgcn_sched_barrier(0);
      if (tap >= 1 && tap <= 7 && (tap & 1)) convert_store(xr, tap >> 1, next_scale, d ^ 1);      // taps 1, 3, 5, 7: z voxel 0..3
      uint4 (&af)[NS] = aq[tap % (AD + 1)];
      uint4 (&bf)[NBW][NS] = bq[tap & 1];
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) acc[nb] = mfma16<NS>(af[1], bf[nb][0], acc[nb]);       // lo x hi
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) acc[nb] = mfma16<NS>(af[0], bf[nb][1], acc[nb]);       // hi x lo
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) acc[nb] = mfma16<NS>(af[0], bf[nb][0], acc[nb]);       // hi x hi
    }
    __syncthreads();
  }

  // ---- epilogue: D[i = co][j = voxel]; lane -> voxel jj, register r -> co row ----
  float *yb = y + (size_t)b * Co * S;
  size_t voff[NBW];
  bool vok[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int m = wn * (32 * NBW) + nb * 32 + jj;
    const int yt = (m / TZ) % TY, xt = m / (TZ * TY);
    const int ox = x0 + xt, oy = y0 + yt;
    vok[nb] = ox < R && oy < R && zt < R;
    voff[nb] = (size_t)ox * RR + (size_t)oy * R + zt;
  }
  const bool want_stats = stats_part != nullptr;
  float2 *stat_lds = reinterpret_cast<float2 *>(lds_u);        // [2 voxel groups][64 channels]
  {
    const int mb = wm;
    float bv[16], unscale[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      bv[r] = (bias != nullptr && co < Co) ? bias[co] : 0.0f;
      unscale[r] = exp2_int(-wexp[co]);                         // wexp covers the padded rows of the tile
    }
    const float x_unscale = exp2_int(-x_shift);
    float ss[16], qq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ss[r] = qq[r] = 0.0f;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        float v = acc[nb][r] * unscale[r] * x_unscale;          // powers of two: exact
        if (want_stats) {
          const float m = vok[nb] ? v : 0.0f;
          ss[r] += m;
          qq[r] += m * m;
        }
        v += bv[r];
        if (vok[nb] && co < Co) yb[(size_t)co * S + voff[nb]] = v;
      }
    if (want_stats) {
      const float st2 = half_wave_sum16(ss, j), qt = half_wave_sum16(qq, j);
      const int rr = (j >> 1) & 15;
      if ((j & 1) == 0) stat_lds[wn * kCoTileB + mb * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * kh] = make_float2(st2, qt);
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < kCoTileB && co0 + tid < Co) {
      float2 t = stat_lds[tid];
      t.x += stat_lds[kCoTileB + tid].x; t.y += stat_lds[kCoTileB + tid].y;
      stats_part[(size_t)(co0 + tid) * gridDim.x + blockIdx.x] = t;
    }
  }
}

// ---- the 512-voxel f16x2 tile of the R = 32 grids, ONE workgroup per CU, persistent (round 6) -----------------------------------
// conv3d_igemm_bf16_kernel<2, 4, 4, 32> stages a 16-channel chunk of its halo tile between two barriers (request the rows in three
// batches, wait, convert, 4-byte LDS stores whose lanes are 32 words apart) and relies on the second workgroup of the CU to keep the
// matrix pipes busy meanwhile: by the counters they are busy in 0.47 of the cycles.  This kernel is pw_gemm_f16_wide_kernel's
// structure (pointwise_bf16.hip) on the implicit GEMM:
//   * the same 4 x 4 x 32 voxel x 64 channel tile and the same wave arrangement (a wave = 64 channels x the 128 voxels of one x plane
//     of the tile: per output element the same products in the same order -- bit-identical results, BatchNorm partials included), but
//     ONE workgroup per CU with the halo tile DOUBLE-buffered in LDS ([plane][channel half][halo voxel][8 channels = 16 bytes]: a B
//     fragment is one ds_read_b128, a tap an immediate offset, 32 consecutive z voxels 512 contiguous bytes) and ONE barrier per chunk;
//   * a chunk is 27 taps x 24 MFMAs per wave, every tap 24 PINNED slots (one MFMA + at most one request or a few vector-ALU
//     instructions): the B fragments of tap t + 1 (LDS) and the A fragments of tap t + 2 (weight image, L2) are requested between the
//     MFMAs of tap t; the rows of the NEXT chunk -- in registers since the previous chunk -- are converted and stored into the other
//     buffer between the MFMAs of taps 1 .. 12 (16-byte stores, packed arithmetic along z), and the rows of the chunk after that
//     are requested between the MFMAs of taps 13 .. 24: a whole chunk (~10 us) of latency budget per row;
//   * the workgroup is PERSISTENT over its tiles (XCD-contiguous ranges of the tile list, as xcd_tile_order): all request streams run
//     on into the next tile, whose first chunk is staged while this tile's last one is multiplied; an item's first MFMAs start from
//     C = 0; the epilogue's stores drain behind the next tile's MFMAs.
// Needs R == 32, Ci % 16 == 0, Ci >= 32, tensors below 4 GiB.  PVCNN_CONV_WIDE=0 keeps the two-workgroup kernel.
// TZ = 16 (the R = 16 grids; round 6, second step): the same kernel on a 4 x 4 x 16 tile -- a wave still owns one x plane of the tile,
// 64 channels x 64 voxels: two 32-voxel column blocks of two z rows each (lanes mapped to voxels by the hardware's ds_read_b128
// service groups, as in conv3d_igemm_f16_pipe_kernel), 12 MFMAs per tap.  The tile (and with it the scale tile) is not the
// 128-voxel tile of conv3d_igemm_f16_pipe_kernel: the same fp32-class result, not the same bits.
template <int TZ> struct CwGeom {
  static constexpr int HX = 6, HY = 6, HZ = TZ + 2, HS = HX * HY * HZ;             // halo of the 4 x 4 x TZ tile
  static constexpr int HALFB = HS * 16, TILEB = 4 * HALFB;                         // bytes of one (plane, half) slab / of one buffer
  static constexpr size_t LDS = (size_t)2 * TILEB + (size_t)4 * kCoTileB * sizeof(float2);
};

template <int TZ, int AB = 0>
__global__ __launch_bounds__(256, 1) void conv3d_igemm_f16_wide_kernel(const float *__restrict__ x, const uint16_t *__restrict__ wts,
                                                                       const float *__restrict__ bias, float *__restrict__ y, int Ci, int Co,
                                                                       int B, float2 *__restrict__ stats_part,
                                                                       const uint32_t *__restrict__ x_absmax, const int *__restrict__ wexp,
                                                                       int amax_seg, unsigned x_bytes, unsigned w_bytes, int stats_parts) {
  static_assert(TZ == 32 || TZ == 16, "the tile spans z: R = 32 or 16");
  using G = CwGeom<TZ>;
  constexpr int NS = 2, R = TZ, TX = 4, TY = 4, HX = G::HX, HY = G::HY, HZ = G::HZ, HS = G::HS, NBW = TZ / 8, MBW = 2;
  constexpr int HALFB = G::HALFB, TILEB = G::TILEB, WBLK = 3 * NS * kCoTileB * kKc;
  constexpr int RR = R * R, S = RR * R, tiles_x = R / TX, tiles_y = R / TY;
  constexpr int QZ = TZ / 4, NITEMS = 2 * HX * HY * QZ, NIT = (NITEMS + 255) / 256;       // staging items: 576 (3 per thread) / 288 (2)
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) uint32_t cw_lds[];
  unsigned char *xs8 = reinterpret_cast<unsigned char *>(cw_lds);
  float2 *stat_lds = reinterpret_cast<float2 *>(xs8 + 2 * TILEB);                 // [4 voxel groups][64 channels]

  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                     // = the x plane of the tile this wave owns
  // lane -> voxel of a 32-voxel column block.  TZ = 32: one z row, lane j = z.  TZ = 16: two z rows (y, y + 1); the two ds_read_b128
  // service groups of a half wave ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}) read one whole row each
  const int jj = TZ == 32 ? j : (j < 4 ? j : j < 12 ? 16 + (j - 4) : j < 16 ? j - 8 : j < 20 ? 24 + (j - 16) : j < 28 ? 8 + (j - 20) : 28 + (j - 28));
  const int vrow = TZ == 32 ? 0 : jj >> 4, vz = TZ == 32 ? jj : jj & 15;          // row inside the block, z
  constexpr int RPB = TZ == 32 ? 1 : 2;                                           // y rows per column block
  const int cotiles = ceil_div(Co, kCoTileB), chunks = Ci / kKc;
  const int n_tiles = B * tiles_x * tiles_y;
  // this workgroup's tiles: XCD x owns a contiguous range of the tile list (xcd_tile_order's), its workgroups take it round by round
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int xq = n_tiles >> 3, xr8 = n_tiles & 7;
  const int x_start = xcd * xq + min(xcd, xr8), x_count = xq + (xcd < xr8 ? 1 : 0);
  const int items_local = x_count * cotiles;
  if (slot >= items_local) return;
  const int rounds = (items_local - slot + nslots - 1) / nslots;

  auto descriptor = [](const void *base, uint32_t bytes) {
    const uintptr_t p = reinterpret_cast<uintptr_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t xrsrc = descriptor(x, x_bytes), wrsrc = descriptor(wts, w_bytes);

  // ---- an item = (tile, 64-channel block); what the request streams need of it ----
  // staging items of a thread: e = tid + 256 i, i < NIT (NITEMS = 2 channel octets x 36 halo rows x QZ z quads = 576 / 288; the last i
  // exists for tid < 64 / 32 only: the other threads stage their item 0 twice -- the same bytes to the same place -- so that the
  // chunk stays straight-line code)
  struct Item {
    int b, x0, y0, cot, stat_slot, shift;
    uint32_t x_base, a_base;         // bytes: x -> (cloud b, channel 0); image -> (chunk 0, dxy 0, cotile)
    float scale;
    uint32_t xoff[NIT];              // per thread: byte offset of (octet, clamped halo row, z quad) inside the cloud's chunk
    float sc[NIT];                   // per thread: the scale, or 0 for a row outside the grid
  };
  auto item_at = [&](int r) {
    Item it;
    const int jdx = slot + min(r, rounds - 1) * nslots, tl = jdx / cotiles;
    it.cot = jdx - tl * cotiles;
    int p = x_start + tl;                                       // position in the permuted tile list
    it.stat_slot = tl * 8 + xcd;                                // = the block index of the two-workgroup kernel that computes this tile
    const int tyi = p % tiles_y; p /= tiles_y;
    const int txi = p % tiles_x;
    it.b = p / tiles_x;
    it.x0 = txi * TX; it.y0 = tyi * TY;
    uint32_t tm = 0;
    if (amax_seg > 0) {                                         // max over the z rows of the halo tile: uniform scalar loads
      const int sx = min(max(it.x0 - 1, 0), R - HX), sy = min(max(it.y0 - 1, 0), R - HY);
      const uint32_t *tab = x_absmax + 1 + ((size_t)it.b * R + sx) * R + sy;
      for (int ix = 0; ix < HX; ++ix)
        for (int iy = 0; iy < HY; ++iy) tm = max(tm, tab[(size_t)ix * R + iy]);
    } else {
      tm = *x_absmax;
    }
    it.shift = __builtin_amdgcn_readfirstlane(scale_shift(tm)); // (a zero tile: shift 0, the products are exact zeros, y = bias)
    it.scale = exp2_int(it.shift);
    it.x_base = __builtin_amdgcn_readfirstlane((uint32_t)it.b * (uint32_t)Ci * (uint32_t)(S * 4));
    it.a_base = __builtin_amdgcn_readfirstlane((uint32_t)it.cot * (uint32_t)(WBLK * 2));
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int e = tid + 256 * i, ec = e < NITEMS ? e : tid;
      const int q = ec % QZ, row = (ec / QZ) % (HX * HY), cg = (ec / QZ) / (HX * HY), hy = row % HY, hx = row / HY;
      const int gx = it.x0 + hx - 1, gy = it.y0 + hy - 1;
      const bool in = (unsigned)gx < (unsigned)R && (unsigned)gy < (unsigned)R;
      it.sc[i] = in ? it.scale : 0.0f;
      it.xoff[i] = (uint32_t)((cg * 8) * S + min(max(gx, 0), R - 1) * RR + min(max(gy, 0), R - 1) * R + 4 * q) * 4u;
    }
    return it;
  };
  // LDS byte offset of staging item i's quad (plane 0, buffer 0), the same for every item
  uint32_t st_off[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int e = tid + 256 * i, ec = e < NITEMS ? e : tid;
    const int q = ec % QZ, row = (ec / QZ) % (HX * HY), cg = (ec / QZ) / (HX * HY);
    st_off[i] = (uint32_t)(cg * HS + row * HZ + 1 + 4 * q) * 16u;
  }
  // B fragment: the lane's voxel (row vrow of column block nb, z) of x plane `wave` of the tile; tap (dx, dy, dz) and nb are immediates
  const uint32_t b_off = (uint32_t)(kh * HS + (wave * HY + vrow) * HZ + vz) * 16u;
  // A fragment: row mb * 32 + j of the 64-row slab (bytes); the swizzle bit is bit 3 of the row
  const uint32_t a_off = (uint32_t)(j * 8 + ((kh ^ ((j >> 3) & 1)) * 4)) * 4u;
  const uint32_t a_chunk = (uint32_t)(9 * cotiles) * (uint32_t)(WBLK * 2);      // bytes from a chunk's block to the next chunk's

  auto load_a1 = [&](uint32_t soff, int tap, int plane, int mb) {          // (tap: compile-time)
    const int dxy = tap / 3, dz = tap - dxy * 3;
    return __builtin_amdgcn_raw_buffer_load_b128(wrsrc, a_off + (uint32_t)(mb * 32 * 32),
                                                 soff + (uint32_t)dxy * (uint32_t)(cotiles * WBLK * 2) + (uint32_t)((dz * NS + plane) * kCoTileB * kKc * 2), 0);
  };
  auto load_b1 = [&](int buf, int tap, int plane, int nb) {
    const int dx = tap / 9, dy = (tap / 3) % 3, dz = tap % 3;
    return *reinterpret_cast<const u32x4 *>(xs8 + buf * TILEB + plane * 2 * HALFB + b_off + ((dx * HY + RPB * nb + dy) * HZ + dz) * 16);
  };
  auto load_x1 = [&](uint32_t voff, uint32_t soff, int k) {               // channel k of the item's octet
    const u32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, soff + (uint32_t)k * (uint32_t)(S * 4), 0);
    return make_float4(__uint_as_float(r4.x), __uint_as_float(r4.y), __uint_as_float(r4.z), __uint_as_float(r4.w));
  };
  auto mma = [](const u32x4 &a, const u32x4 &b, const f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  };
  // conversion of one UNIT = staging item i, z voxels 2 h2 and 2 h2 + 1 (8 channels each), in 12 PIECES that are issued between
  // different MFMAs: pieces 0 .. 3 scale + round channel pair k2 to its hi fp16 pairs, 4 .. 7 the residuals' fp16 pairs, 8 .. 11 one
  // 16-byte store each (voxel, plane).  split_pair's arithmetic, packed along z (two neighbouring voxels of a channel sit in
  // neighbouring registers of the 16-byte load); the fused multiply-subtract a * scale - hi rounds once and is exact.
  f16x2 ch[2][4];                                               // [voxel][channel pair]: hi pairs of the unit in flight
  uint32_t cl[2][4];                                            // ... and its lo pairs
  auto conv_piece = [&](const float4 (&v)[8], float scale, int buf, uint32_t st, int h2, int piece) {
    const f32x2 sv = {scale, scale};
    if (piece < 8) {
      const int k2 = piece & 3;
      const f32x2 a = h2 == 0 ? f32x2{v[2 * k2].x, v[2 * k2].y} : f32x2{v[2 * k2].z, v[2 * k2].w};
      const f32x2 c = h2 == 0 ? f32x2{v[2 * k2 + 1].x, v[2 * k2 + 1].y} : f32x2{v[2 * k2 + 1].z, v[2 * k2 + 1].w};
      if (piece < 4) {
        const f32x2 sa = a * sv, sc2 = c * sv;
        ch[0][k2] = __builtin_convertvector(f32x2{sa[0], sc2[0]}, f16x2);
        ch[1][k2] = __builtin_convertvector(f32x2{sa[1], sc2[1]}, f16x2);
      } else {
        const f32x2 ha = {(float)ch[0][k2][0], (float)ch[1][k2][0]}, hc = {(float)ch[0][k2][1], (float)ch[1][k2][1]};
        const f32x2 ra = __builtin_elementwise_fma(a, sv, -ha), rc = __builtin_elementwise_fma(c, sv, -hc);
        cl[0][k2] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{ra[0], rc[0]}, f16x2));
        cl[1][k2] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{ra[1], rc[1]}, f16x2));
      }
    } else {
      const int e = (piece - 8) & 1, plane = (piece - 8) >> 1;
      unsigned char *dst = xs8 + buf * TILEB + st + (2 * h2 + e) * 16 + plane * 2 * HALFB;
      if (plane == 0)
        *reinterpret_cast<u32x4 *>(dst) = u32x4{__builtin_bit_cast(uint32_t, ch[e][0]), __builtin_bit_cast(uint32_t, ch[e][1]),
                                                __builtin_bit_cast(uint32_t, ch[e][2]), __builtin_bit_cast(uint32_t, ch[e][3])};
      else
        *reinterpret_cast<u32x4 *>(dst) = u32x4{cl[e][0], cl[e][1], cl[e][2], cl[e][3]};
    }
  };

  // ---- prologue: the z halo of both buffers is padding for good; chunk 0 of item 0 staged, chunk 1 in registers ----
  for (int z = tid; z < 2 * 4 * HX * HY * 2 * 4; z += 256) {
    const int w = z & 3, side = (z >> 2) & 1, row = (z >> 3) % (HX * HY), slab = (z >> 3) / (HX * HY);       // slab: (buffer, plane, kh)
    cw_lds[(slab * HS + row * HZ + side * (HZ - 1)) * 4 + w] = 0u;
  }
  Item cur = item_at(0), nxt = item_at(1);
  float4 xv[NIT][8];                                            // the rows of the next chunk to convert
#pragma unroll
  for (int i = 0; i < NIT; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) xv[i][k] = load_x1(cur.xoff[i], cur.x_base, k);
#pragma unroll
  for (int i = 0; i < NIT; ++i)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int piece = 0; piece < 12; ++piece) conv_piece(xv[i], cur.sc[i], 0, st_off[i], h2, piece);
  {
    const bool n1 = 1 >= chunks;                                // (chunks >= 2: never; kept for the form of the stream)
    const uint32_t soff = (n1 ? nxt.x_base : cur.x_base) + (uint32_t)(n1 ? 0 : 1) * (uint32_t)(kKc * S * 4);
#pragma unroll
    for (int i = 0; i < NIT; ++i)
#pragma unroll
      for (int k = 0; k < 8; ++k) xv[i][k] = load_x1(n1 ? nxt.xoff[i] : cur.xoff[i], soff, k);
  }
  u32x4 af[3][NS][MBW], bf[2][NS][NBW];                          // A ring: taps t, t + 1, t + 2; B: taps t, t + 1
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb) af[t][s2][mb] = load_a1(cur.a_base, t, s2, mb);
  lds_barrier();
  int buf = 0;                                                  // the buffer the current chunk multiplies

  f32x16 acc[MBW][NBW];                                         // (every element is defined by the first chunk of an item)
  // One chunk: 27 taps x 24 pinned slots.  FIRST: the item's first chunk (tap 0 starts from C = 0).
  //   tap t, slots 0 .. 7   lo x hi   + the B fragments of tap t + 1 (8 LDS reads)
  //          slots 8 .. 15  hi x lo   + the A fragments of tap t + 2 (4 image reads), + the staging work of the tap (below)
  //          slots 16 .. 23 hi x hi   + the staging work of the tap
  //   staging: taps 1 .. 12 convert + store the six (item, z pair) units of the next chunk's rows, one unit per two taps, six of
  //   its twelve pieces per tap; taps 13 .. 24 request the rows of the chunk after the next (24 loads, 2 per tap)
  auto chunk_body = [&](auto first_tag, int chunk) {
    constexpr bool FIRST = decltype(first_tag)::value;
    // streams: the NEXT chunk's rows are in xv (converted here, stored into buffer buf ^ 1 with the scale of ITS item); the rows
    // requested here are those of chunk + 2; the A fragments of taps 25, 26 request taps 0, 1 of chunk + 1
    const bool n1 = chunk + 1 >= chunks, n2 = chunk + 2 >= chunks;
    const uint32_t a_cur = cur.a_base + (uint32_t)chunk * a_chunk;
    const uint32_t a_nxt = n1 ? nxt.a_base : a_cur + a_chunk;
    const uint32_t x_soff = (n2 ? nxt.x_base : cur.x_base) + (uint32_t)(chunk + 2 - (n2 ? chunks : 0)) * (uint32_t)(kKc * S * 4);
    float csc[NIT];
    uint32_t cxo[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) { csc[i] = n1 ? nxt.sc[i] : cur.sc[i]; cxo[i] = n2 ? nxt.xoff[i] : cur.xoff[i]; }
    // B fragments of tap 0: behind the barrier that published this chunk's tile
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) bf[0][s2][nb] = load_b1(buf, 0, s2, nb);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int GS = 2 * NBW;                                 // MFMAs (= slots) per product group: 8 / 4
    constexpr int PPT = TZ == 32 ? 6 : 4;                       // conversion pieces per tap (taps 1 .. 12: NIT x 2 units x 12 pieces)
    constexpr int ROW_TAPS = NIT * 8 / 2;                       // taps 13 .. : two row requests each
    // one conversion piece: index l of the tap's PPT
    auto tap_piece = [&](int tap, int l) {
      if constexpr (!(AB & 4)) {
        if (tap >= 1 && tap <= 12) {
          const int pidx = (tap - 1) * PPT + l, u = pidx / 12, piece = pidx % 12;
          conv_piece(xv[u >> 1], csc[u >> 1], buf ^ 1, st_off[u >> 1], u & 1, piece);
        }
      }
    };
    auto tap_row = [&](int tap, int l2) {                       // one row request: index l2 of the tap's two
      if constexpr (!(AB & 2)) {
        if (tap >= 13 && tap < 13 + ROW_TAPS) {
          const int l = (tap - 13) * 2 + l2, it3 = l >> 3, k = l & 7;
          xv[it3][k] = load_x1(cxo[it3], x_soff, k);
        }
      }
    };
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int ac = tap % 3, an = (tap + 2) % 3, bc = tap & 1, bn = bc ^ 1;
#pragma unroll
      for (int i = 0; i < GS; ++i) {                            // ---- lo x hi  + the B fragments of tap + 1
        const int nb = i >> 1, mb = i & 1;
        if constexpr (FIRST) {
          if (tap == 0) acc[mb][nb] = mma(af[ac][1][mb], bf[bc][0][nb], f32x16{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f});
          else acc[mb][nb] = mma(af[ac][1][mb], bf[bc][0][nb], acc[mb][nb]);
        } else {
          acc[mb][nb] = mma(af[ac][1][mb], bf[bc][0][nb], acc[mb][nb]);
        }
        if constexpr (!(AB & 8)) { if (tap + 1 < 27) bf[bn][i / NBW][i % NBW] = load_b1(buf, tap + 1 < 27 ? tap + 1 : 0, i / NBW, i % NBW); }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < GS; ++i) {                            // ---- hi x lo  + the A fragments of tap + 2
        const int nb = i >> 1, mb = i & 1;
        acc[mb][nb] = mma(af[ac][0][mb], bf[bc][1][nb], acc[mb][nb]);
        if constexpr (!(AB & 1)) {
          if (i < 4) {
            if (tap + 2 < 27) af[an][i >> 1][i & 1] = load_a1(a_cur, tap + 2 < 27 ? tap + 2 : 0, i >> 1, i & 1);
            else af[an][i >> 1][i & 1] = load_a1(a_nxt, tap + 2 - 27 >= 0 ? tap + 2 - 27 : 0, i >> 1, i & 1);
          }
        }
        if constexpr (TZ == 32) {
          if (i == 4 || i == 5) tap_piece(tap, i - 4);
          if (i >= 6) tap_row(tap, i - 6);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < GS; ++i) {                            // ---- hi x hi
        const int nb = i >> 1, mb = i & 1;
        acc[mb][nb] = mma(af[ac][0][mb], bf[bc][0][nb], acc[mb][nb]);
        if constexpr (TZ == 32) {
          if ((i & 1) == 0) tap_piece(tap, 2 + (i >> 1));
        } else {
          tap_piece(tap, i);
          if (i < 2) tap_row(tap, i);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (!(AB & 16)) lds_barrier();
    buf ^= 1;
  };

  for (int r = 0; r < rounds; ++r) {
    chunk_body(std::true_type{}, 0);
    for (int chunk = 1; chunk < chunks; ++chunk) chunk_body(std::false_type{}, chunk);
    // ---- the item's epilogue: D[i = co][j = voxel]; lane -> z of row (x plane = wave, y = nb); register q -> co row ----
    {
      int tid_e = tid;
      asm volatile("" : "+v"(tid_e));                           // (re-derived here: see pw_gemm_f16_wide_kernel)
      const int je = tid_e & 31, khe = (tid_e >> 5) & 1;
      const int jje = TZ == 32 ? je : (je < 4 ? je : je < 12 ? 16 + (je - 4) : je < 16 ? je - 8 : je < 20 ? 24 + (je - 16) : je < 28 ? 8 + (je - 20) : 28 + (je - 28));
      const int vrow_e = TZ == 32 ? 0 : jje >> 4, vz_e = TZ == 32 ? jje : jje & 15;
      const int co0 = cur.cot * kCoTileB;
      const bool want_stats = stats_part != nullptr;
      const float x_unscale = exp2_int(-cur.shift);
      const __amdgpu_buffer_rsrc_t yrsrc = descriptor(y + (size_t)cur.b * Co * S, (uint32_t)Co * (uint32_t)(S * 4));
      // the lane's byte offset of (co0 + 4 kh, x0 + wave, y0 + its row in the block, its z); row q of block mb: + (mb * 32 + rowq) * S * 4;
      // column block nb: + RPB * nb * R * 4
      const uint32_t yoff = (uint32_t)((co0 + 4 * khe) * S + (cur.x0 + wave) * RR + (cur.y0 + vrow_e) * R + vz_e) * 4u;
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb)
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
          float bv[8], unscale[8], ss[8], qq[8];
#pragma unroll
          for (int qi = 0; qi < 8; ++qi) {
            const int q = h8 * 8 + qi, co = co0 + mb * 32 + (q & 3) + 8 * (q >> 2) + 4 * khe;
            bv[qi] = (bias != nullptr && co < Co) ? bias[co] : 0.0f;
            unscale[qi] = exp2_int(-wexp[co]);                  // wexp covers the padded rows of the tile
            ss[qi] = qq[qi] = 0.0f;
          }
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int qi = 0; qi < 8; ++qi) {
              const int q = h8 * 8 + qi;
              float v = acc[mb][nb][q] * unscale[qi] * x_unscale;          // powers of two: exact
              if (want_stats) {                                 // statistics of (y - bias), see bn_finalize_kernel
                ss[qi] += v;
                qq[qi] += v * v;
              }
              v += bv[qi];
              // (rows >= Co of the last channel block: beyond the descriptor, dropped)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrsrc,
                                                    yoff + (uint32_t)((mb * 32 + (q & 3) + 8 * (q >> 2)) * (S * 4)) + (uint32_t)(RPB * nb * R * 4), 0, 0);
            }
          if (want_stats) {
            const float st2 = half_wave_sum8(ss, je), qt = half_wave_sum8(qq, je);
            const int q = h8 * 8 + ((je >> 2) & 7);
            if ((je & 3) == 0) stat_lds[wave * kCoTileB + mb * 32 + (q & 3) + 8 * (q >> 2) + 4 * khe] = make_float2(st2, qt);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      if (want_stats) {
        lds_barrier();
        if (tid_e < kCoTileB && co0 + tid_e < Co) {
          float2 t = stat_lds[tid_e];
#pragma unroll
          for (int w = 1; w < 4; ++w) { t.x += stat_lds[w * kCoTileB + tid_e].x; t.y += stat_lds[w * kCoTileB + tid_e].y; }
          // stats_parts = the slots per channel the caller allocated (pvcnn_conv3d_fwd_split_stats_parts: the two-workgroup
          // kernel's tile count -- twice n_tiles where a small batch takes its 256-voxel tile): the surplus slots are zeros
          stats_part[(size_t)(co0 + tid_e) * stats_parts + cur.stat_slot] = t;
          if (stats_parts > n_tiles) stats_part[(size_t)(co0 + tid_e) * stats_parts + n_tiles + cur.stat_slot] = make_float2(0.0f, 0.0f);
        }
      }
    }
    cur = nxt;
    nxt = item_at(r + 2);
  }
}

// Workgroup tile and staging path.  Vector staging (whole z rows as 16-byte loads) needs R % 4 == 0 and a tile that spans z:
// tz = 8 / 16 / 32 for R <= 8 / 16 / 32.  At 16 < R <= 32 a 512-voxel tile (a wave owns 64 channels x 128 voxels: every weight
// fragment feeds four MFMA column blocks, the halo overhead drops from 3.0x to 2.25x) when that still leaves two workgroups for
// every CU; the six-product bf16x3 mode has neither the registers nor the LDS for it.  (Measured, (16,64,64,32^3), f16x2:
// scalar staging 0.354 ms, vector 0.320 ms, vector + 512-voxel tile 0.309 ms; a (4,8,16) tile at R = 16 spills and loses.)
struct SplitTile { int tx, ty, tz; bool vec; };
static SplitTile split_tiles(int B, int Co, int R, int nsplit) {
  const bool vec = R % 4 == 0 && R <= 32;
  if (R <= 8) {  // tiny grids (PVCNN++ at R = 8, B = 8: 16 tiles of 256 voxels per 64 channels): halve the tile while the chip is not full
    const long per = (long)B * ceil_div(R, 8) * ceil_div(Co, kCoTileB);
    // (measured, f16x2 forward, (8,128,128,8): 38.5 -> 28.4 us; (8,256,256,8): 71.3 -> 61.6; PVCNN++ step 574.7 -> 581.0 clouds/s in one call)
    if (per * ceil_div(R, 2) < kNumCU) return SplitTile{1, 8, 8, vec};
    return per * ceil_div(R, 4) < kNumCU ? SplitTile{2, 8, 8, vec} : SplitTile{4, 8, 8, vec};
  }
  if (!vec) return {4, 4, 16, false};
  if (R <= 16)   // too few 256-voxel tiles to give every SIMD two waves (R = 16, B = 16: 256 per 64 channels): halve them
    return (long)B * ceil_div(R, 4) * ceil_div(R, 4) * ceil_div(Co, kCoTileB) < 768 ? SplitTile{2, 4, 16, true} : SplitTile{4, 4, 16, true};
  const bool big = nsplit != 3 && (long)B * ceil_div(R, 4) * ceil_div(R, 4) * ceil_div(Co, kCoTileB) >= 512;
  return big ? SplitTile{4, 4, 32, true} : SplitTile{2, 4, 32, true};
}

template <int TX, int TY>
static int launch_igemm_f16_pipe(const float *x, const uint16_t *wts, const float *bias, float *y, int B, int Ci, int Co, int R, hipStream_t s,
                                 float2 *stats_part, const uint32_t *x_absmax, const int *wexp, int amax_seg) {
  constexpr int HS = (TX + 2) * (TY + 2) * 18;
  const size_t lds = (size_t)2 * 2 * HS * 8 * sizeof(uint32_t);                  // two tiles of two planes
  const int tx = ceil_div(R, TX), ty = ceil_div(R, TY);
  hipLaunchKernelGGL((conv3d_igemm_f16_pipe_kernel<TX, TY>), dim3((unsigned)((long)B * tx * ty), ceil_div(Co, kCoTileB)), dim3(256), lds, s, x,
                     wts, bias, y, Ci, Co, R, tx, ty, stats_part, x_absmax, wexp, amax_seg);
  return check_launch("conv3d_igemm_f16_pipe");
}

template <int NS, int TX, int TY, int TZ, bool VEC, bool CO32 = false>
static int launch_igemm_bf16(const float *x, const uint16_t *wts, const float *bias, float *y, int B, int Ci, int Co, int R,
                             hipStream_t s, float2 *stats_part, const uint32_t *x_absmax = nullptr, const int *wexp = nullptr,
                             int amax_seg = 0) {
  constexpr int HS = (TX + 2) * (TY + 2) * (TZ + 2);
  const size_t lds = std::max((size_t)NS * HS * 8 * sizeof(uint32_t), (size_t)4 * kCoTileB * sizeof(float2));
  const int tx = ceil_div(R, TX), ty = ceil_div(R, TY), tz = ceil_div(R, TZ);
  auto k = conv3d_igemm_bf16_kernel<NS, TX, TY, TZ, VEC, CO32>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("conv3d(bf16): LDS attribute: %s", hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(k, dim3((unsigned)((long)B * tx * ty * tz), ceil_div(Co, kCoTileB)), dim3(256), lds, s, x, wts, bias, y,
                     Ci, Co, R, tx, ty, tz, stats_part, x_absmax, wexp, amax_seg);
  return check_launch("conv3d_igemm_bf16");
}

}  // namespace pvcnn

using namespace pvcnn;

static size_t weight_image_bytes(int CiE, int CoE, int nsplit) {
  return (size_t)ceil_div(CiE, kKc) * 9 * ceil_div(CoE, kCoTileB) * 3 * nsplit * kCoTileB * kKc * sizeof(uint16_t);
}

// nsplit: 1 = bf16, 3 = bf16x3, 2 = f16x2 (image followed by one int32 shift per padded output channel)
extern "C" size_t pvcnn_conv3d_weight_split_bytes(int Co, int Ci, int for_bwd_data, int nsplit) {
  if (Co <= 0 || Ci <= 0 || nsplit < 1 || nsplit > 3) return 0;
  const int CiE = for_bwd_data ? Co : Ci, CoE = for_bwd_data ? Ci : Co;
  return weight_image_bytes(CiE, CoE, nsplit) + (nsplit == 2 ? (size_t)ceil_div(CoE, kCoTileB) * kCoTileB * sizeof(int) : 0);
}

// out[0] = max over x of the bit pattern of |x| (0 for an empty tensor); the f16x2 convolution derives its input scale from it
extern "C" int pvcnn_absmax_bits(const float *x, size_t n, void *out, void *stream) {
  PVCNN_REQUIRE(out && (x || n == 0), "null pointer");
  PVCNN_REQUIRE(n == 0 || aligned16(x), "x must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(out, 0, sizeof(uint32_t), s);
  if (e != hipSuccess) { set_error("absmax: memset: %s", hipGetErrorString(e)); return (int)e; }
  if (n == 0) return 0;
  const unsigned grid = (unsigned)std::min<size_t>(512, (n / 4 + 2047) / 2048 + 1);
  hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(512), 0, s, x, n, static_cast<uint32_t *>(out));
  return check_launch("absmax");
}

int pvcnn::launch_amax_reduce(uint32_t *out, long T, hipStream_t s) {
  hipLaunchKernelGGL(absmax_tiles_reduce_kernel, dim3(1), dim3(1024), 0, s, out, T);
  return check_launch("absmax_tiles_reduce");
}

extern "C" size_t pvcnn_absmax_tiles_count(int B, long L, int seg) {
  if (B <= 0 || L <= 0 || seg <= 0) return 1;
  return 1 + (size_t)B * (size_t)((L + seg - 1) / seg);
}

// x (B, C, L) -> amax buffer `out` (pvcnn_absmax_tiles_count(B, L, seg) uint32): [0] global, [1 + b * nseg + l / seg] per segment
extern "C" int pvcnn_absmax_tiles(const float *x, int B, int C, long L, int seg, void *out, void *ticket, void *stream) {
  PVCNN_REQUIRE(out && B >= 0 && C >= 0 && L >= 0 && seg > 0, "bad argument");
  PVCNN_REQUIRE(B <= 65535, "batch > 65535");
  hipStream_t s = static_cast<hipStream_t>(stream);
  uint32_t *o = static_cast<uint32_t *>(out);
  if (B == 0 || C == 0 || L == 0) {
    hipError_t e = hipMemsetAsync(out, 0, pvcnn_absmax_tiles_count(B, L, seg) * sizeof(uint32_t), s);
    if (e != hipSuccess) { set_error("absmax_tiles: memset: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
  }
  PVCNN_REQUIRE(x, "null pointer");
  const long nseg = (L + seg - 1) / seg;
  PVCNN_REQUIRE(nseg <= 0x7fffffffL && (long)B * nseg <= 0x7fffffffL, "too many segments");
  const int spb = amax_segs_per_block(seg);
  const int vec = (L % 4 == 0) && (seg % 4 == 0) && aligned16(x);
  const bool table_only = ticket == PVCNN_TABLE_ONLY;       // (ABI v12) the consumers take the maximum from the table: no word [0], no launch for it
  if (table_only) ticket = nullptr;
  PVCNN_REQUIRE(!ticket || (reinterpret_cast<uintptr_t>(ticket) & 3) == 0, "ticket must be 4-byte aligned");
  if ((long)B * nseg > kFoldTableMax || spb > 64) ticket = nullptr;   // a long table is read faster by the 1024 threads of the reduce launch
  hipLaunchKernelGGL(absmax_tiles_kernel, dim3((unsigned)((nseg + spb - 1) / spb), B), dim3(256), 0, s, x, C, L, seg, (int)nseg, vec, o,
                     static_cast<unsigned *>(ticket));
  if (int rc = check_launch("absmax_tiles")) return rc;
  return ticket || table_only ? 0 : launch_amax_reduce(o, (long)B * nseg, s);
}

extern "C" int pvcnn_conv3d_weight_split(const float *w, int Co, int Ci, int for_bwd_data, int nsplit, void *wts, void *stream) {
  PVCNN_REQUIRE(w && wts && Co > 0 && Ci > 0, "bad argument");
  PVCNN_REQUIRE(nsplit >= 1 && nsplit <= 3, "nsplit must be 1 (bf16), 2 (f16x2) or 3 (bf16x3)");
  PVCNN_REQUIRE(aligned16(wts), "wts must be 16-byte aligned");
  const int CiE = for_bwd_data ? Co : Ci, CoE = for_bwd_data ? Ci : Co;
  if (nsplit == 2) {
    int *wexp = reinterpret_cast<int *>(static_cast<char *>(wts) + weight_image_bytes(CiE, CoE, 2));
    hipLaunchKernelGGL(conv3d_weight_split_f16_kernel, dim3(ceil_div(CoE, kCoTileB) * kCoTileB), dim3(256), 0, static_cast<hipStream_t>(stream),
                       w, Co, Ci, for_bwd_data, static_cast<uint16_t *>(wts), wexp);
    return check_launch("conv3d_weight_split_f16");
  }
  const long total = (long)ceil_div(CiE, kKc) * 9 * ceil_div(CoE, kCoTileB) * 3 * kCoTileB * kKc;
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (nsplit == 1) hipLaunchKernelGGL(conv3d_weight_split_kernel<1>, grid, dim3(256), 0, s, w, Co, Ci, for_bwd_data, static_cast<uint16_t *>(wts));
  else             hipLaunchKernelGGL(conv3d_weight_split_kernel<3>, grid, dim3(256), 0, s, w, Co, Ci, for_bwd_data, static_cast<uint16_t *>(wts));
  return check_launch("conv3d_weight_split");
}

// both f16x2 images of w (forward + backward-data) in one launch; buffers sized by pvcnn_conv3d_weight_split_bytes(.., 0 / 1, 2)
extern "C" int pvcnn_conv3d_weight_split_pair(const float *w, int Co, int Ci, void *wts_fwd, void *wts_bwd, void *stream) {
  PVCNN_REQUIRE(w && wts_fwd && wts_bwd && Co > 0 && Ci > 0, "bad argument");
  PVCNN_REQUIRE(aligned16(wts_fwd) && aligned16(wts_bwd), "images must be 16-byte aligned");
  const int rows_f = ceil_div(Co, kCoTileB) * kCoTileB, rows_b = ceil_div(Ci, kCoTileB) * kCoTileB;
  int *wexp_f = reinterpret_cast<int *>(static_cast<char *>(wts_fwd) + weight_image_bytes(Ci, Co, 2));
  int *wexp_b = reinterpret_cast<int *>(static_cast<char *>(wts_bwd) + weight_image_bytes(Co, Ci, 2));
  hipLaunchKernelGGL(conv3d_weight_split_f16_pair_kernel, dim3(rows_f + rows_b), dim3(256), 0, static_cast<hipStream_t>(stream), w, Co, Ci,
                     rows_f, static_cast<uint16_t *>(wts_fwd), wexp_f, static_cast<uint16_t *>(wts_bwd), wexp_b);
  return check_launch("conv3d_weight_split_pair");
}

// batched form of pvcnn_conv3d_weight_split_pair: `entry` (host, 10 int64) describes one weight and its two image buffers and returns
// the number of workgroups it takes; the caller sets word [9] (row_begin) to the running sum and copies the table to the device.
extern "C" long pvcnn_conv3d_weight_split_pair_entry(const float *w, int Co, int Ci, void *wts_fwd, void *wts_bwd, long long *entry) {
  if (!w || !wts_fwd || !wts_bwd || !entry || Co <= 0 || Ci <= 0 || !aligned16(wts_fwd) || !aligned16(wts_bwd)) return -1;
  const int rows_f = ceil_div(Co, kCoTileB) * kCoTileB, rows_b = ceil_div(Ci, kCoTileB) * kCoTileB;
  SplitEntry e;
  e.w = w;
  e.wts_f = static_cast<uint16_t *>(wts_fwd);
  e.wexp_f = reinterpret_cast<int *>(static_cast<char *>(wts_fwd) + weight_image_bytes(Ci, Co, 2));
  e.wts_b = static_cast<uint16_t *>(wts_bwd);
  e.wexp_b = reinterpret_cast<int *>(static_cast<char *>(wts_bwd) + weight_image_bytes(Co, Ci, 2));
  e.Co = Co; e.Ci = Ci; e.rows_f = rows_f; e.tm = 0; e.row_begin = 0;
  memcpy(entry, &e, sizeof(e));
  return rows_f + rows_b;
}

extern "C" int pvcnn_conv3d_weight_split_pair_batch(const void *table, int n, long total_rows, void *stream) {
  PVCNN_REQUIRE(n >= 0 && total_rows >= 0 && total_rows <= 0x7fffffffL, "bad size");
  if (n == 0 || total_rows == 0) return 0;
  PVCNN_REQUIRE(table && (reinterpret_cast<uintptr_t>(table) & 7) == 0, "null or misaligned table");
  hipLaunchKernelGGL(conv3d_weight_split_f16_batch_kernel, dim3((unsigned)total_rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const SplitEntry *>(table), n);
  return check_launch("conv3d_weight_split_pair_batch");
}

// ... and of the plain-bf16 images (nsplit = 1; buffers sized by pvcnn_conv3d_weight_split_bytes(.., 0 / 1, 1))
static long conv_bf16_image_blocks(int CiE, int CoE) {
  return ((long)ceil_div(CiE, kKc) * 9 * ceil_div(CoE, kCoTileB) * 3 * kCoTileB * kKc + 255) / 256;
}

extern "C" long pvcnn_conv3d_weight_split_pair_entry_bf16(const float *w, int Co, int Ci, void *wts_fwd, void *wts_bwd, long long *entry) {
  if (!w || !wts_fwd || !wts_bwd || !entry || Co <= 0 || Ci <= 0 || !aligned16(wts_fwd) || !aligned16(wts_bwd)) return -1;
  SplitEntry e;
  e.w = w;
  e.wts_f = static_cast<uint16_t *>(wts_fwd); e.wexp_f = nullptr;
  e.wts_b = static_cast<uint16_t *>(wts_bwd); e.wexp_b = nullptr;
  e.Co = Co; e.Ci = Ci; e.rows_f = conv_bf16_image_blocks(Ci, Co); e.tm = 0; e.row_begin = 0;
  memcpy(entry, &e, sizeof(e));
  return (long)e.rows_f + conv_bf16_image_blocks(Co, Ci);
}

extern "C" int pvcnn_conv3d_weight_split_pair_batch_bf16(const void *table, int n, long total_rows, void *stream) {
  PVCNN_REQUIRE(n >= 0 && total_rows >= 0 && total_rows <= 0x7fffffffL, "bad size");
  if (n == 0 || total_rows == 0) return 0;
  PVCNN_REQUIRE(table && (reinterpret_cast<uintptr_t>(table) & 7) == 0, "null or misaligned table");
  hipLaunchKernelGGL(conv3d_weight_split_bf16_batch_kernel, dim3((unsigned)total_rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const SplitEntry *>(table), n);
  return check_launch("conv3d_weight_split_pair_batch_bf16");
}

extern "C" size_t pvcnn_conv3d_fwd_split_stats_parts(int B, int Co, int R, int nsplit) {
  if (B <= 0 || Co <= 0 || R <= 0) return 0;
  const SplitTile t = split_tiles(B, Co, R, nsplit);
  return (size_t)B * ceil_div(R, t.tx) * ceil_div(R, t.ty) * ceil_div(R, t.tz);
}

// the launch shape conv3d_fwd_split_impl's dispatch below takes: (voxels per tile) << 8 | weight rows per tile
extern "C" int pvcnn_conv3d_fwd_split_route(int B, int Ci, int Co, int R, int nsplit) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || R <= 0 || nsplit < 1 || nsplit > 3) return 0;
  const SplitTile t = split_tiles(B, Co, R, nsplit);
  int tx = t.tx, ty = t.ty, tz = t.tz, rows = kCoTileB;
  if (t.tz == 16 && t.vec && t.tx == 2 && nsplit == 2 && Ci % kKc == 0) { tx = 2; ty = 4; tz = 16; }    // the pipelined kernel: same tile
  if (t.vec && t.tz == 32 && Co <= 32 && nsplit == 2) rows = 32;
  return ((tx * ty * tz) << 8) | rows;
}

// y = conv3d(x, w) + bias with the pre-split weights of pvcnn_conv3d_weight_split (forward layout: Ci, Co as given; backward-data:
// call with x = grad_y, Ci = the forward Co, Co = the forward Ci, bias = NULL and the for_bwd_data = 1 weights).
static int conv3d_fwd_split_impl(const float *x, const void *wts, const float *bias, int B, int Ci, int Co, int R, int nsplit,
                                 const void *x_absmax, int amax_seg, float *y, float *stats_part, void *stream) {
  PVCNN_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && R > 0, "bad size");
  PVCNN_REQUIRE(nsplit >= 1 && nsplit <= 3, "nsplit must be 1 (bf16), 2 (f16x2) or 3 (bf16x3)");
  PVCNN_REQUIRE(nsplit != 2 || x_absmax, "f16x2 needs the input's pvcnn_absmax_bits / pvcnn_absmax_tiles");
  PVCNN_REQUIRE(amax_seg == 0 || amax_seg == R, "amax_seg must be 0 (scalar scale) or R (one maximum per z row)");
  if (B == 0) return 0;
  PVCNN_REQUIRE(x && wts && y && aligned16(wts), "null or misaligned pointer");
  PVCNN_REQUIRE(!stats_part || (reinterpret_cast<uintptr_t>(stats_part) & 7) == 0, "stats_part must be 8-byte aligned");
  PVCNN_REQUIRE((long)R * R * R * (long)std::max(Ci, Co) <= 0x7fffffffL, "grid too large");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint16_t *w16 = static_cast<const uint16_t *>(wts);
  float2 *sp = reinterpret_cast<float2 *>(stats_part);
  const SplitTile t = split_tiles(B, Co, R, nsplit);
  PVCNN_REQUIRE(!t.vec || aligned16(x), "x must be 16-byte aligned");
  const uint32_t *am = static_cast<const uint32_t *>(x_absmax);
  const int *wexp = nsplit == 2 ? reinterpret_cast<const int *>(static_cast<const char *>(wts) + weight_image_bytes(Ci, Co, 2)) : nullptr;
#define PVCNN_IGEMM(NS, TX, TY, TZ, VEC) launch_igemm_bf16<NS, TX, TY, TZ, VEC>(x, w16, bias, y, B, Ci, Co, R, s, sp, am, wexp, amax_seg)
#define PVCNN_IGEMM_NS(TX, TY, TZ, VEC) (nsplit == 3 ? PVCNN_IGEMM(3, TX, TY, TZ, VEC) : nsplit == 2 ? PVCNN_IGEMM(2, TX, TY, TZ, VEC) : PVCNN_IGEMM(1, TX, TY, TZ, VEC))
#define PVCNN_IGEMM_BIG(TX, TY, TZ) (nsplit == 2 ? PVCNN_IGEMM(2, TX, TY, TZ, true) : PVCNN_IGEMM(1, TX, TY, TZ, true))
  // round 6: R = 32 / 16, whole 16-channel chunks: the persistent one-workgroup-per-CU kernel (conv3d_igemm_f16_wide_kernel)
  static const bool wide_on = [] { const char *e = getenv("PVCNN_CONV_WIDE"); return !(e && e[0] == '0'); }();
  // (R = 16: measured -- tools/calls_r06/r06_call10: 6 .. 9 % faster per launch than conv3d_igemm_f16_pipe_kernel, 44.8 / 80.2 / 144.0 us
  //  against 47.6 / 87.5 / 156.3 at 64 -> 64 / 64 -> 128 / 128 -> 128, but nothing in the step: 6.076 / 6.083 ms with, 6.066 / 6.060
  //  without -- one item per workgroup at Co = 64, nothing for the persistence to hide.  Opt-in: PVCNN_CONV_WIDE16=1; tested either way)
  static const bool wide16_on = [] { const char *e = getenv("PVCNN_CONV_WIDE16"); return e && e[0] == '1'; }();
  if (wide_on && nsplit == 2 && (R == 32 || (R == 16 && wide16_on)) && Co > 32 && Ci % kKc == 0 && Ci >= 2 * kKc &&
      (long)B * std::max(Ci, Co) * R * R * R * 4 < 0xffffffffL) {
    const int cotiles = ceil_div(Co, kCoTileB), n_tiles = B * (R / 4) * (R / 4);
    const long per_xcd = (long)((n_tiles + 7) / 8) * cotiles;
    const unsigned grid = 8u * (unsigned)std::min<long>(kNumCU / 8, per_xcd);
    const unsigned xb = (unsigned)((size_t)B * Ci * R * R * R * 4), wb = (unsigned)weight_image_bytes(Ci, Co, 2);
    const int parts = (int)pvcnn_conv3d_fwd_split_stats_parts(B, Co, R, nsplit);
#define PVCNN_CW_LAUNCH(TZV, ABV)                                                                                                    \
    do {                                                                                                                               \
      auto kw = conv3d_igemm_f16_wide_kernel<TZV, ABV>;                                                                                \
      const int lds_bytes = (int)CwGeom<TZV>::LDS;                                                                                     \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kw), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);   \
      if (e != hipSuccess) { set_error("conv3d(wide): LDS attribute: %s", hipGetErrorString(e)); return (int)e; }                      \
      hipLaunchKernelGGL(kw, dim3(grid), dim3(256), lds_bytes, s, x, w16, bias, y, Ci, Co, B, sp, am, wexp, amax_seg, xb, wb, parts);   \
    } while (0)
#ifdef PVCNN_ABLATE
    const char *ab_env = getenv("PVCNN_CONV_ABLATE");
    const int ab = ab_env ? atoi(ab_env) : 0;
    if (R == 32) {
      switch (ab) {
        case 1: PVCNN_CW_LAUNCH(32, 1); break;
        case 2: PVCNN_CW_LAUNCH(32, 2); break;
        case 4: PVCNN_CW_LAUNCH(32, 4); break;
        case 8: PVCNN_CW_LAUNCH(32, 8); break;
        case 16: PVCNN_CW_LAUNCH(32, 16); break;
        case 31: PVCNN_CW_LAUNCH(32, 31); break;
        default: PVCNN_CW_LAUNCH(32, 0);
      }
    } else {
      switch (ab) {
        case 31: PVCNN_CW_LAUNCH(16, 31); break;
        default: PVCNN_CW_LAUNCH(16, 0);
      }
    }
#else
    if (R == 32) PVCNN_CW_LAUNCH(32, 0); else PVCNN_CW_LAUNCH(16, 0);
#endif
#undef PVCNN_CW_LAUNCH
    return check_launch("conv3d_igemm_f16_wide");
  }
  if (t.tz == 8 && t.tx == 1) return t.vec ? PVCNN_IGEMM_NS(1, 8, 8, true) : PVCNN_IGEMM_NS(1, 8, 8, false);
  if (t.tz == 8 && t.tx == 2) return t.vec ? PVCNN_IGEMM_NS(2, 8, 8, true) : PVCNN_IGEMM_NS(2, 8, 8, false);
  if (t.tz == 8) return t.vec ? PVCNN_IGEMM_NS(4, 8, 8, true) : PVCNN_IGEMM_NS(4, 8, 8, false);
  if (!t.vec) return PVCNN_IGEMM_NS(4, 4, 16, false);
  // the pipelined 128-voxel kernel (f16x2 only).  Round 3, 64 -> 64 at 16^3 x 16: see profiles/ab/r03u_convbench.jsonl
  if (t.tz == 16 && t.tx == 2 && nsplit == 2 && Ci % kKc == 0) return launch_igemm_f16_pipe<2, 4>(x, w16, bias, y, B, Ci, Co, R, s, sp, am, wexp, amax_seg);
  if (t.tz == 16) return t.tx == 2 ? PVCNN_IGEMM_NS(2, 4, 16, true) : PVCNN_IGEMM_NS(4, 4, 16, true);
  if (Co <= 32 && nsplit == 2)     // a 32-row weight tile: no MFMAs on the padded half (f16x2, the default arithmetic, only)
    return t.tx == 4 ? launch_igemm_bf16<2, 4, 4, 32, true, true>(x, w16, bias, y, B, Ci, Co, R, s, sp, am, wexp, amax_seg)
                     : launch_igemm_bf16<2, 2, 4, 32, true, true>(x, w16, bias, y, B, Ci, Co, R, s, sp, am, wexp, amax_seg);
  return t.tx == 4 ? PVCNN_IGEMM_BIG(4, 4, 32) : PVCNN_IGEMM_NS(2, 4, 32, true);
#undef PVCNN_IGEMM_BIG
#undef PVCNN_IGEMM_NS
#undef PVCNN_IGEMM
}

extern "C" int pvcnn_conv3d_fwd_split(const float *x, const void *wts, const float *bias, int B, int Ci, int Co, int R, int nsplit,
                                      const void *x_absmax, int amax_seg, float *y, float *stats_part, void *stream) {
  return conv3d_fwd_split_impl(x, wts, bias, B, Ci, Co, R, nsplit, x_absmax, amax_seg, y, stats_part, stream);
}
