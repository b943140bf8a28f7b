// split16.h -- fp32 operands as 16-bit pieces for the bf16 / fp16 matrix cores (shared by conv3d_bf16.hip and
// conv3d_wgrad_f16.hip; the arithmetic is described at the top of conv3d_bf16.hip).
#pragma once
#include "common.h"

namespace pvcnn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Two neighbouring channels at once, packed (first value in the low half): v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 round to nearest
// even in hardware.  NS = 1, 3: bf16 pieces as in split_bf16.  NS = 2: fp16 "hi + lo" of PRE-SCALED values (|v| < 2^15, see
// scale_shift): hi = fp16(v) keeps 11 bits, lo = fp16(v - hi) the next 11.
template <int NS>
__device__ __forceinline__ void split_pair(float a, float b, uint32_t (&w)[NS]) {
  f32x2 v = {a, b};
  if constexpr (NS == 2) {
    const f16x2 h = __builtin_convertvector(v, f16x2);
    w[0] = __builtin_bit_cast(uint32_t, h);
    v = v - __builtin_convertvector(h, f32x2);                   // exact
    w[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  } else {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      w[s] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
      if (s + 1 < NS) {
        const f32x2 back = {__uint_as_float(w[s] << 16), __uint_as_float(w[s] & 0xffff0000u)};
        v = v - back;                                            // exact: the residual fits fp32
      }
    }
  }
}

// fp16 has 5 exponent bits: operands of the f16x2 mode are scaled by a power of two that puts the largest magnitude of the
// tensor (bits of max |x|, from absmax_kernel; of a weight row, in the split kernel) into [2^13, 2^14).  Everything within
// 2^-17 of the maximum then keeps 22 bits in hi + lo; smaller elements lose low bits gradually (absolute error <= 2^-38 of
// the maximum).  Zero / inf / NaN maxima: no scaling (inf and NaN then propagate as they would in fp32).
__device__ __forceinline__ int scale_shift(uint32_t absmax_bits) {
  const int e = (int)((absmax_bits >> 23) & 0xffu);
  if (e == 0 || e == 255) return 0;
  return min(max(140 - e, -100), 100);                           // 13 - (e - 127)
}
__device__ __forceinline__ float exp2_int(int s) { return __uint_as_float((uint32_t)(s + 127) << 23); }

template <int NS>
__device__ __forceinline__ f32x16 mfma16(const uint4 &a, const uint4 &b, const f32x16 &c) {
  if constexpr (NS == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

}  // namespace pvcnn
