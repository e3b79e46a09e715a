// Shared by the split-fp16 layer GEMMs (nnet_gemm_b3.hip: FP32 sources split on the fly; nnet_gemm_b3i.hip / nnet_gemm_b3j.hip:
// sources already stored as operand images): vector types, tile constants, the operand split and the order of the products.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "decode_common.h"
#include "kernels.h"
#include "nnet_common.h"

namespace rs {
namespace b3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kB3BN = 256, kB3KS = 16;
constexpr int kB3Parts = 2;                           // fp16 parts per FP32 operand (nnet_gemm_b3.hip)
constexpr int kB3FragBytes = 1024;                    // one 32 x 16 fp16 operand fragment
// LDS of the epilogue (nnet_b3_epilogue.inc): one 32-row slab at a pitch of kB3BN + 4 floats + bias / scale / offset / weight scale of the tile's columns
constexpr size_t kB3EpiBytes = (size_t)(32 * (kB3BN + 4) + 4 * kB3BN + 256) * sizeof(float);      // (+ the row maxima of up to 256 tile rows)
// |x| at or above this rounds to an fp16 infinity: the split cannot carry the value (kernels raise GemmDev::ovf, the host
// repeats the call on the exact-FP32 kernels).  The kernels test the SUM of |x| over the four / eight values split together
// (B3Over below): a sum propagates NaN and infinity where a maximum (fmaxf returns the operand that is a number) drops them, at the
// same instruction count; a group whose sum passes the bound while every member is below it sends the call to the exact kernels
// needlessly, which is correct.
// Small activations (round 6): the low part of |x| < 2^-3 is an fp16 subnormal, so the split carries a value to 2^-25 ABSOLUTE, not to
// 2^-22 relative (15 bits at |x| = 1e-3, nothing below 3e-8).  That is harmless while the operand row the value sits in has an element
// of magnitude >= 2^-3: the error is then below 2^-22 of the row's largest element, the same norm-wise bound the split gives a row of
// normal-range values, and far below what the order of an FP32 sum over the row moves (2^-24 sqrt(K) of the result).  It is NOT
// harmless when a whole row is small -- a network whose hidden activations are O(1e-3) would lose up to half of its bits with no sign of
// it.  So every kernel that splits activations also takes the maximum of |x| over the row (the tile's up to 256 columns of it; the
// layer's whole K for the kernel that splits FP32 sources inside its k loop) and raises GemmDev::ovf[1] for a row that is not all zero
// and has no element >= 2^-3 (B3Under); the host treats the flag like the overflow one: the call is repeated on the exact-FP32
// kernels (engine.cc: CheckGemmRange).  The TDNNs of the suite carry batch-norm'ed, ReLU'd activations of order 1 and never raise it.
constexpr float kB3Overflow = 65520.f;
__device__ __forceinline__ bool B3Over(float group_sum) { return !(group_sum < kB3Overflow); }
constexpr float kB3Tiny = 0.125f;                     // 2^-3: below it the low fp16 part is subnormal
__device__ __forceinline__ bool B3Under(float row_max) { return row_max > 0.f && row_max < kB3Tiny; }
// running maximum of |x| (v_max3_f32 with |.| source modifiers; a NaN operand is dropped here and caught by B3Over's sum)
__device__ __forceinline__ float B3AbsMax(float m, const f32x4 &x) { return fmaxf(fmaxf(m, fmaxf(fabsf(x[0]), fabsf(x[1]))), fmaxf(fabsf(x[2]), fabsf(x[3]))); }
// the larger of a value and its partner's 32 lanes away (the two k-groups of an image unit / the two column halves of an accumulator row)
__device__ __forceinline__ float B3MaxHalves(float m) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}

// x = p1 + p2 up to 2^-22 |x| (fp16 parts, round to nearest even; p2 may be subnormal: the matrix cores keep fp16
// subnormal inputs, profiles/micro/mfma_f16_denorm.hip), 8 values at a time.  Returns the sum of |x| over the eight (NaN / infinity propagate).
__device__ __forceinline__ float Split2(const f32x4 &lo, const f32x4 &hi, f16x8 *p1, f16x8 *p2) {
  const f16x4 a1 = __builtin_convertvector(lo, f16x4), b1 = __builtin_convertvector(hi, f16x4);
  const f32x4 ra = lo - __builtin_convertvector(a1, f32x4), rb = hi - __builtin_convertvector(b1, f32x4);
  const f16x4 a2 = __builtin_convertvector(ra, f16x4), b2 = __builtin_convertvector(rb, f16x4);
  *p1 = __builtin_shufflevector(a1, b1, 0, 1, 2, 3, 4, 5, 6, 7);
  *p2 = __builtin_shufflevector(a2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
  const float m0 = (fabsf(lo[0]) + fabsf(lo[1])) + (fabsf(lo[2]) + fabsf(lo[3]));
  const float m1 = (fabsf(hi[0]) + fabsf(hi[1])) + (fabsf(hi[2]) + fabsf(hi[3]));
  return m0 + m1;
}
__device__ __forceinline__ float Split2(const f32x4 &x, f16x4 *p1, f16x4 *p2) {
  *p1 = __builtin_convertvector(x, f16x4);
  *p2 = __builtin_convertvector(x - __builtin_convertvector(*p1, f32x4), f16x4);
  return (fabsf(x[0]) + fabsf(x[1])) + (fabsf(x[2]) + fabsf(x[3]));
}

}  // namespace b3
}  // namespace rs
