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
constexpr size_t kB3EpiBytes = (size_t)(32 * (kB3BN + 4) + 4 * kB3BN) * sizeof(float);
// |x| at or above this rounds to an fp16 infinity: the split cannot carry the value (kernels raise GemmDev::ovf, the host
// repeats the call on the exact-FP32 kernels)
constexpr float kB3Overflow = 65520.f;

// x = p1 + p2 up to 2^-22 |x| (fp16 parts, round to nearest even; p2 may be subnormal: the matrix cores keep fp16
// subnormal inputs, profiles/micro/mfma_f16_denorm.hip), 8 values at a time.  Returns max |x| of the eight.
__device__ __forceinline__ float Split2(const f32x4 &lo, const f32x4 &hi, f16x8 *p1, f16x8 *p2) {
  const f16x4 a1 = __builtin_convertvector(lo, f16x4), b1 = __builtin_convertvector(hi, f16x4);
  const f32x4 ra = lo - __builtin_convertvector(a1, f32x4), rb = hi - __builtin_convertvector(b1, f32x4);
  const f16x4 a2 = __builtin_convertvector(ra, f16x4), b2 = __builtin_convertvector(rb, f16x4);
  *p1 = __builtin_shufflevector(a1, b1, 0, 1, 2, 3, 4, 5, 6, 7);
  *p2 = __builtin_shufflevector(a2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
  const float m0 = fmaxf(fmaxf(fabsf(lo[0]), fabsf(lo[1])), fmaxf(fabsf(lo[2]), fabsf(lo[3])));
  const float m1 = fmaxf(fmaxf(fabsf(hi[0]), fabsf(hi[1])), fmaxf(fabsf(hi[2]), fabsf(hi[3])));
  return fmaxf(m0, m1);
}
__device__ __forceinline__ float Split2(const f32x4 &x, f16x4 *p1, f16x4 *p2) {
  *p1 = __builtin_convertvector(x, f16x4);
  *p2 = __builtin_convertvector(x - __builtin_convertvector(*p1, f32x4), f16x4);
  return fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
}

}  // namespace b3
}  // namespace rs
