// Shared by the split-bf16 layer GEMMs (nnet_gemm_b3.hip: FP32 sources split on the fly; nnet_gemm_b3i.hip: sources already
// stored as operand images): vector types, tile constants and the epilogue.
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
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kB3BN = 256, kB3KS = 16;
constexpr int kB3FragBytes = 1024;                    // one 32 x 16 bf16 operand fragment
// LDS of the epilogue (nnet_b3_epilogue.inc): one 32-row slab at a pitch of kB3BN + 4 floats + bias / scale / offset of the tile's columns
constexpr size_t kB3EpiBytes = (size_t)(32 * (kB3BN + 4) + 3 * kB3BN) * sizeof(float);

// x = p1 + p2 + p3 (bf16 parts, round to nearest even), 8 values at a time
__device__ __forceinline__ void Split3(const f32x4 &lo, const f32x4 &hi, bf16x8 *p1, bf16x8 *p2, bf16x8 *p3) {
  const bf16x4 a1 = __builtin_convertvector(lo, bf16x4), b1 = __builtin_convertvector(hi, bf16x4);
  const f32x4 ra = lo - __builtin_convertvector(a1, f32x4), rb = hi - __builtin_convertvector(b1, f32x4);
  const bf16x4 a2 = __builtin_convertvector(ra, bf16x4), b2 = __builtin_convertvector(rb, bf16x4);
  const f32x4 sa = ra - __builtin_convertvector(a2, f32x4), sb = rb - __builtin_convertvector(b2, f32x4);
  const bf16x4 a3 = __builtin_convertvector(sa, bf16x4), b3 = __builtin_convertvector(sb, bf16x4);
  *p1 = __builtin_shufflevector(a1, b1, 0, 1, 2, 3, 4, 5, 6, 7);
  *p2 = __builtin_shufflevector(a2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
  *p3 = __builtin_shufflevector(a3, b3, 0, 1, 2, 3, 4, 5, 6, 7);
}


}  // namespace b3
}  // namespace rs
