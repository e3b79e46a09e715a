// Wavefront-wide (64 lanes) reductions on CDNA4 built from DPP row rotations and v_readlane: no LDS traffic, no
// ds_bpermute chains.  Results are wave-uniform (they come back through SGPRs).
#pragma once
#include <hip/hip_runtime.h>

namespace rs {
namespace wv {

template <int CTRL>
__device__ __forceinline__ unsigned Dpp(unsigned v) {
  // EVERY LANE OF THE WAVE MUST BE ACTIVE where these reductions are called (wave-uniform control flow): a disabled source lane reads
  // as 0 under bound_ctrl.
  // (old = 0 with bound_ctrl: a row rotation gives every lane a source, so the result is the same -- and the compiler folds the
  // move into the operation that consumes it, v_min_u32_dpp instead of copy + nop + v_mov_b32_dpp + v_min_u32: two issue slots per
  // step of a reduction instead of four)
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// butterfly of row rotations inside each row of 16 lanes (row_ror:1,2,4,8), then the four row results via SGPRs
#define RS_WAVE_REDUCE(OP)                                                                                              \
  v = OP(v, Dpp<0x121>(v));                                                                                             \
  v = OP(v, Dpp<0x122>(v));                                                                                             \
  v = OP(v, Dpp<0x124>(v));                                                                                             \
  v = OP(v, Dpp<0x128>(v));                                                                                             \
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16); \
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48); \
  return OP(OP(a, b), OP(c, d));

__device__ __forceinline__ unsigned MinU(unsigned v) { RS_WAVE_REDUCE(min) }
__device__ __forceinline__ unsigned MaxU(unsigned v) { RS_WAVE_REDUCE(max) }
__device__ __forceinline__ unsigned AddU_(unsigned x, unsigned y) { return x + y; }
__device__ __forceinline__ int Sum(int x) {
  unsigned v = (unsigned)x;
  RS_WAVE_REDUCE(AddU_)
}
#undef RS_WAVE_REDUCE

// inclusive prefix sum over the 64 lanes (row shifts + row broadcasts; the same sequence as dd::WaveScanIncl, decode_common.h)
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ int DppM(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, BOUND); }
__device__ __forceinline__ int ScanIncl(int v) {
  v += DppM<0x111, 0xF, true>(v);      // row_shr:1
  v += DppM<0x112, 0xF, true>(v);      // row_shr:2
  v += DppM<0x114, 0xF, true>(v);      // row_shr:4
  v += DppM<0x118, 0xF, true>(v);      // row_shr:8
  v += DppM<0x142, 0xA, false>(v);     // row_bcast:15 into rows 1 and 3
  v += DppM<0x143, 0xC, false>(v);     // row_bcast:31 into rows 2 and 3
  return v;
}

// order-preserving float <-> unsigned maps (branch-free)
__device__ __forceinline__ float OrderedToFloat(unsigned u) {
  return __uint_as_float(u ^ ((unsigned)((int)~u >> 31) | 0x80000000u));
}
__device__ __forceinline__ unsigned FloatToOrdered(float f) {
  const unsigned b = __float_as_uint(f);
  return b ^ ((unsigned)((int)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float MaxF(float v) { return OrderedToFloat(MaxU(FloatToOrdered(v))); }
__device__ __forceinline__ float MinF(float v) { return OrderedToFloat(MinU(FloatToOrdered(v))); }

}  // namespace wv
}  // namespace rs
