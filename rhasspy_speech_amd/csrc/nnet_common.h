// Pieces shared by the GEMM kernels (nnet_kernels.hip: exact-FP32 MFMA; nnet_gemm_b3.hip: split-bf16 MFMA).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace rs {

__device__ __forceinline__ float ApplyStage(const EltStageDev &st, float v, int col) {
  switch (st.kind) {
    case 0: return v > 0.f ? v : 0.f;                                   // ReLU
    case 1: { float t = __fmul_rn(v, st.scale[col]); return __fadd_rn(t, st.offset[col]); }  // MulColsVec then AddVecToRows
    case 4: return __fmul_rn(v, st.alpha);
    default: return v;
  }
}

inline int GemmEpiMode(const GemmDev &d, int rows) {
  // fused-stage pattern of the epilogue: 0 none, 1 ReLU, 2 ReLU + per-column scale/offset (BatchNorm), 3 generic
  // (non-temporal stores for the 672 MB log-likelihood matrix were tried: no measurable difference)
  (void)rows;
  if (d.nstages == 0) return 0;
  if (d.nstages == 1 && d.stages[0].kind == 0) return 1;
  if (d.nstages == 2 && d.stages[0].kind == 0 && d.stages[1].kind == 1) return 2;
  return 3;
}


// nnet_gemm_b3.hip
bool GemmB3Usable(const GemmDev &d);
void LaunchGemmB3(const GemmDev &d, int rows, const int *row_ivec, hipStream_t s);
// nnet_gemm_b3i.hip: a layer with a folded residual (GemmDev::res) on a kernel that does not add it in its own epilogue
GemmDev GemmWithoutResidual(const GemmDev &d);
void LaunchResidualAdd(const GemmDev &d, int rows, hipStream_t s);
// nnet_gemm_b3i.hip (sources stored as operand images)
bool GemmB3IUsable(const GemmDev &d);
void LaunchGemmB3I(const GemmDev &d, int rows, hipStream_t s);
// nnet_gemm_b3j.hip: the same GEMM with both operands through LDS-DMA, hand-placed waits and a 256 x 256 tile (large launches)
bool GemmB3JUsable(const GemmDev &d, int rows);
bool GemmB3JSmallUsable(const GemmDev &d);                 // launches of 32-row tiles on GemmKernelB3J (nnet_gemm_b3j.hip)
void LaunchGemmB3JSmall(const GemmDev &d, int rows, hipStream_t s);
void LaunchGemmB3J(const GemmDev &d, int rows, hipStream_t s);

}  // namespace rs
