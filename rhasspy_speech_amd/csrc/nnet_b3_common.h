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

// Epilogue of a (32 MR WM) x 256 tile held as MR x 2 accumulator tiles of 32x32 per wave (4 waves side by side per wave row).
// The 32-row slabs of the tile are transposed through LDS one after the other; each leaves as 16-byte row-contiguous FP32
// stores (d.write_f32) and / or as the operand image the consuming layers read (d.out_img: three bf16 parts in A-fragment
// order, 16 bytes per (row, k-group) -- 512 consecutive bytes per 32 lanes).
template <int MR, bool MIXED, int WM>
__device__ __forceinline__ void Epilogue(f32x16 (&acc)[MR][2], const GemmDev &d, int rows, int row0, int n0, int mr_eff, int epi_mode,
                                         unsigned char *smem) {
  constexpr int RT = MR * WM, BN = kB3BN, NT = 256 * WM;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  Bias
  // and the fused stages are applied in registers; each 32-row slab of the tile is then transposed through LDS (free
  // after the loop; pitch 264 floats keeps both halves of a wave on different banks) and leaves as 16-byte row-contiguous
  // stores, 1 KiB per row and wave instruction -- storing straight from the accumulators (two rows x 128 bytes per
  // instruction) cost 40 us of a 200 us hidden layer.
  constexpr int C_LD = BN + 8;
  float *Cs = reinterpret_cast<float *>(smem);
  const bool vec_out = ((d.ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.out) & 15) == 0) && (((d.n + 3) & ~3) <= d.ldo);
  float bias[2], sc[2], of[2];
  int ccol[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int col = n0 + wn * 64 + j * 32 + (lane & 31);
    const bool cok = col < d.n;
    ccol[j] = cok ? col : 0;
    bias[j] = (d.bias && cok) ? d.bias[ccol[j]] : 0.f;
    sc[j] = 1.f; of[j] = 0.f;
    if (epi_mode == 2) { sc[j] = d.stages[1].scale[ccol[j]]; of[j] = d.stages[1].offset[ccol[j]]; }
  }
#pragma unroll
  for (int sl = 0; sl < RT; sl++) {                // 32-row slab sl of the tile belongs to wave row sl / MR
    constexpr int kDummy = 0; (void)kDummy;
    const int i = sl % MR;
    if (MIXED && sl >= mr_eff) break;              // workgroup-uniform
    if (wm == sl / MR) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int cl = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          float v = __fadd_rn(bias[j], acc[i][j][r]);
          if (epi_mode == 1) {
            v = v > 0.f ? v : 0.f;
          } else if (epi_mode == 2) {               // ReLU then BatchNorm (test mode): MulColsVec, AddVecToRows
            v = v > 0.f ? v : 0.f;
            v = __fadd_rn(__fmul_rn(v, sc[j]), of[j]);
          } else if (epi_mode == 3) {
            for (int st = 0; st < d.nstages; st++) v = ApplyStage(d.stages[st], v, ccol[j]);
          }
          Cs[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * C_LD + cl] = v;
        }
      }
    }
    dd::LdsBarrier();
    if (vec_out && d.write_f32) {
#pragma unroll
      for (int q = 0; q < 2048 / NT; q++) {
        const int unit = tid + NT * q, rl = unit >> 6, c4 = (unit & 63) * 4;
        const int row = row0 + sl * 32 + rl, col = n0 + c4;
        if (row < rows && col < d.n)
          *reinterpret_cast<f32x4 *>(d.out + (size_t)(d.row_map ? d.row_map[row] : row) * d.ldo + col) =
              *reinterpret_cast<const f32x4 *>(&Cs[rl * C_LD + c4]);
      }
    } else if (d.write_f32) {
      for (int idx = tid; idx < 32 * BN; idx += NT) {
        const int rl = idx / BN, cl = idx % BN;
        const int row = row0 + sl * 32 + rl, col = n0 + cl;
        if (row < rows && col < d.n) d.out[(size_t)(d.row_map ? d.row_map[row] : row) * d.ldo + col] = Cs[rl * C_LD + cl];
      }
    }
    if (d.out_img.base) {
      // unit = (k-step of 16 columns, k-group of 8, row): 8 values -> 3 x 16 bytes
#pragma unroll
      for (int q = 0; q < 1024 / NT; q++) {
        const int unit = tid + NT * q, rl = unit & 31, kg = (unit >> 5) & 1, ks = unit >> 6;
        const int row = row0 + sl * 32 + rl, col = n0 + ks * 16 + kg * 8;
        if (row < rows && (col >> 4) < d.out_img.nks) {
          f32x4 lo = *reinterpret_cast<const f32x4 *>(&Cs[rl * C_LD + ks * 16 + kg * 8]);
          f32x4 hi = *reinterpret_cast<const f32x4 *>(&Cs[rl * C_LD + ks * 16 + kg * 8 + 4]);
#pragma unroll
          for (int e = 0; e < 4; e++) { if (col + e >= d.n) lo[e] = 0.f; if (col + 4 + e >= d.n) hi[e] = 0.f; }
          bf16x8 p1, p2, p3;
          Split3(lo, hi, &p1, &p2, &p3);
          const int phys = (d.row_map ? d.row_map[row] : row) + d.out_img.guard;
          unsigned char *dst = d.out_img.base + ((size_t)(phys >> 5) * d.out_img.nks + (col >> 4)) * kB3FragBytes + kg * 512 + (phys & 31) * 16;
          *reinterpret_cast<bf16x8 *>(dst) = p1;
          *reinterpret_cast<bf16x8 *>(dst + d.out_img.part_bytes) = p2;
          *reinterpret_cast<bf16x8 *>(dst + 2 * d.out_img.part_bytes) = p3;
        }
      }
    }
    dd::LdsBarrier();
  }
}

}  // namespace b3
}  // namespace rs
