// Acoustic-model kernels for gfx950: the time-dilated affine layers of a nnet3 TDNN as ONE segmented
// FP32-MFMA GEMM per layer, with the ReLU / BatchNorm(test-mode) epilogue fused.
//
// What the reference does per layer (kaldi/src/nnet3): the compiled computation materialises the spliced
// input (kCopyRows/kAddRows for Append/Offset descriptors, nnet-compute.cc:296-371), calls cblas_sgemm
// (AffineComponent::Propagate nnet-simple-component.cc:1242-1251, TdnnComponent::Propagate
// nnet-tdnn-component.cc:181-213), then separate passes for ReLU and BatchNorm
// (nnet-normalize-component.cc:453-463).  Here the splice never exists in memory: the K loop walks
// "segments" (source buffer, time offset, column range) and reads the rows it needs straight from the
// producer's output; halo rows make the row offset a constant (kernels.h).
//
// FP32 in / FP32 accumulate on v_mfma_f32_16x16x4_f32 (exact f32, 157 TF peak).  Layers at least 192 columns wide take
// the split-bf16 kernel instead (nnet_gemm_b3.hip: three bf16 parts per operand, six bf16 MFMAs per product, the same
// accuracy); this file serves the narrow layers (LDA, bottlenecks) and RS_GEMM_B3=0.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include "env.h"

#include "kernels.h"
#include "nnet_common.h"

namespace rs {


// Block tile BM x BN = (16 MT WM) x (64 WN), 4 waves in a WM x WN grid, each wave a (16 MT) x 64 tile of
// v_mfma_f32_16x16x4_f32 accumulators (16-row granularity lets the launcher pick BM so that the number of tiles
// divides evenly over the 256 CUs: a 128-row tile would leave the last round 1/8 full on the hidden layers).
//
// K loop over (segment, 32-wide k-tile) pairs, two-stage LDS ring, ONE barrier per k-tile:
//   iteration t: fragments of tile t: LDS -> registers (16-byte reads); tile t+1: registers -> the other LDS stage;
//   global loads of tile t+2 -> registers; then MT x 4 x 8 MFMAs.  Loads therefore have two full iterations to land.
// LDS rows are 36 floats: 16-byte aligned for b128 accesses and conflict-free for the fragment reads
// (lane (i, q) reads 8 consecutive floats at 36 i + 8 q: the 16 lanes of a phase cover all 64 banks).
// K index mapping inside a k-tile: MFMA step s multiplies A[i][8 q + s] with B[8 q + s][j] (q = lane / 16 is the
// instruction's own k index), so each lane's 8 operands of a row are contiguous in LDS.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- epilogue shared by the GEMM kernels.  C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg.
// Bias and the fused stages are applied in registers (a lane's column is fixed per j, so the per-column parameters are
// loaded once), the tile is transposed through LDS (pitch BN + 4: conflict-free for that layout) and leaves as 16-byte
// row-contiguous stores.  The caller has finished with the k-loop's LDS (barrier) before calling.
template <int MT, int WM, int WN>
__device__ __forceinline__ void GemmEpilogue(const f32x4 (&acc)[MT][4], const GemmDev &d, int rows, int row0, int n0, int epi_mode,
                                             float *Cs) {
  constexpr int BM = 16 * MT * WM, BN = 64 * WN, C_LD = BN + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int cl = wn * 64 + j * 16 + (lane & 15), col = n0 + cl;
    const bool cok = col < d.n;
    const int cc = cok ? col : 0;
    const float bias = (d.bias && cok) ? d.bias[cc] : 0.f;
    float sc = 1.f, of = 0.f;
    if (epi_mode == 2) { sc = d.stages[1].scale[cc]; of = d.stages[1].offset[cc]; }
#pragma unroll
    for (int i = 0; i < MT; i++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float v = __fadd_rn(bias, acc[i][j][r]);
        if (epi_mode == 1) {
          v = v > 0.f ? v : 0.f;
        } else if (epi_mode == 2) {               // ReLU then BatchNorm (test mode): MulColsVec, AddVecToRows
          v = v > 0.f ? v : 0.f;
          v = __fadd_rn(__fmul_rn(v, sc), of);
        } else if (epi_mode == 3) {
          for (int st = 0; st < d.nstages; st++) v = ApplyStage(d.stages[st], v, cc);
        }
        Cs[(wm * 16 * MT + i * 16 + 4 * (lane >> 4) + r) * C_LD + cl] = v;
      }
    }
  }
  __syncthreads();
  const bool vec_out = ((d.ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.out) & 15) == 0) && (((d.n + 3) & ~3) <= d.ldo);
  if (vec_out) {
    for (int idx = tid; idx < BM * (BN / 4); idx += 256) {
      const int rl = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
      const int row = row0 + rl, col = n0 + c4;
      if (row < rows && col < d.n) {
        const int prow = d.row_map ? d.row_map[row] : row;
        f32x4 v = *reinterpret_cast<const f32x4 *>(&Cs[rl * C_LD + c4]);
        if (d.res) {      // a folded residual sum (LayerOp::res_buf): EltwiseKernel's operations
          const f32x4 r = *reinterpret_cast<const f32x4 *>(d.res + (size_t)prow * d.res_ld + col);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = __fadd_rn(d.res_scale != 1.0f ? __fmul_rn(r[e], d.res_scale) : r[e], v[e]);
        }
        *reinterpret_cast<f32x4 *>(d.out + (size_t)prow * d.ldo + col) = v;
      }
    }
  } else {
    for (int idx = tid; idx < BM * BN; idx += 256) {
      const int rl = idx / BN, cl = idx % BN;
      const int row = row0 + rl, col = n0 + cl;
      if (row < rows && col < d.n) {
        const size_t prow = d.row_map ? d.row_map[row] : row;
        float v = Cs[rl * C_LD + cl];
        if (d.res) { const float r = d.res[prow * d.res_ld + col]; v = __fadd_rn(d.res_scale != 1.0f ? __fmul_rn(r, d.res_scale) : r, v); }
        d.out[prow * d.ldo + col] = v;
      }
    }
  }
}

template <int MT, int WM, int WN, bool VEC>
__global__ __launch_bounds__(256, 2) void GemmKernel(GemmDev d, int rows, const int *__restrict__ row_ivec, int epi_mode) {
  constexpr int BM = 16 * MT * WM, BN = 64 * WN, BK = kGemmBK, LDS_LD = 36;
  constexpr int NA = BM / 32, NB = BN / 32;          // 16-byte staging loads per thread and operand
  constexpr int STAGE = (BM + BN) * LDS_LD;          // floats per LDS stage
  static_assert(BM * (BN + 4) <= 2 * STAGE, "the output tile is staged in the k-loop's LDS");
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order: the hardware deals workgroup b to XCD b % 8; all column tiles of one row tile are given
  // to the same XCD back to back, so the row tile's activations are fetched into that XCD's L2 once and the
  // weights (<= 2 MB) stay L2-resident.  Placement only affects speed, never results.
  const int ncol = (d.n + BN - 1) / BN, nrow = (rows + BM - 1) / BM;
  const int bid = blockIdx.x, xcd = bid & 7, local = bid >> 3;
  const int rt = (local / ncol) * 8 + xcd, ct = local % ncol;
  if (rt >= nrow) return;
  const int row0 = rt * BM, n0 = ct * BN;
  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: thread (lr, kq) owns rows lr + 32 h of each tile, floats 4 kq .. 4 kq + 3 of the k-tile
  const int lr = tid >> 3, kq = (tid & 7) * 4;
  int grow[NA];
#pragma unroll
  for (int h = 0; h < NA; h++) {
    grow[h] = row0 + lr + h * 32;
    if (grow[h] >= rows) grow[h] = 0;          // clamped rows are dropped in the epilogue
    if (d.row_map) grow[h] = d.row_map[grow[h]];
  }
  // W's K axis is the concatenation of the (padded) segments, so the weight pointers simply advance by BK per tile
  const float *wptr[NB];
#pragma unroll
  for (int h = 0; h < NB; h++) wptr[h] = d.W + (size_t)(n0 + lr + h * 32) * d.k_pad + kq;
  const float *aptr[NA];       // activation pointers: recomputed when the cursor enters a segment, then += BK

  f32x4 av[NA], bv[NB];
  int staged_lim = 0;
  int seg = 0, k0 = 0, nt = 0;              // (segment, k0) cursor of the next tile to issue
  for (int sgi = 0; sgi < d.nsegs; sgi++) nt += (d.segs[sgi].ncols + BK - 1) / BK;
  auto enter_segment = [&]() __attribute__((always_inline)) {
    const GemmSegDev &sg = d.segs[seg];
#pragma unroll
    for (int h = 0; h < NA; h++) {
      const long arow = sg.per_utt ? (long)row_ivec[grow[h]] : (long)grow[h] + sg.row_off;
      aptr[h] = sg.src + arow * sg.ld + sg.col0 + kq;
    }
  };
  // Branch-free staging loads: every lane always loads (columns past the segment's width are zeroed with selects
  // when the tile is written to LDS).  Reads past a row's end stay inside the buffer (rows are padded / followed by
  // guard rows; W is zero-padded to k_pad).
  auto issue = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < NA; h++) {
      if (VEC) av[h] = *reinterpret_cast<const f32x4 *>(aptr[h]);     // every segment's rows are 16-byte aligned (checked by the launcher)
      else av[h] = f32x4{aptr[h][0], aptr[h][1], aptr[h][2], aptr[h][3]};
      aptr[h] += BK;
    }
#pragma unroll
    for (int h = 0; h < NB; h++) { bv[h] = *reinterpret_cast<const f32x4 *>(wptr[h]); wptr[h] += BK; }
    staged_lim = d.segs[seg].ncols - k0 - kq;      // valid columns of the staged tile, relative to this lane's first
    k0 += BK;
    if (k0 >= d.segs[seg].ncols) { seg++; k0 = 0; if (seg < d.nsegs) enter_segment(); }
  };
  auto stage_store = [&](int stage) __attribute__((always_inline)) {
    float *As = gsm + stage * STAGE, *Bs = As + BM * LDS_LD;
#pragma unroll
    for (int h = 0; h < NA; h++) {
      const f32x4 x = av[h];
      *reinterpret_cast<f32x4 *>(&As[(lr + h * 32) * LDS_LD + kq]) =
          f32x4{staged_lim > 0 ? x[0] : 0.f, staged_lim > 1 ? x[1] : 0.f, staged_lim > 2 ? x[2] : 0.f, staged_lim > 3 ? x[3] : 0.f};
    }
#pragma unroll
    for (int h = 0; h < NB; h++) *reinterpret_cast<f32x4 *>(&Bs[(lr + h * 32) * LDS_LD + kq]) = bv[h];
  };
  if (nt > 0) { enter_segment(); issue(); stage_store(0); }
  if (nt > 1) issue();
  __syncthreads();
  const int fi = lane & 15, fq = (lane >> 4) * 8;
  for (int t = 0; t < nt; t++) {
    const float *As = gsm + (t & 1) * STAGE, *Bs = As + BM * LDS_LD;
    f32x4 af[MT][2], bf[4][2];
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const float *pa = &As[(wm * 16 * MT + i * 16 + fi) * LDS_LD + fq];
      af[i][0] = *reinterpret_cast<const f32x4 *>(pa);
      af[i][1] = *reinterpret_cast<const f32x4 *>(pa + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float *pb = &Bs[(wn * 64 + j * 16 + fi) * LDS_LD + fq];
      bf[j][0] = *reinterpret_cast<const f32x4 *>(pb);
      bf[j][1] = *reinterpret_cast<const f32x4 *>(pb + 4);
    }
    if (t + 1 < nt) {
      stage_store((t + 1) & 1);
      if (t + 2 < nt) issue();
    }
#pragma unroll
    for (int s = 0; s < 8; s++) {
#pragma unroll
      for (int i = 0; i < MT; i++) {
        const float a = s < 4 ? af[i][0][s & 3] : af[i][1][s & 3];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float b = s < 4 ? bf[j][0][s & 3] : bf[j][1][s & 3];
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  GemmEpilogue<MT, WM, WN>(acc, d, rows, row0, n0, epi_mode, gsm);
}

// ---- the same GEMM with direct-to-LDS staging (global_load_lds_dwordx4: a wave instruction moves 64 x 16 bytes from
// per-lane global addresses to 1 KiB of consecutive LDS, no VGPR round trip, no ds_write).  Consecutive LDS means no
// padding, so the bank conflicts of the fragment reads are removed by an XOR swizzle instead: a tile row is 8 units of 16
// bytes (128-byte pitch) and logical unit u of row r is stored at unit u ^ ((r >> 1) & 7) -- the 16 lanes of a fragment
// read phase then touch 16 different (row parity, unit) pairs = all 64 banks.  The lane that fills physical unit p of row
// r simply fetches logical unit p ^ ((r >> 1) & 7) from global memory (the 8 lanes of a row still cover one contiguous
// 128-byte line).  Two LDS stages: the DMA of tile t+1 is in flight during the MFMAs of tile t and is waited for
// (vmcnt) just before the barrier.  A segment's last, partial k-tile cannot be zero-filled by the DMA, so that one
// tile goes through registers with the usual select.  Requires 16-byte aligned segment rows (the launcher checks).
template <int MT, int WM, int WN>
__global__ __launch_bounds__(256, 2) void GemmKernelDma(GemmDev d, int rows, const int *__restrict__ row_ivec, int epi_mode) {
  constexpr int BM = 16 * MT * WM, BN = 64 * WN, BK = kGemmBK;
  constexpr int NA = BM / 32, NB = BN / 32;          // 1 KiB wave transfers per stage and operand
  constexpr int STAGE = (BM + BN) * BK;              // floats per LDS stage
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ncol = (d.n + BN - 1) / BN, nrow = (rows + BM - 1) / BM;
  const int bid = blockIdx.x, xcd = bid & 7, local = bid >> 3;           // XCD-aware tile order, see GemmKernel
  const int rt = (local / ncol) * 8 + xcd, ct = local % ncol;
  if (rt >= nrow) return;
  const int row0 = rt * BM, n0 = ct * BN;
  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: lane (lr, pu) of the workgroup fills physical unit pu of tile rows lr + 32 h with logical unit lu
  const int lr = tid >> 3, pu = tid & 7, lu = pu ^ ((lr >> 1) & 7), kq = lu * 4;
  const int wbase = __builtin_amdgcn_readfirstlane(wave) * 8 * BK;      // this wave's 8 rows inside a 32-row group (floats)
  int grow[NA];
#pragma unroll
  for (int h = 0; h < NA; h++) {
    grow[h] = row0 + lr + h * 32;
    if (grow[h] >= rows) grow[h] = 0;          // clamped rows are dropped in the epilogue
    if (d.row_map) grow[h] = d.row_map[grow[h]];
  }
  const float *wptr[NB];
#pragma unroll
  for (int h = 0; h < NB; h++) wptr[h] = d.W + (size_t)(n0 + lr + h * 32) * d.k_pad + kq;
  const float *aptr[NA];
  int seg = 0, k0 = 0, nt = 0;              // (segment, k0) cursor of the next tile to stage
  for (int sgi = 0; sgi < d.nsegs; sgi++) nt += (d.segs[sgi].ncols + BK - 1) / BK;
  auto enter_segment = [&]() __attribute__((always_inline)) {
    const GemmSegDev &sg = d.segs[seg];
#pragma unroll
    for (int h = 0; h < NA; h++) {
      const long arow = sg.per_utt ? (long)row_ivec[grow[h]] : (long)grow[h] + sg.row_off;
      aptr[h] = sg.src + arow * sg.ld + sg.col0 + kq;
    }
  };
  // stage the tile under the cursor into LDS stage `stage`, advance the cursor
  auto stage_tile = [&](int stage) __attribute__((always_inline)) {
    float *As = gsm + stage * STAGE, *Bs = As + BM * BK;
    const int ncols = d.segs[seg].ncols;
    if (k0 + BK <= ncols) {           // full tile: DMA (wave-uniform branch)
#pragma unroll
      for (int h = 0; h < NA; h++)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)aptr[h],
                                         (void __attribute__((address_space(3))) *)(As + h * 32 * BK + wbase), 16, 0, 0);
    } else {                          // the segment's partial last tile: through registers, zero-filled
      const int lim = ncols - k0 - kq;
#pragma unroll
      for (int h = 0; h < NA; h++) {
        const f32x4 x = *reinterpret_cast<const f32x4 *>(aptr[h]);
        *reinterpret_cast<f32x4 *>(&As[(lr + h * 32) * BK + pu * 4]) =
            f32x4{lim > 0 ? x[0] : 0.f, lim > 1 ? x[1] : 0.f, lim > 2 ? x[2] : 0.f, lim > 3 ? x[3] : 0.f};
      }
    }
#pragma unroll
    for (int h = 0; h < NA; h++) aptr[h] += BK;
#pragma unroll
    for (int h = 0; h < NB; h++) {
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)wptr[h],
                                       (void __attribute__((address_space(3))) *)(Bs + h * 32 * BK + wbase), 16, 0, 0);
      wptr[h] += BK;
    }
    k0 += BK;
    if (k0 >= ncols) { seg++; k0 = 0; if (seg < d.nsegs) enter_segment(); }
  };
  if (nt > 0) { enter_segment(); stage_tile(0); }
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // fragment addressing: lane (fi, q) reads logical units 2q, 2q+1 of its row
  // 16-column tiles of this wave's 64 columns that hold real columns (LDA: 40 columns = 3 of 4): the others' products are not formed
  // (43.4 -> 39.8 us per LDA launch of the headline batch; a ring of three stages with the DMA two k-tiles ahead changed nothing)
  const int nj = __builtin_amdgcn_readfirstlane(min(4, max(0, (d.n - n0 - wn * 64 + 15) / 16)));
  const int fi = lane & 15, fsw = (fi >> 1) & 7;
  const int u0 = (((lane >> 4) * 2) ^ fsw) * 4, u1 = (((lane >> 4) * 2 + 1) ^ fsw) * 4;
  for (int t = 0; t < nt; t++) {
    const float *As = gsm + (t & 1) * STAGE, *Bs = As + BM * BK;
    if (t + 1 < nt) stage_tile((t + 1) & 1);
    f32x4 af[MT][2], bf[4][2];
#pragma unroll
    for (int i = 0; i < MT; i++) {
      const float *pa = &As[(wm * 16 * MT + i * 16 + fi) * BK];
      af[i][0] = *reinterpret_cast<const f32x4 *>(pa + u0);
      af[i][1] = *reinterpret_cast<const f32x4 *>(pa + u1);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float *pb = &Bs[(wn * 64 + (j < nj ? j : 0) * 16 + fi) * BK];
      bf[j][0] = *reinterpret_cast<const f32x4 *>(pb + u0);
      bf[j][1] = *reinterpret_cast<const f32x4 *>(pb + u1);
    }
#pragma unroll
    for (int s = 0; s < 8; s++) {
#pragma unroll
      for (int i = 0; i < MT; i++) {
        const float a = s < 4 ? af[i][0][s & 3] : af[i][1][s & 3];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float b = s < 4 ? bf[j][0][s & 3] : bf[j][1][s & 3];
          if (j < nj) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i][j], 0, 0, 0);
        }
      }
    }
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile has landed
    __syncthreads();
  }
  GemmEpilogue<MT, WM, WN>(acc, d, rows, row0, n0, epi_mode, gsm);
}

template <int MT, int WM, int WN, bool VEC>
static void LaunchGemmV(const GemmDev &d, int rows, const int *row_ivec, hipStream_t s) {
  constexpr int BM = 16 * MT * WM, BN = 64 * WN;
  constexpr size_t smem = 2 * (size_t)(BM + BN) * 36 * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&GemmKernel<MT, WM, WN, VEC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const int nrow = (rows + BM - 1) / BM, ncol = (d.n + BN - 1) / BN;
  const int nrow8 = (nrow + 7) / 8 * 8;      // row tiles are dealt to the 8 XCDs round-robin
  hipLaunchKernelGGL((GemmKernel<MT, WM, WN, VEC>), dim3(nrow8 * ncol), dim3(256), smem, s, d, rows, row_ivec, GemmEpiMode(d, rows));
}
template <int MT, int WM, int WN>
static void LaunchGemmDma(const GemmDev &d, int rows, const int *row_ivec, hipStream_t s) {
  constexpr int BM = 16 * MT * WM, BN = 64 * WN;
  constexpr size_t stages = 2 * (size_t)(BM + BN) * kGemmBK * sizeof(float), ctile = (size_t)BM * (BN + 4) * sizeof(float);
  constexpr size_t smem = stages > ctile ? stages : ctile;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&GemmKernelDma<MT, WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const int nrow = (rows + BM - 1) / BM, ncol = (d.n + BN - 1) / BN;
  const int nrow8 = (nrow + 7) / 8 * 8;
  hipLaunchKernelGGL((GemmKernelDma<MT, WM, WN>), dim3(nrow8 * ncol), dim3(256), smem, s, d, rows, row_ivec, GemmEpiMode(d, rows));
}

template <int MT, int WM, int WN>
static void LaunchGemmT(const GemmDev &d, int rows, const int *row_ivec, hipStream_t s) {
  bool vec = true;
  for (int i = 0; i < d.nsegs; i++)
    vec = vec && (d.segs[i].ld & 3) == 0 && (d.segs[i].col0 & 3) == 0 && (reinterpret_cast<uintptr_t>(d.segs[i].src) & 15) == 0;
  static int use_dma = [] { const char *e = TuneEnv("RS_GEMM_DMA"); return e ? std::atoi(e) : 1; }();
  if (vec && use_dma) LaunchGemmDma<MT, WM, WN>(d, rows, row_ivec, s);
  else if (vec) LaunchGemmV<MT, WM, WN, true>(d, rows, row_ivec, s);
  else LaunchGemmV<MT, WM, WN, false>(d, rows, row_ivec, s);
}

void LaunchGemm(const GemmDev &d, int rows, const int *row_ivec, hipStream_t s) {
  if (rows <= 0) return;
  if (GemmB3IUsable(d)) {
    if (GemmB3JUsable(d, rows)) LaunchGemmB3J(d, rows, s);
    else LaunchGemmB3I(d, rows, s);
    return;
  }
  if (GemmB3Usable(d)) { LaunchGemmB3(d, rows, row_ivec, s); return; }
  static int num_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  // tile height: the one whose busiest CU does the least work; at equal work the shorter tile wins (more workgroups
  // per CU hide the staging latency better: measured 808 vs 848 us on the output layer for 64- vs 128-row tiles)
  auto cost = [&](int bm, int bn) {
    const long tiles = (long)((rows + bm - 1) / bm) * ((d.n + bn - 1) / bn);
    const double pref = bm == 64 ? 0.97 : (bm == 96 ? 0.985 : 1.0);
    return (double)(((tiles + num_cu - 1) / num_cu) * bm * bn) * pref;
  };
  if (d.n <= 64) {
    static int narrow_bm = [] { const char *e = TuneEnv("RS_GEMM_NARROW_BM"); return e ? std::atoi(e) : 0; }();
    if (narrow_bm ? narrow_bm == 64 : cost(64, 64) < cost(128, 64)) LaunchGemmT<1, 4, 1>(d, rows, row_ivec, s);
    else LaunchGemmT<2, 4, 1>(d, rows, row_ivec, s);
    return;
  }
  double c128 = cost(128, 128), c96 = cost(96, 128), c64 = cost(64, 128);
  static int force_bm = [] { const char *e = TuneEnv("RS_GEMM_BM"); return e ? std::atoi(e) : 0; }();
  if (force_bm == 128) c128 = 0; else if (force_bm == 96) c96 = 0; else if (force_bm == 64) c64 = 0;
  if (c128 <= c96 && c128 <= c64) LaunchGemmT<4, 2, 2>(d, rows, row_ivec, s);
  else if (c96 <= c64) LaunchGemmT<3, 2, 2>(d, rows, row_ivec, s);
  else LaunchGemmT<2, 2, 2>(d, rows, row_ivec, s);
}

// true when the kernel LaunchGemm picks for d writes d.out_img itself (the split-bf16 kernels' epilogue does)
bool GemmWritesImage(const GemmDev &d) { return GemmB3IUsable(d) || GemmB3Usable(d); }

// ------------------------------------------------------------------------------------------ elementwise
__global__ void EltwiseKernel(EltwiseDev d, int rows) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)rows * d.dim;
  if (idx >= total) return;
  int row = (int)(idx / d.dim), col = (int)(idx % d.dim);
  if (d.row_map) row = d.row_map[row];
  float v = 0.f;
  for (int t = 0; t < d.nterms; t++) {
    const SumTermDev &tm = d.terms[t];
    float x = tm.src[((long)row + tm.row_off) * tm.ld + tm.col0 + col];
    if (tm.scale != 1.0f) x = __fmul_rn(x, tm.scale);
    v = (t == 0) ? x : __fadd_rn(v, x);
  }
  for (int st = 0; st < d.nstages; st++) v = ApplyStage(d.stages[st], v, col);
  d.out[(size_t)row * d.ldo + col] = v;
}

// wave per row: log-softmax (cu-math / LogSoftMaxPerRow) or NormalizePerRow (cu-math.cc:280-318)
__global__ __launch_bounds__(256) void RowReduceKernel(EltwiseDev d, int rows) {
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (d.row_map) row = d.row_map[row];
  const SumTermDev &tm = d.terms[0];
  const float *src = tm.src + ((long)row + tm.row_off) * tm.ld + tm.col0;
  float *dst = d.out + (size_t)row * d.ldo;
  if (d.row_reduce == 2) {
    float mx = -INFINITY;
    for (int c = lane; c < d.dim; c += 64) mx = fmaxf(mx, src[c]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int c = lane; c < d.dim; c += 64) sum += expf(src[c] - mx);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    float lse = logf(sum);
    for (int c = lane; c < d.dim; c += 64) dst[c] = src[c] - mx - lse;
  } else {
    float ss = 0.f;
    for (int c = lane; c < d.dim; c += 64) ss += src[c] * src[c];
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    float d_scaled = (float)d.dim * d.alpha * d.alpha;
    float nrm = ss * (1.0f / d_scaled);
    nrm = fmaxf(nrm, 1.3552527156068805425e-20f);
    nrm = powf(nrm, -0.5f);
    for (int c = lane; c < d.dim; c += 64) dst[c] = src[c] * nrm;
  }
}

void LaunchEltwise(const EltwiseDev &d, int rows, hipStream_t s) {
  if (rows <= 0) return;
  if (d.row_reduce) {
    hipLaunchKernelGGL(RowReduceKernel, dim3((rows + 3) / 4), dim3(256), 0, s, d, rows);
  } else {
    size_t total = (size_t)rows * d.dim;
    hipLaunchKernelGGL(EltwiseKernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d, rows);
  }
}

__global__ void PriorScaleKernel(float *x, int ld, int rows, int dim, const float *log_priors, float scale) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)rows * dim) return;
  int row = (int)(idx / dim), col = (int)(idx % dim);
  float v = x[(size_t)row * ld + col];
  if (log_priors) v = __fadd_rn(v, -log_priors[col]);
  x[(size_t)row * ld + col] = __fmul_rn(v, scale);
}

void LaunchPriorScale(float *x, int ld, int rows, int dim, const float *log_priors, float scale, hipStream_t s) {
  size_t total = (size_t)rows * dim;
  if (total == 0) return;
  hipLaunchKernelGGL(PriorScaleKernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, ld, rows, dim, log_priors, scale);
}

}  // namespace rs
