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
// FP32 in / FP32 accumulate on v_mfma_f32_32x32x2_f32 (exact f32, 157 TF peak): the 1e-4 log-likelihood
// bound of the north-star rules out bf16/fp8 operands.
#include <hip/hip_runtime.h>
#include <cmath>

#include "kernels.h"

namespace rs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float ApplyStage(const EltStageDev &st, float v, int col) {
  switch (st.kind) {
    case 0: return v > 0.f ? v : 0.f;                                   // ReLU
    case 1: { float t = __fmul_rn(v, st.scale[col]); return __fadd_rn(t, st.offset[col]); }  // MulColsVec then AddVecToRows
    case 4: return __fmul_rn(v, st.alpha);
    default: return v;
  }
}

// 128 x 128 block tile, 4 waves in a 2 x 2 grid, each wave a 64 x 64 tile = 2 x 2 MFMA 32x32 accumulators.
// The K loop runs over (segment, k-tile) pairs; the global loads of tile i+1 are issued into registers before
// the MFMA work of tile i (software pipelining), so HBM/L2 latency hides behind 32 x BK/16 MFMAs per wave.
__global__ __launch_bounds__(256) void GemmKernel(GemmDev d, int rows, const int *__restrict__ row_ivec) {
  constexpr int BM = kGemmBM, BN = kGemmBN, BK = kGemmBK, LDS_LD = BK + 1;
  constexpr int NV = BK / 16;                  // float4 loads per thread per operand half
  __shared__ float As[BM * LDS_LD];
  __shared__ float Bs[BN * LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: the hardware deals workgroup b to XCD b % 8; all column tiles of one row tile are given
  // to the same XCD back to back, so the row tile's activations are fetched into that XCD's L2 once and the
  // weights (<= 2 MB) stay L2-resident.  Placement only affects speed, never results.
  const int ncol = (d.n + BN - 1) / BN, nrow = (rows + BM - 1) / BM;
  const int bid = blockIdx.x, xcd = bid & 7, local = bid >> 3;
  const int rt = (local / ncol) * 8 + xcd, ct = local % ncol;
  if (rt >= nrow) return;
  const int row0 = rt * BM, n0 = ct * BN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // staging: thread (lr0, lk) owns rows lr0 and lr0 + 64, columns lk + 16 * v (v < NV) of each tile
  const int lr0 = tid >> 2, lk = (tid & 3) * 4;
  int grow[2];
  bool avalid[2];
#pragma unroll
  for (int h = 0; h < 2; h++) { grow[h] = row0 + lr0 + h * 64; avalid[h] = grow[h] < rows; if (!avalid[h]) grow[h] = 0; }
  const float *wrow[2];
#pragma unroll
  for (int h = 0; h < 2; h++) wrow[h] = d.W + (size_t)(n0 + lr0 + h * 64) * d.k_pad;

  float4 av[2][NV], bv[2][NV];
  // Branch-free staging loads: every lane always loads (rows past the end were clamped to row 0 and are dropped in
  // the epilogue; columns past the segment's width are zeroed with selects).  Conditional loads make hipcc
  // branch around each load and serialise them behind s_waitcnt.  Reads past a row's end stay inside the buffer
  // (rows are padded / followed by guard rows; W is zero-padded to k_pad).
  auto issue = [&](int seg, int k0) {
    const GemmSegDev &sg = d.segs[seg];
    const bool vec_ok = ((sg.ld & 3) == 0) && ((sg.col0 & 3) == 0);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const long arow = sg.per_utt ? (long)row_ivec[grow[h]] : (long)grow[h] + sg.row_off;
      const float *ap = sg.src + arow * sg.ld + sg.col0;
#pragma unroll
      for (int v = 0; v < NV; v++) {
        const int kk = k0 + lk + 16 * v;
        float4 x;
        if (vec_ok) {
          x = *reinterpret_cast<const float4 *>(ap + kk);
        } else {
          x.x = ap[kk]; x.y = ap[kk + 1]; x.z = ap[kk + 2]; x.w = ap[kk + 3];
        }
        av[h][v] = x;            // masked later, when it is written to LDS (keeps the loads in flight across the MFMAs)
        bv[h][v] = *reinterpret_cast<const float4 *>(wrow[h] + sg.k0 + kk);
      }
    }
  };
  int seg = 0, k0 = 0;
  int kpad = (d.segs[0].ncols + BK - 1) / BK * BK;
  if (d.nsegs > 0) issue(0, 0);
  while (seg < d.nsegs) {
    const int lim = d.segs[seg].ncols - k0 - lk;      // valid columns of the staged tile, relative to this lane's first
    __syncthreads();                     // previous tile's LDS reads are done
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int v = 0; v < NV; v++) {
        float *pa = &As[(lr0 + h * 64) * LDS_LD + lk + 16 * v];
        const int l2 = lim - 16 * v;
        pa[0] = l2 > 0 ? av[h][v].x : 0.f; pa[1] = l2 > 1 ? av[h][v].y : 0.f;
        pa[2] = l2 > 2 ? av[h][v].z : 0.f; pa[3] = l2 > 3 ? av[h][v].w : 0.f;
        float *pb = &Bs[(lr0 + h * 64) * LDS_LD + lk + 16 * v];
        pb[0] = bv[h][v].x; pb[1] = bv[h][v].y; pb[2] = bv[h][v].z; pb[3] = bv[h][v].w;
      }
    // advance to the next tile and start its loads before computing this one
    k0 += BK;
    if (k0 >= kpad) { seg++; k0 = 0; if (seg < d.nsegs) kpad = (d.segs[seg].ncols + BK - 1) / BK * BK; }
    if (seg < d.nsegs) issue(seg, k0);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int kc = kk + (lane >> 5);
      const float a0 = As[(wm * 64 + (lane & 31)) * LDS_LD + kc];
      const float a1 = As[(wm * 64 + 32 + (lane & 31)) * LDS_LD + kc];
      const float b0 = Bs[(wn * 64 + (lane & 31)) * LDS_LD + kc];
      const float b1 = Bs[(wn * 64 + 32 + (lane & 31)) * LDS_LD + kc];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int col = n0 + wn * 64 + j * 32 + (lane & 31);
    if (col >= d.n) continue;
    const float bias = d.bias ? d.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= rows) continue;
        float v = __fadd_rn(bias, acc[i][j][r]);
        for (int st = 0; st < d.nstages; st++) v = ApplyStage(d.stages[st], v, col);
        d.out[(size_t)row * d.ldo + col] = v;
      }
    }
  }
}

void LaunchGemm(const GemmDev &d, int rows, const int *row_ivec, hipStream_t s) {
  if (rows <= 0) return;
  const int nrow = (rows + kGemmBM - 1) / kGemmBM, ncol = (d.n + kGemmBN - 1) / kGemmBN;
  const int nrow8 = (nrow + 7) / 8 * 8;      // row tiles are dealt to the 8 XCDs round-robin
  hipLaunchKernelGGL(GemmKernel, dim3(nrow8 * ncol), dim3(256), 0, s, d, rows, row_ivec);
}

// ------------------------------------------------------------------------------------------ elementwise
__global__ void EltwiseKernel(EltwiseDev d, int rows) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)rows * d.dim;
  if (idx >= total) return;
  int row = (int)(idx / d.dim), col = (int)(idx % d.dim);
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
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
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
