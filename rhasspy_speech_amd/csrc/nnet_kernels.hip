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
__global__ __launch_bounds__(256) void GemmKernel(GemmDev d, int rows, const int *__restrict__ row_utt) {
  constexpr int BM = kGemmBM, BN = kGemmBN, BK = kGemmBK, LDS_LD = BK + 1;
  __shared__ float As[BM * LDS_LD];
  __shared__ float Bs[BN * LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int row0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // each thread stages 2 float4 of A and 2 of B per K step
  const int lr0 = tid >> 2, lk = (tid & 3) * 4;       // rows lr0 and lr0 + 64
  for (int s = 0; s < d.nsegs; s++) {
    const GemmSegDev sg = d.segs[s];
    long arow[2];
    bool avalid[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      int grow = row0 + lr0 + h * 64;
      avalid[h] = grow < rows;
      int gr = avalid[h] ? grow : 0;
      arow[h] = sg.per_utt ? (long)row_utt[gr] : (long)gr + sg.row_off;
    }
    const bool vec_ok = ((sg.ld & 3) == 0) && ((sg.col0 & 3) == 0);
    const int kpad = (sg.ncols + BK - 1) / BK * BK;
    for (int k0 = 0; k0 < kpad; k0 += BK) {
      float4 av[2], bv[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        av[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int kk = k0 + lk;
        if (avalid[h] && kk < sg.ncols) {
          const float *p = sg.src + arow[h] * sg.ld + sg.col0 + kk;
          if (vec_ok && kk + 4 <= sg.ncols) {
            av[h] = *reinterpret_cast<const float4 *>(p);
          } else {
            av[h].x = p[0];
            if (kk + 1 < sg.ncols) av[h].y = p[1];
            if (kk + 2 < sg.ncols) av[h].z = p[2];
            if (kk + 3 < sg.ncols) av[h].w = p[3];
          }
        }
        const int gn = n0 + lr0 + h * 64;
        bv[h] = (gn < d.n_pad) ? *reinterpret_cast<const float4 *>(d.W + (size_t)gn * d.k_pad + sg.k0 + kk)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      __syncthreads();   // previous step's LDS reads are done
#pragma unroll
      for (int h = 0; h < 2; h++) {
        float *pa = &As[(lr0 + h * 64) * LDS_LD + lk];
        pa[0] = av[h].x; pa[1] = av[h].y; pa[2] = av[h].z; pa[3] = av[h].w;
        float *pb = &Bs[(lr0 + h * 64) * LDS_LD + lk];
        pb[0] = bv[h].x; pb[1] = bv[h].y; pb[2] = bv[h].z; pb[3] = bv[h].w;
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        const int kc = kk + (lane >> 5);
        float a0 = As[(wm * 64 + (lane & 31)) * LDS_LD + kc];
        float a1 = As[(wm * 64 + 32 + (lane & 31)) * LDS_LD + kc];
        float b0 = Bs[(wn * 64 + (lane & 31)) * LDS_LD + kc];
        float b1 = Bs[(wn * 64 + 32 + (lane & 31)) * LDS_LD + kc];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
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

void LaunchGemm(const GemmDev &d, int rows, const int *row_utt, hipStream_t s) {
  if (rows <= 0) return;
  dim3 grid((rows + kGemmBM - 1) / kGemmBM, (d.n + kGemmBN - 1) / kGemmBN);
  hipLaunchKernelGGL(GemmKernel, grid, dim3(256), 0, s, d, rows, row_utt);
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
