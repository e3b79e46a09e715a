// Feature front-end kernels for gfx950: batched MFCC and sliding-window CMVN (the iVector estimator lives in
// ivector_kernels.hip).  One 64-lane wavefront per frame for the per-frame kernels (CDNA4 wave64; never 32).
//
// Reference behaviour being reproduced (kaldi/src):
//   feat/feature-window.cc:90-224 (dither, DC removal, pre-emphasis, window, zero padding)
//   matrix/srfft.cc:356-432 + feat/feature-functions.cc:29-51 (real FFT -> 257-bin power spectrum)
//   feat/mel-computations.cc:226-251, feat/feature-mfcc.cc:28-80 (mel, log, DCT, lifter)
//   feat/online-feature.cc:337-452 + transform/cmvn.cc:64-91 (OnlineCmvn)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cfloat>
#include <cmath>

#include "kernels.h"

namespace rs {

#define RS_WAVE 64

__device__ __forceinline__ float WaveSum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, RS_WAVE);
  return v;
}
__device__ __forceinline__ float WaveMax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, RS_WAVE));
  return v;
}

// ------------------------------------------------------------------------------------------ MFCC
// One wave per output row (halo rows recompute their clamped edge frame: <10 % extra work for 3 s
// utterances, no special cases downstream).
//
// The real FFT is the reference's own algorithm (matrix/srfft.cc: in-place single-precision split radix + a
// post-processing pass whose twiddle comes from a float recurrence), restated as levels of independent butterfly
// tasks (srfft_plan.h) executed lane-parallel with the same float operations on the same operands, so the power
// spectrum matches the reference to the bit instead of to its own ~1e-3 rounding noise.  This file is compiled with
// -ffp-contract=off for that reason.
__device__ __forceinline__ void SrfftRunTask(const int4 tk, const float *__restrict__ tw, float *xr, float *xi) {
  const int kind = tk.x & 0xff, lg = tk.x >> 8, off = tk.y;
  if (kind == 0) {
    // srfft.cc:278-333 for one n: the four points it touches are private to this task
    const int m = 1 << lg, m2 = m >> 1, m4 = m >> 2, n = tk.z;
    const int e0 = off + n, e1 = e0 + m4, e2 = e0 + m2, e3 = e2 + m4;
    const float ar = xr[e0], ai = xi[e0], br = xr[e1], bi = xi[e1], cr = xr[e2], ci = xi[e2], dr = xr[e3], di = xi[e3];
    // step 1 on (e0, e2) and (e1, e3)
    xr[e0] = ar + cr; xi[e0] = ai + ci;
    xr[e1] = br + dr; xi[e1] = bi + di;
    const float p_r = ar - cr, p_i = ai - ci, q_r = br - dr, q_i = bi - di;
    // step 2 on the pair (e2, e3)
    float r1 = p_r + q_i, i2 = p_i + q_r, i1 = p_i - q_r, r2 = p_r - q_i;
    // steps 3 & 4
    if (tk.w == -2) {
      const float sqhalf = 0.70710678118654752440f;
      const float t1 = sqhalf * (r1 + i1);
      i1 = sqhalf * (i1 - r1);
      r1 = t1;
      const float t2 = sqhalf * (i2 - r2);
      i2 = -sqhalf * (r2 + i2);
      r2 = t2;
    } else if (tk.w >= 0) {
      const float *w = tw + (size_t)tk.w * 6;
      const float cn = w[0], spcn = w[1], smcn = w[2], c3n = w[3], spc3n = w[4], smc3n = w[5];
      float t2 = cn * (r1 + i1);
      float t1 = spcn * r1 + t2;
      r1 = smcn * i1 + t2;
      i1 = t1;
      t2 = c3n * (r2 + i2);
      t1 = spc3n * r2 + t2;
      r2 = smc3n * i2 + t2;
      i2 = t1;
    }
    xr[e2] = r1; xi[e2] = i1;
    xr[e3] = r2; xi[e3] = i2;
  } else if (kind == 1) {
    // srfft.cc:227-264: the whole length-4 transform
    float r0 = xr[off], r1 = xr[off + 1], r2 = xr[off + 2], r3 = xr[off + 3];
    float i0 = xi[off], i1 = xi[off + 1], i2 = xi[off + 2], i3 = xi[off + 3];
    float t;
    t = r0 + r2; r2 = r0 - r2; r0 = t;
    t = i0 + i2; i2 = i0 - i2; i0 = t;
    t = r1 + r3; r3 = r1 - r3; r1 = t;
    t = i1 + i3; i3 = i1 - i3; i1 = t;
    t = r0 + r1; r1 = r0 - r1; r0 = t;
    t = i0 + i1; i1 = i0 - i1; i0 = t;
    const float t1 = r2 + i3, t2 = i2 + r3;
    i2 = i2 - r3;
    r3 = r2 - i3;
    r2 = t1;
    i3 = t2;
    xr[off] = r0; xr[off + 1] = r1; xr[off + 2] = r2; xr[off + 3] = r3;
    xi[off] = i0; xi[off + 1] = i1; xi[off + 2] = i2; xi[off + 3] = i3;
  } else {
    // srfft.cc:265-274: length 2
    const float r0 = xr[off], r1 = xr[off + 1], i0 = xi[off], i1 = xi[off + 1];
    xr[off] = r0 + r1; xr[off + 1] = r0 - r1;
    xi[off] = i0 + i1; xi[off + 1] = i0 - i1;
  }
}

// The waves of a workgroup work on different frames, each in its own slices of the LDS arrays: a wave only has to order
// its own LDS traffic (its earlier writes land before its later reads), no wave ever waits for another one.
__device__ __forceinline__ void WaveLdsSync() {
  __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// WPB = waves (frames) per workgroup.  4 by default; 16 with all the CU's LDS requested when several decode pipelines are in
// flight, so that no GemmKernelB3 workgroup of another pipeline can share the CU (DESIGN.md section 5: that kernel
// perturbs this one's LDS-staged arithmetic when they share a CU).
template <int NFFT, int WPB>   // padded window (real points)
__global__ __launch_bounds__(64 * WPB) void MfccKernel(MfccDev m, BatchGeom g, const int16_t *__restrict__ pcm,
                                                       float *__restrict__ feats, int ld, const int *__restrict__ out_rows) {
  constexpr int NC = NFFT / 2;        // complex points
  const int4 *tasks = reinterpret_cast<const int4 *>(m.fft_tasks);
  const float *fft_tw = m.fft_tw;
  extern __shared__ __attribute__((aligned(16))) float mfcc_lds[];
  float (*xbuf)[NFFT] = reinterpret_cast<float (*)[NFFT]>(mfcc_lds);
  float (*xrb)[NC] = reinterpret_cast<float (*)[NC]>(mfcc_lds + WPB * NFFT);
  float (*xib)[NC] = reinterpret_cast<float (*)[NC]>(mfcc_lds + WPB * (NFFT + NC));
  float (*pw)[NC + 1] = reinterpret_cast<float (*)[NC + 1]>(mfcc_lds + WPB * (NFFT + 2 * NC));
  float (*lm)[64] = reinterpret_cast<float (*)[64]>(mfcc_lds + WPB * (NFFT + 3 * NC + 1));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * WPB + wave;
  const bool active = row < g.total_rows;
  int u = 0, t = 0;
  if (active) {
    u = g.d_row_utt[row];
    t = g.d_row_t[row];
    int T = g.d_num_frames[u];
    t = t >= T ? T - 1 : t;
    t = t < 0 ? 0 : t;          // (also the halo rows of an utterance too short for one frame: frame 0 of whatever follows it, never read back)
  }
  float *x = xbuf[wave];
  float raw_energy = 0.f, dc = 0.f;
  if (active) {
    const int16_t *src = pcm + g.d_sample_off[u] + (int64_t)t * m.shift;
    // 1. load (int16 -> float, unscaled), dither, DC removal.  The frame sum follows the reference's BLAS call
    // (VectorBase::Sum() = cblas_sdot(n, x, 1, &one, 0), OpenBLAS kernel/x86_64/sdot.c strided loop): adjacent pairs are added
    // in float and the pair sums accumulated in a double.  With integer samples every order gives the same sum; with the
    // dither noise in, a differently rounded mean re-rounds every sample of a loud frame and moves the cepstra by 1e-3.  The
    // double accumulation is a wave reduction here: the pair sums of a frame span far fewer than 53 bits, so it is exact in
    // any order (it would take a pair cancelling to below 2^-33 of the frame's peak to make the order matter).
    double dsum = 0.0;
    // (dither off: finite stand-ins multiplied by 0 -- v + 0 is v -- so that the loads below carry no condition: written as
    // `if (noise) v += ...` the compiler put every load under a branch with a full wait behind it, 16 dependent round trips per frame)
    const float *noise = m.dither ? m.dither + (size_t)(t + (g.d_frame0 ? g.d_frame0[u] : 0)) * m.win : m.window;
    const float dv = m.dither ? m.dither_value : 0.f;
    constexpr int JP = NFFT / 128;                      // sample pairs per lane
    float s0[JP], s1[JP], n0[JP], n1[JP];
#pragma unroll
    for (int j = 0; j < JP; j++) {                      // all requests first ...
      const int i0 = 2 * (lane + RS_WAVE * j), c0 = i0 < m.win ? i0 : m.win - 1, c1 = i0 + 1 < m.win ? i0 + 1 : m.win - 1;
      s0[j] = (float)src[c0]; s1[j] = (float)src[c1];
      n0[j] = noise[c0]; n1[j] = noise[c1];
    }
#pragma unroll
    for (int j = 0; j < JP; j++) {                      // ... then Dither(): data[i] += RandGauss(&rstate) * dither_value (no FMA: -ffp-contract=off)
      const int i0 = 2 * (lane + RS_WAVE * j), i1 = i0 + 1;
      const float v0 = s0[j] + n0[j] * dv, v1 = s1[j] + n1[j] * dv;
      if (i0 < m.win) x[i0] = v0;
      if (i1 < m.win) { x[i1] = v1; dsum += (double)(v0 + v1); }
      else if (i0 < m.win) dsum += (double)v0;
    }
    for (int i = m.win + lane; i < NFFT; i += RS_WAVE) x[i] = 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dsum += __shfl_xor(dsum, o, RS_WAVE);
    // DC removal (x[i] += -mean) is applied where the samples are read below -- the same subtraction on the same operands, once
    // for a sample itself and once as its right neighbour's predecessor -- instead of in a pass of its own over the LDS copy
    dc = m.remove_dc ? (float)dsum / (float)m.win : 0.f;
    if (m.use_energy && m.raw_energy) {
      WaveLdsSync();      // the sample pairs were written by other lanes than the ones that read them here
      float e = 0.f;
      for (int i = lane; i < m.win; i += RS_WAVE) { const float v = m.remove_dc ? x[i] - dc : x[i]; e += v * v; }
      raw_energy = logf(fmaxf(WaveSum(e), FLT_EPSILON));
    }
  }
  WaveLdsSync();
  float *xr = xrb[wave], *xi = xib[wave];
  if (active) {
    // 2. pre-emphasis (uses the *un-emphasised* left neighbour, as the backwards loop of the reference does) and
    // window; the even / odd samples are the real / imaginary parts of the half-length complex transform
    float e = 0.f;
    for (int i = lane; i < NFFT; i += RS_WAVE) {
      float y = 0.f;
      if (i < m.win) {
        float prev = x[i > 0 ? i - 1 : 0], cur = x[i];
        if (m.remove_dc) { prev -= dc; cur -= dc; }
        const float v = cur - m.preemph * prev;
        y = v * m.window[i];
        e += y * y;
      }
      if (i & 1) xi[i >> 1] = y; else xr[i >> 1] = y;
    }
    if (m.use_energy && !m.raw_energy) raw_energy = logf(fmaxf(WaveSum(e), FLT_EPSILON));
  }
  WaveLdsSync();
  // 3. split-radix complex FFT, level by level (tasks of one level touch disjoint points)
  for (int L = 0; L < m.fft_num_levels; L++) {
    if (active)
      for (int ti = m.fft_level_begin[L] + lane; ti < m.fft_level_begin[L + 1]; ti += RS_WAVE) SrfftRunTask(tasks[ti], fft_tw, xr, xi);
    WaveLdsSync();
  }
  // 4. real-FFT post-processing (srfft.cc:379-417) fused with the power spectrum (feature-functions.cc:41-49);
  // spectrum element k of the bit-reversal pass is element perm[k] of the in-place result
  if (active) {
    for (int k = lane + 1; 2 * k <= NC; k += RS_WAVE) {
      const int kd = NC - k;
      const int pk = m.fft_perm[k], pd = m.fft_perm[kd];
      const float bk_re = xr[pk], bk_im = xi[pk], bd_re = xr[pd], bd_im = xi[pd];
      const float kn_re = m.fft_kn[2 * k], kn_im = m.fft_kn[2 * k + 1];
      const float ck_re = 0.5f * (bk_re + bd_re), ck_im = 0.5f * (bk_im - bd_im);
      const float dk_re = 0.5f * (bk_im + bd_im), dk_im = -0.5f * (bk_re - bd_re);
      // A_k = C_k + kN D_k
      const float a_re = ck_re + (kn_re * dk_re - kn_im * dk_im);
      const float a_im = ck_im + (kn_re * dk_im + kn_im * dk_re);
      pw[wave][k] = a_re * a_re + a_im * a_im;
      if (kd != k) {
        // A_k' = conj(C_k) + (-conj(kN)) conj(D_k)
        const float nd_im = -dk_im, nk_re = -kn_re;
        const float b_re = ck_re + (nk_re * dk_re - kn_im * nd_im);
        const float b_im = -ck_im + (nk_re * nd_im + kn_im * dk_re);
        pw[wave][kd] = b_re * b_re + b_im * b_im;
      }
    }
    if (lane == 0) {
      const float d0 = xr[m.fft_perm[0]], d1 = xi[m.fft_perm[0]];
      const float zeroth = d0 + d1, n2th = d0 - d1;
      pw[wave][0] = zeroth * zeroth;
      pw[wave][NC] = n2th * n2th;
    }
  }
  WaveLdsSync();
  // 5. mel filterbank + log
  if (active && lane < m.nbins) {
    int off = m.mel_offset[lane], len = m.mel_len[lane];
    const float *w = m.mel_weights + m.mel_start[lane];
    float e = 0.f;
    for (int i = 0; i < len; i++) e += w[i] * pw[wave][off + i];
    lm[wave][lane] = logf(fmaxf(e, FLT_EPSILON));
  }
  WaveLdsSync();
  // 6. DCT + lifter, write the row
  if (active && lane < m.nceps) {
    const float *d = m.dct + lane * m.nbins;
    float c = 0.f;
    for (int b = 0; b < m.nbins; b++) c += d[b] * lm[wave][b];
    c *= m.lifter[lane];
    if (m.use_energy && lane == 0) c = fmaxf(raw_energy, m.log_energy_floor);
    feats[(size_t)(out_rows ? out_rows[row] : row) * ld + lane] = c;      // out_rows: streams write into their pool rows
  }
}

template <int NFFT, int WPB>
static void LaunchMfccT(const MfccDev &m, const BatchGeom &g, const int16_t *pcm, float *feats, int ld, bool exclusive, const int *out_rows,
                        hipStream_t s) {
  const int blocks = (g.total_rows + WPB - 1) / WPB;
  if (!blocks) return;
  constexpr size_t need = sizeof(float) * WPB * (NFFT + 3 * (NFFT / 2) + 1 + 64);
  const size_t smem = exclusive ? std::max<size_t>(need, 159 * 1024) : need;
  static size_t attr = 0;
  if (smem > attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&MfccKernel<NFFT, WPB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = smem;
  }
  hipLaunchKernelGGL((MfccKernel<NFFT, WPB>), dim3(blocks), dim3(64 * WPB), smem, s, m, g, pcm, feats, ld, out_rows);
}

void LaunchMfcc(const MfccDev &m, const BatchGeom &g, const int16_t *pcm, float *feats, int ld, hipStream_t s, bool exclusive,
                const int *out_rows) {
  if (m.padded == 512) {
    if (exclusive) LaunchMfccT<512, 16>(m, g, pcm, feats, ld, true, out_rows, s);
    else LaunchMfccT<512, 4>(m, g, pcm, feats, ld, false, out_rows, s);
  } else {
    LaunchMfccT<2048, 4>(m, g, pcm, feats, ld, exclusive, out_rows, s);
  }
}

// ------------------------------------------------------------------------------------------ online CMVN
// Workgroup per utterance, frames in chunks staged through LDS.  Per chunk: (1) thread d walks the frames with the same
// add-new / subtract-old double-precision update ComputeStatsForFrame performs (the only sequential part: two adds
// per frame), (2) one thread per frame does SmoothOnlineCmvnStats' per-frame scalars (the two fp64 divisions), (3) all
// threads apply the mean-only ApplyCmvn to the chunk in parallel with coalesced stores.  No speaker stats: the
// reference starts every utterance from a fresh process.
constexpr int kCmvnTC = 32, kCmvnMaxDim = 128;
//
// Streams (t_begin != null): utterance u resumes at frame t_begin[u] with the running sums and the window count parked in
// state[(D + 1) * state_slot[u]] by the launch that produced frame t_begin[u] - 1, and parks them again at frame T; the frames
// that leave the window are re-read from `in`, which holds the whole stream.
__global__ __launch_bounds__(256) void OnlineCmvnKernel(CmvnDev c, BatchGeom g, const float *__restrict__ in, float *__restrict__ out, int ld,
                                                        const int *__restrict__ t_begin, double *__restrict__ state, const int *__restrict__ state_slot) {
  __shared__ float xs[kCmvnTC][kCmvnMaxDim];      // the chunk
  __shared__ float xp[kCmvnTC][kCmvnMaxDim];      // the frames leaving the window while the chunk enters
  __shared__ double ss[kCmvnTC][kCmvnMaxDim];     // running sums after each frame
  __shared__ double nn[kCmvnTC], aa[kCmvnTC];     // frame count in the window; weight of the global stats
  __shared__ float al[kCmvnTC];                   // -1 / (smoothed count)
  __shared__ float edge[2][kCmvnMaxDim];          // normalised first / last frame (halo rows replicate them)
  __shared__ double gs[kCmvnMaxDim];              // global stats (read per element in step 3 while the window is not full)
  const int u = blockIdx.x, tid = threadIdx.x, D = c.dim, W = c.cmn_window;
  if (tid < D) gs[tid] = c.global_stats[tid];
  const int T = g.d_num_frames[u];
  const size_t base = (size_t)g.d_row_base[u] + g.L;
  double sum = 0.0, count = 0.0;                  // threads < D
  const double gcount = c.global_stats[D];
  const int t_first = t_begin ? t_begin[u] : 0;
  double *park = t_begin ? state + (size_t)(D + 1) * state_slot[u] : nullptr;
  if (park && t_first > 0 && tid < D) { sum = park[tid]; count = park[D]; }
  for (int t0 = t_first; t0 < T; t0 += kCmvnTC) {
    const int n = T - t0 < kCmvnTC ? T - t0 : kCmvnTC;
    for (int idx = tid; idx < n * D; idx += 256) {
      const int i = idx / D, d = idx % D, tp = t0 + i - W;
      xs[i][d] = in[(base + t0 + i) * ld + d];
      xp[i][d] = tp >= 0 ? in[(base + tp) * ld + d] : 0.f;
    }
    __syncthreads();
    if (tid < D) {
#pragma unroll 8
      for (int i = 0; i < n; i++) {
        sum += (double)xs[i][tid];
        count += 1.0;
        if (t0 + i - W >= 0) { sum -= (double)xp[i][tid]; count -= 1.0; }
        ss[i][tid] = sum;
        if (tid == 0) nn[i] = count;
      }
    }
    __syncthreads();
    if (tid < n) {
      double nf = nn[tid], a = 0.0;
      if (nf < (double)W) {
        double from_global = (double)W - nf;
        if (from_global > (double)c.global_frames) from_global = (double)c.global_frames;
        if (from_global > 0.0) { a = from_global / gcount; nf += a * gcount; }
      }
      aa[tid] = a;
      al[tid] = (float)(-1.0 / nf);
    }
    __syncthreads();
    for (int idx = tid; idx < n * D; idx += 256) {
      const int i = idx / D, d = idx % D;
      double sv = ss[i][d];
      const double a = aa[i];
      if (a > 0.0) sv += a * gs[d];
      const float offset = (float)((double)al[i] * sv);
      const float yv = xs[i][d] + offset;
      out[(base + t0 + i) * ld + d] = yv;
      if (t0 + i == 0) edge[0][d] = yv;
      if (t0 + i == T - 1) edge[1][d] = yv;
    }
    __syncthreads();
  }
  if (park && tid < D) { park[tid] = sum; if (tid == 0) park[D] = count; }
  if (T > 0 && !t_begin) {
    for (int idx = tid; idx < g.L * D; idx += 256) out[(base - g.L + idx / D) * ld + idx % D] = edge[0][idx % D];
    for (int idx = tid; idx < g.R * D; idx += 256) out[(base + T + idx / D) * ld + idx % D] = edge[1][idx % D];
  }
}

void LaunchOnlineCmvn(const CmvnDev &c, const BatchGeom &g, const float *in, float *out, int ld, hipStream_t s, const int *t_begin,
                      double *state, const int *state_slot) {
  if (g.n_utts == 0) return;
  hipLaunchKernelGGL(OnlineCmvnKernel, dim3(g.n_utts), dim3(256), 0, s, c, g, in, out, ld, t_begin, state, state_slot);
}

// ------------------------------------------------------------------------------------------ row copies
// dst row dst_row[i] <- src row src_row[i], `width` 4-byte words each (one workgroup per row).  The streaming engine's
// glue: pool rows <-> the dense transient buffers of one advance (clamped context gathers, state parking).
__global__ __launch_bounds__(256) void CopyRowsKernel(const unsigned *__restrict__ src, long src_ld, const int *__restrict__ src_row,
                                                      unsigned *__restrict__ dst, long dst_ld, const int *__restrict__ dst_row, int width) {
  const int i = blockIdx.x;
  const unsigned *sp = src + (size_t)(src_row ? src_row[i] : i) * src_ld;
  unsigned *dp = dst + (size_t)(dst_row ? dst_row[i] : i) * dst_ld;
  if (((reinterpret_cast<uintptr_t>(sp) | reinterpret_cast<uintptr_t>(dp)) & 15) == 0) {
    const int w4 = width >> 2;
    for (int k = threadIdx.x; k < w4; k += 256) reinterpret_cast<uint4 *>(dp)[k] = reinterpret_cast<const uint4 *>(sp)[k];
    for (int k = (w4 << 2) + threadIdx.x; k < width; k += 256) dp[k] = sp[k];
  } else {
    for (int k = threadIdx.x; k < width; k += 256) dp[k] = sp[k];
  }
}
// up to four such copies with the same row lists in one launch (blockIdx.y = which)
__global__ __launch_bounds__(256) void CopyRowsMultiKernel(CopyRowsSet set, const int *__restrict__ src_row, const int *__restrict__ dst_row) {
  const int i = blockIdx.x;
  const CopyRowsSet::One c = set.a[blockIdx.y];
  const unsigned *sp = static_cast<const unsigned *>(c.src) + (size_t)(src_row ? src_row[i] : i) * c.ld;
  unsigned *dp = static_cast<unsigned *>(c.dst) + (size_t)(dst_row ? dst_row[i] : i) * c.ld;
  if (((reinterpret_cast<uintptr_t>(sp) | reinterpret_cast<uintptr_t>(dp)) & 15) == 0) {
    const int w4 = c.width >> 2;
    for (int k = threadIdx.x; k < w4; k += 256) reinterpret_cast<uint4 *>(dp)[k] = reinterpret_cast<const uint4 *>(sp)[k];
    for (int k = (w4 << 2) + threadIdx.x; k < c.width; k += 256) dp[k] = sp[k];
  } else {
    for (int k = threadIdx.x; k < c.width; k += 256) dp[k] = sp[k];
  }
}
void LaunchCopyRowsMulti(const CopyRowsSet &set, const int *src_row, const int *dst_row, int n, hipStream_t s) {
  if (n <= 0 || set.count <= 0) return;
  hipLaunchKernelGGL(CopyRowsMultiKernel, dim3(n, set.count), dim3(256), 0, s, set, src_row, dst_row);
}

// zero fill of a list of regions (blockIdx.y = region, 16 workgroups per region)
__global__ __launch_bounds__(256) void ZeroRegionsKernel(ZeroRegions z) {
  const ZeroRegions::One r = z.r[blockIdx.y];
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
  if (((reinterpret_cast<uintptr_t>(r.p) | r.bytes) & 15) == 0) {
    uint4 *p = static_cast<uint4 *>(r.p);
    for (size_t k = t; k < r.bytes / 16; k += nthreads) p[k] = make_uint4(0u, 0u, 0u, 0u);
  } else {
    unsigned *p = static_cast<unsigned *>(r.p);
    for (size_t k = t; k < r.bytes / 4; k += nthreads) p[k] = 0u;
  }
}
void LaunchZeroRegions(const ZeroRegions &z, hipStream_t s) {
  if (z.count <= 0) return;
  hipLaunchKernelGGL(ZeroRegionsKernel, dim3(16, z.count), dim3(256), 0, s, z);
}

void LaunchCopyRows(const void *src, long src_ld_words, const int *src_row, void *dst, long dst_ld_words, const int *dst_row, int n, int width_words,
                    hipStream_t s) {
  if (n <= 0 || width_words <= 0) return;
  hipLaunchKernelGGL(CopyRowsKernel, dim3(n), dim3(256), 0, s, static_cast<const unsigned *>(src), src_ld_words, src_row,
                     static_cast<unsigned *>(dst), dst_ld_words, dst_row, width_words);
}

// ------------------------------------------------------------------------------------------ row geometry
// Per-row lookup tables of the ragged time-major layout (kernels.h), derived on the device from the per-utterance row
// bases: utterance of a row, its frame index relative to the utterance (negative / >= T in the halo), and -- offline --
// the iVector row it reads.
__global__ void RowGeometryKernel(int n_utts, int rows, int L, const int *__restrict__ row_base, const int *__restrict__ ivrow_base,
                                  int *__restrict__ row_utt, int *__restrict__ row_t, int *__restrict__ row_ivec) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  int lo = 0, hi = n_utts;          // largest u with row_base[u] <= r
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (row_base[mid] <= r) lo = mid; else hi = mid; }
  row_utt[r] = lo;
  row_t[r] = r - row_base[lo] - L;
  if (row_ivec) row_ivec[r] = ivrow_base[lo];
}
__global__ void FrameRowsKernel(int n_utts, int n_segs, int total, int L, int slab_len, const int *__restrict__ seg_off,
                                const int *__restrict__ row_base, int *__restrict__ frame_rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int lo = 0, hi = n_segs;          // largest segment with seg_off[seg] <= i
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_off[mid] <= i) lo = mid; else hi = mid; }
  const int k = lo / n_utts, u = lo % n_utts;
  frame_rows[i] = row_base[u] + L + k * slab_len + (i - seg_off[lo]);
}
void LaunchFrameRows(int n_utts, int n_segs, int total, int L, int slab_len, const int *seg_off, const int *row_base, int *frame_rows,
                     hipStream_t s) {
  if (total <= 0) return;
  hipLaunchKernelGGL(FrameRowsKernel, dim3((total + 255) / 256), dim3(256), 0, s, n_utts, n_segs, total, L, slab_len, seg_off, row_base, frame_rows);
}
void LaunchRowGeometry(int n_utts, int rows, int L, const int *row_base, const int *ivrow_base, int *row_utt, int *row_t, int *row_ivec,
                       hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(RowGeometryKernel, dim3((rows + 255) / 256), dim3(256), 0, s, n_utts, rows, L, row_base, ivrow_base, row_utt, row_t, row_ivec);
}

// Debug aid (RS_LDS_POISON=1): fills the LDS of every CU with a NaN pattern.  LDS is not cleared between workgroups, so a
// kernel that reads LDS it never wrote shows up as a changed result when this runs in front of it.
__global__ __launch_bounds__(1024) void LdsPoisonKernel(unsigned pattern, unsigned *sink) {
  extern __shared__ unsigned poison[];
  const int n = 160 * 1024 / 4;
  for (int i = threadIdx.x; i < n; i += 1024) poison[i] = pattern + i;
  __syncthreads();
  if (poison[(threadIdx.x * 37) % n] == 1u) sink[0] = 1;      // keeps the stores alive
}
// Same idea for the register files (not cleared between waves either): a kernel that reads a register it never wrote
// sees whatever the previous wave left there; this leaves NaN patterns in v4..v255 and s16..s99 of every SIMD.
__global__ __launch_bounds__(64) void RegPoisonKernel(unsigned pattern, unsigned *sink) {
  const unsigned p = pattern + threadIdx.x;
  const unsigned q = __builtin_amdgcn_readfirstlane(pattern);
  __asm__ volatile("v_mov_b32 v4, %0\n\tv_mov_b32 v5, %0\n\tv_mov_b32 v6, %0\n\tv_mov_b32 v7, %0\n\tv_mov_b32 v8, %0\n\tv_mov_b32 v9, %0\n\tv_mov_b32 v10, %0\n\tv_mov_b32 v11, %0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, %0\n\tv_mov_b32 v14, %0\n\tv_mov_b32 v15, %0\n\tv_mov_b32 v16, %0\n\tv_mov_b32 v17, %0\n\tv_mov_b32 v18, %0\n\tv_mov_b32 v19, %0\n\tv_mov_b32 v20, %0\n\tv_mov_b32 v21, %0\n\tv_mov_b32 v22, %0\n\tv_mov_b32 v23, %0\n\tv_mov_b32 v24, %0\n\tv_mov_b32 v25, %0\n\tv_mov_b32 v26, %0\n\tv_mov_b32 v27, %0\n\tv_mov_b32 v28, %0\n\tv_mov_b32 v29, %0\n\tv_mov_b32 v30, %0\n\tv_mov_b32 v31, %0\n\tv_mov_b32 v32, %0\n\tv_mov_b32 v33, %0\n\tv_mov_b32 v34, %0\n\tv_mov_b32 v35, %0\n\tv_mov_b32 v36, %0\n\tv_mov_b32 v37, %0\n\tv_mov_b32 v38, %0\n\tv_mov_b32 v39, %0\n\tv_mov_b32 v40, %0\n\tv_mov_b32 v41, %0\n\tv_mov_b32 v42, %0\n\tv_mov_b32 v43, %0\n\tv_mov_b32 v44, %0\n\tv_mov_b32 v45, %0\n\tv_mov_b32 v46, %0\n\tv_mov_b32 v47, %0\n\tv_mov_b32 v48, %0\n\tv_mov_b32 v49, %0\n\tv_mov_b32 v50, %0\n\tv_mov_b32 v51, %0\n\tv_mov_b32 v52, %0\n\tv_mov_b32 v53, %0\n\tv_mov_b32 v54, %0\n\tv_mov_b32 v55, %0\n\tv_mov_b32 v56, %0\n\tv_mov_b32 v57, %0\n\tv_mov_b32 v58, %0\n\tv_mov_b32 v59, %0\n\tv_mov_b32 v60, %0\n\tv_mov_b32 v61, %0\n\tv_mov_b32 v62, %0\n\tv_mov_b32 v63, %0\n\tv_mov_b32 v64, %0\n\tv_mov_b32 v65, %0\n\tv_mov_b32 v66, %0\n\tv_mov_b32 v67, %0\n\tv_mov_b32 v68, %0\n\tv_mov_b32 v69, %0\n\tv_mov_b32 v70, %0\n\tv_mov_b32 v71, %0\n\tv_mov_b32 v72, %0\n\tv_mov_b32 v73, %0\n\tv_mov_b32 v74, %0\n\tv_mov_b32 v75, %0\n\tv_mov_b32 v76, %0\n\tv_mov_b32 v77, %0\n\tv_mov_b32 v78, %0\n\tv_mov_b32 v79, %0\n\tv_mov_b32 v80, %0\n\tv_mov_b32 v81, %0\n\tv_mov_b32 v82, %0\n\tv_mov_b32 v83, %0\n\tv_mov_b32 v84, %0\n\tv_mov_b32 v85, %0\n\tv_mov_b32 v86, %0\n\tv_mov_b32 v87, %0\n\tv_mov_b32 v88, %0\n\tv_mov_b32 v89, %0\n\tv_mov_b32 v90, %0\n\tv_mov_b32 v91, %0\n\tv_mov_b32 v92, %0\n\tv_mov_b32 v93, %0\n\tv_mov_b32 v94, %0\n\tv_mov_b32 v95, %0\n\tv_mov_b32 v96, %0\n\tv_mov_b32 v97, %0\n\tv_mov_b32 v98, %0\n\tv_mov_b32 v99, %0\n\tv_mov_b32 v100, %0\n\tv_mov_b32 v101, %0\n\tv_mov_b32 v102, %0\n\tv_mov_b32 v103, %0\n\tv_mov_b32 v104, %0\n\tv_mov_b32 v105, %0\n\tv_mov_b32 v106, %0\n\tv_mov_b32 v107, %0\n\tv_mov_b32 v108, %0\n\tv_mov_b32 v109, %0\n\tv_mov_b32 v110, %0\n\tv_mov_b32 v111, %0\n\tv_mov_b32 v112, %0\n\tv_mov_b32 v113, %0\n\tv_mov_b32 v114, %0\n\tv_mov_b32 v115, %0\n\tv_mov_b32 v116, %0\n\tv_mov_b32 v117, %0\n\tv_mov_b32 v118, %0\n\tv_mov_b32 v119, %0\n\tv_mov_b32 v120, %0\n\tv_mov_b32 v121, %0\n\tv_mov_b32 v122, %0\n\tv_mov_b32 v123, %0\n\tv_mov_b32 v124, %0\n\tv_mov_b32 v125, %0\n\tv_mov_b32 v126, %0\n\tv_mov_b32 v127, %0\n\tv_mov_b32 v128, %0\n\tv_mov_b32 v129, %0\n\tv_mov_b32 v130, %0\n\tv_mov_b32 v131, %0\n\tv_mov_b32 v132, %0\n\tv_mov_b32 v133, %0\n\tv_mov_b32 v134, %0\n\tv_mov_b32 v135, %0\n\tv_mov_b32 v136, %0\n\tv_mov_b32 v137, %0\n\tv_mov_b32 v138, %0\n\tv_mov_b32 v139, %0\n\tv_mov_b32 v140, %0\n\tv_mov_b32 v141, %0\n\tv_mov_b32 v142, %0\n\tv_mov_b32 v143, %0\n\tv_mov_b32 v144, %0\n\tv_mov_b32 v145, %0\n\tv_mov_b32 v146, %0\n\tv_mov_b32 v147, %0\n\tv_mov_b32 v148, %0\n\tv_mov_b32 v149, %0\n\tv_mov_b32 v150, %0\n\tv_mov_b32 v151, %0\n\tv_mov_b32 v152, %0\n\tv_mov_b32 v153, %0\n\tv_mov_b32 v154, %0\n\tv_mov_b32 v155, %0\n\tv_mov_b32 v156, %0\n\tv_mov_b32 v157, %0\n\tv_mov_b32 v158, %0\n\tv_mov_b32 v159, %0\n\tv_mov_b32 v160, %0\n\tv_mov_b32 v161, %0\n\tv_mov_b32 v162, %0\n\tv_mov_b32 v163, %0\n\tv_mov_b32 v164, %0\n\tv_mov_b32 v165, %0\n\tv_mov_b32 v166, %0\n\tv_mov_b32 v167, %0\n\tv_mov_b32 v168, %0\n\tv_mov_b32 v169, %0\n\tv_mov_b32 v170, %0\n\tv_mov_b32 v171, %0\n\tv_mov_b32 v172, %0\n\tv_mov_b32 v173, %0\n\tv_mov_b32 v174, %0\n\tv_mov_b32 v175, %0\n\tv_mov_b32 v176, %0\n\tv_mov_b32 v177, %0\n\tv_mov_b32 v178, %0\n\tv_mov_b32 v179, %0\n\tv_mov_b32 v180, %0\n\tv_mov_b32 v181, %0\n\tv_mov_b32 v182, %0\n\tv_mov_b32 v183, %0\n\tv_mov_b32 v184, %0\n\tv_mov_b32 v185, %0\n\tv_mov_b32 v186, %0\n\tv_mov_b32 v187, %0\n\tv_mov_b32 v188, %0\n\tv_mov_b32 v189, %0\n\tv_mov_b32 v190, %0\n\tv_mov_b32 v191, %0\n\tv_mov_b32 v192, %0\n\tv_mov_b32 v193, %0\n\tv_mov_b32 v194, %0\n\tv_mov_b32 v195, %0\n\tv_mov_b32 v196, %0\n\tv_mov_b32 v197, %0\n\tv_mov_b32 v198, %0\n\tv_mov_b32 v199, %0\n\tv_mov_b32 v200, %0\n\tv_mov_b32 v201, %0\n\tv_mov_b32 v202, %0\n\tv_mov_b32 v203, %0\n\tv_mov_b32 v204, %0\n\tv_mov_b32 v205, %0\n\tv_mov_b32 v206, %0\n\tv_mov_b32 v207, %0\n\tv_mov_b32 v208, %0\n\tv_mov_b32 v209, %0\n\tv_mov_b32 v210, %0\n\tv_mov_b32 v211, %0\n\tv_mov_b32 v212, %0\n\tv_mov_b32 v213, %0\n\tv_mov_b32 v214, %0\n\tv_mov_b32 v215, %0\n\tv_mov_b32 v216, %0\n\tv_mov_b32 v217, %0\n\tv_mov_b32 v218, %0\n\tv_mov_b32 v219, %0\n\tv_mov_b32 v220, %0\n\tv_mov_b32 v221, %0\n\tv_mov_b32 v222, %0\n\tv_mov_b32 v223, %0\n\tv_mov_b32 v224, %0\n\tv_mov_b32 v225, %0\n\tv_mov_b32 v226, %0\n\tv_mov_b32 v227, %0\n\tv_mov_b32 v228, %0\n\tv_mov_b32 v229, %0\n\tv_mov_b32 v230, %0\n\tv_mov_b32 v231, %0\n\tv_mov_b32 v232, %0\n\tv_mov_b32 v233, %0\n\tv_mov_b32 v234, %0\n\tv_mov_b32 v235, %0\n\tv_mov_b32 v236, %0\n\tv_mov_b32 v237, %0\n\tv_mov_b32 v238, %0\n\tv_mov_b32 v239, %0\n\tv_mov_b32 v240, %0\n\tv_mov_b32 v241, %0\n\tv_mov_b32 v242, %0\n\tv_mov_b32 v243, %0\n\tv_mov_b32 v244, %0\n\tv_mov_b32 v245, %0\n\tv_mov_b32 v246, %0\n\tv_mov_b32 v247, %0\n\tv_mov_b32 v248, %0\n\tv_mov_b32 v249, %0\n\tv_mov_b32 v250, %0\n\tv_mov_b32 v251, %0\n\tv_mov_b32 v252, %0\n\tv_mov_b32 v253, %0\n\tv_mov_b32 v254, %0\n\tv_mov_b32 v255, %0\n\ts_mov_b32 s16, %1\n\ts_mov_b32 s17, %1\n\ts_mov_b32 s18, %1\n\ts_mov_b32 s19, %1\n\ts_mov_b32 s20, %1\n\ts_mov_b32 s21, %1\n\ts_mov_b32 s22, %1\n\ts_mov_b32 s23, %1\n\ts_mov_b32 s24, %1\n\ts_mov_b32 s25, %1\n\ts_mov_b32 s26, %1\n\ts_mov_b32 s27, %1\n\ts_mov_b32 s28, %1\n\ts_mov_b32 s29, %1\n\ts_mov_b32 s30, %1\n\ts_mov_b32 s31, %1\n\ts_mov_b32 s32, %1\n\ts_mov_b32 s33, %1\n\ts_mov_b32 s34, %1\n\ts_mov_b32 s35, %1\n\ts_mov_b32 s36, %1\n\ts_mov_b32 s37, %1\n\ts_mov_b32 s38, %1\n\ts_mov_b32 s39, %1\n\ts_mov_b32 s40, %1\n\ts_mov_b32 s41, %1\n\ts_mov_b32 s42, %1\n\ts_mov_b32 s43, %1\n\ts_mov_b32 s44, %1\n\ts_mov_b32 s45, %1\n\ts_mov_b32 s46, %1\n\ts_mov_b32 s47, %1\n\ts_mov_b32 s48, %1\n\ts_mov_b32 s49, %1\n\ts_mov_b32 s50, %1\n\ts_mov_b32 s51, %1\n\ts_mov_b32 s52, %1\n\ts_mov_b32 s53, %1\n\ts_mov_b32 s54, %1\n\ts_mov_b32 s55, %1\n\ts_mov_b32 s56, %1\n\ts_mov_b32 s57, %1\n\ts_mov_b32 s58, %1\n\ts_mov_b32 s59, %1\n\ts_mov_b32 s60, %1\n\ts_mov_b32 s61, %1\n\ts_mov_b32 s62, %1\n\ts_mov_b32 s63, %1\n\ts_mov_b32 s64, %1\n\ts_mov_b32 s65, %1\n\ts_mov_b32 s66, %1\n\ts_mov_b32 s67, %1\n\ts_mov_b32 s68, %1\n\ts_mov_b32 s69, %1\n\ts_mov_b32 s70, %1\n\ts_mov_b32 s71, %1\n\ts_mov_b32 s72, %1\n\ts_mov_b32 s73, %1\n\ts_mov_b32 s74, %1\n\ts_mov_b32 s75, %1\n\ts_mov_b32 s76, %1\n\ts_mov_b32 s77, %1\n\ts_mov_b32 s78, %1\n\ts_mov_b32 s79, %1\n\ts_mov_b32 s80, %1\n\ts_mov_b32 s81, %1\n\ts_mov_b32 s82, %1\n\ts_mov_b32 s83, %1\n\ts_mov_b32 s84, %1\n\ts_mov_b32 s85, %1\n\ts_mov_b32 s86, %1\n\ts_mov_b32 s87, %1\n\ts_mov_b32 s88, %1\n\ts_mov_b32 s89, %1\n\ts_mov_b32 s90, %1\n\ts_mov_b32 s91, %1\n\ts_mov_b32 s92, %1\n\ts_mov_b32 s93, %1\n\ts_mov_b32 s94, %1\n\ts_mov_b32 s95, %1\n\ts_mov_b32 s96, %1\n\ts_mov_b32 s97, %1\n\ts_mov_b32 s98, %1\n\ts_mov_b32 s99, %1\n\ts_nop 0" :: "v"(p), "s"(q) : "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191","v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239","v240","v241","v242","v243","v244","v245","v246","v247","v248","v249","v250","v251","v252","v253","v254","v255", "s16","s17","s18","s19","s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35","s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63","s64","s65","s66","s67","s68","s69","s70","s71","s72","s73","s74","s75","s76","s77","s78","s79","s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95","s96","s97","s98","s99");
  if (p == 1u) sink[1] = 1;
}
void LaunchLdsPoison(unsigned *sink, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&LdsPoisonKernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(LdsPoisonKernel, dim3(512), dim3(1024), 160 * 1024, s, 0x7fc00000u, sink);
  hipLaunchKernelGGL(RegPoisonKernel, dim3(16384), dim3(64), 0, s, 0x7fc00000u, sink);
}

}  // namespace rs
