// Feature front-end kernels for gfx950: batched MFCC and sliding-window CMVN (the iVector estimator lives in
// ivector_kernels.hip).  One 64-lane wavefront per frame for the per-frame kernels (CDNA4 wave64; never 32).
//
// Reference behaviour being reproduced (kaldi/src):
//   feat/feature-window.cc:90-224 (DC removal, pre-emphasis, window, zero padding)
//   matrix/srfft.cc:356-432 + feat/feature-functions.cc:29-51 (real FFT -> 257-bin power spectrum)
//   feat/mel-computations.cc:226-251, feat/feature-mfcc.cc:28-80 (mel, log, DCT, lifter)
//   feat/online-feature.cc:337-452 + transform/cmvn.cc:64-91 (OnlineCmvn)
#include <hip/hip_runtime.h>
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

template <int NFFT>   // padded window (real points)
__global__ __launch_bounds__(256) void MfccKernel(MfccDev m, BatchGeom g, const int16_t *__restrict__ pcm,
                                                  float *__restrict__ feats, int ld) {
  constexpr int NC = NFFT / 2;        // complex points
  const int4 *tasks = reinterpret_cast<const int4 *>(m.fft_tasks);
  const float *fft_tw = m.fft_tw;
  constexpr int WPB = 4;              // waves (frames) per block
  __shared__ float xbuf[WPB][NFFT];
  __shared__ float xrb[WPB][NC];
  __shared__ float xib[WPB][NC];
  __shared__ float pw[WPB][NC + 1];
  __shared__ float lm[WPB][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * WPB + wave;
  const bool active = row < g.total_rows;
  int u = 0, t = 0;
  if (active) {
    u = g.d_row_utt[row];
    t = g.d_row_t[row];
    int T = g.d_num_frames[u];
    t = t < 0 ? 0 : (t >= T ? T - 1 : t);
  }
  float *x = xbuf[wave];
  float raw_energy = 0.f;
  if (active) {
    const int16_t *src = pcm + g.d_sample_off[u] + (int64_t)t * m.shift;
    // 1. load (int16 -> float, unscaled), DC removal
    float part = 0.f;
    for (int i = lane; i < m.win; i += RS_WAVE) {
      float v = (float)src[i];
      x[i] = v;
      part += v;
    }
    for (int i = m.win + lane; i < NFFT; i += RS_WAVE) x[i] = 0.f;
    float mean = WaveSum(part) / (float)m.win;
    if (m.remove_dc)
      for (int i = lane; i < m.win; i += RS_WAVE) x[i] -= mean;
    if (m.use_energy && m.raw_energy) {
      float e = 0.f;
      for (int i = lane; i < m.win; i += RS_WAVE) e += x[i] * x[i];
      raw_energy = logf(fmaxf(WaveSum(e), FLT_EPSILON));
    }
  }
  __syncthreads();
  float *xr = xrb[wave], *xi = xib[wave];
  if (active) {
    // 2. pre-emphasis (uses the *un-emphasised* left neighbour, as the backwards loop of the reference does) and
    // window; the even / odd samples are the real / imaginary parts of the half-length complex transform
    float e = 0.f;
    for (int i = lane; i < NFFT; i += RS_WAVE) {
      float y = 0.f;
      if (i < m.win) {
        const float prev = x[i > 0 ? i - 1 : 0];
        const float v = x[i] - m.preemph * prev;
        y = v * m.window[i];
        e += y * y;
      }
      if (i & 1) xi[i >> 1] = y; else xr[i >> 1] = y;
    }
    if (m.use_energy && !m.raw_energy) raw_energy = logf(fmaxf(WaveSum(e), FLT_EPSILON));
  }
  __syncthreads();
  // 3. split-radix complex FFT, level by level (tasks of one level touch disjoint points)
  for (int L = 0; L < m.fft_num_levels; L++) {
    if (active)
      for (int ti = m.fft_level_begin[L] + lane; ti < m.fft_level_begin[L + 1]; ti += RS_WAVE) SrfftRunTask(tasks[ti], fft_tw, xr, xi);
    __syncthreads();
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
  __syncthreads();
  // 5. mel filterbank + log
  if (active && lane < m.nbins) {
    int off = m.mel_offset[lane], len = m.mel_len[lane];
    const float *w = m.mel_weights + m.mel_start[lane];
    float e = 0.f;
    for (int i = 0; i < len; i++) e += w[i] * pw[wave][off + i];
    lm[wave][lane] = logf(fmaxf(e, FLT_EPSILON));
  }
  __syncthreads();
  // 6. DCT + lifter, write the row
  if (active && lane < m.nceps) {
    const float *d = m.dct + lane * m.nbins;
    float c = 0.f;
    for (int b = 0; b < m.nbins; b++) c += d[b] * lm[wave][b];
    c *= m.lifter[lane];
    if (m.use_energy && lane == 0) c = fmaxf(raw_energy, m.log_energy_floor);
    feats[(size_t)row * ld + lane] = c;
  }
}

void LaunchMfcc(const MfccDev &m, const BatchGeom &g, const int16_t *pcm, float *feats, int ld, hipStream_t s) {
  int blocks = (g.total_rows + 3) / 4;
  if (!blocks) return;
  if (m.padded == 512) hipLaunchKernelGGL(MfccKernel<512>, dim3(blocks), dim3(256), 0, s, m, g, pcm, feats, ld);
  else hipLaunchKernelGGL(MfccKernel<2048>, dim3(blocks), dim3(256), 0, s, m, g, pcm, feats, ld);
}

// ------------------------------------------------------------------------------------------ online CMVN
// Workgroup per utterance, frames in chunks staged through LDS.  Per chunk: (1) thread d walks the frames with the same
// add-new / subtract-old double-precision update ComputeStatsForFrame performs (the only sequential part: two adds
// per frame), (2) one thread per frame does SmoothOnlineCmvnStats' per-frame scalars (the two fp64 divisions), (3) all
// threads apply the mean-only ApplyCmvn to the chunk in parallel with coalesced stores.  No speaker stats: the
// reference starts every utterance from a fresh process.
constexpr int kCmvnTC = 32, kCmvnMaxDim = 128;
__global__ __launch_bounds__(256) void OnlineCmvnKernel(CmvnDev c, BatchGeom g, const float *__restrict__ in, float *__restrict__ out, int ld) {
  __shared__ float xs[kCmvnTC][kCmvnMaxDim];      // the chunk
  __shared__ float xp[kCmvnTC][kCmvnMaxDim];      // the frames leaving the window while the chunk enters
  __shared__ double ss[kCmvnTC][kCmvnMaxDim];     // running sums after each frame
  __shared__ double nn[kCmvnTC], aa[kCmvnTC];     // frame count in the window; weight of the global stats
  __shared__ float al[kCmvnTC];                   // -1 / (smoothed count)
  __shared__ float edge[2][kCmvnMaxDim];          // normalised first / last frame (halo rows replicate them)
  const int u = blockIdx.x, tid = threadIdx.x, D = c.dim, W = c.cmn_window;
  const int T = g.d_num_frames[u];
  const size_t base = (size_t)g.d_row_base[u] + g.L;
  double sum = 0.0, count = 0.0;                  // threads < D
  const double gcount = c.global_stats[D];
  for (int t0 = 0; t0 < T; t0 += kCmvnTC) {
    const int n = T - t0 < kCmvnTC ? T - t0 : kCmvnTC;
    for (int idx = tid; idx < n * D; idx += 256) {
      const int i = idx / D, d = idx % D, tp = t0 + i - W;
      xs[i][d] = in[(base + t0 + i) * ld + d];
      xp[i][d] = tp >= 0 ? in[(base + tp) * ld + d] : 0.f;
    }
    __syncthreads();
    if (tid < D) {
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
      if (a > 0.0) sv += a * c.global_stats[d];
      const float offset = (float)((double)al[i] * sv);
      const float yv = xs[i][d] + offset;
      out[(base + t0 + i) * ld + d] = yv;
      if (t0 + i == 0) edge[0][d] = yv;
      if (t0 + i == T - 1) edge[1][d] = yv;
    }
    __syncthreads();
  }
  if (T > 0) {
    for (int idx = tid; idx < g.L * D; idx += 256) out[(base - g.L + idx / D) * ld + idx % D] = edge[0][idx % D];
    for (int idx = tid; idx < g.R * D; idx += 256) out[(base + T + idx / D) * ld + idx % D] = edge[1][idx % D];
  }
}

void LaunchOnlineCmvn(const CmvnDev &c, const BatchGeom &g, const float *in, float *out, int ld, hipStream_t s) {
  if (g.n_utts == 0) return;
  hipLaunchKernelGGL(OnlineCmvnKernel, dim3(g.n_utts), dim3(256), 0, s, c, g, in, out, ld);
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

}  // namespace rs
