// Feature front-end kernels for gfx950: batched MFCC, sliding-window CMVN, UBM posteriors and the online
// iVector estimator.  One 64-lane wavefront per frame for the per-frame kernels (CDNA4 wave64; never 32).
//
// Reference behaviour being reproduced (kaldi/src):
//   feat/feature-window.cc:90-224 (DC removal, pre-emphasis, window, zero padding)
//   matrix/srfft.cc:356-432 + feat/feature-functions.cc:29-51 (real FFT -> 257-bin power spectrum)
//   feat/mel-computations.cc:226-251, feat/feature-mfcc.cc:28-80 (mel, log, DCT, lifter)
//   feat/online-feature.cc:337-452 + transform/cmvn.cc:64-91 (OnlineCmvn)
//   gmm/diag-gmm.cc:546-562 + hmm/posterior.cc:440-509 (UBM log-likes, posterior pruning)
//   ivector/ivector-extractor.cc:611-668,732-756 + matrix/optimization.cc:453-566 (stats, CG solve)
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
// utterances, no special cases downstream).  512-point real FFT = 256-point complex Stockham radix-4 FFT
// (4 passes, one butterfly per lane per pass, ping-pong in LDS) + untangle.
template <int NFFT>   // padded window (real points)
__global__ __launch_bounds__(256) void MfccKernel(MfccDev m, BatchGeom g, const int16_t *__restrict__ pcm,
                                                  float *__restrict__ feats, int ld) {
  constexpr int NC = NFFT / 2;        // complex points
  constexpr int WPB = 4;              // waves (frames) per block
  __shared__ float2 bufA[WPB][NC];
  __shared__ float2 bufB[WPB][NC];
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
  float *x = reinterpret_cast<float *>(bufA[wave]);   // NFFT floats view
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
  float y[(NFFT / 2 + 63) / 64 * 2];   // pre-emphasised + windowed samples owned by this lane
  if (active) {
    // 2. pre-emphasis (uses the *un-emphasised* left neighbour, as the backwards loop of the reference does)
    int n = 0;
    for (int i = lane; i < m.win; i += RS_WAVE, n++) {
      float prev = x[i > 0 ? i - 1 : 0];
      float v = x[i] - m.preemph * prev;
      y[n] = v * m.window[i];
    }
  }
  __syncthreads();
  if (active) {
    int n = 0;
    for (int i = lane; i < m.win; i += RS_WAVE, n++) x[i] = y[n];
    if (m.use_energy && !m.raw_energy) {
      float e = 0.f;
      n = 0;
      for (int i = lane; i < m.win; i += RS_WAVE, n++) e += y[n] * y[n];
      raw_energy = logf(fmaxf(WaveSum(e), FLT_EPSILON));
    }
  }
  __syncthreads();
  // 3. complex FFT of z[n] = x[2n] + i x[2n+1]  (bufA already holds it: float2 view of x)
  float2 *in = bufA[wave], *out = bufB[wave];
  const float2 *tw = reinterpret_cast<const float2 *>(m.twiddle);   // W_NC^k = (cos, -sin)(2 pi k / NC), k < NC
  for (int Ns = 1; Ns < NC; Ns *= 4) {
    if (active) {
      for (int j = lane; j < NC / 4; j += RS_WAVE) {
        int k = j % Ns;
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float2 a = in[j + r * (NC / 4)];
          int ti = (k * r * (NC / (Ns * 4))) % NC;
          float2 w = tw[ti];
          v[r] = make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
        }
        // radix-4 butterfly (forward transform, W_4 = -i)
        float2 s0 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y), d0 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
        float2 s1 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y), d1 = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
        int j0 = (j / Ns) * Ns * 4 + k;
        out[j0] = make_float2(s0.x + s1.x, s0.y + s1.y);
        out[j0 + Ns] = make_float2(d0.x + d1.y, d0.y - d1.x);
        out[j0 + 2 * Ns] = make_float2(s0.x - s1.x, s0.y - s1.y);
        out[j0 + 3 * Ns] = make_float2(d0.x - d1.y, d0.y + d1.x);
      }
    }
    __syncthreads();
    float2 *tmp = in; in = out; out = tmp;
  }
  // NC = 256 -> 4 passes (even) so the result is back in bufA; NC = 128 (3.5 passes) is not a power of 4
  // 4. untangle -> power spectrum of the real transform, bins 0..NC
  if (active) {
    const float2 *tw2 = tw + NC;    // W_NFFT^k, k <= NC
    for (int k = lane; k <= NC; k += RS_WAVE) {
      float2 zk = in[k & (NC - 1)], zn = in[(NC - k) & (NC - 1)];
      float er = 0.5f * (zk.x + zn.x), ei = 0.5f * (zk.y - zn.y);     // even part
      float orr = 0.5f * (zk.y + zn.y), oi = -0.5f * (zk.x - zn.x);   // odd part (already times -i)
      float2 w = tw2[k];
      float xr = er + (orr * w.x - oi * w.y), xi = ei + (orr * w.y + oi * w.x);
      pw[wave][k] = xr * xr + xi * xi;
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
  if (blocks == 0) return;
  if (m.padded == 512) hipLaunchKernelGGL(MfccKernel<512>, dim3(blocks), dim3(256), 0, s, m, g, pcm, feats, ld);
  else hipLaunchKernelGGL(MfccKernel<2048>, dim3(blocks), dim3(256), 0, s, m, g, pcm, feats, ld);
}

// ------------------------------------------------------------------------------------------ online CMVN
// One thread per (utterance, dim); frames are walked sequentially with the same add-new / subtract-old
// double-precision update ComputeStatsForFrame performs, then SmoothOnlineCmvnStats with the global stats
// (no speaker stats: the reference starts every utterance from a fresh process) and mean-only ApplyCmvn.
__global__ void OnlineCmvnKernel(CmvnDev c, BatchGeom g, const float *__restrict__ in, float *__restrict__ out, int ld) {
  int u = blockIdx.x, d = threadIdx.x;
  if (d >= c.dim) return;
  int T = g.d_num_frames[u];
  size_t base = (size_t)g.d_row_base[u] + g.L;
  double sum = 0.0, count = 0.0;
  const double gsum = c.global_stats[d], gcount = c.global_stats[c.dim];
  float first = 0.f, last = 0.f;
  for (int t = 0; t < T; t++) {
    float xv = in[(base + t) * ld + d];
    sum += (double)xv;
    count += 1.0;
    int prev = t - c.cmn_window;
    if (prev >= 0) {
      sum -= (double)in[(base + prev) * ld + d];
      count -= 1.0;
    }
    double s = sum, n = count;
    if (n < (double)c.cmn_window) {
      double from_global = (double)c.cmn_window - n;
      if (from_global > (double)c.global_frames) from_global = (double)c.global_frames;
      if (from_global > 0.0) {
        double a = from_global / gcount;
        s += a * gsum;
        n += a * gcount;
      }
    }
    float alpha = (float)(-1.0 / n);
    float offset = (float)((double)alpha * s);
    float yv = xv + offset;
    out[(base + t) * ld + d] = yv;
    if (t == 0) first = yv;
    last = yv;
  }
  for (int t = -g.L; t < 0; t++) out[(base + t) * ld + d] = first;
  for (int t = T; t < T + g.R; t++) out[(base + t) * ld + d] = last;
}

void LaunchOnlineCmvn(const CmvnDev &c, const BatchGeom &g, const float *in, float *out, int ld, hipStream_t s) {
  if (g.n_utts == 0) return;
  int threads = ((c.dim + 63) / 64) * 64;
  hipLaunchKernelGGL(OnlineCmvnKernel, dim3(g.n_utts), dim3(threads), 0, s, c, g, in, out, ld);
}

// ------------------------------------------------------------------------------------------ UBM posteriors
// One wave per row.  Lane l scores Gaussians l, l+64, ... against the frame (parameters stored transposed,
// D x G, so that a wave reads 64 consecutive floats per dimension), then the wave extracts the top
// num_gselect posteriors exactly as VectorToPosteriorEntry does.
template <int NPL>   // Gaussians per lane (G <= 64 * NPL)
__global__ __launch_bounds__(256) void UbmPostKernel(IvecDev iv, BatchGeom g, const float *__restrict__ feats, int ld,
                                                     int *__restrict__ post_idx, float *__restrict__ post_w) {
  __shared__ float xs[4][128];
  __shared__ float x2[4][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  bool active = row < g.total_rows;
  const int D = iv.feat_dim, G = iv.num_gauss, nsel = iv.num_gselect;
  if (active) {
    int u = g.d_row_utt[row], t = g.d_row_t[row];
    if (t < 0 || t >= g.d_num_frames[u]) active = false;
  }
  if (active)
    for (int d = lane; d < D; d += RS_WAVE) {
      float v = feats[(size_t)row * ld + d];
      xs[wave][d] = v;
      x2[wave][d] = v * v;
    }
  __syncthreads();
  if (!active) {
    if (row < g.total_rows && lane < nsel) post_idx[(size_t)row * nsel + lane] = -1;
    return;
  }
  float a1[NPL], a2[NPL], ll[NPL];
#pragma unroll
  for (int j = 0; j < NPL; j++) { a1[j] = 0.f; a2[j] = 0.f; }
  for (int d = 0; d < D; d++) {
    const float xv = xs[wave][d], xq = x2[wave][d];
    const float *mi = iv.means_invvars_t + (size_t)d * G + lane, *vi = iv.inv_vars_t + (size_t)d * G + lane;
#pragma unroll
    for (int j = 0; j < NPL; j++)
      if (lane + j * RS_WAVE < G) { a1[j] += xv * mi[j * RS_WAVE]; a2[j] += xq * vi[j * RS_WAVE]; }
  }
  float lmax = -INFINITY;
#pragma unroll
  for (int j = 0; j < NPL; j++) {
    int gi = lane + j * RS_WAVE;
    float v = -INFINITY;
    if (gi < G) {
      v = iv.gconsts[gi] + a1[j];
      v = v + (-0.5f) * a2[j];
    }
    ll[j] = v;
    lmax = fmaxf(lmax, v);
  }
  const float max_like = WaveMax(lmax);
  const float like_cutoff = max_like + logf(iv.min_post);
  // posteriors of the candidates (exp in double, as the reference's `exp(like - max_like)` does)
#pragma unroll
  for (int j = 0; j < NPL; j++) ll[j] = (ll[j] > like_cutoff) ? (float)exp((double)(ll[j] - max_like)) : -1.f;
  // top-nsel extraction, best first; ties -> lowest Gaussian index.  Every lane keeps the full selection.
  float sel_w[8];
  int sel_i[8];
  int nfound = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    sel_w[k] = 0.f;
    sel_i[k] = -1;
    if (k < nsel && nfound == k) {
      float bv = -1.f;
      int bg = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < NPL; j++) if (ll[j] > bv) { bv = ll[j]; bg = lane + j * RS_WAVE; }
      float wv = bv;
      int wg = bg;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(wv, o, RS_WAVE);
        int og = __shfl_xor(wg, o, RS_WAVE);
        if (ov > wv || (ov == wv && og < wg)) { wv = ov; wg = og; }
      }
      if (wv >= 0.f) {
        sel_w[k] = wv;
        sel_i[k] = wg;
        nfound = k + 1;
#pragma unroll
        for (int j = 0; j < NPL; j++) if (lane + j * RS_WAVE == wg) ll[j] = -1.f;
      }
    }
  }
  // prune + renormalise (posterior.cc:494-507), identical on every lane
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 8; k++) if (k < nfound) tot += sel_w[k];
  const float cutoff = iv.min_post * tot;
#pragma unroll
  for (int k = 7; k >= 1; k--)
    if (nfound == k + 1 && sel_w[k] < cutoff) { tot -= sel_w[k]; nfound = k; }
  const float inv_tot = (float)(1.0 / (double)tot);
  const float scale = iv.posterior_scale * 1.0f;
  float w = 0.f;
  int gi = -1;
#pragma unroll
  for (int k = 0; k < 8; k++)
    if (k == lane && k < nfound) { w = sel_w[k] * inv_tot; w *= scale; gi = sel_i[k]; }
  if (lane < nsel) {
    post_idx[(size_t)row * nsel + lane] = gi;
    post_w[(size_t)row * nsel + lane] = w;
  }
}

void LaunchUbmPosteriors(const IvecDev &iv, const BatchGeom &g, const float *lda_norm, int ld, int *post_idx,
                         float *post_w, hipStream_t s) {
  int blocks = (g.total_rows + 3) / 4;
  if (!blocks) return;
  int npl = (iv.num_gauss + 63) / 64;
#define RS_UBM(N) hipLaunchKernelGGL(UbmPostKernel<N>, dim3(blocks), dim3(256), 0, s, iv, g, lda_norm, ld, post_idx, post_w)
  if (npl <= 1) RS_UBM(1);
  else if (npl <= 2) RS_UBM(2);
  else if (npl <= 4) RS_UBM(4);
  else if (npl <= 8) RS_UBM(8);
  else if (npl <= 16) RS_UBM(16);
  else RS_UBM(32);
#undef RS_UBM
}

// ------------------------------------------------------------------------------------------ iVector stats
// Block per utterance; frames in order so that every per-Gaussian sum is accumulated in the reference's
// frame order (AccStats: weighted_feats.AddVec per frame, float tot_weight).
__global__ __launch_bounds__(256) void IvecAccumKernel(IvecDev iv, BatchGeom g, const float *__restrict__ lda, int ld,
                                                        const int *__restrict__ post_idx, const float *__restrict__ post_w,
                                                        const int *frame_begin, const int *frame_end,
                                                        float *__restrict__ gamma, double *__restrict__ wfeats) {
  int u = blockIdx.x;
  int T = g.d_num_frames[u];
  int t0 = frame_begin ? frame_begin[u] : 0, t1 = frame_end ? frame_end[u] : T;
  if (t1 > T) t1 = T;
  int D = iv.feat_dim, G = iv.num_gauss, nsel = iv.num_gselect;
  size_t base = (size_t)g.d_row_base[u] + g.L;
  float *gm = gamma + (size_t)u * G;
  double *wf = wfeats + (size_t)u * G * D;
  for (int t = t0; t < t1; t++) {
    size_t row = base + t;
    for (int i = threadIdx.x; i < nsel * D; i += blockDim.x) {
      int j = i / D, d = i % D;
      int gi = post_idx[row * nsel + j];
      if (gi >= 0) {
        float w = post_w[row * nsel + j];
        wf[(size_t)gi * D + d] += (double)w * (double)lda[row * ld + d];
        if (d == 0) gm[gi] += w;
      }
    }
    __syncthreads();
  }
}

void LaunchIvecAccumulate(const IvecDev &iv, const BatchGeom &g, const float *lda, int ld, const int *post_idx,
                          const float *post_w, const int *frame_begin, const int *frame_end, double *gamma,
                          double *wfeats, hipStream_t s) {
  if (g.n_utts == 0) return;
  // gamma is kept in float (GaussInfo::tot_weight is a BaseFloat); the buffer is sized for doubles, we use
  // its first half as floats.
  hipLaunchKernelGGL(IvecAccumKernel, dim3(g.n_utts), dim3(256), 0, s, iv, g, lda, ld, post_idx, post_w, frame_begin,
                     frame_end, reinterpret_cast<float *>(gamma), wfeats);
}

// linear += sum_g Sigma_inv_M_g^T wf_g ; thread per (utt, i)
__global__ void IvecLinearKernel(IvecDev iv, const float *__restrict__ gamma, const double *__restrict__ wfeats,
                                 double *__restrict__ linear) {
  int u = blockIdx.x, i = threadIdx.x;
  int D = iv.feat_dim, G = iv.num_gauss, I = iv.ivec_dim;
  if (i >= I) return;
  double acc = 0.0;
  for (int gi = 0; gi < G; gi++) {
    if (gamma[(size_t)u * G + gi] == 0.f) continue;
    const double *sim = iv.sigma_inv_M + (size_t)gi * D * I;
    const double *wf = wfeats + ((size_t)u * G + gi) * D;
    double a = 0.0;
    for (int d = 0; d < D; d++) a += sim[(size_t)d * I + i] * wf[d];
    acc += a;
  }
  linear[(size_t)u * I + i] += acc;
}

// quadratic += sum_g gamma_g U_g ; prior rescaling for max_count; num_frames update
__global__ void IvecQuadKernel(IvecDev iv, const float *__restrict__ gamma, double *__restrict__ quadratic,
                               double *__restrict__ linear, double *__restrict__ num_frames) {
  int u = blockIdx.y;
  int G = iv.num_gauss, I = iv.ivec_dim;
  int usz = I * (I + 1) / 2;
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  // total weight (double sum of the float per-Gaussian totals), recomputed by every thread (G is small)
  double tot = 0.0;
  for (int gi = 0; gi < G; gi++) tot += (double)gamma[(size_t)u * G + gi];
  double change = 0.0;
  if (iv.max_count > 0.0f) {
    double oldn = num_frames[u], newn = oldn + tot, mc = (double)iv.max_count;
    double old_scale = (oldn > mc ? oldn : mc) / mc, new_scale = (newn > mc ? newn : mc) / mc;
    change = new_scale - old_scale;
  }
  if (k < usz) {
    double acc = 0.0;
    for (int gi = 0; gi < G; gi++) {
      float gm = gamma[(size_t)u * G + gi];
      if (gm == 0.f) continue;
      acc += (double)gm * iv.U[(size_t)gi * usz + k];
    }
    // is k a diagonal element?  k = r(r+1)/2 + r
    int r = (int)((sqrt(8.0 * (double)k + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= k) r++;
    while (r * (r + 1) / 2 > k) r--;
    bool diag = (k == r * (r + 1) / 2 + r);
    quadratic[(size_t)u * usz + k] += acc + ((diag && change != 0.0) ? change : 0.0);
    if (k == 0 && change != 0.0) linear[(size_t)u * I] += iv.prior_offset * change;
  }
}
__global__ void IvecNumFramesKernel(IvecDev iv, int n_utts, const float *__restrict__ gamma, double *__restrict__ num_frames) {
  int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_utts) return;
  double tot = 0.0;
  for (int gi = 0; gi < iv.num_gauss; gi++) tot += (double)gamma[(size_t)u * iv.num_gauss + gi];
  num_frames[u] += tot;
}

// zero the per-step accumulators of the Gaussians that were touched (cheaper than a 40 MB memset per chunk)
__global__ void IvecClearKernel(IvecDev iv, float *__restrict__ gamma, double *__restrict__ wfeats) {
  const int u = blockIdx.y, gi = blockIdx.x * blockDim.y + threadIdx.y;
  if (gi >= iv.num_gauss) return;
  if (gamma[(size_t)u * iv.num_gauss + gi] == 0.f) return;
  double *wf = wfeats + ((size_t)u * iv.num_gauss + gi) * iv.feat_dim;
  for (int d = threadIdx.x; d < iv.feat_dim; d += blockDim.x) wf[d] = 0.0;
  // (one wave per Gaussian: every lane has read gamma before lane 0 clears it)
  if (threadIdx.x == 0) gamma[(size_t)u * iv.num_gauss + gi] = 0.f;
}
void LaunchIvecClear(const IvecDev &iv, int n_utts, double *gamma, double *wfeats, hipStream_t s) {
  if (n_utts == 0) return;
  dim3 block(64, 4), grid((iv.num_gauss + 3) / 4, n_utts);
  hipLaunchKernelGGL(IvecClearKernel, grid, block, 0, s, iv, reinterpret_cast<float *>(gamma), wfeats);
}

void LaunchIvecStats(const IvecDev &iv, int n_utts, const double *gamma, const double *wfeats, double *linear,
                     double *quadratic, double *num_frames, hipStream_t s) {
  if (n_utts == 0) return;
  const float *gm = reinterpret_cast<const float *>(gamma);
  int threads = ((iv.ivec_dim + 63) / 64) * 64;
  hipLaunchKernelGGL(IvecLinearKernel, dim3(n_utts), dim3(threads), 0, s, iv, gm, wfeats, linear);
  int usz = iv.ivec_dim * (iv.ivec_dim + 1) / 2;
  hipLaunchKernelGGL(IvecQuadKernel, dim3((usz + 255) / 256, n_utts), dim3(256), 0, s, iv, gm, quadratic, linear, num_frames);
  hipLaunchKernelGGL(IvecNumFramesKernel, dim3((n_utts + 63) / 64), dim3(64), 0, s, iv, n_utts, gm, num_frames);
}

// ------------------------------------------------------------------------------------------ CG solve
__device__ __forceinline__ double BlockSum(double v, double *scratch) {
  // deterministic tree reduction over the block
  int tid = threadIdx.x;
  scratch[tid] = v;
  __syncthreads();
  for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
    if (tid < o) scratch[tid] += scratch[tid + o];
    __syncthreads();
  }
  double r = scratch[0];
  __syncthreads();
  return r;
}

// y = A x for packed-lower symmetric A (row r: elements r(r+1)/2 .. +r)
__device__ __forceinline__ double SpMatVecRow(const double *A, const double *x, int r, int n) {
  double acc = 0.0;
  const double *row = A + (size_t)r * (r + 1) / 2;
  for (int c = 0; c <= r; c++) acc += row[c] * x[c];
  for (int c = r + 1; c < n; c++) acc += A[(size_t)c * (c + 1) / 2 + r] * x[c];
  return acc;
}

__global__ void IvecSolveKernel(IvecDev iv, const double *__restrict__ linear, const double *__restrict__ quadratic,
                                const double *__restrict__ num_frames, double *__restrict__ xio,
                                float *__restrict__ ivec_out, int ldo, const int *__restrict__ out_row,
                                const int *__restrict__ active) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int u = blockIdx.x, tid = threadIdx.x, n = iv.ivec_dim;
  const int orow = out_row ? out_row[u] : u;
  if (orow < 0) return;                                    // this utterance has no chunk at this step
  const bool solve = active ? active[u] != 0 : true;       // 0: re-emit the current estimate (no new frames)
  const int usz = n * (n + 1) / 2;
  double *A = reinterpret_cast<double *>(smem_raw);
  double *x = A + usz, *r = x + n, *p = r + n, *Ap = p + n, *b = Ap + n, *scratch = b + n;
  for (int i = tid; i < usz; i += blockDim.x) A[i] = quadratic[(size_t)u * usz + i];
  for (int i = tid; i < n; i += blockDim.x) { b[i] = linear[(size_t)u * n + i]; x[i] = xio[(size_t)u * n + i]; }
  __syncthreads();
  const bool have = num_frames[u] > 0.0;
  if (!solve) {
    // nothing
  } else if (have) {
    if (tid == 0 && x[0] == 0.0) x[0] = iv.prior_offset;     // GetIvector: better initial guess
    __syncthreads();
    const bool mine = tid < n;
    // p0 = b - A x0 ; r0 = -p0
    double ax = mine ? SpMatVecRow(A, x, tid, n) : 0.0;
    if (mine) { p[tid] = b[tid] - ax; r[tid] = -p[tid]; }
    __syncthreads();
    double r_cur = BlockSum(mine ? r[tid] * r[tid] : 0.0, scratch);
    const double r_initial = r_cur;
    double r_recompute = r_cur;
    const double max_error_sq = DBL_MIN, residual_factor = (double)(0.01f * 0.01f), inv_residual_factor = 1.0 / residual_factor;
    int k = 0;
    for (; k < n + 5 && k != iv.num_cg_iters; k++) {
      double apv = mine ? SpMatVecRow(A, p, tid, n) : 0.0;
      if (mine) Ap[tid] = apv;
      __syncthreads();
      double pr = BlockSum(mine ? p[tid] * r[tid] : 0.0, scratch);
      double pap = BlockSum(mine ? p[tid] * Ap[tid] : 0.0, scratch);
      double alpha = -pr / pap;
      if (mine) { x[tid] += alpha * p[tid]; r[tid] += alpha * Ap[tid]; }
      __syncthreads();
      double r_next = BlockSum(mine ? r[tid] * r[tid] : 0.0, scratch);
      if (r_next < residual_factor * r_recompute || r_next > inv_residual_factor * r_recompute) {
        double ax2 = mine ? SpMatVecRow(A, x, tid, n) : 0.0;
        if (mine) r[tid] = ax2 - b[tid];
        __syncthreads();
        r_next = BlockSum(mine ? r[tid] * r[tid] : 0.0, scratch);
        r_recompute = r_next;
      }
      if (r_next <= max_error_sq) break;
      double beta = r_next / r_cur;
      if (mine) p[tid] = p[tid] * beta - r[tid];
      __syncthreads();
      r_cur = r_next;
    }
    // (the reference falls back to an exact solve if the residual got worse; with an SPD system and <= 15
    //  iterations CG is monotone in the A-norm, the squared residual only grows in pathological cases)
    (void)r_initial;
  } else {
    if (tid < n) x[tid] = (tid == 0) ? iv.prior_offset : 0.0;
    __syncthreads();
  }
  if (tid < n) {
    if (solve) xio[(size_t)u * n + tid] = x[tid];
    float v = (float)x[tid];
    if (tid == 0) v = (float)((double)v - iv.prior_offset);   // (*feat)(0) -= PriorOffset() on the float copy
    ivec_out[(size_t)orow * ldo + tid] = v;
  }
}

void LaunchIvecSolve(const IvecDev &iv, int n_utts, const double *linear, const double *quadratic,
                     const double *num_frames, double *x, float *ivec_out, int ldo, const int *out_row, const int *active,
                     hipStream_t s) {
  if (n_utts == 0) return;
  int n = iv.ivec_dim;
  int threads = 64;
  while (threads < n) threads <<= 1;
  size_t smem = sizeof(double) * ((size_t)n * (n + 1) / 2 + 5 * (size_t)n + threads);
  hipLaunchKernelGGL(IvecSolveKernel, dim3(n_utts), dim3(threads), smem, s, iv, linear, quadratic, num_frames, x, ivec_out, ldo, out_row, active);
}

}  // namespace rs
