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

}  // namespace rs
