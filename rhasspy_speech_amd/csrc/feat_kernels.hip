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
#include "env.h"

namespace rs {

#define RS_WAVE 64

__device__ __forceinline__ float WaveSum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, RS_WAVE);
  return v;
}
// sum of a double over the wave: row rotations inside the rows of 16 lanes, then the four row results through SGPRs
__device__ __forceinline__ double WaveSumF64(double v) {
#define RS_DPP_D(CTRL)                                                                                                  \
  v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true),                       \
                        __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true));
  RS_DPP_D(0x121) RS_DPP_D(0x122) RS_DPP_D(0x124) RS_DPP_D(0x128)
#undef RS_DPP_D
  auto rl = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
  return (rl(0) + rl(16)) + (rl(32) + rl(48));
}
__device__ __forceinline__ float WaveMax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, RS_WAVE));
  return v;
}

// ------------------------------------------------------------------------------------------ MFCC
// One wave per frame; the rows of an utterance's halo (copies of its first / last frame) are written by the waves of those
// two frames.
//
// The real FFT is the reference's own algorithm (matrix/srfft.cc: in-place single-precision split radix + a
// post-processing pass whose twiddle comes from a float recurrence), restated as levels of independent butterfly
// tasks (srfft_plan.h) executed lane-parallel with the same float operations on the same operands, so the power
// spectrum matches the reference to the bit instead of to its own ~1e-3 rounding noise.  This file is compiled with
// -ffp-contract=off for that reason.
__device__ __forceinline__ void SrfftRunTask(const int4 tk, const float *__restrict__ tw, float *xr, float *xi, const float *tw_inline = nullptr) {
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
      const float *w = tw_inline ? tw_inline : tw + (size_t)tk.w * 6;
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

// The same task from a 48-byte record of the padded, pre-addressed plan (MfccDev::fft_recs: 64 records per level, one per lane):
//   {byte offsets of the task's (up to) four points inside xr / xi; kind (3 = no task for this lane) | twiddle class << 8 (0 factors
//    below, 1 none, 2 the sqrt(1/2) case), the six twiddle factors; -}
// The offsets are those of the SWIZZLED layout (MfccDev::fft_swz): point i of the transform lives at index i ^ g(i >> 5).  In the
// plain layout the small blocks of the in-place split radix -- points 8 k + n, 16 k + n, ... -- put a half wave's 32 reads on 8 or
// 4 of the 32 LDS banks: the kernel's LDS pipe was busy 83 % of the time and half of that was bank conflicts (SQ_LDS_IDX_ACTIVE /
// SQ_LDS_BANK_CONFLICT, profiles/micro/pmc_lds.sh); the swizzle (found by exhaustive search over the linear ones against this plan's
// access patterns, profiles/micro/fft_swizzle.py) halves the passes of the levels and of the post-processing gather (636 -> 308
// per frame; 208 would be conflict-free).  Same float operations on the same operands.
__device__ __forceinline__ void SrfftRunRec(const float4 r0, const float4 r1, const float4 r2, float *xr, int xi_off) {
  const int meta = __float_as_int(r1.x), kind = meta & 0xff, twc = meta >> 8;
  char *xb = reinterpret_cast<char *>(xr);
  float *q0 = reinterpret_cast<float *>(xb + __float_as_int(r0.x)), *q1 = reinterpret_cast<float *>(xb + __float_as_int(r0.y));
  float *q2 = reinterpret_cast<float *>(xb + __float_as_int(r0.z)), *q3 = reinterpret_cast<float *>(xb + __float_as_int(r0.w));
  if (kind == 0) {
    const float ar = q0[0], ai = q0[xi_off], br = q1[0], bi = q1[xi_off], cr = q2[0], ci = q2[xi_off], dr = q3[0], di = q3[xi_off];
    q0[0] = ar + cr; q0[xi_off] = ai + ci;
    q1[0] = br + dr; q1[xi_off] = bi + di;
    const float p_r = ar - cr, p_i = ai - ci, q_r = br - dr, q_i = bi - di;
    float r1v = p_r + q_i, i2 = p_i + q_r, i1 = p_i - q_r, r2v = p_r - q_i;
    if (twc == 2) {
      const float sqhalf = 0.70710678118654752440f;
      const float t1 = sqhalf * (r1v + i1);
      i1 = sqhalf * (i1 - r1v);
      r1v = t1;
      const float t2 = sqhalf * (i2 - r2v);
      i2 = -sqhalf * (r2v + i2);
      r2v = t2;
    } else if (twc == 0) {
      const float cn = r1.y, spcn = r1.z, smcn = r1.w, c3n = r2.x, spc3n = r2.y, smc3n = r2.z;
      float t2 = cn * (r1v + i1);
      float t1 = spcn * r1v + t2;
      r1v = smcn * i1 + t2;
      i1 = t1;
      t2 = c3n * (r2v + i2);
      t1 = spc3n * r2v + t2;
      r2v = smc3n * i2 + t2;
      i2 = t1;
    }
    q2[0] = r1v; q2[xi_off] = i1;
    q3[0] = r2v; q3[xi_off] = i2;
  } else if (kind == 1) {
    float a0 = q0[0], a1 = q1[0], a2 = q2[0], a3 = q3[0];
    float i0 = q0[xi_off], i1 = q1[xi_off], i2 = q2[xi_off], i3 = q3[xi_off];
    float t;
    t = a0 + a2; a2 = a0 - a2; a0 = t;
    t = i0 + i2; i2 = i0 - i2; i0 = t;
    t = a1 + a3; a3 = a1 - a3; a1 = t;
    t = i1 + i3; i3 = i1 - i3; i1 = t;
    t = a0 + a1; a1 = a0 - a1; a0 = t;
    t = i0 + i1; i1 = i0 - i1; i0 = t;
    const float t1 = a2 + i3, t2 = i2 + a3;
    i2 = i2 - a3;
    a3 = a2 - i3;
    a2 = t1;
    i3 = t2;
    q0[0] = a0; q1[0] = a1; q2[0] = a2; q3[0] = a3;
    q0[xi_off] = i0; q1[xi_off] = i1; q2[xi_off] = i2; q3[xi_off] = i3;
  } else if (kind == 2) {
    const float a0 = q0[0], a1 = q1[0], i0 = q0[xi_off], i1 = q1[xi_off];
    q0[0] = a0 + a1; q1[0] = a0 - a1;
    q0[xi_off] = i0 + i1; q1[xi_off] = i0 - i1;
  }
}

// The waves of a workgroup work on different frames, each in its own slices of the LDS arrays: a wave only has to order
// its own LDS traffic (its earlier writes land before its later reads), no wave ever waits for another one.
__device__ __forceinline__ void WaveLdsSync() {
  __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// WPB = waves (frames) per workgroup.  4 by default; 16 with all the CU's LDS requested when several decode pipelines are in
// flight, so that no GemmKernelB3 workgroup of another pipeline can share the CU (that kernel
// perturbs this one's LDS-staged arithmetic when they share a CU).
// TAB: the FFT plan's records (21.5 KB for a 512-point window) are copied into LDS once per workgroup and every wave reads its records
// from there: 21 of a frame's ~60 vector memory requests become 1.3 (WPB = 16), and it is their number the kernel is bound by.
template <int NFFT, int WPB, bool TAB = false>   // padded window (real points)
__global__ __launch_bounds__(64 * WPB) void MfccKernel(MfccDev m, BatchGeom g, const int16_t *__restrict__ pcm,
                                                       float *__restrict__ feats, int ld, const int *__restrict__ out_rows) {
  constexpr int NC = NFFT / 2;        // complex points
  const int4 *tasks = reinterpret_cast<const int4 *>(m.fft_tasks);
  const float *fft_tw = m.fft_tw;
  extern __shared__ __attribute__((aligned(16))) float mfcc_lds[];
  float (*xrb)[NC] = reinterpret_cast<float (*)[NC]>(mfcc_lds);
  float (*xib)[NC] = reinterpret_cast<float (*)[NC]>(mfcc_lds + WPB * NC);
  float (*pw)[NC + 1] = reinterpret_cast<float (*)[NC + 1]>(mfcc_lds + WPB * (2 * NC));
  float (*lm)[64] = reinterpret_cast<float (*)[64]>(mfcc_lds + WPB * (3 * NC + 1));
  // (the wave number through readfirstlane: the row, its utterance, frame, sample and noise addresses are then scalars -- s_load look-ups,
  // one address register per request instead of a 64-bit vector addition each; the front end was 350 of the kernel's ~1 000 vector
  // instructions per frame, and the kernel is bound by their number)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float4 *lrec = reinterpret_cast<float4 *>(mfcc_lds + WPB * (3 * NC + 1 + 64));      // TAB: [level][3][64] float4
  if (TAB) {
    const int n4 = m.fft_num_levels * 64 * 3;
    for (int i = threadIdx.x; i < n4; i += 64 * WPB) {
      const int rec = i / 3, c = i - 3 * rec;
      lrec[((rec >> 6) * 3 + c) * 64 + (rec & 63)] = m.fft_recs[i];
    }
    __syncthreads();
  }
  const int row = blockIdx.x * WPB + wave;
  const bool active = row < g.total_rows;
  // What does not depend on the frame is requested first, ahead of the row's own chain of look-ups (row -> utterance -> offsets ->
  // samples): the first FFT level's record here, the post-processing pass's gather indices and factors and the lane's mel filter and
  // lifter in front of the FFT, whose seven LDS round trips cover them.
  // The kernel is a chain of ~20 dependent trips to L2 per frame, and eight waves per SIMD -- all it can hold -- do not hide them
  // (its LDS pipe is half busy since the swizzle, its vector ALU 63 %: profiles/micro/pmc_lds.sh, pmc_inst.sh; round 6).
  constexpr bool kFast = NFFT == 512;
  float4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
  const bool recs_on = kFast && m.fft_recs != nullptr;
  if (recs_on && !TAB) {
    const char *lv = reinterpret_cast<const char *>(m.fft_recs) + (unsigned)lane * 48u;
    a0 = *reinterpret_cast<const float4 *>(lv); a1 = *reinterpret_cast<const float4 *>(lv + 16); a2 = *reinterpret_cast<const float4 *>(lv + 32);
  }

  int u = 0, t = 0;
  bool first = false, last = false;      // this row is the utterance's frame 0 / frame T - 1: its cepstra are also the left / right halo rows'
  if (active) {
    u = g.d_row_utt[row];
    t = g.d_row_t[row];
    const int T = g.d_num_frames[u];
    // A halo row repeats the edge frame next to it: the wave of that frame writes the copies, the halo rows' own waves have
    // nothing to do (9 % of the rows of a batch of 3 s utterances).  No workgroup-wide step anywhere below: a wave may leave.
    if (T > 0 && (t < 0 || t >= T)) return;
    first = T > 0 && t == 0;
    last = T > 0 && t == T - 1;
    t = t >= T ? T - 1 : t;
    t = t < 0 ? 0 : t;          // (the halo rows of an utterance too short for one frame: frame 0 of whatever follows it, never read back)
  }
#ifdef RS_MFCC_PROFILE
  long long mp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mt = clock64();
#define RS_MT(i) do { const long long n_ = clock64(); mp[i] += n_ - mt; mt = n_; } while (0)
#else
#define RS_MT(i) do { } while (0)
#endif
  float raw_energy = 0.f;
  float *xr = xrb[wave], *xi = xib[wave];
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
    float s0[JP], s1[JP], n0[JP], n1[JP], w0[JP], w1[JP];
    // A lane's two samples, noise values and window values are neighbours: with an even window and aligned rows (the usual case: a
    // wave-uniform test) each pair is ONE request -- 12 per frame instead of 24.  The kernel's texture addresser is busy 89 % of the
    // launch (profiles/micro/pmc_ta.sh): ~80 vector memory requests per frame at 16 cycles each are what a frame costs a CU.
    const bool pairs = (m.win & 1) == 0 && m.win >= 2 && ((reinterpret_cast<uintptr_t>(src) & 3) == 0) &&
                       ((reinterpret_cast<uintptr_t>(noise) & 7) == 0) && ((reinterpret_cast<uintptr_t>(m.window) & 7) == 0);
    if (pairs) {
#pragma unroll
      for (int j = 0; j < JP; j++) {
        const int pi = lane + RS_WAVE * j, pc = 2 * pi < m.win ? pi : (m.win >> 1) - 1;      // pair index, clamped into the window
        const unsigned b = (unsigned)pc;
        const int sp = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(src) + 4u * b);
        const float2 np = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(noise) + 8u * b);
        const float2 wp = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(m.window) + 8u * b);
        s0[j] = (float)(short)(sp & 0xffff); s1[j] = (float)(sp >> 16);
        n0[j] = np.x; n1[j] = np.y;
        w0[j] = wp.x; w1[j] = wp.y;
      }
    } else {
#pragma unroll
      for (int j = 0; j < JP; j++) {                      // all requests first ...
        const int i0 = 2 * (lane + RS_WAVE * j), c0 = i0 < m.win ? i0 : m.win - 1, c1 = i0 + 1 < m.win ? i0 + 1 : m.win - 1;
        // (scalar base + 32-bit lane offset: the addressing mode that costs no vector instruction per request)
        const unsigned b0 = (unsigned)c0, b1 = (unsigned)c1;
        s0[j] = (float)*reinterpret_cast<const int16_t *>(reinterpret_cast<const char *>(src) + 2u * b0);
        s1[j] = (float)*reinterpret_cast<const int16_t *>(reinterpret_cast<const char *>(src) + 2u * b1);
        n0[j] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(noise) + 4u * b0);
        n1[j] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(noise) + 4u * b1);
        w0[j] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(m.window) + 4u * b0);
        w1[j] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(m.window) + 4u * b1);
      }
    }
    float v0[JP], v1[JP];
#pragma unroll
    for (int j = 0; j < JP; j++) {                      // ... then Dither(): data[i] += RandGauss(&rstate) * dither_value (no FMA: -ffp-contract=off)
      const int i0 = 2 * (lane + RS_WAVE * j), i1 = i0 + 1;
      v0[j] = s0[j] + n0[j] * dv; v1[j] = s1[j] + n1[j] * dv;
      if (i1 < m.win) dsum += (double)(v0[j] + v1[j]);
      else if (i0 < m.win) dsum += (double)v0[j];
    }
    dsum = WaveSumF64(dsum);      // (exact in any order, see above: row rotations + four readlanes instead of six ds_bpermute pairs)
    // DC removal (x[i] += -mean), then
    // 2. pre-emphasis (uses the *un-emphasised* left neighbour, as the backwards loop of the reference does) and window; the even /
    // odd samples are the real / imaginary parts of the half-length complex transform.  A lane keeps its sample pairs in registers
    // from the load to here: the odd sample's left neighbour is the pair's even one, the even sample's is the previous lane's odd one
    // (one cross-lane move per pair).  Through an LDS copy of the frame -- written in pairs, read back sample by sample with the
    // window value fetched under the same condition -- this step was 200 of the kernel's 1 300 instructions per frame, and the
    // kernel is bound by their number (round 6).  Same subtractions and products on the same operands.
    const float dc = m.remove_dc ? (float)dsum / (float)m.win : 0.f;      // (x - 0.f is x)
    float e_raw = 0.f, e_win = 0.f;
#pragma unroll
    for (int j = 0; j < JP; j++) {
      const int i0 = 2 * (lane + RS_WAVE * j), i1 = i0 + 1;
      float left = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v1[j]), 0x138, 0xF, 0xF, false));      // wave_shr:1 -- lane l - 1's odd sample
      const float carry = j > 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v1[j > 0 ? j - 1 : 0]), RS_WAVE - 1)) : v0[0];      // lane 0: the previous round's last sample; sample 0 is its own neighbour
      left = lane == 0 ? carry : left;
      const float c0 = v0[j] - dc, c1 = v1[j] - dc, p0 = left - dc;
      const float y0 = i0 < m.win ? (c0 - m.preemph * p0) * w0[j] : 0.f;
      const float y1 = i1 < m.win ? (c1 - m.preemph * c0) * w1[j] : 0.f;
      if (m.use_energy) {      // (kernel-uniform)
        e_raw += (i0 < m.win ? c0 * c0 : 0.f) + (i1 < m.win ? c1 * c1 : 0.f);
        e_win += y0 * y0 + y1 * y1;
      }
      // (point i of the half-length transform lives at i ^ g(i >> 5), g linear in the three bits: MfccDev::fft_swz, zero without the
      // pre-addressed plan; i = lane + 64 j, so bit 5 is the lane's and bits 6, 7 are j's)
      const int pos = ((lane ^ ((lane & 32) ? m.fft_swz[0] : 0)) ^ ((j & 1) ? m.fft_swz[1] : 0) ^ ((j & 2) ? m.fft_swz[2] : 0)) + RS_WAVE * j;
      xr[pos] = y0;
      xi[pos] = y1;
    }
    if (m.use_energy) raw_energy = logf(fmaxf(WaveSum(m.raw_energy ? e_raw : e_win), FLT_EPSILON));
  }
  RS_MT(0);
  WaveLdsSync();
  RS_MT(1);
  // (requested here, where the registers of the sample pairs are free again; first used seven LDS round trips later)
  int pp_k[2] = {0, 0}, pp_d[2] = {0, 0};
  float pp_re[2] = {0.f, 0.f}, pp_im[2] = {0.f, 0.f};
  if (kFast) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int k = lane + 1 + RS_WAVE * q;          // (2 k <= NFFT / 2 for both)
      (void)k;
      const float4 pr = m.fft_post[q * RS_WAVE + lane];      // (one request instead of four: the kernel is bound by the NUMBER of its vector memory requests)
      pp_k[q] = __float_as_int(pr.x); pp_d[q] = __float_as_int(pr.y);
      pp_re[q] = pr.z; pp_im[q] = pr.w;
    }
  }
  const int mlane = lane < m.nbins ? lane : 0, clane = lane < m.nceps ? lane : 0;
  const int4 mrec = m.mel_rec[mlane];
  const int mel_off = mrec.x, mel_n = mrec.y, mel_s = mrec.z;
  const float lift = m.lifter[clane];
  // 3. split-radix complex FFT, level by level (tasks of one level touch disjoint points)
  if (NFFT == 512 && m.fft_recs) {
    // at most one task per lane and level: the record of the NEXT level (task + its twiddle factors, three 16-byte loads of one
    // 48-byte record) is requested before the current level runs, so no level waits for a global round trip -- the task-then-twiddles
    // pair of dependent loads per level was 14 of the frame's ~45 (profiles/r04/mfcc_notes.txt).  Round 6: 64 records per level (a
    // lane's record at a fixed offset from the level's base), the points' LDS offsets in the record, two register sets in turn.
    const int nl = m.fft_num_levels;
    const int xi_off = (int)(xi - xr);
    const char *rec0 = reinterpret_cast<const char *>(m.fft_recs);
    const unsigned lane_off = (unsigned)lane * 48u;
    float4 b0, b1, b2;
#define RS_FFT_FETCH(L, r0, r1, r2)                                                              \
    {                                                                                            \
      if (TAB) {                                                                                 \
        r0 = lrec[((L) * 3 + 0) * 64 + lane]; r1 = lrec[((L) * 3 + 1) * 64 + lane]; r2 = lrec[((L) * 3 + 2) * 64 + lane]; \
      } else {                                                                                   \
        const char *lv = rec0 + (size_t)(L) * (64 * 48);                                         \
        r0 = *reinterpret_cast<const float4 *>(lv + lane_off);                                   \
        r1 = *reinterpret_cast<const float4 *>(lv + lane_off + 16);                              \
        r2 = *reinterpret_cast<const float4 *>(lv + lane_off + 32);                              \
      }                                                                                          \
    }
    if (TAB) RS_FFT_FETCH(0, a0, a1, a2)
    for (int L = 0; L < nl; L += 2) {      // (level 0's record: requested at the top of the kernel)
      if (L + 1 < nl) RS_FFT_FETCH(L + 1, b0, b1, b2)
      if (active) SrfftRunRec(a0, a1, a2, xr, xi_off);
      WaveLdsSync();
      if (L + 1 >= nl) break;
      if (L + 2 < nl) RS_FFT_FETCH(L + 2, a0, a1, a2)
      if (active) SrfftRunRec(b0, b1, b2, xr, xi_off);
      WaveLdsSync();
    }
#undef RS_FFT_FETCH
  } else {
    for (int L = 0; L < m.fft_num_levels; L++) {
      if (active)
        for (int ti = m.fft_level_begin[L] + lane; ti < m.fft_level_begin[L + 1]; ti += RS_WAVE) SrfftRunTask(tasks[ti], fft_tw, xr, xi);
      WaveLdsSync();
    }
  }
  RS_MT(2);
  // 4. real-FFT post-processing (srfft.cc:379-417) fused with the power spectrum (feature-functions.cc:41-49);
  // spectrum element k of the bit-reversal pass is element perm[k] of the in-place result
  if (active) {
    auto post = [&](int k, int pk, int pd, float kn_re, float kn_im) __attribute__((always_inline)) {
      const int kd = NC - k;
      const float bk_re = xr[pk], bk_im = xi[pk], bd_re = xr[pd], bd_im = xi[pd];
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
    };
    if (kFast) {
#pragma unroll
      for (int q = 0; q < 2; q++) post(lane + 1 + RS_WAVE * q, pp_k[q], pp_d[q], pp_re[q], pp_im[q]);
    } else {
      for (int k = lane + 1; 2 * k <= NC; k += RS_WAVE) post(k, m.fft_perm[k], m.fft_perm[NC - k], m.fft_kn[2 * k], m.fft_kn[2 * k + 1]);
    }
    if (lane == 0) {
      const float d0 = xr[m.fft_perm[0]], d1 = xi[m.fft_perm[0]];
      const float zeroth = d0 + d1, n2th = d0 - d1;
      pw[wave][0] = zeroth * zeroth;
      pw[wave][NC] = n2th * n2th;
    }
  }
  WaveLdsSync();
  RS_MT(3);
  // 5. mel filterbank + log
  if (active && lane < m.nbins) {
    const int off = mel_off, len = mel_n;
    const float *w = m.mel_weights + mel_s;
    float e = 0.f;
    for (int i = 0; i < len; i++) e += w[i] * pw[wave][off + i];
    lm[wave][lane] = logf(fmaxf(e, FLT_EPSILON));
  }
  WaveLdsSync();
  RS_MT(4);
  // 6. DCT + lifter, write the row
  if (active && lane < m.nceps) {
    const float *d = m.dct + lane * m.nbins;
    float c = 0.f;
    for (int b = 0; b < m.nbins; b++) c += d[b] * lm[wave][b];
    c *= lift;
    if (m.use_energy && lane == 0) c = fmaxf(raw_energy, m.log_energy_floor);
    feats[(size_t)(out_rows ? out_rows[row] : row) * ld + lane] = c;      // out_rows: streams write into their pool rows
    if (first) for (int k = 1; k <= g.L; k++) feats[(size_t)(out_rows ? out_rows[row - k] : row - k) * ld + lane] = c;
    if (last) for (int k = 1; k <= g.R; k++) feats[(size_t)(out_rows ? out_rows[row + k] : row + k) * ld + lane] = c;
  }
  RS_MT(5);
#ifdef RS_MFCC_PROFILE
  if (lane == 0 && row % 9973 == 0)
    printf("mfcc row %d: load+dc %lld preemph+window %lld fft %lld power %lld mel %lld dct %lld\n", row, mp[0], mp[1], mp[2], mp[3], mp[4], mp[5]);
#endif
#undef RS_MT
}

template <int NFFT, int WPB, bool TAB = false>
static void LaunchMfccT(const MfccDev &m, const BatchGeom &g, const int16_t *pcm, float *feats, int ld, bool exclusive, const int *out_rows,
                        hipStream_t s) {
  const int blocks = (g.total_rows + WPB - 1) / WPB;
  if (!blocks) return;
  const size_t need = sizeof(float) * WPB * (3 * (NFFT / 2) + 1 + 64) + (TAB ? (size_t)m.fft_num_levels * 64 * 48 : 0);
  const size_t smem = exclusive ? std::max<size_t>(need, 159 * 1024) : need;
  static size_t attr = 0;
  if (smem > attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&MfccKernel<NFFT, WPB, TAB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = smem;
  }
  hipLaunchKernelGGL((MfccKernel<NFFT, WPB, TAB>), dim3(blocks), dim3(64 * WPB), smem, s, m, g, pcm, feats, ld, out_rows);
}

void LaunchMfcc(const MfccDev &m, const BatchGeom &g, const int16_t *pcm, float *feats, int ld, hipStream_t s, bool exclusive,
                const int *out_rows) {
  if (m.padded == 512) {
    // Launches of one to about twelve device fills of 16-frame workgroups (8 k to 96 k rows: 64 to 256 utterances of 3 s) take them,
    // with the FFT plan in LDS.  Alone the launch is 4 % slower than with four frames per workgroup (167 against 161 us for the
    // headline batch: coarser tail), but beside other calls' layer GEMMs -- four calls in flight -- the pipelined step is shorter:
    // 0.59 against 0.73 ms for 64 utterances, 1.14 / 1.16 for 128, 1.52-1.56 / 1.63-1.69 for 256 (four boxes); 1.51 / 1.44 for 192 is
    // the exception inside the range, and above it the 4-wave form wins (384: 2.69 / 2.63, 512: 3.56 / 3.36, the mixed workload's
    // 512-utterance calls 6.8 / 6.5-6.6): profiles/micro/mfcc_shape_{ab,utts,conc,mixed}.sh.  Most of the effect is the workgroup size
    // (the 16-wave form without the tables shows it too) -- how this kernel's workgroups share CUs with the GEMMs' -- the rest a third
    // fewer vector memory requests.  An empirical launcher rule, not a model.  Small launches (a stream round, one utterance) keep the
    // 4-wave form.
    static const int shape_env = [] { const char *e = TuneEnv("RS_MFCC_SHAPE"); return e ? std::atoi(e) : -1; }();
    const bool tab_ok = m.fft_recs != nullptr && m.fft_num_levels * 64 * 48 <= 32 * 1024;
    const int shape = shape_env >= 0 ? shape_env : ((g.total_rows >= 16 * 512 && g.total_rows <= 96 * 1024) ? 16 : 0);
    if (exclusive) LaunchMfccT<512, 16>(m, g, pcm, feats, ld, true, out_rows, s);
    else if (shape == 16 && tab_ok) LaunchMfccT<512, 16, true>(m, g, pcm, feats, ld, false, out_rows, s);
    else if (shape == 8 && tab_ok) LaunchMfccT<512, 8, true>(m, g, pcm, feats, ld, false, out_rows, s);
    else if (shape == 17) LaunchMfccT<512, 16>(m, g, pcm, feats, ld, false, out_rows, s);
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
constexpr int kCmvnTC = 32;      // frames per chunk (the feature dimension is at most 128: engine.cc)
//
// Streams (t_begin != null): utterance u resumes at frame t_begin[u] with the running sums and the window count parked in
// state[(D + 1) * state_slot[u]] by the launch that produced frame t_begin[u] - 1, and parks them again at frame T; the frames
// that leave the window are re-read from `in`, which holds the whole stream.
template <int kPer>      // elements of a chunk per thread, at most: kCmvnTC * D <= 256 * kPer
__global__ __launch_bounds__(256) void OnlineCmvnKernel(CmvnDev c, BatchGeom g, const float *__restrict__ in, float *__restrict__ out, int ld,
                                                        const int *__restrict__ t_begin, double *__restrict__ state, const int *__restrict__ state_slot) {
  // LDS by the feature dimension (40: 23 KB; as [32][128] arrays the kernel held 68 KB of a CU, i.e. the place of one of the two
  // layer-GEMM workgroups of the call in flight beside it), and the NEXT chunk's frames are in flight, in registers, while a chunk
  // is processed (every chunk used to start with a trip to memory: ten of them per 3 s utterance).
  extern __shared__ __attribute__((aligned(16))) unsigned char cmvn_smem[];
  const int u = blockIdx.x, tid = threadIdx.x, D = c.dim, W = c.cmn_window;
  double *ss = reinterpret_cast<double *>(cmvn_smem);                 // [TC][D] running sums after each frame
  double *gs = ss + kCmvnTC * D;                                       // [D] global stats (read per element in step 3 while the window is not full)
  double *nn = gs + D, *aa = nn + kCmvnTC;                             // [TC] frame count in the window; weight of the global stats
  float *xs = reinterpret_cast<float *>(aa + kCmvnTC);                // [TC][D] the chunk
  float *xp = xs + kCmvnTC * D;                                        // [TC][D] the frames leaving the window while the chunk enters
  float *al = xp + kCmvnTC * D;                                        // [TC] -1 / (smoothed count)
  float *edge = al + kCmvnTC;                                          // [2][D] normalised first / last frame (halo rows replicate them)
  if (tid < D) gs[tid] = c.global_stats[tid];
  const int T = g.d_num_frames[u];
  const size_t base = (size_t)g.d_row_base[u] + g.L;
  double sum = 0.0, count = 0.0;                  // threads < D
  const double gcount = c.global_stats[D];
  const int t_first = t_begin ? t_begin[u] : 0;
  double *park = t_begin ? state + (size_t)(D + 1) * state_slot[u] : nullptr;
  if (park && t_first > 0 && tid < D) { sum = park[tid]; count = park[D]; }
  float nx[kPer], np[kPer];
  const int i_first = tid / D, d_first = tid - i_first * D, i_step = 256 / D, d_step = 256 - i_step * D;      // element tid + 256 q = (frame, dimension), by steps
  auto fetch = [&](int t0) {                             // element idx = tid + 256 q of the chunk that starts at frame t0
    const int n = T - t0 < kCmvnTC ? T - t0 : kCmvnTC;
    int i = i_first, d = d_first;
#pragma unroll
    for (int q = 0; q < kPer; q++) {
      // (no load under a condition: an element past the chunk's end re-reads the chunk's first frame, a frame that has no
      // predecessor W frames back reads frame 0; neither value is used)
      const int ti = t0 + (i < n ? i : 0), tp = ti - W > 0 ? ti - W : 0;
      nx[q] = in[(base + ti) * ld + d];
      np[q] = in[(base + tp) * ld + d];
      i += i_step; d += d_step;
      if (d >= D) { d -= D; i++; }
    }
  };
#ifdef RS_CMVN_PROFILE
  long long cp[6] = {0, 0, 0, 0, 0, 0}, ct = clock64();
#define RS_CT(i) do { const long long n_ = clock64(); cp[i] += n_ - ct; ct = n_; } while (0)
#else
#define RS_CT(i) do { } while (0)
#endif
  if (t_first < T) fetch(t_first);
  RS_CT(0);
  for (int t0 = t_first; t0 < T; t0 += kCmvnTC) {
    const int n = T - t0 < kCmvnTC ? T - t0 : kCmvnTC;
#pragma unroll
    for (int q = 0; q < kPer; q++) {
      const int idx = tid + 256 * q;
      // (every request is waited for HERE, used or not: one that is consumed under a condition stays "possibly in flight" for the
      // compiler, and the next fetch() into the same register then waits vmcnt(0) -- for this chunk's output stores as well)
      __asm__ volatile("" : "+v"(nx[q]), "+v"(np[q]));
      if (idx < n * D) { xs[idx] = nx[q]; xp[idx] = np[q]; }      // (xp is read where a frame leaves the window, nowhere else)
    }
    __syncthreads();
    RS_CT(1);
    if (t0 + kCmvnTC < T) fetch(t0 + kCmvnTC);
    if (tid < D) {
      if (n == kCmvnTC && (t0 >= W || t0 + kCmvnTC <= W)) {
        // A whole chunk whose frames all push an old frame out of the window, or none does: the chunk's values of this dimension
        // first (independent LDS reads, all in flight together), then the chain of additions in registers, then the running sums
        // back.  Read, add and write frame by frame, every frame waited out an LDS round trip behind the previous frame's write
        // (215 cycles per frame, 60 % of the kernel; round 6).  Same additions in the same order.
        float xv[kCmvnTC], pv[kCmvnTC];
        double sv[kCmvnTC];
        const bool leaving = t0 >= W;
#pragma unroll
        for (int i = 0; i < kCmvnTC; i++) xv[i] = xs[i * D + tid];
        if (leaving) {
#pragma unroll
          for (int i = 0; i < kCmvnTC; i++) pv[i] = xp[i * D + tid];
#pragma unroll
          for (int i = 0; i < kCmvnTC; i++) { sum += (double)xv[i]; sum -= (double)pv[i]; sv[i] = sum; }      // (count + 1 - 1)
          count += 1.0; count -= 1.0;
          if (tid == 0) {
#pragma unroll
            for (int i = 0; i < kCmvnTC; i++) nn[i] = count;
          }
        } else {
#pragma unroll
          for (int i = 0; i < kCmvnTC; i++) { sum += (double)xv[i]; sv[i] = sum; }
          if (tid == 0) {
#pragma unroll
            for (int i = 0; i < kCmvnTC; i++) nn[i] = count + (double)(i + 1);
          }
          count += (double)kCmvnTC;
        }
#pragma unroll
        for (int i = 0; i < kCmvnTC; i++) ss[i * D + tid] = sv[i];
      } else {
#pragma unroll 4
        for (int i = 0; i < n; i++) {
          sum += (double)xs[i * D + tid];
          count += 1.0;
          if (t0 + i - W >= 0) { sum -= (double)xp[i * D + tid]; count -= 1.0; }
          ss[i * D + tid] = sum;
          if (tid == 0) nn[i] = count;
        }
      }
    }
    __syncthreads();
    RS_CT(2);
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
    RS_CT(3);
    {
      // (element tid + 256 q by steps, as in fetch(): no division by the runtime dimension; the LDS reads of all of a thread's
      // elements before the arithmetic)
      int i = i_first, d = d_first;
      double svq[kPer], aq[kPer], gq[kPer];
      float alq[kPer], xq[kPer];
      int iq[kPer], dq[kPer];
#pragma unroll
      for (int q = 0; q < kPer; q++) {
        iq[q] = i; dq[q] = d;
        const bool on = i < n;
        const int idx = on ? i * D + d : 0, ii = on ? i : 0;
        svq[q] = ss[idx]; aq[q] = aa[ii]; gq[q] = gs[d]; alq[q] = al[ii]; xq[q] = xs[idx];
        i += i_step; d += d_step;
        if (d >= D) { d -= D; i++; }
      }
#pragma unroll
      for (int q = 0; q < kPer; q++) {
        if (iq[q] < n) {
          double sv = svq[q];
          if (aq[q] > 0.0) sv += aq[q] * gq[q];
          const float offset = (float)((double)alq[q] * sv);
          const float yv = xq[q] + offset;
          out[(base + t0 + iq[q]) * ld + dq[q]] = yv;
          if (t0 + iq[q] == 0) edge[dq[q]] = yv;
          if (t0 + iq[q] == T - 1) edge[D + dq[q]] = yv;
        }
      }
    }
    __syncthreads();
    RS_CT(4);
  }
#ifdef RS_CMVN_PROFILE
  if (tid == 0 && u % 61 == 0) printf("cmvn utt %d (T=%d): first fetch %lld | stage %lld walk %lld scalars %lld apply %lld\n", u, T, cp[0], cp[1], cp[2], cp[3], cp[4]);
#endif
#undef RS_CT
  if (park && tid < D) { park[tid] = sum; if (tid == 0) park[D] = count; }
  if (T > 0 && !t_begin) {
    for (int idx = tid; idx < g.L * D; idx += 256) out[(base - g.L + idx / D) * ld + idx % D] = edge[idx % D];
    for (int idx = tid; idx < g.R * D; idx += 256) out[(base + T + idx / D) * ld + idx % D] = edge[D + idx % D];
  }
}

void LaunchOnlineCmvn(const CmvnDev &c, const BatchGeom &g, const float *in, float *out, int ld, hipStream_t s, const int *t_begin,
                      double *state, const int *state_slot) {
  if (g.n_utts == 0) return;
  const int D = c.dim;
  // (D <= 128: engine.cc refuses more cepstral coefficients when the model is loaded; kPer = 16 covers 32 frames x 128)
  const size_t smem = sizeof(double) * ((size_t)kCmvnTC * D + D + 2 * kCmvnTC) + sizeof(float) * ((size_t)2 * kCmvnTC * D + kCmvnTC + 2 * D);
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&OnlineCmvnKernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024); attr_set = true; }
  if (D <= 40) hipLaunchKernelGGL(OnlineCmvnKernel<5>, dim3(g.n_utts), dim3(256), smem, s, c, g, in, out, ld, t_begin, state, state_slot);
  else if (D <= 64) hipLaunchKernelGGL(OnlineCmvnKernel<8>, dim3(g.n_utts), dim3(256), smem, s, c, g, in, out, ld, t_begin, state, state_slot);
  else hipLaunchKernelGGL(OnlineCmvnKernel<16>, dim3(g.n_utts), dim3(256), smem, s, c, g, in, out, ld, t_begin, state, state_slot);
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

// A call's results (word counts, costs, counters, the first words of every transcript) stored into page-locked host memory by a
// kernel: four small device-to-host copies on the copy engine wait behind whatever the other calls in flight have queued there
// (their 25 MB sample uploads).
__global__ __launch_bounds__(64) void ResultsToHostKernel(const int *__restrict__ nw, const float *__restrict__ costs, const long long *__restrict__ ctr,
                                                          const int *__restrict__ words, int max_words, int inline_words, int h_stride, int *h_nw,
                                                          float *h_costs, long long *h_ctr, int *h_words) {
  const int u = blockIdx.x, t = threadIdx.x;
  if (t == 0) h_nw[u] = nw[u];
  if (t < 4) h_costs[(size_t)u * 4 + t] = costs[(size_t)u * 4 + t];
  if (t < 8) h_ctr[(size_t)u * 8 + t] = ctr[(size_t)u * 8 + t];
  for (int k = t; k < inline_words; k += 64) h_words[(size_t)u * h_stride + k] = words[(size_t)u * max_words + k];
}
void LaunchResultsToHost(const int *nw, const float *costs, const long long *ctr, const int *words, int max_words, int inline_words, int h_stride, int n,
                         int *h_nw, float *h_costs, long long *h_ctr, int *h_words, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(ResultsToHostKernel, dim3(n), dim3(64), 0, s, nw, costs, ctr, words, max_words, inline_words, h_stride, h_nw, h_costs, h_ctr, h_words);
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
// RowGeometryKernel + one FrameRowsKernel per list, as block ranges of one launch
__global__ __launch_bounds__(256) void BatchSetupKernel(BatchSetup b) {
  int blk = blockIdx.x;
  const int geo_blocks = (b.rows + 255) / 256;
  if (blk < geo_blocks) {
    const int r = blk * 256 + threadIdx.x;
    if (r >= b.rows) return;
    int lo = 0, hi = b.n_utts;          // largest u with row_base[u] <= r
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (b.row_base[mid] <= r) lo = mid; else hi = mid; }
    b.row_utt[r] = lo;
    b.row_t[r] = r - b.row_base[lo] - b.L;
    if (b.row_ivec) b.row_ivec[r] = b.ivrow_base[lo];
    return;
  }
  blk -= geo_blocks;
  for (int l = 0; l < b.n_lists; l++) {      // (block-uniform)
    const int nb = (b.lists[l].total + 255) / 256;
    if (blk < nb) {
      const BatchSetup::List &ls = b.lists[l];
      const int i = blk * 256 + threadIdx.x;
      if (i >= ls.total) return;
      int lo = 0, hi = ls.n_segs;          // largest segment with seg_off[seg] <= i
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ls.seg_off[mid] <= i) lo = mid; else hi = mid; }
      const int k = lo / b.n_utts, u = lo % b.n_utts;
      ls.out[i] = b.row_base[u] + ls.L + k * ls.slab_len + ls.first + ls.stride * (i - ls.seg_off[lo]);
      return;
    }
    blk -= nb;
  }
}
void LaunchBatchSetup(const BatchSetup &b, hipStream_t s) {
  int blocks = (b.rows + 255) / 256;
  for (int l = 0; l < b.n_lists; l++) blocks += (b.lists[l].total + 255) / 256;
  if (blocks <= 0) return;
  hipLaunchKernelGGL(BatchSetupKernel, dim3(blocks), dim3(256), 0, s, b);
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
