// Online iVector estimator kernels for gfx950: UBM posteriors, statistics, conjugate-gradient solve.
//
// Reference behaviour being reproduced (kaldi/src):
//   gmm/diag-gmm.cc:546-562 + hmm/posterior.cc:440-509 (UBM log-likes, posterior pruning)
//   ivector/ivector-extractor.cc:611-668,732-795 + matrix/optimization.cc:453-566 (stats, prior, CG solve)
//   online2/online-ivector-feature.cc:225-330,440-442 (what is accumulated when)
//
// Shape of the work (zamia-like: D = 40, G = 512, I = 100, 5 Gaussians kept per frame):
//   * UBM scoring is a [rows x 2D] x [2D x G] product followed by a per-row top-k: UbmPostMfmaKernel, a wave scores 16 rows
//     against all Gaussians on the FP32 matrix cores and selects with DPP row reductions (UbmPostKernel: the scalar-FMA form it
//     is bit-identical to, kept for shapes the MFMA form does not cover);
//   * the first-order statistics: IvecAccumKernel, a counting sort of each chunk's posteriors by Gaussian in LDS, then a wave per
//     Gaussian adds in frame order;
//   * two batch products over the utterances, in double as the reference keeps them:
//     linear[u] = sum_{g,d} Sigma^-1 M_g[d,:] * wfeats[u,g,d]   ([U x G D] x [G D x I])
//     quadratic[u] = sum_g gamma[u,g] * U_g                    ([U x G] x [G x I(I+1)/2])
//     on v_mfma_f64_16x16x4_f64, 64 utterances per workgroup (IvecLinearMfmaKernel, IvecQuadMfmaAsmKernel / IvecQuadMfmaKernel;
//     the vector forms IvecLinearPartialKernel / IvecQuadKernel remain as the cross-check RS_IVEC_MFMA=0 selects);
//   * the solve is one workgroup per utterance with the quadratic term expanded to a full matrix in LDS (IvecSolveFullKernel).
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include "env.h"

#include "kernels.h"
#include "wave_ops.h"

namespace rs {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------ UBM posteriors
// NPL = Gaussians per lane (G <= 64 * NPL), R = rows per wave.  Gaussian parameters are stored transposed (D x G)
// so that a wave reads 64 consecutive floats per dimension.  Selection (VectorToPosteriorEntry): candidates are
// the Gaussians whose log-likelihood exceeds max + log(min_post); the num_gselect best are kept (exp() is
// monotone, so ranking by log-likelihood is ranking by posterior; ties -> lowest Gaussian index), then pruned and
// renormalised exactly as posterior.cc:494-507 does.  exp() is evaluated in double for the kept ones only.
template <int NPL, int R>
__global__ __launch_bounds__(256) void UbmPostKernel(IvecDev iv, BatchGeom g, const float *__restrict__ feats, int ld,
                                                     int *__restrict__ post_idx, float *__restrict__ post_w) {
  constexpr int NP2 = (NPL + 1) / 2;
  static_assert(R <= 8, "phase B maps (row, slot) to lane = 8 row + slot");
  __shared__ float xs[4][R][128];
  __shared__ unsigned ckey[4][64 * 2 * NP2];      // per-wave candidate list
  __shared__ int cgi[4][64 * 2 * NP2];
  __shared__ float sel_ll[4][R][8], sel_max[4][R];
  __shared__ int sel_gi[4][R][8], sel_n[4][R];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + wave) * R;
  const int D = iv.feat_dim, G = iv.num_gauss, nsel = iv.num_gselect;
  unsigned active = 0;          // bit r: row r is a real frame (wave-uniform)
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int row = row0 + r;
    bool ok = row < g.total_rows;
    if (ok) {
      const int u = g.d_row_utt[row], t = g.d_row_t[row];
      ok = t >= 0 && t < g.d_num_frames[u];
    }
    active |= ok ? (1u << r) : 0u;
    for (int d = lane; d < D; d += 64) xs[wave][r][d] = ok ? feats[(size_t)row * ld + d] : 0.f;
  }
  __syncthreads();
  if (active != 0u) {
    // (scalar FMAs: no packed FP32 VALU arithmetic beside another call's MFMA kernels; this kernel is the
    // fallback for shapes the matrix-core kernel below does not cover)
    float a1[R][2 * NP2], a2[R][2 * NP2];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
      for (int j = 0; j < 2 * NP2; j++) { a1[r][j] = 0.f; a2[r][j] = 0.f; }
    // clamped Gaussian indices: lanes past G score the last Gaussian again and are ignored at selection
    int gidx[2 * NP2];
#pragma unroll
    for (int j = 0; j < 2 * NP2; j++) { const int gi = lane + j * 64; gidx[j] = gi < G ? gi : G - 1; }
    for (int d = 0; d < D; d++) {
      const float *mi = iv.means_invvars_t + (size_t)d * G, *vi = iv.inv_vars_t + (size_t)d * G;
      float m[2 * NP2], v[2 * NP2];
#pragma unroll
      for (int j = 0; j < 2 * NP2; j++) { m[j] = mi[gidx[j]]; v[j] = vi[gidx[j]]; }
#pragma unroll
      for (int r = 0; r < R; r++) {
        const float xv = xs[wave][r][d], xq = xv * xv;
#pragma unroll
        for (int j = 0; j < 2 * NP2; j++) {
          a1[r][j] = fmaf(xv, m[j], a1[r][j]);
          a2[r][j] = fmaf(xq, v[j], a2[r][j]);
        }
      }
    }
    float gc[2 * NP2];
#pragma unroll
    for (int j = 0; j < 2 * NP2; j++) gc[j] = iv.gconsts[gidx[j]];
    const float log_min_post = logf(iv.min_post);
    // ---- phase A, per row: candidates (like > max + log min_post) are compacted into a per-wave LDS list, the
    // num_gselect best are taken from the list (normally a handful of entries, one per lane)
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (!((active >> r) & 1u)) continue;         // wave-uniform
      unsigned key[2 * NP2];
      unsigned lmax = 0u;
#pragma unroll
      for (int j = 0; j < 2 * NP2; j++) {
        const float s1 = a1[r][j], s2 = a2[r][j];
        float v = gc[j] + s1;
        v = v + (-0.5f) * s2;
        key[j] = (lane + j * 64 < G && j < NPL) ? wv::FloatToOrdered(v) : 0u;   // lanes past G: below every real value
        lmax = max(lmax, key[j]);
      }
      const unsigned kmax = wv::MaxU(lmax);
      const float max_like = wv::OrderedToFloat(kmax);
      const unsigned kcut = wv::FloatToOrdered(max_like + log_min_post);
      int C = 0;
#pragma unroll
      for (int j = 0; j < 2 * NP2; j++) {
        const bool cand = key[j] > kcut;
        const unsigned long long mask = __ballot(cand);
        if (mask != 0ull) {                        // wave-uniform, rare per j
          if (cand) {
            const int pos = C + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            ckey[wave][pos] = key[j];
            cgi[wave][pos] = lane + j * 64;
          }
          C += __popcll(mask);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      int nfound = 0;
      if (C <= 64) {
        unsigned kv = lane < C ? ckey[wave][lane] : 0u;
        const int gv = lane < C ? cgi[wave][lane] : 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          if (k < nsel && nfound == k) {
            const unsigned wm = wv::MaxU(kv);
            if (wm != 0u) {
              const unsigned long long tie = __ballot(kv == wm);
              int gi;
              if (__popcll(tie) == 1) gi = __builtin_amdgcn_readlane(gv, __ffsll((long long)tie) - 1);
              else gi = (int)wv::MinU(kv == wm ? (unsigned)gv : 0x7fffffffu);     // ties -> lowest Gaussian index
              if (lane == 0) { sel_ll[wave][r][k] = wv::OrderedToFloat(wm); sel_gi[wave][r][k] = gi; }
              nfound = k + 1;
              if (gv == gi) kv = 0u;
            }
          }
        }
      } else {
        for (int k = 0; k < nsel && nfound == k; k++) {
          unsigned bv = 0u;
          int bg = 0x7fffffff, bi = -1;
          for (int i = lane; i < C; i += 64) {
            const unsigned kk = ckey[wave][i];
            const int gg = cgi[wave][i];
            if (kk > bv || (kk == bv && kk != 0u && gg < bg)) { bv = kk; bg = gg; bi = i; }
          }
          const unsigned wm = wv::MaxU(bv);
          if (wm == 0u) break;
          const int gi = (int)wv::MinU(bv == wm ? (unsigned)bg : 0x7fffffffu);
          if (bv == wm && bg == gi) ckey[wave][bi] = 0u;
          if (lane == 0) { sel_ll[wave][r][k] = wv::OrderedToFloat(wm); sel_gi[wave][r][k] = gi; }
          nfound = k + 1;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
      if (lane == 0) { sel_n[wave][r] = nfound; sel_max[wave][r] = max_like; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // ---- phase B, all rows of the wave at once: lane (r, k) evaluates exp(like - max) in double as the reference
    // does; lane r then prunes and renormalises row r (posterior.cc:494-507) and writes it
    {
      const int r = lane >> 3, k = lane & 7;
      const bool row_ok = r < R && ((active >> r) & 1u);
      const int nf = row_ok ? sel_n[wave][r] : 0;
      float post = 0.f;
      if (k < nf) post = (float)exp((double)(sel_ll[wave][r][k] - sel_max[wave][r]));
      if (r < R) sel_ll[wave][r][k] = post;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < R && ((active >> lane) & 1u)) {
        const int rr = lane;
        int nfound = sel_n[wave][rr];
        float sel_w[8];
#pragma unroll
        for (int q = 0; q < 8; q++) sel_w[q] = sel_ll[wave][rr][q];
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) if (q < nfound) tot += sel_w[q];
        const float cutoff = iv.min_post * tot;
#pragma unroll
        for (int q = 7; q >= 1; q--)
          if (nfound == q + 1 && sel_w[q] < cutoff) { tot -= sel_w[q]; nfound = q; }
        const float inv_tot = (float)(1.0 / (double)tot);
        const float scale = iv.posterior_scale * 1.0f;
#pragma unroll
        for (int q = 0; q < 8; q++) {
          if (q < nsel) {
            float w = 0.f;
            int gi = -1;
            if (q < nfound) { w = sel_w[q] * inv_tot; w *= scale; gi = sel_gi[wave][rr][q]; }
            post_idx[(size_t)(row0 + rr) * nsel + q] = gi;
            post_w[(size_t)(row0 + rr) * nsel + q] = w;
          }
        }
      }
    }
  }
  // rows that are not frames (halo)
#pragma unroll
  for (int r = 0; r < R; r++)
    if (!((active >> r) & 1u) && row0 + r < g.total_rows && lane < nsel) post_idx[(size_t)(row0 + r) * nsel + lane] = -1;
}

// ---- the same scoring on the matrix cores.  [rows x D] x [D x G] twice (x . mean/var and x^2 . 1/var) with
// v_mfma_f32_16x16x4_f32: FP32 in, FP32 out, and -- per MI355X_MICROARCH / the programming guide -- bit for bit a k-ordered
// fmaf chain, i.e. EXACTLY the sums the kernel above forms with v_pk_fma_f32 (same products, same order, one rounding each),
// at the matrix pipe's 64 FLOP/clk/SIMD from one wave per SIMD instead of 18 % of the vector peak.  No packed FP32 VALU
// arithmetic is left in the iVector path (the instructions that a co-resident MFMA kernel of another call corrupted in the
// feature kernel).
// A wave scores 16 rows against all G Gaussians: the 16 x 4 A operand of every k-step (x and x^2) stays in registers for the
// whole wave; the B operands (parameters in MFMA fragment order, prepared on the host: 16 bytes per lane = four k-steps) stream
// from L1/L2.  C layout of the 16x16 tile: lane l holds Gaussian 16 j + (l & 15) of rows 4 (l >> 4) + q, q = 0..3 -- so the
// 16 lanes of a DPP row hold ALL Gaussians of four rows, and the top-k selection of a row is a matter of row-wide DPP
// reductions, four rows at a time.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned RowMaxU(unsigned v) {
  v = max(v, wv::Dpp<0x121>(v)); v = max(v, wv::Dpp<0x122>(v)); v = max(v, wv::Dpp<0x124>(v)); v = max(v, wv::Dpp<0x128>(v));
  return v;
}
__device__ __forceinline__ unsigned RowMinU(unsigned v) {
  v = min(v, wv::Dpp<0x121>(v)); v = min(v, wv::Dpp<0x122>(v)); v = min(v, wv::Dpp<0x124>(v)); v = min(v, wv::Dpp<0x128>(v));
  return v;
}
constexpr int kUbmRows = 16;          // rows per wave
template <int NT, int KG, int KU = 4 * KG>   // Gaussian tiles of 16; groups of four k-steps (16 feature dims); k-steps that carry feature dims
// (two workgroups per CU = two waves per SIMD: 248 registers + 8 accumulators is the budget the tile loop below is written against)
__global__ __launch_bounds__(256, 2) void UbmPostMfmaKernel(IvecDev iv, BatchGeom g, const float *__restrict__ feats, int ld,
                                                         const float *__restrict__ bm, const float *__restrict__ bv,
                                                         int *__restrict__ post_idx, float *__restrict__ post_w, int ablate) {
  constexpr int KS = 4 * KG, KP = 16 * KG + 1;      // k-steps; LDS pitch (odd: the 16 rows of a read hit 16 banks)
  constexpr int CAP = 16 * NT;                        // candidates per row, worst case
  __shared__ float xs[4][kUbmRows][KP];
  __shared__ float gcs[16 * NT];                   // gconsts (read with ds_read: a global load here would make every tile wait vmcnt(0))
  __shared__ unsigned ckey[4][4][CAP];
  __shared__ unsigned short cgi[4][4][CAP];
  __shared__ float sel_ll[4][kUbmRows][8], sel_max[4][kUbmRows];
  __shared__ int sel_gi[4][kUbmRows][8], sel_n[4][kUbmRows];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, lg = lane & 15;
  const int row0 = (blockIdx.x * 4 + wave) * kUbmRows;
  const int D = iv.feat_dim, G = iv.num_gauss, nsel = iv.num_gselect;
#ifdef RS_UBM_PROFILE
  const long long tp0 = clock64();
  long long tp1 = tp0, tp2 = tp0, tp3 = tp0, tp4 = tp0;
  int pc_max = 0, pc_sum = 0, pc_slow = 0;
#endif
  unsigned active = 0;          // bit r: row r is a real frame (wave-uniform)
  for (int i = threadIdx.x; i < 16 * NT; i += 256) gcs[i] = iv.gconsts[i < G ? i : G - 1];
  {
    // which of the wave's rows are frames: lane r looks its row up (row -> utterance, frame index -> frame count: three dependent
    // loads, once, for all 16 rows side by side), then the 16 feature rows are requested together.  Row after row -- each row's
    // lookups and its feature load behind the previous row's -- the wave began with 64 dependent global round trips, about as long
    // as its 640 MFMAs (scoring alone took 115 us for the headline batch against the matrix pipe's 54).
    bool ok = false;
    if (lane < kUbmRows) {
      const int row = row0 + lane;
      ok = row < g.total_rows;
      if (ok) {
        const int u = g.d_row_utt[row], t = g.d_row_t[row];
        ok = t >= 0 && t < g.d_num_frames[u];
      }
    }
    active = (unsigned)(__ballot(ok) & 0xFFFFull);
    float xv[kUbmRows];
#pragma unroll
    for (int r = 0; r < kUbmRows; r++)
      xv[r] = (((active >> r) & 1u) && lane < D && lane < 16 * KG) ? feats[(size_t)(row0 + r) * ld + lane] : 0.f;
    if (lane < 16 * KG) {
#pragma unroll
      for (int r = 0; r < kUbmRows; r++) xs[wave][r][lane] = xv[r];
    }
  }
  __syncthreads();
#ifdef RS_UBM_PROFILE
  tp1 = clock64();
#endif
  if (active != 0u) {
    float a1[KS], a2[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) { a1[ks] = xs[wave][lg][4 * ks + grp]; a2[ks] = a1[ks] * a1[ks]; }
    unsigned key[NT][4];
    // The parameter fragments of tile j + 2 are requested before the MFMAs of tile j: with one or two waves per SIMD nothing else
    // hides the L2 round trip of a tile's six loads (193 us for the headline batch without this, all of it waiting).  The requests
    // are inline asm and the waits are placed by hand: written as C++ loads the compiler allocates a tile's temporaries in registers
    // of requests still in flight and puts s_waitcnt vmcnt(0) in front of them -- every tile then waited for the requests made a
    // moment earlier for two tiles ahead.  Each wait here leaves exactly the newer tiles' requests outstanding.
    constexpr int PF = NT > 2 ? 3 : 2;          // register sets in rotation
    f32x4v pm[PF][KG], pv[PF][KG];
    const unsigned lane_off = (unsigned)lane * 16u;
#define RS_UBM_LOAD(dst, base, off) __asm__ volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory")
    auto fetch = [&](int j, int set) __attribute__((always_inline)) {
#pragma unroll
      for (int kg = 0; kg < KG; kg++) {
        const unsigned off = lane_off + (unsigned)(j * KG + kg) * 1024u;
        RS_UBM_LOAD(pm[set][kg], bm, off);
        RS_UBM_LOAD(pv[set][kg], bv, off);
      }
    };
    // (the wait names the set's registers as in-out operands: nothing that reads them can be scheduled above it)
    auto wait_set = [&](int newer, int set) __attribute__((always_inline)) {
#pragma unroll
      for (int kg = 0; kg < KG; kg++) {
        if (newer >= 2) __asm__ volatile("s_waitcnt vmcnt(%2)" : "+v"(pm[set][kg]), "+v"(pv[set][kg]) : "n"(4 * KG) : "memory");
        else if (newer == 1) __asm__ volatile("s_waitcnt vmcnt(%2)" : "+v"(pm[set][kg]), "+v"(pv[set][kg]) : "n"(2 * KG) : "memory");
        else __asm__ volatile("s_waitcnt vmcnt(0)" : "+v"(pm[set][kg]), "+v"(pv[set][kg]) : : "memory");
      }
    };
#pragma unroll
    for (int j = 0; j < PF - 1 && j < NT; j++) fetch(j, j);
    // The score arithmetic of tile j - 1 (four values per lane) is placed between the MFMA pairs of tile j, whose accumulators are a
    // second register set: behind the tile's last MFMA it waited for the pipe to drain and left it idle for ~220 of a tile's 860 cycles.
    f32x4v c1s[2], c2s[2];
    auto score = [&](int j, int q, float gc) __attribute__((always_inline)) {
      const int gi = j * 16 + lg;
      float v = gc + c1s[j & 1][q];
      v = v + (-0.5f) * c2s[j & 1][q];
      const unsigned ko = wv::FloatToOrdered(v);
      key[j][q] = gi < G ? ko : 0u;            // columns past G: below every real value
    };
#pragma unroll
    for (int j = 0; j < NT; j++) {
      if (j + PF - 1 < NT) fetch(j + PF - 1, (j + PF - 1) % PF);
      wait_set(NT - 1 - j < PF - 1 ? NT - 1 - j : PF - 1, j % PF);
      f32x4v c1 = {0.f, 0.f, 0.f, 0.f}, c2 = {0.f, 0.f, 0.f, 0.f};
      const float gcp = gcs[(j > 0 ? j - 1 : 0) * 16 + lg];
#pragma unroll
      for (int kg = 0; kg < KG; kg++) {
        const f32x4v m = pm[j % PF][kg], v = pv[j % PF][kg];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if (4 * kg + i >= KU) continue;          // all-padding k-steps (D = 40: two of twelve) add exact zeros; skipped
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[4 * kg + i], m[i], c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[4 * kg + i], v[i], c2, 0, 0, 0);
          // (the two chains alternate: an MFMA of this shape issues in 32 cycles and its accumulator is ready after 40 -- left to
          // the scheduler, runs of one chain follow each other)
          if (j > 0 && 4 * kg + i < 4) score(j - 1, 4 * kg + i, gcp);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      c1s[j & 1] = c1; c2s[j & 1] = c2;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) score(NT - 1, q, gcs[(NT - 1) * 16 + lg]);
#undef RS_UBM_LOAD
#ifdef RS_UBM_PROFILE
    tp2 = clock64();
#endif
    if (ablate & 1) {          // measurement only (RS_UBM_ABLATE): scoring without the selection
      unsigned acc = 0;
#pragma unroll
      for (int j = 0; j < NT; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc ^= key[j][q];
      if (acc == 0x12345u) post_idx[0] = 1;
      return;
    }
    const float log_min_post = logf(iv.min_post);
    // ---- phase A: row 4 grp + q of every lane group at once.  Candidates (like > max + log min_post) are compacted into the
    // group's LDS list, the num_gselect best are taken from it (ties -> lowest Gaussian index).
    // (Round 6: this phase was as long as the scoring -- 19 k cycles of a wave alone on its SIMD, twice that beside a second one.
    // The tests against the cutoff are all made first, their lane masks kept in scalar registers, so that no test waits for the
    // branch on the previous one; list positions come from v_mbcnt on the group's share of a mask; and the best num_gselect of a list
    // of at most 16 are found by RANK -- every lane holds one candidate and counts, over 15 row rotations that depend on nothing
    // but the candidate itself, how many of the others are better -- instead of num_gselect rounds of row-wide max / min
    // reductions with an LDS round trip each.)
    const unsigned grp_lo = grp < 2 ? 0xFFFFu << (16 * grp) : 0u, grp_hi = grp >= 2 ? 0xFFFFu << (16 * (grp - 2)) : 0u;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (!((active >> q) & 0x1111u)) continue;                      // none of the four rows is a frame (wave-uniform)
      const int r = 4 * grp + q;
      unsigned lmax = 0u;
#pragma unroll
      for (int j = 0; j < NT; j++) lmax = max(lmax, key[j][q]);
      const unsigned kmax = RowMaxU(lmax);
      const float max_like = wv::OrderedToFloat(kmax);
      unsigned kcut = wv::FloatToOrdered(max_like + log_min_post);
      if (nsel <= 16) {
        // Only the num_gselect best of the candidates are kept, and the num_gselect-th largest of the 16 lanes' own maxima is a lower
        // bound of the num_gselect-th largest score of the row (they are 16 different scores of the row): nothing below it can be
        // selected.  Frames on which dozens of Gaussians lie within log min_post of the best -- a third of the headline batch's --
        // made lists of 20-45 candidates per row; with this bound the list is rarely longer than the selection.
        int above = 0;
#define RS_UBM_ABOVE(S) above += (int)(wv::Dpp<0x120 + S>(lmax) > lmax);
        RS_UBM_ABOVE(1) RS_UBM_ABOVE(2) RS_UBM_ABOVE(3) RS_UBM_ABOVE(4) RS_UBM_ABOVE(5) RS_UBM_ABOVE(6) RS_UBM_ABOVE(7) RS_UBM_ABOVE(8)
        RS_UBM_ABOVE(9) RS_UBM_ABOVE(10) RS_UBM_ABOVE(11) RS_UBM_ABOVE(12) RS_UBM_ABOVE(13) RS_UBM_ABOVE(14) RS_UBM_ABOVE(15)
#undef RS_UBM_ABOVE
        const unsigned nth = RowMinU(above < nsel ? lmax : 0xFFFFFFFFu);      // (lanes that tie share a count: at least num_gselect lanes qualify)
        kcut = max(kcut, nth ? nth - 1u : 0u);                               // candidates are the scores > kcut: nth itself stays one (0: fewer real columns than num_gselect)
      }
      unsigned long long cm[NT];
#pragma unroll
      for (int j = 0; j < NT; j++) cm[j] = __ballot(key[j][q] > kcut);
      __builtin_amdgcn_sched_barrier(0);
      int C = 0;                                                     // group-uniform
#pragma unroll
      for (int j = 0; j < NT; j++) {
        if (cm[j] != 0ull) {                                         // wave-uniform; about 40 % of the tiles hold a candidate of one of the four rows
          const unsigned ml = (unsigned)cm[j] & grp_lo, mh = (unsigned)(cm[j] >> 32) & grp_hi;
          if (key[j][q] > kcut) {
            const int pos = C + (int)__builtin_amdgcn_mbcnt_hi(mh, __builtin_amdgcn_mbcnt_lo(ml, 0u));
            ckey[wave][grp][pos] = key[j][q];
            cgi[wave][grp][pos] = (unsigned short)(j * 16 + lg);
          }
          C += __popc(ml) + __popc(mh);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#ifdef RS_UBM_PROFILE
      { const int cm_ = (int)wv::MaxU((unsigned)C); pc_max = cm_ > pc_max ? cm_ : pc_max; pc_sum += wv::Sum(lg == 0 ? C : 0); pc_slow += __ballot(C > 16) != 0ull; }
#endif
      if (__ballot(C > 16) == 0ull) {
        // one candidate per lane; sort key: likelihood, then the LOWER Gaussian index (distinct keys: no two lanes share a rank)
        const bool mine = lg < C;
        const unsigned k_hi = mine ? ckey[wave][grp][lg] : 0u;
        const unsigned gidx = mine ? (unsigned)cgi[wave][grp][lg] : 0u;
        const unsigned k_lo = mine ? 0xFFFFu - gidx : 0u;
        const unsigned long long mk = ((unsigned long long)k_hi << 32) | k_lo;
        int rank = 0;
#define RS_UBM_RANK(S)                                                                                          \
        {                                                                                                        \
          const unsigned long long ok = ((unsigned long long)wv::Dpp<0x120 + S>(k_hi) << 32) | wv::Dpp<0x120 + S>(k_lo); \
          rank += (int)(ok > mk);                                                                                \
        }
        RS_UBM_RANK(1) RS_UBM_RANK(2) RS_UBM_RANK(3) RS_UBM_RANK(4) RS_UBM_RANK(5) RS_UBM_RANK(6) RS_UBM_RANK(7) RS_UBM_RANK(8)
        RS_UBM_RANK(9) RS_UBM_RANK(10) RS_UBM_RANK(11) RS_UBM_RANK(12) RS_UBM_RANK(13) RS_UBM_RANK(14) RS_UBM_RANK(15)
#undef RS_UBM_RANK
        if (mine && rank < nsel) { sel_ll[wave][r][rank] = wv::OrderedToFloat(k_hi); sel_gi[wave][r][rank] = (int)gidx; }
        if (lg == 0) { sel_n[wave][r] = C < nsel ? C : nsel; sel_max[wave][r] = max_like; }
      } else {
        // a crowded list (more than 16 Gaussians within log min_post of the best): num_gselect rounds of row-wide selection
        int nfound = 0;
        const int cmax = (int)wv::MaxU((unsigned)C);                   // longest list of the four groups
        for (int k = 0; k < nsel; k++) {
          // best remaining candidate of my group: every lane scans its share of the list, then a row-wide reduction
          unsigned bv = 0u;
          int bg = 0x7fffffff, bi = -1;
          for (int i = lg; i < cmax; i += 16) {
            if (i < C) {
              const unsigned kk = ckey[wave][grp][i];
              const int gg = cgi[wave][grp][i];
              if (kk > bv || (kk == bv && kk != 0u && gg < bg)) { bv = kk; bg = gg; bi = i; }
            }
          }
          const unsigned wm = RowMaxU(bv);
          const bool have = wm != 0u && nfound == k;
          const int gi = (int)RowMinU((bv == wm && have) ? (unsigned)bg : 0x7fffffffu);
          if (have && bv == wm && bg == gi) ckey[wave][grp][bi] = 0u;
          if (have && lg == 0) { sel_ll[wave][r][k] = wv::OrderedToFloat(wm); sel_gi[wave][r][k] = gi; }
          if (have) nfound = k + 1;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        if (lg == 0) { sel_n[wave][r] = nfound; sel_max[wave][r] = max_like; }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
#ifdef RS_UBM_PROFILE
    tp3 = clock64();
#endif
    // ---- phase B: exp(like - max) in double as the reference does (lane = 4 row + slot pair), then lane r prunes and
    // renormalises row r (posterior.cc:494-507) and writes it
    {
      const int r = lane >> 2;
      const bool row_ok = (active >> r) & 1u;
      const int nf = row_ok ? sel_n[wave][r] : 0;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int k = (lane & 3) + 4 * h;
        float post = 0.f;
        if (k < nf) post = (float)exp((double)(sel_ll[wave][r][k] - sel_max[wave][r]));
        sel_ll[wave][r][k] = post;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < kUbmRows && ((active >> lane) & 1u)) {
        const int rr = lane;
        int nfound = sel_n[wave][rr];
        float sel_w[8];
#pragma unroll
        for (int q = 0; q < 8; q++) sel_w[q] = sel_ll[wave][rr][q];
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) if (q < nfound) tot += sel_w[q];
        const float cutoff = iv.min_post * tot;
#pragma unroll
        for (int q = 7; q >= 1; q--)
          if (nfound == q + 1 && sel_w[q] < cutoff) { tot -= sel_w[q]; nfound = q; }
        const float inv_tot = (float)(1.0 / (double)tot);
        const float scale = iv.posterior_scale * 1.0f;
#pragma unroll
        for (int q = 0; q < 8; q++) {
          if (q < nsel) {
            float w = 0.f;
            int gi = -1;
            if (q < nfound) { w = sel_w[q] * inv_tot; w *= scale; gi = sel_gi[wave][rr][q]; }
            post_idx[(size_t)(row0 + rr) * nsel + q] = gi;
            post_w[(size_t)(row0 + rr) * nsel + q] = w;
          }
        }
      }
    }
  }
#ifdef RS_UBM_PROFILE
  tp4 = clock64();
  if (threadIdx.x == 0 && blockIdx.x % 149 == 0)
    printf("ubm wg %d: prologue %lld tiles %lld select-A %lld select-B %lld; candidates: longest list %d, %d in 16 rows, %d of 4 passes on the slow path\n", (int)blockIdx.x, tp1 - tp0, tp2 - tp1, tp3 - tp2, tp4 - tp3, pc_max, pc_sum, pc_slow);
#endif
  // rows that are not frames (halo)
  if (lane < kUbmRows && !((active >> lane) & 1u) && row0 + lane < g.total_rows)
    for (int q = 0; q < nsel; q++) post_idx[(size_t)(row0 + lane) * nsel + q] = -1;
}

void LaunchUbmPosteriors(const IvecDev &iv, const BatchGeom &g, const float *lda_norm, int ld, int *post_idx,
                         float *post_w, hipStream_t s) {
  if (g.total_rows <= 0) return;
  const char *ue = std::getenv("RS_UBM_MFMA");          // read per call: the parity test flips it between two decodes
  const int use_mfma = ue ? std::atoi(ue) : 1;
  if (use_mfma && iv.ubm_bm && iv.num_gauss <= 512 && iv.feat_dim <= 48 && iv.num_gselect <= 8) {
    const int nt = (iv.num_gauss + 15) / 16, kg = iv.ubm_kg;
    const dim3 grid((g.total_rows + 4 * kUbmRows - 1) / (4 * kUbmRows));
    static const int ablate = [] { const char *e = TuneEnv("RS_UBM_ABLATE"); return e ? std::atoi(e) : 0; }();
#define RS_UBM_M(N, K) hipLaunchKernelGGL((UbmPostMfmaKernel<N, K>), grid, dim3(256), 0, s, iv, g, lda_norm, ld, iv.ubm_bm, iv.ubm_bv, post_idx, post_w, ablate)
    if (kg == 1) { if (nt <= 2) RS_UBM_M(2, 1); else if (nt <= 8) RS_UBM_M(8, 1); else RS_UBM_M(32, 1); }
    else if (nt > 8 && iv.feat_dim <= 40)
      hipLaunchKernelGGL((UbmPostMfmaKernel<32, 3, 10>), grid, dim3(256), 0, s, iv, g, lda_norm, ld, iv.ubm_bm, iv.ubm_bv, post_idx, post_w, ablate);
    else { if (nt <= 2) RS_UBM_M(2, 3); else if (nt <= 8) RS_UBM_M(8, 3); else RS_UBM_M(32, 3); }
#undef RS_UBM_M
    return;
  }
  const int npl = (iv.num_gauss + 63) / 64;
#define RS_UBM(N, RR)                                                                                                  \
  hipLaunchKernelGGL((UbmPostKernel<N, RR>), dim3((g.total_rows + 4 * RR - 1) / (4 * RR)), dim3(256), 0, s, iv, g, lda_norm, ld, \
                     post_idx, post_w)
  if (npl <= 2) RS_UBM(2, 8);
  else if (npl <= 4) RS_UBM(4, 8);
  else if (npl <= 8) RS_UBM(8, 8);
  else if (npl <= 16) RS_UBM(16, 4);
  else RS_UBM(32, 2);
#undef RS_UBM
}

// ------------------------------------------------------------------------------------------ iVector stats
// OnlineIvectorEstimationStats ctor (ivector-extractor.cc:786-795): quadratic = I, linear = [prior_offset, 0, ...];
// current_ivector_ starts at [prior_offset, 0, ...] (online-ivector-feature.cc:440-442); num_frames = 0.
__global__ void IvecInitKernel(IvecDev iv, double *__restrict__ linear, double *__restrict__ quadratic, double *__restrict__ x,
                               double *__restrict__ num_frames) {
  const int u = blockIdx.x, I = iv.ivec_dim, usz = I * (I + 1) / 2;
  for (int k = threadIdx.x; k < usz; k += blockDim.x) quadratic[(size_t)u * usz + k] = 0.0;
  for (int i = threadIdx.x; i < I; i += blockDim.x) {
    linear[(size_t)u * I + i] = i == 0 ? iv.prior_offset : 0.0;
    x[(size_t)u * I + i] = i == 0 ? iv.prior_offset : 0.0;
  }
  if (threadIdx.x == 0) num_frames[u] = 0.0;
  __syncthreads();
  for (int r = threadIdx.x; r < I; r += blockDim.x) quadratic[(size_t)u * usz + (size_t)r * (r + 1) / 2 + r] = 1.0;
}
void LaunchIvecInit(const IvecDev &iv, int n_utts, double *linear, double *quadratic, double *x, double *num_frames, hipStream_t s) {
  if (n_utts == 0) return;
  hipLaunchKernelGGL(IvecInitKernel, dim3(n_utts), dim3(256), 0, s, iv, linear, quadratic, x, num_frames);
}

// Per-Gaussian first-order statistics (AccStats: weighted_feats.AddVec per frame in double, float tot_weight), every sum
// accumulated in the reference's frame order.  Workgroup per utterance, frames in chunks:
//   1. the chunk's posteriors (<= num_gselect per frame) and feature rows are staged in LDS;
//   2. the entries are counting-sorted by Gaussian, stably in frame order: 16 waves histogram 16 frame segments, a
//      per-Gaussian prefix over the segments gives every segment its write cursor, each wave then walks its segment
//      frame by frame (the Gaussians of one frame are distinct, so a frame's entries can be placed in parallel);
//   3. the (Gaussian, dim) sums are independent: each wave takes Gaussians (lane = dim), walks that Gaussian's entry list
//      in order and does its read-modify-write of wfeats / gamma exactly once -- instead of one global round trip per frame.
#ifndef RS_ACC_TC
#define RS_ACC_TC 320
#endif
#ifndef RS_ACC_ABLATE                // measurement only (profiles/micro/acc_ablate.sh): 1 = no sums, 2 = staging only, 4 = sums without the global read-modify-write
#define RS_ACC_ABLATE 0
#endif
constexpr int kAccTC = RS_ACC_TC;    // frames per chunk
constexpr int kAccNSeg = 16;         // frame segments sorted in parallel (<= waves per workgroup); fewer when G is large
__global__ __launch_bounds__(1024) void IvecAccumKernel(IvecDev iv, BatchGeom g, const float *__restrict__ lda, int ld,
                                                         const int *__restrict__ post_idx, const float *__restrict__ post_w,
                                                         const int *frame_begin, const int *frame_end,
                                                         float *__restrict__ gamma, double *__restrict__ wfeats, int nseg, int fresh, int geo_mod) {
  extern __shared__ __attribute__((aligned(16))) char acc_smem[];
  const int u = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int ug = geo_mod > 0 ? u % geo_mod : u;      // (several pseudo-utterances -- the chunks of a stream -- over one utterance's rows)
  const int T = g.d_num_frames[ug];
  int t_begin = frame_begin ? frame_begin[u] : 0, t_end = frame_end ? frame_end[u] : T;
  if (t_end > T) t_end = T;
  const int D = iv.feat_dim, G = iv.num_gauss, nsel = iv.num_gselect;
  const size_t base = (size_t)g.d_row_base[ug] + g.L;
  float *gm = gamma + (size_t)u * G;
  double *wf = wfeats + (size_t)u * G * D;
  // LDS carve-up
  float *xs = reinterpret_cast<float *>(acc_smem);                    // [kAccTC][D]
  int *eidx = reinterpret_cast<int *>(xs + (size_t)kAccTC * D);       // [kAccTC * nsel] Gaussian of each entry (-1 = none)
  float *ew = reinterpret_cast<float *>(eidx + kAccTC * nsel);        // [kAccTC * nsel] posterior
  int *hist = reinterpret_cast<int *>(ew + kAccTC * nsel);            // [kAccNSeg][G]  counts, then write cursors
  int *cnt = hist + nseg * G;                                         // [G]
  int *off = cnt + G;                                                 // [G + 1]
  float *ows = reinterpret_cast<float *>(off + G + 1);                // [kAccTC * nsel] posteriors sorted by Gaussian (stable in frame order)
  unsigned short *ofr = reinterpret_cast<unsigned short *>(ows + kAccTC * nsel);   // [kAccTC * nsel] ... and their frames (within the chunk)
  unsigned short *ogi = ofr + kAccTC * nsel;                                        // [kAccTC * nsel] ... and their Gaussians
  int *gsplit = reinterpret_cast<int *>(ogi + kAccTC * nsel);                        // [17] first Gaussian of each wave's share of the sorted list
  // fresh: the statistics start from zero with this launch (whole utterances at once), so the first chunk writes every sum
  // instead of the caller clearing 8 G D bytes per utterance (42 MB for the headline batch) for this kernel to read back
  if (fresh && t_begin >= t_end) {
    for (int i = tid; i < G * D; i += 1024) wf[i] = 0.0;
    for (int i = tid; i < G; i += 1024) gm[i] = 0.f;
  }
#ifdef RS_ACC_PROFILE
  long long ap[6] = {0, 0, 0, 0, 0, 0}, at = clock64();
#define RS_AT(i) do { const long long n_ = clock64(); ap[i] += n_ - at; at = n_; } while (0)
#else
#define RS_AT(i) do { } while (0)
#endif
  for (int t0 = t_begin; t0 < t_end; t0 += kAccTC) {
    const int n = t_end - t0 < kAccTC ? t_end - t0 : kAccTC, ne = n * nsel;
    const int seglen = (n + nseg - 1) / nseg;
    __syncthreads();
    if (D <= 64) {
      // (a wave per frame, lane = dimension: no division by the runtime dimension, four rows in flight per wave)
      int f = wave;
      for (; f + 48 < n; f += 64) {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = lda[(base + t0 + f + 16 * q) * ld + (lane < D ? lane : 0)];
#pragma unroll
        for (int q = 0; q < 4; q++) if (lane < D) xs[(f + 16 * q) * D + lane] = v[q];
      }
      for (; f < n; f += 16) if (lane < D) xs[f * D + lane] = lda[(base + t0 + f) * ld + lane];
    } else {
      for (int i = tid; i < n * D; i += 1024) xs[i] = lda[(base + t0 + i / D) * ld + i % D];
    }
    for (int e = tid; e < ne; e += 1024) { eidx[e] = post_idx[(base + t0) * nsel + e]; ew[e] = post_w[(base + t0) * nsel + e]; }
    for (int i = tid; i < nseg * G; i += 1024) hist[i] = 0;
    __syncthreads();
    RS_AT(0);
    if (RS_ACC_ABLATE & 2) continue;
    // 2a. per-segment histograms
    if (wave < nseg) {
      const int f0 = wave * seglen, f1 = f0 + seglen < n ? f0 + seglen : n;
      for (int e = f0 * nsel + lane; e < f1 * nsel; e += 64) { const int gi = eidx[e]; if (gi >= 0) atomicAdd(&hist[wave * G + gi], 1); }
    }
    __syncthreads();
    RS_AT(1);
    // 2b. per Gaussian: exclusive prefix over the segments; then an exclusive scan over the Gaussians
    for (int gi = tid; gi < G; gi += 1024) {
      int run = 0;
      for (int sgm = 0; sgm < nseg; sgm++) { const int c = hist[sgm * G + gi]; hist[sgm * G + gi] = run; run += c; }
      cnt[gi] = run;
    }
    __syncthreads();
    if (wave == 0) {
      int carry = 0;
      for (int g0 = 0; g0 < G; g0 += 64) {
        const int gi = g0 + lane, c = gi < G ? cnt[gi] : 0;
        int inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
        if (gi < G) off[gi] = carry + inc - c;
        carry += __shfl(inc, 63, 64);
      }
      if (lane == 0) off[G] = carry;
    }
    __syncthreads();
    RS_AT(2);
    // 2c. stable placement: a wave walks its segment frame by frame
    if (wave < nseg) {
      const int f0 = wave * seglen, f1 = f0 + seglen < n ? f0 + seglen : n;
      // (four frames' entries and their Gaussians' list starts are read together; the cursors are then advanced frame by frame --
      // two frames of a segment can name the same Gaussian)
      for (int fb = f0; fb < f1; fb += 4) {
        int gq[4], oq[4];
        float wq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int f = fb + q < f1 ? fb + q : f1 - 1, e = f * nsel + (lane < nsel ? lane : 0);
          gq[q] = eidx[e]; wq[q] = ew[e];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) oq[q] = off[gq[q] >= 0 ? gq[q] : 0];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int f = fb + q, gi = gq[q];
          if (f < f1 && lane < nsel && gi >= 0) {
            const int pos = oq[q] + hist[wave * G + gi];
            hist[wave * G + gi]++;
            ofr[pos] = (unsigned short)f; ows[pos] = wq[q]; ogi[pos] = (unsigned short)gi;
          }
        }
      }
    }
    __syncthreads();
    RS_AT(3);
    // 3. independent (Gaussian, dim) sums, each in frame order.  A wave takes whole Gaussians (lane = feature dim), so
    // the walk over a Gaussian's entry list is wave-uniform: the list (frame, posterior: LDS broadcasts) is read four
    // entries ahead of the dependent double adds, and the read-modify-write of wfeats is batched so that PG global loads
    // are in flight at once.  (The earlier thread-per-(Gaussian, dim) form spent 117 of its 141 us here: lanes of a wave
    // straddled Gaussians with different list lengths and every entry cost two dependent LDS round trips plus an integer
    // division -- profiles/r02/acc_ablate.txt.)
    constexpr int PG = 16;
    if (RS_ACC_ABLATE & 1) continue;
    const bool first = fresh && t0 == t_begin;
    if (first && G <= 65535) {
      // The statistics start from zero with this chunk (every call but the second and later chunks of a long utterance): the sorted
      // list is dealt to the waves in contiguous pieces of about equal length that begin at a Gaussian's first entry, and a wave
      // walks its piece ENTRY BY ENTRY -- posterior, frame and Gaussian of the next eight entries are read together, the feature
      // values they name after that, whichever Gaussians they belong to; a change of Gaussian writes the finished sums.  Gaussian
      // by Gaussian (the loop below) a wave paid two dependent LDS round trips per entry -- the lists are three entries long on
      // average, too short for its read-ahead -- and the wave that drew the popular Gaussians held the workgroup: 32-81 k cycles of
      // the kernel's 110 k (round 6).  Same additions in the same order.
      const int E = off[G];
      for (int i = tid; i <= 16; i += 1024) gsplit[i] = G;
      __syncthreads();
      for (int gi = tid; gi < G; gi += 1024) {
        // owner of a Gaussian: the sixteenth of the list its first entry lies in (monotone in gi)
        const int o = E > 0 ? min(15, (int)((long)off[gi] * 16 / E)) : 0, op = gi > 0 ? (E > 0 ? min(15, (int)((long)off[gi - 1] * 16 / E)) : 0) : -1;
        for (int w = op + 1; w <= o; w++) gsplit[w] = gi;
      }
      __syncthreads();
      const int g_lo = __builtin_amdgcn_readfirstlane(gsplit[wave]), g_hi = __builtin_amdgcn_readfirstlane(gsplit[wave + 1]);
      // Gaussians without an entry: zeros (lane = dim)
      for (int gi = wave; gi < G; gi += 16) {
        if (__builtin_amdgcn_readfirstlane(cnt[gi]) != 0) continue;
        for (int d = lane; d < D; d += 64) wf[(size_t)gi * D + d] = 0.0;
        if (lane == 0) gm[gi] = 0.f;
      }
      if (g_lo < g_hi) {
        const int e_lo = __builtin_amdgcn_readfirstlane(off[g_lo]), e_hi = __builtin_amdgcn_readfirstlane(off[g_hi]);
        for (int d0 = 0; d0 < D; d0 += 64) {
          const int d = d0 + lane;
          const bool dv = d < D;
          const float *xd = xs + (dv ? d : 0);
          int gcur = -1;
          double a = 0.0;
          float ga = 0.f;
          for (int e = e_lo; e < e_hi; e += 8) {
            float w8[8], x8[8];
            int g8[8], f8[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { const int ee = e + i < e_hi ? e + i : e_hi - 1; w8[i] = ows[ee]; f8[i] = ofr[ee]; g8[i] = ogi[ee]; }
#pragma unroll
            for (int i = 0; i < 8; i++) x8[i] = xd[f8[i] * D];
            // (the products are exact in double -- 24 x 24 bits -- so a + w * x is the same sum formed in one step or in two; formed
            // here, ahead of the chain, an entry costs the chain one addition)
            double p8[8];
#pragma unroll
            for (int i = 0; i < 8; i++) p8[i] = (double)w8[i] * (double)x8[i];
            if (e + 8 <= e_hi && __builtin_amdgcn_readfirstlane(g8[7]) == gcur) {
              // eight more entries of the current Gaussian (the list is sorted): no test per entry
#pragma unroll
              for (int i = 0; i < 8; i++) { a += p8[i]; ga += w8[i]; }
              continue;
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
              if (e + i < e_hi) {                                   // (wave-uniform)
                const int gn = __builtin_amdgcn_readfirstlane(g8[i]);
                if (gn != gcur) {
                  if (gcur >= 0) { if (dv) wf[(size_t)gcur * D + d] = a; if (d0 == 0 && lane == 0) gm[gcur] = ga; }
                  gcur = gn; a = 0.0; ga = 0.f;
                }
                a += p8[i];
                ga += w8[i];
              }
            }
          }
          if (gcur >= 0) { if (dv) wf[(size_t)gcur * D + d] = a; if (d0 == 0 && lane == 0) gm[gcur] = ga; }
        }
      }
      RS_AT(4);
      continue;
    }
    for (int d0 = 0; d0 < D; d0 += 64) {
      const int d = d0 + lane;
      const bool dv = d < D;
      const float *xd = xs + (dv ? d : 0);
      for (int j0 = wave; j0 < G; j0 += 16 * PG) {
        double acc[PG];
#pragma unroll
        for (int q = 0; q < PG; q++) {
          const int gi = j0 + 16 * q;
          acc[q] = (gi < G && dv && !first && !(RS_ACC_ABLATE & 4)) ? wf[(size_t)gi * D + d] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < PG; q++) {
          const int gi = j0 + 16 * q;
          if (gi >= G) continue;
          const int c = __builtin_amdgcn_readfirstlane(cnt[gi]);
          if (c == 0 && !first) continue;
          const int o0 = __builtin_amdgcn_readfirstlane(off[gi]);
          const unsigned short *fr = ofr + o0;
          const float *ws = ows + o0;
          double a = acc[q];
          float ga = (d0 == 0 && !first) ? gm[gi] : 0.f;
          int k = 0;
          for (; k + 4 <= c; k += 4) {
            const float w0 = ws[k], w1 = ws[k + 1], w2 = ws[k + 2], w3 = ws[k + 3];
            const float x0 = xd[fr[k] * D], x1 = xd[fr[k + 1] * D], x2 = xd[fr[k + 2] * D], x3 = xd[fr[k + 3] * D];
            a += (double)w0 * (double)x0; a += (double)w1 * (double)x1; a += (double)w2 * (double)x2; a += (double)w3 * (double)x3;
            ga += w0; ga += w1; ga += w2; ga += w3;
          }
          for (; k < c; k++) { const float w0 = ws[k]; a += (double)w0 * (double)xd[fr[k] * D]; ga += w0; }
          if (dv && (!(RS_ACC_ABLATE & 4) || a == 12345.0)) wf[(size_t)gi * D + d] = a;
          if (d0 == 0 && lane == 0) gm[gi] = ga;
        }
      }
    }
    RS_AT(4);
  }
#ifdef RS_ACC_PROFILE
  if ((tid & 63) == 0 && (wave == 0 || wave == 15) && u % 61 == 0)
    printf("acc utt %d wave %d: stage %lld hist %lld prefix %lld place %lld sums %lld\n", u, wave, ap[0], ap[1], ap[2], ap[3], ap[4]);
#endif
#undef RS_AT
}

static size_t IvecAccumSmemBytes(const IvecDev &iv, int nseg) {
  const size_t ne = (size_t)kAccTC * iv.num_gselect;
  return (size_t)kAccTC * iv.feat_dim * 4 + ne * 8 + ((size_t)nseg * iv.num_gauss + 2 * (size_t)iv.num_gauss + 1) * 4 + ne * 8 + 17 * 4 + 64;
}

void LaunchIvecAccumulate(const IvecDev &iv, const BatchGeom &g, const float *lda, int ld, const int *post_idx,
                          const float *post_w, const int *frame_begin, const int *frame_end, double *gamma,
                          double *wfeats, bool fresh, hipStream_t s, int geo_mod) {
  if (g.n_utts == 0) return;
  // gamma is kept in float (GaussInfo::tot_weight is a BaseFloat); the buffer is sized for doubles, we use
  // its first half as floats.
  int nseg = kAccNSeg;
  while (nseg > 1 && IvecAccumSmemBytes(iv, nseg) > 150 * 1024) nseg >>= 1;
  const size_t smem = IvecAccumSmemBytes(iv, nseg);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&IvecAccumKernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(IvecAccumKernel, dim3(g.n_utts), dim3(1024), smem, s, iv, g, lda, ld, post_idx, post_w, frame_begin,
                     frame_end, reinterpret_cast<float *>(gamma), wfeats, nseg, fresh ? 1 : 0, geo_mod);
}

constexpr int kIvecUB = 8;        // utterances per workgroup in the two batch products
constexpr int kIvecKS = 32;       // Gaussian ranges the linear-term product is split into

// Per utterance: tot = sum_g gamma_g (double sum of the float per-Gaussian totals, g ascending), the max_count prior
// rescaling step of AccStats (ivector-extractor.cc:634-649) as `change`, and num_frames += tot.
__global__ __launch_bounds__(64) void IvecTotKernel(IvecDev iv, const float *__restrict__ gamma, double *__restrict__ num_frames,
                                                    double *__restrict__ change) {
  __shared__ float gs[4096];
  const int u = blockIdx.x, G = iv.num_gauss;
  double tot = 0.0;
  for (int g0 = 0; g0 < G; g0 += 4096) {
    const int n = G - g0 < 4096 ? G - g0 : 4096;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 64) gs[i] = gamma[(size_t)u * G + g0 + i];
    __syncthreads();
    if (threadIdx.x == 0) for (int i = 0; i < n; i++) tot += (double)gs[i];
  }
  if (threadIdx.x == 0) {
    double ch = 0.0;
    const double oldn = num_frames[u], newn = oldn + tot;
    if (iv.max_count > 0.0f) {
      const double mc = (double)iv.max_count;
      const double old_scale = (oldn > mc ? oldn : mc) / mc, new_scale = (newn > mc ? newn : mc) / mc;
      ch = new_scale - old_scale;
    }
    change[u] = ch;
    num_frames[u] = newn;
  }
}

// partial[ks][u][i] = sum over the Gaussians of range ks and all d of Sigma_inv_M[g][d][i] * wfeats[u][g][d]
__global__ __launch_bounds__(128) void IvecLinearPartialKernel(IvecDev iv, int n_utts, const double *__restrict__ wfeats,
                                                               double *__restrict__ partial) {
  constexpr int UB = kIvecUB, GB = 4;
  __shared__ double wfl[GB][UB][128];
  const int D = iv.feat_dim, G = iv.num_gauss, I = iv.ivec_dim;
  const int u0 = blockIdx.x * UB, ks = blockIdx.y;
  const int per = (G + kIvecKS - 1) / kIvecKS, g_begin = ks * per, g_end = g_begin + per < G ? g_begin + per : G;
  for (int i0 = 0; i0 < I; i0 += 128) {
    const int i = i0 + threadIdx.x, ic = i < I ? i : I - 1;
    double acc[UB];
#pragma unroll
    for (int uu = 0; uu < UB; uu++) acc[uu] = 0.0;
    for (int gb = g_begin; gb < g_end; gb += GB) {
      __syncthreads();
      for (int e = threadIdx.x; e < GB * UB * D; e += 128) {
        const int d = e % D, uu = (e / D) % UB, gg = e / (D * UB);
        const int gi = gb + gg, u = u0 + uu;
        wfl[gg][uu][d] = (gi < g_end && u < n_utts) ? wfeats[((size_t)u * G + gi) * D + d] : 0.0;
      }
      __syncthreads();
      for (int gg = 0; gg < GB; gg++) {
        const int gi = gb + gg < g_end ? gb + gg : g_end - 1;      // past the range: wfl is zero there
        const double *sim = iv.sigma_inv_M + (size_t)gi * D * I + ic;
        // DC independent loads in flight per lane, then DC x UB FMAs against LDS broadcasts
        constexpr int DC = 8;
        int d0 = 0;
        for (; d0 + DC <= D; d0 += DC) {
          double sv[DC];
#pragma unroll
          for (int dd = 0; dd < DC; dd++) sv[dd] = sim[(size_t)(d0 + dd) * I];
#pragma unroll
          for (int dd = 0; dd < DC; dd++)
#pragma unroll
            for (int uu = 0; uu < UB; uu++) acc[uu] += sv[dd] * wfl[gg][uu][d0 + dd];
        }
        for (; d0 < D; d0++) {
          const double sv = sim[(size_t)d0 * I];
#pragma unroll
          for (int uu = 0; uu < UB; uu++) acc[uu] += sv * wfl[gg][uu][d0];
        }
      }
    }
    if (i < I)
#pragma unroll
      for (int uu = 0; uu < UB; uu++)
        if (u0 + uu < n_utts) partial[((size_t)ks * n_utts + u0 + uu) * I + i] = acc[uu];
  }
}
// linear[u][i] += sum_ks partial[ks][u][i] (fixed order) -- and, by the workgroups behind those (blockIdx.x >= reduce_blocks), what
// IvecTotKernel does for one utterance each (one launch instead of two in front of the quadratic product, which needs both)
__global__ __launch_bounds__(256) void IvecLinearReduceKernel(IvecDev iv, int n_utts, const double *__restrict__ partial, double *__restrict__ linear,
                                                              int reduce_blocks, const float *__restrict__ gamma, double *__restrict__ num_frames,
                                                              double *__restrict__ change) {
  const int I = iv.ivec_dim;
  if ((int)blockIdx.x < reduce_blocks) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_utts * I) return;
    double acc = 0.0;
    for (int ks = 0; ks < kIvecKS; ks++) acc += partial[(size_t)ks * n_utts * I + idx];
    linear[idx] += acc;
    return;
  }
  // tot = sum_g gamma_g in double: g ascending within each of 256 contiguous runs, then the runs' sums ascending (IvecTotKernel
  // walked all of them on one thread: 10 us for 512 Gaussians; a double sum of a few hundred floats is exact to 1e-16 either way)
  __shared__ double part[256];
  const int u = blockIdx.x - reduce_blocks, G = iv.num_gauss, tid = threadIdx.x;
  const int per = (G + 255) / 256, g0 = tid * per, g1 = g0 + per < G ? g0 + per : G;
  double t = 0.0;
  for (int gi = g0; gi < g1; gi++) t += (double)gamma[(size_t)u * G + gi];
  part[tid] = t;
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    for (int i = 0; i < 256; i++) tot += part[i];
    double ch = 0.0;
    const double oldn = num_frames[u], newn = oldn + tot;
    if (iv.max_count > 0.0f) {
      const double mc = (double)iv.max_count;
      const double old_scale = (oldn > mc ? oldn : mc) / mc, new_scale = (newn > mc ? newn : mc) / mc;
      ch = new_scale - old_scale;
    }
    change[u] = ch;
    num_frames[u] = newn;
  }
}

// quadratic[u][k] += sum_g gamma[u][g] U_g[k] (+ the prior rescaling on the diagonal and on linear[0])
__global__ __launch_bounds__(128) void IvecQuadKernel(IvecDev iv, int n_utts, const float *__restrict__ gamma,
                                                      const double *__restrict__ change, double *__restrict__ quadratic,
                                                      double *__restrict__ linear) {
  constexpr int UB = kIvecUB, GC = 512;
  __shared__ float gml[UB][GC];
  const int G = iv.num_gauss, I = iv.ivec_dim, usz = I * (I + 1) / 2;
  const int u0 = blockIdx.y * UB;
  const int k = blockIdx.x * 128 + threadIdx.x, kc = k < usz ? k : usz - 1;
  double acc[UB];
#pragma unroll
  for (int uu = 0; uu < UB; uu++) acc[uu] = 0.0;
  for (int g0 = 0; g0 < G; g0 += GC) {
    const int n = G - g0 < GC ? G - g0 : GC;
    __syncthreads();
    for (int e = threadIdx.x; e < UB * n; e += 128) {
      const int uu = e / n, gi = e % n;
      gml[uu][gi] = u0 + uu < n_utts ? gamma[(size_t)(u0 + uu) * G + g0 + gi] : 0.f;
    }
    __syncthreads();
    const double *Uc = iv.U + (size_t)g0 * usz + kc;
    constexpr int GU = 8;         // independent loads in flight per lane
    int gi = 0;
    for (; gi + GU <= n; gi += GU) {
      double uv[GU];
#pragma unroll
      for (int q = 0; q < GU; q++) uv[q] = Uc[(size_t)(gi + q) * usz];
#pragma unroll
      for (int q = 0; q < GU; q++)
#pragma unroll
        for (int uu = 0; uu < UB; uu++) acc[uu] += (double)gml[uu][gi + q] * uv[q];
    }
    for (; gi < n; gi++) {
      const double uv = Uc[(size_t)gi * usz];
#pragma unroll
      for (int uu = 0; uu < UB; uu++) acc[uu] += (double)gml[uu][gi] * uv;
    }
  }
  if (k >= usz) return;
  // is k a diagonal element?  k = r(r+1)/2 + r
  int r = (int)((sqrt(8.0 * (double)k + 1.0) - 1.0) * 0.5);
  while ((r + 1) * (r + 2) / 2 <= k) r++;
  while (r * (r + 1) / 2 > k) r--;
  const bool diag = (k == r * (r + 1) / 2 + r);
#pragma unroll
  for (int uu = 0; uu < UB; uu++) {
    const int u = u0 + uu;
    if (u >= n_utts) break;
    const double ch = change[u];
    quadratic[(size_t)u * usz + k] += acc[uu] + ((diag && ch != 0.0) ? ch : 0.0);
    if (k == 0 && ch != 0.0) linear[(size_t)u * I] += iv.prior_offset * ch;
  }
}

// ---- the two batch products on the fp64 matrix cores (v_mfma_f64_16x16x4_f64; C/D layout of the f64 form: column = lane & 15,
// row = (lane >> 4) + 4 * reg -- NOT the f32 map).  Same sums in the same order (Gaussians / feature dims ascending, one
// accumulator per output element), so the estimator state matches the vector kernels to fp64 rounding; what changes is the
// operand traffic: a wave multiplies a 64-utterance x 4 slab of the per-utterance statistics (LDS) with a 4 x 16 slab of the
// model matrix (one 8-byte load per lane) in four MFMAs, instead of one LDS broadcast read per fused multiply-add.
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kMmU = 64;           // utterances per workgroup (4 M tiles of 16)
// quadratic[u][k] += sum_g gamma[u][g] U_g[k]: [n_utts x G] x [G x usz]; workgroup = 4 waves = 4 column tiles of 16
__global__ __launch_bounds__(256) void IvecQuadMfmaKernel(IvecDev iv, int n_utts, const float *__restrict__ gamma,
                                                          const double *__restrict__ change, double *__restrict__ quadratic,
                                                          double *__restrict__ linear) {
  constexpr int GC = 128;                                   // Gaussians per LDS chunk
  __shared__ float gml[kMmU][GC + 1];
  const int G = iv.num_gauss, I = iv.ivec_dim, usz = I * (I + 1) / 2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int u0 = blockIdx.y * kMmU, k0 = (blockIdx.x * 4 + wave) * 16;
  const int kcol = k0 + lr < usz ? k0 + lr : usz - 1;
  f64x4 acc[4];
#pragma unroll
  for (int m = 0; m < 4; m++) acc[m] = f64x4{0.0, 0.0, 0.0, 0.0};
  // eight k-steps of model-matrix values per request group, and the NEXT group (of this chunk or the first of the next one) is
  // requested before the MFMAs of the current one: with at most one wave per SIMD nothing else hides the round trip (68 us for
  // 64 utterances before, all of it a chain of 16 exposed L2 / HBM latencies)
  auto fetch = [&](int g0, int n, int gi, double (&b)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int gk = gi + 4 * q + lk < n ? gi + 4 * q + lk : n - 1;             // (past the end: gml is zero there)
      b[q] = iv.U[(size_t)(g0 + gk) * usz + kcol];
    }
  };
  // (occupancy chunk of the next step: global -> registers during the MFMAs of this one, to LDS between two barriers afterwards)
  constexpr int SPT = kMmU * GC / 256;
  float stg[SPT];
  auto stage_load = [&](int g0, int n) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < SPT; t++) {
      const int e = threadIdx.x + 256 * t, uu = e / GC, gi = e % GC;
      stg[t] = (u0 + uu < n_utts && gi < n) ? gamma[(size_t)(u0 + uu) * G + g0 + gi] : 0.f;
    }
  };
  auto stage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < SPT; t++) { const int e = threadIdx.x + 256 * t; gml[e / GC][e % GC] = stg[t]; }
  };
  double bc[8], bn[8];
  if (G > 0) {
    const int n0 = G < GC ? G : GC;
    stage_load(0, n0);
    if (k0 < usz) fetch(0, n0, 0, bc);
    if (threadIdx.x < kMmU) gml[threadIdx.x][GC] = 0.f;
    stage_store();
    __syncthreads();
  }
  for (int g0 = 0; g0 < G; g0 += GC) {
    const int n = G - g0 < GC ? G - g0 : GC;
    const bool more = g0 + GC < G;
    const int n_next = more ? (G - g0 - GC < GC ? G - g0 - GC : GC) : 0;
    if (more) stage_load(g0 + GC, n_next);
    if (k0 < usz) {
      for (int gi = 0; gi < n; gi += 32) {
        if (gi + 32 < n) fetch(g0, n, gi + 32, bn);
        else if (more) fetch(g0 + GC, n_next, 0, bn);
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int gc = gi + 4 * q + lk < GC ? gi + 4 * q + lk : GC;
#pragma unroll
          for (int m = 0; m < 4; m++) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)gml[16 * m + lr][gc], bc[q], acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) bc[q] = bn[q];
      }
    }
    if (more) {
      __syncthreads();
      stage_store();
      __syncthreads();
    }
  }
  if (k0 + lr >= usz) return;
  const int k = k0 + lr;
  int r = (int)((sqrt(8.0 * (double)k + 1.0) - 1.0) * 0.5);
  while ((r + 1) * (r + 2) / 2 <= k) r++;
  while (r * (r + 1) / 2 > k) r--;
  const bool diag = (k == r * (r + 1) / 2 + r);
#pragma unroll
  for (int m = 0; m < 4; m++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int u = u0 + 16 * m + lk + 4 * q;
      if (u >= n_utts) continue;
      const double ch = change[u];
      quadratic[(size_t)u * usz + k] += acc[m][q] + ((diag && ch != 0.0) ? ch : 0.0);
      if (k == 0 && ch != 0.0) linear[(size_t)u * I] += iv.prior_offset * ch;
    }
}
// ---- the same product with the model-matrix loads in inline asm and hand-placed waits (num_gauss a multiple of 128).  In the
// kernel above the compiler puts s_waitcnt vmcnt(0) in front of the MFMAs of every request group -- the group requested a moment
// earlier "for the next step" is waited for at once, so the 16 groups of a workgroup are 16 exposed memory round trips (39 us for a
// round of 64 streams against 15.6 us of fp64 MFMA time; profiles/r02/acc_ablate.txt).  Here the two register sets alternate
// without copies (a copy would read a set still in flight), the compiler does not see the loads, and each wait leaves exactly the
// newest group outstanding.  Same products, same order.
#define RS_GLOAD8(dst, ptr) __asm__ volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define RS_WAIT_SET(CNT, b)                                                                                                          \
  __asm__ volatile("s_waitcnt vmcnt(" #CNT ")"                                                                                       \
                   : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) : : "memory")
template <int NCH, int MT>                                          // num_gauss = 128 NCH; the chunk loop is unrolled: a register set that is
                                                            // live around a loop's back edge gets copied there, in flight or not
__global__ __launch_bounds__(256) void IvecQuadMfmaAsmKernel(IvecDev iv, int n_utts, const float *__restrict__ gamma,
                                                          const double *__restrict__ change, double *__restrict__ quadratic,
                                                          double *__restrict__ linear) {
  constexpr int GC = 128;                                   // Gaussians per LDS chunk
  constexpr int MU = 16 * MT;                               // utterances per workgroup (MT = 1: a few dozen streams over four times the workgroups)
  __shared__ float gml[MU][GC + 1];
  const int G = iv.num_gauss, I = iv.ivec_dim, usz = I * (I + 1) / 2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int u0 = blockIdx.y * MU, k0 = (blockIdx.x * 4 + wave) * 16;
  const int kcol = k0 + lr < usz ? k0 + lr : usz - 1;
  f64x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; m++) acc[m] = f64x4{0.0, 0.0, 0.0, 0.0};
  // eight k-steps of model-matrix values per request group, and the NEXT group (of this chunk or the first of the next one) is
  // requested before the MFMAs of the current one: with at most one wave per SIMD nothing else hides the round trip (68 us for
  // 64 utterances before, all of it a chain of 16 exposed L2 / HBM latencies)
#define RS_QFETCH(b, gfirst)                                                                 \
  _Pragma("unroll") for (int q = 0; q < 8; q++) RS_GLOAD8(b[q], iv.U + (size_t)((gfirst) + 4 * q + lk) * usz + kcol)
#define RS_QMMA(b, gi)                                                                       \
  _Pragma("unroll") for (int q = 0; q < 8; q++) {                                           \
    const int gc = (gi) + 4 * q + lk;                                                        \
    _Pragma("unroll") for (int m = 0; m < MT; m++)                                          \
      acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)gml[16 * m + lr][gc], b[q], acc[m], 0, 0, 0); \
  }
  // (occupancy chunk of the next step: global -> registers during the MFMAs of this one, to LDS between two barriers afterwards)
  constexpr int SPT = MU * GC / 256;
  float stg[SPT];
  // (also in asm, all of them always issued -- rows past n_utts re-read the last utterance, their results are never stored -- so
  // that the waits below can count them: a load the compiler knows about makes it wait vmcnt(0) before the LDS store, request
  // group in flight included)
  auto stage_load = [&](int g0) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < SPT; t++) {
      const int e = threadIdx.x + 256 * t, uu = e / GC, gi = e % GC;
      const int ur = u0 + uu < n_utts ? u0 + uu : n_utts - 1;
      __asm__ volatile("global_load_dword %0, %1, off" : "=v"(stg[t]) : "v"(gamma + (size_t)ur * G + g0 + gi) : "memory");
    }
  };
  auto stage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < SPT; t++) { const int e = threadIdx.x + 256 * t; gml[e / GC][e % GC] = stg[t]; }
  };
  // Three register sets in rotation, two request groups ahead of the MFMAs: one group of MFMAs is about 1 us, a model-matrix
  // round trip about 2 (one group ahead: 58 -> 49 us for the headline batch, 39.5 -> 31.5 us for 64 streams, still a chain of
  // round trips).  Wait counts: 8 per group requested after the awaited one, + the 32 staging loads where they were issued in
  // between (the two steps after a chunk's first).
  double bq[3][8];                                          // (column tiles past usz compute on the clamped column and store nothing)
#define RS_WAIT_N(n, b)                                                                                                 \
  do {                                                                                                                  \
    if ((n) == 0) { RS_WAIT_SET(0, b); } else if ((n) == 8) { RS_WAIT_SET(8, b); } else if ((n) == 16) { RS_WAIT_SET(16, b); } \
    else if ((n) == 24) { RS_WAIT_SET(24, b); } else if ((n) == 40) { RS_WAIT_SET(40, b); } else { RS_WAIT_SET(48, b); }  \
  } while (0)
  constexpr int NG = 4 * NCH;
  stage_load(0);
  RS_QFETCH(bq[0], 0);
  if (NG > 1) { RS_QFETCH(bq[1], 32); }
  __asm__ volatile("s_waitcnt vmcnt(16)" ::: "memory");      // the staged chunk; the two groups stay in flight
  stage_store();
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NG; t++) {
    const int ch = t / 4, r = t % 4;
    const bool more = ch + 1 < NCH;
    if (t + 2 < NG) { RS_QFETCH(bq[(t + 2) % 3], (t + 2) * 32); }
    const int newer = NG - 1 - t < 2 ? NG - 1 - t : 2;
    static_assert(SPT == 8 || SPT == 32, "wait counts below");
    RS_WAIT_N(8 * newer + ((more && (r == 1 || r == 2)) ? SPT : 0), bq[t % 3]);
    if (r == 0 && more) stage_load((ch + 1) * GC);
    RS_QMMA(bq[t % 3], r * 32);
    if (r == 3 && more) {
      __syncthreads();
      stage_store();
      __syncthreads();
    }
  }
#undef RS_WAIT_N
#undef RS_QFETCH
#undef RS_QMMA
  if (k0 + lr >= usz) return;
  const int k = k0 + lr;
  int r = (int)((sqrt(8.0 * (double)k + 1.0) - 1.0) * 0.5);
  while ((r + 1) * (r + 2) / 2 <= k) r++;
  while (r * (r + 1) / 2 > k) r--;
  const bool diag = (k == r * (r + 1) / 2 + r);
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int u = u0 + 16 * m + lk + 4 * q;
      if (u >= n_utts) continue;
      const double ch = change[u];
      quadratic[(size_t)u * usz + k] += acc[m][q] + ((diag && ch != 0.0) ? ch : 0.0);
      if (k == 0 && ch != 0.0) linear[(size_t)u * I] += iv.prior_offset * ch;
    }
}
// partial[ks][u][i] = sum over the Gaussians of range ks and all d of Sigma_inv_M[g][d][i] * wfeats[u][g][d]:
// [n_utts x (G D)] x [(G D) x I], K split into kIvecKS Gaussian ranges (reduced in fixed order by IvecLinearReduceKernel)
template <int MT>      // M tiles of 16 utterances per workgroup: 4 for batches, 1 for a few dozen streams (four times the workgroups, a quarter of the MFMAs each)
__global__ __launch_bounds__(256) void IvecLinearMfmaKernel(IvecDev iv, int n_utts, const double *__restrict__ wfeats,
                                                            double *__restrict__ partial) {
  constexpr int KC = 64;                                    // (Gaussian, dim) products per LDS chunk
  constexpr int MU = 16 * MT;
  __shared__ double wfl[MU][KC + 1];
  const int D = iv.feat_dim, G = iv.num_gauss, I = iv.ivec_dim;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int u0 = blockIdx.x * MU, ks = blockIdx.y;
  const int per = (G + kIvecKS - 1) / kIvecKS, g_begin = ks * per, g_end = g_begin + per < G ? g_begin + per : G;
  const long kk_begin = (long)g_begin * D, kk_end = (long)(g_end > g_begin ? g_end : g_begin) * D;      // rows of Sigma_inv_M [G D][I]
  {
    const int i0 = blockIdx.z * 64 + wave * 16;               // 64 columns per workgroup, 16 per wave
    const bool live = i0 < I;
    const int icol = i0 + lr < I ? i0 + lr : I - 1;
    f64x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = f64x4{0.0, 0.0, 0.0, 0.0};
    // (request groups of eight k-steps, the next group in flight during the MFMAs of the current one, as in IvecQuadMfmaKernel)
    auto fetch = [&](long c0, int n, int j, double (&b)[8]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int jk = j + 4 * q + lk < n ? j + 4 * q + lk : n - 1;
        b[q] = iv.sigma_inv_M[(size_t)(c0 + jk) * I + icol];
      }
    };
    // The statistics chunk of the NEXT step travels global -> registers while the MFMAs of this one run, and is written to LDS
    // between two barriers afterwards: staged synchronously (load, barrier, use) the ten chunks of a workgroup were ten exposed
    // memory round trips -- most of the kernel's 78 us.
    constexpr int SPT = MU * KC / 256;                       // staged values per thread and chunk
    double stg[SPT];
    auto stage_load = [&](long c0, int n) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < SPT; t++) {
        const int e = threadIdx.x + 256 * t, uu = e / KC, j = e % KC;
        stg[t] = (u0 + uu < n_utts && j < n) ? wfeats[(size_t)(u0 + uu) * G * D + c0 + j] : 0.0;
      }
    };
    auto stage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < SPT; t++) { const int e = threadIdx.x + 256 * t; wfl[e / KC][e % KC] = stg[t]; }
    };
    double bc[8], bn[8];
    if (kk_begin < kk_end) {
      const int n0 = kk_end - kk_begin < KC ? (int)(kk_end - kk_begin) : KC;
      stage_load(kk_begin, n0);
      if (live) fetch(kk_begin, n0, 0, bc);
      if (threadIdx.x < MU) wfl[threadIdx.x][KC] = 0.0;
      stage_store();
      __syncthreads();
    }
    for (long c0 = kk_begin; c0 < kk_end; c0 += KC) {
      const int n = kk_end - c0 < KC ? (int)(kk_end - c0) : KC;
      const bool more = c0 + KC < kk_end;
      const int n_next = more ? (kk_end - c0 - KC < KC ? (int)(kk_end - c0 - KC) : KC) : 0;
      if (more) stage_load(c0 + KC, n_next);
      for (int j = 0; live && j < n; j += 32) {
        if (j + 32 < n) fetch(c0, n, j + 32, bn);
        else if (more) fetch(c0 + KC, n_next, 0, bn);
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int jc = j + 4 * q + lk < KC ? j + 4 * q + lk : KC;
#pragma unroll
          for (int m = 0; m < MT; m++) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(wfl[16 * m + lr][jc], bc[q], acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) bc[q] = bn[q];
      }
      if (more) {
        __syncthreads();
        stage_store();
        __syncthreads();
      }
    }
    if (i0 + lr < I) {
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int u = u0 + 16 * m + lk + 4 * q;
          if (u < n_utts) partial[((size_t)ks * n_utts + u) * I + i0 + lr] = acc[m][q];
        }
    }
  }
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

size_t IvecStatsScratchDoubles(const IvecDev &iv, int n_utts) { return (size_t)kIvecKS * n_utts * iv.ivec_dim + n_utts; }

void LaunchIvecStats(const IvecDev &iv, int n_utts, const double *gamma, const double *wfeats, double *linear,
                     double *quadratic, double *num_frames, double *scratch, hipStream_t s) {
  if (n_utts == 0) return;
  const float *gm = reinterpret_cast<const float *>(gamma);
  double *partial = scratch, *change = scratch + (size_t)kIvecKS * n_utts * iv.ivec_dim;
  const int ub = (n_utts + kIvecUB - 1) / kIvecUB, usz = iv.ivec_dim * (iv.ivec_dim + 1) / 2;
  const char *me = std::getenv("RS_IVEC_MFMA");          // read per call (tests flip it)
  const bool mfma = !(me && std::atoi(me) == 0);
  // a few dozen utterances (a round of streams): 16 per workgroup instead of 64, so that four times as many CUs share the two
  // products (same sums in the same order per element: the M tiling does not touch the k order)
  const bool narrow = n_utts <= 64;
  const int um64 = (n_utts + kMmU - 1) / kMmU, um = narrow ? (n_utts + 15) / 16 : um64;
  if (mfma && narrow) hipLaunchKernelGGL(IvecLinearMfmaKernel<1>, dim3(um, kIvecKS, (iv.ivec_dim + 63) / 64), dim3(256), 0, s, iv, n_utts, wfeats, partial);
  else if (mfma) hipLaunchKernelGGL(IvecLinearMfmaKernel<4>, dim3(um, kIvecKS, (iv.ivec_dim + 63) / 64), dim3(256), 0, s, iv, n_utts, wfeats, partial);
  else hipLaunchKernelGGL(IvecLinearPartialKernel, dim3(ub, kIvecKS), dim3(128), 0, s, iv, n_utts, wfeats, partial);
  const int reduce_blocks = (n_utts * iv.ivec_dim + 255) / 256;
  hipLaunchKernelGGL(IvecLinearReduceKernel, dim3(reduce_blocks + n_utts), dim3(256), 0, s, iv, n_utts, partial, linear, reduce_blocks, gm, num_frames, change);
  const char *ae = std::getenv("RS_IVEC_ASM");           // read per call (a test compares the two forms)
  const int quad_asm = ae ? std::atoi(ae) : 1;
#define RS_QUAD_ASM(N) do { if (narrow) hipLaunchKernelGGL((IvecQuadMfmaAsmKernel<N, 1>), dim3((usz + 63) / 64, um), dim3(256), 0, s, iv, n_utts, gm, change, quadratic, linear); \
                            else hipLaunchKernelGGL((IvecQuadMfmaAsmKernel<N, 4>), dim3((usz + 63) / 64, um), dim3(256), 0, s, iv, n_utts, gm, change, quadratic, linear); } while (0)
  if (mfma && quad_asm && iv.num_gauss == 512) RS_QUAD_ASM(4);
  else if (mfma && quad_asm && iv.num_gauss == 256) RS_QUAD_ASM(2);
  else if (mfma && quad_asm && iv.num_gauss == 1024) RS_QUAD_ASM(8);
  else if (mfma && quad_asm && iv.num_gauss == 128) RS_QUAD_ASM(1);
#undef RS_QUAD_ASM
  else if (mfma) hipLaunchKernelGGL(IvecQuadMfmaKernel, dim3((usz + 63) / 64, um64), dim3(256), 0, s, iv, n_utts, gm, change, quadratic, linear);
  else hipLaunchKernelGGL(IvecQuadKernel, dim3((usz + 127) / 128, ub), dim3(128), 0, s, iv, n_utts, gm, change, quadratic, linear);
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

// ---- fast path for ivector_dim <= 128: the quadratic term is expanded to a full n x n matrix in LDS so that a row's
// dot product reads consecutive addresses across lanes (conflict-free), the dot products of a CG step are reduced with
// DPP row rotations inside a wave and one LDS exchange across waves (one barrier per reduction instead of a tree).
// Same iteration (LinearCgd, matrix/optimization.cc:453-566) and the same decisions as IvecSolveKernel.
template <int CTRL>
__device__ __forceinline__ double DppD(double v) {
  // (old = 0 with bound_ctrl: the same value for a row rotation with every lane active -- all call sites are -- and one v_mov_b32_dpp
  // per half instead of copy + wait state + v_mov_b32_dpp)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ReadLaneD(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double WaveSumD(double v) {
  v += DppD<0x121>(v);
  v += DppD<0x122>(v);
  v += DppD<0x124>(v);
  v += DppD<0x128>(v);
  return (ReadLaneD(v, 0) + ReadLaneD(v, 16)) + (ReadLaneD(v, 32) + ReadLaneD(v, 48));
}
template <int K, int NW>
__device__ __forceinline__ void BlockSumK(const double (&v)[K], double (&out)[K], double (*xch)[NW][4], int &rb) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < K; k++) {
    const double w = WaveSumD(v[k]);
    if (lane == 0) xch[rb][wave][k] = w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; k++) {
    double t = xch[rb][0][k];
#pragma unroll
    for (int w = 1; w < NW; w++) t += xch[rb][w][k];
    out[k] = t;
  }
  rb ^= 1;
}

#ifndef RS_SOLVE_ABLATE               // measurement only (profiles/micro/kernel_ablate.sh): 1 = no CG iterations, 2 = no expansion either,
#define RS_SOLVE_ABLATE 0            // 4 = one iteration
#endif
// LinearCgd (matrix/optimization.cc:453-566) on the expanded matrix A (n x n in LDS), right-hand side b and start x held one
// element per thread (tid < n); the code of IvecSolveFullKernel, shared with IvecChainKernel so that both run the same arithmetic.
// (NFIX = 100 with four waves: a thread's half row of the matrix -- 50 elements, the same in all of a solve's products -- is read from
// LDS once and kept in registers; a product then reads the vector only.  The four waves of a workgroup sit on four SIMDs and share one
// LDS pipe: a product was 100 wave reads of the matrix + 100 broadcast reads of the vector, and it was the pipe it waited for.)
template <int NW, int NFIX = 0>
__device__ __forceinline__ double IvecCgSolve(const IvecDev &iv, const double *A, double *xs, double *ps, double (*xch)[NW][4], int &rb,
                                              bool mine, int tid, int n_arg, double b, double x) {
  const int n = NFIX ? NFIX : n_arg;
  if (tid == 0 && x == 0.0) x = iv.prior_offset;          // GetIvector: better initial guess
  if (mine) xs[tid] = x;
  __syncthreads();
  // Row products with TWO threads per row when the workgroup has them (n <= 32 NW): thread tid < n takes the first half of the
  // columns, thread tid + 32 NW the second, the halves meet through LDS (hs = ps + n; one more barrier per product).  With one
  // thread per row 100 of the 256 threads worked and a product was 200 LDS reads + 100 dependent-issue fp64 FMAs per thread, the
  // longest piece of the per-chunk chain of a stream advance (four accumulators each, summed pairwise at the end: the order of an
  // fp64 sum is free at the 1e-4 the iVector is held to; batch and stream paths run this same code and stay bit-equal to each other).
  double *hs = ps + n;
  const bool two = NW > 1 && n <= 32 * NW;                 // (tid + 32 NW < 64 NW for every row)
  const bool second = two && tid >= 32 * NW && tid - 32 * NW < n;
  const int row = second ? tid - 32 * NW : tid;
  const int c_lo = second ? n / 2 : 0, c_hi = two ? (second ? n : n / 2) : n;
  constexpr bool in_regs = NFIX == 100 && NW == 4;
  constexpr int HALF = in_regs ? NFIX / 2 : 1;
  double areg[HALF];
  if (in_regs) {
#pragma unroll
    for (int i = 0; i < HALF; i++) areg[i] = (mine || second) ? A[(size_t)(c_lo + i) * n + row] : 0.0;
  }
  auto matvec = [&](const double *vec) __attribute__((always_inline)) {
    constexpr int MB = 10;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (in_regs) {
      if (mine || second) {
        static_assert(!in_regs || HALF % MB == 0, "whole batches");
#pragma unroll
        for (int c = 0; c < HALF; c += MB) {
          double w[MB];
#pragma unroll
          for (int i = 0; i < MB; i++) w[i] = vec[c_lo + c + i];
#pragma unroll
          for (int i = 0; i + 4 <= MB; i += 4) { a0 += areg[c + i] * w[i]; a1 += areg[c + i + 1] * w[i + 1]; a2 += areg[c + i + 2] * w[i + 2]; a3 += areg[c + i + 3] * w[i + 3]; }
#pragma unroll
          for (int i = MB / 4 * 4; i < MB; i++) a0 += areg[c + i] * w[i];
        }
      }
    } else if (mine || second) {
      int c = c_lo;
      for (; c + MB <= c_hi; c += MB) {
        double a[MB], w[MB];
#pragma unroll
        for (int i = 0; i < MB; i++) { a[i] = A[(size_t)(c + i) * n + row]; w[i] = vec[c + i]; }
#pragma unroll
        for (int i = 0; i + 4 <= MB; i += 4) { a0 += a[i] * w[i]; a1 += a[i + 1] * w[i + 1]; a2 += a[i + 2] * w[i + 2]; a3 += a[i + 3] * w[i + 3]; }
#pragma unroll
        for (int i = MB / 4 * 4; i < MB; i++) a0 += a[i] * w[i];
      }
      for (; c < c_hi; c++) a0 += A[(size_t)c * n + row] * vec[c];
    }
    double sum = (a0 + a1) + (a2 + a3);
    if (two) {
      if (second) hs[row] = sum;
      __syncthreads();
      if (mine) sum += hs[tid];
      // (the next product rewrites hs behind the barrier its caller puts in front of it)
    }
    return mine ? sum : 0.0;      // (the threads past ivec_dim add exact zeros to the block sums)
  };
  // p0 = b - A x0 ; r0 = -p0
  double p = b - matvec(xs), r = -p;
  if (!mine) { p = 0.0; r = 0.0; }
  double in1[1] = {r * r}, out1[1];
  BlockSumK<1, NW>(in1, out1, xch, rb);
  double r_cur = out1[0], r_recompute = r_cur;
  const double max_error_sq = DBL_MIN, residual_factor = (double)(0.01f * 0.01f), inv_residual_factor = 1.0 / residual_factor;
  for (int k = 0; k < n + 5 && k != ((RS_SOLVE_ABLATE & 3) ? 0 : (RS_SOLVE_ABLATE & 4) ? 1 : iv.num_cg_iters); k++) {
    if (mine) ps[tid] = p;
    __syncthreads();
    const double ap = matvec(ps);
    double in2[2] = {p * r, p * ap}, out2[2];
    BlockSumK<2, NW>(in2, out2, xch, rb);
    const double alpha = -out2[0] / out2[1];
    x += alpha * p;
    r += alpha * ap;
    in1[0] = r * r;
    BlockSumK<1, NW>(in1, out1, xch, rb);
    double r_next = out1[0];
    if (r_next < residual_factor * r_recompute || r_next > inv_residual_factor * r_recompute) {
      if (mine) xs[tid] = x;
      __syncthreads();
      const double ax = matvec(xs);          // (every thread: the product has a barrier inside)
      r = mine ? ax - b : 0.0;
      in1[0] = r * r;
      BlockSumK<1, NW>(in1, out1, xch, rb);
      r_next = out1[0];
      r_recompute = r_next;
    }
    if (r_next <= max_error_sq) break;
    const double beta = r_next / r_cur;
    p = p * beta - r;
    r_cur = r_next;
  }
  return x;
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void IvecSolveFullKernel(IvecDev iv, const double *__restrict__ linear,
                                                               const double *__restrict__ quadratic, const double *__restrict__ num_frames,
                                                               double *__restrict__ xio, float *__restrict__ ivec_out, int ldo,
                                                               const int *__restrict__ out_row, const int *__restrict__ active) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ double xch[2][NW][4];
  int rb = 0;
  const int u = blockIdx.x, tid = threadIdx.x, n = iv.ivec_dim;
  const int orow = out_row ? out_row[u] : u;
  if (orow < 0) return;                                    // this utterance has no chunk at this step
  const bool solve = active ? active[u] != 0 : true;       // 0: re-emit the current estimate (no new frames)
  const int usz = n * (n + 1) / 2;
  double *A = reinterpret_cast<double *>(smem_raw);        // n x n, symmetric
  double *xs = A + (size_t)n * n, *ps = xs + n;            // the vectors a row product reads
  const bool mine = tid < n;
  const bool have = num_frames[u] > 0.0;
  double x = mine ? xio[(size_t)u * n + tid] : 0.0;
  if (solve && have) {
    // expand the packed lower triangle: element k = r(r+1)/2 + c  (c <= r); EB loads per thread in flight (the expansion was
    // 13 of the kernel's 57 us: a double-precision square root per element and the 40 KB read a few loads at a time)
    constexpr int EB = 10;
    const double *qu = quadratic + (size_t)u * usz;
    for (int k0 = tid; k0 < ((RS_SOLVE_ABLATE & 2) ? 0 : usz); k0 += 64 * NW * EB) {
      double v[EB];
#pragma unroll
      for (int j = 0; j < EB; j++) { const int k = k0 + j * 64 * NW; v[j] = k < usz ? qu[k] : 0.0; }
#pragma unroll
      for (int j = 0; j < EB; j++) {
        const int k = k0 + j * 64 * NW;
        if (k >= usz) continue;
        int r = (int)((sqrtf(8.f * (float)k + 1.f) - 1.f) * 0.5f);
        while ((r + 1) * (r + 2) / 2 <= k) r++;
        while (r * (r + 1) / 2 > k) r--;
        const int cc = k - r * (r + 1) / 2;
        A[(size_t)r * n + cc] = v[j];
        A[(size_t)cc * n + r] = v[j];
      }
    }
    const double b = mine ? linear[(size_t)u * n + tid] : 0.0;
    x = n == 100 ? IvecCgSolve<NW, 100>(iv, A, xs, ps, xch, rb, mine, tid, n, b, x) : IvecCgSolve<NW>(iv, A, xs, ps, xch, rb, mine, tid, n, b, x);
  } else if (solve) {
    x = (tid == 0) ? iv.prior_offset : 0.0;
  }
  if (mine) {
    if (solve) xio[(size_t)u * n + tid] = x;
    float v = (float)x;
    if (tid == 0) v = (float)((double)v - iv.prior_offset);   // (*feat)(0) -= PriorOffset() on the float copy
    ivec_out[(size_t)orow * ldo + tid] = v;
  }
}

// A stream's new chunks in ONE launch (round 5).  The per-chunk chain of an advance -- accumulate, two batch products, CG solve, each a
// latency-bound launch of ~50 us on a few dozen streams, times two or three chunks -- was the busiest queue of the streams workload.
// The statistics of every (chunk, stream) pair are now computed side by side as INCREMENTS (LaunchIvecAccumulate / LaunchIvecStats on
// K x n pseudo-utterances that start from zero), and this kernel, a workgroup per stream, walks the stream's chunks in order:
// AccStats' bookkeeping (ivector-extractor.cc:611-668: num_frames, the max_count prior rescaling, linear and quadratic terms += the
// chunk's increment) on the estimator state in place, then GetIvector's warm-started CG (IvecCgSolve), then the chunk's iVector row.
//   dlin / dquad / dtot: [K][n][.] increments (dtot = the chunk's sum of posteriors); out_row / active: [K][n] (row -1: no such chunk;
//   active 0: no new frames since the last estimate, re-emit it); slot (null = u): the stream's row of lin / quad / numf / x.
template <int NW>
__global__ __launch_bounds__(64 * NW) void IvecChainKernel(IvecDev iv, int n_utts, int K, const double *__restrict__ dlin, const double *__restrict__ dquad,
                                                           const double *__restrict__ dtot, double *__restrict__ lin, double *__restrict__ quad,
                                                           double *__restrict__ numf, double *__restrict__ xst, const int *__restrict__ slot,
                                                           float *__restrict__ ivec_out, int ldo, const int *__restrict__ out_row,
                                                           const int *__restrict__ active) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ double xch[2][NW][4];
  int rb = 0;
  const int u = blockIdx.x, tid = threadIdx.x, n = iv.ivec_dim, usz = n * (n + 1) / 2;
  const size_t sl = slot ? (size_t)slot[u] : (size_t)u;
  double *A = reinterpret_cast<double *>(smem_raw);        // n x n, symmetric
  double *xs = A + (size_t)n * n, *ps = xs + n;
  const bool mine = tid < n;
  double *qu = quad + sl * usz, *li = lin + sl * n;
  double nf = numf[sl];                                    // (every thread carries the same value)
  double x = mine ? xst[sl * n + tid] : 0.0;
  bool touched = false;
  for (int k = 0; k < K; k++) {
    const size_t p = (size_t)k * n_utts + u;
    const int orow = out_row[p];
    if (orow < 0) continue;                                // (workgroup-uniform) this stream has no chunk at this step
    const bool act = active[p] != 0;
    if (act) {
      // AccStats: num_frames, the prior rescaling step, the terms
      const double tot = dtot[p], newn = nf + tot;
      double ch = 0.0;
      if (iv.max_count > 0.0f) {
        const double mc = (double)iv.max_count;
        ch = (newn > mc ? newn : mc) / mc - (nf > mc ? nf : mc) / mc;
      }
      nf = newn;
      __syncthreads();                                     // (the previous chunk's solve is done with A)
      constexpr int EB = 10;
      const double *dq = dquad + p * usz;
      for (int k0 = tid; k0 < usz; k0 += 64 * NW * EB) {
        double v[EB], d[EB];
#pragma unroll
        for (int j = 0; j < EB; j++) { const int e = k0 + j * 64 * NW; v[j] = e < usz ? qu[e] : 0.0; d[j] = e < usz ? dq[e] : 0.0; }
#pragma unroll
        for (int j = 0; j < EB; j++) {
          const int e = k0 + j * 64 * NW;
          if (e >= usz) continue;
          int r = (int)((sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);
          while ((r + 1) * (r + 2) / 2 <= e) r++;
          while (r * (r + 1) / 2 > e) r--;
          const int cc = e - r * (r + 1) / 2;
          double val = v[j] + d[j];
          if (cc == r && ch != 0.0) val += ch;             // quadratic_term_.AddToDiag(prior_scale_change)
          qu[e] = val;
          A[(size_t)r * n + cc] = val;
          A[(size_t)cc * n + r] = val;
        }
      }
      double b = 0.0;
      if (mine) {
        b = li[tid] + dlin[p * n + tid];
        if (tid == 0 && ch != 0.0) b += iv.prior_offset * ch;      // linear_term_(0) += prior_offset_ * prior_scale_change
        li[tid] = b;
      }
      touched = true;
      if (nf > 0.0) x = n == 100 ? IvecCgSolve<NW, 100>(iv, A, xs, ps, xch, rb, mine, tid, n, b, x) : IvecCgSolve<NW>(iv, A, xs, ps, xch, rb, mine, tid, n, b, x);      // (its first barrier orders the expansion)
      else x = (tid == 0) ? iv.prior_offset : 0.0;
    }
    if (mine) {
      float v = (float)x;
      if (tid == 0) v = (float)((double)v - iv.prior_offset);   // (*feat)(0) -= PriorOffset() on the float copy
      ivec_out[(size_t)orow * ldo + tid] = v;
    }
  }
  if (touched) {
    if (mine) xst[sl * n + tid] = x;
    if (tid == 0) numf[sl] = nf;
  }
}

// -> false: the extractor is wider than the kernel's LDS matrix holds (the caller runs the chunks one LaunchIvecStats / LaunchIvecSolve at a time)
bool LaunchIvecChain(const IvecDev &iv, int n_utts, int K, const double *dlin, const double *dquad, const double *dtot, double *lin, double *quad,
                     double *numf, double *x, const int *slot, float *ivec_out, int ldo, const int *out_row, const int *active, hipStream_t s) {
  const int n = iv.ivec_dim;
  if (n > 128) return false;
  if (n_utts == 0 || K == 0) return true;
  const size_t smem = sizeof(double) * ((size_t)n * n + 3 * (size_t)n);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&IvecChainKernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&IvecChainKernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  if (n <= 64) hipLaunchKernelGGL(IvecChainKernel<1>, dim3(n_utts), dim3(64), smem, s, iv, n_utts, K, dlin, dquad, dtot, lin, quad, numf, x, slot, ivec_out, ldo, out_row, active);
  else hipLaunchKernelGGL(IvecChainKernel<4>, dim3(n_utts), dim3(256), smem, s, iv, n_utts, K, dlin, dquad, dtot, lin, quad, numf, x, slot, ivec_out, ldo, out_row, active);
  return true;
}

void LaunchIvecSolve(const IvecDev &iv, int n_utts, const double *linear, const double *quadratic,
                     const double *num_frames, double *x, float *ivec_out, int ldo, const int *out_row, const int *active,
                     hipStream_t s) {
  if (n_utts == 0) return;
  int n = iv.ivec_dim;
  if (n <= 128) {
    const size_t smem = sizeof(double) * ((size_t)n * n + 3 * (size_t)n);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&IvecSolveFullKernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&IvecSolveFullKernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      attr_set = true;
    }
    if (n <= 64) hipLaunchKernelGGL(IvecSolveFullKernel<1>, dim3(n_utts), dim3(64), smem, s, iv, linear, quadratic, num_frames, x, ivec_out, ldo, out_row, active);
    else      // (four waves: the two past ivec_dim only help expanding the matrix; they add exact zeros to the reductions)
      hipLaunchKernelGGL(IvecSolveFullKernel<4>, dim3(n_utts), dim3(256), smem, s, iv, linear, quadratic, num_frames, x, ivec_out, ldo, out_row, active);
    return;
  }
  int threads = 64;
  while (threads < n) threads <<= 1;
  size_t smem = sizeof(double) * ((size_t)n * (n + 1) / 2 + 5 * (size_t)n + threads);
  hipLaunchKernelGGL(IvecSolveKernel, dim3(n_utts), dim3(threads), smem, s, iv, linear, quadratic, num_frames, x, ivec_out, ldo, out_row, active);
}

}  // namespace rs
