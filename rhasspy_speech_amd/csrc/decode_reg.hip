// Register-resident token passing for small grammar graphs (the rhasspy use case: a few hundred to a few
// thousand HCLG states).
//
// Same search and same results as decode_dense.hip / decode_kernels.hip (reference:
// lattice-faster-decoder.cc:56-73,644-887).  Observation that drives the design: with a dense state table the set
// of arcs a thread has to evaluate is the same on every frame.  So thread t OWNS the states {t, t+NT, ...} and
// their incoming arcs for the whole utterance: arc records (source state, pdf, weight, arc id) are loaded into
// VGPRs once, and the log-likelihoods of exactly those arcs are fetched from HBM/L2 one frame ahead.  A frame is
// then a handful of independent LDS gathers (source-token costs) plus two block reductions instead of chains of
// dependent LDS reads: ~10x fewer cycles per frame than the LDS-graph variant on the bench graph.
// LDS holds only the dense cost array of the current frame and the packed keys of the frame under construction.
// Compiled with -ffp-contract=off.
#include "decode_common.h"

namespace rs {
using namespace dd;

#ifdef RS_DECODE_PROFILE
#define RS_T(i) do { long long _n = clock64(); if (tid == 0) prof[i] += _n - t_last; t_last = _n; } while (0)
#else
#define RS_T(i) do { } while (0)
#endif

template <int NT, int MAXS, int KE, int KX>
__global__ __launch_bounds__(NT) void RegDecodeKernel(HclgDev h, RegGraphDev rg, DecodeOptsDev o, BatchGeom g,
                                                      const float *__restrict__ loglikes, int ld, DenseWork w, int smem_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = NT / 64;
  __shared__ Red<NW> red;
  const int u = blockIdx.x, tid = threadIdx.x;
  const int T = g.d_num_frames[u], S = h.num_states;
  unsigned long long *key_next = reinterpret_cast<unsigned long long *>(smem);
  float *cost_cur = reinterpret_cast<float *>(smem + (((size_t)S * 8 + 15) & ~(size_t)15));
  int *bp = w.bp + (size_t)u * (g.max_frames + 1) * S;
  float *finfo = w.frame_info + (size_t)u * (g.max_frames + 1) * 4;
  const float INF = INFINITY;
  const size_t ll_base = (size_t)g.d_row_base[u] + g.L;
  unsigned long long n_expanded = 0, n_arcs = 0, n_insert = 0, n_alive = 0;
  int max_active_frames = 0, min_active_frames = 0;

  // ---- my states and my arcs, in registers for the whole utterance
  int4 ea[KE];         // {src state (-1 = unused), pdf + 1 | slot << 28, weight bits, forward arc index}
  int4 xa[KX];         // {src state (-1 = unused), slot, weight bits, forward arc index}
#pragma unroll
  for (int a = 0; a < KE; a++) ea[a] = rg.e_tab[(size_t)a * NT + tid];
#pragma unroll
  for (int a = 0; a < KX; a++) xa[a] = rg.x_tab[(size_t)a * NT + tid];
  float mycost[MAXS];
  unsigned long long key[MAXS];
#pragma unroll
  for (int j = 0; j < MAXS; j++) {
    mycost[j] = INF;
    key[j] = RS_EMPTY;
    const int s = j * NT + tid;
    if (s < S) { cost_cur[s] = INF; key_next[s] = RS_EMPTY; }
  }
  if (h.start % NT == tid) {
#pragma unroll
    for (int j = 0; j < MAXS; j++) if (h.start / NT == j) key[j] = PackKey(0.0f, RS_NOARC);
  }
  // log-likelihoods of my emitting arcs, fetched one frame ahead
  float ll_nxt[KE];
  if (T > 0) {
    const float *row = loglikes + ll_base * ld;
#pragma unroll
    for (int a = 0; a < KE; a++) ll_nxt[a] = ea[a].x >= 0 ? row[(ea[a].y & 0x0FFFFFFF) - 1] : 0.f;
  }
  float closure_cutoff = o.beam;
  int error = 0;
  __syncthreads();
#ifdef RS_DECODE_PROFILE
  long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t_last = clock64();
  long long n_rounds = 0;
#endif

  for (int f = -1; f < T; f++) {
    if (f >= 0) {
      float llv[KE];
#pragma unroll
      for (int a = 0; a < KE; a++) llv[a] = ll_nxt[a];
      if (f + 1 < T) {
        const float *row = loglikes + (ll_base + f + 1) * ld;
#pragma unroll
        for (int a = 0; a < KE; a++) ll_nxt[a] = ea[a].x >= 0 ? row[(ea[a].y & 0x0FFFFFFF) - 1] : 0.f;
      }
      // ---- best token, token count (my states' costs are in registers)
      float lv = INF;
      int li = 0x7fffffff, cnt = 0;
#pragma unroll
      for (int j = 0; j < MAXS; j++) {
        const float c = mycost[j];
        if (c < INF) { cnt++; if (c < lv) { lv = c; li = j * NT + tid; } }
      }
      float best_cost;
      int best_state;
      BlockMinArg<NT>(red, lv, li, &best_cost, &best_state);
      const float beam_cutoff = best_cost + o.beam;
      int c_le = 0, c_lt = 0;
#pragma unroll
      for (int j = 0; j < MAXS; j++) {
        const float c = mycost[j];
        c_le += (c <= beam_cutoff && c < INF);
        c_lt += (c < beam_cutoff);
      }
      int N;
      BlockSum3<NT>(red, cnt, c_le, c_lt, &N, &c_le, &c_lt);
      if (N == 0) { error = 1; break; }
      RS_T(0);
      // ---- GetCutoff (lattice-faster-decoder.cc:644-711)
      float cur_cutoff, adaptive_beam;
      bool decided = false;
      if (N > o.max_active && c_lt > o.max_active) {
        const float mac = KthSmallest<NT>(red, cost_cur, S, o.max_active, best_cost);
        adaptive_beam = mac - best_cost + o.beam_delta;
        cur_cutoff = mac;
        decided = true;
        max_active_frames++;
      }
      if (!decided) {
        float min_active_cutoff = INF;
        bool loosened;
        if (N > o.min_active) {
          if (o.min_active == 0 || c_le > o.min_active) min_active_cutoff = best_cost;   // tmp[min_active] <= beam_cutoff
          else min_active_cutoff = KthSmallest<NT>(red, cost_cur, S, o.min_active, best_cost);
          loosened = min_active_cutoff > beam_cutoff;
        } else {
          loosened = true;      // fewer than min_active tokens: the cutoff stays +inf (:691-705)
        }
        if (loosened) {
          adaptive_beam = min_active_cutoff - best_cost + o.beam_delta;
          cur_cutoff = min_active_cutoff;
          if (N > o.min_active) min_active_frames++;
        } else {
          adaptive_beam = o.beam;
          cur_cutoff = beam_cutoff;
        }
      }
      const float cost_offset = -best_cost;
      RS_T(1);
      // ---- ProcessEmitting: all source costs are gathered with independent LDS reads
      float csrc[KE];
#pragma unroll
      for (int a = 0; a < KE; a++) csrc[a] = ea[a].x >= 0 ? cost_cur[ea[a].x] : INF;
      float local_min = INF;
#pragma unroll
      for (int j = 0; j < MAXS; j++) {
        key[j] = RS_EMPTY;
        n_expanded += (mycost[j] < INF && mycost[j] <= cur_cutoff);
      }
#pragma unroll
      for (int a = 0; a < KE; a++) {
        const float c = csrc[a];
        if (c < INF && c <= cur_cutoff) {
          const float lk = llv[a];
          const float gc = __int_as_float(ea[a].z);
          const float tot = (c + (cost_offset - lk)) + gc;
          n_arcs++;
          if (ea[a].x == best_state) local_min = fminf(local_min, ((gc + cost_offset) - lk) + c);   // :752-757
          local_min = fminf(local_min, tot);
          const unsigned long long kk = PackKey(tot, (unsigned)ea[a].w);
          const int slot = (unsigned)ea[a].y >> 28;
#pragma unroll
          for (int j = 0; j < MAXS; j++) if (slot == j && kk < key[j]) key[j] = kk;
        }
      }
      float mn;
      int dummy;
      BlockMinArg<NT>(red, local_min, tid, &mn, &dummy);
      const float next_cutoff = mn + adaptive_beam;
      if (tid == 0) { finfo[f * 4 + 0] = cost_offset; finfo[f * 4 + 1] = cur_cutoff; finfo[f * 4 + 2] = next_cutoff; finfo[f * 4 + 3] = adaptive_beam; }
      if (next_cutoff < INF) {
#pragma unroll
        for (int j = 0; j < MAXS; j++) if (!(KeyCost(key[j]) < next_cutoff)) key[j] = RS_EMPTY;
      }
      closure_cutoff = next_cutoff;
      RS_T(2);
    }
    // ---- publish my keys, then ProcessNonemitting to the fixpoint (Jacobi rounds over my epsilon in-arcs)
#pragma unroll
    for (int j = 0; j < MAXS; j++) { const int s = j * NT + tid; if (s < S) key_next[s] = key[j]; }
    __syncthreads();
    for (int round = 0; round < 100000; round++) {
      unsigned long long ksrc[KX];
#pragma unroll
      for (int a = 0; a < KX; a++) ksrc[a] = xa[a].x >= 0 ? key_next[xa[a].x] : RS_EMPTY;
      int changed = 0;
#pragma unroll
      for (int a = 0; a < KX; a++) {
        const float c = KeyCost(ksrc[a]);
        if (c < closure_cutoff) {
          const float tot = c + __int_as_float(xa[a].z);
          if (round == 0) n_arcs++;
          if (tot < closure_cutoff) {
            const unsigned long long kk = PackKey(tot, (unsigned)xa[a].w);
#pragma unroll
            for (int j = 0; j < MAXS; j++) if (xa[a].y == j && kk < key[j]) { key[j] = kk; changed = 1; n_insert++; }
          }
        }
      }
      __syncthreads();          // everybody has read the old keys
      if (changed) {
#pragma unroll
        for (int j = 0; j < MAXS; j++) { const int s = j * NT + tid; if (s < S) key_next[s] = key[j]; }
      }
#ifdef RS_DECODE_PROFILE
      n_rounds++;
#endif
      if (!__syncthreads_or(changed)) break;
    }
    RS_T(3);
    // ---- commit frame f+1
    int *bp_row = bp + (size_t)(f + 1) * S;
#pragma unroll
    for (int j = 0; j < MAXS; j++) {
      const int s = j * NT + tid;
      if (s < S) {
        if (key[j] == RS_EMPTY) { bp_row[s] = -2; mycost[j] = INF; }
        else { bp_row[s] = (int)(unsigned)(key[j] & 0xFFFFFFFFull); mycost[j] = FromOrdered((unsigned)(key[j] >> 32)); n_alive++; }
        cost_cur[s] = mycost[j];
      }
    }
    __syncthreads();
    RS_T(4);
  }
  RS_T(5);
  FinishUtterance<NT>(red, h, g, loglikes, ld, w, cost_cur, bp, finfo, smem, smem_bytes, u, T, S, ll_base, error, n_expanded, n_arcs,
                      n_insert, n_alive, max_active_frames, min_active_frames);
#ifdef RS_DECODE_PROFILE
  RS_T(6);
  if (u == 0 && tid == 0)
    printf("reg decode cycles/frame: stats %lld cutoff %lld emit %lld closure %lld (%.2f rounds) commit %lld | finish total %lld (T=%d)\n",
           prof[0] / T, prof[1] / T, prof[2] / T, prof[3] / T, (double)n_rounds / (T + 1), prof[4] / T, prof[6], T);
#endif
}

bool LaunchDecodeReg(const HclgDev &h, const RegGraphDev &r, const DecodeOptsDev &o, const BatchGeom &g,
                     const float *loglikes, int ld, const DenseWork &w, hipStream_t s) {
  if (g.n_utts == 0) return true;
  size_t smem = (((size_t)h.num_states * 8 + 15) & ~(size_t)15) + (((size_t)h.num_states * 4 + 15) & ~(size_t)15);
  if (smem < 48 * 1024) smem = 48 * 1024;      // room to stage back-pointer rows for the traceback
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&RegDecodeKernel<256, 4, 16, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&RegDecodeKernel<1024, 4, 8, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  if (r.nt == 256)
    hipLaunchKernelGGL((RegDecodeKernel<256, 4, 16, 16>), dim3(g.n_utts), dim3(256), smem, s, h, r, o, g, loglikes, ld, w, (int)smem);
  else if (r.nt == 1024)
    hipLaunchKernelGGL((RegDecodeKernel<1024, 4, 8, 8>), dim3(g.n_utts), dim3(1024), smem, s, h, r, o, g, loglikes, ld, w, (int)smem);
  else
    return false;
  return true;
}

}  // namespace rs
