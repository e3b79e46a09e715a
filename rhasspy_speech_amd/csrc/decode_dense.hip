// Dense LDS-resident token passing for small/medium decoding graphs (grammar HCLGs: the rhasspy use case).
//
// Same search as decode_kernels.hip (reference: lattice-faster-decoder.cc:56-73,644-887), different data flow:
// the frame's tokens are a dense cost array over HCLG states held in LDS; each thread owns destination states
// and PULLS over their incoming arcs (reverse graph), so recombination needs no atomics, no hash and no token
// lists, and one frame costs a handful of LDS passes.  Costs and back-pointers are bit-identical to the
// sparse kernel: both take the minimum over incoming arcs of the packed key (order-preserving bits of
// (cur_cost + (cost_offset - loglike)) + graph_cost, forward arc index).
// Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include "env.h"

#include "decode_common.h"

namespace rs {

#ifdef RS_DECODE_PROFILE
#define RS_T(i) do { long long _n = clock64(); if (tid == 0) prof[i] += _n - t_last; t_last = _n; } while (0)
#else
#define RS_T(i) do { } while (0)
#endif

using namespace dd;

static inline size_t Al16(size_t x) { return (x + 15) & ~(size_t)15; }
size_t DenseDecodeSmemBytes(int S, int P) {
  // cost_cur (f32) + key_next (u64) per state, one log-likelihood row, 16-byte aligned carve-outs
  return Al16((size_t)S * 8) + Al16((size_t)S * 4) + Al16((size_t)P * 4);
}
// bytes of the reverse graph when it is cached in LDS as well
static size_t RevGraphSmemBytes(int S, int n_e, int n_x, int n_eps_dst) {
  return 2 * Al16((size_t)(S + 1) * 4) + Al16((size_t)n_e * 16) + Al16((size_t)n_x * 16) + Al16((size_t)n_eps_dst * 4);
}
static const size_t kDenseSmemBudget = 144 * 1024;
bool DenseDecodeFits(int S, int P) { return DenseDecodeSmemBytes(S, P) + sizeof(Red<4>) + 1024 <= kDenseSmemBudget; }

// NT = 256: one workgroup of four waves per utterance (larger graphs).  NT = 64: ONE WAVEFRONT per utterance --
// every __syncthreads() below is then elided by the compiler (single-wave workgroup) and all reductions are
// pure wave shuffles, which is what makes small grammar graphs run at ~2 us per frame.
template <int NT, bool GRAPH_IN_LDS>
__global__ __launch_bounds__(NT) void DenseDecodeKernel(HclgDev h, RevGraphDev rgg, DecodeOptsDev o, BatchGeom g,
                                                        const float *__restrict__ loglikes, int ld, int P, DenseWork w, int smem_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = NT / 64;
  __shared__ Red<NW> red;
  const int u = blockIdx.x, tid = threadIdx.x;
  const int T = g.d_num_frames[u], S = h.num_states;
  unsigned long long *key_next = reinterpret_cast<unsigned long long *>(smem);
  float *cost_cur = reinterpret_cast<float *>(smem + (((size_t)S * 8 + 15) & ~(size_t)15));
  float *llr = reinterpret_cast<float *>(smem + (((size_t)S * 8 + 15) & ~(size_t)15) + (((size_t)S * 4 + 15) & ~(size_t)15));
  // reverse graph: cached in LDS when it fits (every frame re-reads it several times), else read through L1/L2
  const int n_e = rgg.in_begin_e_host_total, n_x = rgg.in_begin_x_host_total;
  unsigned char *gp = smem + (((size_t)S * 8 + 15) & ~(size_t)15) + (((size_t)S * 4 + 15) & ~(size_t)15) + (((size_t)P * 4 + 15) & ~(size_t)15);
  uint32_t *l_be = reinterpret_cast<uint32_t *>(gp);
  uint32_t *l_bx = reinterpret_cast<uint32_t *>(gp + (((size_t)(S + 1) * 4 + 15) & ~(size_t)15));
  int4 *l_ie = reinterpret_cast<int4 *>(gp + 2 * (((size_t)(S + 1) * 4 + 15) & ~(size_t)15));
  int4 *l_ix = reinterpret_cast<int4 *>(reinterpret_cast<unsigned char *>(l_ie) + (((size_t)n_e * 16 + 15) & ~(size_t)15));
  int *l_ed = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(l_ix) + (((size_t)n_x * 16 + 15) & ~(size_t)15));
  if (GRAPH_IN_LDS) {
    for (int i = tid; i <= S; i += NT) { l_be[i] = rgg.in_begin_e[i]; l_bx[i] = rgg.in_begin_x[i]; }
    for (int i = tid; i < n_e; i += NT) l_ie[i] = rgg.in_e[i];
    for (int i = tid; i < n_x; i += NT) l_ix[i] = rgg.in_x[i];
    for (int i = tid; i < rgg.num_eps_dst; i += NT) l_ed[i] = rgg.eps_dst[i];
  }
  struct { const uint32_t *in_begin_e, *in_begin_x; const int4 *in_e, *in_x; const int *eps_dst; int num_eps_dst; } rg;
  rg.in_begin_e = GRAPH_IN_LDS ? l_be : rgg.in_begin_e;
  rg.in_begin_x = GRAPH_IN_LDS ? l_bx : rgg.in_begin_x;
  rg.in_e = GRAPH_IN_LDS ? l_ie : rgg.in_e;
  rg.in_x = GRAPH_IN_LDS ? l_ix : rgg.in_x;
  rg.eps_dst = GRAPH_IN_LDS ? l_ed : rgg.eps_dst;
  rg.num_eps_dst = rgg.num_eps_dst;
  int *bp = w.bp + (size_t)u * (g.max_frames + 1) * S;
  float *finfo = w.frame_info + (size_t)u * (g.max_frames + 1) * 4;
  const float INF = INFINITY;
  const size_t ll_base = (size_t)g.d_row_base[u] + g.L;
  unsigned long long n_expanded = 0, n_arcs = 0, n_insert = 0, n_alive = 0;
  int max_active_frames = 0, min_active_frames = 0;

  for (int s = tid; s < S; s += NT) { cost_cur[s] = INF; key_next[s] = RS_EMPTY; }
  __syncthreads();
  if (tid == 0) key_next[h.start] = PackKey(0.0f, RS_NOARC);
  __syncthreads();
  float closure_cutoff = o.beam;
  int error = 0;
#ifdef RS_DECODE_PROFILE
  long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t_last = clock64();
#endif
  // log-likelihood rows are prefetched into registers one frame ahead (P <= PF * NT), so the global-load latency
  // is hidden behind the previous frame's LDS work
  constexpr int PF = NT == 64 ? 32 : 16;
  const bool pf_ok = P <= PF * NT;
  float pf[PF];
  if (pf_ok && T > 0) {
    const float *src = loglikes + ll_base * ld;
#pragma unroll
    for (int k = 0; k < PF; k++) { const int i = tid + k * NT; pf[k] = i < P ? src[i] : 0.f; }
  }

  for (int f = -1; f < T; f++) {
    if (f >= 0) {
      // ---- this frame's log-likelihood row -> LDS
      if (pf_ok) {
#pragma unroll
        for (int k = 0; k < PF; k++) { const int i = tid + k * NT; if (i < P) llr[i] = pf[k]; }
        if (f + 1 < T) {
          const float *src = loglikes + (ll_base + f + 1) * ld;
#pragma unroll
          for (int k = 0; k < PF; k++) { const int i = tid + k * NT; pf[k] = i < P ? src[i] : 0.f; }
        }
      } else {
        const float *src = loglikes + (ll_base + f) * ld;
        for (int i = tid; i < P; i += NT) llr[i] = src[i];
      }
      // ---- best token and token count
      float lv = INF;
      int li = 0x7fffffff, cnt = 0;
      for (int s = tid; s < S; s += NT) {
        const float c = cost_cur[s];
        if (c < INF) { cnt++; if (c < lv || (c == lv && s < li)) { lv = c; li = s; } }
      }
      float best_cost;
      int best_state;
      BlockMinArg<NT>(red, lv, li, &best_cost, &best_state);
      RS_T(0);
      const float beam_cutoff = best_cost + o.beam;
      int c_le = 0, c_lt = 0;
      for (int s = tid; s < S; s += NT) {
        const float c = cost_cur[s];
        c_le += (c <= beam_cutoff && c < INF);
        c_lt += (c < beam_cutoff);
      }
      int N;
      BlockSum3<NT>(red, cnt, c_le, c_lt, &N, &c_le, &c_lt);
      if (N == 0) { error = 1; break; }
      // ---- GetCutoff (lattice-faster-decoder.cc:644-711); the k-th smallest is only materialised when it binds
      float cur_cutoff, adaptive_beam;
      bool decided = false;
      if (N > o.max_active && c_lt > o.max_active) {
        // tmp[max_active] < beam_cutoff  <=>  more than max_active costs are below the beam cutoff
        const float mac = KthSmallest<NT>(red, cost_cur, S, o.max_active, best_cost);
        adaptive_beam = mac - best_cost + o.beam_delta;
        cur_cutoff = mac;
        decided = true;
        max_active_frames++;
      }
      if (!decided) {
        float min_active_cutoff = INF;
        bool loosened = false;
        if (N > o.min_active) {
          if (o.min_active == 0) min_active_cutoff = best_cost;
          else if (c_le > o.min_active) min_active_cutoff = best_cost;   // placeholder: tmp[min_active] <= beam_cutoff, not binding
          else min_active_cutoff = KthSmallest<NT>(red, cost_cur, S, o.min_active, best_cost);
          loosened = min_active_cutoff > beam_cutoff;
        } else {
          loosened = true;     // fewer than min_active tokens: cutoff stays +inf (:691-705)
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
      // ---- ProcessEmitting, pull form: every destination state takes the min over its incoming emitting arcs
      float local_min = INF;
      for (int s = tid; s < S; s += NT) {
        unsigned long long key = RS_EMPTY;
        const unsigned b = rg.in_begin_e[s], e = rg.in_begin_e[s + 1];
        for (unsigned k = b; k < e; k++) {
          const int4 a = rg.in_e[k];
          const float c = cost_cur[a.x];
          if (!(c < INF) || !(c <= cur_cutoff)) continue;
          const float lk = llr[a.y - 1];
          const float gc = __int_as_float(a.z);
          const float tot = (c + (cost_offset - lk)) + gc;
          n_arcs++;
          if (a.x == best_state) local_min = fminf(local_min, ((gc + cost_offset) - lk) + c);   // :752-757
          local_min = fminf(local_min, tot);
          const unsigned long long kk = PackKey(tot, (unsigned)a.w);
          if (kk < key) key = kk;
        }
        key_next[s] = key;
        const float c = cost_cur[s];
        n_expanded += (c < INF && c <= cur_cutoff);
      }
      float mn;
      int dummy;
      BlockMinArg<NT>(red, local_min, tid, &mn, &dummy);
      const float next_cutoff = mn + adaptive_beam;
      RS_T(2);
      if (tid == 0) { finfo[f * 4 + 0] = cost_offset; finfo[f * 4 + 1] = cur_cutoff; finfo[f * 4 + 2] = next_cutoff; finfo[f * 4 + 3] = adaptive_beam; }
      if (next_cutoff < INF) {
        for (int s = tid; s < S; s += NT) if (!(KeyCost(key_next[s]) < next_cutoff)) key_next[s] = RS_EMPTY;
      }
      closure_cutoff = next_cutoff;
      __syncthreads();
      RS_T(3);
    }
    // ---- ProcessNonemitting, pull form, to the fixpoint
    for (int round = 0; round < 100000; round++) {
      int changed = 0;
      for (int q = tid; q < rg.num_eps_dst; q += NT) {
        const int s = rg.eps_dst[q];
        unsigned long long key = key_next[s];
        const unsigned long long key0 = key;
        const unsigned b = rg.in_begin_x[s], e = rg.in_begin_x[s + 1];
        for (unsigned k = b; k < e; k++) {
          const int4 a = rg.in_x[k];
          const float c = KeyCost(key_next[a.x]);
          if (!(c < closure_cutoff)) continue;
          const float tot = c + __int_as_float(a.z);
          if (round == 0) n_arcs++;
          if (!(tot < closure_cutoff)) continue;
          const unsigned long long kk = PackKey(tot, (unsigned)a.w);
          if (kk < key) key = kk;
        }
        if (key < key0) { key_next[s] = key; changed = 1; n_insert++; }
      }
      if (tid == 0) red.changed = 0;
      __syncthreads();
      if (changed) red.changed = 1;
      __syncthreads();
      const int any = red.changed;
      __syncthreads();
      if (!any) break;
    }
    RS_T(4);
    // ---- commit frame f+1: back-pointer row to HBM, costs become the current frame
    int *bp_row = bp + (size_t)(f + 1) * S;
    for (int s = tid; s < S; s += NT) {
      const unsigned long long key = key_next[s];
      if (key == RS_EMPTY) { bp_row[s] = -2; cost_cur[s] = INF; }
      else { bp_row[s] = (int)(unsigned)(key & 0xFFFFFFFFull); cost_cur[s] = FromOrdered((unsigned)(key >> 32)); n_alive++; }
      key_next[s] = RS_EMPTY;
    }
    __syncthreads();
    RS_T(5);
  }
  FinishUtterance<NT>(red, h, g, loglikes, ld, w, cost_cur, bp, finfo, smem, smem_bytes, u, T, S, ll_base, error, n_expanded, n_arcs,
                      n_insert, n_alive, max_active_frames, min_active_frames, (size_t)u);
#ifdef RS_DECODE_PROFILE
  RS_T(6);
  if (u == 0 && tid == 0)
    printf("dense decode cycles/frame: stats %lld cutoff %lld pull %lld filter %lld closure %lld commit %lld | finish total %lld (T=%d)\n",
           prof[0] / T, prof[1] / T, prof[2] / T, prof[3] / T, prof[4] / T, prof[5] / T, prof[6], T);
#endif
}

void LaunchDecodeDense(const HclgDev &h, const RevGraphDev &r, const DecodeOptsDev &o, const BatchGeom &g,
                       const float *loglikes, int ld, int num_pdfs, const DenseWork &w, hipStream_t s) {
  if (g.n_utts == 0) return;
  size_t smem = DenseDecodeSmemBytes(h.num_states, num_pdfs);
  const size_t with_graph = smem + RevGraphSmemBytes(h.num_states, r.in_begin_e_host_total, r.in_begin_x_host_total, r.num_eps_dst);
  const bool graph_in_lds = with_graph + sizeof(Red<4>) + 1024 <= kDenseSmemBudget;
  if (graph_in_lds) smem = with_graph;
  // RS_DENSE_NT selects the workgroup size per utterance (64 / 256 / 1024); measured on MI355X (625-state grammar graph,
  // 298 frames): 64 -> 18 us/frame, 256 -> 10 us/frame: the frame is a chain of dependent LDS reads, more lanes hide more.
  static int nt_env = [] { const char *e = TuneEnv("RS_DENSE_NT"); return e ? std::atoi(e) : 256; }();
  const bool one_wave = nt_env == 64 && h.num_states <= 4096 && num_pdfs <= 2048;
  const bool big = nt_env == 1024;
  const size_t stage_min = one_wave ? 40 * 1024 : 64 * 1024;     // room to stage back-pointer rows for the traceback
  if (smem < stage_min) smem = stage_min;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&DenseDecodeKernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&DenseDecodeKernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&DenseDecodeKernel<1024, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&DenseDecodeKernel<1024, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&DenseDecodeKernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&DenseDecodeKernel<64, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
#define RS_DD(NTV, GV) hipLaunchKernelGGL((DenseDecodeKernel<NTV, GV>), dim3(g.n_utts), dim3(NTV), smem, s, h, r, o, g, loglikes, ld, num_pdfs, w, (int)smem)
  if (one_wave) { if (graph_in_lds) RS_DD(64, true); else RS_DD(64, false); }
  else if (big) { if (graph_in_lds) RS_DD(1024, true); else RS_DD(1024, false); }
  else { if (graph_in_lds) RS_DD(256, true); else RS_DD(256, false); }
#undef RS_DD
}

}  // namespace rs
