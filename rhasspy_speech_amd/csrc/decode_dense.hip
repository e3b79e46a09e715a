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

#include "kernels.h"

namespace rs {

#define RS_EMPTY 0xFFFFFFFFFFFFFFFFull
#define RS_NOARC 0xFFFFFFFFu

namespace {
__device__ __forceinline__ unsigned OrderedBits(float f) {
  unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float FromOrdered(unsigned u) {
  unsigned b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  return __uint_as_float(b);
}
__device__ __forceinline__ unsigned long long PackKey(float cost, unsigned arc) {
  return ((unsigned long long)OrderedBits(cost) << 32) | arc;
}
__device__ __forceinline__ float KeyCost(unsigned long long k) {
  return k == RS_EMPTY ? INFINITY : FromOrdered((unsigned)(k >> 32));
}

constexpr int NT = 256;
constexpr int NW = NT / 64;

struct Red {
  float f[NW];
  int i[NW];
  int c0[NW], c1[NW], c2[NW];
  float bf[2];
  int bi[4];
  int changed;
  unsigned hist[256];
  unsigned long long ctr[8];
  double dsum[2 * NW];
};

__device__ __forceinline__ void BlockMinArg(Red &r, float v, int idx, float *ov, int *oi) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float xv = __shfl_xor(v, o, 64);
    int xi = __shfl_xor(idx, o, 64);
    if (xv < v || (xv == v && xi < idx)) { v = xv; idx = xi; }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { r.f[w] = v; r.i[w] = idx; }
  __syncthreads();
  float bv = r.f[0];
  int bi = r.i[0];
#pragma unroll
  for (int k = 1; k < NW; k++) if (r.f[k] < bv || (r.f[k] == bv && r.i[k] < bi)) { bv = r.f[k]; bi = r.i[k]; }
  *ov = bv;
  *oi = bi;
  __syncthreads();
}

__device__ __forceinline__ void BlockSum3(Red &r, int a, int b, int c, int *oa, int *ob, int *oc) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { r.c0[w] = a; r.c1[w] = b; r.c2[w] = c; }
  __syncthreads();
  int sa = 0, sb = 0, sc = 0;
#pragma unroll
  for (int k = 0; k < NW; k++) { sa += r.c0[k]; sb += r.c1[k]; sc += r.c2[k]; }
  *oa = sa; *ob = sb; *oc = sc;
  __syncthreads();
}

// exact k-th smallest (0-based) of the finite entries of cost[0..S): 4-pass radix select on the order-preserving
// bit pattern; the 256-bin prefix scan of each pass is done by one wavefront with shuffles (4 bins per lane).
__device__ float KthSmallest(Red &r, const float *cost, int S, int k) {
  unsigned prefix = 0, mask = 0;
  int kk = k;
  for (int pass = 0; pass < 4; pass++) {
    const int shift = 24 - 8 * pass;
    if (threadIdx.x < 256) r.hist[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < S; i += NT) {
      const float c = cost[i];
      if (c < INFINITY) {
        unsigned u = OrderedBits(c);
        if ((u & mask) == prefix) atomicAdd(&r.hist[(u >> shift) & 255u], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int l = threadIdx.x;
      const int h0 = (int)r.hist[4 * l], h1 = (int)r.hist[4 * l + 1], h2 = (int)r.hist[4 * l + 2], h3 = (int)r.hist[4 * l + 3];
      const int tot = h0 + h1 + h2 + h3;
      int inc = tot;                       // inclusive scan over lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o, 64); if (l >= o) inc += v; }
      const int exc = inc - tot;
      // the lane whose 4 bins contain rank kk: exc <= kk < inc
      if (exc <= kk && kk < inc) {
        int acc = exc, b = 4 * l;
        if (acc + h0 <= kk) { acc += h0; b++; if (acc + h1 <= kk) { acc += h1; b++; if (acc + h2 <= kk) { acc += h2; b++; } } }
        r.bi[1] = b;
        r.bi[2] = kk - acc;
      }
    }
    __syncthreads();
    prefix |= ((unsigned)r.bi[1]) << shift;
    mask |= 255u << shift;
    kk = r.bi[2];
  }
  __syncthreads();
  return FromOrdered(prefix);
}
}  // namespace

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
bool DenseDecodeFits(int S, int P) { return DenseDecodeSmemBytes(S, P) + sizeof(Red) + 1024 <= kDenseSmemBudget; }

template <bool GRAPH_IN_LDS>
__global__ __launch_bounds__(NT) void DenseDecodeKernel(HclgDev h, RevGraphDev rgg, DecodeOptsDev o, BatchGeom g,
                                                        const float *__restrict__ loglikes, int ld, int P, DenseWork w, int smem_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ Red red;
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
  // log-likelihood rows are prefetched into registers one frame ahead (P <= 16 * NT), so the global-load latency
  // is hidden behind the previous frame's LDS work
  constexpr int PF = 16;
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
      BlockMinArg(red, lv, li, &best_cost, &best_state);
      const float beam_cutoff = best_cost + o.beam;
      int c_le = 0, c_lt = 0;
      for (int s = tid; s < S; s += NT) {
        const float c = cost_cur[s];
        c_le += (c <= beam_cutoff && c < INF);
        c_lt += (c < beam_cutoff);
      }
      int N;
      BlockSum3(red, cnt, c_le, c_lt, &N, &c_le, &c_lt);
      if (N == 0) { error = 1; break; }
      // ---- GetCutoff (lattice-faster-decoder.cc:644-711); the k-th smallest is only materialised when it binds
      float cur_cutoff, adaptive_beam;
      bool decided = false;
      if (N > o.max_active && c_lt > o.max_active) {
        // tmp[max_active] < beam_cutoff  <=>  more than max_active costs are below the beam cutoff
        const float mac = KthSmallest(red, cost_cur, S, o.max_active);
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
          else min_active_cutoff = KthSmallest(red, cost_cur, S, o.min_active);
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
      BlockMinArg(red, local_min, tid, &mn, &dummy);
      const float next_cutoff = mn + adaptive_beam;
      if (tid == 0) { finfo[f * 4 + 0] = cost_offset; finfo[f * 4 + 1] = cur_cutoff; finfo[f * 4 + 2] = next_cutoff; finfo[f * 4 + 3] = adaptive_beam; }
      if (next_cutoff < INF) {
        for (int s = tid; s < S; s += NT) if (!(KeyCost(key_next[s]) < next_cutoff)) key_next[s] = RS_EMPTY;
      }
      closure_cutoff = next_cutoff;
      __syncthreads();
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
    // ---- commit frame f+1: back-pointer row to HBM, costs become the current frame
    int *bp_row = bp + (size_t)(f + 1) * S;
    for (int s = tid; s < S; s += NT) {
      const unsigned long long key = key_next[s];
      if (key == RS_EMPTY) { bp_row[s] = -2; cost_cur[s] = INF; }
      else { bp_row[s] = (int)(unsigned)(key & 0xFFFFFFFFull); cost_cur[s] = FromOrdered((unsigned)(key >> 32)); n_alive++; }
      key_next[s] = RS_EMPTY;
    }
    __syncthreads();
  }
  // ---- final costs, best final token (ComputeFinalCosts), traceback
  float lv1 = INF, lv2 = INF;
  int li1 = 0x7fffffff, li2 = 0x7fffffff;
  if (!error)
    for (int s = tid; s < S; s += NT) {
      const float c = cost_cur[s];
      if (!(c < INF)) continue;
      const float wf = c + h.final_cost[s];
      if (wf < lv1 || (wf == lv1 && s < li1)) { lv1 = wf; li1 = s; }
      if (c < lv2 || (c == lv2 && s < li2)) { lv2 = c; li2 = s; }
    }
  float b1, b2;
  int i1, i2;
  BlockMinArg(red, lv1, li1, &b1, &i1);
  BlockMinArg(red, lv2, li2, &b2, &i2);
  // counters: block sums via LDS atomics on the histogram scratch (reused)
  for (int i = tid; i < 8; i += NT) red.ctr[i] = 0;
  __syncthreads();
  unsigned long long *ctr = red.ctr;
  atomicAdd(&ctr[0], n_expanded);
  atomicAdd(&ctr[1], n_arcs);
  atomicAdd(&ctr[2], n_insert);
  atomicAdd(&ctr[3], n_alive);
  __syncthreads();
  const bool reached = b1 < INF;
  const bool ok = !error && T > 0 && b2 < INF;
  // ---- traceback (GetBestPath).  The back-pointer rows are staged through LDS a block of frames at a time so
  // that the inherently sequential walk runs at LDS latency; arc sources come from LDS too when they fit.
  int *stage = reinterpret_cast<int *>(smem);
  const int stage_ints = smem_bytes / 4;
  const bool src_in_lds = (h.num_arcs + S) <= stage_ints / 2 && h.num_arcs > 0;
  int *lds_src = stage;                                   // [num_arcs]
  int *rows = src_in_lds ? stage + h.num_arcs : stage;    // staged back-pointer rows
  const int rows_cap = (stage_ints - (src_in_lds ? h.num_arcs : 0)) / S;
  int *path = w.path + (size_t)u * w.path_cap * 2;        // (arc, source frame) pairs, last arc first
  int path_len = 0;
  __syncthreads();
  if (ok) {
    if (src_in_lds) for (int i = tid; i < h.num_arcs; i += NT) lds_src[i] = h.arc_src[i];
    int F = T, st = reached ? i1 : i2;
    bool done = false;
    while (!done) {
      const int lo = F - rows_cap + 1 > 0 ? F - rows_cap + 1 : 0;     // stage rows lo..F
      const int nrow = F - lo + 1;
      __syncthreads();
      for (int i = tid; i < nrow * S; i += NT) rows[i] = bp[(size_t)lo * S + i];
      __syncthreads();
      if (tid == 0) {
        while (true) {
          const int arc = rows[(F - lo) * S + st];
          if (arc < 0) { done = true; break; }
          // an arc is emitting iff it is not among the first num_ieps arcs of its source state
          const int src = src_in_lds ? lds_src[arc] : h.arc_src[arc];
          const bool emitting = (unsigned)arc >= h.arc_begin[src] + h.num_ieps[src];
          const int Fs = emitting ? F - 1 : F;
          if (path_len < w.path_cap) { path[2 * path_len] = arc; path[2 * path_len + 1] = Fs; }
          path_len++;
          st = src;
          F = Fs;
          if (F < lo) break;       // need older rows
        }
        red.bi[0] = done ? 1 : 0; red.bi[1] = F; red.bi[2] = st; red.bi[3] = path_len;
      }
      __syncthreads();
      done = red.bi[0] != 0; F = red.bi[1]; st = red.bi[2]; path_len = red.bi[3];
    }
  }
  __syncthreads();
  // ---- path costs and words, in parallel over the path
  const bool path_ok = ok && path_len <= w.path_cap;
  double pg = 0.0, pa = 0.0;
  if (path_ok) {
    for (int i = tid; i < path_len; i += NT) {
      const int arc = path[2 * i], Fs = path[2 * i + 1];
      const int4 a = h.arcs[arc];
      pg += (double)__int_as_float(a.z);
      if (a.x != 0) {
        const float off = finfo[Fs * 4 + 0];
        const float lk = loglikes[(ll_base + Fs) * ld + (a.x - 1)];
        const float link_ac = off - lk;             // ForwardLink::acoustic_cost
        pa += (double)(link_ac - off);              // GetRawLattice :166-172
      }
      path[2 * i + 1] = a.y;                        // olabel replaces the frame
    }
  }
  // deterministic block sums in double
  double *dsum = red.dsum;
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) { pg += __shfl_xor(pg, o2, 64); pa += __shfl_xor(pa, o2, 64); }
  if ((tid & 63) == 0) { dsum[(tid >> 6) * 2] = pg; dsum[(tid >> 6) * 2 + 1] = pa; }
  __syncthreads();
  if (tid < 64) {
    // wave 0: ordered compaction of the word labels (path is stored last-arc-first)
    int *words = w.out_words + (size_t)u * w.max_words;
    int nw = 0;
    bool truncated = false;
    if (path_ok) {
      for (int base = path_len - 1; base >= 0; base -= 64) {
        const int i = base - tid;
        const int wl = (i >= 0) ? path[2 * i + 1] : 0;
        const unsigned long long m = __ballot(wl != 0);
        if (wl != 0) {
          const int pos = nw + __popcll(m & ((1ull << tid) - 1ull));
          if (pos < w.max_words) words[pos] = wl; else truncated = true;
        }
        nw += __popcll(m);
      }
    }
    truncated = __any(truncated) || nw > w.max_words;
    if (tid == 0) {
      double graph = 0.0, ac = 0.0;
      for (int k = 0; k < NW; k++) { graph += dsum[k * 2]; ac += dsum[k * 2 + 1]; }
      if (ok && reached) graph += (double)h.final_cost[i1];
      w.out_nwords[u] = (!path_ok || truncated) ? -1 : nw;
      float *oc = w.out_costs + (size_t)u * 4;
      oc[0] = (float)graph; oc[1] = (float)ac; oc[2] = reached ? b1 : b2; oc[3] = reached ? 1.f : 0.f;
      long long *c8 = w.counters + (size_t)u * 8;
      for (int i = 0; i < 4; i++) c8[i] = (long long)ctr[i];
      c8[4] = 0; c8[5] = max_active_frames; c8[6] = min_active_frames; c8[7] = error ? 2 : 0;
    }
  }
}

void LaunchDecodeDense(const HclgDev &h, const RevGraphDev &r, const DecodeOptsDev &o, const BatchGeom &g,
                       const float *loglikes, int ld, int num_pdfs, const DenseWork &w, hipStream_t s) {
  if (g.n_utts == 0) return;
  size_t smem = DenseDecodeSmemBytes(h.num_states, num_pdfs);
  const size_t with_graph = smem + RevGraphSmemBytes(h.num_states, r.in_begin_e_host_total, r.in_begin_x_host_total, r.num_eps_dst);
  const bool graph_in_lds = with_graph + sizeof(Red) + 1024 <= kDenseSmemBudget;
  if (graph_in_lds) smem = with_graph;
  if (smem < 64 * 1024) smem = 64 * 1024;     // room to stage back-pointer rows for the traceback
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&DenseDecodeKernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&DenseDecodeKernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  if (graph_in_lds)
    hipLaunchKernelGGL(DenseDecodeKernel<true>, dim3(g.n_utts), dim3(NT), smem, s, h, r, o, g, loglikes, ld, num_pdfs, w, (int)smem);
  else
    hipLaunchKernelGGL(DenseDecodeKernel<false>, dim3(g.n_utts), dim3(NT), smem, s, h, r, o, g, loglikes, ld, num_pdfs, w, (int)smem);
}

}  // namespace rs
