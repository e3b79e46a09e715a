// Token-passing beam search over the HCLG on gfx950: one workgroup per utterance, persistent over all
// frames (no per-frame launches), every wavefront expanding tokens in parallel.
//
// Reference semantics reproduced (kaldi/src/decoder/lattice-faster-decoder.cc):
//   InitDecoding :56-73, GetCutoff :644-711, ProcessEmitting :714-804, ProcessNonemitting :820-887,
//   FindOrAddToken :253-293, ComputeFinalCosts :536-577, best-path traceback
//   (lattice-faster-online-decoder.cc:56-173 / GetBestPath :95-102).
// Design (not the reference's): the reference keeps a hash of heap-allocated tokens and walks it
// sequentially; here the "hash" is a dense per-utterance table best[state] holding a packed 64-bit key
// (order-preserving cost bits << 32 | arc index) that every lane updates with atomicMin, so recombination
// is order-independent: a token's cost is the minimum over all incoming arcs of the reference's float
// expression (cur_cost + (cost_offset - loglike)) + graph_cost, and exact ties go to the lowest arc index.
// Pruning uses the *final* value of the reference's running next_cutoff: every
// token the reference is guaranteed to create is created, order-dependent extras (>= best + beam) are not.
// This file is compiled with -ffp-contract=off (cost expressions must not be fused).
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdlib>

#include "kernels.h"

#include "decode_tok.h"

namespace rs {
using namespace tok;

// Relax one arc into the frame under construction (FindOrAddToken).  Returns true if the table entry improved.
template <int NT>
__device__ __forceinline__ bool Relax(BlockCtx<NT> &c, unsigned long long *best, int *map_next, int4 *next_toks,
                                      int next_cap, int ns, float tot, unsigned arc) {
  const unsigned long long key = PackKey(tot, arc);
  const unsigned long long old = atomicMin(&best[ns], key);
  if (old == RS_EMPTY) {
    int idx = atomicAdd(&c.n_next, 1);
    if (idx < next_cap) {
      next_toks[idx] = make_int4(ns, 0, -1, -2);      // .w = -2: not made by an emitting winner, the materialise pass fills it in
      map_next[ns] = idx;
    } else {
      c.overflow = 1;
    }
    return true;
  }
  return key < old;
}

template <int NT>
__global__ __launch_bounds__(NT) void DecodeKernel(HclgDev h, DecodeOptsDev o, BatchGeom g,
                                                   const float *__restrict__ loglikes, int ld, DecodeWork w) {
  __shared__ BlockCtx<NT> c;
  const int u = blockIdx.x, tid = threadIdx.x;
  if (w.redo && !w.redo[u]) return;          // decoded by LiveDecodeKernel (decode_live.hip) already
  const int T = g.d_num_frames[u];
  const int S = h.num_states;
  unsigned long long *best = w.best + (size_t)u * S;
  // state -> index of its token in the frame under construction.  Never cleared: it is only read for states that own a token of
  // that frame (the source of a winning epsilon arc), or -- candidate list overflowed -- of the frame just finished.
  int *map_next = w.map_a + (size_t)u * S;
  int *cand_t = w.map_b + (size_t)u * S;          // source token of each candidate of the arc loop
  int4 *tokens = w.tokens + (size_t)u * w.tok_cap;
  int *frame_off = w.frame_tok_off + (size_t)u * (g.max_frames + 2);
  float *finfo = w.frame_info + (size_t)u * (g.max_frames + 1) * 4;
  int *queue[2] = {w.queue_a + (size_t)u * S, w.queue_b + (size_t)u * S};
  int *stamp = w.in_queue + (size_t)u * S;         // closure round in which the state was last pushed (no per-round reset pass)
  int round_id = 0;
  const float INF = INFINITY;
  const size_t ll_base = (size_t)g.d_row_base[u] + g.L;

  for (int i = tid; i < S; i += NT) { StoreKey(&best[i], RS_EMPTY); stamp[i] = 0; }
  if (tid == 0) {
    c.n_next = 0; c.overflow = 0; c.error = 0; c.q_n[0] = c.q_n[1] = 0;
    for (int i = 0; i < 8; i++) c.counters[i] = 0;
    w.out_nwords[u] = 0;
  }
  unsigned long long cnt_expanded = 0, cnt_arcs = 0, cnt_insert = 0;
  __syncthreads();

  int off_cur = 0, n_cur = 0;       // frame f's token list
  int off_next = 0;                 // frame under construction
  if (tid == 0) {
    Relax(c, best, map_next, tokens, w.tok_cap, h.start, 0.0f, RS_NOARC);
    frame_off[0] = 0;
  }
  __syncthreads();
  float closure_cutoff = o.beam;    // InitDecoding: ProcessNonemitting(config_.beam)
#ifdef RS_DECODE_PROFILE
  long long prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long t_last = clock64();
#endif

  for (int f = -1; f < T; f++) {
    int4 *next_toks = tokens + off_next;
    const int next_cap = w.tok_cap - off_next;
    if (f >= 0) {
      // ================================================================ ProcessEmitting(frame f)
      const int4 *cur = tokens + off_cur;
      float lv = INF;
      int li = 0x7fffffff;
      for (int i = tid; i < n_cur; i += NT) {
        float cst = __int_as_float(cur[i].y);
        if (cst < lv || (cst == lv && i < li)) { lv = cst; li = i; }
      }
      float best_cost;
      int best_idx;
      BlockMinArg<NT>(c, lv, li, &best_cost, &best_idx);
      RS_TP(0);
      // ---- GetCutoff
      const float beam_cutoff = best_cost + o.beam;
      float max_active_cutoff = INF, min_active_cutoff = INF, cur_cutoff, adaptive_beam;
      bool decided = false;
      if (n_cur > o.max_active) max_active_cutoff = BlockKthSmallest<NT>(c, cur, n_cur, o.max_active, best_cost);
      if (max_active_cutoff < beam_cutoff) {
        adaptive_beam = max_active_cutoff - best_cost + o.beam_delta;
        cur_cutoff = max_active_cutoff;
        decided = true;
        if (tid == 0) c.counters[5]++;
      }
      if (!decided) {
        if (n_cur > o.min_active) {
          if (o.min_active == 0) min_active_cutoff = best_cost;
          else min_active_cutoff = BlockKthSmallest<NT>(c, cur, n_cur, o.min_active, best_cost);
        }
        if (min_active_cutoff > beam_cutoff) {
          adaptive_beam = min_active_cutoff - best_cost + o.beam_delta;
          cur_cutoff = min_active_cutoff;
          if (tid == 0 && n_cur > o.min_active) c.counters[6]++;
        } else {
          adaptive_beam = o.beam;
          cur_cutoff = beam_cutoff;
        }
      }
      RS_TP(1);
      const float cost_offset = (n_cur > 0) ? -best_cost : 0.f;
      const float *ll_row = loglikes + (ll_base + f) * ld;
      float local_min = INF;
      // Arc-parallel expansion: the out-degrees of the frame's tokens are prefix-summed in LDS and every thread takes
      // arcs j = tid, tid + NT, ... of the concatenated arc list (binary search for the owning token).  A token-per-thread
      // loop would serialise thousands of L2 atomics on the back-off / unigram states of an n-gram graph.
      // An arc is relaxed in two steps so that no thread ever waits for an atomic to come back from L2:
      //   1. atomicMin on best[dst] WITHOUT using the returned value (fire and forget) + the pair (dst, arc) appended to a
      //      candidate list -- unless the arc is already known to lose: tot >= (smallest candidate seen so far) + adaptive
      //      beam can only be >= the frame's final next_cutoff, i.e. it would be dropped below anyway;
      //   2. after a barrier, the candidate whose arc id sits in best[dst] is the one that created / owns the token and
      //      appends the state to the frame's token list (exactly one winner per state: keys are unique).
      // Before: one returning 64-bit atomic per arc, ~15 dependent L2 round trips per thread and frame on the ARPA graph.
      int *cand_s = queue[0], *cand_a = queue[1];
      const int cand_cap = S;
      // (arc record and log-likelihood are loaded by the caller, four arcs at a time: a thread's ~15 arcs per frame used to be 15
      // serialised chains of dependent L2 round trips -- token, arc range, arc, log-likelihood, atomic)
      auto relax_arc = [&](unsigned a, const int4 arc, float lk, float cur_cost, bool is_best, int src_tok) __attribute__((always_inline)) {
        const float graph_cost = __int_as_float(arc.z);
        const float ac_cost = cost_offset - lk;
        const float tot = (cur_cost + ac_cost) + graph_cost;
        if (is_best) {
          // :752-757  arc.weight + cost_offset - loglike + tok->tot_cost  (the reference's first bound)
          const float nw = ((graph_cost + cost_offset) - lk) + cur_cost;
          local_min = fminf(local_min, nw);
        }
        local_min = fminf(local_min, tot);
        cnt_arcs++;
        const float bound = FromOrdered(c.run_min) + adaptive_beam;
        if (!(tot < bound)) return;
        cnt_insert++;
        const unsigned ot = OrderedBits(tot);
        if (ot < c.run_min) atomicMin(&c.run_min, ot);
        atomicMin(&best[arc.w], PackKey(tot, a));            // result unused: non-returning
        const int ci = atomicAdd(&c.n_cand, 1);
        if (ci < cand_cap) { cand_s[ci] = arc.w; cand_a[ci] = (int)a; cand_t[ci] = src_tok; }   // (else: list full, step 2 scans the table)
      };
      if (tid == 0) { c.run_min = OrderedBits(INF); c.n_cand = 0; }
      __syncthreads();
      for (int c0 = 0; c0 < n_cur; c0 += kPrefixCap) {         // token chunks whose degree prefix fits LDS
        const int nc = n_cur - c0 < kPrefixCap ? n_cur - c0 : kPrefixCap;
        const int4 *ctok = cur + c0;
        if (c0 > 0) __syncthreads();      // the previous chunk's arc loop still reads c.pre (binary search) in other threads
        // degrees (0 for tokens beyond the cutoff), contiguous segment per thread
        const int seg = (nc + NT - 1) / NT;
        const int i0 = tid * seg < nc ? tid * seg : nc, i1 = i0 + seg < nc ? i0 + seg : nc;
        int lsum = 0;
        for (int ib = i0; ib < i1; ib += 4) {
          int4 tk[4];
          uint4 sr[4];
#pragma unroll
          for (int q = 0; q < 4; q++) tk[q] = ctok[ib + q < i1 ? ib + q : i1 - 1];
#pragma unroll
          for (int q = 0; q < 4; q++) sr[q] = h.state_rec[tk[q].x];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            if (ib + q >= i1) break;
            int deg = 0;
            if (__int_as_float(tk[q].y) <= cur_cutoff) { deg = (int)sr[q].z; cnt_expanded++; }
            c.pre[ib + q] = deg;
            c.tok_a0[ib + q] = sr[q].x + sr[q].y;
            c.tok_cost[ib + q] = __int_as_float(tk[q].y);
            lsum += deg;
          }
        }
        // exclusive scan of the per-thread sums over the block
        int inc = lsum;
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) { const int v = __shfl_up(inc, o2, 64); if ((tid & 63) >= o2) inc += v; }
        __syncthreads();          // previous chunk's readers of c.red_i / c.pre are done
        if ((tid & 63) == 63) c.red_i[tid >> 6] = inc;
        __syncthreads();
        int wbase = 0;
        for (int wv = 0; wv < (tid >> 6); wv++) wbase += c.red_i[wv];
        int total = 0;
        for (int wv = 0; wv < NT / 64; wv++) total += c.red_i[wv];
        int run = wbase + inc - lsum;
        for (int i = i0; i < i1; i++) { const int dgr = c.pre[i]; c.pre[i] = run; run += dgr; }
        if (tid == 0) c.pre[nc] = total;
        __syncthreads();
        for (int jb = tid; jb < total; jb += 4 * NT) {
          unsigned a[4];
          float cc[4];
          bool bst[4], on[4];
          int tki[4];
          int4 arc[4];
          float lk[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int j = jb + q * NT;
            on[q] = j < total;
            const int jj = on[q] ? j : total - 1;
            int lo = 0, hi = nc;            // last token with pre[t] <= j
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (c.pre[mid] <= jj) lo = mid; else hi = mid; }
            a[q] = c.tok_a0[lo] + (unsigned)(jj - c.pre[lo]);
            cc[q] = c.tok_cost[lo];
            tki[q] = c0 + lo;
            bst[q] = tki[q] == best_idx;
          }
#pragma unroll
          for (int q = 0; q < 4; q++) arc[q] = h.arcs[a[q]];
#pragma unroll
          for (int q = 0; q < 4; q++) lk[q] = ll_row[arc[q].x - 1];
#pragma unroll
          for (int q = 0; q < 4; q++) if (on[q]) relax_arc(a[q], arc[q], lk[q], cc[q], bst[q], tki[q]);
        }
      }
      __syncthreads();
      RS_TP(2);
      float mn;
      int dummy;
      BlockMinArg<NT>(c, local_min, tid, &mn, &dummy);
      const float next_cutoff = mn + adaptive_beam;   // = min over candidates of (tot_cost + adaptive_beam)
      if (tid == 0) {
        finfo[f * 4 + 0] = cost_offset;
        finfo[f * 4 + 1] = cur_cutoff;
        finfo[f * 4 + 2] = next_cutoff;
        finfo[f * 4 + 3] = adaptive_beam;
      }
      RS_TP(3);
      {
        // step 2 of the relaxation, and the cutoff in the same pass: the winner of a state either appends it to the frame's token
        // list or -- at or above the final cutoff, where the reference never keeps a token alive past the next frame's beam
        // (see header) -- clears the table entry again
        // The winner knows its source token (the candidate record), so the new token is written complete -- back pointer and
        // arc -- right here: no second resolution pass through arc -> source state -> token map for emitting arcs.
        const bool listed = c.n_cand <= cand_cap;      // workgroup-uniform
        if (listed) {
          const int nc2 = c.n_cand;
          for (int ib = tid; ib < nc2; ib += 4 * NT) {
            int s2[4], ca[4], ct[4];
            unsigned long long key[4];
            bool on[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int i = ib + q * NT;
              on[q] = i < nc2;
              const int ii = on[q] ? i : nc2 - 1;
              s2[q] = cand_s[ii];
              ca[q] = cand_a[ii];
              ct[q] = cand_t[ii];
            }
#pragma unroll
            for (int q = 0; q < 4; q++) key[q] = LoadKey(&best[s2[q]]);
#pragma unroll
            for (int q = 0; q < 4; q++) {
              if (!on[q] || (unsigned)(key[q] & 0xFFFFFFFFull) != (unsigned)ca[q]) continue;
              if (KeyCost(key[q]) < next_cutoff) {
                const int idx = atomicAdd(&c.n_next, 1);
                if (idx < next_cap) { next_toks[idx] = make_int4(s2[q], __float_as_int(KeyCost(key[q])), ct[q], ca[q]); map_next[s2[q]] = idx; }
                else c.overflow = 1;
              } else {
                StoreKey(&best[s2[q]], RS_EMPTY);
              }
            }
          }
        } else {
          // more candidates than the list holds: scan the table.  map_next still describes the frame just finished (its tokens
          // were entered when they were made and nothing has been written since), so the back pointers are read from it first and
          // the new frame's entries written after a barrier.
          for (int s2 = tid; s2 < S; s2 += NT) {
            const unsigned long long key = LoadKey(&best[s2]);
            if (key == RS_EMPTY) continue;
            if (KeyCost(key) < next_cutoff) {
              const int idx = atomicAdd(&c.n_next, 1);
              const unsigned arc = (unsigned)(key & 0xFFFFFFFFull);
              if (idx < next_cap) next_toks[idx] = make_int4(s2, __float_as_int(KeyCost(key)), map_next[h.arc_src[arc]], (int)arc);
              else c.overflow = 1;
            } else {
              StoreKey(&best[s2], RS_EMPTY);
            }
          }
          __syncthreads();
          const int nn2 = c.n_next < next_cap ? c.n_next : next_cap;
          for (int i = tid; i < nn2; i += NT) map_next[next_toks[i].x] = i;
        }
      }
      closure_cutoff = next_cutoff;
      __syncthreads();
      RS_TP(4);
    }
    // ================================================================ ProcessNonemitting(closure_cutoff)
    {
      int qi = 0;
      const int n0 = c.n_next < next_cap ? c.n_next : next_cap;
      __syncthreads();
      if (tid == 0) { c.q_n[0] = 0; c.q_n[1] = 0; }
      __syncthreads();
      for (int ib = tid; ib < n0; ib += 4 * NT) {
        int st[4];
        unsigned ne[4];
        bool on[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const int i = ib + q * NT; on[q] = i < n0; st[q] = next_toks[on[q] ? i : n0 - 1].x; }
#pragma unroll
        for (int q = 0; q < 4; q++) ne[q] = h.state_rec[st[q]].y;
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (on[q] && ne[q] != 0) queue[0][atomicAdd(&c.q_n[0], 1)] = st[q];
      }
      __syncthreads();
      int guard_rounds = 0;
      while (c.q_n[qi] > 0) {
        const int qn = c.q_n[qi];
        __syncthreads();
        if (tid == 0) c.q_n[qi ^ 1] = 0;
        round_id++;                       // a state is pushed once per round: stamp[s] == round_id says "already on the next queue"
        __syncthreads();
        // One queue entry per thread and pass (uniform trip count: the wave votes below).  In an n-gram graph thousands of
        // history states back off into ONE unigram state: when every relaxing lane of a wave targets the same state the wave
        // reduces its keys first and issues a single atomic instead of 64 serialised ones on one L2 address.
        for (int ib = 0; ib < qn; ib += NT) {
          const int i = ib + tid;
          const bool have = i < qn;
          const int s = have ? queue[qi][i] : 0;
          const float cur_cost = have ? KeyCost(LoadKey(&best[s])) : INF;
          const bool live = have && cur_cost < closure_cutoff;
          if (live) cnt_expanded++;
          const uint4 sr = live ? h.state_rec[s] : make_uint4(0u, 0u, 0u, 0u);
          const unsigned a0 = sr.x, ne = sr.y;
          unsigned ne_max = ne;
#pragma unroll
          for (int o2 = 32; o2 > 0; o2 >>= 1) ne_max = max(ne_max, (unsigned)__shfl_xor((int)ne_max, o2, 64));
          for (unsigned k = 0; k < ne_max; k++) {
            const bool has_arc = k < ne;
            const unsigned a = a0 + (has_arc ? k : 0u);
            const int4 arc = has_arc ? h.arcs[a] : make_int4(0, 0, 0, 0);
            const float tot = cur_cost + __int_as_float(arc.z);
            if (has_arc) cnt_arcs++;
            const bool act = has_arc && tot < closure_cutoff;
            if (act) cnt_insert++;
            const unsigned long long m = __ballot(act);
            if (m == 0ull) continue;
            const int first = __ffsll((long long)m) - 1;
            const int d0 = __shfl(arc.w, first, 64);
            const bool uniform = __ballot(act && arc.w != d0) == 0ull;
            unsigned long long key = act ? PackKey(tot, a) : RS_EMPTY;
            bool mine = act;
            if (uniform && __popcll(m) > 1) {
              unsigned long long kmin = key;
#pragma unroll
              for (int o2 = 32; o2 > 0; o2 >>= 1) {
                const unsigned lo32 = (unsigned)__shfl_xor((int)(unsigned)(kmin & 0xFFFFFFFFull), o2, 64);
                const unsigned hi32 = (unsigned)__shfl_xor((int)(unsigned)(kmin >> 32), o2, 64);
                const unsigned long long other = ((unsigned long long)hi32 << 32) | lo32;
                kmin = other < kmin ? other : kmin;
              }
              mine = act && key == kmin;          // keys are unique (arc ids): exactly one lane
              key = kmin;
            }
            if (mine) {
              const unsigned wa = (unsigned)(key & 0xFFFFFFFFull);
              if (Relax(c, best, map_next, next_toks, next_cap, arc.w, KeyCost(key), wa) && h.state_rec[arc.w].y != 0) {
                if (atomicExch(&stamp[arc.w], round_id) != round_id) queue[qi ^ 1][atomicAdd(&c.q_n[qi ^ 1], 1)] = arc.w;
              }
            }
          }
        }
        __syncthreads();
        qi ^= 1;
        if (++guard_rounds > 100000) { if (tid == 0) c.error = 2; break; }   // epsilon cycle in the graph
      }
      __syncthreads();
    }
    RS_TP(5);
    // ================================================================ materialise frame f+1
    {
      const int nn = c.n_next < next_cap ? c.n_next : next_cap;
      // Final cost of every token of the new frame, the back pointer of those an epsilon arc made or improved (their .w is not the
      // arc in the table: the source then owns a token of this same frame and map_next names it), and -- same visit -- the table entry
      // cleared for the next frame.
      for (int ib = tid; ib < nn; ib += 4 * NT) {          // four tokens per thread at a time: every load stage of the four in one go
        int4 tk[4];
        int sx[4], bp[4];
        unsigned long long key[4];
        bool on[4], eps[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const int i = ib + q * NT; on[q] = i < nn; tk[q] = next_toks[on[q] ? i : nn - 1]; }
#pragma unroll
        for (int q = 0; q < 4; q++) key[q] = LoadKey(&best[tk[q].x]);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const unsigned arc = (unsigned)(key[q] & 0xFFFFFFFFull);
          eps[q] = arc != RS_NOARC && (int)arc != tk[q].w;
          sx[q] = eps[q] ? h.arc_srcx[arc] & 0x7fffffff : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const unsigned arc = (unsigned)(key[q] & 0xFFFFFFFFull);
          bp[q] = eps[q] ? map_next[sx[q]] : (arc == RS_NOARC ? -1 : tk[q].z);
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (on[q]) {
            next_toks[ib + q * NT] = make_int4(tk[q].x, __float_as_int(KeyCost(key[q])), bp[q], (int)(unsigned)(key[q] & 0xFFFFFFFFull));
            StoreKey(&best[tk[q].x], RS_EMPTY);
          }
      }
      __syncthreads();
      off_cur = off_next;
      n_cur = nn;
      off_next = off_cur + n_cur;
      if (tid == 0) {
        frame_off[f + 2] = off_next;
        c.counters[3] += (unsigned long long)nn;
        {      // [4]: largest frame of the utterance: tokens (low half), candidate records of its arc loop (high half)
          const unsigned long long mt = c.counters[4] & 0xFFFFFFFFull, mc = c.counters[4] >> 32;
          const unsigned long long nc_ = f >= 0 ? (unsigned long long)c.n_cand : 0ull;
          c.counters[4] = ((nc_ > mc ? nc_ : mc) << 32) | ((unsigned long long)nn > mt ? (unsigned long long)nn : mt);
        }
        c.n_next = 0;
        if (nn == 0 && c.error == 0) c.error = 1;     // "no surviving tokens"
      }
      __syncthreads();
      RS_TP(6);
      if (c.error) break;
    }
  }
#ifdef RS_DECODE_PROFILE
  if (tid == 0 && T > 0)
    printf("token-list block %d: %lld cycles, %d tokens, T=%d phases %lld %lld %lld %lld %lld %lld %lld\n", u,
           prof[0] + prof[1] + prof[2] + prof[3] + prof[4] + prof[5] + prof[6], off_next, T, prof[0], prof[1], prof[2], prof[3], prof[4], prof[5],
           prof[6]);
  if (u == 0 && tid == 0 && T > 0)
    printf("token-list decode cycles/frame: best %lld cutoff %lld expand %lld next-min %lld winners+cutoff %lld closure %lld materialise %lld (T=%d)\n",
           prof[0] / T, prof[1] / T, prof[2] / T, prof[3] / T, prof[4] / T, prof[5] / T, prof[6] / T, T);
#endif
  // ================================================================ final costs + best-path traceback
  // (frame T's tokens are tokens[off_cur .. off_cur + n_cur))
  {
    const int4 *cur = tokens + off_cur;
    float lv1 = INF, lv2 = INF;
    int li1 = 0x7fffffff, li2 = 0x7fffffff;
    for (int i = tid; i < n_cur; i += NT) {
      const float cst = __int_as_float(cur[i].y);
      const float wf = cst + h.final_cost[cur[i].x];
      if (wf < lv1 || (wf == lv1 && i < li1)) { lv1 = wf; li1 = i; }
      if (cst < lv2 || (cst == lv2 && i < li2)) { lv2 = cst; li2 = i; }
    }
    float b1, b2;
    int i1, i2;
    BlockMinArg<NT>(c, lv1, li1, &b1, &i1);
    BlockMinArg<NT>(c, lv2, li2, &b2, &i2);
    atomicAdd(&c.counters[0], cnt_expanded);
    atomicAdd(&c.counters[1], cnt_arcs);
    atomicAdd(&c.counters[2], cnt_insert);
    __syncthreads();
    if (tid == 0) {
      const bool reached = b1 < INF;
      int idx = reached ? i1 : i2;
      int F = c.error ? -1 : T;     // frame index of the token list (0..T)
      double graph = 0.0, ac = 0.0;
      int nw = 0;
      int *words = w.out_words + (size_t)u * w.max_words;
      bool truncated = false;
      if (F >= 0 && n_cur > 0) {
        if (reached) graph += (double)h.final_cost[cur[idx].x];
        while (true) {
          const int4 tk = tokens[frame_off[F] + idx];
          if (tk.w < 0) break;
          const int4 arc = h.arcs[tk.w];
          graph += (double)__int_as_float(arc.z);
          if (arc.x != 0) {
            F -= 1;
            const float off = finfo[F * 4 + 0];
            const float lk = loglikes[(ll_base + F) * ld + (arc.x - 1)];
            const float link_ac = off - lk;                   // ForwardLink::acoustic_cost
            ac += (double)(link_ac - off);                    // GetRawLattice :166-172
          }
          if (arc.y != 0) {
            if (nw < w.max_words) words[nw++] = arc.y;
            else truncated = true;
          }
          idx = tk.z;
        }
        for (int a = 0, b = nw - 1; a < b; a++, b--) { int t2 = words[a]; words[a] = words[b]; words[b] = t2; }
      }
      w.out_nwords[u] = (c.error || truncated) ? -1 : nw;
      float *oc = w.out_costs + (size_t)u * 4;
      oc[0] = (float)graph;
      oc[1] = (float)ac;
      oc[2] = reached ? b1 : b2;
      oc[3] = reached ? 1.f : 0.f;
      c.counters[7] = (unsigned long long)c.overflow + 2ull * (unsigned long long)c.error;
      long long *ctr = w.counters + (size_t)u * 8;
      for (int i = 0; i < 8; i++) ctr[i] = (long long)c.counters[i];
      frame_off[T + 1] = off_next;
    }
    // leave both state->token maps empty (the lattice pass rebuilds them frame by frame)
    for (int i = tid; i < S; i += NT) { w.map_a[(size_t)u * S + i] = -1; w.map_b[(size_t)u * S + i] = -1; }
  }
}

// ===================================================================================== lattice extraction
// Backward pass = FinalizeDecoding: PruneForwardLinksFinal on the last frame, PruneForwardLinks(delta = 0) on
// every earlier frame (lattice-faster-decoder.cc:376-458,299-370,625-640).  Forward links are not stored by
// the search kernel; they are re-derived here from the token lists with the same float expressions and the
// same existence rules (emitting link: source cost <= cur_cutoff and tot < next_cutoff; epsilon link: source
// cost < closure cutoff and tot < closure cutoff), so link costs are bit-identical to the search's.
// extra_cost is the exact fixpoint the reference's "while (changed)" loops converge to.
template <int NT>
struct LatticeCtx { float red_f[NT / 64]; int red_i[NT / 64]; float bcast_f[2]; int bcast_i[4]; };
template <int NT>
__global__ __launch_bounds__(NT) void LatticeKernel(HclgDev h, DecodeOptsDev o, BatchGeom g,
                                                    const float *__restrict__ loglikes, int ld, DecodeWork w, LatticeWork lw) {
  // (only what BlockMinArg needs: the search's BlockCtx is 97 KB of LDS, which kept every other kernel that needs LDS -- the next
  // call's layer GEMMs -- off the CUs of a lattice pass for its 2.3 ms)
  __shared__ LatticeCtx<NT> c;
  __shared__ int s_changed;
  const int u = blockIdx.x, tid = threadIdx.x;
  const int T = g.d_num_frames[u];
  if (T <= 0 || w.out_nwords[u] < 0) return;
  const int S = h.num_states;
  int *map_cur = w.map_a + (size_t)u * S, *map_nxt = w.map_b + (size_t)u * S;
  const int4 *tokens = w.tokens + (size_t)u * w.tok_cap;
  const int *frame_off = w.frame_tok_off + (size_t)u * (g.max_frames + 2);
  const float *finfo = w.frame_info + (size_t)u * (g.max_frames + 1) * 4;
  float *extra = lw.extra_cost + (size_t)u * w.tok_cap;
  LatArc *my_arcs = lw.arcs + (size_t)u * lw.utt_cap;
  const float INF = INFINITY, beam = o.lattice_beam;
  const size_t ll_base = (size_t)g.d_row_base[u] + g.L;

  // ---- final costs on the last frame (ComputeFinalCosts)
  float final_best;
  bool have_final;
  {
    const int4 *cur = tokens + frame_off[T];
    const int n = frame_off[T + 1] - frame_off[T];
    float lv1 = INF, lv2 = INF;
    for (int i = tid; i < n; i += NT) {
      const float cst = __int_as_float(cur[i].y);
      lv1 = fminf(lv1, cst + h.final_cost[cur[i].x]);
      lv2 = fminf(lv2, cst);
    }
    float b1, b2;
    int d1, d2;
    BlockMinArg<NT>(c, lv1, tid, &b1, &d1);
    BlockMinArg<NT>(c, lv2, tid, &b2, &d2);
    have_final = b1 < INF;
    final_best = have_final ? b1 : b2;
  }
  for (int f = T; f >= 0; f--) {
    const int off = frame_off[f], n = frame_off[f + 1] - off;
    const int off_n = frame_off[f + 1];
    const int4 *cur = tokens + off;
    const int4 *nxt = tokens + off_n;
    for (int i = tid; i < n; i += NT) map_cur[cur[i].x] = i;
    __syncthreads();
    const float cost_offset = f < T ? finfo[f * 4 + 0] : 0.f, cur_cutoff = f < T ? finfo[f * 4 + 1] : 0.f,
                next_cutoff = f < T ? finfo[f * 4 + 2] : 0.f;
    const float closure_cutoff = f > 0 ? finfo[(f - 1) * 4 + 2] : o.beam;
    const float *ll_row = loglikes + (ll_base + (f < T ? f : 0)) * ld;
    // ---- pass 1: the part of extra_cost that does not depend on this frame's other tokens
    for (int i = tid; i < n; i += NT) {
      const int4 tk = cur[i];
      const float cost = __int_as_float(tk.y);
      float e = INF;
      if (f == T) e = cost + (have_final ? h.final_cost[tk.x] : 0.f) - final_best;
      else if (cost <= cur_cutoff) {
        const unsigned a0 = h.arc_begin[tk.x] + h.num_ieps[tk.x], a1 = h.arc_begin[tk.x + 1];
        for (unsigned a = a0; a < a1; a++) {
          const int4 arc = h.arcs[a];
          const float ac = cost_offset - ll_row[arc.x - 1];
          const float tot = (cost + ac) + __int_as_float(arc.z);
          if (!(tot < next_cutoff)) continue;
          const int j = map_nxt[arc.w];
          if (j < 0) continue;
          float le = extra[off_n + j] + (tot - __int_as_float(nxt[j].y));
          if (le > beam) continue;
          if (le < 0.f) le = 0.f;
          e = fminf(e, le);
        }
      }
      extra[off + i] = e;
    }
    __syncthreads();
    // ---- pass 2: epsilon links inside the frame, to the fixpoint
    for (int round = 0; round < 1000; round++) {
      if (tid == 0) s_changed = 0;
      __syncthreads();
      for (int i = tid; i < n; i += NT) {
        const int4 tk = cur[i];
        if (h.num_ieps[tk.x] == 0) continue;
        const float cost = __int_as_float(tk.y);
        if (cost >= closure_cutoff) continue;
        float e = extra[off + i];
        const float e0 = e;
        const unsigned a0 = h.arc_begin[tk.x], a1 = a0 + h.num_ieps[tk.x];
        for (unsigned a = a0; a < a1; a++) {
          const int4 arc = h.arcs[a];
          const float tot = cost + __int_as_float(arc.z);
          if (!(tot < closure_cutoff)) continue;
          const int j = map_cur[arc.w];
          if (j < 0) continue;
          float le = __hip_atomic_load(&extra[off + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + (tot - __int_as_float(cur[j].y));
          if (le > beam) continue;
          if (le < 0.f) le = 0.f;
          e = fminf(e, le);
        }
        if (e < e0) { __hip_atomic_store(&extra[off + i], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); s_changed = 1; }
      }
      __syncthreads();
      const int ch = s_changed;
      __syncthreads();
      if (!ch) break;
    }
    if (f == T) {
      for (int i = tid; i < n; i += NT) if (extra[off + i] > beam) extra[off + i] = INF;
      __syncthreads();
    }
    // ---- pass 3: emit surviving links (and the final-cost records of the last frame)
    for (int i = tid; i < n; i += NT) {
      const int4 tk = cur[i];
      const float cost = __int_as_float(tk.y);
      if (!(extra[off + i] < INF)) continue;
      if (f == T) {
        const float fc = have_final ? h.final_cost[tk.x] : 0.f;
        if (fc < INF) {
          const int k = atomicAdd(&lw.arcs_count[u], 1);
          if (k < lw.utt_cap) my_arcs[k] = LatArc{u, off + i, -1, -1, fc, 0.f};
        }
      } else if (cost <= cur_cutoff) {
        const unsigned a0 = h.arc_begin[tk.x] + h.num_ieps[tk.x], a1 = h.arc_begin[tk.x + 1];
        for (unsigned a = a0; a < a1; a++) {
          const int4 arc = h.arcs[a];
          const float ac = cost_offset - ll_row[arc.x - 1];
          const float gc = __int_as_float(arc.z);
          const float tot = (cost + ac) + gc;
          if (!(tot < next_cutoff)) continue;
          const int j = map_nxt[arc.w];
          if (j < 0) continue;
          const float le = extra[off_n + j] + (tot - __int_as_float(nxt[j].y));
          if (le > beam) continue;
          const int k = atomicAdd(&lw.arcs_count[u], 1);
          if (k < lw.utt_cap) my_arcs[k] = LatArc{u, off + i, off_n + j, (int)a, gc, ac - cost_offset};
        }
      }
      if (h.num_ieps[tk.x] != 0 && cost < closure_cutoff) {
        const unsigned a0 = h.arc_begin[tk.x], a1 = a0 + h.num_ieps[tk.x];
        for (unsigned a = a0; a < a1; a++) {
          const int4 arc = h.arcs[a];
          const float gc = __int_as_float(arc.z);
          const float tot = cost + gc;
          if (!(tot < closure_cutoff)) continue;
          const int j = map_cur[arc.w];
          if (j < 0) continue;
          const float le = extra[off + j] + (tot - __int_as_float(cur[j].y));
          if (le > beam) continue;
          const int k = atomicAdd(&lw.arcs_count[u], 1);
          if (k < lw.utt_cap) my_arcs[k] = LatArc{u, off + i, off + j, (int)a, gc, 0.f};
        }
      }
    }
    __syncthreads();
    // retire frame f+1's map, keep frame f's as "next"
    if (f < T) {
      const int n_nxt = frame_off[f + 2] - off_n;
      for (int i = tid; i < n_nxt; i += NT) map_nxt[nxt[i].x] = -1;
    }
    __syncthreads();
    int *tmp = map_cur; map_cur = map_nxt; map_nxt = tmp;
  }
  // clear the last map (frame 0)
  {
    const int n = frame_off[1] - frame_off[0];
    for (int i = tid; i < n; i += NT) map_nxt[tokens[frame_off[0] + i].x] = -1;
  }
}

__global__ __launch_bounds__(256) void CompactArcsKernel(const LatArc *__restrict__ arcs, int utt_cap, const int *__restrict__ counts, int n_utts, LatArc *__restrict__ dst) {
  __shared__ int s_part[4];
  const int u = blockIdx.x, tid = threadIdx.x;
  int before = 0;
  for (int v = tid; v < u; v += 256) before += min(counts[v], utt_cap);
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) before += __shfl_xor(before, o2, 64);
  if ((tid & 63) == 0) s_part[tid >> 6] = before;
  __syncthreads();
  const size_t base = (size_t)s_part[0] + s_part[1] + s_part[2] + s_part[3];
  const int n = min(counts[u], utt_cap);
  static_assert(sizeof(LatArc) == 24, "copied as 8-byte words");
  const unsigned long long *src = reinterpret_cast<const unsigned long long *>(arcs + (size_t)u * utt_cap);
  unsigned long long *out = reinterpret_cast<unsigned long long *>(dst + base);
  for (int i = tid; i < 3 * n; i += 256) out[i] = src[i];
}
void LaunchCompactArcs(const LatArc *arcs, int utt_cap, const int *counts, int n_utts, LatArc *dst, hipStream_t s) {
  if (n_utts <= 0) return;
  hipLaunchKernelGGL(CompactArcsKernel, dim3(n_utts), dim3(256), 0, s, arcs, utt_cap, counts, n_utts, dst);
}

void LaunchLatticePrune(const HclgDev &h, const DecodeOptsDev &o, const BatchGeom &g, const float *loglikes, int ld,
                        const DecodeWork &w, const LatticeWork &lw, hipStream_t s) {
  if (g.n_utts == 0) return;
  hipLaunchKernelGGL(LatticeKernel<256>, dim3(g.n_utts), dim3(256), 0, s, h, o, g, loglikes, ld, w, lw);
}

void LaunchDecode(const HclgDev &h, const DecodeOptsDev &o, const BatchGeom &g, const float *loglikes, int ld,
                  const DecodeWork &w, hipStream_t s) {
  if (g.n_utts == 0) return;
  if (h.num_states > 20000)
    hipLaunchKernelGGL(DecodeKernel<1024>, dim3(g.n_utts), dim3(1024), 0, s, h, o, g, loglikes, ld, w);
  else
    hipLaunchKernelGGL(DecodeKernel<256>, dim3(g.n_utts), dim3(256), 0, s, h, o, g, loglikes, ld, w);
}

}  // namespace rs
