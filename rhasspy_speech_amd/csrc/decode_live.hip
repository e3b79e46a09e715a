// Large-graph token-passing search, round 5: the live-state-table search rebuilt around the number of DEPENDENT memory round
// trips per frame and around how many utterances a CU holds at once.
//
// Reference semantics (kaldi/src/decoder/lattice-faster-decoder.cc): InitDecoding :56-73, GetCutoff :644-711, ProcessEmitting
// :714-804, ProcessNonemitting :820-887, FindOrAddToken :253-293, ComputeFinalCosts :536-577, GetBestPath
// (lattice-faster-online-decoder.cc:56-173).  Cut-offs, float expressions, tie rule (lowest arc index) and the token lists it
// leaves behind (up to order within a frame) are DecodeKernel's / round 3's HashDecodeKernel's; what changed is how a frame is run:
//
//  * the kernel is latency-bound (round 3/4: ~35 us of fixed latency per frame -- ~40 dependent L2 round trips and ~25
//    workgroup barriers -- plus ~10 ns per token; 0.08 of the HBM roofline).  A frame is now ~17 dependent round trips:
//      - GetCutoff needs NO sweep in the common frame: the pass that completes a frame's tokens leaves a 256-bin histogram of
//        their costs over [cheapest candidate, next_cutoff) in LDS, every token is known to lie below next_cutoff, so "how many
//        tokens are below best + beam" is the frame's size, and when max-active binds the rank's bin comes from the histogram
//        and ONE sweep collects it (round 3: count sweep + histogram sweep + collecting sweep + recount);
//      - expansion is token-parallel: a thread loads a token, its state record and its first two emitting arcs (the HCLG of an
//        n-gram LM is HMM chains: 99 % of the states have two), four tokens per thread in flight; no compacted list, no degree
//        prefix over the frame, no binary search per arc.  Tokens with more arcs go onto a list and only THOSE arcs are dealt out
//        over the threads by a (small) prefix + search;
//      - candidate records are one 16-byte record; the winners' pass reads it, the key and -- only when the arc says the
//        destination has epsilon arcs -- the destination's state record in one round trip;
//      - the closure's work-list entries carry everything the next round needs (slot, first epsilon arc, count, the key that
//        was written, the token that wrote it): pop = entry -> {key, token index, arcs} -> {atomicMin, next state record} ->
//        push, three dependent round trips per round (round 3: seven), no stamp array -- an entry is stale iff the slot's key is
//        no longer the one it carries -- and the popped entry writes the back pointer into its token, so
//      - completing the frame's tokens is token -> key -> store (round 3: + source state of the winning epsilon arc -> table
//        lookup -> slot's token);
//  * the state -> slot table has two levels: 2^HLOG entries in LDS (eight probes) and, behind it, a 32 K-entry table in global
//    memory (L2) for the states that find their eight LDS entries taken.  The LDS part no longer has to hold the largest frame
//    of the batch, so it is 64 KB (or 32) instead of 128 and TWO (four) workgroups share a CU: while one waits at a barrier or
//    for a round trip the other runs.  Nothing is restarted when a frame outgrows the LDS part; only > 24 576 live states (the
//    slot arrays' length) hands the utterance to DecodeKernel as before.
// Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstdio>

#include "kernels.h"
#include "decode_tok.h"

namespace rs {
using namespace tok;
namespace {

constexpr int kGlobalLog = 15;                      // second-level table: 32768 entries per utterance in global memory
constexpr int kGlobalSize = 1 << kGlobalLog;
constexpr int kSlotBits = 15;
constexpr int kSlotCap = (1 << kSlotBits) * 3 / 4;  // live states per frame (compact form): 24576
constexpr unsigned kFree = 0xFFFFFFFFu;
constexpr int kLdsProbes = 8;
constexpr int kBigCap = 512;                        // high-degree tokens whose arcs are dealt out per chunk
constexpr int kInline = 2;                          // emitting arcs a thread relaxes itself
#ifndef RS_LIVE_Q
#define RS_LIVE_Q 4
#endif
constexpr int kNoBp = 0xFFFF;                       // work-list entry that must not write a back pointer (made by an emitting arc)

__device__ __forceinline__ unsigned LdsTag(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned GlbTag(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Monotone binning of a token cost over [lo, lo + 256 / scale): both the pass that fills the histogram and the pass that collects
// the rank's bin use this one expression.
__device__ __forceinline__ int CostBin(float c, float lo, float scale) {
  int b = (int)((c - lo) * scale);
  return b > 255 ? 255 : (b < 0 ? 0 : b);
}

template <int NT>
struct LiveCtx {
  float red_f[NT / 64];
  int red_i[NT / 64];
  unsigned hist[256];                // scratch of the radix select (BlockKthSmallest)
  float bcast_f[2];
  int bcast_i[4];
  int n_next, q_n[2], overflow, error;
  unsigned run_min;                  // ordered bits of the smallest candidate cost seen so far in this frame
  unsigned min_bits;                 // ordered bits of the cheapest candidate of the frame, pruned ones included (next_cutoff - adaptive beam)
  int n_cand, n_big, n_slots, g_used, redo, kth_n;
  unsigned long long best_key;       // (ordered cost bits << 32 | index) of the cheapest token of the frame just completed
  unsigned long long counters[8];
  unsigned chist[256];               // costs of the frame just completed, binned over [hist_lo, hist_hi)
  float kth_cand[256];
  int big_pre[kBigCap + 1];
  int4 big_ent[kBigCap];
};

// slot of `state`, claiming a fresh one if the state is not in the table yet; -1: no slot / no entry left.
// WIDE = false: entry = state << 15 | slot, slots handed out consecutively (graphs below 131 071 states); WIDE = true: entry = state
// id, the slot is the entry's position (LDS part: [0, 2^HLOG), global part behind it).
template <int HLOG, bool WIDE>
__device__ __forceinline__ int SlotFindOrInsert(unsigned *tags, unsigned *gtags, int *n_slots, int *g_used, unsigned lds_mask,
                                                unsigned state, int slot_limit) {
  unsigned hp = ((state * 2654435761u) >> (32 - HLOG)) & lds_mask;
  int mine = -1;
#pragma unroll 1
  for (int probe = 0; probe < kLdsProbes; probe++) {
    unsigned e = LdsTag(&tags[hp]);
    if (e == kFree) {
      if (mine < 0) {
        mine = atomicAdd(n_slots, 1);
        if (mine >= slot_limit) return -1;
      }
      e = atomicCAS(&tags[hp], kFree, WIDE ? state : ((state << kSlotBits) | (unsigned)mine));
      if (e == kFree) return WIDE ? (int)hp : mine;
      // another lane claimed this entry first (possibly for the same state: then `mine` stays unused, its key stays empty)
    }
    if (WIDE) { if (e == state) return (int)hp; }
    else if ((e >> kSlotBits) == state) return (int)(e & ((1u << kSlotBits) - 1u));
    hp = (hp + 1) & lds_mask;
  }
  // All eight LDS entries belong to other states.  Entries are never released within a frame, so every lane looking for this state
  // finds them taken too and continues here.
  *g_used = 1;
  unsigned gp = (state * 2246822519u) >> (32 - kGlobalLog);
#pragma unroll 1
  for (int probe = 0; probe < 1024; probe++) {
    unsigned e = GlbTag(&gtags[gp]);
    if (e == kFree) {
      if (mine < 0) {
        mine = atomicAdd(n_slots, 1);
        if (mine >= slot_limit) return -1;
      }
      e = atomicCAS(&gtags[gp], kFree, WIDE ? state : ((state << kSlotBits) | (unsigned)mine));
      if (e == kFree) return WIDE ? (1 << HLOG) + (int)gp : mine;
    }
    if (WIDE) { if (e == state) return (1 << HLOG) + (int)gp; }
    else if ((e >> kSlotBits) == state) return (int)(e & ((1u << kSlotBits) - 1u));
    gp = (gp + 1) & (kGlobalSize - 1);
  }
  return -1;
}

// k-th smallest (0-based) of the n token costs whose histogram (CostBin over lo / scale) is chist: the bin that holds the rank
// from the histogram, one sweep that collects that bin's values, direct ranking.  Radix select when the bin is crowded.
template <int NT, class Ctx>
__device__ float KthFromCommitHist(Ctx &c, const int4 *toks, int n, int k, float lo, float scale, float min_cost) {
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < 64) {
    const int h0 = (int)c.chist[4 * lane], h1 = (int)c.chist[4 * lane + 1], h2 = (int)c.chist[4 * lane + 2], h3 = (int)c.chist[4 * lane + 3];
    const int tot = h0 + h1 + h2 + h3;
    int inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
    const int exc = inc - tot;
    if (exc <= k && k < inc) {
      int acc = exc, b = 4 * lane, cnt = h0;
      if (acc + h0 <= k) { acc += h0; b++; cnt = h1; if (acc + h1 <= k) { acc += h1; b++; cnt = h2; if (acc + h2 <= k) { acc += h2; b++; cnt = h3; } } }
      c.bcast_i[0] = b; c.bcast_i[1] = k - acc; c.bcast_i[2] = cnt;
    }
    if (lane == 63 && inc <= k) c.bcast_i[2] = -1;      // fewer than k + 1 values (caller error): radix select
    if (lane == 0) c.kth_n = 0;
  }
  __syncthreads();
  const int bin = c.bcast_i[0], kk = c.bcast_i[1], cnt = c.bcast_i[2];
  __syncthreads();
  if (cnt < 0 || cnt > 256) return BlockKthSmallest<NT>(c, toks, n, k, min_cost);
  for (int i0 = tid; i0 < n; i0 += 4 * NT) {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const int i = i0 + q * NT; v[q] = i < n ? __int_as_float(toks[i].y) : INFINITY; }
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (i0 + q * NT < n && CostBin(v[q], lo, scale) == bin) { const int p = atomicAdd(&c.kth_n, 1); if (p < 256) c.kth_cand[p] = v[q]; }
  }
  __syncthreads();
  if (c.kth_n != cnt) return BlockKthSmallest<NT>(c, toks, n, k, min_cost);      // (cannot happen: the histogram is of these tokens)
  if (tid < cnt) {
    const float v = c.kth_cand[tid];
    int lt = 0, le = 0;
    for (int j = 0; j < cnt; j++) { const float x = c.kth_cand[j]; lt += (int)(x < v); le += (int)(x <= v); }
    if (lt <= kk && kk < le) c.bcast_f[1] = v;
  }
  __syncthreads();
  const float ans = c.bcast_f[1];
  __syncthreads();
  return ans;
}

#ifdef RS_DECODE_PROFILE
#define RS_LP(i) do { __syncthreads(); long long _n = clock64(); if (threadIdx.x == 0) prof[i] += _n - t_last; t_last = clock64(); } while (0)
#else
#define RS_LP(i) do { } while (0)
#endif

template <int NT, int HLOG, bool WIDE>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void LiveDecodeKernel(HclgDev h, DecodeOptsDev o, BatchGeom g,
                                                       const float *__restrict__ loglikes, int ld, DecodeWork w) {
  using Ctx = LiveCtx<NT>;
  constexpr int HS = 1 << HLOG;
  constexpr int NW = NT / 64;
  __shared__ Ctx c;
  __shared__ __attribute__((aligned(16))) unsigned tags[HS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int u = blockIdx.x, tid = threadIdx.x;
  const int T = g.d_num_frames[u];
  const size_t tab = (size_t)w.h_tab;
  unsigned long long *keys = w.h_keys + (size_t)u * tab;
  int *slot_tok = w.h_slot_tok + (size_t)u * tab;            // slot -> index of its token in the frame under construction
  unsigned *gtags = w.h_gtags + (size_t)u * kGlobalSize;
  int4 *cand = reinterpret_cast<int4 *>(w.h_cand) + (size_t)u * w.h_cand_cap;      // {arc, destination | has-epsilon-arcs << 31, slot << 16 | source token, cost bits}
  const int qcap = w.h_qcap;
  int4 *queue[2] = {w.h_q4 + (size_t)u * 2 * qcap, w.h_q4 + (size_t)u * 2 * qcap + qcap};          // {slot | writer token << 16, first epsilon arc, key low, key high}
  int *queue_ne[2] = {w.h_qne + (size_t)u * 2 * qcap, w.h_qne + (size_t)u * 2 * qcap + qcap};      // epsilon arcs of the entry's state
  int4 *big = w.h_comp + (size_t)u * kSlotCap;               // tokens with more than kInline emitting arcs: {first arc left, cost bits, token index, arcs left}
  int4 *tokens = w.tokens + (size_t)u * w.tok_cap;
  int *frame_off = w.frame_tok_off + (size_t)u * (g.max_frames + 2);
  float *finfo = w.frame_info + (size_t)u * (g.max_frames + 1) * 4;
  const int4 *arcsf = h.arcs_f;                              // arcs with bit 31 of .x = "the destination state has epsilon arcs"
  const float INF = INFINITY;
  const size_t ll_base = (size_t)g.d_row_base[u] + g.L;
  const int cand_cap = w.h_cand_cap;
  const int slot_limit = w.h_slot_limit < kSlotCap ? w.h_slot_limit : kSlotCap;      // (tests lower it to send utterances to DecodeKernel)
  const unsigned lds_mask = (1u << (w.h_lds_log > 0 && w.h_lds_log < HLOG ? w.h_lds_log : HLOG)) - 1u;      // (tests shrink the LDS part)

  for (int i = tid; i < (WIDE ? HS + kGlobalSize : kSlotCap); i += NT) StoreKey(&keys[i], RS_EMPTY);
  for (int i = tid; i < kGlobalSize; i += NT) __hip_atomic_store(&gtags[i], kFree, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int i = tid; i < HS; i += NT) tags[i] = kFree;
  for (int i = tid; i < 256; i += NT) c.chist[i] = 0;
  if (tid == 0) {
    c.n_next = 0; c.overflow = 0; c.error = 0; c.q_n[0] = c.q_n[1] = 0;
    for (int i = 0; i < 8; i++) c.counters[i] = 0;
    c.n_slots = 0; c.redo = 0; c.g_used = 0; c.n_big = 0; c.n_cand = 0;
    c.best_key = RS_EMPTY;
    w.out_nwords[u] = 0;
    w.redo[u] = 0;
  }
  unsigned cnt_expanded = 0, cnt_arcs = 0, cnt_insert = 0;      // (per thread: 32 bits hold an utterance's share)
  __syncthreads();
  auto find_or_insert = [&](unsigned state) __attribute__((always_inline)) {
    return SlotFindOrInsert<HLOG, WIDE>(tags, gtags, &c.n_slots, &c.g_used, lds_mask, state, slot_limit);
  };

  int off_cur = 0, n_cur = 0;       // frame f's token list
  int off_next = 0;                 // frame under construction
  if (tid == 0) {                   // InitDecoding: the start state's token
    const int sl = find_or_insert((unsigned)h.start);
    const unsigned long long k0 = PackKey(0.0f, RS_NOARC);
    StoreKey(&keys[sl], k0);
    tokens[0] = make_int4(h.start, sl, -1, -2);
    slot_tok[sl] = 0;
    c.n_next = 1;
    frame_off[0] = 0;
    const uint4 sr = h.state_rec[h.start];
    if (sr.y != 0) {
      queue[0][0] = make_int4(sl | (kNoBp << 16), (int)sr.x, (int)(unsigned)(k0 & 0xFFFFFFFFull), (int)(unsigned)(k0 >> 32));
      queue_ne[0][0] = (int)sr.y;
      c.q_n[0] = 1;
    }
  }
  __syncthreads();
  float closure_cutoff = o.beam;    // InitDecoding: ProcessNonemitting(config_.beam)
  float hist_lo = 0.f;              // the histogram the completion pass is about to fill: bins over [hist_lo, closure cutoff)
  float best_cost = INF;            // cheapest token of frame f and its index: found while the frame's tokens were completed
  int best_idx = 0;
#ifdef RS_DECODE_PROFILE
  long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t_last = clock64();
#endif

  for (int f = -1; f < T; f++) {
    int4 *next_toks = tokens + off_next;
    // a frame holds at most kSlotCap tokens (one per slot); records name tokens of a frame in 15 / 16 bits
    const int next_cap = w.tok_cap - off_next < slot_limit ? w.tok_cap - off_next : slot_limit;
    if (f >= 0) {
      // ================================================================ ProcessEmitting(frame f)
      const int4 *cur = tokens + off_cur;
      // ---- GetCutoff.  The reference selects the (max_active+1)-th / (min_active+1)-th cheapest cost and then only asks on which
      // side of best + beam it lies.  Every token of this frame is below hist_hi = the cutoff it was created under, so with
      // best + beam >= hist_hi (always, unless beam_delta pushed an adaptive beam beyond the beam) all n_cur tokens are below it.
      const float hist_hi = closure_cutoff;
      const float hist_scale = 256.0f / (hist_hi - hist_lo);
      const float beam_cutoff = best_cost + o.beam;
      int n_lt = n_cur, n_le = n_cur;
      if (!(beam_cutoff >= hist_hi)) {       // workgroup-uniform; count
        if (tid == 0) { c.bcast_i[2] = 0; c.bcast_i[3] = 0; }
        __syncthreads();
        int a_lt = 0, a_le = 0;
        for (int i = tid; i < n_cur; i += NT) { const float cst = __int_as_float(cur[i].y); a_lt += (int)(cst < beam_cutoff); a_le += (int)(cst <= beam_cutoff); }
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { a_lt += __shfl_xor(a_lt, o2, 64); a_le += __shfl_xor(a_le, o2, 64); }
        if (lane == 0) { atomicAdd(&c.bcast_i[2], a_lt); atomicAdd(&c.bcast_i[3], a_le); }
        __syncthreads();
        n_lt = c.bcast_i[2]; n_le = c.bcast_i[3];
        __syncthreads();
      }
      float max_active_cutoff = INF, min_active_cutoff = INF, cur_cutoff, adaptive_beam;
      bool decided = false;
      // sorted[max_active] < beam_cutoff  <=>  more than max_active costs lie below beam_cutoff
      if (n_cur > o.max_active && n_lt > o.max_active) max_active_cutoff = KthFromCommitHist<NT>(c, cur, n_cur, o.max_active, hist_lo, hist_scale, best_cost);
      if (max_active_cutoff < beam_cutoff) {
        adaptive_beam = max_active_cutoff - best_cost + o.beam_delta;
        cur_cutoff = max_active_cutoff;
        decided = true;
        if (tid == 0) c.counters[5]++;
      }
      if (!decided) {
        if (n_cur > o.min_active) {
          if (o.min_active == 0) min_active_cutoff = best_cost;
          // sorted[min_active] > beam_cutoff  <=>  at most min_active costs lie at or below beam_cutoff
          else if (n_le <= o.min_active) min_active_cutoff = BlockKthSmallest<NT>(c, cur, n_cur, o.min_active, best_cost);
          else min_active_cutoff = beam_cutoff;      // (any value <= beam_cutoff takes the branch below)
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
      const float cost_offset = (n_cur > 0) ? -best_cost : 0.f;
      const float *ll_row = loglikes + (ll_base + f) * ld;
      if (tid == 0) { c.run_min = OrderedBits(INF); c.min_bits = OrderedBits(INF); c.n_cand = 0; c.n_big = 0; c.best_key = RS_EMPTY; }
      for (int i = tid; i < 256; i += NT) c.chist[i] = 0;
      __syncthreads();
      RS_LP(0);
      float local_min = INF;
      // A first bound for the early-out below (an arc at or above "cheapest candidate so far + adaptive beam" cannot end up below
      // the frame's next_cutoff): the best token's own arcs -- the reference starts its next_cutoff the same way (:752-757).  Four
      // dependent loads for one wave: worth it in the frames that make many candidates.
      if (wave == NW - 1 && n_cur > 2048) {
        const int4 bt = cur[best_idx];
        const uint4 bsr = h.state_rec[bt.x];
        float first_bound = INF;
        for (unsigned k = lane; k < bsr.z; k += 64) {
          const int4 arc = arcsf[bsr.x + bsr.y + k];
          first_bound = fminf(first_bound, (__int_as_float(bt.y) + (cost_offset - ll_row[(arc.x & 0x7fffffff) - 1])) + __int_as_float(arc.z));
        }
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) first_bound = fminf(first_bound, __shfl_xor(first_bound, o2, 64));
        if (lane == 0 && first_bound < INF) atomicMin(&c.run_min, OrderedBits(first_bound));
      }
      auto relax_arc = [&](unsigned a, const int4 arc, float lk, float cur_cost, bool is_best, int src_tok) __attribute__((always_inline)) {
        const float graph_cost = __int_as_float(arc.z);
        const float ac_cost = cost_offset - lk;
        const float tot = (cur_cost + ac_cost) + graph_cost;
        if (is_best) {
          const float nw = ((graph_cost + cost_offset) - lk) + cur_cost;      // :752-757, the reference's first bound
          local_min = fminf(local_min, nw);
        }
        local_min = fminf(local_min, tot);
        cnt_arcs++;
        const float bound = FromOrdered(c.run_min) + adaptive_beam;
        if (!(tot < bound)) return;
        cnt_insert++;
        const unsigned ot = OrderedBits(tot);
        if (ot < c.run_min) atomicMin(&c.run_min, ot);
        const int sl = find_or_insert((unsigned)arc.w);
        if (sl < 0) { c.redo = 1; return; }
        atomicMin(&keys[sl], PackKey(tot, a));                 // result unused: non-returning
        const int ci = atomicAdd(&c.n_cand, 1);
        if (ci < cand_cap) cand[ci] = make_int4((int)a, arc.w | (arc.x & (int)0x80000000), (sl << 16) | src_tok, __float_as_int(tot));
        else c.redo = 1;
      };
      {
        constexpr int Q = RS_LIVE_Q;
        for (int ib = tid; ib < n_cur; ib += Q * NT) {
          int2 tk[Q];
          bool act[Q];
          uint4 sr[Q];
          int4 arc[Q][kInline];
          float lk[Q][kInline];
#pragma unroll
          for (int q = 0; q < Q; q++) {
            const int i = ib + q * NT;
            tk[q] = i < n_cur ? *reinterpret_cast<const int2 *>(&cur[i]) : make_int2(0, __float_as_int(INF));
          }
#pragma unroll
          for (int q = 0; q < Q; q++) {
            act[q] = ib + q * NT < n_cur && __int_as_float(tk[q].y) <= cur_cutoff;
            sr[q] = act[q] ? h.state_rec[tk[q].x] : make_uint4(0u, 0u, 0u, 0u);
          }
#pragma unroll
          for (int q = 0; q < Q; q++)
#pragma unroll
            for (int k = 0; k < kInline; k++)
              arc[q][k] = (unsigned)k < sr[q].z ? arcsf[sr[q].x + sr[q].y + k] : make_int4(1, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < Q; q++)
#pragma unroll
            for (int k = 0; k < kInline; k++) lk[q][k] = ll_row[(arc[q][k].x & 0x7fffffff) - 1];
#pragma unroll
          for (int q = 0; q < Q; q++) {
            if (!act[q]) continue;
            const int i = ib + q * NT;
            cnt_expanded++;
#pragma unroll
            for (int k = 0; k < kInline; k++)
              if ((unsigned)k < sr[q].z) relax_arc(sr[q].x + sr[q].y + k, arc[q][k], lk[q][k], __int_as_float(tk[q].y), i == best_idx, i);
            if (sr[q].z > (unsigned)kInline) big[atomicAdd(&c.n_big, 1)] = make_int4((int)(sr[q].x + sr[q].y + kInline), tk[q].y, i, (int)sr[q].z - kInline);
          }
        }
      }
      __syncthreads();
      RS_LP(1);
      // ---- the arcs beyond the first two of the (few) tokens that have them, dealt out over the threads
      {
        const int nb = c.n_big;
        for (int c0 = 0; c0 < nb; c0 += kBigCap) {
          const int nc = nb - c0 < kBigCap ? nb - c0 : kBigCap;
          if (c0 > 0) __syncthreads();
          for (int i = tid; i < nc; i += NT) { const int4 e = big[c0 + i]; c.big_ent[i] = e; c.big_pre[i] = e.w; }
          __syncthreads();
          // exclusive prefix of the degrees in place: a run of consecutive entries per thread
          const int per = (nc + NT - 1) / NT;
          const int i0 = tid * per < nc ? tid * per : nc, i1 = i0 + per < nc ? i0 + per : nc;
          int lsum = 0;
          for (int i = i0; i < i1; i++) lsum += c.big_pre[i];
          int inc = lsum;
#pragma unroll
          for (int o2 = 1; o2 < 64; o2 <<= 1) { const int v = __shfl_up(inc, o2, 64); if (lane >= o2) inc += v; }
          if (lane == 63) c.red_i[wave] = inc;
          __syncthreads();
          int wbase = 0, total = 0;
          for (int wv = 0; wv < NW; wv++) { if (wv < wave) wbase += c.red_i[wv]; total += c.red_i[wv]; }
          int run = wbase + inc - lsum;
          for (int i = i0; i < i1; i++) { const int dgr = c.big_pre[i]; c.big_pre[i] = run; run += dgr; }
          if (tid == 0) c.big_pre[nc] = total;
          __syncthreads();
          for (int jb = tid; jb < total; jb += 4 * NT) {
            unsigned a[4];
            float cc[4];
            bool on[4];
            int tki[4];
            int4 arc[4];
            float lk[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int j = jb + q * NT;
              on[q] = j < total;
              const int jj = on[q] ? j : total - 1;
              int lo = 0, hi = nc;            // last entry with pre[t] <= j
              while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (c.big_pre[mid] <= jj) lo = mid; else hi = mid; }
              const int4 e = c.big_ent[lo];
              a[q] = (unsigned)e.x + (unsigned)(jj - c.big_pre[lo]);
              cc[q] = __int_as_float(e.y);
              tki[q] = e.z;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) arc[q] = arcsf[a[q]];
#pragma unroll
            for (int q = 0; q < 4; q++) lk[q] = ll_row[(arc[q].x & 0x7fffffff) - 1];
#pragma unroll
            for (int q = 0; q < 4; q++) if (on[q]) relax_arc(a[q], arc[q], lk[q], cc[q], tki[q] == best_idx, tki[q]);
          }
        }
      }
      // next_cutoff = min over the candidates of (tot_cost + adaptive_beam): one LDS atomic per wave instead of a block reduction
#pragma unroll
      for (int o2 = 32; o2 > 0; o2 >>= 1) local_min = fminf(local_min, __shfl_xor(local_min, o2, 64));
      if (lane == 0) atomicMin(&c.min_bits, OrderedBits(local_min));
      __syncthreads();
      RS_LP(2);
      const float next_cutoff = FromOrdered(c.min_bits) + adaptive_beam;
      if (tid == 0) {
        finfo[f * 4 + 0] = cost_offset;
        finfo[f * 4 + 1] = cur_cutoff;
        finfo[f * 4 + 2] = next_cutoff;
        finfo[f * 4 + 3] = adaptive_beam;
      }
      if (c.redo) break;                 // workgroup-uniform: read after the barrier of the reduction
      hist_lo = FromOrdered(c.min_bits);
      {
        // the winner of a slot (= the candidate whose key is the slot's) appends the token, complete with back pointer and arc,
        // or -- at or above the final cutoff -- empties the key again; a token whose state has epsilon arcs goes onto the
        // closure's first work list
        constexpr int WB = 4;
        const int nc2 = c.n_cand;
        for (int ib = tid; ib < nc2; ib += WB * NT) {
          int4 cr[WB];
          unsigned long long key[WB];
          uint4 sr[WB];
#pragma unroll
          for (int q = 0; q < WB; q++) { const int i = ib + q * NT; cr[q] = cand[i < nc2 ? i : nc2 - 1]; }
#pragma unroll
          for (int q = 0; q < WB; q++) key[q] = LoadKey(&keys[(unsigned)cr[q].z >> 16]);
#pragma unroll
          for (int q = 0; q < WB; q++) sr[q] = cr[q].y < 0 ? h.state_rec[cr[q].y & 0x7fffffff] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
          for (int q = 0; q < WB; q++) {
            if (ib + q * NT >= nc2 || (unsigned)(key[q] & 0xFFFFFFFFull) != (unsigned)cr[q].x) continue;
            const int sl = (int)((unsigned)cr[q].z >> 16);
            if (KeyCost(key[q]) < next_cutoff) {
              const int idx = atomicAdd(&c.n_next, 1);
              if (idx < next_cap) {
                next_toks[idx] = make_int4(cr[q].y & 0x7fffffff, sl, cr[q].z & 0xFFFF, cr[q].x);
                slot_tok[sl] = idx;
                if (sr[q].y != 0) {
                  const int qp = atomicAdd(&c.q_n[0], 1);
                  if (qp < qcap) {
                    queue[0][qp] = make_int4(sl | (kNoBp << 16), (int)sr[q].x, (int)(unsigned)(key[q] & 0xFFFFFFFFull), (int)(unsigned)(key[q] >> 32));
                    queue_ne[0][qp] = (int)sr[q].y;
                  } else c.redo = 1;
                }
              } else {
                c.overflow = 1;
              }
            } else {
              StoreKey(&keys[sl], RS_EMPTY);
            }
          }
        }
      }
      closure_cutoff = next_cutoff;
      __syncthreads();
      RS_LP(3);
    }
    // ================================================================ ProcessNonemitting(closure_cutoff)
    {
      int qi = 0;      // (queue[0] was filled by the winners' pass / InitDecoding; c.q_n[1] is 0)
      int guard_rounds = 0;
      while (c.q_n[qi] > 0) {
        const int qn = c.q_n[qi] < qcap ? c.q_n[qi] : qcap;
        __syncthreads();
        if (tid == 0) c.q_n[qi ^ 1] = 0;
        __syncthreads();
        for (int ib = 0; ib < qn; ib += NT) {
          const int i = ib + tid;
          const bool have = i < qn;
          const int4 qe = have ? queue[qi][i] : make_int4(0, 0, 0, 0);
          const int ne_in = have ? queue_ne[qi][i] : 0;
          const int sl = qe.x & 0xFFFF, wtok = (int)((unsigned)qe.x >> 16);
          const unsigned long long ekey = ((unsigned long long)(unsigned)qe.w << 32) | (unsigned)qe.z;
          const unsigned long long key = have ? LoadKey(&keys[sl]) : RS_EMPTY;
          const int my_tok = have ? slot_tok[sl] : 0;
          const bool fresh = have && key == ekey;      // else: the slot was improved again, the entry of that improvement is on a list too
          if (fresh && wtok != kNoBp) next_toks[my_tok].z = wtok;      // the token's back pointer: the token that relaxed the winning epsilon arc
          const float cur_cost = KeyCost(key);
          const bool live = fresh && ne_in > 0 && cur_cost < closure_cutoff;
          if (live) cnt_expanded++;
          const unsigned a0 = (unsigned)qe.y, ne = live ? (unsigned)ne_in : 0u;
          unsigned ne_max = ne;
#pragma unroll
          for (int o2 = 32; o2 > 0; o2 >>= 1) ne_max = max(ne_max, (unsigned)__shfl_xor((int)ne_max, o2, 64));
          for (unsigned k = 0; k < ne_max; k++) {
            const bool has_arc = k < ne;
            const unsigned a = a0 + (has_arc ? k : 0u);
            const int4 arc = has_arc ? arcsf[a] : make_int4(0, 0, 0, 0);
            const float tot = cur_cost + __int_as_float(arc.z);
            if (has_arc) cnt_arcs++;
            const bool act = has_arc && tot < closure_cutoff;
            if (act) cnt_insert++;
            const unsigned long long m = __ballot(act);
            if (m == 0ull) continue;
            // (thousands of history states back off into ONE unigram state: a wave whose relaxing lanes all target the same
            // state reduces its keys first and issues a single atomic)
            const int first = __ffsll((long long)m) - 1;
            const int d0 = __shfl(arc.w, first, 64);
            const bool uniform = __ballot(act && arc.w != d0) == 0ull;
            unsigned long long nkey = act ? PackKey(tot, a) : RS_EMPTY;
            bool mine = act;
            if (uniform && __popcll(m) > 1) {
              unsigned long long kmin = nkey;
#pragma unroll
              for (int o2 = 32; o2 > 0; o2 >>= 1) {
                const unsigned lo32 = (unsigned)__shfl_xor((int)(unsigned)(kmin & 0xFFFFFFFFull), o2, 64);
                const unsigned hi32 = (unsigned)__shfl_xor((int)(unsigned)(kmin >> 32), o2, 64);
                const unsigned long long other = ((unsigned long long)hi32 << 32) | lo32;
                kmin = other < kmin ? other : kmin;
              }
              mine = act && nkey == kmin;          // keys are unique (arc ids): exactly one lane
            }
            if (mine) {
              const int sl2 = find_or_insert((unsigned)arc.w);
              if (sl2 < 0) { c.redo = 1; continue; }
              const bool dst_eps = arc.x < 0;
              const unsigned long long old = atomicMin(&keys[sl2], nkey);
              const uint4 dsr = dst_eps ? h.state_rec[arc.w] : make_uint4(0u, 0u, 0u, 0u);
              if (old == RS_EMPTY) {              // FindOrAddToken made a token: its back pointer is this lane's token
                const int idx = atomicAdd(&c.n_next, 1);
                if (idx < next_cap) { next_toks[idx] = make_int4(arc.w, sl2, my_tok, -2); slot_tok[sl2] = idx; }
                else c.overflow = 1;
              }
              // work-list entry: for a state with epsilon arcs so that they are followed, for an improved token so that its
              // back pointer is rewritten (a new token already carries it)
              if ((old == RS_EMPTY && dst_eps) || (old != RS_EMPTY && nkey < old)) {
                const int qp = atomicAdd(&c.q_n[qi ^ 1], 1);
                if (qp < qcap) {
                  queue[qi ^ 1][qp] = make_int4(sl2 | ((old == RS_EMPTY ? kNoBp : my_tok) << 16), (int)dsr.x, (int)(unsigned)(nkey & 0xFFFFFFFFull), (int)(unsigned)(nkey >> 32));
                  queue_ne[qi ^ 1][qp] = (int)dsr.y;
                } else c.redo = 1;
              }
            }
          }
        }
        __syncthreads();
        qi ^= 1;
        if (c.redo) break;
        if (++guard_rounds > 100000) { if (tid == 0) c.error = 2; break; }   // epsilon cycle in the graph
      }
      __syncthreads();
      if (tid == 0) { c.q_n[0] = 0; c.q_n[1] = 0; }
      if (c.redo) break;
    }
    RS_LP(4);
    // ================================================================ complete the tokens of frame f+1
    {
      const int nn = c.n_next < next_cap ? c.n_next : next_cap;
      const float hscale = 256.0f / (closure_cutoff - hist_lo);
      float lv = INF;                                     // cheapest token of the new frame (lowest index on ties)
      int li = 0x7fffffff;
      constexpr int MB = 4;                               // tokens per thread and trip, both load stages of all of them in flight together
      for (int ib = tid; ib < nn; ib += MB * NT) {
        int slot[MB];
        unsigned long long key[MB];
#pragma unroll
        for (int q = 0; q < MB; q++) { const int i = ib + q * NT; slot[q] = i < nn ? next_toks[i].y : 0; }
#pragma unroll
        for (int q = 0; q < MB; q++) key[q] = LoadKey(&keys[slot[q]]);
#pragma unroll
        for (int q = 0; q < MB; q++) {
          const int i = ib + q * NT;
          if (i >= nn) continue;
          const float cst = KeyCost(key[q]);
          next_toks[i].y = __float_as_int(cst);
          next_toks[i].w = (int)(unsigned)(key[q] & 0xFFFFFFFFull);
          StoreKey(&keys[slot[q]], RS_EMPTY);
          atomicAdd(&c.chist[CostBin(cst, hist_lo, hscale)], 1u);
          if (cst < lv || (cst == lv && i < li)) { lv = cst; li = i; }
        }
      }
      {
        unsigned long long bk = li == 0x7fffffff ? RS_EMPTY : PackKey(lv, (unsigned)li);
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) {
          const unsigned lo32 = (unsigned)__shfl_xor((int)(unsigned)(bk & 0xFFFFFFFFull), o2, 64);
          const unsigned hi32 = (unsigned)__shfl_xor((int)(unsigned)(bk >> 32), o2, 64);
          const unsigned long long other = ((unsigned long long)hi32 << 32) | lo32;
          bk = other < bk ? other : bk;
        }
        if (lane == 0 && bk != RS_EMPTY) atomicMin(&c.best_key, bk);      // (reset in the prologue of ProcessEmitting, behind barriers)
      }
      // the table starts the next frame empty (every key a slot held was emptied by its owner above or in the winners' pass)
      for (int i = tid; i < HS / 4; i += NT) reinterpret_cast<uint4 *>(tags)[i] = make_uint4(kFree, kFree, kFree, kFree);
      if (c.g_used) for (int i = tid; i < kGlobalSize; i += NT) __hip_atomic_store(&gtags[i], kFree, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      best_cost = c.best_key == RS_EMPTY ? INF : KeyCost(c.best_key);
      best_idx = (int)(unsigned)(c.best_key & 0xFFFFFFFFull);
      off_cur = off_next;
      n_cur = nn;
      off_next = off_cur + n_cur;
      const bool stop = c.overflow != 0 || c.error != 0 || nn == 0;
      __syncthreads();
      if (tid == 0) {
        frame_off[f + 2] = off_next;
        c.counters[3] += (unsigned long long)nn;
        {
          const unsigned long long mt = c.counters[4] & 0xFFFFFFFFull, mc = c.counters[4] >> 32;
          const unsigned long long nc_ = f >= 0 ? (unsigned long long)c.n_cand : 0ull;
          c.counters[4] = ((nc_ > mc ? nc_ : mc) << 32) | ((unsigned long long)nn > mt ? (unsigned long long)nn : mt);
        }
        c.n_next = 0;
        c.n_slots = 0;
        c.g_used = 0;
        if (c.overflow) c.redo = 1;                    // a frame of more than kSlotCap tokens: the dense-table kernel has room
        if (nn == 0 && c.error == 0) c.error = 1;      // "no surviving tokens"
      }
      RS_LP(5);
      if (stop) break;
    }
  }
  __syncthreads();
#ifdef RS_DECODE_PROFILE
  if (tid == 0 && T > 0)
    printf("live block %d: %lld cycles, %d tokens, T=%d phases cutoff %lld sweep %lld big %lld winners %lld closure %lld complete %lld\n", u,
           prof[0] + prof[1] + prof[2] + prof[3] + prof[4] + prof[5], off_next, T, prof[0], prof[1], prof[2], prof[3], prof[4], prof[5]);
#endif
  if (c.redo) {                       // workgroup-uniform
    if (tid == 0) { w.redo[u] = 1; w.out_nwords[u] = -1; }
    return;
  }
  // ================================================================ final costs + best-path traceback
  {
    const int4 *cur = tokens + off_cur;
    float lv1 = INF, lv2 = INF;
    int li1 = 0x7fffffff, li2 = 0x7fffffff;
    for (int i = tid; i < n_cur; i += NT) {
      const int2 t2 = *reinterpret_cast<const int2 *>(&cur[i]);
      const float cst = __int_as_float(t2.y);
      const float wf = cst + h.final_cost[t2.x];
      if (wf < lv1 || (wf == lv1 && i < li1)) { lv1 = wf; li1 = i; }
      if (cst < lv2 || (cst == lv2 && i < li2)) { lv2 = cst; li2 = i; }
    }
    float b1, b2;
    int i1, i2;
    BlockMinArg<NT>(c, lv1, li1, &b1, &i1);
    BlockMinArg<NT>(c, lv2, li2, &b2, &i2);
    atomicAdd(&c.counters[0], (unsigned long long)cnt_expanded);
    atomicAdd(&c.counters[1], (unsigned long long)cnt_arcs);
    atomicAdd(&c.counters[2], (unsigned long long)cnt_insert);
    __syncthreads();
    const bool reached = b1 < INF;
    // The walk itself (one dependent load per hop) is thread 0's; what every hop adds -- the arc's weight, its frame's
    // log-likelihood, its word -- is looked up by all threads a block of hops at a time.
    constexpr int kHops = kBigCap;                 // (arc, frame of the acoustic score) pairs per block, in c.big_ent's LDS
    int2 *hops = reinterpret_cast<int2 *>(c.big_ent);
    int *words = w.out_words + (size_t)u * w.max_words;
    double graph = 0.0, ac = 0.0;                  // (thread-local partial sums, reduced at the end)
    int nw = 0;                                    // (wave 0's lanes all hold it)
    bool truncated = false;
    if (tid == 0) { c.bcast_i[0] = reached ? i1 : i2; c.bcast_i[1] = c.error ? -1 : T; c.bcast_i[2] = 0; }
    __syncthreads();
    bool done = c.error != 0 || n_cur == 0;
    while (!done) {
      if (tid == 0) {
        // one round trip per hop: whether the hop's arc is emitting (sign of arc_srcx) decides in which frame the next token lies,
        // so both candidates and the offset of the frame below are requested together with it
        int idx = c.bcast_i[0], F = c.bcast_i[1], n = 0;
        bool end = false;
        int fo0 = frame_off[F], fo1 = F > 0 ? frame_off[F - 1] : 0;
        int4 tk = tokens[fo0 + idx];
        while (n < kHops) {
          if (tk.w < 0) { end = true; break; }
          const int arc_id = tk.w;
          const int sx = h.arc_srcx[arc_id];
          const int4 ta = tokens[fo0 + tk.z], tb = tokens[fo1 + tk.z];      // (fo1 + tk.z stays inside the utterance's token array)
          const int fo2 = F > 1 ? frame_off[F - 2] : 0;
          const bool emitting = sx >= 0;
          idx = tk.z;
          if (emitting) { F -= 1; fo0 = fo1; fo1 = fo2; tk = tb; } else { tk = ta; }
          hops[n++] = make_int2(arc_id, emitting ? F : -1);
        }
        c.bcast_i[0] = idx; c.bcast_i[1] = F; c.bcast_i[2] = n; c.bcast_i[3] = end ? 1 : 0;
      }
      __syncthreads();
      const int n = c.bcast_i[2];
      done = c.bcast_i[3] != 0;
      for (int i = tid; i < n; i += NT) {
        const int2 hp = hops[i];
        const int4 arc = h.arcs[hp.x];
        graph += (double)__int_as_float(arc.z);
        if (hp.y >= 0) {
          const float off = finfo[hp.y * 4 + 0];
          const float lk = loglikes[(ll_base + hp.y) * ld + (arc.x - 1)];
          const float link_ac = off - lk;                   // ForwardLink::acoustic_cost
          ac += (double)(link_ac - off);                    // GetRawLattice :166-172
        }
        hops[i].y = arc.y;                                  // the word label replaces the frame
      }
      __syncthreads();
      if (tid < 64) {      // wave 0: ordered compaction of the word labels (last word first)
        for (int base = 0; base < n; base += 64) {
          const int i = base + tid;
          const int wl = i < n ? hops[i].y : 0;
          const unsigned long long m = __ballot(wl != 0);
          if (wl != 0) {
            const int pos = nw + __popcll(m & ((1ull << tid) - 1ull));
            if (pos < w.max_words) words[pos] = wl; else truncated = true;
          }
          nw += __popcll(m);
        }
      }
      __syncthreads();
    }
    // block sums in double
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) { graph += __shfl_xor(graph, o2, 64); ac += __shfl_xor(ac, o2, 64); }
    double *dsum = reinterpret_cast<double *>(c.kth_cand);
    if (lane == 0) { dsum[wave * 2] = graph; dsum[wave * 2 + 1] = ac; }
    truncated = __any(truncated) || nw > w.max_words;
    __syncthreads();
    if (tid < 64) {
      const int nwc = nw < w.max_words ? nw : w.max_words;
      for (int a = tid; a < nwc / 2; a += 64) { const int b = nwc - 1 - a; const int t2 = words[a]; words[a] = words[b]; words[b] = t2; }
    }
    if (tid == 0) {
      double gsum = 0.0, asum = 0.0;
      for (int k = 0; k < NW; k++) { gsum += dsum[k * 2]; asum += dsum[k * 2 + 1]; }
      if (!c.error && n_cur > 0 && reached) gsum += (double)h.final_cost[cur[i1].x];
      w.out_nwords[u] = (c.error || truncated) ? -1 : nw;
      float *oc = w.out_costs + (size_t)u * 4;
      oc[0] = (float)gsum;
      oc[1] = (float)asum;
      oc[2] = reached ? b1 : b2;
      oc[3] = reached ? 1.f : 0.f;
      c.counters[7] = 2ull * (unsigned long long)c.error;
      long long *ctr = w.counters + (size_t)u * 8;
      for (int i = 0; i < 8; i++) ctr[i] = (long long)c.counters[i];
      frame_off[T + 1] = off_next;
    }
  }
}

}  // namespace

bool DecodeLiveUsable(const HclgDev &h) { return h.num_states > 0 && h.arcs_f != nullptr; }
int DecodeLiveSlotCap() { return kSlotCap; }
int DecodeLiveGlobalTable() { return kGlobalSize; }
// length of the slot-indexed arrays: the compact form numbers slots consecutively (< kSlotCap); the position-addressed form uses
// LDS positions [0, 2^HLOG) and global positions behind them
int DecodeLiveTableSize() { return (1 << 15) + kGlobalSize; }

namespace {
template <int NT, int HLOG>
void LaunchLive(bool wide, const HclgDev &h, const DecodeOptsDev &o, const BatchGeom &g, const float *loglikes, int ld, const DecodeWork &w, hipStream_t s) {
  if (wide) hipLaunchKernelGGL((LiveDecodeKernel<NT, HLOG, true>), dim3(g.n_utts), dim3(NT), 0, s, h, o, g, loglikes, ld, w);
  else hipLaunchKernelGGL((LiveDecodeKernel<NT, HLOG, false>), dim3(g.n_utts), dim3(NT), 0, s, h, o, g, loglikes, ld, w);
}
}  // namespace

void LaunchDecodeLive(const HclgDev &h, const DecodeOptsDev &o, const BatchGeom &g, const float *loglikes, int ld,
                      const DecodeWork &w, hipStream_t s) {
  if (g.n_utts == 0) return;
  // graphs whose state ids do not fit beside a slot number in one table word use the position-addressed form (RS_HASH_WIDE=1 forces it)
  const char *we = std::getenv("RS_HASH_WIDE");          // (read per launch: a test flips it)
  const bool force_wide = we && std::atoi(we) != 0;
  const bool wide = force_wide || (unsigned)h.num_states >= (1u << (32 - kSlotBits)) - 1u;
  // shape: threads per utterance / LDS entries.  512 / 16 K (two workgroups per CU) unless RS_LIVE_SHAPE says otherwise
  static const int shape = [] { const char *e = std::getenv("RS_LIVE_SHAPE"); return e ? std::atoi(e) : 512; }();
  if (shape == 1024) LaunchLive<1024, 15>(wide, h, o, g, loglikes, ld, w, s);
  else if (shape == 256) LaunchLive<256, 13>(wide, h, o, g, loglikes, ld, w, s);
  else LaunchLive<512, 14>(wide, h, o, g, loglikes, ld, w, s);
}

}  // namespace rs
