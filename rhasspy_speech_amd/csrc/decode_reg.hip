// Register-resident token passing for small grammar graphs (the rhasspy use case: a few hundred to a few
// thousand HCLG states, a few thousand arcs).
//
// Same search and same results as decode_dense.hip / decode_kernels.hip (reference:
// lattice-faster-decoder.cc:56-73,644-887).  Observation that drives the design: with a dense state table the set
// of arcs a workgroup has to evaluate is the same on every frame.  So the ARCS are dealt out to the threads once
// (arc i -> thread i % NT) and stay in VGPRs for the whole utterance: {LDS address of the source cost, LDS address
// of the destination key, pdf, weight, arc id}; the log-likelihoods of exactly those arcs are fetched from HBM/L2
// one frame ahead.  A frame is then
//   * one pass over the registers: independent LDS gathers of the source costs, a few VALU ops per arc and one
//     non-returning 64-bit LDS atomic min per surviving arc (key = ordered cost bits << 32 | arc id, so the
//     result does not depend on the order in which arcs arrive),
//   * the same for the epsilon arcs, repeated to the fixpoint (the number of rounds is the epsilon depth of the
//     graph, known on the host, so no convergence vote is needed for acyclic epsilon subgraphs),
//   * a pass over the states that commits keys -> costs / back-pointers and collects the next frame's statistics,
//   * three block reductions done with DPP row rotations + readlane (no ds_bpermute chains) and ONE barrier each.
// Everything in the frame loop is written branch-free on purpose: hipcc turns conditional loads and short-circuit
// conditions into exec-mask branches with a wait in front of each, which is what made the first version slow.
// Compiled with -ffp-contract=off.
#include "decode_common.h"
#include "decode_tok.h"
#include "wave_ops.h"
#include "env.h"

#include <string>

#include <cstdlib>

namespace rs {
using namespace dd;

#ifdef RS_DECODE_PROFILE
#define RS_T(i) do { long long _n = clock64(); if (tid == 0) prof[i] += _n - t_last; t_last = _n; } while (0)
#define RS_T2(i) do { long long _n = clock64(); prof2[i] += _n - t2_last; t2_last = _n; } while (0)
#else
#define RS_T(i) do { } while (0)
#define RS_T2(i) do { } while (0)
#endif


template <int NT, int KE, int KX>
__global__ __launch_bounds__(NT) void RegDecodeKernel(HclgDev h, RegGraphDev rg, DecodeOptsDev o, BatchGeom g,
                                                      const float *__restrict__ loglikes, int ld, DenseWork w, int smem_bytes,
                                                      int f_begin, int f_end) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // the search is a chain of short dependent steps: when it shares a CU with GEMM waves (pipelined output layer), its
  // instructions should win the issue arbitration
#ifndef RS_REG_PRIO
#define RS_REG_PRIO 3
#endif
  __builtin_amdgcn_s_setprio(RS_REG_PRIO);
  constexpr int NW = NT / 64;
  __shared__ Red<NW> red;
  __shared__ int4 xr[2][NW];       // cross-wave exchange, ping-pong so that a reduction needs one barrier
  // 256-bin histograms of the committed frame's costs, filled by the commit pass for the NEXT frame's GetCutoff (two, alternating:
  // the one the current frame has read is cleared by the pass that fills the other)
  __shared__ __attribute__((aligned(16))) unsigned hist2[2][256];
  int rb = 0;
  const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef RS_DEBUG_SEARCH_FRAMES      // timing experiments only: the search stops after that many frames of every utterance
  const int T = min(g.d_num_frames[u], RS_DEBUG_SEARCH_FRAMES), S = h.num_states;
#else
  const int T = g.d_num_frames[u], S = h.num_states;
#endif
  // time slab [f_begin, f_end): an utterance is started by the slab with f_begin == -1, resumed from w.state_cost by later
  // ones, and finished (traceback, results) by the slab that holds its last frame.  Streams bring their own window per
  // utterance and say explicitly when the stream ends (T is then the number of frames that exist so far).
  const bool windows = w.win_begin != nullptr;
  if (windows) { f_begin = w.win_begin[u]; f_end = w.win_end[u]; }
  const bool finishing = windows ? w.win_final[u] != 0 : f_end >= T;
  if (windows ? (f_begin >= f_end && !finishing) : (f_begin >= 0 && f_begin >= T)) return;
  const int f_stop = f_end < T ? f_end : T;
  const size_t slot = windows ? (size_t)w.slot[u] : (size_t)u;
  const size_t frame_row0 = windows ? (size_t)w.pool_row[u] : (size_t)u * (g.max_frames + 1);
  float *state = w.state_cost + slot * (2 * (size_t)S + 4);
  float *cost_cur = reinterpret_cast<float *>(smem);                                       // [S + 1], [S] = +inf forever
  unsigned long long *key_next = reinterpret_cast<unsigned long long *>(smem + rg.key_base);   // [S + 1], [S] = empty forever
  int *bp = w.bp + frame_row0 * S;
  float *finfo = w.frame_info + frame_row0 * 4;
  const float INF = INFINITY;
  // (the utterance's first row through readfirstlane: loaded by a vector instruction it stays in a VGPR, and with it the 64-bit row
  // address of every frame -- eight VALU instructions per frame and a 64-bit add per load instead of base register + 32-bit offset)
  const size_t ll_base = (size_t)__builtin_amdgcn_readfirstlane(g.d_row_base[u]) + g.L;
  unsigned n_expanded = 0, n_arcs = 0, n_insert = 0, n_alive = 0;      // (a thread's share of one launch: 32 bits; as 64-bit counters, two instructions per arc and frame)
  int max_active_frames = 0, min_active_frames = 0;

  // ---- my arcs, in registers for the whole utterance
  int4 ea[KE];         // {src cost addr | dst key addr << 16, pdf * 4 (a byte offset into the frame's row), weight bits, forward arc index}
  int4 xa[KX];         // {src key-high-word addr | dst key addr << 16, -, weight bits, forward arc index}
#pragma unroll
  for (int a = 0; a < KE; a++) { ea[a] = rg.e_tab[(size_t)a * NT + tid]; ea[a].y <<= 2; }
#pragma unroll
  for (int a = 0; a < KX; a++) xa[a] = rg.x_tab[(size_t)a * NT + tid];
  for (int s = tid; s <= S; s += NT) { cost_cur[s] = (f_begin >= 0 && s < S) ? state[s] : INF; key_next[s] = RS_EMPTY; }
  for (int i = tid; i < 256; i += NT) { red.hist[i] = 0; hist2[0][i] = 0; hist2[1][i] = 0; }      // (red.hist: KthFromHist's invariant)
  if (tid == 0) red.ncand = 0;
  // log-likelihoods of my emitting arcs, fetched one frame ahead (padding arcs read pdf 0 and never pass the cutoff)
  const int f_first = f_begin < 0 ? 0 : f_begin;
  float llv[KE];
#pragma unroll
  for (int a = 0; a < KE; a++) llv[a] = 0.f;
  if (f_first < T) {
    const float *row = loglikes + (ll_base + f_first) * ld;
#pragma unroll
    for (int a = 0; a < KE; a++) llv[a] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(row) + (unsigned)ea[a].y);
  }
  __syncthreads();
  if (f_begin < 0 && tid == 0) key_next[h.start] = PackKey(0.0f, RS_NOARC);
  float closure_cutoff = f_begin < 0 ? o.beam : state[S];
  int error = f_begin < 0 ? 0 : (int)state[S + 1];
  // the committed frame's histogram: hist2[hpar], bins = HistBin(cost) over [hist_lo, the cutoff it was committed against)
  // (hist_ok: there is one -- a resumed slab starts without)
  int hpar = 0;
  bool hist_ok = false, hist_open = false;
  float hist_lo = 0.f, hist_scale = 0.f, hist_reach = 1.f;
  auto HistBin = [&](float c) -> int {
    const int b = (int)((c - hist_lo) * hist_scale);
    return b < 0 ? 0 : (b > 255 ? 255 : b);
  };
  // statistics of the committed frame, collected by the commit pass (recomputed when resuming)
  float st_min = INF;
  int st_arg = 0x7fffffff, st_cnt = 0;
  if (f_begin >= 0) {
    for (int s = tid; s < S; s += NT) {
      const float c = cost_cur[s];
      const bool alive = c < INF;
      st_cnt += (int)alive;
      const bool better = alive & (c < st_min);
      st_min = better ? c : st_min;
      st_arg = better ? s : st_arg;
    }
  }
  __syncthreads();
#ifdef RS_DECODE_PROFILE
  long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long prof2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t_last = clock64(), t2_last = t_last;
#endif

  for (int f = f_begin; f < f_stop && !error; f++) {
    if (f >= 0) {
      // ---- best token (ties: smallest state), token count
      float best_cost;
      int best_state, N;
      {
        const unsigned ub = wv::FloatToOrdered(st_min);
        const unsigned wm = wv::MinU(ub);
        const unsigned long long tie = __ballot(ub == wm);
        unsigned wa;
        if (__popcll(tie) == 1) wa = (unsigned)__builtin_amdgcn_readlane(st_arg, __ffsll((long long)tie) - 1);
        else wa = wv::MinU(ub == wm ? (unsigned)st_arg : 0x7fffffffu);
        const int wn = wv::Sum(st_cnt);
        if (lane == 0) xr[rb][wave] = make_int4((int)wm, (int)wa, wn, 0);
        LdsBarrier();
        unsigned long long bk = ~0ull;
        N = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) {
          const int4 e = xr[rb][k];
          const unsigned long long kk = ((unsigned long long)(unsigned)e.x << 32) | (unsigned)e.y;
          bk = kk < bk ? kk : bk;
          N += e.z;
        }
        rb ^= 1;
        best_cost = wv::OrderedToFloat((unsigned)(bk >> 32));
        best_state = (int)(unsigned)(bk & 0xFFFFFFFFull);
      }
      if (N == 0) { error = 1; break; }
      RS_T(0);
      // ---- GetCutoff (lattice-faster-decoder.cc:644-711), order statistic FIRST: "more than max_active tokens are below the beam
      // cutoff" is "the max_active-th smallest cost is below it", and "at most min_active tokens are at or below it" is "the
      // min_active-th smallest cost is above it" -- so no pass counts tokens.  The statistic comes out of the histogram the commit
      // pass left: every wave scans it for itself, the handful of costs in the bin that holds the rank are collected (one barrier)
      // and every wave ranks them for itself.  Without a usable histogram (infinite cutoff last frame, a crowded bin, a second
      // statistic in one frame): the exact selection of rounds 2-3 (KthFromHist, decode_common.h).
      const float beam_cutoff = best_cost + o.beam;
      const bool over_max = N > o.max_active, over_min = N > o.min_active;      // wave-uniform
      // Every live token passed `cost < closure_cutoff` when the previous frame was committed.  When that cutoff is not
      // above this frame's beam cutoff (the usual case: best + beam on both sides), all N tokens are inside the beam and
      // min-active cannot bind.
      const bool all_inside = closure_cutoff <= beam_cutoff;
      bool fast_used = false;
      auto Kth = [&](int k, int *n_le) -> float {
#ifdef RS_DECODE_PROFILE
        prof[7] += 1;
#endif
        if (hist_ok && !fast_used) {
          RS_T2(0);
          const uint4 hv = *reinterpret_cast<const uint4 *>(&hist2[hpar][4 * lane]);
          const int h0 = (int)hv.x, h1 = (int)hv.y, h2 = (int)hv.z, h3 = (int)hv.w;
          const int tot = h0 + h1 + h2 + h3;
          const int inc = WaveScanIncl(tot), exc = inc - tot;
          const bool hit = (exc <= k) & (k < inc);
          int bb = 4 * lane, acc = exc, m = h0;
          if (acc + h0 <= k) { acc += h0; bb++; m = h1; if (acc + h1 <= k) { acc += h1; bb++; m = h2; if (acc + h2 <= k) { acc += h2; bb++; m = h3; } } }
          const unsigned long long hm = __ballot(hit);
          const int hl = hm ? __ffsll((long long)hm) - 1 : 0;
          const int bin = __builtin_amdgcn_readlane(bb, hl), before = __builtin_amdgcn_readlane(acc, hl), cnt = __builtin_amdgcn_readlane(m, hl);
          if (hist_open && hm != 0ull) {       // keep the rank in the middle of the bins
            if (bin >= 192 && hist_reach < 1024.f) hist_reach *= 2.f;
            else if (bin < 64 && hist_reach > 0.03125f) hist_reach *= 0.5f;
          }
          if (hm != 0ull && cnt <= 64) {
            fast_used = true;
            RS_T2(1);
            {
              constexpr int kCS = 3;      // (the costs first, then the appends: as in the commit pass)
              float cq[kCS];
#pragma unroll
              for (int q = 0; q < kCS; q++) { const int s = tid + q * NT; cq[q] = cost_cur[s < S ? s : S]; }      // (cost_cur[S] = +inf)
#pragma unroll
              for (int q = 0; q < kCS; q++) if (cq[q] < INF && HistBin(cq[q]) == bin) red.cand[atomicAdd(&red.ncand, 1)] = cq[q];
              for (int s = tid + kCS * NT; s < S; s += NT) {
                const float c = cost_cur[s];
                if (c < INF && HistBin(c) == bin) red.cand[atomicAdd(&red.ncand, 1)] = c;
              }
            }
            LdsBarrier();           // (red.ncand goes back to zero behind the arc pass's barrier)
            RS_T2(2);
            const int kk = k - before;
            const float v = lane < cnt ? red.cand[lane] : INF;
            int lt = 0, le = 0;
            for (int j = 0; j < cnt; j++) {
              const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
              lt += (int)(x < v);
              le += (int)(x <= v);
            }
            const unsigned long long sel = __ballot((lane < cnt) & (lt <= kk) & (kk < le));
            const int sl = sel ? __ffsll((long long)sel) - 1 : 0;
            *n_le = before + __builtin_amdgcn_readlane(le, sl);
            RS_T2(3);
#ifdef RS_DECODE_PROFILE
            prof2[4] += cnt; prof2[5] += 1;
#endif
            return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), sl));
          }
        }
#ifdef RS_DECODE_PROFILE
        if (!hist_ok) prof2[6] += 1; else prof2[7] += 1;
#endif
        if (fast_used) { LdsBarrier(); if (tid == 0) red.ncand = 0; }
        float hi = closure_cutoff;
        if (!(hi < INF)) {
          float mx = -INF;
          for (int s = tid; s < S; s += NT) { const float c = cost_cur[s]; mx = c < INF ? fmaxf(mx, c) : mx; }
          const unsigned wx = wv::MaxU(wv::FloatToOrdered(mx));
          if (lane == 0) xr[rb][wave] = make_int4((int)wx, 0, 0, 0);
          LdsBarrier();
          unsigned bx = 0u;
#pragma unroll
          for (int q = 0; q < NW; q++) bx = max(bx, (unsigned)xr[rb][q].x);
          rb ^= 1;
          hi = wv::OrderedToFloat(bx);
        }
        const float hscale = hi > best_cost ? 255.0f / (hi - best_cost) : 0.f;
        for (int s = tid; s < S; s += NT) {
          const float c = cost_cur[s];
          if (c < INF) atomicAdd(&red.hist[KthBin(c, best_cost, hscale)], 1u);
        }
        LdsBarrier();
        return KthFromHist<NT>(red, cost_cur, S, k, best_cost, hi, n_le);
      };
      float cur_cutoff, adaptive_beam;
      int n_exp;
      bool decided = false;
      if (over_max) {
        int le;
        const float v = Kth(o.max_active, &le);
        if (v < beam_cutoff) {       // more than max_active tokens below the beam cutoff
          adaptive_beam = v - best_cost + o.beam_delta;
          cur_cutoff = v;
          n_exp = le;
          max_active_frames++;
          decided = true;
        }
      }
      if (!decided) {
        float min_active_cutoff = INF;
        bool loosened;
        int kth_le = 0;
        if (over_min) {
          min_active_cutoff = best_cost;           // best_cost stands for "tmp[min_active] <= beam_cutoff"
          if (o.min_active != 0 && !all_inside) {
            const float v = Kth(o.min_active, &kth_le);
            if (v > beam_cutoff) min_active_cutoff = v;      // fewer than min_active + 1 tokens at or below the beam cutoff
          }
          loosened = min_active_cutoff > beam_cutoff;
        } else {
          loosened = true;      // fewer than min_active tokens: the cutoff stays +inf (:691-705)
        }
        if (loosened) {
          adaptive_beam = min_active_cutoff - best_cost + o.beam_delta;
          cur_cutoff = min_active_cutoff;
          n_exp = over_min ? kth_le : N;
          if (over_min) min_active_frames++;
        } else {
          adaptive_beam = o.beam;
          cur_cutoff = beam_cutoff;
          n_exp = N;
          if ((over_max || over_min) && !all_inside) {      // (a statistic only: the tokens at or below the beam cutoff)
            int c_le = 0;
            for (int s = tid; s < S; s += NT) c_le += (int)(cost_cur[s] <= beam_cutoff);
            const int wa = wv::Sum(c_le);
            if (lane == 0) xr[rb][wave] = make_int4(wa, 0, 0, 0);
            LdsBarrier();
            n_exp = 0;
#pragma unroll
            for (int q = 0; q < NW; q++) n_exp += xr[rb][q].x;
            rb ^= 1;
          }
        }
      }
      const float cost_offset = -best_cost;
      if (tid == 0) n_expanded += (unsigned)n_exp;
      RS_T(1);
      // ---- ProcessEmitting: independent LDS gathers, then one LDS atomic min per surviving arc
      float csrc[KE];
#pragma unroll
      for (int a = 0; a < KE; a++) csrc[a] = *reinterpret_cast<const float *>(smem + (ea[a].x & 0xFFFF));
      float local_min = INF;
      const int best_addr = best_state * 4;
#pragma unroll
      for (int a = 0; a < KE; a++) {
        const float c = csrc[a];
        const bool pass = (c < INF) & (c <= cur_cutoff);
        const float lk = llv[a];
        const float gc = __int_as_float(ea[a].z);
        const float tot = (c + (cost_offset - lk)) + gc;
        const float alt = ((gc + cost_offset) - lk) + c;                           // :752-757, arcs of the best token
        const float m = ((ea[a].x & 0xFFFF) == best_addr) ? fminf(tot, alt) : tot;   // (under a wave-uniform branch instead: the arc pass 1560 -> 2290 clocks)
        local_min = fminf(local_min, pass ? m : INF);
        n_arcs += (unsigned)pass;
        if (pass)
          atomicMin(reinterpret_cast<unsigned long long *>(smem + ((unsigned)ea[a].x >> 16)),
                    ((unsigned long long)wv::FloatToOrdered(tot) << 32) | (unsigned)ea[a].w);
      }
      // The next frame's log-likelihoods are requested here, into the registers this frame has just finished with, and are first
      // touched by the next frame's arc pass.  (Requested at the top of the frame into a second set of registers, the copy at the
      // loop's back edge made the compiler wait for vmcnt(0) there -- which on this ISA also counts the back-pointer stores the
      // commit pass has just issued: every frame waited for its own stores to be acknowledged.)
      // (unconditional -- the last frame asks for its own row again: a load under a branch left the compiler with a copy of
      // the registers, and a wait for the loads, right behind their issue)
      {
        const float *row = loglikes + (ll_base + (f + 1 < T ? f + 1 : f)) * ld;
#pragma unroll
        for (int a = 0; a < KE; a++) {
          // (the empty asm keeps the 32 -> 64 bit extension of the offset inside the loop, where instruction selection can fold it
          // into the load: scalar base + 32-bit vector offset.  Hoisted, the offsets are eight 64-bit register pairs and every
          // load is preceded by a 64-bit vector add)
          asm volatile("" : "+v"(ea[a].y));
          llv[a] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(row) + (unsigned)ea[a].y);
        }
      }
      float next_cutoff;
      {
        const unsigned wm = wv::MinU(wv::FloatToOrdered(local_min));
        if (lane == 0) xr[rb][wave] = make_int4((int)wm, 0, 0, 0);
        LdsBarrier();          // also: every emitting insertion has landed
        unsigned bm = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < NW; k++) bm = min(bm, (unsigned)xr[rb][k].x);
        rb ^= 1;
        next_cutoff = wv::OrderedToFloat(bm) + adaptive_beam;
        if (tid == 0) red.ncand = 0;           // (every wave is past its ranking; the next collection is a barrier away)
        hist_lo = wv::OrderedToFloat(bm);      // no emitted cost is below the smallest one
      }
      if (tid == 0) *reinterpret_cast<float4 *>(finfo + (size_t)f * 4) = make_float4(cost_offset, cur_cutoff, next_cutoff, adaptive_beam);
      closure_cutoff = next_cutoff;      // tokens at or above it are neither expanded below nor committed
      RS_T(2);
    }
    // ---- ProcessNonemitting to the fixpoint
    if (KX > 0 && rg.eps_depth != 0) {
      const int rounds = rg.eps_depth > 0 ? rg.eps_depth : 1 << 30;
      for (int round = 0; round < rounds; round++) {
        unsigned hsrc[KX];
#pragma unroll
        for (int a = 0; a < KX; a++) hsrc[a] = *reinterpret_cast<const unsigned *>(smem + (xa[a].x & 0xFFFF));
        int changed = 0;
#pragma unroll
        for (int a = 0; a < KX; a++) {
          const float c = wv::OrderedToFloat(hsrc[a]);               // empty key -> NaN -> fails both tests
          const float tot = c + __int_as_float(xa[a].z);
          const bool live = c < closure_cutoff;
          const bool pass = live & (tot < closure_cutoff);
          if (round == 0) n_arcs += (unsigned)live;
          if (pass) {
            const unsigned long long kk = ((unsigned long long)wv::FloatToOrdered(tot) << 32) | (unsigned)xa[a].w;
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(smem + ((unsigned)xa[a].x >> 16));
            if (rg.eps_depth > 0) {
              atomicMin(dst, kk);
            } else {
              const unsigned long long old = atomicMin(dst, kk);
              if (kk < old) { changed = 1; n_insert++; }
            }
          }
        }
        if (rg.eps_depth > 0) {
          LdsBarrier();
        } else {
          // cyclic or deep epsilon subgraph: vote
          const int any = __ballot(changed) != 0ull;
          if (lane == 0) xr[rb][wave] = make_int4(any, 0, 0, 0);
          LdsBarrier();
          int tot_any = 0;
#pragma unroll
          for (int k = 0; k < NW; k++) tot_any |= xr[rb][k].x;
          rb ^= 1;
          if (!tot_any) break;
        }
      }
    }
    RS_T(3);
    // ---- commit frame f+1: keys -> costs and back-pointers; statistics for the next frame
    int *bp_row = bp + (size_t)(f + 1) * S;
    float *cost_row = w.cost_rows ? w.cost_rows + (frame_row0 + (size_t)(f + 1)) * S : nullptr;      // (n-best / lattice calls)
    st_min = INF; st_arg = 0x7fffffff; st_cnt = 0;
    // ... and the histogram of the new frame's costs over [smallest emitted cost, cutoff) (any monotone binning will do: what falls
    // outside -- an epsilon arc of negative weight -- lands in the first bin)
    // With an infinite cutoff (fewer than min_active tokens were expanded: every other frame of a small grammar graph) the bins cover
    // hist_reach beams above the smallest cost and the last one takes the rest; hist_reach follows the rank (GetCutoff keeps it in the
    // middle half of the bins).
    if (f < 0) hist_lo = 0.f;
    hist_open = !(closure_cutoff < INF);
    hist_ok = hist_lo < INF && !o.no_commit_hist;
    {
      const float width = hist_open ? hist_reach * o.beam : closure_cutoff - hist_lo;
      hist_scale = (hist_ok && width > 0.f) ? 255.0f / width : 0.f;
    }
    hpar ^= 1;
    for (int i = tid; i < 256; i += NT) hist2[hpar ^ 1][i] = 0;        // the one this frame's GetCutoff has read
    auto commit_state = [&](int s, unsigned long long k) {
      const float c = wv::OrderedToFloat((unsigned)(k >> 32));
      const bool alive = c < closure_cutoff;                     // empty -> NaN -> false
      bp_row[s] = alive ? (int)(unsigned)(k & 0xFFFFFFFFull) : -2;
      cost_cur[s] = alive ? c : INF;
      if (cost_row) cost_row[s] = alive ? c : INF;
      key_next[s] = RS_EMPTY;
      if (alive & hist_ok) atomicAdd(&hist2[hpar][HistBin(c)], 1u);      // (unconditional, into a spare word for the dead: 1.10 -> 1.19 ms)
      st_cnt += (int)alive;
      const bool better = alive & (c < st_min);
      st_min = better ? c : st_min;
      st_arg = better ? s : st_arg;
    };
    {
      // a thread's first kCS keys are read together, then committed: written as one loop the LDS atomic of state s (which may alias
      // anything, as far as the compiler knows) stood between the reads, one LDS round trip per state behind the other
      constexpr int kCS = 3;
      unsigned long long kq[kCS];
#pragma unroll
      for (int q = 0; q < kCS; q++) { const int s = tid + q * NT; kq[q] = key_next[s < S ? s : S]; }      // (key_next has S + 1 entries)
#pragma unroll
      for (int q = 0; q < kCS; q++) { const int s = tid + q * NT; if (s < S) commit_state(s, kq[q]); }
      for (int s = tid + kCS * NT; s < S; s += NT) commit_state(s, key_next[s]);
    }
    n_alive += (unsigned)st_cnt;
    RS_T(4);
    // no barrier here: the first reduction of the next frame has one before anybody reads cost_cur / adds to key_next
  }
  __syncthreads();
  RS_T(5);
  if (!finishing) {
    // park the utterance: token costs and the scalars the next slab needs; counters are flushed (they add up)
    for (int s = tid; s < S; s += NT) state[s] = cost_cur[s];
    if (tid == 0) { state[S] = closure_cutoff; state[S + 1] = (float)error; }
    for (int i = tid; i < 8; i += NT) red.ctr[i] = 0;
    __syncthreads();
    atomicAdd(&red.ctr[0], (unsigned long long)n_expanded);
    atomicAdd(&red.ctr[1], (unsigned long long)n_arcs);
    atomicAdd(&red.ctr[2], (unsigned long long)n_insert);
    atomicAdd(&red.ctr[3], (unsigned long long)n_alive);
    __syncthreads();
    if (tid == 0) {
      long long *c8 = w.counters + slot * 8;
      for (int i = 0; i < 4; i++) c8[i] += (long long)red.ctr[i];
      c8[5] += max_active_frames;
      c8[6] += min_active_frames;
    }
    return;
  }
  FinishUtterance<NT>(red, h, g, loglikes, ld, w, cost_cur, bp, finfo, smem, smem_bytes, u, T, S, ll_base, error, n_expanded, n_arcs,
                      n_insert, n_alive, max_active_frames, min_active_frames, slot);
#ifdef RS_DECODE_PROFILE
  RS_T(6);
  if (u == 0 && tid == 0)
    printf("reg decode Kth fallback: %lld without a histogram, %lld with a crowded bin\n", prof2[6], prof2[7]);
  if (u == 0 && tid == 0)
    printf("reg decode Kth fast path: %lld uses, mean bin population %.1f; clocks per use: scan %lld collect %lld rank %lld\n", prof2[5],
           prof2[5] ? (double)prof2[4] / prof2[5] : 0.0, prof2[5] ? prof2[1] / prof2[5] : 0, prof2[5] ? prof2[2] / prof2[5] : 0, prof2[5] ? prof2[3] / prof2[5] : 0);
  if (u == 0 && tid == 0)
    printf("reg decode cycles/frame: stats %lld cutoff %lld emit %lld closure %lld commit %lld | finish total %lld (T=%d) | counting passes %lld, max-active frames %d, min-active frames %d\n",
           prof[0] / T, prof[1] / T, prof[2] / T, prof[3] / T, prof[4] / T, prof[6], T, prof[7], max_active_frames, min_active_frames);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// The same search with the reference's ORDER-DEPENDENT token creation (lattice-faster-decoder.cc:734-787, hash-list-inl.h:125-165).
//
// The reference walks a frame's tokens in the order its HashList holds them and lowers `next_cutoff` as it goes: arc i creates
// (or improves) a token iff tot_i < c_i, where c_i is the cutoff at the moment the arc is looked at -- the seed from the best
// token's arcs, lowered by every arc looked at before (tot_j + adaptive_beam).  RegDecodeKernel tests against the FINAL cutoff, so
// the reference owns a few tokens more ("extras": final cutoff <= cost < c_i).  They are never expanded by the closure, but the
// next frame's GetCutoff counts them, and when min-active / max-active binds the k-th smallest cost -- the cutoff -- can be one
// of theirs.  Here:
//   * c_i is an exclusive PREFIX MINIMUM over the arcs in the reference's order.  An arc that fails its test cannot lower the
//     cutoff (tot_j + adaptive_beam > c_j >= every later c), so the prefix runs over ALL arcs of the expanded tokens: per token the
//     minimum over its arcs (LDS atomic at the token's list position), one block scan over the positions, and inside a token the
//     handful of arcs before arc i (their values sit in LDS in arc order).
//   * The order.  With at most 1000 states the reference's table (1000 buckets at least, grown by doubling the token count) never
//     collides: bucket = state, and the list is the order in which the frame's states were first inserted -- by the emitting arcs in
//     (position of the source token, arc number) order, then by ProcessNonemitting, which pops its queue from the BACK: source
//     tokens in reverse list order, arcs in order (graphs whose epsilon arcs never chain: the queue only shrinks).  Every state
//     keeps the smallest such key among the arcs that passed (LDS atomic min); dense positions follow from a bit mask per source
//     position + one prefix sum of the masks' populations.
// Everything else -- GetCutoff on all tokens of the list, the closure against the final cutoff, ties, back-pointers -- is the
// reference's already.  Costs equal the CPU oracle's (oracle/decoder.c follows the hash order too) bit for bit.
// Eligible graphs: <= 1000 states, epsilon depth <= 1, at most 32 emitting and 32 epsilon arcs per state (RegGraphDev::exact_ok).
// Seven block-wide steps more per frame than RegDecodeKernel and 14 S + 4 E bytes of LDS: rs_decode_opts.exact_token_order.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned DppId(unsigned v, unsigned ident) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)ident, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ unsigned WaveScanInclMinU(unsigned v) {
  const unsigned I = 0xFFFFFFFFu;
  v = min(v, DppId<0x111, 0xF>(v, I));
  v = min(v, DppId<0x112, 0xF>(v, I));
  v = min(v, DppId<0x114, 0xF>(v, I));
  v = min(v, DppId<0x118, 0xF>(v, I));
  v = min(v, DppId<0x142, 0xA>(v, I));
  v = min(v, DppId<0x143, 0xC>(v, I));
  return v;
}
// exclusive scan (minimum of ordered keys, or sum) of arr[0 .. n) in place, n <= 4 NT; returns the total; one barrier inside, one after
// (popc_of: scan the populations of these masks instead of arr's contents; arr receives the result)
template <int NT, bool IS_MIN>
__device__ __forceinline__ unsigned BlockScanExcl(unsigned *arr, int n, int4 (*xr)[NT / 64], int &rb, const unsigned *popc_of = nullptr) {
  constexpr int NW = NT / 64, PER = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned I = IS_MIN ? 0xFFFFFFFFu : 0u;
  unsigned v[PER], run = I;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int idx = tid * PER + j;
    v[j] = idx < n ? (popc_of ? (unsigned)__popc(popc_of[idx]) : arr[idx]) : I;
    run = IS_MIN ? min(run, v[j]) : run + v[j];
  }
  const unsigned incl = IS_MIN ? WaveScanInclMinU(run) : (unsigned)WaveScanIncl((int)run);
  unsigned excl = (unsigned)__shfl_up((int)incl, 1, 64);
  if (lane == 0) excl = I;
  if (lane == 63) xr[rb][wave] = make_int4((int)incl, 0, 0, 0);
  LdsBarrier();
  unsigned before = I, total = I;
#pragma unroll
  for (int k = 0; k < NW; k++) {
    const unsigned t = (unsigned)xr[rb][k].x;
    if (k < wave) before = IS_MIN ? min(before, t) : before + t;
    total = IS_MIN ? min(total, t) : total + t;
  }
  rb ^= 1;
  unsigned acc = IS_MIN ? min(before, excl) : before + excl;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int idx = tid * PER + j;
    if (idx < n) arr[idx] = acc;
    acc = IS_MIN ? min(acc, v[j]) : acc + v[j];
  }
  LdsBarrier();
  return total;
}

template <int NT, int KE, int KX>
__global__ __launch_bounds__(NT) void RegDecodeExactKernel(HclgDev h, RegGraphDev rg, DecodeOptsDev o, BatchGeom g,
                                                           const float *__restrict__ loglikes, int ld, DenseWork w, int smem_bytes,
                                                           int f_begin, int f_end) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);
  constexpr int NW = NT / 64, E = NT * KE;
  __shared__ Red<NW> red;
  __shared__ int4 xr[2][NW];
  __shared__ unsigned seed_u;
  int rb = 0;
  const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = g.d_num_frames[u], S = h.num_states;
  const bool windows = w.win_begin != nullptr;
  if (windows) { f_begin = w.win_begin[u]; f_end = w.win_end[u]; }
  const bool finishing = windows ? w.win_final[u] != 0 : f_end >= T;
  if (windows ? (f_begin >= f_end && !finishing) : (f_begin >= 0 && f_begin >= T)) return;
  const int f_stop = f_end < T ? f_end : T;
  const size_t slot = windows ? (size_t)w.slot[u] : (size_t)u;
  const size_t frame_row0 = windows ? (size_t)w.pool_row[u] : (size_t)u * (g.max_frames + 1);
  float *state = w.state_cost + slot * (2 * (size_t)S + 4);
  float *cost_cur = reinterpret_cast<float *>(smem);
  unsigned long long *key_next = reinterpret_cast<unsigned long long *>(smem + rg.key_base);
  // the order's arrays, behind the keys
  const int xb = (rg.key_base + 8 * (S + 1) + 15) & ~15;
  unsigned short *rank16 = reinterpret_cast<unsigned short *>(smem + xb);                       // position of a state's token in the frame's list
  unsigned *fkey = reinterpret_cast<unsigned *>(smem + xb + ((2 * S + 15) & ~15));               // smallest insertion key of the frame under construction
  unsigned *ordm = fkey + S;                                                                     // per list position: min over the token's arcs of tot + adaptive_beam; later the prefix sums
  unsigned *maskE = ordm + S;                                                                    // per source position: which of its arcs inserted a state first
  float *arcv = reinterpret_cast<float *>(maskE + S);                                            // per emitting arc (table order): tot + adaptive_beam
  unsigned *maskX = reinterpret_cast<unsigned *>(arcv);                                          // (after the arc pass) the same for the closure's arcs
  int *bp = w.bp + frame_row0 * S;
  float *finfo = w.frame_info + frame_row0 * 4;
  const float INF = INFINITY;
  const size_t ll_base = (size_t)g.d_row_base[u] + g.L;
  unsigned long long n_expanded = 0, n_arcs = 0, n_insert = 0, n_alive = 0;
  int max_active_frames = 0, min_active_frames = 0;
  int4 ea[KE], xa[KX];
  int eaux[KE], xaux[KX];
#pragma unroll
  for (int a = 0; a < KE; a++) { ea[a] = rg.e_tab[(size_t)a * NT + tid]; eaux[a] = rg.e_aux[(size_t)a * NT + tid]; }
#pragma unroll
  for (int a = 0; a < KX; a++) { xa[a] = rg.x_tab[(size_t)a * NT + tid]; xaux[a] = rg.x_aux[(size_t)a * NT + tid]; }
  for (int s = tid; s <= S; s += NT) { cost_cur[s] = (f_begin >= 0 && s < S) ? state[s] : INF; key_next[s] = RS_EMPTY; }
  for (int s = tid; s < S; s += NT) {
    rank16[s] = f_begin >= 0 ? (unsigned short)state[S + 4 + s] : (unsigned short)0;
    fkey[s] = 0xFFFFFFFFu; ordm[s] = 0xFFFFFFFFu; maskE[s] = 0u;
  }
  for (int i = tid; i < (E > S ? E : S); i += NT) maskX[i] = 0u;
  for (int i = tid; i < 256; i += NT) red.hist[i] = 0;
  if (tid == 0) { red.ncand = 0; seed_u = 0xFFFFFFFFu; }
  const int f_first = f_begin < 0 ? 0 : f_begin;
  float llv[KE];
#pragma unroll
  for (int a = 0; a < KE; a++) llv[a] = 0.f;
  if (f_first < T) {
    const float *row = loglikes + (ll_base + f_first) * ld;
#pragma unroll
    for (int a = 0; a < KE; a++) llv[a] = row[ea[a].y];
  }
  __syncthreads();
  if (f_begin < 0 && tid == 0) { key_next[h.start] = PackKey(0.0f, RS_NOARC); fkey[h.start] = 0u; rank16[h.start] = 0; }
  float closure_cutoff = f_begin < 0 ? o.beam : state[S];
  int error = f_begin < 0 ? 0 : (int)state[S + 1];
  float st_min = INF, st_max = -INF;
  int st_arg = 0x7fffffff, st_cnt = 0;
  if (f_begin >= 0) {
    for (int s = tid; s < S; s += NT) {
      const float c = cost_cur[s];
      const bool alive = c < INF;
      st_cnt += (int)alive;
      st_max = alive ? fmaxf(st_max, c) : st_max;
      const bool better = alive & (c < st_min);
      st_min = better ? c : st_min;
      st_arg = better ? s : st_arg;
    }
  }
  __syncthreads();
  int n_emit = 1;                                 // tokens the emitting arcs put on the list (the start token for the first closure)
#ifdef RS_DECODE_PROFILE
  long long prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long t_last = clock64();
#endif

  for (int f = f_begin; f < f_stop && !error; f++) {
    if (f >= 0) {
      // ---- best token (ties: smallest state), token count, largest cost (the list holds tokens beyond last frame's cutoff)
      float best_cost, hist_hi;
      int best_state, N;
      {
        const unsigned ub = wv::FloatToOrdered(st_min);
        const unsigned wm = wv::MinU(ub);
        const unsigned long long tie = __ballot(ub == wm);
        unsigned wa;
        if (__popcll(tie) == 1) wa = (unsigned)__builtin_amdgcn_readlane(st_arg, __ffsll((long long)tie) - 1);
        else wa = wv::MinU(ub == wm ? (unsigned)st_arg : 0x7fffffffu);
        const int wn = wv::Sum(st_cnt);
        const unsigned wx = wv::MaxU(wv::FloatToOrdered(st_max));
        if (lane == 0) xr[rb][wave] = make_int4((int)wm, (int)wa, wn, (int)wx);
        LdsBarrier();
        unsigned long long bk = ~0ull;
        unsigned bx = 0u;
        N = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) {
          const int4 e = xr[rb][k];
          const unsigned long long kk = ((unsigned long long)(unsigned)e.x << 32) | (unsigned)e.y;
          bk = kk < bk ? kk : bk;
          N += e.z;
          bx = max(bx, (unsigned)e.w);
        }
        rb ^= 1;
        best_cost = wv::OrderedToFloat((unsigned)(bk >> 32));
        best_state = (int)(unsigned)(bk & 0xFFFFFFFFull);
        hist_hi = wv::OrderedToFloat(bx);
      }
      if (N == 0) { error = 1; break; }
      RS_T(0);
      // ---- GetCutoff over every token of the list (:644-711)
      const float beam_cutoff = best_cost + o.beam;
      int c_le = 0, c_lt = 0;
      const bool need_pass = N > o.max_active || N > o.min_active;      // workgroup-uniform
      if (need_pass) {
        const float hscale = hist_hi > best_cost ? 255.0f / (hist_hi - best_cost) : 0.f;
        for (int s = tid; s < S; s += NT) {
          const float c = cost_cur[s];
          c_le += (int)(c <= beam_cutoff) & (int)(c < INF);
          c_lt += (int)(c < beam_cutoff);
          if (c < INF) atomicAdd(&red.hist[KthBin(c, best_cost, hscale)], 1u);
        }
        const int wa = wv::Sum(c_le), wb = wv::Sum(c_lt);
        if (lane == 0) xr[rb][wave] = make_int4(wa, wb, 0, 0);
        LdsBarrier();
        c_le = 0; c_lt = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) { const int4 e = xr[rb][k]; c_le += e.x; c_lt += e.y; }
        rb ^= 1;
      }
      int kth = -1;
      if (N > o.max_active && c_lt > o.max_active) kth = o.max_active;
      else if (N > o.min_active && !(o.min_active == 0 || c_le > o.min_active)) kth = o.min_active;
      float kth_cost = 0.f;
      int kth_le = 0;
      if (need_pass) {
        if (kth >= 0) kth_cost = KthFromHist<NT>(red, cost_cur, S, kth, best_cost, hist_hi, &kth_le);
        else for (int i = tid; i < 256; i += NT) red.hist[i] = 0;
      }
      float cur_cutoff, adaptive_beam;
      int n_exp;
      if (N > o.max_active && c_lt > o.max_active) {
        adaptive_beam = kth_cost - best_cost + o.beam_delta;
        cur_cutoff = kth_cost;
        n_exp = kth_le;
        max_active_frames++;
      } else {
        float min_active_cutoff = INF;
        bool loosened;
        if (N > o.min_active) {
          min_active_cutoff = kth >= 0 ? kth_cost : best_cost;
          loosened = min_active_cutoff > beam_cutoff;
        } else {
          loosened = true;
        }
        if (loosened) {
          adaptive_beam = min_active_cutoff - best_cost + o.beam_delta;
          cur_cutoff = min_active_cutoff;
          n_exp = N > o.min_active ? kth_le : N;
          if (N > o.min_active) min_active_frames++;
        } else {
          adaptive_beam = o.beam;
          cur_cutoff = beam_cutoff;
          n_exp = need_pass ? c_le : N;
        }
      }
      const float cost_offset = -best_cost;
      if (tid == 0) n_expanded += (unsigned)n_exp;
      RS_T(1);
      // ---- arc pass A: every expanded token's arcs -- tot + adaptive_beam per arc and (minimum) per list position, the seed
      float tot_a[KE];
      bool exp_a[KE];
      const int best_addr = best_state * 4;
#pragma unroll
      for (int a = 0; a < KE; a++) {
        const int saddr = ea[a].x & 0xFFFF;
        const float c = *reinterpret_cast<const float *>(smem + saddr);
        const bool expd = (c < INF) & (c <= cur_cutoff);
        const float lk = llv[a];
        const float gc = __int_as_float(ea[a].z);
        const float tot = (c + (cost_offset - lk)) + gc;
        const float alt = ((gc + cost_offset) - lk) + c;                           // :752-757, the seed's own expression
        const float v = tot + adaptive_beam;
        tot_a[a] = tot;
        exp_a[a] = expd;
        arcv[a * NT + tid] = expd ? v : INF;
        if (expd) {
          atomicMin(&ordm[rank16[saddr >> 2]], wv::FloatToOrdered(v));
          if (saddr == best_addr) atomicMin(&seed_u, wv::FloatToOrdered(alt + adaptive_beam));
        }
      }
      {
        const float *row = loglikes + (ll_base + (f + 1 < T ? f + 1 : f)) * ld;
#pragma unroll
        for (int a = 0; a < KE; a++) llv[a] = row[ea[a].y];
      }
      LdsBarrier();
      RS_T(2);
      // ---- the cutoff in front of every list position; the frame's final one
      const unsigned tot_min_u = BlockScanExcl<NT, true>(ordm, N, xr, rb);
      const unsigned seed_now = seed_u;
      const float next_cutoff = wv::OrderedToFloat(min(seed_now, tot_min_u));
      RS_T(3);
      // ---- arc pass B: the reference's test, arc by arc
#pragma unroll
      for (int a = 0; a < KE; a++) {
        if (exp_a[a]) {
          const int saddr = ea[a].x & 0xFFFF, first = eaux[a] >> 8, local = eaux[a] & 255;
          const unsigned r_src = rank16[saddr >> 2];
          unsigned cu = min(seed_now, ordm[r_src]);
          for (int k = 0; k < local; k++) cu = min(cu, wv::FloatToOrdered(arcv[first + k]));
          const unsigned tu = wv::FloatToOrdered(tot_a[a]);
          n_arcs++;
          if (tu < cu) {
            const unsigned daddr = (unsigned)ea[a].x >> 16;
            atomicMin(reinterpret_cast<unsigned long long *>(smem + daddr), ((unsigned long long)tu << 32) | (unsigned)ea[a].w);
            atomicMin(&fkey[(daddr - (unsigned)rg.key_base) >> 3], (r_src << 8) | (unsigned)local);
          }
        }
      }
      if (tid == 0) *reinterpret_cast<float4 *>(finfo + (size_t)f * 4) = make_float4(cost_offset, cur_cutoff, next_cutoff, adaptive_beam);
      closure_cutoff = next_cutoff;
      LdsBarrier();          // every emitting insertion has landed
      RS_T(4);
      // ---- list positions of the states the emitting arcs inserted
      for (int s = tid; s < S; s += NT) { const unsigned k = fkey[s]; if (k != 0xFFFFFFFFu) atomicOr(&maskE[k >> 8], 1u << (k & 255u)); }
      for (int i = tid; i < S; i += NT) maskX[i] = 0u;      // (arcv is done; positions < n_emit <= S are used)
      LdsBarrier();
      n_emit = (int)BlockScanExcl<NT, false>(ordm, N, xr, rb, maskE);
      for (int s = tid; s < S; s += NT) {
        const unsigned k = fkey[s];
        if (k != 0xFFFFFFFFu) rank16[s] = (unsigned short)(ordm[k >> 8] + (unsigned)__popc(maskE[k >> 8] & ((1u << (k & 255u)) - 1u)));
      }
      LdsBarrier();
    }
    RS_T(5);
    // ---- ProcessNonemitting: source tokens in reverse list order, one round (no epsilon chains)
    int n_eps = 0;
    if (KX > 0 && rg.eps_depth != 0) {
#pragma unroll
      for (int a = 0; a < KX; a++) {
        const unsigned saddr = (unsigned)xa[a].x & 0xFFFFu;
        const float c = wv::OrderedToFloat(*reinterpret_cast<const unsigned *>(smem + saddr));      // empty key -> NaN -> fails both tests
        const float tot = c + __int_as_float(xa[a].z);
        const bool live = c < closure_cutoff;
        n_arcs += (unsigned)live;
        if (live & (tot < closure_cutoff)) {
          const unsigned daddr = (unsigned)xa[a].x >> 16;
          atomicMin(reinterpret_cast<unsigned long long *>(smem + daddr), ((unsigned long long)wv::FloatToOrdered(tot) << 32) | (unsigned)xa[a].w);
          const unsigned rs = rank16[(saddr - 4u - (unsigned)rg.key_base) >> 3];
          atomicMin(&fkey[(daddr - (unsigned)rg.key_base) >> 3], 0x80000000u | (((unsigned)n_emit - 1u - rs) << 8) | (unsigned)xaux[a]);
        }
      }
      LdsBarrier();
      for (int s = tid; s < S; s += NT) { const unsigned k = fkey[s]; if (k != 0xFFFFFFFFu && (k & 0x80000000u)) atomicOr(&maskX[(k & 0x7FFFFFFFu) >> 8], 1u << (k & 255u)); }
      LdsBarrier();
      n_eps = (int)BlockScanExcl<NT, false>(ordm, n_emit, xr, rb, maskX);
      for (int s = tid; s < S; s += NT) {
        const unsigned k = fkey[s];
        if (k != 0xFFFFFFFFu && (k & 0x80000000u)) {
          const unsigned r = (k & 0x7FFFFFFFu) >> 8;
          rank16[s] = (unsigned short)((unsigned)n_emit + ordm[r] + (unsigned)__popc(maskX[r] & ((1u << (k & 255u)) - 1u)));
        }
      }
      LdsBarrier();
    }
    (void)n_eps;
    RS_T(6);
    // ---- commit frame f+1: every token of the list (the ones at or beyond the cutoff too), statistics for the next frame
    int *bp_row = bp + (size_t)(f + 1) * S;
    st_min = INF; st_max = -INF; st_arg = 0x7fffffff; st_cnt = 0;
    for (int s = tid; s < S; s += NT) {
      const unsigned long long k = key_next[s];
      const bool exists = k != RS_EMPTY;
      const float c = wv::OrderedToFloat((unsigned)(k >> 32));
      bp_row[s] = exists ? (int)(unsigned)(k & 0xFFFFFFFFull) : -2;
      cost_cur[s] = exists ? c : INF;
      key_next[s] = RS_EMPTY;
      fkey[s] = 0xFFFFFFFFu; ordm[s] = 0xFFFFFFFFu; maskE[s] = 0u;
      st_cnt += (int)exists;
      st_max = exists ? fmaxf(st_max, c) : st_max;
      const bool better = exists & (c < st_min);
      st_min = better ? c : st_min;
      st_arg = better ? s : st_arg;
    }
    if (tid == 0) seed_u = 0xFFFFFFFFu;
    n_alive += (unsigned)st_cnt;
    RS_T(7);
  }
  __syncthreads();
  if (!finishing) {
    for (int s = tid; s < S; s += NT) { state[s] = cost_cur[s]; state[S + 4 + s] = (float)rank16[s]; }
    if (tid == 0) { state[S] = closure_cutoff; state[S + 1] = (float)error; }
    for (int i = tid; i < 8; i += NT) red.ctr[i] = 0;
    __syncthreads();
    atomicAdd(&red.ctr[0], n_expanded);
    atomicAdd(&red.ctr[1], n_arcs);
    atomicAdd(&red.ctr[2], n_insert);
    atomicAdd(&red.ctr[3], n_alive);
    __syncthreads();
    if (tid == 0) {
      long long *c8 = w.counters + slot * 8;
      for (int i = 0; i < 4; i++) c8[i] += (long long)red.ctr[i];
      c8[5] += max_active_frames;
      c8[6] += min_active_frames;
    }
    return;
  }
  FinishUtterance<NT>(red, h, g, loglikes, ld, w, cost_cur, bp, finfo, smem, smem_bytes, u, T, S, ll_base, error, n_expanded, n_arcs,
                      n_insert, n_alive, max_active_frames, min_active_frames, slot);
#ifdef RS_DECODE_PROFILE
  if (u == 0 && tid == 0)
    printf("reg decode (exact order) cycles/frame: stats %lld cutoff %lld arcs-A %lld scan %lld arcs-B %lld emit-ranks %lld closure+ranks %lld commit %lld (T=%d)\n",
           prof[0] / T, prof[1] / T, prof[2] / T, prof[3] / T, prof[4] / T, prof[5] / T, prof[6] / T, prof[7] / T, T);
#endif
}

template <int NT, int KE, int KX>
static void LaunchOne(const HclgDev &h, const RegGraphDev &r, const DecodeOptsDev &o, const BatchGeom &g, const float *loglikes, int ld,
                      const DenseWork &w, size_t smem, int f_begin, int f_end, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&RegDecodeKernel<NT, KE, KX>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&RegDecodeExactKernel<NT, KE, KX>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  if (o.exact_order && r.exact_ok && h.num_states <= 4 * NT) {
    // the order's arrays behind the keys (RegDecodeExactKernel): list positions, insertion keys, per-position minima / sums, masks, per-arc values
    const int S = h.num_states, E = NT * KE;
    const size_t xb = ((size_t)r.key_base + 8 * (size_t)(S + 1) + 15) & ~(size_t)15;
    const size_t need = xb + (((size_t)2 * S + 15) & ~(size_t)15) + (size_t)12 * S + (size_t)4 * (E > S ? E : S);
    const size_t sm = smem > need ? smem : need;
    hipLaunchKernelGGL((RegDecodeExactKernel<NT, KE, KX>), dim3(g.n_utts), dim3(NT), sm, s, h, r, o, g, loglikes, ld, w, (int)sm, f_begin, f_end);
    return;
  }
  hipLaunchKernelGGL((RegDecodeKernel<NT, KE, KX>), dim3(g.n_utts), dim3(NT), smem, s, h, r, o, g, loglikes, ld, w, (int)smem, f_begin, f_end);
}

// Dense rows -> token lists.  Three launches, the two that read the rows parallel over (utterance, frame): a wave per frame counts its
// tokens (into frame_tok_off itself: a stream's finishing call hands over every frame of the stream), one wave per utterance
// prefix-sums the counts in place and checks the capacity, a wave per frame writes its tokens in state order (frame 0 starts with the
// start state's token: the lattice's start is token 0).  More tokens than the utterance's slice of the token array holds: capacity
// flag, no lists.  (Round 4/5: one workgroup per utterance walked its frames in turn -- 150 dependent global reads per wave and pass,
// 1.0 ms per 256 x 298 frames; this takes the time of reading the rows twice.)
constexpr int kD2TFrames = 4;      // frames (waves) per workgroup
__global__ __launch_bounds__(64 * kD2TFrames) void DenseCountKernel(HclgDev h, BatchGeom g, DenseWork dw, DecodeWork w) {
  const int u = blockIdx.y, lane = threadIdx.x & 63, f = blockIdx.x * kD2TFrames + (threadIdx.x >> 6);
  const int T = g.d_num_frames[u], S = h.num_states;
  int *frame_off = w.frame_tok_off + (size_t)u * (g.max_frames + 2);
  if (f > g.max_frames + 1) return;
  if (T <= 0 || w.out_nwords[u] < 0 || f > T) { if (lane == 0) frame_off[f] = 0; return; }
  const float *cost = dw.cost_rows + ((size_t)u * (g.max_frames + 1) + f) * S;
  int n = 0;
  for (int s0 = 0; s0 < S; s0 += 64) { const int s = s0 + lane; n += __popcll(__ballot(s < S && cost[s] < INFINITY)); }
  if (lane == 0) frame_off[f] = n;
}
__global__ __launch_bounds__(64) void DenseScanKernel(BatchGeom g, DenseWork dw, DecodeWork w) {
  const int u = blockIdx.x, lane = threadIdx.x;
  const int T = g.d_num_frames[u];
  int *frame_off = w.frame_tok_off + (size_t)u * (g.max_frames + 2);
  if (T <= 0 || w.out_nwords[u] < 0) return;      // (all zero already)
  int carry = 0;
  for (int f0 = 0; f0 <= T; f0 += 64) {            // exclusive prefix over the frames, 64 at a time
    const int f = f0 + lane;
    const int n = f <= T ? frame_off[f] : 0;
    int inc = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
    if (f <= T) frame_off[f] = carry + inc - n;
    carry += __shfl(inc, 63, 64);
  }
  if (carry > w.tok_cap) {      // (PlanSearch sizes the slice for every state of every frame, so this is a caller error)
    for (int f = lane; f <= g.max_frames + 1; f += 64) frame_off[f] = 0;
    if (lane == 0) { w.out_nwords[u] = -1; dw.counters[(size_t)u * 8 + 7] |= 1; }      // "decoder token capacity exceeded"
    return;
  }
  if (lane == 0) frame_off[T + 1] = carry;
}
__global__ __launch_bounds__(64 * kD2TFrames) void DenseWriteKernel(HclgDev h, BatchGeom g, DenseWork dw, DecodeWork w) {
  const int u = blockIdx.y, lane = threadIdx.x & 63, f = blockIdx.x * kD2TFrames + (threadIdx.x >> 6);
  const int T = g.d_num_frames[u], S = h.num_states;
  if (T <= 0 || w.out_nwords[u] < 0 || f > T) return;
  const int *frame_off = w.frame_tok_off + (size_t)u * (g.max_frames + 2);
  const size_t row = ((size_t)u * (g.max_frames + 1) + f) * S;
  const float *cost = dw.cost_rows + row;
  const int *bp = dw.bp + row;
  int4 *tokens = w.tokens + (size_t)u * w.tok_cap;
  int run = frame_off[f];
  if (f == 0) {      // the start token first
    if (lane == 0) tokens[run] = make_int4(h.start, __float_as_int(cost[h.start]), -1, bp[h.start]);
    run++;
  }
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const float c = s < S ? cost[s] : INFINITY;
    const bool on = c < INFINITY && !(f == 0 && s == h.start);
    const unsigned long long m = __ballot(on);
    if (on) tokens[run + __popcll(m & ((1ull << lane) - 1ull))] = make_int4(s, __float_as_int(c), -1, bp[s]);
    run += __popcll(m);
  }
}

void LaunchDenseToTokens(const HclgDev &h, const BatchGeom &g, const DenseWork &dw, const DecodeWork &w, hipStream_t s, bool write_tokens) {
  if (g.n_utts == 0) return;
  const dim3 grid((g.max_frames + 2 + kD2TFrames - 1) / kD2TFrames, g.n_utts);
  hipLaunchKernelGGL(DenseCountKernel, grid, dim3(64 * kD2TFrames), 0, s, h, g, dw, w);
  hipLaunchKernelGGL(DenseScanKernel, dim3(g.n_utts), dim3(64), 0, s, g, dw, w);
  // (DenseLatticeKernel works on the rows themselves: it needs the frames' token offsets only)
  if (write_tokens) hipLaunchKernelGGL(DenseWriteKernel, grid, dim3(64 * kD2TFrames), 0, s, h, g, dw, w);
}

// ===================================================================================== lattice extraction from the dense rows
// LatticeKernel's backward pass (decode_kernels.hip: FinalizeDecoding = PruneForwardLinksFinal on the last frame, PruneForwardLinks
// with delta 0 on every earlier one, lattice-faster-decoder.cc:299-458,625-640) for the graphs the register-resident search holds:
// the same links, float expressions and existence rules, on the cost rows that search leaves behind instead of token lists.
// Arcs live in registers for the whole utterance (arc i -> thread i % 256, slot i / 256) and are looked at arc-parallel; per-state
// arrays (cost of this and the next frame, extra_cost of both as ordered bits -- non-negative floats order like their bit
// patterns, so the min over a token's links is one LDS atomic --, token numbers of both) are in LDS; the next frame's cost row and
// log-likelihoods are requested a frame ahead.  Token numbers are DenseWriteKernel's: the live states of a frame in state order
// behind frame_tok_off (frame 0: the start state first).  A frame is ~6 barriers of LDS work; the token-list kernel's frame was
// ~11 barriers with two to four dependent global round trips in each phase (2.3 ms per 256 x 298 frames).
constexpr int kDLMaxStates = 2048, kDLMaxArcs = 8192;
#ifdef RS_DL_PROFILE
#define RS_DLP(i) do { const long long _n = clock64(); if (tid == 0) dlp[i] += _n - dl_last; dl_last = _n; } while (0)
#else
#define RS_DLP(i) do { } while (0)
#endif
constexpr unsigned kInfBits = 0x7f800000u;
template <int NT>
struct DenseLatticeCtx { float red_f[NT / 64]; int red_i[NT / 64]; float bcast_f[2]; int bcast_i[4]; };
template <int NT, int KA>
__global__ __launch_bounds__(NT) void DenseLatticeKernel(HclgDev h, DecodeOptsDev o, BatchGeom g, const float *__restrict__ loglikes, int ld,
                                                          DenseWork dw, DecodeWork w, LatticeWork lw, int eps_rounds) {
  constexpr int kDLStatesPerThread = kDLMaxStates / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char dl_smem[];
  __shared__ DenseLatticeCtx<NT> c;
  __shared__ int s_narcs;
  __shared__ int s_flag[3];
  __shared__ int s_wtot[NT / 64];
  const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = g.d_num_frames[u];
  if (T <= 0 || w.out_nwords[u] < 0) return;
  if (tid == 0) s_narcs = 0;              // (barriers follow before its first use)
  LatArc *my_arcs = lw.arcs + (size_t)u * lw.utt_cap;
  const int S = h.num_states, A = h.num_arcs;
  float *cost_a = reinterpret_cast<float *>(dl_smem), *cost_b = cost_a + S, *cost_c = cost_b + S;
  unsigned *ex_a = reinterpret_cast<unsigned *>(cost_c + S), *ex_b = ex_a + S;
  unsigned short *rk_a = reinterpret_cast<unsigned short *>(ex_b + S), *rk_b = rk_a + S;
  const int *frame_off = w.frame_tok_off + (size_t)u * (g.max_frames + 2);
  const float *finfo = dw.frame_info + (size_t)u * (g.max_frames + 1) * 4;
  const float *rows = dw.cost_rows + (size_t)u * (g.max_frames + 1) * S;
  const float INF = INFINITY, beam = o.lattice_beam;
  const size_t ll_base = (size_t)g.d_row_base[u] + g.L;
  // ---- this thread's arcs
  int ax[KA];            // pdf + 1, 0 = epsilon arc, -1 = no arc
  float aw[KA];
  unsigned asd[KA];      // source | destination << 16
#pragma unroll
  for (int k = 0; k < KA; k++) {
    const int a = tid + k * NT;
    if (a < A) { const int4 arc = h.arcs[a]; ax[k] = arc.x; aw[k] = __int_as_float(arc.z); asd[k] = (unsigned)h.arc_src[a] | ((unsigned)arc.w << 16); }
    else { ax[k] = -1; aw[k] = 0.f; asd[k] = 0u; }
  }
  // ---- the last frame's costs, final costs (ComputeFinalCosts)
  float *cost_cur = cost_a, *cost_nxt = cost_b, *cost_pre = cost_c;
  unsigned *ex_cur = ex_a, *ex_nxt = ex_b;
  unsigned short *rk_cur = rk_a, *rk_nxt = rk_b;
  float final_best;
  bool have_final;
  {
    float lv1 = INF, lv2 = INF;
    for (int s0 = 0; s0 < S; s0 += NT) {
      const int st = s0 + tid;
      if (st < S) {
        const float cst = rows[(size_t)T * S + st];
        cost_cur[st] = cst;
        cost_nxt[st] = INF;
        ex_nxt[st] = kInfBits;
        lv1 = fminf(lv1, cst + h.final_cost[st]);
        lv2 = fminf(lv2, cst);
      }
    }
    float b1, b2;
    int d1, d2;
    tok::BlockMinArg<NT>(c, lv1, tid, &b1, &d1);
    tok::BlockMinArg<NT>(c, lv2, tid, &b2, &d2);
    have_final = b1 < INF;
    final_best = have_final ? b1 : b2;
  }
  float llv[KA], pre_cost[kDLStatesPerThread];
#pragma unroll
  for (int k = 0; k < KA; k++) llv[k] = 0.f;
#ifdef RS_DL_PROFILE
  long long dlp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dl_last = clock64();
#endif
  int flag_round = 0;                    // (vote mode of the closure pass: three flags in rotation, one barrier per round)
  if (tid < 3) s_flag[tid] = 0;
  // The frame loop communicates through LDS only: its barriers order LDS traffic (dd::LdsBarrier) and leave the vector-memory counter
  // alone, so the rows requested at the top of a frame and the arc records stored at its end are in flight across them.
  // a frame's scalars (its token offset, cost offset and cutoffs) are requested two frames ahead as ordinary loads and ride along in
  // registers: as scalar loads at the top of the frame they were a cache miss in front of the frame's first barrier
  const float4 *finfo4 = reinterpret_cast<const float4 *>(finfo);
  float4 fi_cur = make_float4(0.f, 0.f, 0.f, 0.f), fi_below = T > 0 ? finfo4[T - 1] : make_float4(0.f, 0.f, 0.f, 0.f);
  int off = frame_off[T], off_n = frame_off[T + 1], off_below = T > 0 ? frame_off[T - 1] : 0;
  for (int f = T; f >= 0; f--) {
    const float cost_offset = fi_cur.x, cur_cutoff = fi_cur.y, next_cutoff = fi_cur.z;      // (f == T: unused)
    const float closure_cutoff = f > 0 ? fi_below.z : o.beam;
    const float4 fi_pre = f > 1 ? finfo4[f - 2] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int off_pre = f > 1 ? frame_off[f - 2] : 0;
    // requests for the frame below: its cost row, the log-likelihoods of this thread's emitting arcs (taken in before pass 3)
    float ll_next[KA];
    if (f > 0) {
      const float *ll_row = loglikes + (ll_base + (size_t)(f - 1)) * ld;
#pragma unroll
      for (int k = 0; k < KA; k++) ll_next[k] = ax[k] > 0 ? ll_row[ax[k] - 1] : 0.f;
#pragma unroll
      for (int q = 0; q < kDLStatesPerThread; q++) { const int st = q * NT + tid; pre_cost[q] = st < S ? rows[(size_t)(f - 1) * S + st] : INF; }
    } else {
#pragma unroll
      for (int k = 0; k < KA; k++) ll_next[k] = 0.f;
#pragma unroll
      for (int q = 0; q < kDLStatesPerThread; q++) pre_cost[q] = INF;
    }
    RS_DLP(0);
    dd::LdsBarrier();                    // cost_cur holds row f
    RS_DLP(1);
    // ---- token numbers of this frame: a thread takes kDLStatesPerThread consecutive states, the block prefix of the live counts is a
    // wave scan + four partial sums through LDS (a loop over 64-state chunks cost 3 300 cycles of a 12 000-cycle frame); extra_cost's
    // start values
    {
      const int st0 = tid * kDLStatesPerThread;
      float cst[kDLStatesPerThread];
      unsigned live = 0u;
#pragma unroll
      for (int q = 0; q < kDLStatesPerThread; q++) {
        const int st = st0 + q;
        cst[q] = st < S ? cost_cur[st] : INF;
        live |= ((cst[q] < INF && !(f == 0 && st == h.start)) ? 1u : 0u) << q;      // (frame 0: the start state's token is number 0, the others follow)
      }
      const int n_live = __popc(live);
      const int inc = dd::WaveScanIncl(n_live);      // (DPP row shifts + broadcasts; as six __shfl_up steps: six dependent ds_bpermute round trips)
      if (lane == 63) s_wtot[wave] = inc;
      dd::LdsBarrier();
      int run = (f == 0 ? 1 : 0) + inc - n_live;
#pragma unroll
      for (int w2 = 0; w2 < NT / 64 - 1; w2++) run += w2 < wave ? s_wtot[w2] : 0;
#pragma unroll
      for (int q = 0; q < kDLStatesPerThread; q++) {
        const int st = st0 + q;
        if (st < S) {
          rk_cur[st] = (unsigned short)((f == 0 && st == h.start) ? 0 : run);
          float e = INF;
          if (f == T && cst[q] < INF) e = cst[q] + (have_final ? h.final_cost[st] : 0.f) - final_best;
          ex_cur[st] = e < INF ? (__float_as_uint(e) & 0x7fffffffu) : kInfBits;
        }
        run += (int)(live >> q & 1u);
      }
    }
    RS_DLP(2);
    dd::LdsBarrier();
    // ---- the per-arc operands of the frame, requested together (with `if ... continue` chains a thread went through four
    // dependent LDS round trips per arc, twelve arcs, three passes: 5 us of a frame's 6)
    float cs[KA], cdn[KA], tot[KA];
    unsigned edn[KA];
#pragma unroll
    for (int k = 0; k < KA; k++) {
      const unsigned src = asd[k] & 0xFFFFu, dst = asd[k] >> 16;
      cs[k] = cost_cur[src];
      cdn[k] = (ax[k] > 0 ? cost_nxt : cost_cur)[dst];      // the destination token's cost: next frame (emitting arc) or this one (epsilon arc)
      edn[k] = ex_nxt[dst];
    }
    // ---- pass 1: emitting links into the next frame
    unsigned link_e = 0u, link_x = 0u;      // bit k: arc k meets the conditions that do not depend on this frame's extra costs
    float acv[KA];
#pragma unroll
    for (int k = 0; k < KA; k++) {
      acv[k] = cost_offset - llv[k];
      const bool em = ax[k] > 0;
      tot[k] = em ? (cs[k] + acv[k]) + aw[k] : cs[k] + aw[k];
      const float le = __uint_as_float(edn[k]) + (tot[k] - cdn[k]);
      const bool ok_e = em && f < T && cs[k] <= cur_cutoff && tot[k] < next_cutoff && cdn[k] < INF && !(le > beam);
      const bool ok_x = ax[k] == 0 && cs[k] < closure_cutoff && tot[k] < closure_cutoff && cdn[k] < INF;
      link_e |= (ok_e ? 1u : 0u) << k;
      link_x |= (ok_x ? 1u : 0u) << k;
      if (ok_e) atomicMin(&ex_cur[asd[k] & 0xFFFFu], __float_as_uint(le < 0.f ? 0.f : le) & 0x7fffffffu);
    }
    if (f < T) dd::LdsBarrier();
    RS_DLP(3);
    // ---- pass 2: epsilon links inside the frame, to the fixpoint: as many rounds as the longest epsilon path has arcs (known on
    // the host for an acyclic epsilon subgraph), else until a round changes nothing
    if (eps_rounds != 0) {
      for (int round = 0; round < (eps_rounds > 0 ? eps_rounds : 1000); round++) {
        int changed = 0;
        unsigned ed[KA];
#pragma unroll
        for (int k = 0; k < KA; k++) ed[k] = __hip_atomic_load(&ex_cur[asd[k] >> 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int k = 0; k < KA; k++) {
          const float le = __uint_as_float(ed[k]) + (tot[k] - cdn[k]);
          if ((link_x >> k & 1u) && !(le > beam)) {
            const unsigned nb = __float_as_uint(le < 0.f ? 0.f : le) & 0x7fffffffu;
            if (eps_rounds > 0) atomicMin(&ex_cur[asd[k] & 0xFFFFu], nb);
            else if (nb < atomicMin(&ex_cur[asd[k] & 0xFFFFu], nb)) changed = 1;
          }
        }
        if (eps_rounds > 0) { dd::LdsBarrier(); continue; }
        // flag r % 3 collects this round's votes; thread 0 clears the flag of the round after next in front of this round's barrier:
        // that flag's last readers passed the barrier before, its next writers come after this one
        const int fr = flag_round % 3;
        if (tid == 0) s_flag[(flag_round + 1) % 3] = 0;
        if (changed) s_flag[fr] = 1;
        flag_round++;
        dd::LdsBarrier();
        if (!s_flag[fr]) break;
      }
    }
    if (f == T) {
      for (int st = tid; st < S; st += NT) if (__uint_as_float(ex_cur[st]) > beam) ex_cur[st] = kInfBits;
      dd::LdsBarrier();
    }
    RS_DLP(4);
    // ---- the rows requested at the top are taken in here, before this frame's arc records are stored behind them
    if (f > 0) {
#pragma unroll
      for (int q = 0; q < kDLStatesPerThread; q++) { const int st = q * NT + tid; if (st < S) cost_pre[st] = pre_cost[q]; }
    }
#pragma unroll
    for (int k = 0; k < KA; k++) llv[k] = ll_next[k];
    RS_DLP(5);
    // ---- pass 3: the surviving links (and the final-cost records of the last frame); the utterance's own region of the arc buffer,
    // an LDS counter, one update per wave
    {
      unsigned emit = 0u;                 // bit k: arc k is a link of the lattice
      {
        unsigned es[KA], ed[KA];
#pragma unroll
        for (int k = 0; k < KA; k++) { es[k] = ex_cur[asd[k] & 0xFFFFu]; ed[k] = ex_cur[asd[k] >> 16]; }
#pragma unroll
        for (int k = 0; k < KA; k++) {
          const bool src_ok = __uint_as_float(es[k]) < INF;
          const bool x_ok = (link_x >> k & 1u) && !(__uint_as_float(ed[k]) + (tot[k] - cdn[k]) > beam);
          emit |= ((src_ok && ((link_e >> k & 1u) || x_ok)) ? 1u : 0u) << k;
        }
      }
      const int n_mine = __popc(emit);
      const int inc = dd::WaveScanIncl(n_mine);
      const int wave_total = __builtin_amdgcn_readlane(inc, 63);
      if (wave_total > 0) {
        // the token numbers of both ends of every arc first (LDS), then the records (global stores)
        int rs_[KA], rd_[KA];
#pragma unroll
        for (int k = 0; k < KA; k++) {
          const unsigned src = asd[k] & 0xFFFFu, dst = asd[k] >> 16;
          rs_[k] = off + (int)rk_cur[src];
          rd_[k] = ax[k] > 0 ? off_n + (int)rk_nxt[dst] : off + (int)rk_cur[dst];
        }
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_narcs, wave_total);
        base = __builtin_amdgcn_readfirstlane(base) + inc - n_mine;
#pragma unroll
        for (int k = 0; k < KA; k++) {
          if (!(emit >> k & 1u)) continue;
          if (base < lw.utt_cap) my_arcs[base] = LatArc{u, rs_[k], rd_[k], tid + k * NT, aw[k], ax[k] > 0 ? acv[k] - cost_offset : 0.f};
          base++;
        }
      }
      if (f == T) {
        for (int st = tid; st < S; st += NT) {
          if (!(__uint_as_float(ex_cur[st]) < INF)) continue;
          const float fc = have_final ? h.final_cost[st] : 0.f;
          if (fc < INF) {
            const int k2 = atomicAdd(&s_narcs, 1);
            if (k2 < lw.utt_cap) my_arcs[k2] = LatArc{u, off + (int)rk_cur[st], -1, -1, fc, 0.f};
          }
        }
      }
    }
    // ---- frame f becomes "next", the row taken in above "current", the old "next" is free (its readers are behind the next barrier:
    // nobody writes it before the top-of-frame barrier of the frame after)
    RS_DLP(6);
    { float *t1 = cost_nxt; cost_nxt = cost_cur; cost_cur = cost_pre; cost_pre = t1; }
    fi_cur = fi_below; fi_below = fi_pre;
    off_n = off; off = off_below; off_below = off_pre;
    { unsigned *t2 = ex_cur; ex_cur = ex_nxt; ex_nxt = t2; }
    { unsigned short *t3 = rk_cur; rk_cur = rk_nxt; rk_nxt = t3; }
  }
  __syncthreads();
  if (tid == 0) lw.arcs_count[u] = s_narcs;
#ifdef RS_DL_PROFILE
  if (tid == 0 && (u == 0 || u == 100)) printf("dense lattice block %d T=%d arcs=%d: top %lld barrier0 %lld rank %lld operands+pass1 %lld pass2 %lld take-in %lld pass3 %lld\n", u, T, s_narcs, dlp[0], dlp[1], dlp[2], dlp[3], dlp[4], dlp[5], dlp[6]);
#endif
}

bool DenseLatticeUsable(const HclgDev &h) {
  const char *e = std::getenv("RS_LATTICE_KERNEL");      // "tokens": the token-list kernel (read per call: a test compares the two)
  if (e && std::string(e) == "tokens") return false;
  return h.num_states <= kDLMaxStates && h.num_arcs <= kDLMaxArcs;
}

void LaunchDenseLattice(const HclgDev &h, const DecodeOptsDev &o, const BatchGeom &g, const float *loglikes, int ld, const DenseWork &dw,
                        const DecodeWork &w, const LatticeWork &lw, int eps_rounds, hipStream_t s) {
  if (g.n_utts == 0) return;
  { const char *e = std::getenv("RS_LATTICE_KERNEL"); if (e && std::string(e) == "vote" && eps_rounds != 0) eps_rounds = -1; }      // (tests: closure rounds until nothing changes)
  const size_t smem = (size_t)h.num_states * (3 * 4 + 2 * 4 + 2 * 2) + 16;
  // 512 threads: a wave alone on its SIMD issues an instruction every ~10 cycles whatever it is, and a frame is per-arc instructions
  // (256 / 512 / 1024 threads: 1.9 / 1.3 / 1.3 ms per 256 x 298 frames, profiles/micro/dl_nt.sh; RS_DL_NT in a -DRS_TUNING build)
  static const int nt = [] { const char *e = TuneEnv("RS_DL_NT"); const int v = e ? std::atoi(e) : 512; return v == 256 || v == 1024 ? v : 512; }();
  const int ka = (h.num_arcs + nt - 1) / nt;
  const dim3 grid(g.n_utts);
#define RS_DL(NT, KA) hipLaunchKernelGGL((DenseLatticeKernel<NT, KA>), grid, dim3(NT), smem, s, h, o, g, loglikes, ld, dw, w, lw, eps_rounds)
  if (nt == 1024) { if (ka <= 1) RS_DL(1024, 1); else if (ka <= 2) RS_DL(1024, 2); else if (ka <= 3) RS_DL(1024, 3); else if (ka <= 4) RS_DL(1024, 4); else if (ka <= 6) RS_DL(1024, 6); else RS_DL(1024, 8); }
  else if (nt == 512) { if (ka <= 2) RS_DL(512, 2); else if (ka <= 4) RS_DL(512, 4); else if (ka <= 6) RS_DL(512, 6); else if (ka <= 8) RS_DL(512, 8); else if (ka <= 12) RS_DL(512, 12); else RS_DL(512, 16); }
  else { if (ka <= 4) RS_DL(256, 4); else if (ka <= 8) RS_DL(256, 8); else if (ka <= 12) RS_DL(256, 12); else if (ka <= 16) RS_DL(256, 16); else if (ka <= 24) RS_DL(256, 24); else RS_DL(256, 32); }
#undef RS_DL
}

// the instantiations; RegDecodeConfig picks the first one the graph fits.  Workgroup size measured on MI355X (625-state
// grammar graph, 298 frames): 64 threads 7.5 us/frame, 256 -> 4.2, 512 -> 3.7, 1024 -> 5.1: the per-lane instruction count
// dominates until the barriers of 16 waves take over.
static const int kRegConfigs[][3] = {{512, 4, 2}, {256, 8, 4}, {512, 8, 4}, {256, 16, 8}, {256, 32, 16}};

bool RegDecodeConfig(int num_states, int num_emitting, int num_eps, int *nt, int *ke, int *kx) {
  if (num_states > kRegMaxStates) return false;
  static int force_nt = [] { const char *e = TuneEnv("RS_REG_NT"); return e ? std::atoi(e) : 0; }();
  for (const auto &c : kRegConfigs) {
    if (force_nt && c[0] != force_nt) continue;
    if ((long long)c[0] * c[1] >= num_emitting && (long long)c[0] * c[2] >= num_eps) { *nt = c[0]; *ke = c[1]; *kx = c[2]; return true; }
  }
  return false;
}

bool LaunchDecodeReg(const HclgDev &h, const RegGraphDev &r, const DecodeOptsDev &o_in, const BatchGeom &g,
                     const float *loglikes, int ld, const DenseWork &w, int f_begin, int f_end, hipStream_t s, bool any_final) {
  if (g.n_utts == 0) return true;
  DecodeOptsDev o = o_in;
  { const char *e = std::getenv("RS_REG_NO_HIST"); o.no_commit_hist = e && std::atoi(e) != 0 ? 1 : 0; }      // (read per launch: a test flips it)
  size_t smem = (size_t)r.key_base + (size_t)(h.num_states + 1) * 8;
  // room to stage back-pointer rows for the traceback.  48 KB, not more: with 128 KB a search workgroup left no room for
  // the GEMM workgroups (33 KB each) of the next decode call on its CU, and the overlap of calls in flight was limited to
  // the feature / iVector stages (3.7 ms per headline batch against 3.35 with 48 KB; the search itself takes the same
  // time).  A slab that finishes no utterance does not trace back and keeps its LDS footprint minimal.
  // (12 KB since the calls' stages are chained, engine.cc: 2.51 -> 2.47-2.49 ms per headline step; 4-16 KB are within 1 % of each other.
  // Round 4: 32 KB -- the 16-bit arc -> source table now sits in front of the rows, and the layer GEMM's 72 KB leave one of its
  // workgroups room beside a search whatever this is; 12 / 20 / 32 / 44 KB: search 1.26 / 1.23 / 1.20 / 1.20 ms, profiles/micro/stage_kb.sh)
  static const size_t stage_kb = [] { const char *e = TuneEnv("RS_DECODE_STAGE_KB"); return e ? (size_t)std::atoi(e) : 32; }();
  const size_t stage = (w.win_begin ? !any_final : f_end <= g.max_frames) ? 0 : stage_kb * 1024;
  if (smem < stage) smem = stage;
  // A batch that puts a search workgroup on (nearly) every CU shares those CUs with the GEMM workgroups of the next call: with
  // half the waves and twice the arcs per thread the search alone is 8 % slower (1.12 -> 1.21 ms for 256 x 3 s) and the step with
  // calls in flight 2.5 % faster (2.51 -> 2.45 ms together with the smaller traceback staging below).  The tables are the same --
  // arc i sits in slot i of e_tab / x_tab whatever the shape.  RS_REG_NT pins the shape chosen at load.
  static const bool pinned = TuneEnv("RS_REG_NT") != nullptr;
  static const int num_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const bool crowded = !pinned && !w.win_begin && 4 * (long)g.n_utts >= 3 * (long)num_cu;
  if (crowded && r.nt == 512 && r.ke == 4 && r.kx == 2) LaunchOne<256, 8, 4>(h, r, o, g, loglikes, ld, w, smem, f_begin, f_end, s);
  else if (crowded && r.nt == 512 && r.ke == 8 && r.kx == 4) LaunchOne<256, 16, 8>(h, r, o, g, loglikes, ld, w, smem, f_begin, f_end, s);
  else if (r.nt == 512 && r.ke == 4 && r.kx == 2) LaunchOne<512, 4, 2>(h, r, o, g, loglikes, ld, w, smem, f_begin, f_end, s);
  else if (r.nt == 512 && r.ke == 8 && r.kx == 4) LaunchOne<512, 8, 4>(h, r, o, g, loglikes, ld, w, smem, f_begin, f_end, s);
  else if (r.nt == 256 && r.ke == 16 && r.kx == 8) LaunchOne<256, 16, 8>(h, r, o, g, loglikes, ld, w, smem, f_begin, f_end, s);
  else if (r.nt == 256 && r.ke == 32 && r.kx == 16) LaunchOne<256, 32, 16>(h, r, o, g, loglikes, ld, w, smem, f_begin, f_end, s);
  else if (r.nt == 256 && r.ke == 8 && r.kx == 4) LaunchOne<256, 8, 4>(h, r, o, g, loglikes, ld, w, smem, f_begin, f_end, s);
  else return false;
  return true;
}

}  // namespace rs
