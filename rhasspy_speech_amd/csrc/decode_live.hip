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
//  * first measurement of the above (profiles/r05/live_notes.txt): the round trips were not the bound.  A workgroup ALONE on
//    its CU ran the heaviest utterance as fast with 512 threads as with 1024, and two workgroups on a CU ran half as fast each:
//    a CU's path to L2 was saturated by scattered 16-byte accesses, each pulling a whole cache line -- ~77 k lines per 9 k-token
//    frame (state record, arcs, key atomicMin, key read by the winners, slot's token, key read + clear by the completion pass).
//    So the recombination keys moved into LDS: the table is tags[HS] + keys[HS] (12 bytes per live state), claimed by one LDS
//    compare-and-swap, raced on by ds_min_u64, read by the winners' and the completion pass from LDS and emptied wholesale at
//    the end of the frame.  Behind it a second level in global memory (32 K entries: tag + key) takes the states whose probe
//    window in LDS is taken, so nothing is restarted when a frame outgrows LDS; only > 24 576 live states hands the utterance
//    to DecodeKernel as before.  The slot IS the entry's position (one form for graphs of any size);
//  * a state's record and its first two emitting arcs are ONE 64-byte record (HclgDev::nodes): one line per expanded token instead
//    of two; candidate records are 8 bytes (arc | flags, slot | source token: the destination state is the slot's tag);
//  * slot -> token index is only kept for the states the closure can touch (destinations of epsilon arcs, states with epsilon
//    arcs, tokens the closure creates): a flag on the arc says so.
// Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstdio>

#include "kernels.h"
#include "decode_tok.h"
#include "wave_ops.h"

namespace rs {
using namespace tok;
namespace {

constexpr int kGlobalLog = 15;                      // second-level table: 32768 entries (tag + key) per utterance in global memory
constexpr int kGlobalSize = 1 << kGlobalLog;
constexpr int kSlotCap = 24576;                     // live states / tokens per frame (records name a token of a frame in 16 bits)
constexpr unsigned kFree = 0xFFFFFFFFu;
#ifndef RS_LIVE_BUCKETS
#define RS_LIVE_BUCKETS 3
#endif
constexpr int kLdsBuckets = RS_LIVE_BUCKETS;      // buckets of four LDS entries looked at before a state goes to the global part
constexpr int kBigCap = 256;                        // high-degree tokens whose arcs are dealt out per chunk
constexpr int kStageCap = 128;                      // candidate arcs a wave parks in LDS until 64 of them can be inserted with every lane busy
constexpr int kInline = 2;                          // emitting arcs a thread relaxes itself (they sit in the state's node record)
constexpr int kNoBp = 0xFFFF;                       // work-list entry that must not write a back pointer (made by an emitting arc)
constexpr int kDstHasEps = (int)0x80000000;         // arcs_f.x flags: the destination state has epsilon arcs / is the destination of one
constexpr int kDstEpsDst = 0x40000000;
constexpr int kPdfMask = 0x3fffffff;

typedef unsigned v4u __attribute__((ext_vector_type(4)));
// The LDS part of the table is addressed through LDS-typed pointers: a `volatile` access or a pointer selected between the LDS and the
// global part through a GENERIC pointer compiles to flat_load / flat_atomic (address check, vmcnt(0) + lgkmcnt(0) behind every one of
// them: the sweep's outstanding global loads and stores were waited for at every table probe).
#ifdef RS_LIVE_GENERIC      // (A/B: profiles/micro/live_variants.sh)
typedef unsigned LdsU32;
typedef unsigned long long LdsU64;
typedef v4u LdsV4;
#else
typedef __attribute__((address_space(3))) unsigned LdsU32;
typedef __attribute__((address_space(3))) unsigned long long LdsU64;
typedef __attribute__((address_space(3))) v4u LdsV4;
#endif
__device__ __forceinline__ unsigned LdsTag(const LdsU32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
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
  int big_pre[kBigCap + 1];          // exclusive prefix of the arcs left of a chunk of high-degree tokens
};

// The live-state table of one utterance: HS entries (tag + key) in LDS, kGlobalSize behind them in global memory.
template <int HS>
struct LiveTable {
  LdsU32 *tags;                      // LDS
  LdsU64 *keys;                      // LDS
  unsigned *gtags;                   // global
  unsigned long long *gkeys;         // global
  int *n_slots, *g_used;             // LDS counters
  unsigned lds_size;                 // entries of the LDS part in use (HS; tests shrink it)
  int slot_limit;

  // slot of `state`, claiming a fresh entry if the state is not in the table yet (*claimed says so: the caller counts the frame's live
  // states); -1: no entry left
  __device__ __forceinline__ int FindOrInsert(unsigned state, bool *claimed) const {
    // LDS part: buckets of four tags, read with one 16-byte load; a state lives in the first bucket from its home bucket that had a
    // free entry when it arrived (entries are never released within a frame, so a bucket with a free entry and without the state
    // ends the search).  Linear probing one tag at a time cost 8 dependent LDS round trips per insertion at the load a heavy
    // frame puts on the table (9 k states in 9.7 k entries).
    const unsigned nb = lds_size >> 2;
    unsigned b = (unsigned)(((unsigned long long)(state * 2654435761u) * nb) >> 32);
    *claimed = false;
#pragma unroll 1
    for (int tries = 0, moved = 0; moved < kLdsBuckets && tries < 4 * kLdsBuckets + 8; tries++) {
      const v4u e = *(const volatile LdsV4 *)(tags + 4 * b);
      const int hit = e.x == state ? 0 : e.y == state ? 1 : e.z == state ? 2 : e.w == state ? 3 : -1;
      if (hit >= 0) return (int)(4 * b) + hit;
      const int j = e.x == kFree ? 0 : e.y == kFree ? 1 : e.z == kFree ? 2 : e.w == kFree ? 3 : -1;
      if (j >= 0) {
        unsigned old = kFree;
        __hip_atomic_compare_exchange_strong(tags + 4 * b + j, &old, state, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (old == kFree) { *claimed = true; return (int)(4 * b + j); }
        if (old == state) return (int)(4 * b + j);
        continue;      // another state took the entry first: look at this bucket again
      }
      b = b + 1 == nb ? 0u : b + 1;
      moved++;
    }
    // All probed LDS entries belong to other states.  Entries are never released within a frame, so every lane looking for this
    // state finds them taken too and continues here.
    *g_used = 1;
    unsigned gp = (state * 2246822519u) >> (32 - kGlobalLog);
#pragma unroll 1
    for (int probe = 0; probe < 1024; probe++) {
      unsigned e = GlbTag(&gtags[gp]);
      if (e == kFree) {
        e = atomicCAS(&gtags[gp], kFree, state);
        if (e == kFree) { *claimed = true; return HS + (int)gp; }
      }
      if (e == state) return HS + (int)gp;
      gp = (gp + 1) & (kGlobalSize - 1);
    }
    return -1;
  }
  // one lane's claim (InitDecoding, the closure): counted at once
  __device__ __forceinline__ int FindOrInsertCounted(unsigned state) const {
    bool claimed;
    const int sl = FindOrInsert(state, &claimed);
    if (claimed && atomicAdd(n_slots, 1) >= slot_limit) return -1;
    return sl;
  }
  __device__ __forceinline__ unsigned State(int slot) const { return slot < HS ? LdsTag(tags + slot) : GlbTag(&gtags[slot - HS]); }
  __device__ __forceinline__ void KeyMin(int slot, unsigned long long k) const {      // result unused: non-returning atomics
    if (slot < HS) __hip_atomic_fetch_min(keys + slot, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicMin(&gkeys[slot - HS], k);
  }
  __device__ __forceinline__ unsigned long long KeyMinRet(int slot, unsigned long long k) const {
    if (slot < HS) return __hip_atomic_fetch_min(keys + slot, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return atomicMin(&gkeys[slot - HS], k);
  }
  __device__ __forceinline__ unsigned long long KeyLoad(int slot) const {
    if (slot < HS) return __hip_atomic_load(keys + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return LoadKey(&gkeys[slot - HS]);
  }
  __device__ __forceinline__ void KeyStore(int slot, unsigned long long k) const {
    if (slot < HS) __hip_atomic_store(keys + slot, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); else StoreKey(&gkeys[slot - HS], k);
  }
};

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

#ifndef RS_LIVE_Q
#define RS_LIVE_Q 2
#endif
#ifndef RS_LIVE_HS
#define RS_LIVE_HS 10496
#endif

template <int NT, int HS>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void LiveDecodeKernel(HclgDev h, DecodeOptsDev o, BatchGeom g,
                                                       const float *__restrict__ loglikes, int ld, DecodeWork w) {
  using Ctx = LiveCtx<NT>;
  constexpr int NW = NT / 64;
  static_assert(HS % 4 == 0 && HS + kGlobalSize <= 65536, "slots are named in 16 bits");
  // One LDS object with the workgroup's scalars FIRST: an LDS word at a constant address below 64 KB is a zero base register plus
  // an instruction offset; laid out by the linker the scalars sat behind the table (0x27090...) and every one of them that a loop
  // touches held a VGPR with its address for the whole kernel -- fifteen of the 128 a wave of this workgroup has, with spills.
  struct Lds {
    alignas(16) Ctx c;
    alignas(16) unsigned long long lkeys[HS];
    alignas(16) unsigned tags[HS];
    alignas(16) int4 stage_all[NT / 64][kStageCap];         // per wave: candidate arcs waiting for insertion {arc | flags, destination, cost bits, source token}
  };
  __shared__ Lds lds;
  Ctx &c = lds.c;
  auto &tags = lds.tags;
  auto &lkeys = lds.lkeys;
  auto &stage_all = lds.stage_all;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int u = blockIdx.x, tid = threadIdx.x;
  const int T = g.d_num_frames[u];
  const size_t tab = (size_t)w.h_tab;
  int *slot_tok = w.h_slot_tok + (size_t)u * tab;            // slot -> index of its token in the frame under construction (closure-reachable states only)
  int2 *cand = reinterpret_cast<int2 *>(w.h_cand) + (size_t)u * w.h_cand_cap;      // {arc | flags of arcs_f.x, slot << 16 | source token}
  const int qcap = w.h_qcap;
  int4 *const queue0 = w.h_q4 + (size_t)u * 2 * qcap;        // work lists {slot | writer token << 16, first epsilon arc, key low, key high} ...
  int *const queue_ne0 = w.h_qne + (size_t)u * 2 * qcap;     // ... and the number of epsilon arcs of the entry's state
  int4 *big = w.h_comp + (size_t)u * kSlotCap;               // tokens with more than kInline emitting arcs: {first arc left, cost bits, token index, arcs left}
  int4 *const my_stage = stage_all[threadIdx.x >> 6];
  int4 *tokens = w.tokens + (size_t)u * w.tok_cap;
  int *frame_off = w.frame_tok_off + (size_t)u * (g.max_frames + 2);
  float *finfo = w.frame_info + (size_t)u * (g.max_frames + 1) * 4;
  const int4 *arcsf = h.arcs_f;
  const uint4 *nodes = h.nodes;                              // 4 x 16 bytes per state: {first arc, epsilon arcs, emitting arcs, 0}, emitting arc 0, emitting arc 1, -
  const float INF = INFINITY;
  const size_t ll_base = (size_t)g.d_row_base[u] + g.L;
  const int cand_cap = w.h_cand_cap;
  LiveTable<HS> tb;
  tb.tags = (LdsU32 *)tags; tb.keys = (LdsU64 *)lkeys;
  tb.gtags = w.h_gtags + (size_t)u * kGlobalSize;
  tb.gkeys = w.h_keys + (size_t)u * tab;
  tb.n_slots = &c.n_slots; tb.g_used = &c.g_used;
  tb.lds_size = w.h_lds_log > 1 && (1 << w.h_lds_log) < HS ? 1u << w.h_lds_log : (unsigned)HS;      // (tests shrink the LDS part; a multiple of four)
  tb.slot_limit = w.h_slot_limit < kSlotCap ? w.h_slot_limit : kSlotCap;                            // (tests lower it to send utterances to DecodeKernel)

  for (int i = tid; i < kGlobalSize; i += NT) { StoreKey(&tb.gkeys[i], RS_EMPTY); __hip_atomic_store(&tb.gtags[i], kFree, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  for (int i = tid; i < HS; i += NT) { tags[i] = kFree; lkeys[i] = RS_EMPTY; }
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

  int off_cur = 0, n_cur = 0;       // frame f's token list
  int off_next = 0;                 // frame under construction
  if (tid == 0) {                   // InitDecoding: the start state's token
    const int sl = tb.FindOrInsertCounted((unsigned)h.start);
    const unsigned long long k0 = PackKey(0.0f, RS_NOARC);
    tb.KeyStore(sl, k0);
    tokens[0] = make_int4(h.start, sl, -1, -2);
    slot_tok[sl] = 0;
    c.n_next = 1;
    frame_off[0] = 0;
    const uint4 sr = nodes[(size_t)h.start * 4];
    if (sr.y != 0) {
      queue0[0] = make_int4(sl | (kNoBp << 16), (int)sr.x, (int)(unsigned)(k0 & 0xFFFFFFFFull), (int)(unsigned)(k0 >> 32));
      queue_ne0[0] = (int)sr.y;
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
    // a frame holds at most kSlotCap tokens (one per slot); records name tokens of a frame in 16 bits
    const int next_cap = w.tok_cap - off_next < tb.slot_limit ? w.tok_cap - off_next : tb.slot_limit;
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
        const uint4 bsr = nodes[(size_t)bt.x * 4];
        float first_bound = INF;
        for (unsigned k = lane; k < bsr.z; k += 64) {
          const int4 arc = arcsf[bsr.x + bsr.y + k];
          first_bound = fminf(first_bound, (__int_as_float(bt.y) + (cost_offset - ll_row[(arc.x & kPdfMask) - 1])) + __int_as_float(arc.z));
        }
        const unsigned fb = wv::MinU(OrderedBits(first_bound));
        if (lane == 0 && fb < OrderedBits(INF)) atomicMin(&c.run_min, fb);
      }
      // Expansion in two steps, so that the expensive one runs with every lane busy:
      //   eval:  a lane looks at an arc -- cost, bounds, counters -- and parks the survivors (tot below "cheapest candidate seen so
      //          far + adaptive beam": anything else cannot end up below the frame's next_cutoff) in the wave's LDS stage;
      //   flush: as soon as 64 are parked, every lane takes one: table entry of the destination, ds_min on its key, candidate record
      //          (one counter update per 64 records).
      // Relaxing arcs where they were evaluated left the table code running for the lanes whose arc survived only, eight times per
      // trip (profiles/r05/live_notes.txt: 7 G wave instructions per launch, 0.77 of the SIMDs' issue slots).
      float lane_min = INF;             // the cheapest candidate this lane has seen; the wave's minimum goes to c.run_min once per trip
      int st_n = 0;                     // parked arcs (wave-uniform)
      const unsigned long long lanes_below = (1ull << lane) - 1ull;
      auto publish_min = [&]() __attribute__((always_inline)) {
        const unsigned mb = wv::MinU(OrderedBits(lane_min));      // (DPP; as __shfl_xor steps: six dependent ds_bpermute round trips per trip)
        if (lane == 0 && mb < c.run_min) __hip_atomic_fetch_min(&c.run_min, mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      auto flush = [&](int first, int cnt) __attribute__((always_inline)) {      // parked arcs [first, first + cnt), cnt <= 64 (wave-uniform)
        const bool on = lane < cnt;
        const int4 it = my_stage[first + (on ? lane : 0)];
        bool claimed = false;
        int sl = 0;
        if (on) {
          sl = tb.FindOrInsert((unsigned)it.y, &claimed);
          if (sl >= 0) tb.KeyMin(sl, ((unsigned long long)OrderedBits(__int_as_float(it.z)) << 32) | (unsigned)(it.x & kPdfMask));
        }
        const int n_claimed = __popcll(__ballot(on && claimed));
        const bool failed = __ballot(on && sl < 0) != 0ull;
        int base = 0;
        if (lane == 0) {
          base = atomicAdd(&c.n_cand, cnt);
          if (n_claimed && atomicAdd(&c.n_slots, n_claimed) + n_claimed > tb.slot_limit) c.redo = 1;
          if (failed || base + cnt > cand_cap) c.redo = 1;
        }
        base = __builtin_amdgcn_readfirstlane(base);
        if (on && base + lane < cand_cap) cand[base + lane] = make_int2(it.x, (sl << 16) | it.w);
      };
      auto eval_arc = [&](bool valid, unsigned a, const int4 arc, float lk, float cur_cost, bool is_best, int src_tok) __attribute__((always_inline)) {
        const float graph_cost = __int_as_float(arc.z);
        const float ac_cost = cost_offset - lk;
        const float tot = (cur_cost + ac_cost) + graph_cost;
        if (valid) {
          if (is_best) {
            const float nw = ((graph_cost + cost_offset) - lk) + cur_cost;      // :752-757, the reference's first bound
            local_min = fminf(local_min, nw);
          }
          local_min = fminf(local_min, tot);
          cnt_arcs++;
        }
        const float bound = fminf(FromOrdered(c.run_min), lane_min) + adaptive_beam;
        const bool keep = valid && tot < bound;
        if (keep) { cnt_insert++; lane_min = fminf(lane_min, tot); }
        const unsigned long long m = __ballot(keep);
        if (keep) my_stage[st_n + __popcll(m & lanes_below)] = make_int4((int)a | (arc.x & (kDstHasEps | kDstEpsDst)), arc.w, __float_as_int(tot), src_tok);
        st_n += __popcll(m);
        if (st_n >= 64) { st_n -= 64; flush(st_n, 64); }
      };
      {
        constexpr int Q = RS_LIVE_Q;
        for (int base = wave * 64; base < n_cur; base += Q * NT) {      // (wave-uniform trip count: the flush needs every lane)
          int2 tk[Q];
          bool act[Q];
          uint4 sr[Q];
          int4 arc[Q][kInline];
          float lk[Q][kInline];
#pragma unroll
          for (int q = 0; q < Q; q++) {
            const int i = base + q * NT + lane;
            tk[q] = i < n_cur ? *reinterpret_cast<const int2 *>(&cur[i]) : make_int2(0, __float_as_int(INF));
          }
#pragma unroll
          for (int q = 0; q < Q; q++) {
            act[q] = base + q * NT + lane < n_cur && __int_as_float(tk[q].y) <= cur_cutoff;
            const uint4 *nd = nodes + (size_t)tk[q].x * 4;      // one 64-byte record: the state's arc ranges and its first two emitting arcs
            sr[q] = act[q] ? nd[0] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int k = 0; k < kInline; k++) {
              const uint4 ar = act[q] ? nd[1 + k] : make_uint4(1u, 0u, 0u, 0u);
              arc[q][k] = make_int4((int)ar.x, (int)ar.y, (int)ar.z, (int)ar.w);
            }
          }
#pragma unroll
          for (int q = 0; q < Q; q++)
#pragma unroll
            for (int k = 0; k < kInline; k++) lk[q][k] = ll_row[(unsigned)k < sr[q].z ? (arc[q][k].x & kPdfMask) - 1 : 0];
#pragma unroll
          for (int q = 0; q < Q; q++) {
            const int i = base + q * NT + lane;
            if (act[q]) cnt_expanded++;
#pragma unroll
            for (int k = 0; k < kInline; k++)
              eval_arc(act[q] && (unsigned)k < sr[q].z, sr[q].x + sr[q].y + k, arc[q][k], lk[q][k], __int_as_float(tk[q].y), i == best_idx, i);
            if (act[q] && sr[q].z > (unsigned)kInline) big[atomicAdd(&c.n_big, 1)] = make_int4((int)(sr[q].x + sr[q].y + kInline), tk[q].y, i, (int)sr[q].z - kInline);
          }
          publish_min();
        }
      }
      __syncthreads();
      RS_LP(1);
      // ---- the arcs beyond the first two of the (few) tokens that have them, dealt out over the threads: degree prefix of a chunk of
      // such tokens in LDS, a search per arc (a wave per token was tried: its three dependent loads per token ran one token behind
      // the other -- 6.6 -> 11.6 G cycles per launch for this phase)
      {
        const int nb = c.n_big;
        int *big_pre = c.big_pre;
        for (int c0 = 0; c0 < nb; c0 += kBigCap) {
          const int nc = nb - c0 < kBigCap ? nb - c0 : kBigCap;
          const int4 *ent = big + c0;
          __syncthreads();
          // inclusive scan of the chunk's degrees by wave 0 (nc <= 256: four per lane)
          if (wave == 0) {
            int d[4], sum = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) { const int i = 4 * lane + q; d[q] = i < nc ? ent[i].w : 0; sum += d[q]; }
            const int inc = wv::ScanIncl(sum);
            int run = inc - sum;
#pragma unroll
            for (int q = 0; q < 4; q++) { const int i = 4 * lane + q; if (i < nc) big_pre[i] = run; run += d[q]; }
            if (lane == 63) big_pre[nc] = inc;
          }
          __syncthreads();
          const int total = big_pre[nc];
          for (int jb = wave * 64; jb < total; jb += 4 * NT) {      // (wave-uniform trip count: the flush needs every lane)
            unsigned a[4];
            float cc[4];
            bool on[4];
            int tki[4];
            int4 arc[4];
            float lk[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int j = jb + q * NT + lane;
              on[q] = j < total;
              const int jj = on[q] ? j : total - 1;
              int lo = 0, hi = nc;            // last entry with pre[t] <= j
              while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (big_pre[mid] <= jj) lo = mid; else hi = mid; }
              const int4 e = ent[lo];
              a[q] = (unsigned)e.x + (unsigned)(jj - big_pre[lo]);
              cc[q] = __int_as_float(e.y);
              tki[q] = e.z;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) arc[q] = arcsf[a[q]];
#pragma unroll
            for (int q = 0; q < 4; q++) lk[q] = ll_row[(arc[q].x & kPdfMask) - 1];
#pragma unroll
            for (int q = 0; q < 4; q++) eval_arc(on[q], a[q], arc[q], lk[q], cc[q], tki[q] == best_idx, tki[q]);
            publish_min();
          }
        }
        if (st_n > 0) flush(0, st_n);
        st_n = 0;
      }
      // next_cutoff = min over the candidates of (tot_cost + adaptive_beam): one LDS atomic per wave instead of a block reduction
      { const unsigned lb = wv::MinU(OrderedBits(local_min)); if (lane == 0) atomicMin(&c.min_bits, lb); }
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
        // the winner of a slot (= the candidate whose arc is left in the slot's key) appends the token, complete with back pointer
        // and arc, or -- at or above the final cutoff -- empties the key again; a token whose state has epsilon arcs goes onto the
        // closure's first work list
        constexpr int WB = 4;
        const int nc2 = c.n_cand;
        for (int ib = tid; ib < nc2; ib += WB * NT) {
          int2 cr[WB];
          unsigned long long key[WB];
          unsigned st[WB];
          bool win[WB];
          uint4 sr[WB];
#pragma unroll
          for (int q = 0; q < WB; q++) { const int i = ib + q * NT; cr[q] = cand[i < nc2 ? i : nc2 - 1]; }
#pragma unroll
          for (int q = 0; q < WB; q++) {
            const int sl = (int)((unsigned)cr[q].y >> 16);
            key[q] = tb.KeyLoad(sl);
            win[q] = ib + q * NT < nc2 && (unsigned)(key[q] & 0xFFFFFFFFull) == (unsigned)(cr[q].x & kPdfMask);
            st[q] = win[q] ? tb.State(sl) : 0u;
          }
#pragma unroll
          for (int q = 0; q < WB; q++) sr[q] = win[q] && cr[q].x < 0 && KeyCost(key[q]) < next_cutoff ? nodes[(size_t)st[q] * 4] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
          for (int q = 0; q < WB; q++) {
            if (!win[q]) continue;
            const int sl = (int)((unsigned)cr[q].y >> 16);
            if (KeyCost(key[q]) < next_cutoff) {
              const int idx = atomicAdd(&c.n_next, 1);
              if (idx < next_cap) {
                next_toks[idx] = make_int4((int)st[q], sl, cr[q].y & 0xFFFF, cr[q].x & kPdfMask);
                if (cr[q].x & (kDstHasEps | kDstEpsDst)) slot_tok[sl] = idx;      // (only the states the closure can reach need it)
                if (sr[q].y != 0) {
                  const int qp = atomicAdd(&c.q_n[0], 1);
                  if (qp < qcap) {
                    queue0[qp] = make_int4(sl | (kNoBp << 16), (int)sr[q].x, (int)(unsigned)(key[q] & 0xFFFFFFFFull), (int)(unsigned)(key[q] >> 32));
                    queue_ne0[qp] = (int)sr[q].y;
                  } else c.redo = 1;
                }
              } else {
                c.overflow = 1;
              }
            } else {
              tb.KeyStore(sl, RS_EMPTY);
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
      int qi = 0;      // (list 0 was filled by the winners' pass / InitDecoding; c.q_n[1] is 0)
      int guard_rounds = 0;
      while (c.q_n[qi] > 0) {
        const int qn = c.q_n[qi] < qcap ? c.q_n[qi] : qcap;
        const int4 *qin = queue0 + (size_t)qi * qcap;
        const int *qin_ne = queue_ne0 + (size_t)qi * qcap;
        int4 *qout = queue0 + (size_t)(qi ^ 1) * qcap;
        int *qout_ne = queue_ne0 + (size_t)(qi ^ 1) * qcap;
        __syncthreads();
        if (tid == 0) c.q_n[qi ^ 1] = 0;
        __syncthreads();
        for (int ib = 0; ib < qn; ib += NT) {
          const int i = ib + tid;
          const bool have = i < qn;
          const int4 qe = have ? qin[i] : make_int4(0, 0, 0, 0);
          const int ne_in = have ? qin_ne[i] : 0;
          const int sl = qe.x & 0xFFFF, wtok = (int)((unsigned)qe.x >> 16);
          const unsigned long long ekey = ((unsigned long long)(unsigned)qe.w << 32) | (unsigned)qe.z;
          const unsigned long long key = have ? tb.KeyLoad(sl) : RS_EMPTY;
          const int my_tok = have ? slot_tok[sl] : 0;
          const bool fresh = have && key == ekey;      // else: the slot was improved again, the entry of that improvement is on a list too
          if (fresh && wtok != kNoBp) next_toks[my_tok].z = wtok;      // the token's back pointer: the token that relaxed the winning epsilon arc
          const float cur_cost = KeyCost(key);
          const bool live = fresh && ne_in > 0 && cur_cost < closure_cutoff;
          if (live) cnt_expanded++;
          const unsigned a0 = (unsigned)qe.y, ne = live ? (unsigned)ne_in : 0u;
          const unsigned ne_max = wv::MaxU(ne);
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
            const int d0 = __builtin_amdgcn_readlane(arc.w, first);
            const bool uniform = __ballot(act && arc.w != d0) == 0ull;
            unsigned long long nkey = act ? PackKey(tot, a) : RS_EMPTY;
            bool mine = act;
            if (uniform && __popcll(m) > 1) {
              // (the smallest 64-bit key = the smallest high word, then the smallest low word among its holders: two DPP reductions)
              const unsigned khi = (unsigned)(nkey >> 32), klo = (unsigned)(nkey & 0xFFFFFFFFull);
              const unsigned mhi = wv::MinU(khi);
              const unsigned mlo = wv::MinU(khi == mhi ? klo : 0xFFFFFFFFu);
              mine = act && khi == mhi && klo == mlo;          // keys are unique (arc ids): exactly one lane
            }
            if (mine) {
              const int sl2 = tb.FindOrInsertCounted((unsigned)arc.w);
              if (sl2 < 0) { c.redo = 1; continue; }
              const bool dst_eps = arc.x < 0;
              const unsigned long long old = tb.KeyMinRet(sl2, nkey);
              const uint4 dsr = dst_eps ? nodes[(size_t)arc.w * 4] : make_uint4(0u, 0u, 0u, 0u);
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
                  qout[qp] = make_int4(sl2 | ((old == RS_EMPTY ? kNoBp : my_tok) << 16), (int)dsr.x, (int)(unsigned)(nkey & 0xFFFFFFFFull), (int)(unsigned)(nkey >> 32));
                  qout_ne[qp] = (int)dsr.y;
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
      constexpr int MB = 4;                               // tokens per thread and trip
      for (int ib = tid; ib < nn; ib += MB * NT) {
        int4 tk[MB];
#pragma unroll
        for (int q = 0; q < MB; q++) { const int i = ib + q * NT; tk[q] = next_toks[i < nn ? i : nn - 1]; }
#pragma unroll
        for (int q = 0; q < MB; q++) {
          const int i = ib + q * NT;
          if (i >= nn) continue;
          const unsigned long long key = tb.KeyLoad(tk[q].y);
          const float cst = KeyCost(key);
          next_toks[i] = make_int4(tk[q].x, __float_as_int(cst), tk[q].z, (int)(unsigned)(key & 0xFFFFFFFFull));
          if (tk[q].y >= HS) tb.KeyStore(tk[q].y, RS_EMPTY);      // (the LDS part is emptied wholesale below)
          atomicAdd(&c.chist[CostBin(cst, hist_lo, hscale)], 1u);
          if (cst < lv || (cst == lv && i < li)) { lv = cst; li = i; }
        }
      }
      {
        const unsigned long long bk0 = li == 0x7fffffff ? RS_EMPTY : PackKey(lv, (unsigned)li);
        const unsigned bhi = (unsigned)(bk0 >> 32), blo = (unsigned)(bk0 & 0xFFFFFFFFull);
        const unsigned mhi = wv::MinU(bhi);
        const unsigned mlo = wv::MinU(bhi == mhi ? blo : 0xFFFFFFFFu);
        const unsigned long long bk = ((unsigned long long)mhi << 32) | mlo;
        if (lane == 0 && bk != RS_EMPTY) atomicMin(&c.best_key, bk);      // (reset in the prologue of ProcessEmitting, behind barriers)
      }
      __syncthreads();
      // the table starts the next frame empty (global keys were emptied by their owners above or in the winners' pass)
      for (int i = tid; i < HS / 4; i += NT) reinterpret_cast<uint4 *>(tags)[i] = make_uint4(kFree, kFree, kFree, kFree);
      for (int i = tid; i < HS / 2; i += NT) reinterpret_cast<uint4 *>(lkeys)[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
      if (c.g_used) for (int i = tid; i < kGlobalSize; i += NT) __hip_atomic_store(&tb.gtags[i], kFree, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    constexpr int kHops = (NT / 64) * kStageCap * 2;      // (arc, frame of the acoustic score) pairs per block, in the waves' staging LDS
    int2 *hops = reinterpret_cast<int2 *>(&stage_all[0][0]);
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

bool DecodeLiveUsable(const HclgDev &h) { return h.num_states > 0 && h.arcs_f != nullptr && h.nodes != nullptr && h.num_arcs < (1 << 30); }
int DecodeLiveSlotCap() { return kSlotCap; }
int DecodeLiveGlobalTable() { return kGlobalSize; }
// length of the slot-indexed arrays (a slot is the position of the state's table entry: LDS part, then the global part)
int DecodeLiveTableSize() { return 65536; }

void LaunchDecodeLive(const HclgDev &h, const DecodeOptsDev &o, const BatchGeom &g, const float *loglikes, int ld,
                      const DecodeWork &w, hipStream_t s) {
  if (g.n_utts == 0) return;
  // one workgroup of 1024 threads per utterance and CU: 10496 table entries (tags + keys: 123 KB) + 32 KB of staging in LDS (all but 0.7 KB of the CU's 160).  Shapes
  // with two or four smaller workgroups per CU were measured and lost (profiles/r05/live_notes.txt).
  hipLaunchKernelGGL((LiveDecodeKernel<1024, RS_LIVE_HS>), dim3(g.n_utts), dim3(1024), 0, s, h, o, g, loglikes, ld, w);
}

}  // namespace rs
