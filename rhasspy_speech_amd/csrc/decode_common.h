// Device helpers shared by the LDS-resident decoders (decode_dense.hip, decode_reg.hip): order-preserving cost
// keys, block reductions, exact radix select, and the final-cost / traceback / result stage.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>

#include "kernels.h"

namespace rs {
namespace dd {

#define RS_EMPTY 0xFFFFFFFFFFFFFFFFull
#define RS_NOARC 0xFFFFFFFFu

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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, i.e. every
// barrier of the frame loop would wait for the back-pointer stores just issued and for the log-likelihood prefetch of the
// next frame (~1 us each); nothing in the loop communicates through global memory.
__device__ __forceinline__ void LdsBarrier() { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NW>
struct Red {
  float f[NW];
  int i[NW];
  int c0[NW], c1[NW], c2[NW];
  float bf[2];
  int bi[4];
  int changed;
  unsigned hist[256];
  float cand[64];
  int ncand;
  unsigned long long ctr[8];
  double dsum[2 * NW];
};

template <int NT>
__device__ __forceinline__ void BlockMinArg(Red<NT / 64> &r, float v, int idx, float *ov, int *oi) {
  constexpr int NW = NT / 64;
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

template <int NT>
__device__ __forceinline__ void BlockSum3(Red<NT / 64> &r, int a, int b, int c, int *oa, int *ob, int *oc) {
  constexpr int NW = NT / 64;
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

// exact k-th smallest (0-based) of the finite entries of cost[0..S): radix select on the order-preserving bit
// pattern.  The bits shared by the smallest and the largest finite cost are skipped (a frame's costs share sign,
// exponent and usually several mantissa bits, which would otherwise pile every element onto one histogram bin
// and serialise the LDS atomics); the 256-bin prefix scan of each pass is one wavefront of shuffles.
template <int NT>
__device__ float KthSmallest(Red<NT / 64> &r, const float *cost, int S, int k, float min_cost) {
  constexpr int NW = NT / 64;
  // block max of the finite costs
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < S; i += NT) { const float c = cost[i]; if (c < INFINITY) mx = fmaxf(mx, c); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) r.f[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = r.f[0];
#pragma unroll
  for (int w = 1; w < NW; w++) mx = fmaxf(mx, r.f[w]);
  __syncthreads();
  const unsigned umin = OrderedBits(min_cost), umax = OrderedBits(mx);
  const unsigned diff = umin ^ umax;
  if (diff == 0) return min_cost;
  const int nbits = (32 - __clz((int)diff) + 7) & ~7;          // differing low bits, rounded up to whole digits
  unsigned mask = nbits >= 32 ? 0u : ~((1u << nbits) - 1u);
  unsigned prefix = umin & mask;
  int kk = k;
  for (int shift = nbits - 8; shift >= 0; shift -= 8) {
    for (int i = threadIdx.x; i < 256; i += NT) r.hist[i] = 0;
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
      if (exc <= kk && kk < inc) {
        int acc = exc, bb = 4 * l;
        if (acc + h0 <= kk) { acc += h0; bb++; if (acc + h1 <= kk) { acc += h1; bb++; if (acc + h2 <= kk) { acc += h2; bb++; } } }
        r.bi[1] = bb;
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

template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ int DppZ(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, BOUND); }
// inclusive prefix sum over the 64 lanes of a wavefront (DPP row shifts + row broadcasts, no LDS traffic)
__device__ __forceinline__ int WaveScanIncl(int v) {
  v += DppZ<0x111, 0xF, true>(v);      // row_shr:1
  v += DppZ<0x112, 0xF, true>(v);      // row_shr:2
  v += DppZ<0x114, 0xF, true>(v);      // row_shr:4
  v += DppZ<0x118, 0xF, true>(v);      // row_shr:8
  v += DppZ<0x142, 0xA, false>(v);     // row_bcast:15 into rows 1 and 3
  v += DppZ<0x143, 0xC, false>(v);     // row_bcast:31 into rows 2 and 3
  return v;
}

// Exact k-th smallest (0-based) of the finite entries of cost[0..S), all of which lie in [lo, hi]: ONE histogram pass
// over 256 linear bins (float subtract / multiply / truncate are monotone, so bin order agrees with value order), every
// wave scans the histogram itself, then the handful of values in the bin that holds rank k are ranked directly.
// Falls back to the radix select when that bin is crowded.  The histogram pass is the caller's (KthBin per finite
// entry into r.hist, then a barrier) so that it can share a loop; KthFromHist needs r.ncand == 0 on entry and leaves
// r.hist zeroed and r.ncand == 0.  *n_le = number of entries <= the result.
__device__ __forceinline__ int KthBin(float c, float lo, float scale) {
  const int b = (int)((c - lo) * scale);
  return b > 255 ? 255 : b;
}
template <int NT>
__device__ float KthFromHist(Red<NT / 64> &r, const float *cost, int S, int k, float lo, float hi, int *n_le) {
  const int tid = threadIdx.x, lane = tid & 63;
  const float scale = hi > lo ? 255.0f / (hi - lo) : 0.f;
  const uint4 hv = *reinterpret_cast<const uint4 *>(&r.hist[4 * lane]);
  const int h0 = (int)hv.x, h1 = (int)hv.y, h2 = (int)hv.z, h3 = (int)hv.w;
  const int tot = h0 + h1 + h2 + h3;
  const int inc = WaveScanIncl(tot), exc = inc - tot;
  const bool hit = (exc <= k) & (k < inc);
  int bb = 4 * lane, acc = exc, m = h0;
  if (acc + h0 <= k) { acc += h0; bb++; m = h1; if (acc + h1 <= k) { acc += h1; bb++; m = h2; if (acc + h2 <= k) { acc += h2; bb++; m = h3; } } }
  const unsigned long long hm = __ballot(hit);
  const int hl = hm ? __ffsll((long long)hm) - 1 : 0;
  const int bin = __builtin_amdgcn_readlane(bb, hl), before = __builtin_amdgcn_readlane(acc, hl), cnt = __builtin_amdgcn_readlane(m, hl);
  const int kk = k - before;
  if (hm == 0ull || cnt > 64) {
    // k beyond the number of entries (caller error), or a crowded bin (many equal costs): radix select
    __syncthreads();
    const float v = KthSmallest<NT>(r, cost, S, k, lo);
    int le = 0;
    for (int i = tid; i < S; i += NT) le += (int)(cost[i] <= v);
    for (int i = tid; i < 256; i += NT) r.hist[i] = 0;
    if (tid == 0) r.bi[3] = 0;
    __syncthreads();
    atomicAdd(&r.bi[3], le);
    __syncthreads();
    *n_le = r.bi[3];
    return v;
  }
  for (int i = tid; i < S; i += NT) {
    const float c = cost[i];
    if (c < INFINITY && KthBin(c, lo, scale) == bin) r.cand[atomicAdd(&r.ncand, 1)] = c;
  }
  LdsBarrier();
  for (int i = tid; i < 256; i += NT) r.hist[i] = 0;
  if (tid < cnt) {
    const float v = r.cand[tid];
    int lt = 0, le = 0;
    for (int j = 0; j < cnt; j++) { const float x = r.cand[j]; lt += (int)(x < v); le += (int)(x <= v); }
    if (lt <= kk && kk < le) { r.bf[0] = v; r.bi[3] = before + le; }
  }
  if (tid == 0) r.ncand = 0;
  LdsBarrier();
  *n_le = r.bi[3];
  return r.bf[0];
}

// Final costs (ComputeFinalCosts), best-path traceback (GetBestPath) and result records.  cost_cur = the last frame's
// dense costs in LDS; bp = back-pointer rows [T+1][S] in HBM; smem = the dynamic LDS region (reused for staging).
template <int NT>
__device__ void FinishUtterance(Red<NT / 64> &red, const HclgDev &h, const BatchGeom &g, const float *__restrict__ loglikes, int ld,
                                const DenseWork &w, const float *cost_cur, const int *bp, const float *finfo, unsigned char *smem,
                                int smem_bytes, int u, int T, int S, size_t ll_base, int error, unsigned long long n_expanded,
                                unsigned long long n_arcs, unsigned long long n_insert, unsigned long long n_alive,
                                int max_active_frames, int min_active_frames, size_t counter_slot) {
  constexpr int NW = NT / 64;
  const int tid = threadIdx.x;
  const float INF = INFINITY;
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
  BlockMinArg<NT>(red, lv1, li1, &b1, &i1);
  BlockMinArg<NT>(red, lv2, li2, &b2, &i2);
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
#ifdef RS_DECODE_PROFILE
  long long fin_t[5] = {clock64(), 0, 0, 0, 0};
  int fin_blocks = 0;
#define RS_FIN_T(i) fin_t[i] = clock64()
#define RS_FIN_BLOCK fin_blocks++
#else
#define RS_FIN_T(i)
#define RS_FIN_BLOCK
#endif
  // ---- traceback (GetBestPath).  The back-pointer rows are staged through LDS a block of frames at a time so that the
  // inherently sequential walk runs at LDS latency: a block is a contiguous run of the buffer, fetched 16 bytes per thread and
  // load, and the NEXT block (the walk always continues with the frames right below this one) is in flight, in registers, while
  // thread 0 walks the current one.  Arc sources come from LDS too (16 bits per arc) when the table fits beside two rows.
  constexpr int kPF = 8;                                  // 16-byte loads per thread and block that travel ahead (what a block holds beyond them is copied when it is staged)
  const bool src_in_lds = S <= 0x7fff && h.num_arcs > 0 && (size_t)2 * h.num_arcs + (size_t)8 * S + 64 <= (size_t)smem_bytes;
  unsigned short *lds_src = reinterpret_cast<unsigned short *>(smem);       // [num_arcs]: source state | epsilon flag << 15
  const int src_bytes = src_in_lds ? (2 * h.num_arcs + 15) & ~15 : 0;
  int *rows = reinterpret_cast<int *>(smem + src_bytes);  // the staged run (from a 16-byte boundary of the buffer)
  const int cap_ints = (smem_bytes - src_bytes) / 4 - 4;
  const int rows_cap = cap_ints / S;                      // >= 1: the launchers' LDS holds the search's own 12 bytes per state
  int *path = w.path + (size_t)u * w.path_cap * 2;        // (arc, source frame) pairs, last arc first
  int path_len = 0;
  __syncthreads();
  if (ok) {
    if (src_in_lds)
      for (int i = tid; i < h.num_arcs; i += NT) { const int sx = h.arc_srcx[i]; lds_src[i] = (unsigned short)((sx & 0x7fff) | (sx < 0 ? 0x8000 : 0)); }
    int F = T, st = reached ? i1 : i2;
    bool done = false;
    const long long idx0 = (long long)(bp - w.bp);        // my first row in the buffer, in ints (the buffer itself is 16-byte aligned)
    // (eight named registers: as an array the block in flight was kept in scratch memory, with a wait after every load)
    int4 pf0 = {0, 0, 0, 0}, pf1 = pf0, pf2 = pf0, pf3 = pf0, pf4 = pf0, pf5 = pf0, pf6 = pf0, pf7 = pf0;
    static_assert(kPF == 8, "the block in flight is eight named registers");
    int pf_tail = 0;
    int lo = F - rows_cap + 1 > 0 ? F - rows_cap + 1 : 0; // rows lo..F
    auto fetch = [&](int lo_, int F_) {
      const long long first = idx0 + (long long)lo_ * S, last = idx0 + (long long)(F_ + 1) * S, start = first & ~3ll;
      const int nfull = (int)((last - start) >> 2), ntail = (int)((last - start) & 3);
      const int4 *src4 = reinterpret_cast<const int4 *>(w.bp + start);
#define RS_PF_LOAD(q) if ((q) * NT + tid < nfull) pf##q = src4[(q) * NT + tid];
      RS_PF_LOAD(0) RS_PF_LOAD(1) RS_PF_LOAD(2) RS_PF_LOAD(3) RS_PF_LOAD(4) RS_PF_LOAD(5) RS_PF_LOAD(6) RS_PF_LOAD(7)
#undef RS_PF_LOAD
      if (tid < ntail) pf_tail = w.bp[start + 4ll * nfull + tid];
    };
    fetch(lo, F);
    RS_FIN_T(1);
    while (!done) {
      RS_FIN_BLOCK;
      const long long first = idx0 + (long long)lo * S, last = idx0 + (long long)(F + 1) * S, start = first & ~3ll;
      const int skip = (int)(first - start), nfull = (int)((last - start) >> 2), ntail = (int)((last - start) & 3);
      __syncthreads();                                    // (everybody is done with the previous block)
#define RS_PF_STAGE(q) if ((q) * NT + tid < nfull) reinterpret_cast<int4 *>(rows)[(q) * NT + tid] = pf##q;
      RS_PF_STAGE(0) RS_PF_STAGE(1) RS_PF_STAGE(2) RS_PF_STAGE(3) RS_PF_STAGE(4) RS_PF_STAGE(5) RS_PF_STAGE(6) RS_PF_STAGE(7)
#undef RS_PF_STAGE
      for (int i = kPF * NT + tid; i < nfull; i += NT) reinterpret_cast<int4 *>(rows)[i] = reinterpret_cast<const int4 *>(w.bp + start)[i];      // (rows of more than 32 NT states)
      if (tid < ntail) rows[4 * nfull + tid] = pf_tail;
      __syncthreads();
      const int next_F = lo - 1, next_lo = next_F - rows_cap + 1 > 0 ? next_F - rows_cap + 1 : 0;
      if (next_F >= 0) fetch(next_lo, next_F);
      if (tid == 0) {
        // (the walk is ~300 dependent hops of one lane, ~470 clocks each: two LDS reads and as few instructions as possible per hop --
        // the row's address and the path's are carried along, the choice of the source table is made outside the loop.  It is the
        // walk, not the trips to memory, that the traceback's time is made of: with two blocks in flight instead of one it takes
        // the same 143 k clocks for 303 hops in 25 blocks, profiles/micro/prof_reg_decode.sh)
        const int *row = rows + skip + (F - lo) * S;
        int *out = path + 2 * path_len;
        int room = w.path_cap - path_len;
        if (src_in_lds) {
          while (true) {
            const int arc = row[st];
            if (arc < 0) { done = true; break; }
            const unsigned sx = lds_src[arc];
            const int eps = (int)(sx >> 15);                // 1: epsilon arc, the source is in the same row
            F -= 1 - eps;
            if (room > 0) { out[0] = arc; out[1] = F; out += 2; }
            room--;
            st = (int)(sx & 0x7fffu);
            row -= eps ? 0 : S;
            if (F < lo) break;       // need older rows
          }
        } else {
          while (true) {
            const int arc = row[st];
            if (arc < 0) { done = true; break; }
            const int sx = h.arc_srcx[arc];                 // source state | (epsilon arc ? 1 << 31 : 0)
            const int eps = (int)((unsigned)sx >> 31);
            F -= 1 - eps;
            if (room > 0) { out[0] = arc; out[1] = F; out += 2; }
            room--;
            st = sx & 0x7fffffff;
            row -= eps ? 0 : S;
            if (F < lo) break;
          }
        }
        path_len = w.path_cap - room;
        red.bi[0] = done ? 1 : 0; red.bi[1] = F; red.bi[2] = st; red.bi[3] = path_len;
      }
      __syncthreads();
      done = red.bi[0] != 0; F = red.bi[1]; st = red.bi[2]; path_len = red.bi[3];
      lo = next_lo;
      if (!done && F != next_F) { done = true; path_len = w.path_cap + 1; }      // (cannot happen: a walk leaves its block one frame below it)
    }
  }
  __syncthreads();
  RS_FIN_T(2);
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
    RS_FIN_T(3);
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
      long long *c8 = w.counters + counter_slot * 8;
      // += : a resumable decoder has already flushed the counts of earlier time slabs (the buffer starts zeroed)
      for (int i = 0; i < 4; i++) c8[i] += (long long)ctr[i];
      c8[4] = 0; c8[5] += max_active_frames; c8[6] += min_active_frames; c8[7] = error ? 2 : 0;
#ifdef RS_DECODE_PROFILE
      RS_FIN_T(4);
      if (u == 0)
        printf("reg decode finish cycles: pick + arc sources %lld, traceback %lld (%d blocks of %d rows, path %d), path costs %lld, words + results %lld\n",
               fin_t[1] - fin_t[0], fin_t[2] - fin_t[1], fin_blocks, rows_cap, path_len, fin_t[3] - fin_t[2], fin_t[4] - fin_t[3]);
#endif
    }
  }
#undef RS_FIN_T
#undef RS_FIN_BLOCK
}

}  // namespace dd
}  // namespace rs
