// Helpers shared by the token-list searches (decode_kernels.hip: dense per-state tables; decode_live.hip: live-state table):
// order-preserving cost keys, block reductions, exact selection of the k-th smallest token cost.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>

#include "kernels.h"

namespace rs {
namespace tok {

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
__device__ __forceinline__ float KeyCost(unsigned long long k) { return FromOrdered((unsigned)(k >> 32)); }
// best[] is only ever touched with device-scope atomics / L1-bypassing loads (atomics execute in L2)
__device__ __forceinline__ unsigned long long LoadKey(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void StoreKey(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef RS_PREFIX_CAP
#define RS_PREFIX_CAP 8192
#endif
constexpr int kPrefixCap = RS_PREFIX_CAP;

#ifdef RS_DECODE_PROFILE
#define RS_TP(i) do { __syncthreads(); long long _n = clock64(); if (threadIdx.x == 0) prof[i] += _n - t_last; t_last = clock64(); } while (0)
#else
#define RS_TP(i) do { } while (0)
#endif

template <int NT, int PCAP = kPrefixCap>
struct BlockCtx {
  float red_f[NT / 64];
  int red_i[NT / 64];
  unsigned hist[256];
  int n_next, q_n[2], overflow, error;
  unsigned run_min;                  // ordered bits of the smallest candidate cost seen so far in this frame (a bound on next_cutoff)
  int n_cand;                        // (destination state, arc) pairs recorded by the arc loop
  int pre[PCAP + 1];                 // exclusive prefix of the tokens' emitting out-degrees
  unsigned tok_a0[PCAP];             // first emitting arc of each token's state, and the token's cost: the arc loop then needs no
  float tok_cost[PCAP];              // global access to find out WHICH arc it is working on
  float bcast_f[2];
  int bcast_i[4];
  unsigned long long counters[8];
};

// min over the block of (v, idx), lowest idx on ties.  Contains barriers: call from uniform control flow.
template <int NT, class Ctx>
__device__ __forceinline__ void BlockMinArg(Ctx &c, float v, int idx, float *out_v, int *out_i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(v, o, 64);
    int oi = __shfl_xor(idx, o, 64);
    if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { c.red_f[wave] = v; c.red_i[wave] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float bv = c.red_f[0];
    int bi = c.red_i[0];
    for (int w = 1; w < NT / 64; w++)
      if (c.red_f[w] < bv || (c.red_f[w] == bv && c.red_i[w] < bi)) { bv = c.red_f[w]; bi = c.red_i[w]; }
    c.bcast_f[0] = bv;
    c.bcast_i[0] = bi;
  }
  __syncthreads();
  *out_v = c.bcast_f[0];
  *out_i = c.bcast_i[0];
  __syncthreads();
}

// k-th smallest (0-based) cost of toks[0..n): exact radix select on the order-preserving bit pattern
// (the value std::nth_element leaves at position k, lattice-faster-decoder.cc:679-695).  Bits shared by the
// smallest and largest cost are skipped (they would pile every token onto one histogram bin); the 256-bin prefix
// scan of each pass is one wavefront of shuffles.
template <int NT, class Ctx>
__device__ float BlockKthSmallest(Ctx &c, const int4 *toks, int n, int k, float min_cost) {
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < n; i += NT) mx = fmaxf(mx, __int_as_float(toks[i].y));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) c.red_f[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = c.red_f[0];
  for (int w = 1; w < NT / 64; w++) mx = fmaxf(mx, c.red_f[w]);
  __syncthreads();
  const unsigned umin = OrderedBits(min_cost), umax = OrderedBits(mx);
  const unsigned diff = umin ^ umax;
  if (diff == 0) return min_cost;
  const int nbits = (32 - __clz((int)diff) + 7) & ~7;
  unsigned mask = nbits >= 32 ? 0u : ~((1u << nbits) - 1u);
  unsigned prefix = umin & mask;
  int kk = k;
  for (int shift = nbits - 8; shift >= 0; shift -= 8) {
    for (int i = threadIdx.x; i < 256; i += NT) c.hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += NT) {
      unsigned uu = OrderedBits(__int_as_float(toks[i].y));
      if ((uu & mask) == prefix) atomicAdd(&c.hist[(uu >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int l = threadIdx.x;
      const int h0 = (int)c.hist[4 * l], h1 = (int)c.hist[4 * l + 1], h2 = (int)c.hist[4 * l + 2], h3 = (int)c.hist[4 * l + 3];
      const int tot = h0 + h1 + h2 + h3;
      int inc = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o, 64); if (l >= o) inc += v; }
      const int exc = inc - tot;
      if (exc <= kk && kk < inc) {
        int acc = exc, b = 4 * l;
        if (acc + h0 <= kk) { acc += h0; b++; if (acc + h1 <= kk) { acc += h1; b++; if (acc + h2 <= kk) { acc += h2; b++; } } }
        c.bcast_i[1] = b;
        c.bcast_i[2] = kk - acc;
      }
    }
    __syncthreads();
    prefix |= ((unsigned)c.bcast_i[1]) << shift;
    mask |= 255u << shift;
    kk = c.bcast_i[2];
  }
  __syncthreads();
  return FromOrdered(prefix);
}

// The same k-th smallest cost when it is known to lie in [lo, hi) and the frame is large: ONE histogram sweep over 256 linear bins
// of that range (float subtract / multiply / truncate are monotone, so bins are in value order), then a sweep that collects the
// values of the bin holding rank k (a few dozen of 16 k) and ranks them directly -- two sweeps over the frame's tokens instead of
// the radix select's six.  Falls back to the radix select when that bin is crowded.  cand: 256 floats of LDS.
template <int NT, class Ctx>
__device__ float BlockKthSmallestHist(Ctx &c, const int4 *toks, int n, int k, float lo, float hi, float *cand, int *cand_n) {
  const int tid = threadIdx.x, lane = tid & 63;
  const float scale = hi > lo ? 256.0f / (hi - lo) : 0.f;
  for (int i = tid; i < 256; i += NT) c.hist[i] = 0;
  if (tid == 0) *cand_n = 0;
  __syncthreads();
  for (int i = tid; i < n; i += NT) {
    const float v = __int_as_float(toks[i].y);
    if (v < hi) { int b = (int)((v - lo) * scale); b = b > 255 ? 255 : (b < 0 ? 0 : b); atomicAdd(&c.hist[b], 1u); }
  }
  __syncthreads();
  if (tid < 64) {      // wave 0: the bin that holds rank k (four bins per lane)
    const int h0 = (int)c.hist[4 * lane], h1 = (int)c.hist[4 * lane + 1], h2 = (int)c.hist[4 * lane + 2], h3 = (int)c.hist[4 * lane + 3];
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
    if (lane == 63 && inc <= k) c.bcast_i[2] = -1;      // fewer than k + 1 values below hi (caller error): radix select
  }
  __syncthreads();
  const int bin = c.bcast_i[0], kk = c.bcast_i[1], cnt = c.bcast_i[2];
  __syncthreads();
  if (cnt < 0 || cnt > 256) return BlockKthSmallest<NT>(c, toks, n, k, lo);
  for (int i = tid; i < n; i += NT) {
    const float v = __int_as_float(toks[i].y);
    if (v < hi) { int b = (int)((v - lo) * scale); b = b > 255 ? 255 : (b < 0 ? 0 : b); if (b == bin) cand[atomicAdd(cand_n, 1)] = v; }
  }
  __syncthreads();
  if (tid < cnt) {
    const float v = cand[tid];
    int lt = 0, le = 0;
    for (int j = 0; j < cnt; j++) { const float x = cand[j]; lt += (int)(x < v); le += (int)(x <= v); }
    if (lt <= kk && kk < le) c.bcast_f[1] = v;
  }
  __syncthreads();
  const float ans = c.bcast_f[1];
  __syncthreads();
  return ans;
}
}  // namespace tok
}  // namespace rs
