// What bounds a workgroup's global -> LDS DMA stream (global_load_lds_dwordx4) in the pattern of GemmKernelB3J's k loop:
// every k-step each wave waits for its own DMAs of the step (counted vmcnt), the workgroup meets at a barrier, each wave
// issues its DMAs for the step DEPTH-1 ahead, then "computes" (s_sleep or MFMAs).  Sources: one L2-resident block that every
// workgroup reads (the weights: 16 KiB per k-step), or a private stream per workgroup (the activations).
// Reports GB/s per CU for waves per workgroup x workgroups per CU x ring depth x DMAs per wave and step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define DMA16(lds_addr, gptr) __asm__ volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lds_addr), "v"(gptr) : "memory")
#define VMWAIT(N) __asm__ volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory")

template <int WAVES, int DEPTH, int PER, int MFMAS, int REUSE = 3, int READS = 0>
__global__ __launch_bounds__(64 * WAVES) void k(const unsigned char *shared_src, const unsigned char *priv_src, int steps, int shared_per, float *out, int stagger) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int STAGE = WAVES * PER * 1024;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  // DMA p of a wave: p < shared_per reads the shared block (same bytes for every workgroup), the others this workgroup's stream
  const size_t shared_step = (size_t)WAVES * shared_per * 1024, priv_step = (size_t)WAVES * (PER - shared_per) * 1024;
  const unsigned char *ss = shared_src + (size_t)wave * shared_per * 1024 + lane * 16;
  const unsigned char *ps = priv_src + (size_t)blockIdx.x * priv_step * steps + (size_t)wave * (PER - shared_per) * 1024 + lane * 16;
  f32x16 acc[4];
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = (_Float16)(0.5f + lane * 0.001f); b[e] = (_Float16)(0.25f + e * 0.01f); }
  auto issue = [&](int t) __attribute__((always_inline)) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((t % DEPTH) * STAGE + wave * PER * 1024));
#pragma unroll
    for (int p = 0; p < PER; p++) {
      const unsigned char *g = p < shared_per ? ss + (size_t)t * shared_step + p * 1024 : ps + (size_t)(t / REUSE) * priv_step + (size_t)(t % REUSE) * 48 + (p - shared_per) * 1024;      // (the TDNN's three row offsets: the same rows again, three rows further on)
      DMA16(dst + p * 1024, g);
    }
  };
  if (stagger && (blockIdx.x & 256)) __builtin_amdgcn_s_sleep(100);
  for (int t = 0; t < DEPTH - 1 && t < steps; t++) issue(t);
#pragma nounroll
  for (int t = 0; t < steps; t++) {
    const int later = steps - 1 - t < DEPTH - 2 ? steps - 1 - t : DEPTH - 2;
    if (later >= 3) VMWAIT((PER * 3 > 63 ? 63 : PER * 3)); else if (later == 2) VMWAIT((PER * 2 > 63 ? 63 : PER * 2)); else if (later == 1) VMWAIT((PER > 63 ? 63 : PER)); else VMWAIT(0);
    __builtin_amdgcn_s_barrier();
    if (t + DEPTH - 1 < steps) issue(t + DEPTH - 1);
    if constexpr (READS > 0) {        // fragment reads of the stage that has landed
      const unsigned src = lds0 + (unsigned)((t % DEPTH) * STAGE) + lane * 16;
      f16x8 f[READS > 0 ? READS : 1];
#pragma unroll
      for (int i = 0; i < READS; i++) __asm__ volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[i]) : "v"(src), "n"((i % (WAVES * PER)) * 1024));
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < READS; i++) a[0] += f[i][0];
    }
    if (MFMAS > 0) {
#pragma unroll
      for (int i = 0; i < MFMAS; i++) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
    }
  }
  float s = 0; for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  if (s == 12345.f) out[0] = s;
}

// The same skeleton with the WEIGHTS going straight to registers (global_load_dwordx4, one step ahead, two register sets) and
// only the activations through LDS (2 DMAs per wave and step, two steps ahead): is an ordinary load cheaper to issue than a DMA?
#define GLOAD16(dst, gptr) __asm__ volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(gptr) : "memory")
template <int MFMAS, int READS>
__global__ __launch_bounds__(256, 2) void kw(const unsigned char *shared_src, const unsigned char *priv_src, int steps, float *out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WAVES = 4, PA = 2, PW = 4, STAGE = WAVES * PA * 1024;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const size_t shared_step = (size_t)WAVES * PW * 1024, priv_step = (size_t)WAVES * PA * 1024;
  const unsigned char *ss = shared_src + (size_t)wave * PW * 1024 + lane * 16;
  const unsigned char *ps = priv_src + (size_t)blockIdx.x * priv_step * steps + (size_t)wave * PA * 1024 + lane * 16;
  f32x16 acc[4];
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  f16x8 a;
  for (int e = 0; e < 8; e++) a[e] = (_Float16)(0.5f + lane * 0.001f);
  f16x8 w0[PW], w1[PW];
  auto issueA = [&](int t) __attribute__((always_inline)) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((t % 3) * STAGE + wave * PA * 1024));
#pragma unroll
    for (int p = 0; p < PA; p++) DMA16(dst + p * 1024, ps + (size_t)(t / 3) * priv_step + (size_t)(t % 3) * 48 + p * 1024);
  };
#define ISSUE_W(SET, T) _Pragma("unroll") for (int p = 0; p < PW; p++) GLOAD16(SET[p], ss + (size_t)(T) * shared_step + p * 1024)
#define STEP(SET, NEXT, T)                                                                               \
  {                                                                                                      \
    if ((T) + 1 < steps) { __asm__ volatile("s_waitcnt vmcnt(2)" : "+v"(SET[0]), "+v"(SET[1]), "+v"(SET[2]), "+v"(SET[3]) : : "memory"); }      \
    else { __asm__ volatile("s_waitcnt vmcnt(0)" : "+v"(SET[0]), "+v"(SET[1]), "+v"(SET[2]), "+v"(SET[3]) : : "memory"); }                      \
    __builtin_amdgcn_s_barrier();                                                                        \
    if ((T) + 1 < steps) { ISSUE_W(NEXT, (T) + 1); }                                                     \
    if ((T) + 2 < steps) issueA((T) + 2);                                                                \
    if constexpr (READS > 0) {                                                                           \
      const unsigned src = lds0 + (unsigned)(((T) % 3) * STAGE) + lane * 16;                             \
      f16x8 f[READS > 0 ? READS : 1];                                                                    \
      _Pragma("unroll") for (int i = 0; i < READS; i++) __asm__ volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[i]) : "v"(src), "n"((i % (WAVES * PA)) * 1024)); \
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
      _Pragma("unroll") for (int i = 0; i < READS; i++) a[0] += f[i][0];                                 \
    }                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < MFMAS; i++) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(SET[i & 3], a, acc[i & 3], 0, 0, 0); \
  }
  ISSUE_W(w0, 0);
  issueA(0);
  if (steps > 1) issueA(1);
  int t = 0;
#pragma nounroll
  for (; t + 2 <= steps; t += 2) {
    STEP(w0, w1, t)
    STEP(w1, w0, t + 1)
  }
  if (t < steps) STEP(w0, w1, t)
#undef STEP
#undef ISSUE_W
  float s = 0; for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  if (s == 12345.f) out[0] = s;
}
template <int MFMAS, int READS>
void runw(const char *name, const unsigned char *shared_src, const unsigned char *priv_src, float *out, int steps) {
  const int blocks = 512;
  const size_t smem = 3 * 4 * 2 * 1024;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kw<MFMAS, READS>), dim3(blocks), dim3(256), smem, 0, shared_src, priv_src, steps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes_cu = 2.0 * 4 * 6 * 1024.0 * steps;
  printf("%-46s waves 4 x 2 wg/CU, 2 DMAs + 4 register loads per wave, mfma %2d: %7.1f us  %6.1f GB/s per CU  %5.2f us per step\n", name, MFMAS, best * 1e3,
         bytes_cu / (best * 1e-3) / 1e9, best * 1e3 / steps);
}

template <int WAVES, int DEPTH, int PER, int MFMAS, int REUSE = 3, int READS = 0>
void run(const char *name, int per_cu, int shared_per, const unsigned char *shared_src, const unsigned char *priv_src, float *out, int steps) {
  const int blocks = 256 * per_cu;
  size_t smem = (size_t)DEPTH * WAVES * PER * 1024;
  if (per_cu == 1 && smem < 90 * 1024) smem = 90 * 1024;           // keep a second workgroup off the CU
  if (per_cu == 2 && smem > 80 * 1024) { printf("%s: skipped (LDS)\n", name); return; }
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k<WAVES, DEPTH, PER, MFMAS, REUSE, READS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WAVES, DEPTH, PER, MFMAS, REUSE, READS>), dim3(blocks), dim3(64 * WAVES), smem, 0, shared_src, priv_src, steps, shared_per, out, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes_cu = (double)per_cu * WAVES * PER * 1024.0 * steps;
  printf("%-46s waves %d x %d wg/CU depth %d dma/wave %d (shared %d) mfma %2d: %7.1f us  %6.1f GB/s per CU  %5.2f us per step\n", name, WAVES, per_cu, DEPTH, PER, shared_per,
         MFMAS, best * 1e3, bytes_cu / (best * 1e-3) / 1e9, best * 1e3 / steps);
}

int main() {
  const int steps = 96;
  unsigned char *shared_src, *priv_src; float *out;
  hipMalloc(&shared_src, (size_t)8 * 8 * 1024 * steps + (1 << 20));
  const size_t priv_bytes = (size_t)512 * 8 * 8 * 1024 * steps;       // 512 workgroups x up to 64 KiB per step
  hipMalloc(&priv_src, priv_bytes + (1 << 20));
  hipMalloc(&out, 4096);
  hipMemset(shared_src, 1, (size_t)8 * 8 * 1024 * steps);
  hipMemset(priv_src, 2, priv_bytes);
  // the GEMM's shape today: 4 waves, 2 workgroups per CU, depth 3, 6 DMAs per wave (4 weights shared + 2 activations private)
  run<4, 3, 6, 0, 1>("B3J WM=1 shape, private stream never re-read", 2, 4, shared_src, priv_src, out, steps);
  run<4, 3, 6, 0>("B3J WM=1 shape, no compute", 2, 4, shared_src, priv_src, out, steps);
  run<4, 3, 6, 0, 3, 12>("B3J WM=1 shape, 12 fragment reads per step", 2, 4, shared_src, priv_src, out, steps);
  run<4, 3, 6, 24, 3, 12>("B3J WM=1 shape, 12 reads + 24 MFMAs", 2, 4, shared_src, priv_src, out, steps);
  run<4, 3, 6, 24>("B3J WM=1 shape, 24 MFMAs per step", 2, 4, shared_src, priv_src, out, steps);
  runw<0, 0>("weights to registers, no compute", shared_src, priv_src, out, steps);
  runw<0, 8>("weights to registers, 8 fragment reads", shared_src, priv_src, out, steps);
  runw<24, 8>("weights to registers, 8 reads + 24 MFMAs", shared_src, priv_src, out, steps);
  runw<24, 0>("weights to registers, 24 MFMAs", shared_src, priv_src, out, steps);
  run<4, 3, 6, 0>("  one workgroup per CU", 1, 4, shared_src, priv_src, out, steps);
  run<4, 3, 6, 0>("  all shared (L2 hits only)", 2, 6, shared_src, priv_src, out, steps);
  run<4, 3, 6, 0>("  all private (HBM stream)", 2, 0, shared_src, priv_src, out, steps);
  run<4, 2, 6, 0>("  depth 2", 2, 4, shared_src, priv_src, out, steps);
  run<4, 3, 3, 0>("  half the bytes per step", 2, 2, shared_src, priv_src, out, steps);
  run<4, 3, 12, 0>("  twice the bytes per step (depth 3 = 144K)", 1, 8, shared_src, priv_src, out, steps);
  run<8, 3, 4, 0>("B3J WM=2 shape (8 waves, 4 DMAs each)", 1, 2, shared_src, priv_src, out, steps);
  run<8, 4, 4, 0>("  depth 4", 1, 2, shared_src, priv_src, out, steps);
  run<8, 5, 4, 0>("  depth 5", 1, 2, shared_src, priv_src, out, steps);
  run<8, 4, 4, 24>("  depth 4, 24 MFMAs", 1, 2, shared_src, priv_src, out, steps);
  run<8, 4, 4, 24, 3, 12>("  depth 4, 12 reads + 24 MFMAs", 1, 2, shared_src, priv_src, out, steps);
  run<8, 4, 3, 24, 3, 12>("  depth 4, 3 DMAs per wave, 12 reads + 24 MFMAs", 1, 2, shared_src, priv_src, out, steps);
  run<8, 4, 4, 0>("  depth 4 all shared", 1, 4, shared_src, priv_src, out, steps);
  run<8, 4, 4, 0>("  depth 4 all private", 1, 0, shared_src, priv_src, out, steps);
  run<16, 3, 2, 0>("16 waves, 2 DMAs each, depth 3", 1, 1, shared_src, priv_src, out, steps);
  run<2, 4, 16, 0>("2 loader waves, 16 DMAs each, depth 4 (128K)", 1, 8, shared_src, priv_src, out, steps);
  run<1, 3, 20, 0>("1 loader wave, 20 DMAs, depth 3 (60K)", 1, 10, shared_src, priv_src, out, steps);
  return 0;
}
