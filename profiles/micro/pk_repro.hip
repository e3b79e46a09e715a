// Does packed FP32 VALU arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) give wrong results when waves of ANOTHER kernel on the
// same CU issue MFMAs?  (MfccKernel built with compiler-formed packed math produced wrong features in 16-lane
// groups while GemmKernelB3 of another decode call shared its CUs; the library has avoided the instructions since.)
// Isolated repro attempt: a known-answer victim (every lane runs chains of packed operations on exactly representable values, with
// and without LDS traffic between them, like the feature kernel's window / FFT stages) beside a burner that keeps the matrix cores of
// every CU busy from another stream, both sized to co-reside (victim: 4 waves x 2 workgroups, burner: 4 waves x 2 per CU, small
// register footprints, 16 KiB of LDS for the victim).  Prints the number of wrong lanes per configuration.
//   hipcc --offload-arch=gfx950 -O2 -o pk_repro pk_repro.hip && ./pk_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool LDS>
__global__ __launch_bounds__(256, 2) void victim(unsigned *errors, unsigned *first_bad, int rounds) {
  __shared__ v2f stage[LDS ? 2048 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v2f x[8];
  const v2f one = {1.0f, 1.0f}, zero = {0.0f, 0.0f}, two = {2.0f, 2.0f}, half = {0.5f, 0.5f};
  for (int k = 0; k < 8; k++) x[k] = v2f{(float)(lane + 8 * k), (float)(lane + 8 * k + 1000)};
  for (int it = 0; it < rounds; it++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      v2f t;
      // x <- ((x * 2) * 0.5) * 1 + 1 + 0: every step exact in FP32 while the values stay below 2^23
      __asm__ volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(x[k]), "v"(two));
      __asm__ volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(half));
      __asm__ volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(one), "v"(one));
      __asm__ volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x[k]) : "v"(t), "v"(zero));
    }
    if (LDS) {      // through LDS and back, rotated by one lane within the wave's private slice (nobody else touches it)
#pragma unroll
      for (int k = 0; k < 8; k++) stage[wave * 512 + k * 64 + lane] = x[k];
      __builtin_amdgcn_wave_barrier();
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 8; k++) x[k] = stage[wave * 512 + k * 64 + ((lane + 1) & 63)];
      __builtin_amdgcn_wave_barrier();
      __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  unsigned bad = 0;
  for (int k = 0; k < 8; k++) {
    const int src = LDS ? (lane + rounds) & 63 : lane;        // the lane whose chain this register now holds
    const float e0 = (float)(src + 8 * k) + (float)rounds, e1 = (float)(src + 8 * k + 1000) + (float)rounds;
    if (x[k].x != e0) bad++;
    if (x[k].y != e1) bad++;
  }
  if (bad) { atomicAdd(errors, bad); atomicMin(first_bad, (unsigned)(blockIdx.x * 256 + threadIdx.x)); }
}

// The operand forms hipcc actually emits in the feature kernel when it may vectorize (disassembly of MfccKernel<512, 4> built
// without -fno-slp-vectorize -fno-vectorize): negated second operands, op_sel half swaps, an SGPR pair and an inline constant as
// sources, v_pk_mov_b32.  No closed-form answer here: the kernel is deterministic, so a run beside the burner must give the bits
// a run alone gave.
// PARTIAL: the packed operations run under EXEC masks that change every round (single lanes, half rows, whole 16-lane quarters off),
// as in the feature kernel, where they sit inside divergent branches of the butterfly tasks.
template <bool PARTIAL>
__global__ __launch_bounds__(256, 2) void victim_forms(v2f *out, int rounds, float sa, float sb) {
  const int lane = threadIdx.x & 63;
  v2f s = {sa, sb};
  __asm__ volatile("v_readfirstlane_b32 s40, %0\n\tv_readfirstlane_b32 s41, %1" : : "v"(s.x), "v"(s.y) : "s40", "s41");
  unsigned x0 = 0, x1 = 0, x2 = 0, x3 = 0;          // checksums (XOR of result bits): no floating-point feedback, nothing can overflow
  for (int it = 0; it < rounds; it++) {
    // fresh, finite inputs every round
    const float f = (float)((it * 37 + lane * 11) & 1023) * 0.0078125f;
    v2f a = {1.0f + f, 2.0f - f}, b = {0.25f + f, 0.75f - 0.5f * f}, c = {0.5f - f, 1.5f + f}, d = {1.25f, 0.125f + f};
    v2f t, u, v, w, z;
    if (PARTIAL) {
      const unsigned h = (unsigned)it * 2654435761u;
      const unsigned long long quarters = ((h >> 3) & 1 ? 0xFFFFull : 0) | ((h >> 4) & 1 ? 0xFFFF0000ull : 0) | ((h >> 5) & 1 ? 0xFFFF00000000ull : 0) |
                                          ((h >> 6) & 1 ? 0xFFFF000000000000ull : 0);
      const unsigned long long mask = (h & 1) ? quarters : ((h & 2) ? 0x5555555555555555ull << ((h >> 7) & 1) : ~(1ull << ((h >> 8) & 63)));
      if (!((mask >> lane) & 1ull)) continue;
    }
    __asm__ volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    __asm__ volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(u) : "v"(c), "v"(d));
    __asm__ volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(v) : "v"(t), "v"(u));
    __asm__ volatile("v_pk_mul_f32 %0, %1, s[40:41] op_sel_hi:[1,0]" : "=v"(w) : "v"(u));
    __asm__ volatile("v_pk_mul_f32 %0, %1, s[40:41]" : "=v"(z) : "v"(t));
    __asm__ volatile("v_pk_mul_f32 %0, %1, 0.5 op_sel_hi:[1,0]" : "=v"(d) : "v"(v));
    __asm__ volatile("s_nop 0\n\tv_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=v"(t) : "v"(w));
    __asm__ volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(u) : "v"(z), "v"(d));
    __asm__ volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(v), "v"(t));
    __asm__ volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(b) : "v"(w), "v"(u));
    x0 ^= __float_as_uint(a.x) + (unsigned)it; x1 ^= __float_as_uint(a.y); x2 ^= __float_as_uint(b.x); x3 ^= __float_as_uint(b.y) + __float_as_uint(d.x) + __float_as_uint(u.y);
  }
  v2f *o = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  o[0] = v2f{__uint_as_float(x0 & 0x007fffffu), __uint_as_float(x1 & 0x007fffffu)};      // (stored as denormal bit patterns: compared bitwise on the host)
  o[1] = v2f{__uint_as_float(x2 & 0x007fffffu), __uint_as_float(x3 & 0x007fffffu)};
  o[2] = v2f{0.f, 0.f}; o[3] = v2f{0.f, 0.f};
}

__global__ __launch_bounds__(256, 2) void burner(float *out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  u32x4 ua, ub;
  for (int k = 0; k < 4; k++) { h = h * 1664525u + 1013904223u; ua[k] = (h & 0x807f807fu) | 0x3f003f00u; h = h * 1664525u + 1013904223u; ub[k] = (h & 0x807f807fu) | 0x3f003f00u; }
  const bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  float s = 0;
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  unsigned *d_err, *d_first;
  float *d_out;
  CHECK(hipMalloc(&d_err, 4)); CHECK(hipMalloc(&d_first, 4)); CHECK(hipMalloc(&d_out, 4096 * 256 * 4));
  hipStream_t sv, sb;
  CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const int rounds = 200000;      // values reach lane + 1000 + 8 k + 200000 < 2^23
  for (int lds = 0; lds < 2; lds++)
    for (int with_burner = 0; with_burner < 2; with_burner++) {
      unsigned total = 0, first = 0xFFFFFFFFu;
      for (int rep = 0; rep < 5; rep++) {
        CHECK(hipMemsetAsync(d_err, 0, 4, sv)); CHECK(hipMemsetAsync(d_first, 0xFF, 4, sv));
        CHECK(hipStreamSynchronize(sv));
        if (with_burner) for (int l = 0; l < 6; l++) hipLaunchKernelGGL(burner, dim3(512), dim3(256), 0, sb, d_out, 150000);
        if (lds) hipLaunchKernelGGL(victim<true>, dim3(512), dim3(256), 0, sv, d_err, d_first, rounds);
        else hipLaunchKernelGGL(victim<false>, dim3(512), dim3(256), 0, sv, d_err, d_first, rounds);
        unsigned e = 0, f = 0;
        CHECK(hipMemcpyAsync(&e, d_err, 4, hipMemcpyDeviceToHost, sv)); CHECK(hipMemcpyAsync(&f, d_first, 4, hipMemcpyDeviceToHost, sv));
        CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sb));
        total += e; if (f < first) first = f;
      }
      printf("victim %s, %s: %u wrong values in 5 launches x 512 workgroups x 256 lanes x 16 values%s\n", lds ? "with LDS round trips" : "registers only",
             with_burner ? "MFMA burner on every CU" : "alone", total, total ? " (first bad thread below)" : "");
      if (total) printf("  first bad thread %u\n", first);
    }
  // ---- the compiler's operand forms: bits beside the burner against bits alone
  {
    const size_t n = (size_t)512 * 256 * 4;
    v2f *d_forms;
    CHECK(hipMalloc(&d_forms, n * sizeof(v2f)));
    std::vector<v2f> alone(n), beside(n);
    for (int partial = 0; partial < 2; partial++) {
    if (partial) hipLaunchKernelGGL(victim_forms<true>, dim3(512), dim3(256), 0, sv, d_forms, 100000, 0.75f, 1.25f);
    else hipLaunchKernelGGL(victim_forms<false>, dim3(512), dim3(256), 0, sv, d_forms, 100000, 0.75f, 1.25f);
    CHECK(hipMemcpyAsync(alone.data(), d_forms, n * sizeof(v2f), hipMemcpyDeviceToHost, sv));
    CHECK(hipStreamSynchronize(sv));
    size_t wrong = 0, nonfinite = 0;
    for (int rep = 0; rep < 5; rep++) {
      for (int l = 0; l < 6; l++) hipLaunchKernelGGL(burner, dim3(512), dim3(256), 0, sb, d_out, 150000);
      if (partial) hipLaunchKernelGGL(victim_forms<true>, dim3(512), dim3(256), 0, sv, d_forms, 100000, 0.75f, 1.25f);
      else hipLaunchKernelGGL(victim_forms<false>, dim3(512), dim3(256), 0, sv, d_forms, 100000, 0.75f, 1.25f);
      CHECK(hipMemcpyAsync(beside.data(), d_forms, n * sizeof(v2f), hipMemcpyDeviceToHost, sv));
      CHECK(hipStreamSynchronize(sv)); CHECK(hipStreamSynchronize(sb));
      for (size_t i = 0; i < n; i++) {
        unsigned a[2], b[2];
        memcpy(a, &alone[i], 8); memcpy(b, &beside[i], 8);
        wrong += (a[0] != b[0]) + (a[1] != b[1]);
      }
    }
    for (size_t i = 0; i < n; i++) { const float x0 = alone[i][0], x1 = alone[i][1]; nonfinite += !(x0 == x0) + !(x1 == x1); }
    printf("victim with the compiler's operand forms (neg, op_sel, SGPR pair, inline constant, v_pk_mov_b32)%s: %zu values differ from the run alone in 5 launches beside the burner (%zu NaNs in the reference run; first value %g)\n", partial ? ", under changing partial EXEC masks" : "", wrong, nonfinite, (double)alone[0][0]);
    }
  }
  return 0;
}
