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
  return 0;
}
