// Sustained rate of v_mfma_f32_32x32x16_bf16 with register-resident operands: zeros vs. random bit patterns
// (the matrix cores' power draw, and with it the clock, depends on the data).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void kb(float *out, int iters, unsigned seed) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  unsigned h = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) * seed;
  u32x4 ua, ub;
  for (int k = 0; k < 4; k++) {
    h = h * 1664525u + 1013904223u; ua[k] = seed ? ((h & 0x807f807fu) | 0x3f003f00u) : 0u;       // bf16 pairs in [0.5, 1) with random sign / mantissa
    h = h * 1664525u + 1013904223u; ub[k] = seed ? ((h & 0x807f807fu) | 0x3f003f00u) : 0u;
  }
  bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float *out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (unsigned seed : {0u, 12345u}) for (int blocks : {256, 512}) for (int iters : {2000, 20000, 100000}) {
    hipLaunchKernelGGL(kb<4>, dim3(blocks), dim3(256), 0, 0, out, 10, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kb<4>, dim3(blocks), dim3(256), 0, 0, out, iters, seed); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 4.0 * 4 * 2.0 * 32 * 32 * 16 * (double)blocks * iters;
    printf("bf16 32x32x16 %s blocks=%d iters=%d: %.3f ms  %.0f TFLOP/s\n", seed ? "random" : "zeros ", blocks, iters, ms, flop / (ms * 1e-3) / 1e12);
  }
  return 0;
}
