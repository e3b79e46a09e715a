// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal INPUTS on gfx950, and at what rate does it issue against the bf16 form?
// (The 2 x fp16 operand split of the layer GEMMs relies on subnormal low parts: x = hi + lo, |lo| < 2^-14 for |x| < 8.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void kd(float *out, float aval, float bval) {
  f16x8 a, b;
  for (int k = 0; k < 8; k++) { a[k] = (_Float16)aval; b[k] = (_Float16)bval; }
  f32x16 acc;
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}
template <bool F16>
__global__ __launch_bounds__(256) void kb(float *out, int iters, unsigned seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  unsigned h = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) * seed;
  u32x4 ua, ub;
  for (int k = 0; k < 4; k++) {
    h = h * 1664525u + 1013904223u; ua[k] = F16 ? ((h & 0x83ff83ffu) | 0x38003800u) : ((h & 0x807f807fu) | 0x3f003f00u);
    h = h * 1664525u + 1013904223u; ub[k] = F16 ? ((h & 0x83ff83ffu) | 0x38003800u) : ((h & 0x807f807fu) | 0x3f003f00u);
  }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (F16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ua), __builtin_bit_cast(f16x8, ub), acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc[i], 0, 0, 0);
    }
  }
  float s = 0; for (int i = 0; i < 4; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float *out; hipMalloc(&out, 4096 * 256 * 4);
  const float cases[][2] = {{ldexpf(1.f, -20), 1024.f}, {ldexpf(1.f, -24), 1.f}, {ldexpf(3.f, -24), ldexpf(1.f, -24)}, {ldexpf(1.f, -15), ldexpf(1.f, -15)}, {1.f, ldexpf(5.f, -24)}};
  for (auto &c : cases) {
    hipLaunchKernelGGL(kd, dim3(1), dim3(64), 0, 0, out, c[0], c[1]);
    float v; hipMemcpy(&v, out, 4, hipMemcpyDeviceToHost);
    const double want = 16.0 * (double)c[0] * (double)c[1];
    printf("f16 mfma a=%g b=%g: got %.9g want %.9g %s\n", c[0], c[1], v, want, v == (float)want ? "OK" : "MISMATCH (subnormal flushed?)");
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int f16 = 0; f16 < 2; f16++) for (int blocks : {256, 512}) {
    const int iters = 50000;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (f16) hipLaunchKernelGGL(kb<true>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u); else hipLaunchKernelGGL(kb<false>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 4.0 * 4 * 2.0 * 32 * 32 * 16 * (double)blocks * iters;
    printf("%s 32x32x16 random blocks=%d: %.3f ms  %.0f TFLOP/s\n", f16 ? "f16 " : "bf16", blocks, ms, flop / (ms * 1e-3) / 1e12);
  }
  return 0;
}
