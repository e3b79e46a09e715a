// v_mfma_f64_16x16x4_f64 issue rate on gfx950: NACC independent accumulators per wave, one wave per SIMD (256 threads per workgroup,
// one workgroup per CU at blocks = 256).  Prints cycles per MFMA per SIMD at the measured time and the device's FP64 matrix rate.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k64(double *out, int iters) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; i++) for (int r = 0; r < 4; r++) acc[i][r] = 0.0;
  double a = threadIdx.x * 0.001, b = blockIdx.x * 0.002;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0; for (int i = 0; i < NACC; i++) for (int r = 0; r < 4; r++) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(double *out, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k64<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k64<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)NACC * iters * ((blocks + 255) / 256);
  printf("f64 16x16x4 x%dacc blocks=%d: %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD (%.0f cycles at 2.4 GHz)\n", NACC, blocks, ms,
         2048.0 * 4 * NACC * iters * blocks / (ms * 1e-3) / 1e12, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4);
}
int main() {
  double *out; hipMalloc(&out, 4096 * 256 * 8);
  for (int blocks : {256, 512}) { run<1>(out, blocks, 20000); run<4>(out, blocks, 20000); run<8>(out, blocks, 20000); }
  return 0;
}
