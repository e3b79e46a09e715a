// LDS atomic throughput seen by ONE wave (and by four): cycles per wave-wide instruction for ds_min_u64 / ds_min_u32 / ds_write_b64 /
// ds_read_b32 at random addresses in a 5 KB table, all 64 lanes or a third of them active.  hipcc --offload-arch=gfx950 -O3 lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) unsigned long long LdsU64;
typedef __attribute__((address_space(3))) unsigned LdsU;
template <int MODE>
__global__ void k(const unsigned *addr, long long *out, int iters, int active) {
  __shared__ unsigned long long tab[1024];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) tab[i] = ~0ull;
  __syncthreads();
  unsigned a[32];
  const unsigned base = (unsigned)(uintptr_t)tab;
  for (int j = 0; j < 32; j++) a[j] = base + 8u * (addr[j * 256 + threadIdx.x] % 640u);
  unsigned long long key = ((unsigned long long)threadIdx.x << 32) | 7u;
  unsigned acc = 0;
  const bool on = lane < active;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 32; j++) {
      if (MODE == 0) { if (on) (void)__hip_atomic_fetch_min((LdsU64 *)(uintptr_t)a[j], key + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
      if (MODE == 1) { if (on) (void)__hip_atomic_fetch_min((LdsU *)(uintptr_t)a[j], (unsigned)key + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
      if (MODE == 2) { if (on) *(LdsU64 *)(uintptr_t)a[j] = key + it; }
      if (MODE == 3) { if (on) acc += *(LdsU *)(uintptr_t)a[j]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  __asm__ volatile("s_waitcnt lgkmcnt(0)");
  long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 12345u) out[1000] = acc;
}
int main() {
  std::vector<unsigned> h(32 * 256);
  unsigned s = 12345u;
  for (auto &x : h) { s = s * 1664525u + 1013904223u; x = s >> 8; }
  unsigned *d; long long *o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 8192);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const char *names[4] = {"ds_min_u64", "ds_min_u32", "ds_write_b64", "ds_read_b32"};
  for (int waves = 1; waves <= 4; waves *= 4)
    for (int active = 64; active >= 21; active -= 43)
      for (int m = 0; m < 4; m++) {
        const int iters = 200;
        for (int rep = 0; rep < 2; rep++) {
          if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, d, o, iters, active);
          if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, d, o, iters, active);
          if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * waves), 0, 0, d, o, iters, active);
          if (m == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64 * waves), 0, 0, d, o, iters, active);
          hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, o, 8, hipMemcpyDeviceToHost);
        printf("%d wave(s), %2d lanes active, %-12s: %.1f cycles per wave instruction\n", waves, active, names[m], (double)c / (iters * 32));
      }
  return 0;
}
