// Which CUs a stream made by hipExtStreamCreateWithCUMask runs its workgroups on (MI355X: 8 XCDs x 32 CUs): every workgroup records its
// XCC id and HW id; the host counts the distinct (XCC, SE, CU) triples per mask.  hipcc --offload-arch=gfx950 -O2 cu_mask_probe.hip -o cu_mask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void Probe(unsigned long long *out) {
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    __asm__ volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x] = ((unsigned long long)xcc << 32) | hw;
  }
  // stay a little so that workgroups spread
  long long t0 = clock64();
  while (clock64() - t0 < 20000) {}
}
int main() {
  const int n = 4096;
  unsigned long long *d;
  hipMalloc(&d, n * 8);
  std::vector<unsigned long long> h(n);
  for (int variant = 0; variant < 6; variant++) {
    std::vector<uint32_t> mask(8, 0u);
    const char *what = "";
    switch (variant) {
      case 0: for (auto &m : mask) m = 0xFFFFFFFFu; what = "all 256 bits"; break;
      case 1: for (int i = 0; i < 4; i++) mask[i] = 0xFFFFFFFFu; what = "bits 0..127"; break;
      case 2: for (auto &m : mask) m = 0x55555555u; what = "every second bit"; break;
      case 3: for (auto &m : mask) m = 0x11111111u; what = "every fourth bit"; break;
      case 4: mask[0] = 0xFFFFFFFFu; what = "bits 0..31"; break;
      case 5: for (auto &m : mask) m = 0x0000FFFFu; what = "low 16 bits of every word"; break;
    }
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", what, hipGetErrorString(e)); continue; }
    hipMemsetAsync(d, 0, n * 8, s);
    hipLaunchKernelGGL(Probe, dim3(n), dim3(256), 0, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::set<unsigned long long> cus;
    std::set<unsigned> xccs;
    for (auto v : h) {
      const unsigned hw = (unsigned)v, xcc = (unsigned)(v >> 32) & 0xF;
      const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
      cus.insert(((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu);
      xccs.insert(xcc);
    }
    printf("%-28s: %zu distinct CUs on %zu XCCs\n", what, cus.size(), xccs.size());
    hipStreamDestroy(s);
  }
  return 0;
}
