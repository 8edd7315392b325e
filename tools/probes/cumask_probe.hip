// Probe: does a stream created with hipExtStreamCreateWithCUMask confine a kernel to the masked CUs on this box, and
// how do mask bits map to (XCC, SE, CU)?  Build: hipcc --offload-arch=gfx950 -O2 cumask_probe.hip -o cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

__global__ void where(uint32_t* out, int spin) {
  __shared__ char big[96 * 1024];  // one workgroup per CU
  if (threadIdx.x == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    big[0] = (char)hw;
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    out[2 * blockIdx.x] = hw + (big[0] == 127 ? 1 : 0) * 0;
  }
}

static void run(hipStream_t st, int n_wg, const char* tag) {
  uint32_t* d;
  hipMalloc(&d, n_wg * 8);
  hipLaunchKernelGGL(where, dim3(n_wg), dim3(64), 0, st, d, 2000);  // 20 us spin at 100 MHz
  hipStreamSynchronize(st);
  std::vector<uint32_t> h(2 * n_wg);
  hipMemcpy(h.data(), d, n_wg * 8, hipMemcpyDeviceToHost);
  std::set<uint32_t> cus;
  int per_xcc[16] = {0};
  for (int i = 0; i < n_wg; ++i) {
    const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
    const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
  }
  for (uint32_t c : cus) per_xcc[c >> 16]++;
  printf("%s: %d workgroups ran on %zu distinct CUs; per XCC:", tag, n_wg, cus.size());
  for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
  printf("\n");
  hipFree(d);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("CUs %d\n", p.multiProcessorCount);
  hipStream_t s0;
  hipStreamCreate(&s0);
  run(s0, 256, "plain stream, 256 WGs");
  for (int keep : {240, 224, 16}) {
    uint32_t mask[8] = {0};
    for (int i = 0; i < keep; ++i) mask[i / 32] |= 1u << (i % 32);
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    printf("mask low %d bits: create -> %s\n", keep, hipGetErrorString(e));
    if (e == hipSuccess) {
      char tag[64];
      snprintf(tag, sizeof tag, "masked stream (%d CUs), %d WGs", keep, keep);
      run(s, keep, tag);
      run(s, 256, "masked stream, 256 WGs");
      hipStreamDestroy(s);
    }
  }
  // complement mask: the top 16 bits
  {
    uint32_t mask[8] = {0};
    for (int i = 240; i < 256; ++i) mask[i / 32] |= 1u << (i % 32);
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    printf("mask bits 240..255: create -> %s\n", hipGetErrorString(e));
    if (e == hipSuccess) { run(s, 16, "top-16 mask, 16 WGs"); run(s, 64, "top-16 mask, 64 WGs"); }
  }
  return 0;
}
