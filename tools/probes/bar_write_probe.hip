// Can the host write device memory directly (large BAR), and how long does a 16 KB block take against hipMemcpyAsync?
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/bar_write_probe.hip -o /tmp/bar_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k_sum(const unsigned* p, int n, unsigned long long* out) {
  unsigned long long s = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
  atomicAdd(out, s);
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int n = 4096;  // 16 KB
  unsigned *d = nullptr, *h = nullptr;
  unsigned long long *dout = nullptr, *hout = nullptr;
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipHostMalloc((void**)&h, n * 4);
  hipHostMalloc((void**)&hout, 8);
  hipMalloc((void**)&dout, 8);
  int attr = 0;
  hipDeviceGetAttribute(&attr, hipDeviceAttributeIsLargeBar, 0);
  printf("large BAR attribute: %d\n", attr);
  for (int flavour = 0; flavour < 2; ++flavour) {
    hipError_t e = flavour == 0 ? hipMalloc((void**)&d, n * 4)
                                : hipExtMallocWithFlags((void**)&d, n * 4, hipDeviceMallocFinegrained);
    printf("flavour %d (%s): alloc %s\n", flavour, flavour ? "fine-grained" : "hipMalloc", hipGetErrorString(e));
    if (e != hipSuccess) continue;
    hipPointerAttribute_t pa;
    memset(&pa, 0, sizeof pa);
    hipPointerGetAttributes(&pa, d);
    printf("  hostPointer %p devicePointer %p\n", pa.hostPointer, pa.devicePointer);
    fflush(stdout);
    if (!attr) continue;
    double t_direct = 0, t_copy = 0;
    unsigned long long want = 0;
    bool ok = true;
    for (int it = 0; it < 200; ++it) {
      for (int i = 0; i < n; ++i) h[i] = (unsigned)(i * 7 + it);
      want = 0;
      for (int i = 0; i < n; ++i) want += h[i];
      double t0 = now();
      memcpy(d, h, n * 4);            // host stores straight into device memory
      __sync_synchronize();
      double t1 = now();
      hipMemsetAsync(dout, 0, 8, st);
      hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, st, d, n, dout);
      hipMemcpyAsync(hout, dout, 8, hipMemcpyDeviceToHost, st);
      hipStreamSynchronize(st);
      if (*hout != want) ok = false;
      double t2 = now();
      hipMemcpyAsync(d, h, n * 4, hipMemcpyHostToDevice, st);
      double t3 = now();
      hipStreamSynchronize(st);
      if (it >= 20) { t_direct += t1 - t0; t_copy += t3 - t2; }
    }
    printf("  direct host writes %s; 16 KB: memcpy into device memory %.2f us, hipMemcpyAsync call %.2f us\n",
           ok ? "seen by the kernel" : "NOT seen", t_direct / 180, t_copy / 180);
    // stale-line check: only direct writes between launches, 512 workgroups (every XCD's L2 caches the block), the
    // result read back from pinned memory written by the kernel -- no copy command ever touches `d`
    {
      unsigned long long* hacc = nullptr;
      hipHostMalloc((void**)&hacc, 8);
      int bad = 0;
      for (int it = 0; it < 2000; ++it) {
        for (int i = 0; i < n; ++i) h[i] = (unsigned)(i * 3 + it * 11);
        unsigned long long w1 = 0;
        for (int i = 0; i < n; ++i) w1 += h[i];
        memcpy(d, h, n * 4);
        __sync_synchronize();
        *hacc = 0;
        hipLaunchKernelGGL(k_sum, dim3(512), dim3(256), 0, st, d, n, hacc);
        hipStreamSynchronize(st);
        if (*hacc != 512ull * w1) ++bad;
      }
      printf("  2000 launches with only direct writes in between: %d stale results\n", bad);
      hipHostFree(hacc);
    }
    hipFree(d);
  }
  return 0;
}
