// Probe: host cost of one hipGraphLaunch (memcpy + 4 dependent kernels) against the same work issued call by call.
// Build: hipcc --offload-arch=gfx950 -O2 graph_probe.hip -o graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Args { double* p; const int* hdr; int n; long long a, b, c, d, e, f, g; };
__global__ void k(Args a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[0] += (double)a.hdr[0]; }

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const int NS = 4, ITERS = 4000;
  hipStream_t st[NS]; hipEvent_t done[NS];
  char* h_in; char* d_in[NS]; double* d_p[NS];
  CK(hipHostMalloc((void**)&h_in, 64 * 1024));
  for (int i = 0; i < NS; ++i) {
    CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    CK(hipMalloc((void**)&d_in[i], 64 * 1024)); CK(hipMalloc((void**)&d_p[i], 64));
  }
  // (1) call by call
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    for (int it = 0; it < ITERS; ++it) {
      const int s = it % NS;
      if (it >= NS) CK(hipEventSynchronize(done[s]));
      CK(hipMemcpyAsync(d_in[s], h_in, 14 * 1024, hipMemcpyHostToDevice, st[s]));
      Args a{d_p[s], (const int*)d_in[s], it, 0, 0, 0, 0, 0, 0, 0};
      hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, st[s], a);
      hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, st[s], a);
      hipLaunchKernelGGL(k, dim3(104), dim3(256), 0, st[s], a);
      hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, st[s], a);
      CK(hipEventRecord(done[s], st[s]));
    }
    CK(hipDeviceSynchronize());
    printf("call by call: %.2f us per batch (memcpy + 4 kernels + event)\n", (now() - t0) / ITERS);
  }
  // (2) one graph per stream, captured once
  hipGraphExec_t ge[NS];
  for (int s = 0; s < NS; ++s) {
    hipGraph_t g;
    CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
    CK(hipMemcpyAsync(d_in[s], h_in, 14 * 1024, hipMemcpyHostToDevice, st[s]));
    Args a{d_p[s], (const int*)d_in[s], 0, 0, 0, 0, 0, 0, 0, 0};
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, st[s], a);
    hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, st[s], a);
    hipLaunchKernelGGL(k, dim3(104), dim3(256), 0, st[s], a);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, st[s], a);
    CK(hipStreamEndCapture(st[s], &g));
    CK(hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0));
  }
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    for (int it = 0; it < ITERS; ++it) {
      const int s = it % NS;
      if (it >= NS) CK(hipEventSynchronize(done[s]));
      CK(hipGraphLaunch(ge[s], st[s]));
      CK(hipEventRecord(done[s], st[s]));
    }
    CK(hipDeviceSynchronize());
    printf("graph: %.2f us per batch (one hipGraphLaunch + event)\n", (now() - t0) / ITERS);
  }
  // (3) no memcpy in the graph variant: kernels only, call by call
  {
    double t0 = now();
    for (int it = 0; it < ITERS; ++it) {
      const int s = it % NS;
      if (it >= NS) CK(hipEventSynchronize(done[s]));
      Args a{d_p[s], (const int*)d_in[s], it, 0, 0, 0, 0, 0, 0, 0};
      hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, st[s], a);
      hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, st[s], a);
      CK(hipEventRecord(done[s], st[s]));
    }
    CK(hipDeviceSynchronize());
    printf("two kernels + event, no memcpy: %.2f us per batch\n", (now() - t0) / ITERS);
  }
  {
    double t0 = now();
    for (int it = 0; it < ITERS; ++it) {
      const int s = it % NS;
      if (it >= NS) CK(hipEventSynchronize(done[s]));
      CK(hipMemcpyAsync(d_in[s], h_in, 14 * 1024, hipMemcpyHostToDevice, st[s]));
      CK(hipEventRecord(done[s], st[s]));
    }
    CK(hipDeviceSynchronize());
    printf("memcpy + event only: %.2f us per batch\n", (now() - t0) / ITERS);
  }
  return 0;
}
