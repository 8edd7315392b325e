// Host cost of submitting one scoring batch: 2 copies + 4 kernels + event, eager vs one hipGraphLaunch.
// Build & run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/micro/graph_launch.hip -o /tmp/gl && /tmp/gl
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(int* p, int n) { if (threadIdx.x == 0 && blockIdx.x == 0 && n < 0) p[0] = n; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  char *h_in, *d_in, *h_out, *d_out; int* d_p;
  hipHostMalloc((void**)&h_in, 16384); hipMalloc((void**)&d_in, 16384);
  hipHostMalloc((void**)&h_out, 8192); hipMalloc((void**)&d_out, 8192); hipMalloc((void**)&d_p, 64);
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  const int iters = 2000;
  auto eager = [&]() {
    hipMemcpyAsync(d_in, h_in, 9000, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(k, dim3(1280), dim3(256), 0, st, d_p, 1);
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, st, d_p, 2);
    hipLaunchKernelGGL(k, dim3(1600), dim3(256), 0, st, d_p, 3);
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, st, d_p, 4);
    hipMemcpyAsync(h_out, d_out, 7680, hipMemcpyDeviceToHost, st);
    hipEventRecord(ev, st);
  };
  for (int i = 0; i < 50; ++i) eager();
  hipStreamSynchronize(st);
  double t0 = now();
  for (int i = 0; i < iters; ++i) { eager(); if ((i & 3) == 3) hipEventSynchronize(ev); }
  hipStreamSynchronize(st);
  double t1 = now();
  printf("eager: %.2f us per batch (submission + completion, 4 deep)\n", (t1 - t0) / iters * 1e6);
  t0 = now();
  for (int i = 0; i < iters; ++i) eager();
  double t2 = now();
  hipStreamSynchronize(st);
  printf("eager: %.2f us per batch host submission only\n", (t2 - t0) / iters * 1e6);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  hipMemcpyAsync(d_in, h_in, 9000, hipMemcpyHostToDevice, st);
  hipLaunchKernelGGL(k, dim3(1280), dim3(256), 0, st, d_p, 1);
  hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, st, d_p, 2);
  hipLaunchKernelGGL(k, dim3(1600), dim3(256), 0, st, d_p, 3);
  hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, st, d_p, 4);
  hipMemcpyAsync(h_out, d_out, 7680, hipMemcpyDeviceToHost, st);
  hipStreamEndCapture(st, &g);
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  printf("instantiate: %s\n", hipGetErrorString(e));
  for (int i = 0; i < 50; ++i) { hipGraphLaunch(ge, st); hipEventRecord(ev, st); }
  hipStreamSynchronize(st);
  t0 = now();
  for (int i = 0; i < iters; ++i) { hipGraphLaunch(ge, st); hipEventRecord(ev, st); }
  t2 = now();
  hipStreamSynchronize(st);
  t1 = now();
  printf("graph: %.2f us per batch host submission only; %.2f us per batch incl. completion\n", (t2 - t0) / iters * 1e6, (t1 - t0) / iters * 1e6);
  return 0;
}
