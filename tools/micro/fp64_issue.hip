// fp64 issue rate of one SIMD as a function of waves per SIMD and independent chains per wave (gfx950):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64_issue tools/micro/fp64_issue.hip && /tmp/fp64_issue
// One workgroup of W waves on one CU (waves w, w+4, ... share a SIMD); every wave runs N dependent v_fma_f64 per chain,
// C chains interleaved.  Prints cycles per instruction per wave and per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int C>
__global__ void k(double* out, unsigned long long* cyc, int iters) {
  double v[C];
#pragma unroll
  for (int c = 0; c < C; ++c) v[c] = 1.0 + threadIdx.x * 1e-9 + c;
  const double a = 1.0000001, b = 1e-9;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int c = 0; c < C; ++c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int c = 0; c < C; ++c) s += v[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int C>
void run(int waves, int iters) {
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, 1024 * sizeof(double));
  hipMalloc(&cyc, 16 * sizeof(unsigned long long));
  k<C><<<1, waves * 64>>>(out, cyc, iters);
  k<C><<<1, waves * 64>>>(out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(16);
  hipMemcpy(h.data(), cyc, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  const double n = (double)iters * 16 * C;
  double mx = 0;
  for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
  const int per_simd = (waves + 3) / 4;
  printf("waves %2d (%d per SIMD) chains %d: %.2f cycles per instruction per wave, %.2f per SIMD\n", waves, per_simd, C,
         mx / n, mx / (n * per_simd));
  hipFree(out);
  hipFree(cyc);
}

int main() {
  const int iters = 2000;
  for (int waves : {1, 4, 8, 12, 16}) {
    run<1>(waves, iters);
    run<2>(waves, iters);
    run<4>(waves, iters);
  }
  return 0;
}
