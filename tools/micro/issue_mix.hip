// Does scalar issue ride along with vector issue on gfx950?  One workgroup of W waves on one CU; every wave runs blocks of
// NV dependent-free v_fma_f64 (4 chains) and NS s_add_u32 (4 chains) interleaved.  Prints cycles per block per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/issue_mix.bin tools/micro/issue_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NV, int NS, int NB>
__global__ void k(double* out, unsigned long long* cyc, int iters) {
  double v0 = 1.0 + threadIdx.x * 1e-9, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3;
  const double a = 1.0000001, b = 1e-9;
  int s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (NV >= 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v0) : "v"(a), "v"(b));
      if (NS >= 1) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0) : : "scc");
      if (NB >= 1) asm volatile("s_cbranch_scc0 .Lb%=\n.Lb%=:" : : : );
      if (NV >= 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v1) : "v"(a), "v"(b));
      if (NS >= 2) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s1) : : "scc");
      if (NV >= 3) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v2) : "v"(a), "v"(b));
      if (NS >= 3) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s2) : : "scc");
      if (NV >= 4) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v3) : "v"(a), "v"(b));
      if (NS >= 4) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s3) : : "scc");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + (double)(s0 + s1 + s2 + s3);
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

template <int NV, int NS, int NB>
void run(int waves, int iters) {
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, 1024 * sizeof(double));
  hipMalloc(&cyc, 16 * sizeof(unsigned long long));
  k<NV, NS, NB><<<1, waves * 64>>>(out, cyc, iters);
  k<NV, NS, NB><<<1, waves * 64>>>(out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(16);
  hipMemcpy(h.data(), cyc, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double mx = 0;
  for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
  const int per_simd = (waves + 3) / 4;
  const double groups = (double)iters * 8;   // groups of (NV vector + NS scalar + NB branch) instructions per wave
  printf("waves/SIMD %d  vector %d scalar %d branch %d per group: %.1f cycles per group per wave, %.2f per SIMD per group (vector alone would be %.1f)\n",
         per_simd, NV, NS, NB, mx / groups, mx / groups / per_simd, 4.4 * NV);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  const int iters = 1000;
  for (int waves : {4, 16}) {
    run<4, 0, 0>(waves, iters);
    run<4, 2, 0>(waves, iters);
    run<4, 4, 0>(waves, iters);
    run<2, 4, 0>(waves, iters);
    run<1, 4, 0>(waves, iters);
    run<0, 4, 0>(waves, iters);
    run<2, 2, 1>(waves, iters);
    run<1, 1, 1>(waves, iters);
  }
  return 0;
}
