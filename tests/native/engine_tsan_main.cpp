// TEST INFRASTRUCTURE.  Drives the product's native sampler (csrc/bsr_engine.hip) with worker threads and batches
// generated ahead, against the CPU stand-in of the data side (stub_scorer.cpp), for ThreadSanitizer.  Prints one
// digest line per chain: the chains' outcomes must not depend on how they were grouped over threads.
//   engine_tsan <n_chains> <props_per_chain> [trace_cap]    (groups / look-ahead / score memo through BSR_ENGINE_GROUPS /
//   BSR_ENGINE_LOOKAHEAD / BSR_ENGINE_MEMO; trace_cap > 0: the traced, single-threaded ticket path with all chains in one batch)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/bsr_hip.h"

int main(int argc, char** argv) {
  const int n_chains = argc > 1 ? atoi(argv[1]) : 8;
  const long props = argc > 2 ? atol(argv[2]) : 300;
  const long trace_cap = argc > 3 ? atol(argv[3]) : 0;
  const int N = 200, d = 3, K = 3;
  std::vector<double> X((size_t)N * d), y(N);
  uint64_t s = 12345;
  auto u = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; };
  for (int r = 0; r < N; ++r) {
    for (int f = 0; f < d; ++f) X[(size_t)r * d + f] = -3.0 + 6.0 * u();
    y[r] = 1.35 * X[(size_t)r * d] * X[(size_t)r * d + 1] + 5.5 * std::sin((X[(size_t)r * d] - 1) * (X[(size_t)r * d + 1] - 1)) + 0.1 * (u() - 0.5);
  }
  bsr_ctx* ctx = nullptr;
  if (bsr_ctx_create(&ctx, 0, N, d, X.data(), y.data(), K, n_chains, 8 * 64, BSR_DTYPE_F64) != BSR_OK) return 2;
  bsr_engine* e = nullptr;
  if (bsr_engine_create(&e, ctx, n_chains, K, N, d, -1.0, 1000000, 1) != BSR_OK) return 3;
  bsr_engine_set_nan_policy(e, 1);
  for (int c = 0; c < n_chains; ++c) {
    bsr_engine_seed(e, c, 1000u + (uint32_t)c);
    if (bsr_engine_init_chain(e, c) != BSR_OK) { fprintf(stderr, "init: %s\n", bsr_engine_last_error(e)); return 4; }
  }
  int64_t n_trace = 0;
  std::vector<bsr_trace> trace((size_t)(trace_cap > 0 ? trace_cap : 1));
  const int rc = bsr_engine_run(e, 16, props, trace_cap > 0 ? trace.data() : nullptr, trace_cap, &n_trace, 8 * 64);
  if (rc != BSR_OK) { fprintf(stderr, "run: %d %s\n", rc, bsr_engine_last_error(e)); return 5; }
  for (int c = 0; c < n_chains; ++c) {
    std::vector<bsr_node> tapes((size_t)K * 512);
    int32_t len[BSR_MAX_K];
    double beta[BSR_MAX_K + 1], errs[4096], sigma = 0;
    int32_t n_errs = 0;
    int64_t cnt[5];
    if (bsr_engine_chain_result(e, c, tapes.data(), 512, len, beta, errs, 4096, &n_errs, cnt, &sigma, 1) != BSR_OK) return 6;
    uint64_t h = 1469598103934665603ull;
    for (int k = 0; k < K; ++k)
      for (int i = 0; i < len[k]; ++i) {
        const bsr_node& n = tapes[(size_t)k * 512 + i];
        h = (h ^ (uint64_t)(uint32_t)n.opcode) * 1099511628211ull;
        h = (h ^ (uint64_t)(uint32_t)n.feature) * 1099511628211ull;
      }
    printf("chain %d props %lld accepts %lld gate %lld trees %016llx beta %.12g %.12g sigma %.12g\n", c, (long long)cnt[0],
           (long long)cnt[1], (long long)cnt[2], (unsigned long long)h, beta[0], beta[1], sigma);
  }
  bsr_engine_destroy(e);
  bsr_ctx_destroy(ctx);
  return 0;
}
