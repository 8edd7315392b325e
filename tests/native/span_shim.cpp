// C shim around csrc/bsr_span.h for tests/test_span_host.py (CPU only).
#include "../../mcmc-symreg_amd/csrc/bsr_span.h"

extern "C" {
// tapes: concatenated bsr_node rows, off[n+1].  The first K tapes are the chain's current trees, tape K.. are candidates;
// out[i] = 1: candidate i is in the span of the K current trees, 2: it repeats tree which_k[i] up to sign (also in span)
int span_check(const bsr_node* rows, const int* off, int K, int n_cand, const int* which_k, int* out) {
  std::vector<bsr_span::LinForm> forms(K);
  std::vector<char> ok(K);
  for (int k = 0; k < K; ++k) ok[k] = bsr_span::lin_form(rows + off[k], off[k + 1] - off[k], &forms[k]) ? 1 : 0;
  bsr_span::SpanBasis b;
  b.build(forms, ok);
  for (int i = 0; i < n_cand; ++i) {
    bsr_span::LinForm f;
    out[i] = 0;
    if (!bsr_span::lin_form(rows + off[K + i], off[K + i + 1] - off[K + i], &f)) { out[i] = -1; continue; }
    if (b.in_span(f)) out[i] = 1;
    if (ok[which_k[i]] && bsr_span::same_up_to_sign(f, forms[which_k[i]])) out[i] = 2;
  }
  return (int)b.rows.size();
}
}
