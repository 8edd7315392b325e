// Native host-side sampler: the reference's proposal generator and chain loop in C++, driving the
// GPU scorer through the same C ABI entry points the Python engine uses.  Not data-parallel; it exists because once
// the O(N) work is on the GPU the Python proposal generator (~100 us/proposal) caps chain throughput.
//
// Every branch follows the reference's behaviour, quirks included, because the accepted-tree sequence for a seed is
// pinned by the ORDER and KIND of random draws:
//   grow      codes/funcs.py:74-119      fStruc   codes/funcs.py:349-398
//   Prop      codes/funcs.py:406-923     auxProp  codes/funcs.py:935-1138
//   newProp   codes/funcs.py:1184-1306   chain loop codes/bsr_class.py:99-273
// The random stream is numpy's legacy RandomState: MT19937, random_sample = (a>>5, b>>6) doubles, masked-rejection
// randint on 32-bit draws, polar-method standard_normal with a cached second value, choice via cdf search, and
// scipy's invgamma.rvs(a) = 1 / gammainccinv(a, U).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bsr_hip.h"
#include "bsr_internal.h"
#include "bsr_span.h"

#if defined(__x86_64__) || defined(__i386__)
#define BSR_CPU_RELAX() __builtin_ia32_pause()
#else
#define BSR_CPU_RELAX() std::this_thread::yield()
#endif

namespace {

// lgamma() stores the sign in the global `signgam`: a data race between the sampler's worker threads (found by the
// ThreadSanitizer build, tests/native).  The reentrant form keeps it local.
inline double lgamma_mt(double x) {
  int sign = 0;
  return lgamma_r(x, &sign);
}

const double kInf = std::numeric_limits<double>::infinity();
const double kNaN = std::numeric_limits<double>::quiet_NaN();

// ------------------------------------------------------------------------------------------------ RNG
struct LegacyRng {
  uint32_t key[624];
  int pos = 624;
  int has_gauss = 0;
  double gauss = 0.0;
  uint64_t draws = 0;   // 32-bit outputs taken so far (a position a copy of the state can be replayed to: RngMark)

  void seed(uint32_t s) {  // np.random.seed(int): mt19937_seed
    for (int i = 0; i < 624; ++i) {
      key[i] = s;
      s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
    }
    pos = 624;
    has_gauss = 0;
    gauss = 0.0;
  }
  void refill() {
    const uint32_t UP = 0x80000000u, LOW = 0x7fffffffu, MAT = 0x9908b0dfu;
    int i;
    uint32_t yv;
    for (i = 0; i < 624 - 397; ++i) {
      yv = (key[i] & UP) | (key[i + 1] & LOW);
      key[i] = key[i + 397] ^ (yv >> 1) ^ ((yv & 1u) ? MAT : 0u);
    }
    for (; i < 623; ++i) {
      yv = (key[i] & UP) | (key[i + 1] & LOW);
      key[i] = key[i + (397 - 624)] ^ (yv >> 1) ^ ((yv & 1u) ? MAT : 0u);
    }
    yv = (key[623] & UP) | (key[0] & LOW);
    key[623] = key[396] ^ (yv >> 1) ^ ((yv & 1u) ? MAT : 0u);
    pos = 0;
  }
  uint32_t next32() {
    if (pos == 624) refill();
    uint32_t yv = key[pos++];
    ++draws;
    yv ^= (yv >> 11);
    yv ^= (yv << 7) & 0x9d2c5680u;
    yv ^= (yv << 15) & 0xefc60000u;
    yv ^= (yv >> 18);
    return yv;
  }
  double uniform() {  // random_sample / uniform(0,1,1)[0]
    const uint32_t a = next32() >> 5, b = next32() >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
  int64_t randint(int64_t lo, int64_t hi) {  // np.random.randint(lo, hi): masked rejection on 32-bit draws
    const uint64_t rng = (uint64_t)(hi - 1 - lo);
    if (rng == 0) return lo;
    uint64_t mask = rng;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    mask |= mask >> 32;
    if (rng == 0xFFFFFFFFull) return lo + (int64_t)next32();
    uint32_t v;
    do {
      v = next32() & (uint32_t)mask;
    } while (v > rng);
    return lo + (int64_t)v;
  }
  double standard_normal() {  // legacy_gauss
    if (has_gauss) {
      const double t = gauss;
      has_gauss = 0;
      gauss = 0.0;
      return t;
    }
    double f, x1, x2, r2;
    do {
      x1 = 2.0 * uniform() - 1.0;
      x2 = 2.0 * uniform() - 1.0;
      r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = std::sqrt(-2.0 * std::log(r2) / r2);
    gauss = f * x1;
    has_gauss = 1;
    return f * x2;
  }
  double normal(double loc, double scale) { return loc + scale * standard_normal(); }
};

// A position in a random stream, 24 bytes: a candidate used to carry a copy of the whole state (2.5 KB, copied for every
// proposal generated); it now carries the position, and the one time in a thousand the stream has to go back there the
// state at the start of the candidate's batch is replayed up to it.
struct RngMark {
  uint64_t draws = 0;
  int has_gauss = 0;
  double gauss = 0.0;
};
inline RngMark mark_of(const LegacyRng& r) { return RngMark{r.draws, r.has_gauss, r.gauss}; }
inline void replay_to(LegacyRng& r, const LegacyRng& start, const RngMark& m) {
  r = start;
  while (r.draws < m.draws) (void)r.next32();
  r.has_gauss = m.has_gauss;
  r.gauss = m.gauss;
}

// Q(a,x) = p solved for x, a in {1,4} (scipy.special.gammainccinv; invgamma.rvs(a) = 1/gammainccinv(a, U))
double gammainccinv_int(int a, double p) {
  if (a == 1) return -std::log(p);
  if (p <= 0.0) return kInf;
  if (p >= 1.0) return 0.0;
  // a == 4: Q = e^-x (1 + x + x^2/2 + x^3/6); P = 1 - Q = e^-x sum_{n>=4} x^n/n!
  // (both take e^-x from the caller: the Newton step needs it for the density as well -- one exp per iteration, same bits)
  auto Qf = [](double x, double ex) { return ex * (1.0 + x * (1.0 + x * (0.5 + x / 6.0))); };
  auto Pf = [](double x, double ex) {
    double term = x * x * x * x / 24.0, sum = term;
    for (int n = 5; n < 200; ++n) {
      term *= x / n;
      sum += term;
      if (term < 1e-18 * sum) break;
    }
    return ex * sum;
  };
  // start from the Wilson-Hilferty approximation of the chi-square quantile (x = chi2_{2a}/2)
  double x;
  {
    const double q = p;  // upper tail probability
    // normal quantile by a coarse rational approximation, refined by Newton below anyway
    double t = std::sqrt(-2.0 * std::log(q < 0.5 ? q : 1.0 - q));
    double z = t - (2.515517 + 0.802853 * t + 0.010328 * t * t) / (1.0 + 1.432788 * t + 0.189269 * t * t + 0.001308 * t * t * t);
    if (q >= 0.5) z = -z;
    const double k = 8.0;  // degrees of freedom 2a
    double c = 1.0 - 2.0 / (9.0 * k) + z * std::sqrt(2.0 / (9.0 * k));
    x = 0.5 * k * c * c * c;
    if (!(x > 1e-6)) x = std::pow(24.0 * (1.0 - q), 0.25);
  }
  // near p = 1 solve P(x) = 1 - p instead (no cancellation in 1 - Q; below 0.9 the closed form of Q is as good as the
  // series and costs an eighth of it: the series' forty divisions were half of what this function took)
  const bool use_p = p > 0.9;
  for (int it = 0; it < 100; ++it) {
    const double ex = std::exp(-x);
    const double dens = x * x * x * ex / 6.0;  // dP/dx = -dQ/dx
    const double step = use_p ? ((1.0 - p) - Pf(x, ex)) / dens : (Qf(x, ex) - p) / dens;
    double xn = x + step;
    if (!(xn > 0.0)) xn = 0.5 * x;
    // (Newton converges quadratically here: a step below 1e-9 relative leaves an error below 1e-18 -- the step that
    // would only confirm it is not taken)
    const bool conv = std::fabs(xn - x) <= 1e-9 * xn;
    x = xn;
    if (conv) break;
  }
  return x;
}

double invgamma_rvs(LegacyRng& r, int a) { return 1.0 / gammainccinv_int(a, r.uniform()); }

inline double flog(double x) { return x > 0 ? std::log(x) : (x == 0 ? -kInf : kNaN); }
inline double fexp(double x) { return std::exp(x); }
inline double fdiv(double a, double b) {
  if (b != 0) return a / b;
  if (a != a || a == 0) return kNaN;
  return ((a > 0) == !std::signbit(b)) ? kInf : -kInf;
}
inline double invgamma_pdf(double x, int a) {
  if (!(x > 0)) return 0.0;
  return fexp(-(a + 1) * std::log(x) - lgamma_mt((double)a) - 1.0 / x);
}
const double kSqrt2Pi = std::sqrt(2 * M_PI);
inline double norm_pdf(double x, double loc, double scale) {
  const double z = (x - loc) / scale;
  return std::exp(-z * z / 2.0) / kSqrt2Pi / scale;
}
inline double pymax(double a, double b) { return b > a ? b : a; }  // Python max(a, b), NaN included

// ------------------------------------------------------------------------------------------------ trees
enum { OP_INV = 0, OP_LN = 1, OP_TERMINAL = 10 };
// Operator table (Ops / Op_weights / Op_type of codes/bsr_class.py:110-112).  A tree node keeps the operator's OPCODE
// (bsr_hip.h) in `op` and the TABLE INDEX it was created with in `op_ind` -- never updated by ReassignOperator, as in
// the reference, which matters as soon as the weights are not uniform.
#define BSR_MAX_OPS 16

struct TNode {
  int type = -1;     // -1 not grown, 0 terminal, 1 unary, 2 binary
  int op = -1;       // operator index (current), -1 none
  int op_ind = -1;   // index assigned at creation (stale after ReassignOperator, as in the reference)
  int depth = 0;
  int left = -1, right = -1, parent = -1;
  int feature = 0;
  double a = 0.0, b = 0.0;
};

// A scratch vector that keeps its capacity from call to call (one per call site and thread; never in a function that can
// be entered again while it is in use): generating a proposal ran 40 heap allocations, a third of its time.
#define BSR_SCRATCH(T, name) static thread_local std::vector<T> name; name.clear()
typedef std::pair<int, int> BSR_PAIR_II;   // (a type without a comma, for the macro)

struct Tree {
  std::vector<TNode> n;
  int root = 0;
  int add(int depth) {
    n.emplace_back();
    n.back().depth = depth;
    return (int)n.size() - 1;
  }
};

void preorder(const Tree& t, int root, std::vector<int>& out) {
  out.clear();
  BSR_SCRATCH(int, st);
  st.push_back(root);
  while (!st.empty()) {
    const int i = st.back();
    st.pop_back();
    out.push_back(i);
    if (t.n[i].left >= 0) {
      if (t.n[i].right >= 0) st.push_back(t.n[i].right);
      st.push_back(t.n[i].left);
    }
  }
}

// compact copy of the subtree reachable from root, into `d` (whose storage is reused)
void clone_into(const Tree& src, int root, Tree& d) {
  BSR_SCRATCH(int, order);
  preorder(src, root, order);
  BSR_SCRATCH(int, map);
  map.assign(src.n.size(), -1);
  d.n.clear();
  d.n.reserve(order.size() + 8);
  for (int i : order) {
    map[i] = (int)d.n.size();
    d.n.push_back(src.n[i]);
  }
  for (auto& nd : d.n) {
    nd.left = nd.left >= 0 ? map[nd.left] : -1;
    nd.right = nd.right >= 0 ? map[nd.right] : -1;
    nd.parent = (nd.parent >= 0 && map[nd.parent] >= 0) ? map[nd.parent] : -1;
  }
  d.root = map[root];
  d.n[d.root].parent = -1;
}
Tree clone_tree(const Tree& src, int root) {
  Tree d;
  clone_into(src, root, d);
  return d;
}

int count_nodes(const Tree& t, int i) {
  int c = 0;
  BSR_SCRATCH(int, st);
  st.push_back(i);
  while (!st.empty()) {
    const int j = st.back();
    st.pop_back();
    ++c;
    if (t.n[j].type == 1) st.push_back(t.n[j].left);
    else if (t.n[j].type != 0) {
      st.push_back(t.n[j].left);
      st.push_back(t.n[j].right);
    }
  }
  return c;
}
int count_ln(const Tree& t, int i) {
  int c = 0;
  BSR_SCRATCH(int, st);
  st.push_back(i);
  while (!st.empty()) {
    const int j = st.back();
    st.pop_back();
    if (t.n[j].type == 1) {
      if (t.n[j].op == OP_LN) ++c;
      st.push_back(t.n[j].left);
    } else if (t.n[j].type != 0) {
      st.push_back(t.n[j].left);
      st.push_back(t.n[j].right);
    }
  }
  return c;
}
void up_depth(Tree& t, int root) {
  t.n[root].depth = t.n[root].parent < 0 ? 0 : t.n[t.n[root].parent].depth + 1;
  BSR_SCRATCH(int, st);
  st.push_back(root);
  while (!st.empty()) {
    const int i = st.back();
    st.pop_back();
    if (t.n[i].left >= 0) {
      t.n[t.n[i].left].depth = t.n[i].depth + 1;
      st.push_back(t.n[i].left);
      if (t.n[i].right >= 0) {
        t.n[t.n[i].right].depth = t.n[i].depth + 1;
        st.push_back(t.n[i].right);
      }
    }
  }
}

struct Params {
  int n_feature;
  double beta;
  int n_ops = 0;
  int op_code[BSR_MAX_OPS];    // opcode of table entry k
  int op_type[BSR_MAX_OPS];    // arity of table entry k
  double w[BSR_MAX_OPS], cdf[BSR_MAX_OPS], logw[BSR_MAX_OPS];
  // fstruc's per-node terms for the depths trees reach, computed once by the very calls fstruc made per node (the same
  // libm results, bit for bit): the terminal's log(1 - 1 / (1 + depth)^-beta) - log(n_feature), the operator's
  // log(1 + depth) beta
  enum { DEPTH_TAB = 64 };
  double fs_term[DEPTH_TAB], fs_op[DEPTH_TAB];
  double grow_prob[DEPTH_TAB];   // grow's 1 / (1 + depth)^-beta (codes/funcs.py:79), by the call grow made per node
  void fill_depth_tables() {
    for (int dpt = 0; dpt < DEPTH_TAB; ++dpt) {
      grow_prob[dpt] = 1 / std::pow(1 + dpt, -beta);
      double ls = 0;
      ls += flog(1 - 1 / std::pow(1 + dpt, -beta));
      ls -= std::log((double)n_feature);
      fs_term[dpt] = ls;
      fs_op[dpt] = std::log((double)(1 + dpt)) * beta;
    }
  }

  static int arity_of(int code) {
    return (code == BSR_OP_ADD || code == BSR_OP_MUL || code == BSR_OP_SUB || code == BSR_OP_DIV) ? 2 : 1;
  }
  void set_table(int n, const int* codes, const double* weights) {
    n_ops = n;
    double c = 0;
    for (int i = 0; i < n; ++i) {
      op_code[i] = codes[i];
      op_type[i] = arity_of(codes[i]);
      w[i] = weights[i];
      logw[i] = flog(weights[i]);
      c += weights[i];                 // np.random.choice: cdf = cumsum(p); cdf /= cdf[-1]
      cdf[i] = c;
    }
    const double last = cdf[n - 1];
    for (int i = 0; i < n; ++i) cdf[i] /= last;
  }
  void set_default_table() {           // codes/bsr_class.py:110-112: the ten operators, uniform weights
    int codes[10];
    double weights[10];
    for (int i = 0; i < 10; ++i) {
      codes[i] = i;
      weights[i] = 1.0 / 10;
    }
    set_table(10, codes, weights);
  }
};

int choose_op(const Params& P, LegacyRng& r) {  // np.random.choice(arange(n), p=w): cdf.searchsorted(u, 'right')
  const double u = r.uniform();
  return (int)(std::upper_bound(P.cdf, P.cdf + P.n_ops, u) - P.cdf);
}

void grow(Tree& t, int i, const Params& P, double sigma_a, double sigma_b, LegacyRng& r) {  // codes/funcs.py:74-119
  const int depth = t.n[i].depth;
  bool pick = true;
  if (depth > 0) {
    const double prob = depth < Params::DEPTH_TAB ? P.grow_prob[depth] : 1 / std::pow(1 + depth, -P.beta);
    if (r.uniform() > prob) {
      t.n[i].feature = (int)r.randint(0, P.n_feature);  // :83, overwritten by the second draw at :99
      t.n[i].type = 0;
      pick = false;
    }
  }
  if (pick) {
    const int k = choose_op(P, r);
    t.n[i].op = P.op_code[k];
    t.n[i].type = P.op_type[k];
    t.n[i].op_ind = k;
  }
  if (t.n[i].type == 0) {
    t.n[i].feature = (int)r.randint(0, P.n_feature);
  } else if (t.n[i].type == 1) {
    const int l = t.add(depth + 1);
    t.n[i].left = l;
    t.n[l].parent = i;
    if (t.n[i].op == OP_LN) {
      t.n[i].a = r.normal(1, std::sqrt(sigma_a));
      t.n[i].b = r.normal(0, std::sqrt(sigma_b));
    }
    grow(t, l, P, sigma_a, sigma_b, r);
  } else {
    const int l = t.add(depth + 1);
    t.n[i].left = l;
    t.n[l].parent = i;
    const int rr = t.add(depth + 1);
    t.n[i].right = rr;
    t.n[rr].parent = i;
    grow(t, l, P, sigma_a, sigma_b, r);
    grow(t, rr, P, sigma_a, sigma_b, r);
  }
}

void fstruc(const Tree& t, int i, const Params& P, double sigma_a, double sigma_b, double* ls_out, double* lp_out) {
  // codes/funcs.py:349-398; uses each node's STORED depth and op_ind
  const TNode& nd = t.n[i];
  double ls = 0, lp = 0;
  const double logw = (nd.type != 0 && nd.op_ind >= 0) ? P.logw[nd.op_ind] : 0.0;
  if (nd.type == 0) {
    if (nd.depth >= 0 && nd.depth < Params::DEPTH_TAB) {
      ls += P.fs_term[nd.depth];   // (0 + a - b == (0 + a) - b: the table holds exactly what the two lines below add up to)
    } else {
      ls += flog(1 - 1 / std::pow(1 + nd.depth, -P.beta));
      ls -= std::log((double)P.n_feature);
    }
  } else {
    if (nd.depth == 0) ls += logw;
    else if (nd.depth > 0 && nd.depth < Params::DEPTH_TAB) ls += P.fs_op[nd.depth] + logw;
    else ls += std::log((double)(1 + nd.depth)) * P.beta + logw;
    if (nd.type == 1 && nd.op == OP_LN) {
      lp -= std::pow(nd.a - 1, 2) / (2 * sigma_a);
      lp -= std::pow(nd.b, 2) / (2 * sigma_b);
      lp -= 0.5 * std::log(2 * M_PI * sigma_a);
      lp -= 0.5 * std::log(2 * M_PI * sigma_b);
    }
  }
  if (nd.left >= 0) {
    double a, b;
    fstruc(t, nd.left, P, sigma_a, sigma_b, &a, &b);
    ls += a;
    lp += b;
    if (nd.right >= 0) {
      fstruc(t, nd.right, P, sigma_a, sigma_b, &a, &b);
      ls += a;
      lp += b;
    }
  }
  *ls_out = ls;
  *lp_out = lp;
}
double fstruc0(const Tree& t, int i, const Params& P, double sa, double sb) {
  double a, b;
  fstruc(t, i, P, sa, sb, &a, &b);
  return a;
}

void detr_candidates(const Tree& t, const std::vector<int>& order, std::vector<int>& out) {  // codes/funcs.py:454-468
  out.clear();
  for (int i : order) {
    const TNode& nd = t.n[i];
    if (nd.type == 0) continue;
    if (nd.parent < 0) {
      if (nd.right < 0) {
        if (t.n[nd.left].type == 0) continue;
      } else if (t.n[nd.left].type == 0 && t.n[nd.right].type == 0) {
        continue;
      }
    }
    out.push_back(i);
  }
}

void swap_child(Tree& t, int parent, int old_c, int new_c) {
  if (t.n[parent].left == old_c) t.n[parent].left = new_c;
  else t.n[parent].right = new_c;
  t.n[new_c].parent = parent;
}

enum { CH_NONE = 0, CH_SHRINK = 1, CH_EXPAND = 2 };
enum { A_STAY = 0, A_GROW, A_PRUNE, A_DETR, A_TRANS, A_ROP, A_RFEAT };

struct Move {
  int root = 0;
  std::vector<int> ln_nodes;
  std::vector<double> last_a, last_b;
  int change = CH_NONE;
  double Q = 1, Qinv = 1;
  int action = 0;
};

int count_terms(const Tree& t, int root, int* total) {
  BSR_SCRATCH(int, o);
  preorder(t, root, o);
  int nt = 0;
  for (int i : o) nt += t.n[i].type == 0;
  *total = (int)o.size();
  return nt;
}

// What a proposal's first lines derive from the tree it starts from (codes/funcs.py:406-480: the node lists of the tree,
// its ln nodes and their parameters, the candidates for a detransformation) -- the same for every proposal of tree k of
// a chain until that tree is replaced, and a chain makes ~1 000 proposals between two accepted ones: kept per (chain, k)
// with the compact copy of the tree the proposals start from (`base`: nodes in pre-order, so its node list is 0..n-1).
struct PropSetup {
  Tree base;
  std::vector<int> term, nterm, detcd, ln_nodes;
  std::vector<double> last_a, last_b;
  double p_stay = 0, p_grow = 0, p_prune = 0, p_detr = 0, p_trans = 0, p_rop = 0;   // the move probabilities of codes/funcs.py:475-480
};
// (by the reference's own expressions, from the three counts they depend on: seven divisions per proposal otherwise)
inline void move_probabilities(int ltNum, size_t n_nterm, size_t n_detcd, double* p_stay, double* p_grow, double* p_prune,
                               double* p_detr, double* p_trans, double* p_rop) {
  *p_stay = 0.25 * ltNum / (ltNum + 3);                                    // :475-480
  *p_grow = (1 - *p_stay) * std::min(1.0, 4.0 / ((double)n_nterm + 2)) / 3;
  *p_prune = (1 - *p_stay) / 3 - *p_grow;
  *p_detr = (1 - *p_stay) * (1.0 / 3) * (double)n_detcd / (3 + (double)n_detcd);
  *p_trans = (1 - *p_stay) / 3 - *p_detr;
  *p_rop = (1 - *p_stay) / 6;
}
void prop_lists(const Tree& t, const std::vector<int>& tree, std::vector<int>& term, std::vector<int>& nterm,
                std::vector<int>& detcd, std::vector<int>& ln_nodes, std::vector<double>& last_a, std::vector<double>& last_b) {
  ln_nodes.clear();
  last_a.clear();
  last_b.clear();
  term.clear();
  nterm.clear();
  for (int i : tree) {
    if (t.n[i].op == OP_LN && t.n[i].type == 1) {
      ln_nodes.push_back(i);
      last_a.push_back(t.n[i].a);
      last_b.push_back(t.n[i].b);
    }
    (t.n[i].type == 0 ? term : nterm).push_back(i);
  }
  detr_candidates(t, tree, detcd);
}
void build_setup(const Tree& src, PropSetup& ps) {
  clone_into(src, src.root, ps.base);
  BSR_SCRATCH(int, tree);
  preorder(ps.base, ps.base.root, tree);
  prop_lists(ps.base, tree, ps.term, ps.nterm, ps.detcd, ps.ln_nodes, ps.last_a, ps.last_b);
  move_probabilities((int)ps.ln_nodes.size(), ps.nterm.size(), ps.detcd.size(), &ps.p_stay, &ps.p_grow, &ps.p_prune, &ps.p_detr,
                     &ps.p_trans, &ps.p_rop);
}

// One structural proposal on the private copy `t` (codes/funcs.py:406-923); ps: `t` is a copy of ps->base
void prop_inplace(Tree& t, const Params& P, double sigma_a, double sigma_b, LegacyRng& r, Move& mv, const PropSetup* ps = nullptr) {
  int Root = t.root;
  BSR_SCRATCH(int, tree_own);
  BSR_SCRATCH(int, term_own);
  BSR_SCRATCH(int, nterm_own);
  BSR_SCRATCH(int, detcd_own);
  if (ps) {
    mv.ln_nodes = ps->ln_nodes;
    mv.last_a = ps->last_a;
    mv.last_b = ps->last_b;
  } else {
    preorder(t, Root, tree_own);
    prop_lists(t, tree_own, term_own, nterm_own, detcd_own, mv.ln_nodes, mv.last_a, mv.last_b);
  }
  const std::vector<int>& term = ps ? ps->term : term_own;
  const std::vector<int>& nterm = ps ? ps->nterm : nterm_own;
  const std::vector<int>& detcd = ps ? ps->detcd : detcd_own;
  const size_t n_tree = ps ? ps->base.n.size() : tree_own.size();   // nodes of the tree (ps: node i is the i-th in pre-order)
  const int ltNum = (int)mv.ln_nodes.size();
  int change = CH_NONE;
  double Q = 1, Qinv = 1;

  double p_stay, p_grow, p_prune, p_detr, p_trans, p_rop;
  if (ps) {
    p_stay = ps->p_stay; p_grow = ps->p_grow; p_prune = ps->p_prune; p_detr = ps->p_detr; p_trans = ps->p_trans; p_rop = ps->p_rop;
  } else {
    move_probabilities(ltNum, nterm.size(), detcd.size(), &p_stay, &p_grow, &p_prune, &p_detr, &p_trans, &p_rop);
  }
  const double u = r.uniform();                                                         // :483
  int action;

  if (u <= p_stay) {                                                                    // :490-500
    action = A_STAY;
    Q = Qinv = p_stay;
    const double sa = std::sqrt(sigma_a), sb = std::sqrt(sigma_b);
    for (int i : mv.ln_nodes) {
      t.n[i].a = r.normal(1, sa);
      t.n[i].b = r.normal(1, sb);
    }
  } else if (u <= p_stay + p_grow) {                                                    // :503-536
    action = A_GROW;
    const int tgt = term[r.randint(0, (int64_t)term.size())];
    grow(t, tgt, P, sigma_a, sigma_b, r);
    if (t.n[tgt].type != 0) {
      const double fs = fstruc0(t, tgt, P, sigma_a, sigma_b);
      Q = p_grow * fexp(fs) / (double)term.size();
      const int new_lt = count_ln(t, Root);
      int new_n;
      const int nt = count_terms(t, Root, &new_n);
      const double new_p = (1 - 0.25 * new_lt / (new_lt + 3)) * (1 - std::min(1.0, 4.0 / ((new_n - nt) + 2))) / 3;
      Qinv = new_p / std::max(1, new_n - nt - 1);
      if (new_lt > ltNum) change = CH_EXPAND;
    }
  } else if (u <= p_stay + p_grow + p_prune) {                                          // :539-579
    action = A_PRUNE;
    const int tgt = nterm[r.randint(1, (int64_t)nterm.size())];
    const double fs = fstruc0(t, tgt, P, sigma_a, sigma_b);
    if (count_ln(t, tgt) > 0) change = CH_SHRINK;
    t.n[tgt].left = -1;
    t.n[tgt].right = -1;
    t.n[tgt].op = -1;
    t.n[tgt].type = 0;
    t.n[tgt].feature = (int)r.randint(0, P.n_feature);
    const int new_lt = count_ln(t, Root);
    int new_n;
    const int nt = count_terms(t, Root, &new_n);
    Q = p_prune / (((double)nterm.size() - 1) * P.n_feature);
    const double pg = 1 - 0.25 * new_lt / (new_lt + 3) * 0.75 * std::min(1.0, 4.0 / ((new_n - nt) + 2));
    Qinv = pg * fexp(fs) / nt;
  } else if (u <= p_stay + p_grow + p_prune + p_detr) {                                 // :582-673
    action = A_DETR;
    const int dn = detcd[r.randint(0, (int64_t)detcd.size())];
    const int det_oi = t.n[dn].op_ind;  // the removed node's table index as assigned at its creation
    int cut = -1;
    Q = p_detr / (double)detcd.size();
    if (t.n[dn].parent < 0) {
      if (t.n[dn].right < 0) {
        Root = t.n[Root].left;
      } else if (t.n[t.n[dn].left].type == 0) {
        cut = t.n[Root].left;
        Root = t.n[Root].right;
      } else if (t.n[t.n[dn].right].type == 0) {
        cut = t.n[Root].right;
        Root = t.n[Root].left;
      } else {
        if (r.uniform() <= 0.5) {
          cut = t.n[Root].right;
          Root = t.n[Root].left;
        } else {
          cut = t.n[Root].left;
          Root = t.n[Root].right;
        }
        Q = Q / 2;
      }
    } else if (t.n[dn].type == 1) {
      swap_child(t, t.n[dn].parent, dn, t.n[dn].left);
    } else {
      if (r.uniform() <= 0.5) {
        cut = t.n[dn].right;
        swap_child(t, t.n[dn].parent, dn, t.n[dn].left);
      } else {
        cut = t.n[dn].left;
        swap_child(t, t.n[dn].parent, dn, t.n[dn].right);
      }
      Q = Q / 2;
    }
    t.n[Root].parent = -1;
    up_depth(t, Root);
    BSR_SCRATCH(int, nt_order);
    BSR_SCRATCH(int, ndet);
    preorder(t, Root, nt_order);
    int new_lt = 0;
    for (int i : nt_order) new_lt += (t.n[i].op == OP_LN && t.n[i].type == 1);
    if (new_lt < ltNum) change = CH_SHRINK;
    const double new_pstay = 0.25 * new_lt / (new_lt + 3);
    detr_candidates(t, nt_order, ndet);
    const double nd = (double)ndet.size();
    const double new_pdetr = (1 - new_pstay) * (1.0 / 3) * nd / (nd + 3);
    const double new_ptr = (1 - new_pstay) / 3 - new_pdetr;
    Qinv = new_ptr * P.w[det_oi] / (double)nt_order.size();
    if (cut >= 0) Qinv = Qinv * fexp(fstruc0(t, cut, P, sigma_a, sigma_b));  // cut keeps its stale depths
  } else if (u <= p_stay + p_grow + p_prune + p_detr + p_trans) {                       // :679-786
    action = A_TRANS;
    const int64_t ins_at = r.randint(0, (int64_t)n_tree);
    const int ins = ps ? (int)ins_at : tree_own[(size_t)ins_at];
    const int k = choose_op(P, r);
    const int nn = t.add(t.n[ins].depth);
    t.n[nn].op = P.op_code[k];
    t.n[nn].type = P.op_type[k];
    t.n[nn].op_ind = k;
    if (t.n[nn].type == 1 && t.n[nn].op == OP_LN) change = CH_EXPAND;
    const int par = t.n[ins].parent;
    if (par < 0) {
      Root = nn;
    } else {
      if (t.n[par].left == ins) t.n[par].left = nn;
      else t.n[par].right = nn;
      t.n[nn].parent = par;
    }
    t.n[nn].left = ins;
    t.n[ins].parent = nn;
    if (t.n[nn].type == 1) {
      up_depth(t, Root);
      Q = p_trans * P.w[k] / (double)n_tree;
    } else {
      const int nr = t.add(t.n[nn].depth + 1);
      t.n[nn].right = nr;
      t.n[nr].parent = nn;
      up_depth(t, Root);
      grow(t, nr, P, sigma_a, sigma_b, r);
      Q = p_trans * P.w[k] * fexp(fstruc0(t, nr, P, sigma_a, sigma_b)) / (double)n_tree;
    }
    BSR_SCRATCH(int, nt_order);
    BSR_SCRATCH(int, ndet);
    preorder(t, Root, nt_order);
    int new_lt = 0;
    for (int i : nt_order) new_lt += (t.n[i].op == OP_LN && t.n[i].type == 1);
    if (new_lt > ltNum) change = CH_EXPAND;
    const double new_pstay = 0.25 * new_lt / (new_lt + 3);
    detr_candidates(t, nt_order, ndet);
    const double nd = (double)ndet.size();
    const double new_pdetr = (1 - new_pstay) * (1.0 / 3) * nd / (nd + 3);
    Qinv = new_pdetr / nd;
    if (t.n[nn].type == 2 && t.n[t.n[nn].left].type > 0 && t.n[t.n[nn].right].type > 0) Qinv = Qinv / 2;
  } else if (u <= p_stay + p_grow + p_prune + p_detr + p_trans + p_rop) {               // :791-903
    action = A_ROP;
    const int cn = nterm[r.randint(0, (int64_t)nterm.size())];
    const int last_op = t.n[cn].op, last_type = t.n[cn].type, last_oi = t.n[cn].op_ind;
    const int k = choose_op(P, r);
    const int new_type = P.op_type[k], new_op = P.op_code[k];
    if (last_type == 1 && new_type == 1) {  // unary -> unary (op_ind not updated)
      t.n[cn].op = new_op;
      if (last_op == OP_LN) {
        if (new_op != OP_LN) change = CH_SHRINK;
      } else if (new_op == OP_LN) {
        change = CH_EXPAND;
      }
      Q = P.w[k];
      Qinv = P.w[last_oi];
    } else if (last_type == 1) {  // unary -> binary
      t.n[cn].op = new_op;
      t.n[cn].type = 2;
      const int rr = t.add(t.n[cn].depth + 1);
      t.n[cn].right = rr;
      t.n[rr].parent = cn;
      grow(t, rr, P, sigma_a, sigma_b, r);
      const double fs = fstruc0(t, rr, P, sigma_a, sigma_b);
      Q = p_rop * fexp(fs) * P.w[k] / (double)nterm.size();
      int new_n;
      const int nt = count_terms(t, Root, &new_n);
      const int new_lt = count_ln(t, Root);
      const double new_p0 = (double)new_lt / (4 * (new_lt + 3));
      Qinv = 0.125 * (1 - new_p0) * P.w[last_oi] / (new_n - nt);
      if (new_lt > ltNum) change = CH_EXPAND;
      else if (new_lt < ltNum) change = CH_SHRINK;
    } else if (new_type == 1) {  // binary -> unary
      const int cut = t.n[cn].right;
      const int p_lt = count_ln(t, cut);
      if (p_lt > 1) change = CH_SHRINK;
      else if (new_op == OP_LN && p_lt == 0) change = CH_EXPAND;
      t.n[cn].right = -1;
      t.n[cn].op = new_op;
      t.n[cn].type = new_type;
      Q = p_rop * P.w[k] / (double)nterm.size();
      const int new_n = count_nodes(t, Root);
      const int new_lt = count_ln(t, Root);
      const double new_p0 = (double)new_lt / (4 * (new_lt + 3));
      const double fs = fstruc0(t, cut, P, sigma_a, sigma_b);
      Qinv = 0.125 * (1 - new_p0) * fexp(fs) * P.w[last_oi] / new_n;  // newTerm empty at :893-894
    } else {  // binary -> binary
      t.n[cn].op = new_op;
      Q = P.w[k];
      Qinv = P.w[last_oi];
    }
  } else {                                                                              // :907-917
    action = A_RFEAT;
    const int tgt = term[r.randint(0, (int64_t)term.size())];
    t.n[tgt].feature = (int)r.randint(0, P.n_feature);
    Q = Qinv = 1;
  }
  t.n[Root].parent = -1;
  up_depth(t, Root);
  t.root = Root;
  mv.root = Root;
  mv.change = change;
  mv.Q = Q;
  mv.Qinv = Qinv;
  mv.action = action;
}

// codes/funcs.py:935-1138
void aux_inplace(Tree& t, const Move& mv, double sigma_a, double sigma_b, LegacyRng& r, double* sa2_out,
                 double* sb2_out, double* hratio, double* detjacob) {
  BSR_SCRATCH(int, order);
  BSR_SCRATCH(int, lns);
  bool any_ln = false;   // (four trees in five hold no ln node at all, attached or cut off: nothing to list then)
  for (const TNode& nd : t.n)
    if (nd.op == OP_LN && nd.type == 1) {
      any_ln = true;
      break;
    }
  if (any_ln) {
    preorder(t, t.root, order);
    for (int i : order)
      if (t.n[i].op == OP_LN && t.n[i].type == 1) lns.push_back(i);
  }
  // :945-946 draw a pair that only the shrinking move keeps (:1030-1031 and :1127-1128 draw it again): the other moves
  // take the two uniforms off the stream and skip the logarithm and the division of values nobody reads
  double new_sa2 = 0.0, new_sb2 = 0.0;
  if (mv.change == CH_SHRINK) {
    new_sa2 = invgamma_rvs(r, 1);
    new_sb2 = invgamma_rvs(r, 1);
  } else {
    (void)r.uniform();
    (void)r.uniform();
  }
  const std::vector<double>& last_a = mv.last_a;
  const std::vector<double>& last_b = mv.last_b;
  *hratio = kNaN;
  *detjacob = kNaN;
  if (mv.change == CH_SHRINK) {                                                         // :950-1026
    std::vector<double> keep_a, keep_b, cut_a, cut_b;
    for (size_t i = 0; i < mv.ln_nodes.size(); ++i) {
      const TNode& p = t.n[mv.ln_nodes[i]];
      if (p.op == OP_LN) {  // includes nodes detached with a cut subtree: nobody reset them
        keep_a.push_back(last_a[i]);
        keep_b.push_back(last_b[i]);
      } else {
        cut_a.push_back(last_a[i]);
        cut_b.push_back(last_b[i]);
      }
    }
    for (int i = 0; i < (int)lns.size() - (int)keep_a.size(); ++i) {
      keep_a.push_back(cut_a[i]);
      keep_b.push_back(cut_b[i]);
    }
    const int n0 = (int)keep_a.size();
    const double sa = std::sqrt(new_sa2), sb = std::sqrt(new_sb2);
    std::vector<double> Ua(n0), Ub(n0), Na(n0), Nb(n0), NUa, NUb;
    for (int i = 0; i < n0; ++i) {
      Ua[i] = r.normal(0, sa);
      Ub[i] = r.normal(0, sb);
    }
    for (int i = 0; i < n0; ++i) {
      Na[i] = keep_a[i] + Ua[i];
      Nb[i] = keep_b[i] + Ub[i];
      NUa.push_back(keep_a[i] - Ua[i]);
      NUb.push_back(keep_b[i] - Ub[i]);
    }
    NUa.insert(NUa.end(), last_a.begin(), last_a.end());
    NUb.insert(NUb.end(), last_b.begin(), last_b.end());
    double logh = 0, loghstar = 0;
    logh += flog(invgamma_pdf(new_sa2, 1));
    logh += flog(invgamma_pdf(new_sb2, 1));
    loghstar += flog(invgamma_pdf(sigma_a, 1));
    loghstar += flog(invgamma_pdf(sigma_b, 1));
    for (int i = 0; i < n0; ++i) {
      logh += flog(norm_pdf(Ua[i], 0, sa));
      logh += flog(norm_pdf(Ub[i], 0, sb));
    }
    const double osa = std::sqrt(sigma_a), osb = std::sqrt(sigma_b);
    for (size_t i = 0; i < NUa.size(); ++i) {
      loghstar += flog(norm_pdf(NUa[i], 0, osa));
      loghstar += flog(norm_pdf(NUb[i], 0, osb));
    }
    *hratio = fexp(loghstar - logh);
    *detjacob = std::ldexp(1.0, 2 * n0);
    for (size_t i = 0; i < lns.size(); ++i) {
      t.n[lns[i]].a = Na[i];
      t.n[lns[i]].b = Nb[i];
    }
  } else if (mv.change == CH_EXPAND) {                                                  // :1030-1110
    new_sa2 = invgamma_rvs(r, 1);
    new_sb2 = invgamma_rvs(r, 1);
    const int m = (int)last_a.size();
    const double sa = std::sqrt(new_sa2), sb = std::sqrt(new_sb2);
    std::vector<double> Ua(m), Ub(m), Na, Nb, NUa(m), NUb(m);
    for (int i = 0; i < m; ++i) {
      Ua[i] = r.normal(0, sa);
      Ub[i] = r.normal(0, sb);
    }
    for (int i = 0; i < m; ++i) {
      Na.push_back((last_a[i] + Ua[i]) / 2);
      Nb.push_back((last_b[i] + Ub[i]) / 2);
      NUa[i] = (last_a[i] - Ua[i]) / 2;
      NUb[i] = (last_b[i] - Ub[i]) / 2;
    }
    const int nn = (int)lns.size() - m;
    for (int i = 0; i < nn; ++i) {
      Na.push_back(r.normal(1, sa));
      Nb.push_back(r.normal(0, sb));
    }
    double logh = 0, loghstar = 0;
    logh += flog(invgamma_pdf(new_sa2, 1));
    logh += flog(invgamma_pdf(new_sb2, 1));
    loghstar += flog(invgamma_pdf(sigma_a, 1));
    loghstar += flog(invgamma_pdf(sigma_b, 1));
    for (int i = m; i < nn; ++i) {  // :1084-1086 adds plain pdf values
      logh += norm_pdf(Na[i], 1, sa);
      logh += norm_pdf(Nb[i], 0, sb);
    }
    for (int i = 0; i < m; ++i) {
      logh += flog(norm_pdf(Ua[i], 0, sa));
      logh += flog(norm_pdf(Ub[i], 0, sb));
    }
    const double osa = std::sqrt(sigma_a), osb = std::sqrt(sigma_b);
    for (int i = 0; i < m; ++i) {
      loghstar += flog(norm_pdf(NUa[i], 0, osa));
      loghstar += flog(norm_pdf(NUb[i], 0, osb));
    }
    *hratio = fexp(loghstar - logh);
    *detjacob = 1.0 / std::ldexp(1.0, 2 * m);
    for (size_t i = 0; i < lns.size(); ++i) {
      t.n[lns[i]].a = Na[i];
      t.n[lns[i]].b = Nb[i];
    }
  } else {                                                                              // :1127-1136
    new_sa2 = invgamma_rvs(r, 1);
    new_sb2 = invgamma_rvs(r, 1);
    if (!lns.empty()) {   // (four trees in five: no ln node, no square roots either)
      const double sa = std::sqrt(new_sa2), sb = std::sqrt(new_sb2);
      BSR_SCRATCH(double, va);
      BSR_SCRATCH(double, vb);
      va.resize(lns.size());
      vb.resize(lns.size());
      for (size_t i = 0; i < lns.size(); ++i) {
        va[i] = r.normal(1, sa);
        vb[i] = r.normal(0, sb);
      }
      for (size_t i = 0; i < lns.size(); ++i) {
        t.n[lns[i]].a = va[i];
        t.n[lns[i]].b = vb[i];
      }
    }
  }
  *sa2_out = new_sa2;
  *sb2_out = new_sb2;
}

// Postfix rows of the tree (heavier child of +/* first: Sethi-Ullman), as bsr/tape.py:flatten
void flatten(const Tree& t, int root, std::vector<bsr_node>& rows) {
  rows.clear();
  rows.reserve(t.n.size());
  BSR_SCRATCH(int, need);
  need.assign(t.n.size(), 0);
  {
    BSR_SCRATCH(BSR_PAIR_II, st);
    st.push_back({root, 0});
    while (!st.empty()) {
      auto [i, seen] = st.back();
      st.pop_back();
      const TNode& nd = t.n[i];
      if (nd.type == 0) {
        need[i] = 1;
        continue;
      }
      if (!seen) {
        st.push_back({i, 1});
        st.push_back({nd.left, 0});
        if (nd.type == 2) st.push_back({nd.right, 0});
      } else if (nd.type == 1) {
        need[i] = need[nd.left];
      } else {
        const int a = need[nd.left], b = need[nd.right];
        const bool comm = nd.op == BSR_OP_ADD || nd.op == BSR_OP_MUL;
        need[i] = comm ? ((a == b) ? a + 1 : std::max(a, b)) : std::max(a, b + 1);  // sub/div: left waits on the stack
      }
    }
  }
  BSR_SCRATCH(int, index);
  index.assign(t.n.size(), -1);
  BSR_SCRATCH(BSR_PAIR_II, st);
  st.push_back({root, 0});
  while (!st.empty()) {
    auto [i, seen] = st.back();
    st.pop_back();
    const TNode& nd = t.n[i];
    bsr_node r;
    memset(&r, 0, sizeof r);
    if (nd.type == 0) {
      index[i] = (int)rows.size();
      r.opcode = BSR_OP_TERMINAL;
      r.left = r.right = -1;
      r.feature = nd.feature;
      rows.push_back(r);
    } else if (!seen) {
      st.push_back({i, 1});
      if (nd.type == 1) {
        st.push_back({nd.left, 0});
      } else {
        int first = nd.left, second = nd.right;
        const bool comm = nd.op == BSR_OP_ADD || nd.op == BSR_OP_MUL;
        if (comm && need[nd.right] > need[nd.left]) std::swap(first, second);
        st.push_back({second, 0});
        st.push_back({first, 0});
      }
    } else {
      index[i] = (int)rows.size();
      r.opcode = nd.op;
      r.left = index[nd.left];
      r.right = nd.type == 2 ? index[nd.right] : -1;
      r.feature = -1;
      if (nd.op == OP_LN) {
        r.a = nd.a;
        r.b = nd.b;
      }
      rows.push_back(r);
    }
  }
}

uint64_t tree_hash(const Tree& t, int root) {  // FNV-1a over the pre-order (type, operator, feature) sequence
  BSR_SCRATCH(int, o);
  preorder(t, root, o);
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](uint64_t v) {
    for (int k = 0; k < 4; ++k) {
      h ^= (v >> (8 * k)) & 0xFF;
      h *= 1099511628211ull;
    }
  };
  for (int i : o) {
    mix((uint64_t)(t.n[i].type + 1));
    mix((uint64_t)(t.n[i].type == 0 ? 100 + t.n[i].feature : t.n[i].op));
  }
  return h;
}

// ... and over everything that decides the column: the ln nodes' parameters too (two trees with the same key compute
// the same column, bit for bit)
// *check (optional): a second, independently mixed word over the same stream of node words (rotate-xor-add; two cheap
// operations per word) -- what the score memo compares next to the 64-bit key, so that a key collision between two
// different trees is a miss and not a wrong score.
uint64_t tree_hash_full(const Tree& t, int root, uint32_t* check = nullptr) {
  BSR_SCRATCH(int, o);
  preorder(t, root, o);
  uint64_t h = 1469598103934665603ull;
  uint64_t h2 = 0x243F6A8885A308D3ull;
  auto mix = [&](uint64_t v) {   // (a word at a time: multiply-xorshift, two multiplications instead of FNV's eight)
    h = (h ^ v) * 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 29;
    h2 = ((h2 << 7) | (h2 >> 57)) + (v ^ 0x9E3779B97F4A7C15ull) * 0x100000001B3ull;
  };
  for (int i : o) {
    const TNode& nd = t.n[i];
    mix(((uint64_t)(nd.type + 1) << 32) | (uint64_t)(uint32_t)(nd.type == 0 ? 100 + nd.feature : nd.op));
    if (nd.type == 1 && nd.op == OP_LN) {
      uint64_t a, b;
      memcpy(&a, &nd.a, 8);
      memcpy(&b, &nd.b, 8);
      mix(a);
      mix(b);
    }
  }
  if (check) *check = (uint32_t)(h2 ^ (h2 >> 32)) + (uint32_t)o.size();
  return h;
}

// Canonical key of the column a tree computes, as far as structure tells: equal keys = equal columns up to sign.
// Children of + and * are combined order-free; a negation at the root is dropped (a column and its negative are
// collinear).  Used to guess the rank gate: a candidate that repeats a sibling, or siblings that repeat each other.
uint64_t canon_key_at(const Tree& t, int i) {
  const TNode& nd = t.n[i];
  auto mix = [](uint64_t h, uint64_t v) {
    h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    return h * 0xff51afd7ed558ccdull;
  };
  if (nd.type == 0) return mix(0x1234567ull, (uint64_t)nd.feature + 1000);
  uint64_t h = mix(0x77ull, (uint64_t)nd.op + 1);
  if (nd.op == OP_LN) {
    uint64_t ba, bb;
    memcpy(&ba, &nd.a, 8);
    memcpy(&bb, &nd.b, 8);
    h = mix(mix(h, ba), bb);
  }
  const uint64_t l = canon_key_at(t, nd.left);
  if (nd.type == 1) return mix(h, l);
  const uint64_t r = canon_key_at(t, nd.right);
  if (nd.op == BSR_OP_ADD || nd.op == BSR_OP_MUL) return mix(h, (l < r ? mix(l, r) : mix(r, l)));
  return mix(mix(h, l), r);
}
uint64_t canon_key(const Tree& t) {
  int i = t.root;
  while (t.n[i].type == 1 && t.n[i].op == BSR_OP_NEG) i = t.n[i].left;
  return canon_key_at(t, i);
}

struct Cand {
  int k;
  Tree tree;
  int change;
  double Q, Qinv, hratio, detjacob, new_sigma, new_sa2, new_sb2, u;
  int action;
  bool pred_def = false;  // speculated as a rank-gate rejection: no accept-uniform was drawn behind it
  uint64_t ghash = 0;     // tree_hash_full of the candidate, mixed with k (the gate's memory below)
  uint32_t gcheck = 0;    // the hash's second word (ScoreMemo compares both)
  double sn_s = 0, sn_p = 0;  // fStruc of the proposed tree (structure / ln-parameter parts)
  double terms[8];            // what the device-side MH step needs (include/bsr_hip.h: bsr_score_submit_mh)
  int mhflags = 0;
  RngMark before_u;   // where the stream stood in front of the candidate's accept-uniform (replayed from ChainS::start_state)
  std::vector<bsr_node> tape;
};

// Scores of the candidates a chain has had scored in its CURRENT state (emptied by every accepted move).  A score is a
// function of (candidate column, slot k, the chain's K current columns) and -- through the log-likelihood only -- of the
// candidate's sigma (codes/funcs.py:1162-1173: SSE does not see it), and a chain proposes the same few mutations of its
// trees over and over between two accepts: 40-49 % of the consumed proposals repeat a (tree incl. ln parameters, k)
// already scored (tools/memo_probe.py, profiles/r05_memo_probe.txt).  Those are answered from here: rank and SSE from
// the table, the log-likelihood recomputed for the candidate's own sigma by the formula the device uses.
struct ScoreMemo {
  static constexpr int CAP = 2048;   // slots (open addressing, linear probing); full at CAP / 2 entries: no more inserts
  std::vector<uint64_t> key;         // 0: empty
  std::vector<uint32_t> chk;         // second hash word of the tree behind the key: a key that matches with another check
                                     // word is a collision -- a miss (the candidate goes to the GPU), never a wrong score
  std::vector<bsr_score> val;
  int n = 0;
  int64_t hits = 0, lookups = 0, collisions = 0;
  void clear() {
    if (n > 0) std::fill(key.begin(), key.end(), 0ull);
    n = 0;
  }
  const bsr_score* peek(uint64_t k, uint32_t ck) const {   // (find without the statistics: the generator's look)
    if (n == 0) return nullptr;
    if (k == 0) k = 1;
    for (size_t i = (size_t)(k * 0x9E3779B97F4A7C15ull >> 53) & (CAP - 1);; i = (i + 1) & (CAP - 1)) {
      if (key[i] == k) return chk[i] == ck ? &val[i] : nullptr;
      if (key[i] == 0) return nullptr;
    }
  }
  const bsr_score* find(uint64_t k, uint32_t ck) {
    ++lookups;
    if (n == 0) return nullptr;
    if (k == 0) k = 1;
    for (size_t i = (size_t)(k * 0x9E3779B97F4A7C15ull >> 53) & (CAP - 1);; i = (i + 1) & (CAP - 1)) {
      if (key[i] == k) {
        if (chk[i] != ck) { ++collisions; return nullptr; }
        ++hits;
        return &val[i];
      }
      if (key[i] == 0) return nullptr;
    }
  }
  void put(uint64_t k, uint32_t ck, const bsr_score& v) {
    if (key.empty()) { key.assign(CAP, 0ull); chk.assign(CAP, 0u); val.resize(CAP); }
    if (n >= CAP / 2) return;
    if (k == 0) k = 1;
    for (size_t i = (size_t)(k * 0x9E3779B97F4A7C15ull >> 53) & (CAP - 1);; i = (i + 1) & (CAP - 1)) {
      if (key[i] == k) return;
      if (key[i] == 0) { key[i] = k; chk[i] = ck; val[i] = v; ++n; return; }
    }
  }
};

struct ChainS {
  int index = 0;
  ScoreMemo memo;
  LegacyRng rng;
  std::vector<Tree> roots;
  std::vector<std::vector<bsr_node>> tapes;
  // structure of the current trees (bsr_span.h): per tree k the echelon basis of the OTHER trees' linear forms, and
  // whether those siblings are dependent among themselves -- rank-gate speculation (predict_gate_reject)
  std::vector<bsr_span::SpanBasis> sib_basis;
  std::vector<char> sib_dependent;
  std::vector<Tree> last_roots;  // `Roots` as built before the latest newProp (codes/bsr_class.py:180-182)
  std::vector<double> siga, sigb, Beta, errs;
  std::vector<double> fs_old_s, fs_old_p;
  std::vector<char> fs_old_ok;
  double yll_sse = kNaN, yll_sigma = kNaN, yll_val = 0.0, lp_sigma_val = 0.0;   // consume(): the state's own terms of the log-ratio
  std::vector<PropSetup> setup;     // per tree k: what its proposals start from (valid while setup_ok[k])
  std::vector<char> setup_ok;
  // storage of the candidates that have been consumed or thrown away, for the ones generated next (a candidate's tree
  // and tape were two allocations per proposal)
  std::vector<std::vector<TNode>> pool_nodes;
  std::vector<std::vector<bsr_node>> pool_tape;
  double sigma = 1.0, sse_old = 0.0;
  int total = 0, count = 0;
  bool done = false, inited = false, last_stale = false;
  int64_t n_props = 0, n_accept = 0, n_rank_rej = 0, n_discard = 0;
  double run_ema = 1e9;  // typical number of proposals consumed per batch (speculation length that pays off)
  int verify_expect_event = -1;  // BSR_ENGINE_VERIFY_MH: what the device said about the proposal being consumed
  // Rank-gate predictor.  The reference draws no accept-uniform behind a proposal its rank gate rejects
  // (codes/funcs.py:1226-1228), so a speculative batch stays on the right random stream only if the gate's verdicts
  // are guessed.  Two cheap guesses, both exact-by-construction (a wrong one costs the tail of the batch):
  //  * magnitude: an interval bound of max|z| over the features' ranges against the siblings' max|.| -- a column
  //    1/(N eps) times larger or smaller than its siblings is what most rejections are (exp(x^3)^3 next to O(1) columns);
  //  * structure: a candidate that computes the same column as a sibling (up to sign), or siblings that repeat each
  //    other (then every candidate for the remaining trees is rejected until one of the two changes): canon_key;
  //  * history: def_ema[k], the share of tree k's recent proposals that were rejected, for dependent-sibling states
  //    the structure does not show.
  std::vector<double> def_ema;
  std::vector<uint64_t> ckey;      // canon_key of the chain's current trees
  std::vector<double> colmax;      // max|.| of the chain's current columns (from the last refresh)
  std::vector<uint32_t> colflags;
  std::vector<Cand> cands;
  // Candidates the rank gate has rejected in the chain's CURRENT state (emptied by every accepted move).  The gate's
  // verdict is a function of the candidate and its siblings alone, and a chain proposes the same few mutations of its
  // trees over and over (`inv(cub(x5))` for a current `cub(x5) * x0`: a column of range 1e16 next to siblings of range 1,
  // rejected on scale, which the interval estimate below underrates): 93 % of the events that end a speculative run were
  // rejections nobody predicted, most of them repeats.
  std::vector<uint64_t> gate_memo;
  std::vector<uint64_t> gate_pass_memo;   // ... and the ones predicted rejected that passed (the estimates' false alarms)
  int64_t n_evt_gate = 0, n_evt_pass = 0, n_pred_ok = 0;   // events by kind (BSR_ENGINE_PROF): a rejection nobody predicted, a predicted one that passed; predicted and right
  LegacyRng end_state;
  LegacyRng gen_start;     // the stream in front of the batch generate() built last (travels with the batch: Lane::start_state)
  LegacyRng start_state;   // the stream in front of the batch being consumed (what the candidates' marks are replayed from)
};

// the end of a list of candidates: their storage goes to the chain's pools
void recycle(ChainS& c, std::vector<Cand>& v) {
  for (Cand& cd : v) {
    if (cd.tree.n.capacity() > 0 && c.pool_nodes.size() < 256) {
      c.pool_nodes.emplace_back();
      c.pool_nodes.back().swap(cd.tree.n);
    }
    if (cd.tape.capacity() > 0 && c.pool_tape.size() < 256) {
      c.pool_tape.emplace_back();
      c.pool_tape.back().swap(cd.tape);
    }
  }
  v.clear();
}

}  // namespace

struct bsr_engine {
  bsr_ctx* ctx = nullptr;
  int K = 0, n_chains = 0, val = 100, y_is_series = 1;
  int nan_reject = 0;  // 0: a NaN candidate aborts like the reference (LinAlgError); 1: treat it as a rank-gate rejection
  int predict_gate = 1;  // speculate the rank gate's verdict (BSR_ENGINE_PREDICT=0: off)
  int score_memo = 1;    // answer exact repeats of a candidate in an unchanged chain state from a table (BSR_ENGINE_MEMO=0: off)
  int device_mh = 0;     // BSR_ENGINE_DEVICE_MH=1: log-ratio, accept test and first-event scan on the device (k_events).
                         // Off by default: measured 8 % slower end to end (one more launch per batch, 4 KB more
                         // upload) than the 0.2 us per proposal the host spends on the same arithmetic.
  int verify_mh = 0;     // BSR_ENGINE_VERIFY_MH=1: the host recomputes every decision and compares
  const double* x_lo = nullptr;  // per-feature range of X, owned by the context
  const double* x_hi = nullptr;
  int64_t N = 0;
  Params P;
  std::vector<ChainS> chains;
  std::string err;
  bsr_trace* trace = nullptr;
  int64_t trace_cap = 0, n_trace = 0;
  double t_gen = 0, t_submit = 0, t_wait = 0, t_consume = 0;  // seconds, reported when BSR_ENGINE_PROF is set
  std::mutex mu;        // guards err (worker threads)
};

namespace {

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int efail(bsr_engine* e, int code, const std::string& msg) {
  std::lock_guard<std::mutex> lk(e->mu);
  e->err = msg;
  return code;
}
#define ECHK(e, call)                                                         \
  do {                                                                        \
    int rc_ = (call);                                                         \
    if (rc_ != BSR_OK) return efail(e, rc_, std::string(#call) + ": " + bsr_last_error((e)->ctx)); \
  } while (0)

int refresh_chain(bsr_engine* e, ChainS& c) {
  bsr_chain_info info;
  ECHK(e, bsr_refresh(e->ctx, c.index, &info));
  bool any = false;
  c.colmax.assign(e->K, 0.0);
  c.colflags.assign(e->K, 0u);
  for (int k = 0; k < e->K; ++k) {
    any |= info.colflags[k] != 0;
    c.colmax[k] = info.maxabs[k];
    c.colflags[k] = info.colflags[k];
  }
  // every fitted value of a non-finite old state is NaN: Series.sum(skipna=True) gives 0.0, ndarray sum NaN
  c.sse_old = any ? (e->y_is_series ? 0.0 : kNaN) : info.sse_old;
  std::fill(c.fs_old_ok.begin(), c.fs_old_ok.end(), 0);
  std::fill(c.setup_ok.begin(), c.setup_ok.end(), 0);
  return BSR_OK;
}

void rebuild_sibling_spans(ChainS& c, int K);
int init_chain(bsr_engine* e, ChainS& c) {  // codes/bsr_class.py:116-163
  const int K = e->K;
  c.sigma = invgamma_rvs(c.rng, 1);
  c.roots.assign(K, Tree());
  c.siga.assign(K, 0.0);
  c.sigb.assign(K, 0.0);
  c.tapes.assign(K, {});
  for (int k = 0; k < K; ++k) {
    Tree t;
    t.add(0);
    t.root = 0;
    const double sa = invgamma_rvs(c.rng, 1);
    const double sb = invgamma_rvs(c.rng, 1);
    grow(t, 0, e->P, sa, sb, c.rng);
    c.roots[k] = t;
    c.siga[k] = sa;
    c.sigb[k] = sb;
  }
  for (int k = 0; k < K; ++k) {
    flatten(c.roots[k], c.roots[k].root, c.tapes[k]);
    ECHK(e, bsr_set_current(e->ctx, c.index, k, c.tapes[k].data(), (int)c.tapes[k].size()));
  }
  c.fs_old_s.assign(K, 0.0);
  c.fs_old_p.assign(K, 0.0);
  c.fs_old_ok.assign(K, 0);
  c.setup.resize(K);
  c.setup_ok.assign(K, 0);
  c.def_ema.assign(K, 0.0);
  c.gate_memo.clear();        // (a chain object is used again for the next restart: nothing of the last one's state stands)
  c.gate_pass_memo.clear();
  c.memo.clear();
  c.ckey.assign(K, 0);
  for (int k = 0; k < K; ++k) c.ckey[k] = canon_key(c.roots[k]);
  rebuild_sibling_spans(c, K);
  int rc = refresh_chain(e, c);
  if (rc != BSR_OK) return rc;
  c.Beta.assign(K + 1, 0.0);
  double rmse;
  ECHK(e, bsr_fit_beta(e->ctx, c.index, c.Beta.data(), &rmse));
  c.total = 0;
  c.count = 0;
  c.errs.clear();
  c.last_roots = c.roots;
  c.last_stale = false;
  c.done = false;
  c.inited = true;
  c.n_props = c.n_accept = c.n_rank_rej = c.n_discard = 0;
  return BSR_OK;
}

// Interval bound of a tree's values over the per-feature ranges of X (what max|z| can be; loose for non-monotone
// compositions, which only makes the predictor miss).  1/x and log|x| across zero are bounded by the spacing of N
// values spread over the interval.
struct Iv {
  double lo, hi;
  double mn;   // an estimate of the smallest |value| among the N rows (what 1/x blows up on); < 0: not tracked
};
// The range of a tree's values over the data, by interval arithmetic from the features' ranges; and, next to it, how close
// to zero the values are likely to come: a feature that straddles zero comes within range / 2N of it, its cube within the
// cube of that, and the interval alone (the cube's range / 2N) is ten orders of magnitude off -- `inv(cub(x5))`, a column
// of range 1e16, used to be predicted harmless.
Iv tree_range(const Tree& t, int i, const double* xlo, const double* xhi, double N) {
  const TNode& nd = t.n[i];
  auto straddles = [](double lo, double hi) { return lo <= 0 && 0 <= hi; };
  auto fresh = [&](double lo, double hi) {   // nothing known but the interval
    return Iv{lo, hi, straddles(lo, hi) ? (hi - lo) / (2 * N) : std::min(std::fabs(lo), std::fabs(hi))};
  };
  if (nd.type == 0) return fresh(xlo[nd.feature], xhi[nd.feature]);
  const Iv a = tree_range(t, nd.left, xlo, xhi, N);
  auto mag = [](const Iv& v) { return std::max(std::fabs(v.lo), std::fabs(v.hi)); };
  auto mul = [&](const Iv& p, const Iv& q) {
    double c[4] = {p.lo * q.lo, p.lo * q.hi, p.hi * q.lo, p.hi * q.hi};
    double lo = kInf, hi = -kInf;
    for (double v : c) {
      if (v != v) v = 0.0;  // 0 * inf
      lo = std::min(lo, v);
      hi = std::max(hi, v);
    }
    return Iv{lo, hi, p.mn * q.mn};   // (the smallest values of both rarely share a row: an underestimate, the safe side)
  };
  auto inv = [&](const Iv& v) {
    if (straddles(v.lo, v.hi)) {
      const double m = 1 / std::max(v.mn, 1e-300);
      return Iv{-m, m, 1 / std::max(mag(v), 1e-300)};
    }
    return Iv{std::min(1 / v.lo, 1 / v.hi), std::max(1 / v.lo, 1 / v.hi), 1 / std::max(mag(v), 1e-300)};
  };
  if (nd.type == 1) {
    switch (nd.op) {
      case BSR_OP_LN: {
        const double p = nd.a * a.lo + nd.b, q = nd.a * a.hi + nd.b;
        return fresh(std::min(p, q), std::max(p, q));   // (the zero crossing moves with b)
      }
      case BSR_OP_NEG: return {-a.hi, -a.lo, a.mn};
      case BSR_OP_SIN:
      case BSR_OP_COS: return fresh(-1.0, 1.0);
      case BSR_OP_EXP: {
        auto ex = [](double v) { return v > 200 ? 1e10 : std::exp(v); };
        return fresh(ex(a.lo), std::max(ex(a.hi), ex(std::min(a.hi, 200.0))));
      }
      case BSR_OP_SQUARE: {
        const double m = std::max(a.lo * a.lo, a.hi * a.hi);
        return {straddles(a.lo, a.hi) ? 0.0 : std::min(a.lo * a.lo, a.hi * a.hi), m, a.mn * a.mn};
      }
      case BSR_OP_CUBIC: return {a.lo * a.lo * a.lo, a.hi * a.hi * a.hi, a.mn * a.mn * a.mn};
      case BSR_OP_INV: return inv(a);
      case BSR_OP_LOG: {
        const double top = std::log(std::max(mag(a), 1e-300));
        const double bot = std::log(std::max(a.mn, 1e-300));
        return fresh(std::min(bot, top), top);
      }
      default: return a;
    }
  }
  const Iv b = tree_range(t, nd.right, xlo, xhi, N);
  switch (nd.op) {
    case BSR_OP_ADD: return fresh(a.lo + b.lo, a.hi + b.hi);
    case BSR_OP_SUB: return fresh(a.lo - b.hi, a.hi - b.lo);
    case BSR_OP_DIV: return mul(a, inv(b));
    default: return mul(a, b);
  }
}

// the sibling bases of a chain (after initialisation and after every accepted move)
void rebuild_sibling_spans(ChainS& c, int K) {
  std::vector<bsr_span::LinForm> forms((size_t)K);
  std::vector<char> ok((size_t)K, 0);
  for (int j = 0; j < K; ++j)
    ok[j] = (j < (int)c.tapes.size() && !c.tapes[j].empty() &&
             bsr_span::lin_form(c.tapes[j].data(), (int)c.tapes[j].size(), &forms[j])) ? 1 : 0;
  c.sib_basis.assign((size_t)K, bsr_span::SpanBasis());
  c.sib_dependent.assign((size_t)K, 0);
  for (int k = 0; k < K; ++k) {
    std::vector<char> okk = ok;
    okk[k] = 0;
    c.sib_basis[k].build(forms, okk);
    int known = 0;
    for (int j = 0; j < K; ++j) known += (okk[j] && !forms[j].inexact) ? 1 : 0;   // (the basis takes only forms that stand for their columns)
    c.sib_dependent[k] = (int)c.sib_basis[k].rows.size() < known ? 1 : 0;
  }
}

bool predict_gate_reject(bsr_engine* e, const ChainS& c, const Tree& t, int k, const std::vector<bsr_node>& tape) {
  if (!e->predict_gate) return false;
  if (c.def_ema[k] > 0.9) return true;   // a state in which (nearly) every candidate for k is rejected
  if (e->K < 2) return false;
  const uint64_t key = canon_key(t);
  for (int i = 0; i < e->K; ++i) {
    if (i == k) continue;
    if (c.ckey[i] == key) return true;                      // repeats a sibling
    for (int j = i + 1; j < e->K; ++j)
      if (j != k && c.ckey[i] == c.ckey[j]) return true;    // two siblings repeat each other
  }
  // linear structure: the siblings are dependent among themselves, or the candidate is a linear combination of them
  // (`x1 + x6` next to `x1 + x1` and `-x6`): the new matrix has rank < K whatever the numbers are
  if ((size_t)k < c.sib_basis.size()) {
    if (c.sib_dependent[k]) return true;
    bsr_span::LinForm f;
    if (!tape.empty() && bsr_span::lin_form(tape.data(), (int)tape.size(), &f) && c.sib_basis[k].in_span(f)) return true;
  }
  if (!e->x_lo) return false;
  double sib = 0.0;
  for (int j = 0; j < e->K; ++j) {
    if (j == k) continue;
    if (c.colflags[j]) return true;       // an inf/NaN sibling: rank 0 (or LinAlgError) whatever the candidate is
    sib = std::max(sib, c.colmax[j]);
  }
  const Iv r = tree_range(t, t.root, e->x_lo, e->x_hi, (double)e->N);
  const double est = std::max(std::fabs(r.lo), std::fabs(r.hi));
  if (!(est == est)) return false;
  if (!std::isfinite(est)) return true;  // overflows: inf in the column, rank 0
  const double tol = (double)std::max<int64_t>(e->N, e->K) * 2.220446049250313e-16;  // numpy's relative rank tolerance
  return est > sib * (10.0 / tol) || est < sib * (tol / 10.0);
}


// diagnostics (BSR_ENGINE_DUMP_GATE=1): an expression as text
std::string tree_text(const Tree& t, int i) {
  static const char* nm[16] = {"inv", "ln", "neg", "sin", "cos", "exp", "sq", "cub", "+", "*", "x", "?", "?", "-", "/", "log"};
  const TNode& nd = t.n[i];
  if (nd.type == 0) return "x" + std::to_string(nd.feature);
  if (nd.type == 1) {
    if (nd.op == OP_LN) { char b[64]; snprintf(b, sizeof b, "ln[%.3g,%.3g](", nd.a, nd.b); return b + tree_text(t, nd.left) + ")"; }
    return std::string(nm[nd.op & 15]) + "(" + tree_text(t, nd.left) + ")";
  }
  return "(" + tree_text(t, nd.left) + " " + nm[nd.op & 15] + " " + tree_text(t, nd.right) + ")";
}

// `ahead`: candidates of this chain generated before and not consumed yet (a batch in flight: they are assumed to
// end as speculated); the new ones continue the sweep behind them, from the random stream where it stands
void generate(bsr_engine* e, ChainS& c, int max_n, int ahead = 0, bool memo_on = false, bool device_mh = true) {
  recycle(c, c.cands);
  c.cands.reserve((size_t)std::max(0, max_n));
  c.gen_start = c.rng;   // (one copy of the stream's state per batch; the candidates carry positions in it)
  int total = c.total + ahead, count = (c.count + ahead) % e->K;
  while ((int)c.cands.size() < max_n) {
    if (count == 0 && total >= e->val) break;  // `while total < val` is only tested between sweeps
    c.cands.emplace_back();
    Cand& cd = c.cands.back();
    if (!c.pool_nodes.empty()) {
      cd.tree.n.swap(c.pool_nodes.back());
      c.pool_nodes.pop_back();
    }
    if (!c.pool_tape.empty()) {
      cd.tape.swap(c.pool_tape.back());
      c.pool_tape.pop_back();
      cd.tape.clear();
    }
    const int k = count;
    cd.k = k;
    if (!c.setup_ok[k]) {
      build_setup(c.roots[k], c.setup[k]);
      c.setup_ok[k] = 1;
    }
    cd.tree.n.reserve(c.setup[k].base.n.size() + 8);
    cd.tree = c.setup[k].base;
    static thread_local Move mv;   // (keeps its lists' storage; every field is set by prop_inplace)
    prop_inplace(cd.tree, e->P, c.siga[k], c.sigb[k], c.rng, mv, &c.setup[k]);
    cd.new_sigma = invgamma_rvs(c.rng, 4);
    aux_inplace(cd.tree, mv, c.siga[k], c.sigb[k], c.rng, &cd.new_sa2, &cd.new_sb2, &cd.hratio, &cd.detjacob);
    cd.change = mv.change;
    cd.Q = mv.Q;
    cd.Qinv = mv.Qinv;
    cd.action = mv.action;
    cd.before_u = mark_of(c.rng);
    cd.ghash = tree_hash_full(cd.tree, cd.tree.root, &cd.gcheck) * 0x9E3779B97F4A7C15ull + (uint64_t)k;
    // A candidate the chain has had scored in its current state (ScoreMemo: more than half of what a chain generates
    // between two accepts) is not sent to the GPU again -- so it needs no tape here (an accept flattens it then), and
    // its rank is not a guess: the table has the gate's verdict.
    const bsr_score* known = memo_on ? c.memo.peek(cd.ghash, cd.gcheck) : nullptr;
    if (known && known->rank >= 0) {
      cd.tape.clear();
      cd.pred_def = e->predict_gate && known->rank < e->K;
    } else {
      flatten(cd.tree, cd.tree.root, cd.tape);
      cd.pred_def = predict_gate_reject(e, c, cd.tree, k, cd.tape) ||
                    (e->predict_gate && std::find(c.gate_memo.begin(), c.gate_memo.end(), cd.ghash) != c.gate_memo.end());
      if (cd.pred_def && std::find(c.gate_pass_memo.begin(), c.gate_pass_memo.end(), cd.ghash) != c.gate_pass_memo.end())
        cd.pred_def = false;
    }
    cd.u = cd.pred_def ? kNaN : c.rng.uniform();  // a proposal speculated as gate-rejected draws no uniform
    {  // the scalar terms of codes/funcs.py:1230-1296 that do not depend on the score
      fstruc(cd.tree, cd.tree.root, e->P, cd.new_sa2, cd.new_sb2, &cd.sn_s, &cd.sn_p);
      if (!c.fs_old_ok[k]) {
        fstruc(c.roots[k], c.roots[k].root, e->P, c.siga[k], c.sigb[k], &c.fs_old_s[k], &c.fs_old_p[k]);
        c.fs_old_ok[k] = 1;
      }
      cd.mhflags = (cd.change != CH_NONE ? BSR_MH_JUMP : 0) | (cd.pred_def ? BSR_MH_NO_UNIFORM : 0);
      if (device_mh) {   // (only the device-side MH step reads them: six logarithms per candidate the host-side step forms itself)
        cd.terms[0] = -c.sse_old / (2 * c.sigma * c.sigma) - 0.5 * (double)e->N * std::log(2 * M_PI * c.sigma * c.sigma);
        cd.terms[1] = (cd.change != CH_NONE) ? (c.fs_old_s[k] + c.fs_old_p[k]) - (cd.sn_s + cd.sn_p) : (c.fs_old_s[k] - cd.sn_s);
        cd.terms[2] = flog(pymax(1e-5, fdiv(cd.Qinv, cd.Q)));
        cd.terms[3] = (cd.change != CH_NONE) ? flog(pymax(1e-5, cd.hratio)) : 0.0;
        cd.terms[4] = (cd.change != CH_NONE) ? flog(pymax(1e-5, cd.detjacob)) : 0.0;
        cd.terms[5] = flog(invgamma_pdf(cd.new_sigma, 4));
        cd.terms[6] = flog(invgamma_pdf(c.sigma, 4));
        cd.terms[7] = cd.pred_def ? kNaN : flog(cd.u);
      } else {
        for (double& v : cd.terms) v = 0.0;
      }
    }
    ++total;
    count = (count + 1) % e->K;
  }
  c.end_state = c.rng;
}

// lp_sigma: flog(invgamma_pdf(sigma, 4)) of the chain's current sigma (the caller keeps it from proposal to proposal)
double log_ratio(const Cand& c, double yllstar, double yll, double sn_s, double sn_p, double so_s, double so_p,
                 double lp_sigma) {  // codes/funcs.py:1230-1296
  const double log_y = yllstar - yll;
  const double log_q = flog(pymax(1e-5, fdiv(c.Qinv, c.Q)));
  double logR;
  if (c.change != CH_NONE) {
    const double log_s = (so_s + so_p) - (sn_s + sn_p);
    logR = log_y + log_s + log_q + flog(pymax(1e-5, c.hratio)) + flog(pymax(1e-5, c.detjacob));
  } else {
    logR = log_y + (so_s - sn_s) + log_q;
  }
  return logR + flog(invgamma_pdf(c.new_sigma, 4)) - lp_sigma;
}

// The device half of an accepted move (codes/bsr_class.py:211-252): the accepted tree's column into the chain's cache,
// the basis and both fits refreshed, the RMSE appended, stop rule 2.  at >= 0: the accepted proposal's place in the batch
// last scored on batch_slot (or, batch_slot < 0, in the context's last waited batch); at < 0: the proposal was answered
// from the score memo and is in no batch -- its tape (c.tapes[k], already the chain's own) is scored alone first, on
// that slot, and committed from there: bsr_commit re-runs the staged tape, the same bytes whichever batch staged it.
struct DeferredAccept {
  ChainS* c;
  int k;
  bsr_trace* tr;
};
int finish_accept(bsr_engine* e, ChainS& c, int k, int batch_slot, int at, bsr_trace* tr) {
  double rmse;
  {
    struct CtxLock {
      bsr_ctx* c;
      explicit CtxLock(bsr_ctx* c_) : c(c_) { if (c) bsr_internal_lock(c); }
      ~CtxLock() { if (c) bsr_internal_unlock(c); }
    } lk(batch_slot >= 0 ? e->ctx : nullptr);
    if (at < 0) {
      const std::vector<bsr_node>& tape = c.tapes[k];
      const int32_t off2[2] = {0, (int32_t)tape.size()};
      const int32_t ch1 = c.index, k1 = k;
      const double sg1 = c.sigma;
      bsr_score one;
      if (batch_slot >= 0) {
        ECHK(e, bsr_internal_submit(e->ctx, batch_slot, tape.data(), off2, &ch1, &k1, &sg1, 1));
        ECHK(e, bsr_internal_wait(e->ctx, batch_slot, &one));
      } else {
        int32_t tk = -1;
        ECHK(e, bsr_score_submit(e->ctx, tape.data(), off2, &ch1, &k1, &sg1, 1, &tk));
        ECHK(e, bsr_score_wait(e->ctx, tk, &one));
      }
      at = 0;
    }
    if (batch_slot >= 0) ECHK(e, bsr_internal_commit(e->ctx, batch_slot, c.index, k, at));
    else ECHK(e, bsr_commit(e->ctx, c.index, k, at));
    int rc = refresh_chain(e, c);
    if (rc != BSR_OK) return rc;
    ECHK(e, bsr_fit_beta(e->ctx, c.index, c.Beta.data(), &rmse));
  }
  c.errs.push_back(rmse);
  if (tr) tr->rmse = rmse;
  const int m = std::min<int>(10, (int)c.errs.size());  // codes/bsr_class.py:248-252
  if (c.errs.size() > 100) {
    double mn = kInf, sum = 0;
    for (int j = 0; j < m; ++j) {
      const double v = c.errs[c.errs.size() - m + j];
      mn = std::min(mn, v);
      sum += v;
    }
    if (1 - mn / (sum / m) < 0.05) c.done = true;
  }
  return BSR_OK;
}

// batch_slot < 0: the batch was scored through the public ticket API (one thread); otherwise by a worker thread that
// owns that batch slot, and the accept path takes the context lock (commit, refresh and fit share the main stream)
// ev != nullptr: the device has already formed every log-ratio and found the first proposal of the run that is not
// "rejected as speculated" (k_events); the proposals in front of it only need their bookkeeping.
// keep_rng: more candidates of this chain were generated behind these (a batch ahead): when every candidate here ends
// as speculated the random stream already stands where it should; *broke_out tells the caller whether an event
// (accept, gate verdict against the speculation, error) ended the run early -- what was generated behind is then void
int consume(bsr_engine* e, ChainS& c, const bsr_score* res, int slot0, int batch_slot, const bsr_event* ev,
            bool keep_rng = false, bool* broke_out = nullptr, const int32_t* gpu_index = nullptr,
            std::vector<DeferredAccept>* deferred = nullptr) {
  const int K = e->K;
  int used = 0;
  bool broke = false;
  if (broke_out) *broke_out = true;   // (early error returns count as events)
  for (size_t i = 0; i < c.cands.size(); ++i) {
    Cand& cd = c.cands[i];
    const bsr_score& sc = res[i];
    ++used;
    ++c.n_props;
    const int k = cd.k;
    if (c.last_stale) {  // `Roots` is rebuilt before every newProp (codes/bsr_class.py:180-182)
      c.last_roots = c.roots;
      c.last_stale = false;
    }
    if (ev && !e->verify_mh && (int)i < ev->index && !(sc.rank < 0 && !e->nan_reject)) {
      ++c.total;
      c.count = (k + 1) % K;
      if (sc.rank < K) {
        ++c.n_rank_rej;
        c.def_ema[k] = 0.75 * c.def_ema[k] + 0.25;
      } else {
        c.def_ema[k] *= 0.75;
      }
      continue;
    }
    if (ev && e->verify_mh) {  // every decision recomputed below must agree with the device's scan
      const bool device_says_event = (int)i == ev->index;
      if ((int)i > ev->index) return efail(e, BSR_E_STATE, "device MH scan stopped early");
      c.verify_expect_event = device_says_event ? ev->kind : BSR_EV_NONE;
    }
    bsr_trace* tr = nullptr;
    if (e->trace && e->n_trace < e->trace_cap) {
      tr = &e->trace[e->n_trace++];
      memset(tr, 0, sizeof *tr);
      tr->chain = c.index;
      tr->count = k;
      tr->action = cd.action;
      tr->change = cd.change;
      tr->rank = sc.rank;
      tr->Q = cd.Q;
      tr->Qinv = cd.Qinv;
      tr->new_sigma = cd.new_sigma;
      tr->new_sa2 = cd.new_sa2;
      tr->new_sb2 = cd.new_sb2;
      tr->tree_hash = tree_hash(cd.tree, cd.tree.root);
      tr->n_nodes = count_nodes(cd.tree, cd.tree.root);
    }
    if (sc.rank < 0 && !e->nan_reject) {
      replay_to(c.rng, c.start_state, cd.before_u);
      return efail(e, BSR_E_LINALG, "SVD did not converge");  // NaN in new_outputs, codes/funcs.py:1226
    }
    ++c.total;
    c.count = (k + 1) % K;
    if (sc.rank < K) {  // codes/funcs.py:1226-1228: no uniform drawn
      ++c.n_rank_rej;
      c.def_ema[k] = 0.75 * c.def_ema[k] + 0.25;
      if (cd.pred_def) {  // speculated exactly that: the candidates behind it are on the right stream
        ++c.n_pred_ok;
        if (e->verify_mh && ev && c.verify_expect_event != BSR_EV_NONE) return efail(e, BSR_E_STATE, "device MH scan: spurious event");
        continue;
      }
      if (e->verify_mh && ev && c.verify_expect_event != BSR_EV_GATE) return efail(e, BSR_E_STATE, "device MH scan missed a gate rejection");
      replay_to(c.rng, c.start_state, cd.before_u);
      ++c.n_evt_gate;
      if (c.gate_memo.size() < 256) c.gate_memo.push_back(cd.ghash);
      if (getenv("BSR_ENGINE_DUMP_GATE")) {
        std::string l = "unpredicted gate rejection k=" + std::to_string(k) + ": " + tree_text(cd.tree, cd.tree.root) + "  | siblings:";
        for (int j = 0; j < K; ++j) if (j != k) l += " [" + tree_text(c.roots[j], c.roots[j].root) + "]";
        fprintf(stderr, "%s  smin/smax %.3g/%.3g\n", l.c_str(), sc.smin, sc.smax);
      }
      broke = true;
      break;
    }
    c.def_ema[k] *= 0.75;
    bool tail_invalid = false;
    if (cd.pred_def) {  // the gate passed a proposal speculated as rejected: its uniform is drawn now, from the state
      replay_to(c.rng, c.start_state, cd.before_u);  // in front of it; whatever follows in the batch was generated on a shifted stream
      cd.u = c.rng.uniform();
      ++c.n_evt_pass;
      if (c.gate_pass_memo.size() < 256) c.gate_pass_memo.push_back(cd.ghash);
      tail_invalid = true;
    }
    // Without the device-side MH scan the log-likelihood is formed HERE from the scored SSE (the device's formula,
    // csrc/bsr_solve.h, evaluated by the host's libm -- as `yll` below always was): a proposal answered from the score
    // memo and the same proposal scored by the GPU then give the same bits, so the memo cannot move a chain.
    const double yllstar = ev ? sc.loglik
                              : -sc.sse / (2 * cd.new_sigma * cd.new_sigma) - 0.5 * (double)e->N * std::log(2 * M_PI * cd.new_sigma * cd.new_sigma);
    // (the two terms that belong to the chain's state, not to the proposal: the same value for the ~1 000 proposals
    // between two accepted ones -- three logarithms and an exponential per proposal otherwise)
    if (!(c.yll_sse == c.sse_old && c.yll_sigma == c.sigma)) {
      c.yll_sse = c.sse_old;
      c.yll_sigma = c.sigma;
      c.yll_val = -c.sse_old / (2 * c.sigma * c.sigma) - 0.5 * (double)e->N * std::log(2 * M_PI * c.sigma * c.sigma);
      c.lp_sigma_val = flog(invgamma_pdf(c.sigma, 4));
    }
    const double yll = c.yll_val;
    const double sn_s = cd.sn_s, sn_p = cd.sn_p;  // fStruc of the proposed tree, computed when it was generated
    if (!c.fs_old_ok[k]) {
      fstruc(c.roots[k], c.roots[k].root, e->P, c.siga[k], c.sigb[k], &c.fs_old_s[k], &c.fs_old_p[k]);
      c.fs_old_ok[k] = 1;
    }
    const double logR = log_ratio(cd, yllstar, yll, sn_s, sn_p, c.fs_old_s[k], c.fs_old_p[k], c.lp_sigma_val);
    const double alpha = (0 < logR) ? 0 : logR;  // Python's min(logR, 0)
    const bool accepted = !(flog(cd.u) >= alpha);
    if (tr) {
      tr->yllstar = yllstar;
      tr->yll = yll;
      tr->logR = logR;
      tr->u = cd.u;
      tr->accepted = accepted;
    }
    if (e->verify_mh && ev) {
      const int want = tail_invalid ? BSR_EV_GATE_PASSED : (accepted ? BSR_EV_ACCEPT : BSR_EV_NONE);
      if (c.verify_expect_event != want) return efail(e, BSR_E_STATE, "device MH scan disagrees with the host's accept test");
      if (want != BSR_EV_NONE && !(ev->logR == logR || (ev->logR != ev->logR && logR != logR)))
        return efail(e, BSR_E_STATE, "device logR differs from the host's");
    }
    if (!accepted) {
      if (tail_invalid) {  // rejected on the uniform just drawn: the chain goes on from the state behind it
        broke = true;
        break;
      }
      continue;
    }
    // ---- accepted: codes/bsr_class.py:200-243
    ++c.n_accept;
    c.gate_memo.clear();   // (the siblings of every k change with this move)
    c.gate_pass_memo.clear();
    c.memo.clear();        // (... and so does every score)
    c.last_roots = c.roots;  // the list built before this newProp: stale by this accept if the chain stops now
    c.last_stale = true;
    if (cd.tape.empty()) flatten(cd.tree, cd.tree.root, cd.tape);   // (a candidate answered from the score memo: generated without its tape)
    c.roots[k] = cd.tree;
    c.ckey[k] = canon_key(cd.tree);
    c.tapes[k] = cd.tape;
    rebuild_sibling_spans(c, K);
    c.sigma = cd.new_sigma;
    c.siga[k] = cd.new_sa2;
    c.sigb[k] = cd.new_sb2;
    c.total = 0;
    for (int j = 0; j < K; ++j)
      if (j != k) c.def_ema[j] = 0.0;  // their sibling set has changed
    replay_to(c.rng, c.start_state, cd.before_u);
    c.rng.uniform();
    // the accepted proposal by its place in the batch the GPU scored -- or, where it was answered from the score memo
    // (it is in no batch), by its tape
    const int at = gpu_index ? gpu_index[slot0 + (int)i] : slot0 + (int)i;
    if (at < 0) {
      // Answered from the score memo: its tape is staged in no batch, so it goes to the GPU alone and the commit is made
      // from THAT one-tape batch.  Not here: the one-tape batch takes the lane's slot (or moves the context's "last waited
      // batch"), and the chains of this batch that are consumed behind this one still commit by their place in it.  The
      // device half of the accept waits until every chain of the batch has been consumed (finish_accept, run by the
      // caller); nothing the other chains do depends on this chain's refreshed state.
      if (!deferred) return efail(e, BSR_E_STATE, "a memo-answered accept needs the caller's deferred list");
      deferred->push_back({&c, k, tr});
    } else {
      int rc = finish_accept(e, c, k, batch_slot, at, tr);
      if (rc != BSR_OK) return rc;
    }
    broke = true;
    break;
  }
  if (!broke && !keep_rng) c.rng = c.end_state;  // every candidate was consumed as a plain rejection
  if (broke_out) *broke_out = broke;
  c.n_discard += (int64_t)c.cands.size() - used;
  if (broke) c.run_ema = (c.run_ema > 1e8) ? used : 0.7 * c.run_ema + 0.3 * used;
  else if (c.run_ema < 1e8) c.run_ema = 0.7 * c.run_ema + 0.3 * (2.0 * used);
  if (!c.done && c.count == 0 && c.total >= e->val) c.done = true;
  recycle(c, c.cands);
  return BSR_OK;
}

}  // namespace

extern "C" int bsr_engine_create(bsr_engine** out, bsr_ctx* ctx, int32_t n_chains, int32_t K, int64_t N,
                                 int32_t n_feature, double beta, int32_t val, int32_t y_is_series) {
  if (!out || !ctx || n_chains <= 0 || K <= 0 || K > BSR_MAX_K) return BSR_E_ARG;
  bsr_engine* e = new bsr_engine();
  e->ctx = ctx;
  e->K = K;
  e->N = N;
  e->n_chains = n_chains;
  e->val = val;
  e->y_is_series = y_is_series;
  e->P.n_feature = n_feature;
  e->P.beta = beta;
  e->P.fill_depth_tables();
  e->P.set_default_table();
  if (getenv("BSR_ENGINE_PREDICT")) e->predict_gate = atoi(getenv("BSR_ENGINE_PREDICT")) != 0;
  if (getenv("BSR_ENGINE_MEMO")) e->score_memo = atoi(getenv("BSR_ENGINE_MEMO")) != 0;
  bsr_internal_feature_range(ctx, &e->x_lo, &e->x_hi);
  if (getenv("BSR_ENGINE_DEVICE_MH")) e->device_mh = atoi(getenv("BSR_ENGINE_DEVICE_MH")) != 0;
  if (getenv("BSR_ENGINE_VERIFY_MH")) e->verify_mh = atoi(getenv("BSR_ENGINE_VERIFY_MH")) != 0;
  e->chains.resize(n_chains);
  for (int c = 0; c < n_chains; ++c) {
    e->chains[c].index = c;
    e->chains[c].rng.seed(0);
  }
  *out = e;
  return BSR_OK;
}

extern "C" int bsr_engine_destroy(bsr_engine* e) {
  delete e;
  return BSR_OK;
}

extern "C" const char* bsr_engine_last_error(const bsr_engine* e) { return e ? e->err.c_str() : ""; }

// Ops / Op_weights of codes/bsr_class.py:110-112 as data: opcodes[i] is the opcode of table entry i (arity follows
// from it), weights[i] its prior weight (np.random.choice normalises them).  Call before bsr_engine_init_chain.
extern "C" int bsr_engine_set_ops(bsr_engine* e, int32_t n_ops, const int32_t* opcodes, const double* weights) {
  if (!e || !opcodes || !weights || n_ops <= 0 || n_ops > BSR_MAX_OPS) return BSR_E_ARG;
  int codes[BSR_MAX_OPS];
  double total = 0;
  for (int i = 0; i < n_ops; ++i) {
    const int c = opcodes[i];
    const bool known = (c >= 0 && c < BSR_OP_TERMINAL) || c == BSR_OP_SUB || c == BSR_OP_DIV || c == BSR_OP_LOG;
    if (!known || !(weights[i] >= 0)) return efail(e, BSR_E_ARG, "bsr_engine_set_ops: unknown opcode or negative weight");
    codes[i] = c;
    total += weights[i];
  }
  if (!(total > 0)) return efail(e, BSR_E_ARG, "bsr_engine_set_ops: weights sum to zero");
  e->P.set_table(n_ops, codes, weights);
  return BSR_OK;
}

extern "C" int bsr_engine_set_nan_policy(bsr_engine* e, int32_t reject) {
  if (!e) return BSR_E_ARG;
  e->nan_reject = reject ? 1 : 0;
  return BSR_OK;
}

extern "C" int bsr_engine_seed(bsr_engine* e, int32_t chain, uint32_t seed) {
  if (!e || chain < 0 || chain >= e->n_chains) return BSR_E_ARG;
  e->chains[chain].rng.seed(seed);
  return BSR_OK;
}

extern "C" int bsr_engine_set_rng(bsr_engine* e, int32_t chain, const uint32_t* key624, int32_t pos, int32_t has_gauss,
                                  double gauss) {
  if (!e || !key624 || chain < 0 || chain >= e->n_chains || pos < 0 || pos > 624) return BSR_E_ARG;
  LegacyRng& r = e->chains[chain].rng;
  memcpy(r.key, key624, sizeof r.key);
  r.pos = pos;
  r.has_gauss = has_gauss;
  r.gauss = gauss;
  return BSR_OK;
}

extern "C" int bsr_engine_get_rng(bsr_engine* e, int32_t chain, uint32_t* key624, int32_t* pos, int32_t* has_gauss,
                                  double* gauss) {
  if (!e || !key624 || !pos || !has_gauss || !gauss || chain < 0 || chain >= e->n_chains) return BSR_E_ARG;
  const LegacyRng& r = e->chains[chain].rng;
  memcpy(key624, r.key, sizeof r.key);
  *pos = r.pos;
  *has_gauss = r.has_gauss;
  *gauss = r.gauss;
  return BSR_OK;
}

extern "C" int bsr_engine_init_chain(bsr_engine* e, int32_t chain) {
  if (!e || chain < 0 || chain >= e->n_chains) return BSR_E_ARG;
  return init_chain(e, e->chains[chain]);
}

// Advances every initialised, unfinished chain until it is done (or has consumed max_props proposals).
// Chains are dealt into up to four groups (BSR_ENGINE_GROUPS; each needs a batch slot of its own); each group's batch (up to batch_per_chain speculative
// proposals per chain) is one asynchronous submission, so proposal generation and result handling of one group
// overlap the GPU work of the others.  A chain belongs to one group only, hence never depends on a batch in flight.
extern "C" int bsr_engine_run(bsr_engine* e, int32_t batch_per_chain, int64_t max_props, bsr_trace* trace,
                              int64_t trace_cap, int64_t* n_trace, int32_t max_batch) {
  if (!e || batch_per_chain <= 0 || max_batch <= 0) return BSR_E_ARG;
  e->trace = trace;
  e->trace_cap = trace ? trace_cap : 0;
  e->n_trace = 0;
  // One batch in flight: its tapes, what was asked per proposal, the results, and per chain of the group the
  // candidates it carries (a chain's candidate list travels with the batch: with a second batch generated ahead the
  // chain object itself only holds what is being consumed).
  struct Lane {
    std::vector<bsr_node> rows;
    std::vector<int32_t> off, chs, ks;
    std::vector<double> sig, terms;
    std::vector<int32_t> mhflags, spans;
    std::vector<bsr_event> events;
    std::vector<bsr_score> res;
    // the score memo: which proposals of the batch are answered from their chain's table (their scores, ready), and the
    // batch that actually goes to the GPU -- the others, compacted; gpu_pos: where a GPU result belongs in `res`
    std::vector<char> hit;
    std::vector<bsr_score> hit_val;
    std::vector<bsr_node> g_rows;
    std::vector<int32_t> g_off, g_chs, g_ks, gpu_pos, gpu_of;   // gpu_of: a proposal's place in the GPU's batch, -1: none
    std::vector<double> g_sig;
    std::vector<bsr_score> g_res;
    bool compact = false, no_gpu = false;
    std::vector<std::pair<int, int>> span;       // per chain of the group: first proposal, count
    std::vector<std::vector<Cand>> cands;        // per chain: its candidates in this batch
    std::vector<LegacyRng> end_state;            // per chain: the random stream behind its last candidate
    std::vector<LegacyRng> start_state;          // ... and in front of its first
    std::vector<char> valid;                     // per chain: 0 once an event in the batch before made them void
    int32_t ticket = -1;
    int slot = -1;  // >= 0: this group's worker thread owns that batch slot
    bool inflight = false;
    double t_sent = 0;   // when the batch was handed to the library (BSR_ENGINE_PROF: the latency of the batches waited for)
    int n_sent = 0;      // ... and how many tapes it held
  };
  // A group's chains are independent of each other between the batch's assembly and its results (a chain's candidates
  // depend on its own trees and its own random stream; consuming them touches that chain alone, the rare accept takes the
  // context lock), so a group of several chains deals the per-chain halves of its cycle -- generate() and consume() -- to
  // HELPER threads of its own: the batch the GPU launches stays the group's (64 tapes cost the chip little more than
  // 32), the threads that generate and consume are as many as the chains.  A chain always goes to the same thread
  // (index in the group modulo the group's threads: what it generated is in that core's cache when it is consumed);
  // the worker posts a round (kind + per-chain arguments), does its own share and goes on once every helper has
  // reported the round done, so nothing else touches a chain while a helper works on it.
  enum { JOB_GENERATE = 0, JOB_CONSUME = 1 };
  struct HelperCtl {
    std::atomic<uint32_t> go{0}, done{0};
    char pad[56];
  };
  struct Group {
    std::vector<ChainS*> chains;
    Lane lane[3];   // the batch being consumed next and up to two generated ahead of it
    int fifo[3] = {0, 0, 0}, n_fly = 0;   // lanes in flight, oldest first
    double t_gen = 0, t_submit = 0, t_wait = 0, t_consume = 0;
    double lat_waited = 0;                                   // BSR_ENGINE_PROF: submit -> results, summed over the batches waited for
    int64_t n_batches = 0, n_waited = 0, n_tapes = 0, n_cands = 0;
    double evt_ema = 0.0;   // share of this group's chain batches that ended in an event lately (lookahead pays while it is low)
    // the round on offer (written by the worker between rounds only)
    int job_kind = JOB_GENERATE, job_lane = 0;
    std::vector<int> room_of, ahead_of;               // JOB_GENERATE: per chain, room < 0: not live
    std::vector<int> ev_of, rc_of;                    // JOB_CONSUME: per chain, index of its span in the MH scan; its return code
    std::vector<char> broke_of;
    std::vector<std::vector<DeferredAccept>> deferred_of;
    std::unique_ptr<HelperCtl[]> ctl;
    std::atomic<int> quit{0};
    uint32_t round = 0;
    int n_helpers = 0;
  };
  auto is_live = [&](const ChainS& c) { return c.inited && !c.done && (max_props < 0 || c.n_props < max_props); };
  std::vector<ChainS*> live;
  for (auto& c : e->chains)
    if (is_live(c)) live.push_back(&c);
  int rc = BSR_OK;
  if (live.empty()) {
    if (n_trace) *n_trace = 0;
    return rc;
  }
  // tracing wants proposals in chain order: keep one group then
  // A group's cycle is serial (generate, submit, wait for the batch's dependent kernels, consume), so the batches in
  // flight are what hides it: four groups (a worker thread each) with a second, lookahead batch per group (below).
  // 8 chains x 32 at K = 3, N = 100k, consumed proposals/s (tools/probes/engine_lookahead_ab.sh, three rounds in one
  // box): 4 groups with lookahead 1.69-1.82 M, 8 groups with 1.49-1.63 M, 8 without 1.56-1.69 M (round 2's default),
  // 4 without 1.44-1.45 M.  At K = 8, where k_solve takes 27 us per launch whatever its size, eight launches of 32
  // proposals lose against four of 64 anyway.
  const int dflt_groups = 4;
  const int max_groups = std::max(1, std::min<int>(BSR_MAX_INFLIGHT, getenv("BSR_ENGINE_GROUPS") ? atoi(getenv("BSR_ENGINE_GROUPS")) : dflt_groups));
  const int n_groups = trace ? 1 : std::max(1, std::min<int>(max_groups, (int)live.size()));
  // One worker thread per group (each with its own batch slots and HIP streams): proposal generation, staging and the
  // HIP calls of a submission cost the host ~1.3 us per proposal, more than the GPU needs to score it, so a single
  // host thread leaves the GPU two thirds idle.  K == 1 keeps the single-threaded ticket path (its rescoring step
  // drains every slot).
  const int want_threads = getenv("BSR_ENGINE_THREADS") ? atoi(getenv("BSR_ENGINE_THREADS")) : 1;
  // (a single group -- one chain -- takes the same path on the caller's thread: what it gains is the lookahead batch;
  // served by the plain ticket loop below it sat out every batch's time on the GPU, 0.32 M consumed proposals/s at C2)
  const bool threaded = !trace && e->K > 1 && want_threads != 0;
  // A SECOND batch per group, generated while the first is on the GPU on the assumption that the first ends as
  // speculated (four in five do): a worker used to wait 63 % of its time for its batch.  An event in the first batch
  // (accept, gate verdict against the speculation) makes the chain's share of the second one void -- it is skipped
  // when it arrives, and the chain generates afresh from the state behind the event: the sequence of consumed
  // proposals is the reference's whatever is thrown away (codes/funcs.py:1300-1303, :1226-1228).
  // (Until round 4 it was off for K >= 5: there, before the gate's memory, six in seven batches ended in an event.)
  // BSR_ENGINE_LOOKAHEAD: batches generated ahead per group, 0..2 (two need a third batch slot per group: up to four
  // groups).  Default: two for a single group -- a lone chain's thread still spent a quarter of its cycle waiting with
  // one; 0.61-0.66 -> 0.72 M consumed proposals/s -- one otherwise (eight chains in four groups: 2.4-2.7 M either way;
  // K = 8: one chain 0.31 -> 0.48 M with one ahead, the same with two; eight chains 1.0 M with none or one, 0.83 M with
  // two -- until the gate's memory six in seven of its batches ended in an event and lookahead was off for K >= 5).
  // The second only pays since the gate's memory (ChainS::gate_memo) made events rare: with one every ~100 proposals it
  // voided two batches instead of one and gained nothing (discarded 6 711 -> 11 064 of 20 000 consumed, 0.56 M/s either
  // way).
  const int la_env = getenv("BSR_ENGINE_LOOKAHEAD") ? atoi(getenv("BSR_ENGINE_LOOKAHEAD")) : -1;
  const int la_max = n_groups <= 4 ? 2 : 1;
  const int la_depth = !threaded ? 0 : (la_env >= 0 ? std::min(la_env, la_max) : ((e->K <= 4 && n_groups == 1) ? 2 : 1));
  const bool lookahead = la_depth > 0;
  std::vector<Group> groups(n_groups);
  for (size_t i = 0; i < live.size(); ++i) groups[i % n_groups].chains.push_back(live[i]);
  for (int gi = 0; gi < n_groups; ++gi) {
    groups[gi].lane[0].slot = threaded ? gi : -1;
    groups[gi].lane[1].slot = threaded ? BSR_MAX_INFLIGHT + gi : -1;
    groups[gi].lane[2].slot = threaded ? BSR_MAX_INFLIGHT + 4 + gi : -1;   // (only with up to four groups: la_max)
  }
  const int per_group_cap = std::max(1, max_batch / n_groups);
  const bool use_mh = e->device_mh && !trace && e->K > 1;
  const bool memo_on = e->score_memo && !use_mh;   // (the device-side MH scan wants every proposal's score on the device)

  // the per-chain half of a round
  auto chain_job = [&](Group& g, size_t ci) {
    ChainS& c = *g.chains[ci];
    if (g.job_kind == JOB_GENERATE) {
      if (g.room_of[ci] > 0) generate(e, c, g.room_of[ci], g.ahead_of[ci], memo_on, use_mh);
      else if (g.room_of[ci] == 0) recycle(c, c.cands);
      return;
    }
    Lane& L = g.lane[g.job_lane];
    const int li = g.job_lane;
    g.rc_of[ci] = BSR_OK;
    g.broke_of[ci] = 0;
    g.deferred_of[ci].clear();
    if (L.span[ci].second == 0) return;
    if (!L.valid[ci] || c.done) {   // generated behind a batch that did not end as speculated (or that ended the chain): thrown away unseen
      c.n_discard += (int64_t)L.cands[ci].size();
      recycle(c, L.cands[ci]);
      return;
    }
    const bsr_event* ev = use_mh ? &L.events[g.ev_of[ci]] : nullptr;
    c.cands.swap(L.cands[ci]);
    c.end_state = L.end_state[ci];
    c.start_state = L.start_state[ci];
    if (memo_on)   // what the GPU scored for this chain's (unchanged) state: kept for the repeats to come
      for (size_t q = 0; q < c.cands.size(); ++q) {
        const size_t at = (size_t)L.span[ci].first + q;
        if (at < L.hit.size() && !L.hit[at]) c.memo.put(c.cands[q].ghash, c.cands[q].gcheck, L.res[at]);
      }
    bool more_ahead = false;
    for (int ol = 0; ol < 3; ++ol) {
      const Lane& O = g.lane[ol];
      more_ahead = more_ahead || (ol != li && O.inflight && ci < O.valid.size() && O.valid[ci] && O.span[ci].second > 0);
    }
    bool broke = false;
    g.rc_of[ci] = consume(e, c, L.res.data() + L.span[ci].first, L.span[ci].first, L.slot, ev, more_ahead, &broke,
                          L.compact ? L.gpu_of.data() : nullptr, &g.deferred_of[ci]);
    g.broke_of[ci] = broke ? 1 : 0;
  };
  auto run_round = [&](Group& g, int kind, int lane) {
    g.job_kind = kind;
    g.job_lane = lane;
    const size_t nc = g.chains.size();
    if (g.n_helpers == 0) {
      for (size_t ci = 0; ci < nc; ++ci) chain_job(g, ci);
      return;
    }
    const size_t T = (size_t)g.n_helpers + 1;
    ++g.round;
    for (int h = 0; h < g.n_helpers; ++h) g.ctl[h].go.store(g.round, std::memory_order_release);
    for (size_t ci = 0; ci < nc; ci += T) chain_job(g, ci);
    for (int h = 0; h < g.n_helpers; ++h)
      for (int spins = 0; g.ctl[h].done.load(std::memory_order_acquire) != g.round; ++spins) {
        if (spins < 4096) BSR_CPU_RELAX();
        else std::this_thread::yield();
      }
  };
  auto helper = [&](Group& g, int h) {
    bsr_internal_place_thread();
    const size_t T = (size_t)g.n_helpers + 1;
    uint32_t seen = 0;
    for (int spins = 0;; ++spins) {
      if (g.quit.load(std::memory_order_acquire)) break;
      const uint32_t r = g.ctl[h].go.load(std::memory_order_acquire);
      if (r != seen) {
        seen = r;
        for (size_t ci = (size_t)h + 1; ci < g.chains.size(); ci += T) chain_job(g, ci);
        g.ctl[h].done.store(r, std::memory_order_release);
        spins = 0;
        continue;
      }
      // a worker's round comes every few tens of microseconds while the run lasts: spin for about that long, then give
      // the CPU away between looks
      if (spins < 20000) BSR_CPU_RELAX();
      else std::this_thread::yield();
    }
  };
  // generates and submits the batch of lane `li`; ahead: behind the candidates the group's other lane has in flight
  auto submit = [&](Group& g, int li, bool ahead) -> int {
    Lane& L = g.lane[li];
    L.rows.clear();
    L.off.assign(1, 0);
    L.chs.clear();
    L.ks.clear();
    L.sig.clear();
    L.span.clear();
    L.terms.clear();
    L.mhflags.clear();
    L.spans.assign(1, 0);
    L.hit.clear();
    L.hit_val.clear();
    L.compact = L.no_gpu = false;
    int n_hit = 0;
    const size_t nc = g.chains.size();
    L.cands.resize(nc);
    L.end_state.resize(nc);
    L.start_state.resize(nc);
    L.valid.assign(nc, 1);
    int n_live = 0;
    for (ChainS* c : g.chains) n_live += is_live(*c) ? 1 : 0;
    if (n_live == 0) return BSR_OK;
    const int per = std::max(1, std::min<int>(batch_per_chain, per_group_cap / n_live));
    // generating phase: every live chain's candidates (the group's helper threads take their chains' share)
    const double tg0 = now_s();
    g.room_of.assign(nc, -1);
    g.ahead_of.assign(nc, 0);
    for (size_t ci = 0; ci < nc; ++ci) {
      ChainS* c = g.chains[ci];
      if (!is_live(*c)) continue;
      // what this chain has in flight ahead of the new candidates (the other lane's share, unless an event voided it)
      int n_ahead = 0;
      if (ahead)
        for (int ol = 0; ol < 3; ++ol) {
          const Lane& O = g.lane[ol];
          if (ol != li && O.inflight && ci < O.valid.size() && O.valid[ci]) n_ahead += O.span[ci].second;
        }
      int room = per;
      // speculate only about as far as this chain's batches have recently been consumed
      if (c->run_ema < 1e8) room = std::min(room, std::max(2, (int)std::ceil(2.0 * c->run_ema)));
      if (max_props >= 0) room = (int)std::min<int64_t>(room, max_props - c->n_props - n_ahead);
      g.room_of[ci] = std::max(0, room);
      g.ahead_of[ci] = n_ahead;
    }
    run_round(g, JOB_GENERATE, li);
    g.t_gen += now_s() - tg0;
    for (size_t ci = 0; ci < nc; ++ci) {
      ChainS* c = g.chains[ci];
      recycle(*c, L.cands[ci]);
      if (g.room_of[ci] < 0) {
        L.span.push_back({(int)L.chs.size(), 0});
        continue;
      }
      const int room = g.room_of[ci];
      L.span.push_back({(int)L.chs.size(), (int)c->cands.size()});
      for (const Cand& cd : c->cands) {
        if (memo_on) {
          const bsr_score* h = c->memo.find(cd.ghash, cd.gcheck);
          L.hit.push_back(h ? 1 : 0);
          if (h) {   // rank and SSE; the log-likelihood for this candidate's own sigma is formed where it is used (consume)
            L.hit_val.push_back(*h);
            ++n_hit;
          }
        }
        L.rows.insert(L.rows.end(), cd.tape.begin(), cd.tape.end());
        L.off.push_back((int32_t)L.rows.size());
        L.chs.push_back(c->index);
        L.ks.push_back(cd.k);
        L.sig.push_back(cd.new_sigma);
        L.terms.insert(L.terms.end(), cd.terms, cd.terms + 8);
        L.mhflags.push_back(cd.mhflags);
      }
      if (!c->cands.empty()) L.spans.push_back((int32_t)L.chs.size());
      if (c->cands.empty() && !ahead && room > 0) c->done = true;
      L.end_state[ci] = c->end_state;
      L.start_state[ci] = c->gen_start;
      L.cands[ci].swap(c->cands);
    }
    if (L.chs.empty()) return BSR_OK;
    L.res.resize(L.chs.size());
    const double ts0 = now_s();
    int r;
    const int n_sp = (int)L.spans.size() - 1;
    L.events.resize(std::max(1, n_sp));
    // proposals answered from the memo do not travel: the batch for the GPU is the others, compacted
    const bsr_node* rows_p = L.rows.data();
    const int32_t* off_p = L.off.data();
    const int32_t* chs_p = L.chs.data();
    const int32_t* ks_p = L.ks.data();
    const double* sig_p = L.sig.data();
    int n_gpu = (int)L.chs.size();
    if (n_hit > 0) {
      L.compact = true;
      L.g_rows.clear(); L.g_off.assign(1, 0); L.g_chs.clear(); L.g_ks.clear(); L.g_sig.clear(); L.gpu_pos.clear();
      L.gpu_of.assign(L.chs.size(), -1);
      for (size_t i = 0; i < L.chs.size(); ++i) {
        if (L.hit[i]) continue;
        L.gpu_of[i] = (int32_t)L.gpu_pos.size();
        L.g_rows.insert(L.g_rows.end(), L.rows.begin() + L.off[i], L.rows.begin() + L.off[i + 1]);
        L.g_off.push_back((int32_t)L.g_rows.size());
        L.g_chs.push_back(L.chs[i]);
        L.g_ks.push_back(L.ks[i]);
        L.g_sig.push_back(L.sig[i]);
        L.gpu_pos.push_back((int32_t)i);
      }
      n_gpu = (int)L.g_chs.size();
      L.g_res.resize(std::max(1, n_gpu));
      rows_p = L.g_rows.data(); off_p = L.g_off.data(); chs_p = L.g_chs.data(); ks_p = L.g_ks.data(); sig_p = L.g_sig.data();
    }
    if (n_gpu == 0) {   // every proposal of the batch is a repeat: nothing to launch
      L.no_gpu = true;
      L.inflight = true;
      return BSR_OK;
    }
    if (L.slot >= 0)
      r = use_mh ? bsr_internal_submit_mh(e->ctx, L.slot, L.rows.data(), L.off.data(), L.chs.data(), L.ks.data(),
                                          L.sig.data(), (int)L.chs.size(), L.terms.data(), L.mhflags.data(),
                                          L.spans.data(), n_sp)
                 : bsr_internal_submit(e->ctx, L.slot, rows_p, off_p, chs_p, ks_p, sig_p, n_gpu);
    else
      r = use_mh ? bsr_score_submit_mh(e->ctx, L.rows.data(), L.off.data(), L.chs.data(), L.ks.data(), L.sig.data(),
                                       (int)L.chs.size(), L.terms.data(), L.mhflags.data(), L.spans.data(), n_sp,
                                       &L.ticket)
                 : bsr_score_submit(e->ctx, rows_p, off_p, chs_p, ks_p, sig_p, n_gpu, &L.ticket);
    L.t_sent = now_s();
    L.n_sent = n_gpu;
    g.t_submit += L.t_sent - ts0;
    g.n_batches += 1;
    g.n_tapes += n_gpu;
    g.n_cands += (int64_t)L.chs.size();
    if (r != BSR_OK) return efail(e, r, std::string("bsr_score_submit: ") + bsr_last_error(e->ctx));
    L.inflight = true;
    return BSR_OK;
  };
  // waits for lane `li`'s batch and consumes it chain by chain
  auto collect = [&](Group& g, int li) -> int {
    Lane& L = g.lane[li];
    if (!L.inflight) return BSR_OK;
    L.inflight = false;
    const double tw0 = now_s();
    int r;
    bsr_score* res_p = L.compact ? L.g_res.data() : L.res.data();
    if (L.no_gpu)
      r = BSR_OK;
    else if (use_mh)
      r = (L.slot >= 0) ? bsr_internal_wait_mh(e->ctx, L.slot, L.res.data(), L.events.data())
                        : bsr_score_wait_mh(e->ctx, L.ticket, L.res.data(), L.events.data());
    else
      r = (L.slot >= 0) ? bsr_internal_wait(e->ctx, L.slot, res_p)
                        : bsr_score_wait(e->ctx, L.ticket, res_p);
    const double tw1 = now_s();
    g.t_wait += tw1 - tw0;
    if (!L.no_gpu && tw1 - tw0 > 2e-6) {   // (waited for: its age is its latency)
      g.n_waited += 1;
      g.lat_waited += tw1 - L.t_sent;
    }
    if (r != BSR_OK) return efail(e, r, std::string("bsr_score_wait: ") + bsr_last_error(e->ctx));
    if (L.compact) {   // the GPU's scores to their places, the memo's next to them
      for (size_t j = 0; j < L.gpu_pos.size(); ++j) L.res[L.gpu_pos[j]] = L.g_res[j];
      size_t hv = 0;
      for (size_t i = 0; i < L.hit.size(); ++i)
        if (L.hit[i]) L.res[i] = L.hit_val[hv++];
    }
    const size_t nc = g.chains.size();
    g.ev_of.assign(nc, 0);
    g.rc_of.assign(nc, BSR_OK);
    g.broke_of.assign(nc, 0);
    g.deferred_of.resize(nc);
    int sp = 0;  // chains with proposals in this batch, in order: the spans of the MH scan
    for (size_t i = 0; i < nc; ++i)
      if (L.span[i].second != 0) g.ev_of[i] = sp++;
    std::vector<char> counted(nc, 0);
    for (size_t i = 0; i < nc; ++i) counted[i] = L.span[i].second != 0 && L.valid[i] && !g.chains[i]->done;
    run_round(g, JOB_CONSUME, li);
    std::vector<DeferredAccept> deferred;   // accepts answered from the score memo: their device half runs behind the loop
    for (size_t i = 0; i < nc; ++i) {
      if (!counted[i]) continue;
      g.evt_ema = 0.9 * g.evt_ema + (g.broke_of[i] ? 0.1 : 0.0);
      if (g.broke_of[i])   // what was generated behind these is void
        for (int ol = 0; ol < 3; ++ol) {
          Lane& O = g.lane[ol];
          if (ol != li && O.inflight && i < O.valid.size()) O.valid[i] = 0;
        }
      if (g.rc_of[i] != BSR_OK) return g.rc_of[i];
      deferred.insert(deferred.end(), g.deferred_of[i].begin(), g.deferred_of[i].end());
    }
    // every chain of the batch has been consumed: the lane's slot is free for the one-tape batches of the accepts that
    // were answered from the memo (run inside the loop they replaced the staged batch the later chains of it
    // still committed from)
    for (const DeferredAccept& d : deferred) {
      r = finish_accept(e, *d.c, d.k, L.slot, -1, d.tr);
      if (r != BSR_OK) return r;
    }
    g.t_consume += now_s() - tw1;
    return BSR_OK;
  };

  if (threaded) {
    std::atomic<int> first_rc{BSR_OK};
    auto worker = [&](Group& g, bool own_thread) {
      if (own_thread) bsr_internal_place_thread();   // (the caller's thread, which runs group 0, stays where it is)
      auto free_lane = [&]() {
        for (int l = 0; l < 3; ++l)
          if (!g.lane[l].inflight) return l;
        return -1;
      };
      int r = submit(g, 0, false);
      g.n_fly = 0;
      if (g.lane[0].inflight) g.fifo[g.n_fly++] = 0;
      while (r == BSR_OK && first_rc.load(std::memory_order_relaxed) == BSR_OK && g.n_fly > 0) {
        // batches generated ahead are consumed only if the ones in front of them end as speculated: while more than half
        // of the group's chain batches end in an event (K = 8: six in seven) they would mostly be scored for nothing
        while (r == BSR_OK && lookahead && g.evt_ema < 0.5 && g.n_fly < 1 + la_depth) {
          const int l = free_lane();
          if (l < 0) break;
          r = submit(g, l, true);
          if (r != BSR_OK || !g.lane[l].inflight) break;   // (nothing left to generate ahead)
          g.fifo[g.n_fly++] = l;
        }
        if (r != BSR_OK) break;
        const int cur = g.fifo[0];
        for (int q = 1; q < g.n_fly; ++q) g.fifo[q - 1] = g.fifo[q];
        --g.n_fly;
        r = collect(g, cur);
        if (r != BSR_OK) break;
        if (g.n_fly == 0) {
          r = submit(g, cur, false);
          if (r == BSR_OK && g.lane[cur].inflight) g.fifo[g.n_fly++] = cur;
        }
      }
      for (int li = 0; li < 3; ++li) {
        Lane& L = g.lane[li];
        if (!L.inflight) continue;   // left in flight by an error (here or elsewhere): drain so the context stays usable
        L.inflight = false;
        (void)bsr_internal_wait(e->ctx, L.slot, L.res.data());
        for (size_t i = 0; i < g.chains.size() && i < L.cands.size(); ++i) g.chains[i]->n_discard += (int64_t)L.cands[i].size();
      }
      if (r != BSR_OK) {
        int expect = BSR_OK;
        first_rc.compare_exchange_strong(expect, r);
      }
      g.quit.store(1, std::memory_order_release);   // this group's helpers are done with it (they spin between rounds)
    };
    // Helper threads (Group::ctl): ONE per group of several chains, as far as HALF the CPUs left over next to the workers and
    // the library's two submission threads allow -- the CPU budget of the rank and the L3 domain its threads are
    // confined to (csrc/bsr_place.hip: 8 cores and their SMT siblings on the hosts of MI355X boxes; helpers spin between
    // rounds, so a thread more than there are CPUs stalls everybody: 16 chains with 12 helpers 3.4 M consumed
    // proposals/s against 6.3 M with 4).  Measured, 8 chains x 32, K = 3, N = 100k (profiles/r06_engine_helpers.txt): none
    // 4.1 M/s, one per group 4.4-4.7 M/s; 16 chains 6.3-6.4 M/s; a second batch ahead next to them, threads outside the
    // L3 domain or two groups of four threads: no better.  BSR_ENGINE_HELPERS: the total (0 = none), dealt one per group
    // and round.
    int cpus = (int)std::floor(bsr_internal_cpu_budget());
    if (bsr_internal_placed_cpus() > 0) cpus = std::min(cpus, bsr_internal_placed_cpus());
    int helpers_left = std::max(0, (cpus - n_groups - 2) / 2);   // (half of what is left: with every CPU taken they cost more than they bring -- 8 CPUs: 4.36 M/s with two helpers, 4.44 M/s with none)
    int per_group = 1;
    if (getenv("BSR_ENGINE_HELPERS")) {
      helpers_left = std::max(0, atoi(getenv("BSR_ENGINE_HELPERS")));
      per_group = 1 << 20;
    }
    for (bool any = true; any && helpers_left > 0;) {
      any = false;
      for (Group& g : groups)
        if (helpers_left > 0 && g.n_helpers < per_group && g.n_helpers + 1 < (int)g.chains.size()) {
          ++g.n_helpers;
          --helpers_left;
          any = true;
        }
    }
    std::vector<std::thread> th, hth;
    for (Group& g : groups)
      if (g.n_helpers > 0) g.ctl.reset(new HelperCtl[g.n_helpers]);
    for (Group& g : groups)
      for (int h = 0; h < g.n_helpers; ++h) hth.emplace_back(helper, std::ref(g), h);
    for (int gi = 1; gi < n_groups; ++gi) th.emplace_back(worker, std::ref(groups[gi]), true);
    worker(groups[0], false);
    for (auto& t : th) t.join();
    for (Group& g : groups) g.quit.store(1, std::memory_order_release);
    for (auto& t : hth) t.join();
    rc = first_rc.load();
  } else {
    for (Group& g : groups) {
      rc = submit(g, 0, false);
      if (rc != BSR_OK) break;
    }
    while (rc == BSR_OK) {
      bool any = false;
      for (Group& g : groups) {
        if (!g.lane[0].inflight) continue;
        any = true;
        rc = collect(g, 0);
        if (rc != BSR_OK) break;
        rc = submit(g, 0, false);
        if (rc != BSR_OK) break;
      }
      if (!any) break;
    }
    if (rc != BSR_OK) {  // drain what is still in flight so the context stays usable
      for (Group& g : groups)
        if (g.lane[0].inflight) {
          g.lane[0].inflight = false;
          (void)bsr_score_wait(e->ctx, g.lane[0].ticket, g.lane[0].res.data());
        }
    }
  }
  double lat_w = 0;
  int64_t nb = 0, nw = 0, nt = 0, ncd = 0;
  for (const Group& g : groups) {
    lat_w += g.lat_waited; nb += g.n_batches; nw += g.n_waited; nt += g.n_tapes; ncd += g.n_cands;
    e->t_gen += g.t_gen;
    e->t_submit += g.t_submit;
    e->t_wait += g.t_wait;
    e->t_consume += g.t_consume;
  }
  if (n_trace) *n_trace = e->n_trace;
  e->trace = nullptr;
  if (getenv("BSR_ENGINE_PROF")) {
    fprintf(stderr, "bsr_engine_run (%s): generate %.3f s, submit %.3f s, wait %.3f s, consume %.3f s (thread-seconds)\n",
            threaded ? "worker threads" : "one thread", e->t_gen, e->t_submit, e->t_wait, e->t_consume);
    fprintf(stderr, "  this call: %lld batches to the GPU, %.1f tapes of %.1f candidates each; %lld waited for, %.1f us from submission to results\n",
            (long long)nb, nb ? (double)nt / nb : 0.0, nb ? (double)ncd / nb : 0.0, (long long)nw, nw ? 1e6 * lat_w / nw : 0.0);
    int64_t a = 0, g = 0, ps = 0, ok = 0, rj = 0, np_ = 0;
    for (const ChainS& c : e->chains) {
      a += c.n_accept; g += c.n_evt_gate; ps += c.n_evt_pass; ok += c.n_pred_ok; rj += c.n_rank_rej; np_ += c.n_props;
    }
    fprintf(stderr, "  events in %lld consumed proposals: %lld accepts, %lld gate rejections nobody predicted, %lld predicted "
            "rejections that passed the gate; %lld of %lld gate rejections predicted\n", (long long)np_, (long long)a,
            (long long)g, (long long)ps, (long long)ok, (long long)rj);
  }
  return rc;
}

extern "C" int bsr_engine_memo_stats(const bsr_engine* e, int32_t chain, int64_t* stats2) {
  if (!e || !stats2 || chain < 0 || chain >= (int)e->chains.size()) return BSR_E_ARG;
  stats2[0] = e->chains[chain].memo.hits;
  stats2[1] = e->chains[chain].memo.lookups;
  return BSR_OK;
}

extern "C" int bsr_engine_chain_result(bsr_engine* e, int32_t chain, bsr_node* tapes, int32_t tape_cap,
                                       int32_t* tape_len, double* beta, double* errs, int32_t errs_cap,
                                       int32_t* n_errs, int64_t* counters, double* sigma, int32_t current) {
  if (!e || chain < 0 || chain >= e->n_chains || !tape_len) return BSR_E_ARG;
  ChainS& c = e->chains[chain];
  if (!c.inited) return efail(e, BSR_E_STATE, "chain not initialised");
  std::vector<bsr_node> t;
  const std::vector<Tree>& src = current ? c.roots : c.last_roots;
  for (int k = 0; k < e->K; ++k) {
    flatten(src[k], src[k].root, t);
    tape_len[k] = (int32_t)t.size();
    if (tapes && (int32_t)t.size() <= tape_cap) memcpy(tapes + (size_t)k * tape_cap, t.data(), t.size() * sizeof(bsr_node));
    else if (tapes) tape_len[k] = -(int32_t)t.size();
  }
  if (beta) memcpy(beta, c.Beta.data(), sizeof(double) * (e->K + 1));
  if (n_errs) *n_errs = (int32_t)c.errs.size();
  if (errs && !c.errs.empty())   // (memcpy from a null data() is undefined even for zero bytes: UBSan, tests/native)
    memcpy(errs, c.errs.data(), sizeof(double) * std::min<size_t>(c.errs.size(), (size_t)errs_cap));
  if (counters) {
    counters[0] = c.n_props;
    counters[1] = c.n_accept;
    counters[2] = c.n_rank_rej;
    counters[3] = c.n_discard;
    counters[4] = c.done ? 1 : 0;
  }
  if (sigma) *sigma = c.sigma;
  return BSR_OK;
}

// Draw sequence for tests: kind[i] in {0 uniform, 1 randint(lo,hi), 2 standard_normal, 3 choice10, 4 invgamma(a=lo)}
extern "C" int bsr_rng_selftest(uint32_t seed, int32_t n, const int32_t* kind, const int64_t* lo, const int64_t* hi,
                                double* out) {
  if (!kind || !out) return BSR_E_ARG;
  LegacyRng r;
  r.seed(seed);
  Params P10;
  P10.set_default_table();
  for (int i = 0; i < n; ++i) {
    switch (kind[i]) {
      case 0: out[i] = r.uniform(); break;
      case 1: out[i] = (double)r.randint(lo[i], hi[i]); break;
      case 2: out[i] = r.standard_normal(); break;
      case 3: out[i] = (double)choose_op(P10, r); break;
      default: out[i] = invgamma_rvs(r, (int)lo[i]); break;
    }
  }
  return BSR_OK;
}
