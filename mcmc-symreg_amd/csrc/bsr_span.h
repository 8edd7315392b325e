// Host-side structure analysis of tapes: is a candidate's column inside the span of a chain's K current columns?
//
// The solve step (k_solve) measures rho^2 = |s z|^2 - |Q^T s z|^2 in one pass; when the candidate's column lies
// (nearly) inside the span of the current columns that difference cancels and the proposal would go through the
// residual step -- two more kernels in the batch's queue.  Most such candidates are in the span BY CONSTRUCTION: the tree
// they would replace again (4.6 % of the real mix at K = 3), the same with a negation moved (1.1 %), and -- at K = 8,
// a dozen per batch -- linear combinations of current trees (`-x1` next to `x1 + x1`, `x1 + x6` next to `x1 + x1` and
// `-x6`, `cos(x3) + x6`, `x1 + -x1`).  For those w = 0 exactly is what the residual step would make of them (it
// measures |w|^2 ~ 1e-31 |s z|^2, far below its cut), so the host says so and k_solve skips the step -- provided its own
// one-pass figure agrees that the candidate is in the span (a wrong claim costs nothing but the shortcut).
//
// A tape's column is written as a LINEAR FORM over atoms: sum of coef_i x atom_i, where +, -, neg and ln (a x + b)
// act on the coefficients and everything else makes an atom -- the canonical hash of the subtree with the signs
// carried to the root: neg A -> -A; A*B, A/B multiply the signs; 1/A, A^3, sin A pass the sign on; A^2, cos A drop
// it; exp, log keep it inside (IEEE negation commutes with every rounded operation involved; bsr_fastmath.h's sin is
// odd and its cos even).  Operands of + and * in a fixed order.  The constant column is an atom of its own.
// A chain's K forms are reduced to an echelon basis once per accepted move; a candidate's form is reduced against it.
// Hashes are 64 bits and not confirmed on an exact form: a collision would have to coincide with a column that is
// numerically inside the span anyway (the device's own test) to change a single result.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/bsr_hip.h"

namespace bsr_span {

constexpr int LF_CAP = 6;                               // terms a form holds; a longer sum becomes one atom
constexpr uint64_t LF_CONST = 0x434F4E5354414E55ull;    // atom of the constant column

inline uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

struct LinForm {
  int n = 0;                 // terms, sorted by atom, no zero coefficient
  bool approx = false;       // the form stands for the tape's column to ROUNDING only (an inverse of an inverse was undone)
  uint64_t atom[LF_CAP];
  double coef[LF_CAP];
};

inline uint64_t form_hash(const LinForm& f) {
  uint64_t h = 0x4C494E464F524Dull;
  for (int i = 0; i < f.n; ++i) {
    uint64_t cb;
    memcpy(&cb, &f.coef[i], 8);
    h = mix64(h ^ f.atom[i]) + 3 * mix64(cb);
  }
  return mix64(h);
}
inline void set_atom(LinForm& f, uint64_t a, double c) {
  f.n = 1;
  f.atom[0] = a;
  f.coef[0] = c;
}
// (atom, sign) of a form that stands as an operand of a non-linear operator
inline void as_atom(const LinForm& f, uint64_t* a, int* sign) {
  if (f.n == 1 && (f.coef[0] == 1.0 || f.coef[0] == -1.0)) {
    *a = f.atom[0];
    *sign = f.coef[0] < 0 ? -1 : 1;
  } else {
    *a = form_hash(f);
    *sign = 1;
  }
}
// out = a + sb * b (sb = +-1); a sum longer than LF_CAP terms becomes an atom of its own
inline void add_forms(const LinForm& a, const LinForm& b, double sb, LinForm* out) {
  LinForm r;
  int i = 0, j = 0;
  bool over = false;
  while (i < a.n || j < b.n) {
    uint64_t at;
    double c;
    if (j >= b.n || (i < a.n && a.atom[i] < b.atom[j])) { at = a.atom[i]; c = a.coef[i]; ++i; }
    else if (i >= a.n || b.atom[j] < a.atom[i]) { at = b.atom[j]; c = sb * b.coef[j]; ++j; }
    else { at = a.atom[i]; c = a.coef[i] + sb * b.coef[j]; ++i; ++j; }
    if (c == 0.0) continue;
    if (r.n == LF_CAP) { over = true; break; }
    r.atom[r.n] = at;
    r.coef[r.n] = c;
    ++r.n;
  }
  if (over) {
    uint64_t ha = form_hash(a), hb = form_hash(b);
    if (sb > 0 && hb < ha) std::swap(ha, hb);
    set_atom(r, mix64(mix64(ha ^ 0x2B00000000000000ull) + 3 * hb + (sb < 0 ? 0x5A5Aull : 0)), 1.0);
  }
  *out = r;
}

// what a unary operator does with its operand's sign: +1 passes it on (odd), 0 drops it (even), 2 keeps it inside
inline int unary_sign_rule(int op) {
  switch (op) {
    case BSR_OP_INV: case BSR_OP_CUBIC: case BSR_OP_SIN: return 1;
    case BSR_OP_SQUARE: case BSR_OP_COS: return 0;
    default: return 2;   // exp, log
  }
}

// linear form of the column a postfix tape computes; false: malformed tape / deeper than max_stack
inline bool lin_form(const bsr_node* t, int len, LinForm* out, int max_stack = 26) {
  LinForm st[28];
  if (max_stack > 26) max_stack = 26;
  int sp = 0;
  // 1/(1/A) is A to rounding (the moves stack an inverse on an inverse often enough: `1/[1/[x30]]`): the atoms made
  // by `inv` of a plain +-atom remember it, and an inverse on top of one gives it back
  constexpr int INV_CAP = 8;
  uint64_t inv_of[INV_CAP], inv_base[INV_CAP];
  int n_inv = 0;
  bool undone = false;
  for (int i = 0; i < len; ++i) {
    const int op = t[i].opcode;
    if (op == BSR_OP_TERMINAL) {
      if (sp >= max_stack) return false;
      set_atom(st[sp], mix64(0x7465726Dull ^ ((uint64_t)(uint32_t)t[i].feature << 32)), 1.0);
      ++sp;
    } else if (op == BSR_OP_ADD || op == BSR_OP_SUB) {
      if (sp < 2) return false;
      LinForm r;
      add_forms(st[sp - 2], st[sp - 1], op == BSR_OP_SUB ? -1.0 : 1.0, &r);
      --sp;
      st[sp - 1] = r;
    } else if (op == BSR_OP_MUL || op == BSR_OP_DIV) {
      if (sp < 2) return false;
      uint64_t l, r;
      int sl, sr;
      as_atom(st[sp - 2], &l, &sl);
      as_atom(st[sp - 1], &r, &sr);
      if (op == BSR_OP_MUL && r < l) std::swap(l, r);
      --sp;
      set_atom(st[sp - 1], mix64(mix64(l ^ ((uint64_t)op << 56)) + 3 * r), (double)(sl * sr));
    } else if (op == BSR_OP_NEG) {
      if (sp < 1) return false;
      for (int q = 0; q < st[sp - 1].n; ++q) st[sp - 1].coef[q] = -st[sp - 1].coef[q];
    } else if (op == BSR_OP_LN) {
      if (sp < 1) return false;
      LinForm& f = st[sp - 1];
      const double a = t[i].a, b = t[i].b;
      if (!(std::isfinite(a) && std::isfinite(b))) {
        uint64_t h;
        int s;
        as_atom(f, &h, &s);
        uint64_t ab[2];
        memcpy(&ab[0], &a, 8);
        memcpy(&ab[1], &b, 8);
        set_atom(f, mix64(mix64(h + (s < 0)) ^ mix64(ab[0]) ^ (mix64(ab[1]) << 1)), 1.0);
        continue;
      }
      int m = 0;
      for (int q = 0; q < f.n; ++q) {
        const double c = f.coef[q] * a;
        if (c != 0.0) { f.atom[m] = f.atom[q]; f.coef[m] = c; ++m; }
      }
      f.n = m;
      if (b != 0.0) {
        LinForm k, r;
        set_atom(k, LF_CONST, b);
        add_forms(f, k, 1.0, &r);
        f = r;
      }
    } else if ((op >= 0 && op < BSR_OP_ADD) || op == BSR_OP_LOG) {
      if (sp < 1) return false;
      uint64_t h;
      int s;
      as_atom(st[sp - 1], &h, &s);
      const int rule = unary_sign_rule(op);
      const bool plain = st[sp - 1].n == 1 && (st[sp - 1].coef[0] == 1.0 || st[sp - 1].coef[0] == -1.0);
      if (op == BSR_OP_INV && plain) {
        int hit = -1;
        for (int q = 0; q < n_inv; ++q)
          if (inv_of[q] == h) hit = q;
        if (hit >= 0) {   // 1/(s * 1/B) = s B
          set_atom(st[sp - 1], inv_base[hit], (double)s);
          undone = true;
          continue;
        }
      }
      uint64_t x = mix64(h ^ ((uint64_t)(op + 1) << 48));
      if (op == BSR_OP_INV && plain && n_inv < INV_CAP) {
        inv_of[n_inv] = x;
        inv_base[n_inv] = h;
        ++n_inv;
      }
      if (rule == 2) { x = mix64(x + (s < 0 ? 0xA5A5ull : 0)); s = 1; }
      else if (rule == 0) s = 1;
      set_atom(st[sp - 1], x, (double)s);
    } else {
      return false;
    }
  }
  if (sp != 1) return false;
  *out = st[0];
  out->approx = undone;
  return true;
}

// f == +-g term by term (the column of one is the column of the other up to sign, bit for bit)
inline bool same_up_to_sign(const LinForm& f, const LinForm& g) {
  if (f.approx || g.approx) return false;
  if (f.n != g.n) return false;
  if (f.n == 0) return true;
  const double r = (f.coef[0] == g.coef[0]) ? 1.0 : -1.0;
  for (int i = 0; i < f.n; ++i)
    if (f.atom[i] != g.atom[i] || f.coef[i] != r * g.coef[i]) return false;
  return true;
}

// Echelon basis of a chain's K forms.
struct SpanBasis {
  static constexpr int VCAP = 56;   // terms a working vector holds (K <= 8 forms of <= 6 terms and a candidate's)
  struct Vec {                      // sorted by atom, no zero coefficient; inline storage: no allocation per candidate
    int n = 0;
    bool over = false;              // more terms than VCAP: the owner gives up (no claim)
    uint64_t a[VCAP];
    double c[VCAP];
    double get(uint64_t at) const {
      for (int i = 0; i < n; ++i)
        if (a[i] == at) return c[i];
      return 0.0;
    }
    double max_abs() const {
      double m = 0.0;
      for (int i = 0; i < n; ++i) m = std::max(m, std::fabs(c[i]));
      return m;
    }
    void from(const LinForm& f) {
      n = f.n;
      over = false;
      for (int i = 0; i < f.n; ++i) { a[i] = f.atom[i]; c[i] = f.coef[i]; }
    }
  };
  struct Row {
    uint64_t pivot;
    Vec t;   // the pivot's coefficient is 1, no other row holds it
  };
  std::vector<Row> rows;
  std::vector<LinForm> forms;   // the K forms themselves (same_up_to_sign against tree k)
  std::vector<char> known;      // form k is valid

  // v -= x * r; coefficients below 1e-12 of `scale` are treated as zero
  static void axpy(Vec& v, double x, const Vec& r, double scale) {
    Vec o;
    int i = 0, j = 0;
    while (i < v.n || j < r.n) {
      uint64_t at;
      double y;
      if (j >= r.n || (i < v.n && v.a[i] < r.a[j])) { at = v.a[i]; y = v.c[i]; ++i; }
      else if (i >= v.n || r.a[j] < v.a[i]) { at = r.a[j]; y = -x * r.c[j]; ++j; }
      else { at = v.a[i]; y = v.c[i] - x * r.c[j]; ++i; ++j; }
      if (!(std::fabs(y) > 1e-12 * scale)) continue;
      if (o.n == VCAP) { o.over = true; break; }
      o.a[o.n] = at;
      o.c[o.n] = y;
      ++o.n;
    }
    o.over = o.over || v.over || r.over;
    v = o;
  }
  void reduce(Vec& v, double scale) const {
    for (const Row& r : rows) {
      const double x = v.get(r.pivot);
      if (x != 0.0) axpy(v, x, r.t, scale);
    }
  }
  void add(const LinForm& f) {
    Vec v;
    v.from(f);
    const double scale = std::max(1.0, v.max_abs());
    reduce(v, scale);
    if (v.n == 0 || v.over) return;
    int best = 0;
    for (int i = 1; i < v.n; ++i)
      if (std::fabs(v.c[i]) > std::fabs(v.c[best])) best = i;
    const double pc = v.c[best];
    Row r;
    r.pivot = v.a[best];
    for (int i = 0; i < v.n; ++i) v.c[i] /= pc;
    v.c[best] = 1.0;
    r.t = v;
    for (Row& o : rows) {   // keep the basis fully reduced: the new pivot leaves the older rows
      const double x = o.t.get(r.pivot);
      if (x != 0.0) axpy(o.t, x, r.t, std::max(1.0, o.t.max_abs()));
    }
    rows.push_back(r);
  }
  void build(const std::vector<LinForm>& fs, const std::vector<char>& ok) {
    rows.clear();
    forms = fs;
    known = ok;
    for (size_t k = 0; k < fs.size(); ++k)
      if (ok[k]) add(fs[k]);
    for (const Row& r : rows)
      if (r.t.over) { rows.clear(); break; }   // a working vector overflowed: no claims from this basis
  }
  // the candidate's column is a linear combination of the chain's current columns (old column k included)
  bool in_span(const LinForm& f) const {
    Vec v;
    v.from(f);
    const double scale = v.max_abs();   // relative to the candidate's own coefficients (a form of tiny ones is not "zero")
    if (scale == 0.0) return true;      // the zero column
    reduce(v, scale);
    return !v.over && v.max_abs() <= 1e-9 * scale;
  }
};

}  // namespace bsr_span
