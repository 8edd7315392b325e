// Host-side structure analysis of tapes: is a candidate's column inside the span of a chain's K current columns?
//
// The solve step (k_solve) measures rho^2 = |s z|^2 - |Q^T s z|^2 in one pass; when the candidate's column lies
// (nearly) inside the span of the current columns that difference cancels and the proposal would go through the
// residual step -- two more kernels in the batch's queue.  Most such candidates are in the span BY CONSTRUCTION: the tree
// they would replace again (4.6 % of the real mix at K = 3), the same with a negation moved (1.1 %), and -- at K = 8,
// a dozen per batch -- linear combinations of current trees (`-x1` next to `x1 + x1`, `x1 + x6` next to `x1 + x1` and
// `-x6`, `cos(x3) + x6`, `x1 + -x1`).  For those w = 0 exactly is what the residual step would make of them (it
// measures |w|^2 ~ 1e-31 |s z|^2, far below its cut), so the host says so and k_solve skips the step -- provided its own
// one-pass figure agrees that the candidate is in the span.  A claim is therefore a statement about NUMBERS, not about
// algebra: it is made only where the column the tape computes equals the combination to a few roundings OF THE RESULT.
//
// Two descriptions of the column a tape computes travel up the tape together:
//  * its IDENTITY (sign, hash): structure-preserving, with the signs carried to the root -- neg A -> -A; A*B, A/B
//    multiply the signs; 1/A, A^3, sin A pass the sign on; A^2, cos A drop it; exp, log keep it inside; a sum pulls out
//    the sign of its first operand (operands of + and * in a fixed order: a + b == b + a bit for bit, and
//    -(a + b) == (-a) + (-b)); a x + b folds the sign into (a, b).  IEEE negation commutes with every rounded
//    operation involved and bsr_fastmath.h's sin is odd, its cos even: equal identities mean equal columns up to sign
//    BIT FOR BIT.  Association is part of the structure: (x1 + x2) + x3 and x1 + (x2 + x3) are different identities.
//  * its LINEAR FORM over atoms: sum of coef_i x atom_i, where +, -, neg and ln (a x + b) act on the coefficients and
//    every other operator makes an atom of its operand's identity.  The constant column is an atom of its own.  The
//    form stands for the column to rounding as long as nothing cancelled on the way: a sum in which a coefficient
//    drops to less than a quarter of its larger addend -- `(x2 + x1) + -x1` -- carries the rounding of the LARGE
//    intermediate, which against a small result is not rounding any more (features of very different magnitude: the
//    reference's rank gate sees full rank there, codes/funcs.py:1226).  Such a form is marked `inexact` and makes no
//    claim, neither as a candidate nor as a row of the basis; as an operand of a non-linear operator it is known by its
//    identity alone.  The exception that needs no flag: A - A of one and the same column is exactly zero.
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
constexpr double LF_CANCEL = 0.25;                      // a coefficient that falls below this share of its larger addend has cancelled
constexpr double LF_EPS = 64 * 2.220446049250313e-16;   // "zero" in the echelon arithmetic: a few roundings

inline uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
inline uint64_t bits_of(double v) {
  uint64_t b;
  memcpy(&b, &v, 8);
  return b;
}

struct LinForm {
  int n = 0;                 // terms, sorted by atom, no zero coefficient
  bool inexact = false;      // a term cancelled on the way: the form does not stand for the column to rounding (no claims)
  int ssign = 1;             // identity of the column: ssign x (the column `shash` names), bit for bit
  uint64_t shash = 0;
  uint64_t atom[LF_CAP];
  double coef[LF_CAP];
};

inline void set_atom(LinForm& f, uint64_t a, double c) {
  f.n = 1;
  f.atom[0] = a;
  f.coef[0] = c;
}
// out = a + sb * b (sb = +-1) on the forms (identity and flags are the caller's); *over: more than LF_CAP terms
inline void add_forms(const LinForm& a, const LinForm& b, double sb, LinForm* out, bool* over) {
  LinForm r;
  r.inexact = a.inexact || b.inexact;
  int i = 0, j = 0;
  *over = false;
  // A - A of one and the same column (one term each, the same atom, opposite coefficients) is exactly zero
  const bool same_col = a.n == 1 && b.n == 1 && a.atom[0] == b.atom[0] && a.coef[0] == -sb * b.coef[0] && !r.inexact;
  while (i < a.n || j < b.n) {
    uint64_t at;
    double c;
    if (j >= b.n || (i < a.n && a.atom[i] < b.atom[j])) { at = a.atom[i]; c = a.coef[i]; ++i; }
    else if (i >= a.n || b.atom[j] < a.atom[i]) { at = b.atom[j]; c = sb * b.coef[j]; ++j; }
    else {
      at = a.atom[i];
      const double ca = a.coef[i], cb = sb * b.coef[j];
      c = ca + cb;
      if (!same_col && !(std::fabs(c) >= LF_CANCEL * std::max(std::fabs(ca), std::fabs(cb)))) r.inexact = true;
      ++i;
      ++j;
    }
    if (c == 0.0) continue;
    if (r.n == LF_CAP) { *over = true; break; }
    r.atom[r.n] = at;
    r.coef[r.n] = c;
    ++r.n;
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

// linear form and identity of the column a postfix tape computes; false: malformed tape / deeper than max_stack
inline bool lin_form(const bsr_node* t, int len, LinForm* out, int max_stack = 26) {
  // (raw storage: an entry is written whole where it is pushed -- twenty-eight default-constructed forms were a fifth of
  // what a five-node tape's form cost)
  alignas(LinForm) unsigned char st_raw[28 * sizeof(LinForm)];
  LinForm* st = reinterpret_cast<LinForm*>(st_raw);
  if (max_stack > 26) max_stack = 26;
  int sp = 0;
  // 1/(1/A) is A to rounding (the moves stack an inverse on an inverse often enough: `1/[1/[x30]]`): the atoms made
  // by `inv` of a plain +-atom remember it, and an inverse on top of one gives it back -- in the FORM; the identity
  // keeps both inverses (the column is A to rounding, not bit for bit)
  constexpr int INV_CAP = 8;
  uint64_t inv_of[INV_CAP], inv_base[INV_CAP];
  int n_inv = 0;
  // the form of an entry that is known by its identity alone: one atom
  auto as_own_atom = [](LinForm& f) {
    set_atom(f, f.shash, (double)f.ssign);
    f.inexact = false;
  };
  for (int i = 0; i < len; ++i) {
    const int op = t[i].opcode;
    if (op == BSR_OP_TERMINAL) {
      if (sp >= max_stack) return false;
      LinForm& f = st[sp];
      f.inexact = false;
      f.shash = mix64(0x7465726Dull ^ ((uint64_t)(uint32_t)t[i].feature << 32));
      f.ssign = 1;
      as_own_atom(f);   // (n, atom[0], coef[0], inexact: with the three fields above, everything a form of one term holds)
      ++sp;
    } else if (op == BSR_OP_ADD || op == BSR_OP_SUB) {
      if (sp < 2) return false;
      const LinForm& a = st[sp - 2];
      const LinForm& b = st[sp - 1];
      const double sb = op == BSR_OP_SUB ? -1.0 : 1.0;
      LinForm r;
      bool over;
      add_forms(a, b, sb, &r, &over);
      // identity: x - y == x + (-y); operands in hash order, the first one's sign in front
      uint64_t h1 = a.shash, h2 = b.shash;
      int s1 = a.ssign, s2 = b.ssign * (sb < 0 ? -1 : 1);
      if (h2 < h1 || (h2 == h1 && s2 > s1)) { std::swap(h1, h2); std::swap(s1, s2); }
      r.ssign = s1;
      r.shash = mix64(mix64(h1 ^ 0x2B00000000000000ull) + 3 * h2 + (s1 * s2 < 0 ? 0x5A5Aull : 0));
      if (over) as_own_atom(r);
      --sp;
      st[sp - 1] = r;
    } else if (op == BSR_OP_MUL || op == BSR_OP_DIV) {
      if (sp < 2) return false;
      uint64_t l = st[sp - 2].shash, r = st[sp - 1].shash;
      const int s = st[sp - 2].ssign * st[sp - 1].ssign;
      if (op == BSR_OP_MUL && r < l) std::swap(l, r);
      --sp;
      LinForm& f = st[sp - 1];
      f.shash = mix64(mix64(l ^ ((uint64_t)op << 56)) + 3 * r);
      f.ssign = s;
      as_own_atom(f);
    } else if (op == BSR_OP_NEG) {
      if (sp < 1) return false;
      for (int q = 0; q < st[sp - 1].n; ++q) st[sp - 1].coef[q] = -st[sp - 1].coef[q];
      st[sp - 1].ssign = -st[sp - 1].ssign;
    } else if (op == BSR_OP_LN) {
      if (sp < 1) return false;
      LinForm& f = st[sp - 1];
      double a = t[i].a * (double)f.ssign, b = t[i].b;   // a (s X) + b == (a s) X + b bit for bit
      // identity: -(a X + b) == (-a) X + (-b): the sign of a (of b where a is zero) goes in front
      int s = 1;
      if (std::signbit(a) && !(a != a)) { s = -1; a = -a; b = -b; }
      f.shash = mix64(mix64(f.shash ^ 0x6C6E000000000000ull) ^ mix64(bits_of(a)) ^ (mix64(bits_of(b)) << 1));
      f.ssign = s;
      if (!(std::isfinite(t[i].a) && std::isfinite(t[i].b))) {
        as_own_atom(f);
        continue;
      }
      int m = 0;
      for (int q = 0; q < f.n; ++q) {
        const double c = f.coef[q] * t[i].a;
        if (c != 0.0) { f.atom[m] = f.atom[q]; f.coef[m] = c; ++m; }
        else if (t[i].a != 0.0) f.inexact = true;   // a product that underflowed: not what the column holds
      }
      f.n = m;
      if (t[i].b != 0.0) {
        LinForm k, r;
        bool over;
        set_atom(k, LF_CONST, t[i].b);
        add_forms(f, k, 1.0, &r, &over);
        r.shash = f.shash;
        r.ssign = f.ssign;
        if (over) as_own_atom(r);
        f = r;
      }
    } else if ((op >= 0 && op < BSR_OP_ADD) || op == BSR_OP_LOG) {
      if (sp < 1) return false;
      LinForm& f = st[sp - 1];
      const int rule = unary_sign_rule(op);
      // an inverse on the inverse of a plain +-atom: the form gets the atom back
      const bool plain = f.n == 1 && !f.inexact && (f.coef[0] == 1.0 || f.coef[0] == -1.0);
      int undo = -1;
      if (op == BSR_OP_INV && plain)
        for (int q = 0; q < n_inv; ++q)
          if (inv_of[q] == f.atom[0]) undo = q;
      const uint64_t base_atom = plain ? f.atom[0] : 0;
      const double base_coef = plain ? f.coef[0] : 0.0;
      int s = f.ssign;
      uint64_t x = mix64(f.shash ^ ((uint64_t)(op + 1) << 48));
      if (rule == 2) { x = mix64(x + (s < 0 ? 0xA5A5ull : 0)); s = 1; }
      else if (rule == 0) s = 1;
      f.shash = x;
      f.ssign = s;
      if (undo >= 0) {   // 1/(c * 1/B) = c B, c = +-1 (to rounding)
        set_atom(f, inv_base[undo], base_coef);
        f.inexact = false;
        continue;
      }
      as_own_atom(f);
      if (op == BSR_OP_INV && plain && n_inv < INV_CAP) {   // 1/(c A) = c (1/A): an inverse of THIS atom gives A back
        inv_of[n_inv] = x;
        inv_base[n_inv] = base_atom;
        ++n_inv;
      }
    } else {
      return false;
    }
  }
  if (sp != 1) return false;
  *out = st[0];
  return true;
}

// the column of one is the column of the other up to sign, bit for bit (equal identities)
inline bool same_up_to_sign(const LinForm& f, const LinForm& g) { return f.shash == g.shash; }

// features a tape reads, feature f at bit f mod 64
inline uint64_t feature_mask(const bsr_node* t, int len) {
  uint64_t m = 0;
  for (int i = 0; i < len; ++i)
    if (t[i].opcode == BSR_OP_TERMINAL) m |= 1ull << (t[i].feature & 63);
  return m;
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
  // features the chain's current trees read, feature f at bit f mod 64 (all ones: unknown).  A candidate that reads a
  // feature outside them is not looked at (mark_in_span): it could only be in the span through an exact cancellation of
  // that feature (x5 - x5 + x1), which costs it a residual pass it did not need -- and nothing else.
  uint64_t feat_mask = ~0ull;
  std::vector<LinForm> forms;   // the K forms themselves (same_up_to_sign against tree k)
  std::vector<char> known;      // form k is valid

  // v -= x * r; coefficients within a few roundings of `scale` are treated as zero
  static void axpy(Vec& v, double x, const Vec& r, double scale) {
    Vec o;
    int i = 0, j = 0;
    while (i < v.n || j < r.n) {
      uint64_t at;
      double y;
      if (j >= r.n || (i < v.n && v.a[i] < r.a[j])) { at = v.a[i]; y = v.c[i]; ++i; }
      else if (i >= v.n || r.a[j] < v.a[i]) { at = r.a[j]; y = -x * r.c[j]; ++j; }
      else { at = v.a[i]; y = v.c[i] - x * r.c[j]; ++i; ++j; }
      if (!(std::fabs(y) > 0.5 * LF_EPS * scale)) continue;
      if (o.n == VCAP) { o.over = true; break; }
      o.a[o.n] = at;
      o.c[o.n] = y;
      ++o.n;
    }
    o.over = o.over || v.over || r.over;
    v.n = o.n;                       // (its o.n terms, not the whole 900-byte vector)
    v.over = o.over;
    for (int q = 0; q < o.n; ++q) { v.a[q] = o.a[q]; v.c[q] = o.c[q]; }
  }
  void reduce(Vec& v, double scale) const {
    for (const Row& r : rows) {
      const double x = v.get(r.pivot);
      if (x != 0.0) axpy(v, x, r.t, scale);
    }
  }
  // A form becomes a row only where what the older rows leave of it is a substantial part of it (an eighth of its
  // largest coefficient): a row that is nearly a combination of the others would let a candidate be "in the span" with
  // large coefficients of nearly cancelling columns -- to the algebra, not to the numbers.
  void add(const LinForm& f) {
    Vec v;
    v.from(f);
    const double scale = v.max_abs();
    if (!(scale > 0.0)) return;
    reduce(v, scale);
    if (v.n == 0 || v.over || !(v.max_abs() >= 0.125 * scale)) return;
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
      if (ok[k] && !fs[k].inexact) add(fs[k]);   // (a form in which a term cancelled is no statement about its column)
    for (const Row& r : rows)
      if (r.t.over) { rows.clear(); break; }   // a working vector overflowed: no claims from this basis
  }
  // the candidate's column is a linear combination of the chain's current columns (old column k included)
  // -- to rounding: what is left of the form after the reduction is within a few roundings of its own coefficients
  // (the reference's rank tolerance is max(N, K) eps of the largest singular value, codes/funcs.py:1226; a claim must
  // hold for every N)
  bool in_span(const LinForm& f) const {
    if (f.inexact) return false;
    Vec v;
    v.from(f);
    const double scale = v.max_abs();   // relative to the candidate's own coefficients (a form of tiny ones is not "zero")
    if (scale == 0.0) return true;      // the zero column (A - A of one column: exact)
    reduce(v, scale);
    return !v.over && v.max_abs() <= LF_EPS * scale;
  }
};

}  // namespace bsr_span
