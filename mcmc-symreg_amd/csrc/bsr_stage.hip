// Staging of a batch on the host: tape validation, the compact streams (fusions, operand order, derived columns, cost
// model), the tile pass's tape groups / LDS slot maps / schedule / tape records.  No HIP calls beyond buffer growth.
#include "bsr_ctx.h"

bool is_binary_op(int op) {
  return op == BSR_OP_ADD || op == BSR_OP_MUL || op == BSR_OP_SUB || op == BSR_OP_DIV;
}

int check_tape(bsr_ctx* c, const bsr_node* t, int len, int* max_sp) {
  if (len <= 0) return fail(c, BSR_E_TAPE, "empty tape");
  if (len > BSR_MAX_TAPE) return fail(c, BSR_E_TOOBIG, "tape longer than BSR_MAX_TAPE");
  int sp = 0, mx = 0;
  for (int i = 0; i < len; ++i) {
    const int op = t[i].opcode;
    if (op == BSR_OP_TERMINAL) {
      if (t[i].feature < 0 || t[i].feature >= c->d) return fail(c, BSR_E_TAPE, "terminal feature out of range");
      ++sp;
    } else if ((op >= 0 && op < BSR_OP_ADD) || op == BSR_OP_LOG) {
      if (sp < 1) return fail(c, BSR_E_TAPE, "unary operator on empty stack");
    } else if (is_binary_op(op)) {
      if (sp < 2) return fail(c, BSR_E_TAPE, "binary operator needs two operands");
      --sp;
    } else {
      return fail(c, BSR_E_TAPE, "unknown opcode");
    }
    mx = std::max(mx, sp);
  }
  if (sp != 1) return fail(c, BSR_E_TAPE, "tape does not reduce to one value");
  if (mx > BSR_MAX_STACK) return fail(c, BSR_E_TOOBIG, "tape needs a deeper stack than BSR_MAX_STACK");
  *max_sp = mx;
  return BSR_OK;
}

// Operand order of the commutative operators in the streams.  A bare terminal (or a `terminal, unary op` pair that
// became a derived column) as the SECOND operand of + or * fuses with the operator into one stream entry
// (BSR_SOP_ADD_T / BSR_SOP_MUL_T) and needs no stack slot; as the first operand, with anything else second, it is
// pushed and popped.  So where exactly the first operand is such a terminal the two subtrees trade places
// (x3 + ln(x1) is written `x1 ln x3 +`): a + b == b + a and a * b == b * a bit for bit, the value of every row is
// unchanged, and almost every tape of the real move mix becomes a chain (bsr_device.h: chain_eval).
// `admitted(j)`: nodes j, j+1 are a `terminal, unary op` pair whose derived column this batch uses.
// Writes the reordered tape to `out` (may alias nothing of `t`) and returns true, or returns false: order kept.
template <typename Admitted>
static bool reorder_tape(const bsr_node* t, int len, bsr_node* out, std::vector<int32_t>& kid, std::vector<int32_t>& stk,
                         const Admitted& admitted) {
  if (len < 4) return false;   // the shortest tape with something to swap: T, T, op, +
  // pass 1: children of every node, "is one terminal entry" per subtree; is there anything to swap at all?
  kid.resize((size_t)len * 2);
  stk.clear();
  bool any = false;
  // kid[2j], kid[2j+1]: roots of the left / right subtree (-1: none); termlike(j): the subtree is one terminal entry
  auto termlike = [&](int j) {
    if (t[j].opcode == BSR_OP_TERMINAL) return true;
    return j > 0 && t[j - 1].opcode == BSR_OP_TERMINAL && kid[2 * j] == j - 1 && kid[2 * j + 1] < 0 && admitted(j - 1);
  };
  for (int j = 0; j < len; ++j) {
    const int op = t[j].opcode;
    kid[2 * j] = kid[2 * j + 1] = -1;
    if (op == BSR_OP_TERMINAL) {
      stk.push_back(j);
    } else if (op == BSR_OP_ADD || op == BSR_OP_MUL || op == BSR_OP_SUB || op == BSR_OP_DIV) {
      if (stk.size() < 2) return false;
      const int r = stk.back();
      stk.pop_back();
      const int l = stk.back();
      kid[2 * j] = l;
      kid[2 * j + 1] = r;
      stk.back() = j;
      if ((op == BSR_OP_ADD || op == BSR_OP_MUL) && termlike(l) && !termlike(r)) any = true;
    } else {
      if (stk.empty()) return false;
      kid[2 * j] = stk.back();
      stk.back() = j;
    }
  }
  if (!any || stk.size() != 1) return false;
  // pass 2: post-order walk from the root with the swapped child order (explicit stack; entry = node, or ~node once its
  // children have been pushed)
  int n_out = 0;
  stk.clear();
  stk.push_back(len - 1);
  while (!stk.empty()) {
    const int e = stk.back();
    stk.pop_back();
    if (e < 0) {
      out[n_out++] = t[~e];
      continue;
    }
    const int l = kid[2 * e], r = kid[2 * e + 1];
    if (l < 0) {
      out[n_out++] = t[e];
      continue;
    }
    stk.push_back(~e);
    if (r < 0) {
      stk.push_back(l);
    } else {
      const int op = t[e].opcode;
      const bool swap = (op == BSR_OP_ADD || op == BSR_OP_MUL) && termlike(l) && !termlike(r);
      // popped first = evaluated first
      if (swap) { stk.push_back(l); stk.push_back(r); }
      else { stk.push_back(r); stk.push_back(l); }
    }
  }
  return n_out == len;
}

// Tapes by cost, heaviest first, equal costs in batch order: one key per tape, sorted in place (std::stable_sort
// allocates a buffer on every call: two of them were a tenth of the caller's time per batch).
// Validates the tapes, chooses LDS staging, and writes the compact streams the interpreter reads into the slot's
// pinned input block:
//   opcode stream  : 4 bits per entry, 16 per 64-bit word, one padding word per tape; an entry is a tape node, or a
//                    `terminal, +|*` pair fused into BSR_SOP_ADD_T / BSR_SOP_MUL_T
//   column stream  : 16 bits per terminal in tape order (LDS slot when staging, else the X column), 4 per word,
//                    padded with a valid id so the kernel may request one terminal past the end
//   ln stream      : (a,b) per ln node in tape order plus one padding pair
int stage_tapes(bsr_ctx* c, BatchSlot& s, const bsr_node* rows, const int32_t* tape_off, int n,
                std::vector<TapeLoc>* loc, int tile_chains) {
  if (!rows || !tape_off || n <= 0) return fail(c, BSR_E_ARG, "null tapes / empty batch");
  if (n > c->max_batch) return fail(c, BSR_E_TOOBIG, "batch larger than max_batch");
  if (tape_off[0] != 0) return fail(c, BSR_E_ARG, "tape_off[0] must be 0");
  loc->resize(n);
  s.order_n = -1;
  size_t cw = 0, fw = 0, lw = 0;
  int max_fused_sp = 0;
  std::fill(s.slot_of.begin(), s.slot_of.end(), -1);
  // How many derived columns this batch may use.  A context that scores through the tile pass must keep the batch's
  // columns inside LDS: the allowance is what the slice leaves after y, the chains' bases and every X column the tapes
  // name (an upper bound of what stays in use).  Derived columns are admitted in tape order until it is spent; the
  // values do not depend on which are (same routine either way), so neither do the results.
  // (not with the LDS-staging variant of k_rows, BSR_NO_LDS=0: there the column count picks the row-block size, and the
  // block size fixes the order of the sums)
  const bool derive = c->derived_ready && c->n_cols > c->d && c->no_lds;
  int allowance = derive ? c->n_cols : 0;
  if (derive && c->tile_on && tile_chains > 0) {
    int n_base = 0;
    if (c->tile_whole || c->tile_stream) {   // (the chunked k_tile: the allowance does not depend on the batch's base columns)
      for (int j = 0; j < tape_off[n]; ++j)
        if (rows[j].opcode == BSR_OP_TERMINAL && rows[j].feature >= 0 && rows[j].feature < c->d &&
            s.slot_of[rows[j].feature] < 0) {
          s.slot_of[rows[j].feature] = 0;
          ++n_base;
        }
      std::fill(s.slot_of.begin(), s.slot_of.begin() + c->d, -1);
    }
    const long fixed = (long)n_base + 1 + (long)((c->tile_by_chain && c->tile_T > 1) ? (tile_chains + c->tile_T - 1) / c->tile_T : tile_chains) * c->K;
    size_t fit = (tile_lds_bytes_max() - 1024) / ((size_t)std::max(1, c->tile_bps + (c->tile_long > 0 ? 1 : 0)) * BSR_TILE_BLOCK * c->esz);
    long room = (long)fit - fixed;
    if (!c->tile_whole) room = c->derived_max;   // chunked: a group's columns set the chunk length, not whether the batch fits
    // ... but the streaming kernel's waves copy at most BSR_STREAM_UNITS_MAX columns per chunk: a batch whose tapes name
    // nearly every feature (deep trees at d = 50: 54 base columns) must not be pushed over that by its derived columns --
    // it fell back to the chunked k_tile at four times the streaming kernel's time (round 6: that, not the stack machine,
    // was most of config 5's "deep tree" cliff)
    if (!c->tile_whole && c->tile_stream) room = std::min<long>(room, std::max<long>(0, (long)BSR_STREAM_UNITS_MAX - fixed));
    // (streaming contexts too: trading derived columns for a deeper LDS ring -- 3 buffers instead of 2 in the batches
    // of 49+ columns -- measured slower, 95.8 against 91.6 us: a derived column saves more than the ring's depth)
    // no room at all: the batch would not take the tile pass anyway (k_rows reads columns from L2: no limit there)
    if (room >= 0) allowance = (int)std::min<long>(room, c->n_cols);
    allowance = std::min(allowance, c->derived_max);
  }
  s.derived_used = 0;
  // Which derived columns: the ones that save the most.  One pass over the batch adds up, per (op, feature), the
  // interpreter cost of the op on every `terminal f, op` it would replace; the `allowance` best are admitted (ties: the
  // lower column), a single cheap use (neg, square) is not worth a column.  -2 marks an admitted column until the main
  // pass below gives it its place.
  if (derive && allowance > 0) {
    std::vector<std::pair<int, int>>& cand = s.derived_cand;   // (benefit, column)
    cand.clear();
    std::vector<int>& ben = s.derived_benefit;
    if ((int)ben.size() < c->n_cols) ben.assign(c->n_cols, 0);
    for (int j = 0; j + 1 < tape_off[n]; ++j) {
      if (rows[j].opcode != BSR_OP_TERMINAL || rows[j].feature < 0 || rows[j].feature >= c->d) continue;
      const int op = rows[j + 1].opcode, m = derived_index(op);
      if (m < 0) continue;
      const int dc = c->d * (1 + m) + rows[j].feature;
      const int w = (op == BSR_OP_SIN || op == BSR_OP_COS) ? 77 : (op == BSR_OP_EXP) ? 59 : (op == BSR_OP_LOG) ? 90
                    : (op == BSR_OP_INV) ? 35 : (op == BSR_OP_CUBIC) ? 23 : 3;   // the operator's cost (stage_tapes' cost model)
      if (ben[dc] == 0) cand.push_back({0, dc});
      ben[dc] += w;
    }
    // (a `terminal, op` pair that straddles two tapes cannot occur: a tape never ends in a terminal unless it is one)
    for (auto& cd : cand) {
      cd.first = ben[cd.second];
      ben[cd.second] = 0;
    }
    std::sort(cand.begin(), cand.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) {
      return a.first != b.first ? a.first > b.first : a.second < b.second;
    });
    // worth a column: one use of cubic or anything dearer where the slice is staged from L2; where it streams from HBM
    // (8 bytes per row and group that uses it), an operator of 17 instructions or more per value
    const int worth = c->tile_whole ? 15 : 35;
    for (size_t q = 0; q < cand.size() && (int)q < allowance; ++q)
      if (cand[q].first >= worth) s.slot_of[cand[q].second] = -2;
  }
  // Pass 1: validation, sizes, the columns in use.  A tape whose fused encoding still pushes more than one terminal
  // (it is not a chain) is tried once more with its commutative operands in fusing order (reorder_tape): few tapes
  // get there, so the reordering costs the batch next to nothing.
  if ((int)s.tape_src.size() < n) s.tape_src.resize((size_t)n);
  for (int i = 0; i < n; ++i) {
    const int len = tape_off[i + 1] - tape_off[i];
    TapeLoc& L = (*loc)[i];
    const bsr_node* t = rows + tape_off[i];
    int rc = check_tape(c, t, len, &L.max_sp);
    if (rc != BSR_OK) return rc;
    int nt = 0, nl = 0, fmx = 0, n_push = 0;
    auto scan = [&](const bsr_node* tp) {
      nt = nl = fmx = n_push = 0;
      int fsp = 0;  // stack depth with `terminal, +|*` pairs fused (what the kernels run)
      for (int j = 0; j < len; ++j) {
        if (tp[j].opcode == BSR_OP_TERMINAL) {
          ++nt;
          int col = tp[j].feature;
          if (derive && j + 1 < len) {   // `terminal, unary op` -> the op's derived column
            const int m = derived_index(tp[j + 1].opcode);
            if (m >= 0) {
              const int dc = c->d * (1 + m) + col;
              if (s.slot_of[dc] == -2 || s.slot_of[dc] == 0) {   // admitted above
                if (s.slot_of[dc] == -2) ++s.derived_used;
                col = dc;
                ++j;
              }
            }
          }
          s.slot_of[col] = 0;
          const int nxt = (j + 1 < len) ? tp[j + 1].opcode : -1;
          if (nt > 1 && (nxt == BSR_OP_ADD || nxt == BSR_OP_MUL)) ++j; else { ++fsp; ++n_push; }
        } else if (tp[j].opcode == BSR_OP_LN) {
          ++nl;
        } else if (is_binary_op(tp[j].opcode)) {
          --fsp;
        }
        fmx = std::max(fmx, fsp);
      }
    };
    scan(t);
    s.tape_src[i] = t;
    if (c->reorder && n_push > 1 && len >= 4) {
      auto admitted = [&](int j) {   // nodes j, j+1: `terminal, unary op` served by a derived column of this batch
        if (!derive || j + 1 >= len) return false;
        const int m = derived_index(t[j + 1].opcode), f = t[j].feature;
        if (m < 0 || f < 0 || f >= c->d) return false;
        const int sl = s.slot_of[c->d * (1 + m) + f];
        return sl == -2 || sl == 0;
      };
      if ((int)s.rows_perm.size() < tape_off[n]) {
        // (pointers into the copy handed out for earlier tapes must survive: size it once for the whole batch)
        std::vector<bsr_node> grown((size_t)tape_off[n]);
        for (int q = 0; q < i; ++q)
          if (s.tape_src[q] != rows + tape_off[q]) {
            memcpy(grown.data() + tape_off[q], s.tape_src[q], (size_t)(tape_off[q + 1] - tape_off[q]) * sizeof(bsr_node));
            s.tape_src[q] = grown.data() + tape_off[q];
          }
        s.rows_perm.swap(grown);
      }
      if (reorder_tape(t, len, s.rows_perm.data() + tape_off[i], s.perm_kid, s.perm_stack, admitted)) {
        s.tape_src[i] = s.rows_perm.data() + tape_off[i];
        scan(s.tape_src[i]);
      }
    }
    max_fused_sp = std::max(max_fused_sp, fmx);
    L.n_nodes = len;
    L.code_off = (int)cw;
    L.feat_off = (int)fw;
    L.ln_off = (int)lw;
    cw += (size_t)(len + 15) / 16 + 1;
    fw += (size_t)(nt + 1 + 3) / 4 + 1;
    lw += (size_t)nl + 1;
  }
  const size_t rec_words = (c->tile_sched_cap * sizeof(TapeRec) + (size_t)(1 + c->tile_T) * (c->max_batch + 1) * sizeof(int32_t) +
                            (c->tile_stream ? (c->tile_sched_cap + 2) * sizeof(StreamRec) + (c->max_batch + 2) * (size_t)BSR_STREAM_EXT_MAX * 16 : 0) +
                            (c->tile_asm ? ((size_t)c->tile_T * (c->max_batch + 1) + 2) * sizeof(TileProg) : 0)) / 8 + 32;
  int rc = ensure_input(c, s, cw + 2 * fw + 2 * lw + rec_words);
  if (rc != BSR_OK) return rc;
  s.off_recs = (s.off_streams + (cw + 2 * fw + 2 * lw) * 8 + 127) / 128 * 128;
  s.recs_bytes = 0;
  // columns of X referenced by this batch -> LDS slots (ascending feature order)
  s.nF = 0;
  int32_t* hfeat = s.h_feat();
  for (int f = 0; f < c->n_cols; ++f)
    if (s.slot_of[f] == 0) {
      s.slot_of[f] = s.nF;
      hfeat[s.nF++] = f;
    }
  const size_t lds_budget = 64 * 1024;
  s.use_lds = false;
  s.rb_rows = c->rb_rows;
  if (!c->no_lds) {
    for (int rb = c->rb_rows; rb >= 64 * c->rows_per_lane && rb >= 256; rb >>= 1) {
      if ((size_t)(s.nF + 1) * rb * c->esz <= lds_budget) {
        s.use_lds = true;
        s.rb_rows = rb;
        break;
      }
    }
  }
  s.tile = false;
  s.tile_chains = tile_chains;
  s.code_words = cw;
  s.feat_words = fw;
  s.ln_words = 2 * lw;
  uint64_t* hc = s.h_streams();
  uint64_t* hf = hc + cw;
  double* hl = reinterpret_cast<double*>(hf + fw);
  memset(hc, 0, (cw + fw) * 8);
  for (int i = 0; i < n; ++i) {
    const TapeLoc L = (*loc)[i];
    uint64_t* pc = hc + L.code_off;
    uint64_t* pf = hf + L.feat_off;
    double* pl = hl + 2 * (size_t)L.ln_off;
    const bsr_node* tsrc = s.tape_src[i];
    // (the cost model's unit differs by row pass: see the two tables below)
    const bool stream_costs = c->tile_stream != 0;
    int nt = 0, nl = 0, ns = 0, sp = 0, mx = 0, cost = stream_costs ? 73 : 23;
    int n_push = 0, n_stack_ops = 0;   // chain tape: one push (the leading terminal), no operator that pops
    uint8_t ss_codes[64];
    int ss_n = 0;
    for (int j = 0; j < L.n_nodes; ++j) {
      const bsr_node& r = tsrc[j];
      int code = r.opcode & 15;
      if (r.opcode == BSR_OP_TERMINAL) {
        int col = r.feature;
        if (derive && j + 1 < L.n_nodes) {   // the column pass 1 admitted for `terminal, unary op` (slot assigned)
          const int m = derived_index(tsrc[j + 1].opcode);
          if (m >= 0 && s.slot_of[c->d * (1 + m) + col] >= 0) {
            col = c->d * (1 + m) + col;
            ++j;
          }
        }
        const uint64_t id = (uint64_t)(s.use_lds ? s.slot_of[col] : col);
        pf[nt >> 2] |= id << (16 * (nt & 3));
        ++nt;
        // a terminal consumed at once by + or * (the lighter child in the tape's heavy-child-first order) becomes
        // one stream entry: acc = acc op X[:,f], no push/pop
        const int nxt = (j + 1 < L.n_nodes) ? tsrc[j + 1].opcode : -1;
        if (ns > 0 && (nxt == BSR_OP_ADD || nxt == BSR_OP_MUL)) {
          code = (nxt == BSR_OP_ADD) ? BSR_SOP_ADD_T : BSR_SOP_MUL_T;
          ++j;
        } else {
          ++sp;
          ++n_push;
        }
      } else if (r.opcode == BSR_OP_LN) {
        pl[2 * nl] = r.a;
        pl[2 * nl + 1] = r.b;
        ++nl;
      } else if (is_binary_op(r.opcode)) {
        --sp;
        ++n_stack_ops;
      }
      mx = std::max(mx, sp);
      pc[ns >> 4] |= (uint64_t)code << (4 * (ns & 15));
      ++ns;
      if (g_stream_stats) ss_codes[ss_n < 64 ? ss_n++ : 63] = (uint8_t)code;
      // vector instructions per 64 rows of the tile pass's chain evaluator, in halves (ISA listing of k_tile1 / k_tile):
      // a fused terminal 1.5 (its reads are LDS work), neg / square 1.5, ln 2.5, cubic 11.5, inv 17.5, exp 29.5,
      // sin / cos 38.5 (31 of arithmetic, the huge-argument test, the call's register moves), log 45, a pushed terminal or
      // a popping operator 3 (operand copies of the stack machine); each includes ~0.5 for its decode.  The base 11.5
      // is the projection sums (7), the pass's set-up and the block's share of the lane reduction.
      if (!stream_costs) {
        cost += (code == BSR_SOP_ADD_T || code == BSR_SOP_MUL_T) ? 3
                : (r.opcode == BSR_OP_TERMINAL) ? (ns > 1 ? 6 : 0)
                : (r.opcode == BSR_OP_SIN || r.opcode == BSR_OP_COS) ? 77
                : (r.opcode == BSR_OP_EXP) ? 59 : (r.opcode == BSR_OP_LOG) ? 90
                : (r.opcode == BSR_OP_INV || r.opcode == BSR_OP_DIV) ? 35 : (r.opcode == BSR_OP_CUBIC) ? 23
                : (r.opcode == BSR_OP_LN) ? 5 : is_binary_op(r.opcode) ? 6 : 3;
      } else {
        // The streaming pass (bsr_stream.hip, its assembly interpreter): a wave's time, not its vector instructions --
        // half-microseconds per tape over a 30-block slice, from batches of one shape (tools/probes/op_costs, round 4):
        // a leaf 36.5 us of fetching the program, reading y and the basis and adding up (the base, 73); a fused
        // terminal +2.0, neg / square +2.5, ln +3.0, a pushed terminal or a popping operator +4, cubic +6, inv +11,
        // exp +14, cos +19, sin +20.  With the tile pass's table (an operator 3 to 77 against a base of 23) the
        // schedule gave tapes with a sin a wave almost to themselves: busiest wave / mean wave 1.50 measured.
        cost += (code == BSR_SOP_ADD_T || code == BSR_SOP_MUL_T) ? 4
                : (r.opcode == BSR_OP_TERMINAL) ? (ns > 1 ? 8 : 0)
                : (r.opcode == BSR_OP_SIN) ? 40 : (r.opcode == BSR_OP_COS) ? 38
                : (r.opcode == BSR_OP_EXP) ? 28 : (r.opcode == BSR_OP_LOG) ? 400   // (log: the stack machine, out of line)
                : (r.opcode == BSR_OP_INV || r.opcode == BSR_OP_DIV) ? 22 : (r.opcode == BSR_OP_CUBIC) ? 12
                : (r.opcode == BSR_OP_LN) ? 6 : is_binary_op(r.opcode) ? 8 : 5;
      }
    }
    // (a tape the streaming pass hands to the stack machine: a call, scratch-indexed stack slots -- four times a chain's time.
    // Round 6: the chunk block of assembly also takes programs of several words, up to eight ln nodes and, at K <= 3, a
    // second value below the accumulator -- build_tile_launch decides per tape; the cost estimate follows the same limits)
    {
      const bool classic = ns <= 16 && nt <= 8 && nl <= 3 && mx <= 2;
      const bool deep = c->stream_deep > 0 && c->K <= 4 && ns <= 16 * (1 + BSR_STREAM_EXT_MAX) && nl <= BSR_STREAM_LN_PAIRS &&
                        mx <= ((c->stream_deep >= 2 && c->K <= 3) ? 3 : 2);
      if (stream_costs && !classic && !deep) cost *= 4;
    }
    (*loc)[i].n_stream = ns;
    (*loc)[i].nt = nt;
    (*loc)[i].nl = nl;
    (*loc)[i].grp = 0;
    (*loc)[i].cost = cost;
    (*loc)[i].max_sp = mx;
    (*loc)[i].acc_only = (c->chain_eval && n_push == 1 && n_stack_ops == 0) ? 1 : 0;
    if (g_stream_stats) {
      const int kind = (*loc)[i].acc_only ? 0 : 1;
      g_ss_tapes[kind].fetch_add(1, std::memory_order_relaxed);
      for (int q = 0; q < ss_n; ++q) g_ss_entries[kind][ss_codes[q]].fetch_add(1, std::memory_order_relaxed);
    }
    pl[2 * nl] = 1.0;
    pl[2 * nl + 1] = 0.0;
    if (!s.use_lds) {  // padding ids must name a valid column: repeat the first terminal's
      const uint64_t id0 = pf[0] & 0xFFFFu;
      const int words = (nt + 1 + 3) / 4 + 1;
      for (int t = nt; t < words * 4; ++t) pf[t >> 2] |= id0 << (16 * (t & 3));
    }
  }
  s.tile_possible = c->tile_on && c->tile_ever && tile_chains > 0 && max_fused_sp - 1 <= BSR_REG_STACK && !s.use_lds;
  if (g_stream_stats) {
    g_ss_batches.fetch_add(1, std::memory_order_relaxed);
    g_ss_derived.fetch_add(s.derived_used, std::memory_order_relaxed);
    g_ss_cols.fetch_add(s.nF, std::memory_order_relaxed);
  }
  return BSR_OK;
}

// Second half of staging a scoring batch for the tile pass: the tape groups, their LDS slot maps and column-pointer
// tables, and the column stream in LDS slots; the descriptors learn their group and the slot of their chain's basis.
// It needs nothing from the caller but what stage_tapes left in the slot, so it runs wherever the batch's launches
// are issued -- on a submission thread where the context has one: 1.5 us off the caller's path per batch.
void stage_tile(bsr_ctx* c, BatchSlot& s, int n) {
  std::vector<TapeLoc>& loc = s.loc_tmp;
  const int tile_chains = s.tile_chains;
  uint64_t* hc = s.h_streams();
  uint64_t* hf = hc + s.code_words;
  double* hl = reinterpret_cast<double*>(hf + s.feat_words);
  uint64_t* hf2 = reinterpret_cast<uint64_t*>(hl + s.ln_words);  // the column stream in LDS slots
  const size_t fw = s.feat_words;
  s.tile = false;
  if (!s.tile_possible || (int)loc.size() < n) return;
  // ---- tile pass: tape groups, their LDS slot maps and column-pointer tables, the column stream in LDS slots.
  // No tape may need more value-stack slots than the register stack holds.
  {
    const int T = c->tile_T, K = c->K;
    const int group_chains = (c->tile_by_chain && T > 1) ? (tile_chains + T - 1) / T : tile_chains;   // basis column sets a group stages
    const int cap = (n + T - 1) / T;   // tapes per group at most (keeps the groups' passes even)
    // tapes by cost, heaviest first (stable: equal costs keep batch order)
    cost_order(s.order_tmp, s.order_keys, n, [&](int i) { return loc[i].cost; });
    s.order_n = n;
    if ((int)s.grp_slot.size() < T * c->n_cols) s.grp_slot.resize((size_t)T * c->n_cols);
    std::fill(s.grp_slot.begin(), s.grp_slot.begin() + (size_t)T * c->n_cols, (int16_t)-1);
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ncol[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long total_cost = 0;
    for (int i = 0; i < n; ++i) total_cost += loc[i].cost;
    const long share = total_cost / T + total_cost / (8 * T) + 1;   // an even share of the batch's cost and an eighth
    auto tape_cols = [&](int i, auto&& fn) {   // the columns tape i reads, in stream order
      const uint64_t* pf = hf + loc[i].feat_off;
      for (int t = 0; t < loc[i].nt; ++t) fn((int)((pf[t >> 2] >> (16 * (t & 3))) & 0xFFFFu));
    };
    for (int oi = 0; oi < n; ++oi) {
      const int i = s.order_tmp[oi];
      int g = 0;
      if (T > 1 && c->tile_by_chain) {
        g = s.chain_slot[s.h_desc()[i].ck] % T;   // chain groups: the batch's chains dealt round robin
      } else if (T > 1 && c->tile_whole) {
        // the slice sits in LDS whole: columns are cheap, balance the cost -- dealt back and forth (0 1 1 0 ...), so that
        // no group gets the heavier tape of every round (LPT inside the group follows)
        const int r = oi / T, k = oi - r * T;
        g = (r & 1) ? T - 1 - k : k;
      } else if (T > 1) {
        // chunked: every column of a group costs LDS (shorter chunks) and HBM traffic (the group streams it over all
        // rows), and the launch ends with its heaviest group: among the groups this tape does not lift above an even
        // share of the cost, the one that needs the fewest new columns for it; failing that, the lightest
        int best_new = 1 << 30;
        g = -1;
        for (int gi = 0; gi < T; ++gi) {
          if (cnt[gi] >= cap || load[gi] + loc[i].cost > share) continue;
          int n_new = 0;
          tape_cols(i, [&](int col) { if (s.grp_slot[(size_t)gi * c->n_cols + col] < 0) ++n_new; });
          if (g < 0 || n_new < best_new || (n_new == best_new && load[gi] < load[g])) { best_new = n_new; g = gi; }
        }
        if (g < 0) {
          for (int gi = 0; gi < T; ++gi)
            if (cnt[gi] < cap && (g < 0 || load[gi] < load[g])) g = gi;
        }
      }
      loc[i].grp = g;
      ++cnt[g];
      load[g] += loc[i].cost;
      tape_cols(i, [&](int col) {
        int16_t& sl = s.grp_slot[(size_t)g * c->n_cols + col];
        if (sl < 0) { sl = 0; ++ncol[g]; }
      });
    }
    // slots in ascending column order per group; the group's table: X columns, y, every chain's basis
    const void** hcols = s.h_cols();
    int max_ncols = 0;
    for (int g = 0; g < T; ++g) {
      int nf = 0;
      for (int col = 0; col < c->n_cols; ++col) {
        int16_t& sl = s.grp_slot[(size_t)g * c->n_cols + col];
        if (sl < 0) continue;
        sl = (int16_t)nf;
        hcols[(size_t)g * s.cols_stride + nf] = col_ptr(c, c->Xt, col);
        ++nf;
      }
      s.grp_nF[g] = nf;
      hcols[(size_t)g * s.cols_stride + nf] = c->y;
      if (c->tile_by_chain && T > 1) {
        // the group's own chains (batch chain ci belongs to group ci % T, as its (ci / T)-th); the places of a group
        // that holds fewer chains than the fullest repeat y: staged, never read
        for (int gc = 0; gc < group_chains; ++gc) {
          const size_t ci = (size_t)gc * T + g;
          for (int k = 0; k < K; ++k)
            hcols[(size_t)g * s.cols_stride + nf + 1 + gc * K + k] =
                ci < s.batch_chains.size() ? col_ptr(c, c->Q, (int64_t)s.batch_chains[ci] * K + k) : (const void*)c->y;
        }
      } else {
        for (size_t ci = 0; ci < s.batch_chains.size(); ++ci)
          for (int k = 0; k < K; ++k)
            hcols[(size_t)g * s.cols_stride + nf + 1 + ci * K + k] = col_ptr(c, c->Q, (int64_t)s.batch_chains[ci] * K + k);
      }
      max_ncols = std::max(max_ncols, nf + 1 + group_chains * K);
    }
    s.tile_group_chains = group_chains;
    // the whole slice in LDS at once, or chunks through two buffers (f32: one, staged through registers): as many
    // blocks as fit, a whole number of chain passes where there is room for one
    const size_t budget = tile_lds_bytes_max() - 1024;
    const size_t per_block = (size_t)max_ncols * BSR_TILE_BLOCK * c->esz;
    int chunk = 0, ring = 1;
    static const int force_chunk = env_int("BSR_TILE_CHUNK", 0);   // test hooks: chunks of at most this many blocks,
    static const int force_ring = env_int("BSR_TILE_RING", 0);     // a ring of this many buffers
    s.tile_stream = false;
    if (c->tile_stream && c->esz == 4) {
      // f32 storage: a unit of the ring is 1 KiB = 256 rows of one column (one LDS-DMA instruction): chunks of two blocks,
      // as many buffers as LDS holds (at most four); the waves copy at most 64 units per chunk
      const int room2 = (int)((budget - stream_ln_bytes(c->tile_qmax)) / ((size_t)max_ncols * 1024));
      if (room2 >= 2) { chunk = 2; ring = std::min(4, room2); }
      if (force_ring >= 2 && force_ring <= 4 && chunk > 0 && room2 >= force_ring) ring = force_ring;
      s.tile_stream = chunk > 0 && max_ncols <= BSR_STREAM_UNITS_MAX && group_chains == 1;   // (one chain's basis: the chunk block)
      if (!s.tile_stream) { chunk = 0; ring = 1; }
    } else if (c->tile_stream) {
      // the streaming kernel: chunks of one 128-row block through the deepest ring LDS holds (what is in flight keeps HBM
      // busy); two-block chunks where even four of those fit (a narrow batch).  Its waves copy at most 64 (column, block)
      // pieces per chunk; a wider batch takes k_tile over the same slices (the same sums, bit for bit).
      const int room = (int)((budget - stream_ln_bytes(c->tile_qmax)) / per_block);
      // two-block chunks (half the barriers and hand-overs per row) where the kernel evaluates them a block at a time --
      // the chunk block of assembly; the C++ interpreter would hold four values per lane and tape, and spills
      // (BSR_STREAM_CB2: 0 never, 1 always, default: by that rule)
      static const int two_env = env_int("BSR_STREAM_CB2", -1);
      const bool two = two_env >= 0 ? two_env != 0 : stream_chunk_block(K, tile_chains * K);
      if (room >= 4 && two) { ring = std::min(4, room / 2); chunk = 2; }
      else if (room >= 4) { ring = 4; chunk = 1; }
      else if (room >= 2) { ring = room; chunk = 1; }
      if (force_chunk > 0 && chunk > 0) chunk = std::min(chunk, std::min(force_chunk, 2));
      if (force_ring >= 2 && force_ring <= 4 && chunk > 0 && room >= force_ring * chunk) ring = force_ring;
      // (a batch of 33..39 columns that was given two-block chunks: one-block chunks fit the waves' 64 pieces)
      if (chunk == 2 && max_ncols * chunk > BSR_STREAM_UNITS_MAX && max_ncols <= BSR_STREAM_UNITS_MAX && force_chunk <= 0) {
        chunk = 1;
        ring = std::min(4, room);
      }
      s.tile_stream = chunk > 0 && max_ncols * chunk <= BSR_STREAM_UNITS_MAX;
      if (!s.tile_stream) { chunk = 0; ring = 1; }
    }
    if (s.tile_stream) {
    } else if (per_block * (size_t)(c->tile_bps + (c->tile_long > 0 ? 1 : 0)) <= budget &&
               (force_chunk <= 0 || force_chunk >= c->tile_bps + (c->tile_long > 0 ? 1 : 0))) {
      chunk = c->tile_bps + (c->tile_long > 0 ? 1 : 0);   // the whole slice, the longest one's blocks per column
    } else if (c->esz == 4) {   // f32: one buffer, staged through registers
      chunk = (int)std::min<size_t>((size_t)c->tile_bps, budget / per_block);
      if (chunk >= BSR_TILE_NB) chunk = chunk / BSR_TILE_NB * BSR_TILE_NB;
      if (force_chunk > 0) chunk = std::min(chunk, force_chunk);
    } else {
      // a ring of buffers, ring - 1 chunks in flight while the waves compute on one: what is in flight keeps HBM busy,
      // so prefer the deepest ring that still leaves chunks of two blocks (a chain pass of four values per lane)
      const int room = (int)(budget / per_block);   // blocks of all columns LDS holds
      if (room >= 8) { ring = 4; chunk = room / 4 >= BSR_TILE_NB ? BSR_TILE_NB : 2; }
      else if (room >= 6) { ring = 3; chunk = 2; }
      else if (room >= 4) { ring = 2; chunk = 2; }
      else if (room >= 2) { ring = 2; chunk = 1; }
      if (force_ring >= 2 && force_ring <= 4 && room >= force_ring) { ring = force_ring; chunk = std::max(1, std::min(room / ring, BSR_TILE_NB)); }
      if (force_chunk > 0 && chunk > 0) chunk = std::min(chunk, force_chunk);
      chunk = std::min(chunk, c->tile_bps);
    }
    if (chunk >= 1 && max_ncols < 32768) {
      s.tile = true;
      s.tile_ncols = max_ncols;
      s.tile_chunk = chunk;
      s.tile_ring = ring;
      memset(hf2, 0, fw * 8);
      for (int i = 0; i < n; ++i) {
        const TapeLoc& L = loc[i];
        const uint64_t* pf = hf + L.feat_off;
        uint64_t* pf2 = hf2 + L.feat_off;
        const int16_t* map = s.grp_slot.data() + (size_t)L.grp * c->n_cols;
        const int words = (L.nt + 1 + 3) / 4 + 1;   // the padding ids repeat the first terminal's
        for (int t = 0; t < words * 4; ++t) {
          const int col = (int)((pf[t >> 2] >> (16 * (t & 3))) & 0xFFFFu);
          pf2[t >> 2] |= (uint64_t)(uint16_t)map[col] << (16 * (t & 3));
        }
      }
    }
  }
  if (!s.tile) return;
  PropDesc* hd = s.h_desc();
  for (int i = 0; i < n; ++i) {
    hd[i].grp = loc[i].grp;
    const int in_group = (c->tile_by_chain && c->tile_T > 1) ? s.chain_slot[hd[i].ck] / c->tile_T : s.chain_slot[hd[i].ck];
    hd[i].qslot = s.grp_nF[loc[i].grp] + 1 + in_group * c->K;   // slot of the chain's basis in the group's LDS map
  }
}

// Geometry, schedule and tape records of a tile launch (after stage_tile): runs with the batch's launches.
int build_tile_launch(bsr_ctx* c, BatchSlot& s, int P, TileGeom* tgp) {
  TileGeom& tg = *tgp;
  memset(&tg, 0, sizeof tg);
  PropDesc* hd = s.h_desc();
  const int n_part = c->tile_slices + c->tile_left;
  if (s.order_n != P) cost_order(s.order_tmp, s.order_keys, P, [&](int i) { return hd[i].cost; });
  {
    // geometry of this launch: the context's slices, the chunk the batch's columns leave room for, and the schedule:
    // inside its group a tape goes -- heaviest first -- to the wave with the least work so far that still has a free
    // set of sums (waves w, w+4, w+8, w+12 share a SIMD, but a light wave frees issue slots for its SIMD mates, so
    // per-wave balance is what is worth having)
    tg.T = c->tile_T;
    tg.n_slices = c->tile_slices;
    tg.bps = c->tile_bps;
    tg.n_blocks = c->tile_blocks;
    tg.n_left = c->tile_left;
    tg.n_long = c->tile_long;
    tg.n_part = n_part;
    tg.ncols = s.tile_ncols;
    tg.ncols_fixed = s.tile_group_chains * c->K;
    tg.chunk_blocks = s.tile_chunk;
    tg.ring = s.tile_ring;
    tg.qmax = s.tile_stream ? c->tile_qmax : tile_qmax(c->K);   // (a streaming context's batch that takes k_tile after all)
    int cnt_g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < P; ++i) ++cnt_g[hd[i].grp & 7];
    int most = 0;
    for (int gi = 0; gi < tg.T; ++gi) most = std::max(most, cnt_g[gi]);
    const int per_pass = BSR_TILE_WAVES * tg.qmax;
    tg.n_pass = std::max(1, (most + per_pass - 1) / per_pass);
    const int slots_per_wave = tg.n_pass * tg.qmax;
    const size_t n_sched = (size_t)tg.T * tg.n_pass * per_pass;
    if (n_sched > c->tile_sched_cap) return fail(c, BSR_E_TOOBIG, "tile schedule larger than its buffer");
    TapeRec* sc = s.h_sched();
    // whole slice: the waves pull from the group's list (fp64; fp32 columns keep k_tile's static schedule over the slice
    // staged whole -- the same sums, one kernel family fewer in the library)
    tg.per_group = c->tile_whole && c->dtype == BSR_DTYPE_F64 && tg.chunk_blocks == tg.bps + (tg.n_long > 0 ? 1 : 0) ? most : 0;
    s.recs_bytes = n_sched * sizeof(TapeRec) + ((size_t)P + (size_t)tg.T * tg.per_group) * sizeof(int32_t);
    for (size_t i = 0; i < n_sched; ++i) sc[i].p = -1;
    int32_t* left_idx = reinterpret_cast<int32_t*>(sc + n_sched);   // tapes in cost order -> their records (leftover units)
    const uint64_t* hcodes = s.h_streams();
    const uint64_t* hfeats = hcodes + s.code_words;
    const double* hln = reinterpret_cast<const double*>(hfeats + s.feat_words);
    const uint64_t* hfeats_lds = reinterpret_cast<const uint64_t*>(hln + s.ln_words);
    s.wave_load.assign((size_t)tg.T * BSR_TILE_WAVES, 0.0);
    s.wave_cnt.assign((size_t)tg.T * BSR_TILE_WAVES, 0);
    for (int i = 0; i < P; ++i) {
      const int p = s.order_tmp[i];
      const PropDesc& D = hd[p];
      const int grp = D.grp;
      if (grp < 0 || grp >= tg.T) return fail(c, BSR_E_STATE, "tile schedule: tape group out of range");
      size_t ri;
      if (tg.per_group > 0) {
        // the waves pull their tapes: where in the group's share of the record array a record sits does not matter
        ri = (size_t)grp * tg.n_pass * per_pass + (size_t)s.wave_cnt[grp * BSR_TILE_WAVES]++;
      } else {
        int best = -1;
        for (int w = 0; w < BSR_TILE_WAVES; ++w) {
          const int idx = grp * BSR_TILE_WAVES + w;
          if (s.wave_cnt[idx] >= slots_per_wave) continue;
          if (best < 0 || s.wave_load[idx] < s.wave_load[grp * BSR_TILE_WAVES + best]) best = w;
        }
        if (best < 0) return fail(c, BSR_E_STATE, "tile schedule: no free set of sums");
        const int idx = grp * BSR_TILE_WAVES + best;
        const int slot = s.wave_cnt[idx]++;
        const int pass = slot / tg.qmax, q = slot % tg.qmax;
        ri = (((size_t)grp * tg.n_pass + pass) * BSR_TILE_WAVES + best) * tg.qmax + q;
        // (tried: less work for the younger waves of a SIMD -- it issues for its oldest wave first, the youngest runs the end
        // of every chunk alone -- by 15 and 30 %: no difference, 84.6-85.7 us either way)
        s.wave_load[idx] += (double)D.cost;
      }
      left_idx[i] = (int32_t)ri;
      // the tape's record: what the wave needs to start it, and the heads of its streams (a long tape reads on from them)
      TapeRec& R = sc[ri];
      R.p = p;
      R.n_nodes = D.n_nodes;
      // bit 1: the streaming kernel's scalar-register interpreter takes the tape (at most 16 entries, 8 terminals in
      // slots below 255, 3 ln nodes, one value on the stack below the accumulator)
      R.chain = D.chain | ((D.n_nodes <= 16 && D.n_term <= 8 && D.n_ln <= 3 && D.max_sp <= 2 && tg.ncols < 255) ? 2 : 0);
      R.qslot = D.qslot;
      R.s = D.s;
      R.code0 = hcodes[D.code_off];
      R.code1 = hcodes[D.code_off + 1];
      R.f0 = hfeats_lds[D.feat_off];
      R.f1 = hfeats_lds[D.feat_off + 1];
      const double* pl = hln + 2 * (size_t)D.ln_off;
      for (int t = 0; t < 3; ++t) {   // (the stream holds n_ln pairs and a padding pair)
        R.ln[2 * t] = (t <= D.n_ln) ? pl[2 * t] : 1.0;
        R.ln[2 * t + 1] = (t <= D.n_ln) ? pl[2 * t + 1] : 0.0;
      }
      R.code_off = D.code_off;
      R.feat_off = D.feat_off;
      R.ln_off = D.ln_off;
      R.n_ln = D.n_ln;
      R.n_term = D.n_term;
      R.grp = grp;
    }
    static const int gdump = env_int("BSR_GEOM_DUMP", 0);   // diagnostics: a batch's geometry, once per distinct one
    if (gdump) {
      static int last = -1;
      const int key = tg.ring * 1000000 + tg.chunk_blocks * 100000 + tg.ncols * 100 + tg.T;
      if (key != last) {
        last = key;
        fprintf(stderr, "geometry: stream %d ring %d chunk_blocks %d ncols %d (fixed %d) T %d qmax %d passes %d\n",
                (int)s.tile_stream, tg.ring, tg.chunk_blocks, tg.ncols, tg.ncols_fixed, tg.T, tg.qmax, tg.n_pass);
      }
    }
    static const int dump = env_int("BSR_SCHED_DUMP", 0);   // diagnostics: the schedule's loads (cost model units) per wave
    if (dump && tg.per_group == 0) {
      std::string line = "sched T=" + std::to_string(tg.T) + " qmax=" + std::to_string(tg.qmax) + " costs:";
      for (int i = 0; i < P; ++i) line += " " + std::to_string(hd[s.order_tmp[i]].cost);
      line += " | wave loads:";
      for (size_t i = 0; i < s.wave_load.size(); ++i) line += " " + std::to_string((int)s.wave_load[i]);
      fprintf(stderr, "%s\n", line.c_str());
      for (int w = 0; w < BSR_TILE_WAVES * tg.T; ++w) {
        std::string l2 = "  wave " + std::to_string(w) + ":";
        for (int q = 0; q < tg.qmax; ++q) {
          const TapeRec& R = sc[(size_t)w * tg.qmax + q];
          if (R.p < 0) continue;
          char b[96];
          snprintf(b, sizeof b, " [cost %d n %d chain %d code %llx]", hd[R.p].cost, R.n_nodes, R.chain, (unsigned long long)R.code0);
          l2 += b;
        }
        fprintf(stderr, "%s\n", l2.c_str());
      }
    }
    s.stat_tapes = P;
    s.stat_fast = s.stat_chain = s.stat_entries = 0;
    for (int i = 0; i < P; ++i) {
      s.stat_chain += hd[i].chain ? 1 : 0;
      s.stat_entries += hd[i].n_nodes;
    }
    s.srec_off = 0;
    if (s.tile_stream) {
      // the streaming kernel's 32-byte view of every set of sums, in schedule order, one record of padding behind the
      // last (a wave requests the next tape's program while it adds up this one's rows)
      s.srec_off = (s.recs_bytes + 31) / 32 * 32;
      StreamRec* sr = reinterpret_cast<StreamRec*>(reinterpret_cast<char*>(sc) + s.srec_off);
      std::vector<uint64_t>& ext_words = s.ext_words;   // (code, slots) pairs of the batch's long tapes, in record order
      ext_words.clear();
      // The kernel with a SECOND value below the accumulator (K <= 3, the pass block) costs every batch ~8 % (four more pinned
      // registers: C5's real mix 77 -> 83-86 us): taken where the batch holds enough trees of Strahler number 3 to pay for
      // it -- a sixteenth of its tapes; elsewhere those few go to the stack machine as before (the same bytes)
      int n_sp3 = 0;
      for (int i = 0; i < P; ++i) n_sp3 += (hd[i].max_sp == 3 && hd[i].n_ln <= BSR_STREAM_LN_PAIRS) ? 1 : 0;
      s.tile_deep2 = c->stream_deep >= 2 && c->esz == 8 && stream_deep2_applies(c->K, tg.ncols_fixed, tg.chunk_blocks) &&
                     n_sp3 * 16 >= P && n_sp3 >= 2;
      const int sp_max = s.tile_deep2 ? 3 : 2;
      for (size_t i = 0; i <= n_sched; ++i) {
        StreamRec& Q = sr[i];
        memset(&Q, 0, sizeof Q);
        if (i == n_sched || sc[i].p < 0) continue;
        const TapeRec& R = sc[i];
        const PropDesc& D = hd[R.p];
        bool fast = (R.chain & 2) != 0;
        // the entries behind the leading terminal as operator + 1, 0 behind the last (bsr_stream_asm.h); `log`, which
        // would need code 16, sends its tape to the stack machine
        const uint64_t* pcode = hcodes + D.code_off;
        auto entry = [&](int e) { return (int)((pcode[e >> 4] >> (4 * (e & 15))) & 15u); };
        bool has_log = false;
        for (int e = 1; e < R.n_nodes; ++e) has_log = has_log || entry(e) == BSR_OP_LOG;
        if (has_log) fast = false;
        uint64_t enc = 0;
        for (int e = 1; fast && e < R.n_nodes; ++e) enc |= (uint64_t)((entry(e) + 1) & 15) << (4 * (e - 1));
        uint64_t slots = 0;
        for (int t = 1; t < 8; ++t) {
          const uint64_t w = (t < 4) ? R.f0 : R.f1;
          const uint64_t id = (w >> (16 * (t & 3))) & 0xFFu;
          slots |= ((t < R.n_term) ? id : (uint64_t)0) << (8 * (t - 1));
        }
        // Round 6: a tape the 64-bit program does not hold -- more than 16 entries, 8 terminals or 3 ln nodes, or (K <= 3) a
        // second value below the accumulator -- as a program of SEVERAL words for the chunk block of assembly: the word in
        // the record (16 entries, 7 terminal slots) and up to BSR_STREAM_EXT_MAX extension words (16 entries, 8 slots each)
        // behind the record array; `end of word` fetches the next one (bsr_stream_chunk_asm.h: .Lsc_more).  A word closes
        // when its entries or its slots are used up.  Bit 31 without bit 5: the C++ and the tape-at-a-time interpreters
        // (test library) hand such a tape to the stack machine -- which is what the byte-equality tests compare it with.
        int n_ext = 0;
        size_t ext_first = 0;
        bool deep = false;
        if (!fast && !has_log && c->stream_deep > 0 && c->K <= 4 && tg.ncols_fixed == c->K && tg.ncols < 255 &&
            D.n_ln <= BSR_STREAM_LN_PAIRS && D.max_sp <= sp_max) {
          const uint64_t* pfeat = hfeats_lds + D.feat_off;
          auto term_id = [&](int t) { return (uint64_t)((pfeat[t >> 2] >> (16 * (t & 3))) & 0xFFu); };
          uint64_t wc[1 + BSR_STREAM_EXT_MAX] = {0}, ws[1 + BSR_STREAM_EXT_MAX] = {0};
          int ne = 0, ns_ = 0, wi = 0, t = 1;
          bool ok = true;
          for (int e = 1; e < R.n_nodes && ok; ++e) {
            const int op = entry(e);
            const bool needs = op == BSR_OP_TERMINAL || op == BSR_SOP_ADD_T || op == BSR_SOP_MUL_T;
            const int cap = (wi == 0) ? 7 : 8;
            if (ne == 16 || (needs && ns_ == cap)) {
              ++wi;
              ne = ns_ = 0;
              if (wi > BSR_STREAM_EXT_MAX) { ok = false; break; }
            }
            wc[wi] |= (uint64_t)((op + 1) & 15) << (4 * ne);
            ++ne;
            if (needs) {
              if (t >= R.n_term) { ok = false; break; }
              ws[wi] |= term_id(t++) << (8 * ns_);
              ++ns_;
            }
          }
          const size_t base = i - (i % (size_t)tg.qmax);                       // the wave's first record: the block's %[sr]
          const size_t off_units = (n_sched + 1 - base) * 2 + ext_words.size() / 2;
          if (ok && off_units + (size_t)wi < 1023) {
            deep = true;
            n_ext = wi;
            ext_first = off_units;
            enc = wc[0];
            slots = ws[0];
            for (int w = 1; w <= wi; ++w) { ext_words.push_back(wc[w]); ext_words.push_back(ws[w]); }
          }
        }
        static const int deep_dbg = env_int("BSR_DEEP_DUMP", 0);   // diagnostics: why a tape is (not) taken
        if (deep_dbg && !fast)
          fprintf(stderr, "tape %d: entries %d terms %d ln %d max_sp %d log %d ncols %d fixed %d K %d deep_knob %d -> deep %d (ext %d)\n", R.p,
                  R.n_nodes, R.n_term, D.n_ln, D.max_sp, (int)has_log, tg.ncols, tg.ncols_fixed, c->K, c->stream_deep, (int)deep, n_ext);
        s.stat_fast += (fast || deep) ? 1 : 0;
        Q.meta = ((R.n_nodes - 1) & 31) | (fast ? (int32_t)0x80000020 : deep ? (int32_t)0x40000000 : 0) | 64 | ((R.qslot & 0xFF) << 8) |
                 (n_ext << 16) | (int32_t)(ext_first << 20);
        Q.s = R.s;
        Q.code = (fast || deep) ? enc : 0;
        // (its byte offset in a chunk buffer: a column of a chunk is 1 KiB per f64 block -- and 1 KiB per two f32 blocks)
        Q.first = (int32_t)(R.f0 & 0xFFu) << ((tg.chunk_blocks == 2 && c->esz == 8) ? 11 : 10);
        Q.slots = slots;
      }
      // the extension words behind the records (and one word of padding: a wave's request never runs off the block)
      uint64_t* ext_dst = reinterpret_cast<uint64_t*>(sr + n_sched + 1);
      for (size_t w = 0; w < ext_words.size(); ++w) ext_dst[w] = ext_words[w];
      ext_dst[ext_words.size()] = ext_dst[ext_words.size() + 1] = 0;
      s.recs_bytes = s.srec_off + (n_sched + 1) * sizeof(StreamRec) + (ext_words.size() + 2) * 8;
    }
    s.tprog_off = 0;
    if (tg.per_group > 0) {   // per group: its tapes' records in cost order, -1 padded
      int32_t* glist = left_idx + P;
      for (size_t i = 0; i < (size_t)tg.T * tg.per_group; ++i) glist[i] = -1;
      int fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < P; ++i) {
        const int grp = hd[s.order_tmp[i]].grp;
        glist[(size_t)grp * tg.per_group + fill[grp]++] = left_idx[i];
      }
      if (c->tile_asm) {
        // the assembly tape loop's 64-byte programs (bsr_tile_asm.h), entry i of a group's array = entry i of its list,
        // one record of padding behind the last: what the block takes is a tape that holds at most one value below the
        // accumulator and whose entries behind the leading terminal fit one 64-bit word as operator + 1 (0 ends the tape; `log` would need 16), with its terminals'
        // slots in one word of bytes and at most two ln nodes
        s.tprog_off = (s.recs_bytes + 63) / 64 * 64;
        TileProg* tp = reinterpret_cast<TileProg*>(reinterpret_cast<char*>(sc) + s.tprog_off);
        const size_t n_prog = (size_t)tg.T * (tg.per_group + 1);
        for (size_t gi = 0; gi < (size_t)tg.T; ++gi) {
          for (int i = 0; i <= tg.per_group; ++i) {
            TileProg& Q = tp[gi * (tg.per_group + 1) + i];
            memset(&Q, 0, sizeof Q);
            Q.p = -1;
            const int32_t ri = i < tg.per_group ? glist[gi * tg.per_group + i] : -1;
            if (ri < 0) continue;
            const TapeRec& R = sc[ri];
            bool fast = ((R.chain & 1) != 0 || hd[R.p].max_sp <= 2) && R.n_nodes <= 17 && R.n_term <= 8 && R.n_ln <= 2 &&
                        tg.ncols < 256 && R.qslot < 256;
            static const int asm_mode = env_int("BSR_TILE_ASM", 1);   // (2: test hook -- every tape back to the C++ interpreter)
            if (asm_mode == 2) fast = false;
            uint64_t enc = 0;
            for (int e = 1; fast && e < R.n_nodes; ++e) {
              const uint64_t w = e < 16 ? R.code0 : R.code1;
              const int op = (int)((w >> (4 * (e & 15))) & 15u);
              if (op == BSR_OP_LOG) fast = false;
              enc |= (uint64_t)((op + 1) & 15) << (4 * (e - 1));
            }
            uint64_t slots = 0;
            for (int t = 0; t < 8 && t < R.n_term; ++t) {
              const uint64_t w = (t < 4) ? R.f0 : R.f1;
              slots |= ((w >> (16 * (t & 3))) & 0xFFu) << (8 * t);
            }
            static const int astats = env_int("BSR_ASM_STATS", 0);   // diagnostics: what the block of assembly does not take
            if (astats) {
              static std::atomic<long> n_all{0}, n_fast{0}, n_stack{0};
              const long a = ++n_all, f = n_fast += fast ? 1 : 0, st = n_stack += (R.chain & 1) ? 0 : 1;
              if (a % 6400 == 0) fprintf(stderr, "tile asm: %ld tapes, %ld for the block, %ld not chains\n", a, f, st);
            }
            s.stat_fast += fast ? 1 : 0;
            Q.meta = (fast ? (int32_t)0x80000000 : 0) | (R.qslot & 0xFF);
            Q.p = R.p;
            Q.s = R.s;
            Q.code = fast ? enc : 0;
            Q.slots = slots;
            for (int t = 0; t < 4; ++t) Q.ln[t] = R.ln[t];
          }
        }
        s.recs_bytes = s.tprog_off + n_prog * sizeof(TileProg);
      }
    }
  }
  return BSR_OK;
}

