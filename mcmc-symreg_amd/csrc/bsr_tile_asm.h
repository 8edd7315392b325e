// The tape loop of the whole-slice row pass (bsr_tile_asm.hip: k_tile1a) for ONE WAVE in gfx950 assembly: pull a tape from the
// group's list, fetch its program, run it over the slice in passes of four 128-row blocks, add the rows into the tape's
// sums, reduce them over the lanes, store the (tape, slice) partial record, go on with the next tape.
//
// Why assembly (round 5; the measurements are in DESIGN.md 7): the compiler's k_tile1 spends ~240 scalar and ~100 vector
// instructions per (tape, slice) before the first row is touched -- record fetch through two dependent loads, compare trees
// per entry, the stream words' bookkeeping, 32 spilled scalar registers that come back by v_readlane (a VECTOR
// instruction), the store's predication -- and a CU has one scalar unit for its sixteen waves: at C2 it was busy 80 % of
// the compute phase next to a vector unit busy 90 %.  Here a tape costs ~45 scalar instructions plus 5 per entry and
// pass, and the vector instructions that are not arithmetic are the address adds of the loads and the lane reduction.
//
// What it runs: tapes that hold at most ONE value below the accumulator (chains -- bsr_device.h: chain_eval -- and shapes
// like (x0 + x1) * sin(x2)), of at most 17 entries, 8 terminals in LDS slots below 256, 2 ln nodes, no `log` -- the host
// marks them (TileProg.meta bit 31; bsr_stage.hip).  Anything else, and a sin / cos whose
// argument is huge, infinite or NaN, leaves the block with st = 1 and the tape's list index in idx: the caller runs the
// tape through the C++ interpreter and comes back with the next index.  Every operator is the instruction sequence of
// bsr_stream_chunk_asm.h (the compiler's own for the C++ of bsr_device.h / bsr_fastmath.h), a lane's rows reach the
// sums block by block in row order and the lane reduction combines in the order of store_partial (halves, rows 16
// apart, then lane ^ 1, 2, 4, 8): the record is the same BYTES as k_tile1's (tests/test_gpu_edges.py).
//
// A pass holds four blocks: eight values per lane in v[40:55] (block j: v[40+4j : 43+4j], one ds_read_b128), the operand
// of a fused `acc (+|*) column` in v[56:71]; the per-value operator macros of bsr_stream_chunk_asm.h keep their fixed
// temporaries v[8:19], v20, v[22:39].  A pass of fewer blocks (slices of a length that is no multiple of four) reads
// block 0 again for the missing ones: computed along, never added up.
//
// Fixed registers (all declared clobbered):
//   v[0:39]   temporaries of the operators; in the add-up: y and the basis values of two blocks (v[0:19], v[20:39])
//   v[40:55]  the pass's values, v[56:71] operand / scaled values, v[72:85] the sums: c0 c1 c2 c3 |sz|^2 sz.y max|z|
//   v86 LDS address of the lane's pair in column 0, block 0 of the pass; v87 scratch; v[88:92] addresses of y and the basis;
//   v[94:109] the saved value (the operand a binary operator pops)
//   s[8:13] scratch (s[12:13]: the record's address), s[16:31] the program (TileProg), s[78:79] operator table,
//   s[80:81] jump target (s32..s35 are the stack's: left alone), s[36:59] the math constants, s[60:61] entries / s[62:63] slots / s[64:71] ln pairs being
//   consumed, s72 first block of the pass, s73 its blocks, s74 list index of the tape, s75 of the next, s76 the second
//   half of the slice is still to be waited for
#pragma once
#include "bsr_stream_chunk_asm.h"

// clang-format off
#define BSR_TA_DISPATCH                          \
  "s_lshl_b32 s10, s60, 7\n\t"                   \
  "s_and_b32 s10, s10, 0x780\n\t"                \
  "s_lshr_b64 s[60:61], s[60:61], 4\n\t"         \
  "s_or_b32 s80, s78, s10\n\t"                   \
  "s_setpc_b64 s[80:81]\n\t"

// the next terminal's column: LDS address of the lane's pair in block 0 of the pass into v20
#define BSR_TA_SLOT_ADDR                         \
  "s_and_b32 s10, s62, 0xff\n\t"                 \
  "s_lshr_b64 s[62:63], s[62:63], 8\n\t"         \
  "s_mul_i32 s10, s10, %[stride]\n\t"            \
  "v_add_u32_e32 v20, s10, v86\n\t"

// four blocks of the column at v20 into r0..r3 (a pass of fewer blocks: block 0 for the missing ones)
#define BSR_TA_LOAD4(r0, r1, r2, r3, tag)                      \
  "s_cmp_eq_u32 s73, 4\n\t"                                    \
  "s_cbranch_scc0 .Lta_ldp" tag "_%=\n\t"                      \
  "ds_read_b128 " r0 ", v20\n\t"                               \
  "ds_read_b128 " r1 ", v20 offset:1024\n\t"                   \
  "ds_read_b128 " r2 ", v20 offset:2048\n\t"                   \
  "ds_read_b128 " r3 ", v20 offset:3072\n\t"                   \
  "s_branch .Lta_ldd" tag "_%=\n"                              \
  ".Lta_ldp" tag "_%=:\n\t"                                    \
  "ds_read_b128 " r0 ", v20\n\t"                               \
  "ds_read_b128 " r1 ", v20\n\t"                               \
  "ds_read_b128 " r2 ", v20\n\t"                               \
  "ds_read_b128 " r3 ", v20\n\t"                               \
  "s_cmp_lt_u32 s73, 2\n\t"                                    \
  "s_cbranch_scc1 .Lta_ldd" tag "_%=\n\t"                      \
  "ds_read_b128 " r1 ", v20 offset:1024\n\t"                   \
  "s_cmp_lt_u32 s73, 3\n\t"                                    \
  "s_cbranch_scc1 .Lta_ldd" tag "_%=\n\t"                      \
  "ds_read_b128 " r2 ", v20 offset:2048\n"                     \
  ".Lta_ldd" tag "_%=:\n\t"

// M(value) for the pass's eight values / M(value, operand) for the values and the operand column
#define BSR_TA_EACH(M)                                                                                   \
  M("v[40:41]", "v40", "v41") M("v[42:43]", "v42", "v43") M("v[44:45]", "v44", "v45") M("v[46:47]", "v46", "v47") \
  M("v[48:49]", "v48", "v49") M("v[50:51]", "v50", "v51") M("v[52:53]", "v52", "v53") M("v[54:55]", "v54", "v55")
#define BSR_TA_EACH2(M)                                                                                  \
  M("v[40:41]", "v[56:57]") M("v[42:43]", "v[58:59]") M("v[44:45]", "v[60:61]") M("v[46:47]", "v[62:63]") \
  M("v[48:49]", "v[64:65]") M("v[50:51]", "v[66:67]") M("v[52:53]", "v[68:69]") M("v[54:55]", "v[70:71]")

// ... and M(value, lo, hi, saved) for the values and the one value a tape may hold below them (v[94:109])
#define BSR_TA_EACHS(M)                                                                                  \
  M("v[40:41]", "v40", "v41", "v[94:95]") M("v[42:43]", "v42", "v43", "v[96:97]") M("v[44:45]", "v44", "v45", "v[98:99]") \
  M("v[46:47]", "v46", "v47", "v[100:101]") M("v[48:49]", "v48", "v49", "v[102:103]") M("v[50:51]", "v50", "v51", "v[104:105]") \
  M("v[52:53]", "v52", "v53", "v[106:107]") M("v[54:55]", "v54", "v55", "v[108:109]")
#define BSR_TA_SAVE1(x, lo, hi, sv) "v_mov_b64_e32 " sv ", " x "\n\t"
#define BSR_TA_SADD1(x, lo, hi, sv) "v_add_f64 " x ", " sv ", " x "\n\t"
#define BSR_TA_SMUL1(x, lo, hi, sv) "v_mul_f64 " x ", " sv ", " x "\n\t"
#define BSR_TA_SSUB1(x, lo, hi, sv) "v_add_f64 " x ", " sv ", -" x "\n\t"
#define BSR_TA_SDIV1(x, lo, hi, sv) BSR_SC_DIV(x, lo, hi, sv)
#define BSR_TA_ADD2(x, p) "v_add_f64 " x ", " x ", " p "\n\t"
#define BSR_TA_MUL2(x, p) "v_mul_f64 " x ", " x ", " p "\n\t"
#define BSR_TA_BIN_T(M, tag)                                                     \
  BSR_TA_SLOT_ADDR                                                               \
  BSR_TA_LOAD4("v[56:59]", "v[60:63]", "v[64:67]", "v[68:71]", tag)              \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_EACH2(M)                                                                \
  BSR_TA_DISPATCH
#define BSR_TA_LN1(x, lo, hi)    "v_mul_f64 " x ", s[64:65], " x "\n\t"
#define BSR_TA_LN2(x, lo, hi)    "v_add_f64 " x ", " x ", s[66:67]\n\t"
#define BSR_TA_NEG1(x, lo, hi)   "v_xor_b32_e32 " hi ", 0x80000000, " hi "\n\t"
#define BSR_TA_SQ1(x, lo, hi)    "v_mul_f64 " x ", " x ", " x "\n\t"
#define BSR_TA_INV1(x, lo, hi)   BSR_SC_DIV(x, lo, hi, "1.0")
#define BSR_TA_SIN1(x, lo, hi)   BSR_SC_SINCOS(x, lo, hi, "", BSR_SC_SIN_TINY(x, lo, hi))
#define BSR_TA_COS1(x, lo, hi)   BSR_SC_SINCOS(x, lo, hi, "v_add_u32_e32 v20, 64, v20\n\t", BSR_SC_COS_MOVE(x))
// a huge, infinite or NaN argument in any lane: the caller's interpreter takes the tape (the library routine for those lanes)
#define BSR_TA_BIG1(x, lo, hi)                                                   \
  "v_cmp_nlt_f64_e64 vcc, |" x "|, s[56:57]\n\t"                                 \
  "s_or_b64 s[8:9], s[8:9], vcc\n\t"
#define BSR_TA_BIG_CHECK                                                         \
  "s_mov_b64 s[8:9], 0\n\t"                                                      \
  BSR_TA_EACH(BSR_TA_BIG1)                                                       \
  "s_cmp_lg_u64 s[8:9], 0\n\t"                                                   \
  "s_cbranch_scc1 .Lta_generic%=\n\t"
#define BSR_TA_SLOT(n) ".p2align 7\n.Lta_op" n "_%=:\n\t"

// ---- the rows of one block into the sums.  z0, z1: the lane's two values; B: register number of the block's y pair
// (y at B, basis column i at B + 4 + 4 i); the scaled values go through v[56:59].
#define BSR_TA_ROW_K1(zs, y, qa, qb, qc, qd)                                     \
  "v_fmac_f64_e32 v[80:81], " zs ", " zs "\n\t"                                  \
  "v_fmac_f64_e32 v[82:83], " zs ", " y "\n\t"                                   \
  "v_fmac_f64_e32 v[72:73], " qa ", " zs "\n\t"
#define BSR_TA_ROW_K2(zs, y, qa, qb, qc, qd)  BSR_TA_ROW_K1(zs, y, qa, qb, qc, qd) "v_fmac_f64_e32 v[74:75], " qb ", " zs "\n\t"
#define BSR_TA_ROW_K3(zs, y, qa, qb, qc, qd)  BSR_TA_ROW_K2(zs, y, qa, qb, qc, qd) "v_fmac_f64_e32 v[76:77], " qc ", " zs "\n\t"
#define BSR_TA_ROW_K4(zs, y, qa, qb, qc, qd)  BSR_TA_ROW_K3(zs, y, qa, qb, qc, qd) "v_fmac_f64_e32 v[78:79], " qd ", " zs "\n\t"
#define BSR_TA_ACC_X(ROW, z0, z1)                                                \
  "v_mul_f64 v[56:57], " z0 ", s[18:19]\n\t"                                     \
  "v_mul_f64 v[58:59], " z1 ", s[18:19]\n\t"                                     \
  "v_max_f64 v[84:85], v[84:85], |" z0 "|\n\t"                                   \
  ROW("v[56:57]", "v[0:1]", "v[4:5]", "v[8:9]", "v[12:13]", "v[16:17]")          \
  "v_max_f64 v[84:85], v[84:85], |" z1 "|\n\t"                                   \
  ROW("v[58:59]", "v[2:3]", "v[6:7]", "v[10:11]", "v[14:15]", "v[18:19]")
#define BSR_TA_ACC_Y(ROW, z0, z1)                                                \
  "v_mul_f64 v[56:57], " z0 ", s[18:19]\n\t"                                     \
  "v_mul_f64 v[58:59], " z1 ", s[18:19]\n\t"                                     \
  "v_max_f64 v[84:85], v[84:85], |" z0 "|\n\t"                                   \
  ROW("v[56:57]", "v[20:21]", "v[24:25]", "v[28:29]", "v[32:33]", "v[36:37]")    \
  "v_max_f64 v[84:85], v[84:85], |" z1 "|\n\t"                                   \
  ROW("v[58:59]", "v[22:23]", "v[26:27]", "v[30:31]", "v[34:35]", "v[38:39]")
// y and the K basis columns of block `off` (bytes behind block 0 of the pass) into buffer X (v[0:19]) / Y (v[20:39])
#define BSR_TA_RDX_K1(off) "ds_read_b128 v[0:3], v88 offset:" off "\n\tds_read_b128 v[4:7], v89 offset:" off "\n\t"
#define BSR_TA_RDX_K2(off) BSR_TA_RDX_K1(off) "ds_read_b128 v[8:11], v90 offset:" off "\n\t"
#define BSR_TA_RDX_K3(off) BSR_TA_RDX_K2(off) "ds_read_b128 v[12:15], v91 offset:" off "\n\t"
#define BSR_TA_RDX_K4(off) BSR_TA_RDX_K3(off) "ds_read_b128 v[16:19], v92 offset:" off "\n\t"
#define BSR_TA_RDY_K1(off) "ds_read_b128 v[20:23], v88 offset:" off "\n\tds_read_b128 v[24:27], v89 offset:" off "\n\t"
#define BSR_TA_RDY_K2(off) BSR_TA_RDY_K1(off) "ds_read_b128 v[28:31], v90 offset:" off "\n\t"
#define BSR_TA_RDY_K3(off) BSR_TA_RDY_K2(off) "ds_read_b128 v[32:35], v91 offset:" off "\n\t"
#define BSR_TA_RDY_K4(off) BSR_TA_RDY_K3(off) "ds_read_b128 v[36:39], v92 offset:" off "\n\t"
// the basis columns' addresses behind y's (v88): column i of the tape's basis at v89 + i
#define BSR_TA_QADDR_K1 ""
#define BSR_TA_QADDR_K2 "v_add_u32_e32 v90, %[stride], v89\n\t"
#define BSR_TA_QADDR_K3 BSR_TA_QADDR_K2 "v_add_u32_e32 v91, %[stride], v90\n\t"
#define BSR_TA_QADDR_K4 BSR_TA_QADDR_K3 "v_add_u32_e32 v92, %[stride], v91\n\t"
// "wait until only the reads of one block are in flight": K + 1 of them
#define BSR_TA_WAIT1_K1 "s_waitcnt lgkmcnt(2)\n\t"
#define BSR_TA_WAIT1_K2 "s_waitcnt lgkmcnt(3)\n\t"
#define BSR_TA_WAIT1_K3 "s_waitcnt lgkmcnt(4)\n\t"
#define BSR_TA_WAIT1_K4 "s_waitcnt lgkmcnt(5)\n\t"

// the add-up of a pass: four blocks through the two buffers (the next block's reads under this one's sums); a pass of
// fewer blocks one at a time
#define BSR_TA_ADDUP(ROW, RDX, RDY, QADDR, WAIT1)                                \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_add_u32_e32 v88, %[yoff], v86\n\t"                                          \
  "s_and_b32 s10, s16, 0xff\n\t"                                                 \
  "s_mul_i32 s10, s10, %[stride]\n\t"                                            \
  "v_add_u32_e32 v89, s10, v86\n\t"                                              \
  QADDR                                                                          \
  "s_cmp_eq_u32 s73, 4\n\t"                                                      \
  "s_cbranch_scc0 .Lta_addp%=\n\t"                                               \
  RDX("0") RDY("1024")                                                           \
  WAIT1                                                                          \
  BSR_TA_ACC_X(ROW, "v[40:41]", "v[42:43]")                                      \
  RDX("2048")                                                                    \
  WAIT1                                                                          \
  BSR_TA_ACC_Y(ROW, "v[44:45]", "v[46:47]")                                      \
  RDY("3072")                                                                    \
  WAIT1                                                                          \
  BSR_TA_ACC_X(ROW, "v[48:49]", "v[50:51]")                                      \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_ACC_Y(ROW, "v[52:53]", "v[54:55]")                                      \
  "s_branch .Lta_added%=\n"                                                      \
  ".Lta_addp%=:\n\t"                                                             \
  RDX("0")                                                                       \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_ACC_X(ROW, "v[40:41]", "v[42:43]")                                      \
  "s_cmp_lt_u32 s73, 2\n\t"                                                      \
  "s_cbranch_scc1 .Lta_added%=\n\t"                                              \
  RDX("1024")                                                                    \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_ACC_X(ROW, "v[44:45]", "v[46:47]")                                      \
  "s_cmp_lt_u32 s73, 3\n\t"                                                      \
  "s_cbranch_scc1 .Lta_added%=\n\t"                                              \
  RDX("2048")                                                                    \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_ACC_X(ROW, "v[48:49]", "v[50:51]")                                      \
  ".Lta_added%=:\n\t"

// ---- lane reduction.  swap32(a, b) + add: lanes 0..31 end with a over both halves, lanes 32..63 with b; swap16 + add:
// rows 0, 1, 2, 3 end with the first register's halves 0, the second's halves 0, the first's halves 1, the second's
// halves 1 -- for quantities (p, q) | (r, s) that is p, r, q, s; then lane ^ 1, 2, 4, 8 on the LDS crossbar.
#define BSR_TA_SWAP32(alo, ahi, blo, bhi)                                        \
  "v_permlane32_swap_b32_e32 " alo ", " blo "\n\t"                               \
  "v_permlane32_swap_b32_e32 " ahi ", " bhi "\n\t"
#define BSR_TA_SWAP16(alo, ahi, blo, bhi)                                        \
  "v_permlane16_swap_b32_e32 " alo ", " blo "\n\t"                               \
  "v_permlane16_swap_b32_e32 " ahi ", " bhi "\n\t"
#define BSR_TA_SWZ(dlo, dhi, slo, shi, x)                                        \
  "ds_swizzle_b32 " dlo ", " slo " offset:swizzle(SWAP," x ")\n\t"               \
  "ds_swizzle_b32 " dhi ", " shi " offset:swizzle(SWAP," x ")\n\t"
// one butterfly step of the two sums in v[0:1], v[4:5] and of the maximum in v[8:9] (partners through v[2:3], v[6:7], v[10:11])
#define BSR_TA_STEP3(x)                                                          \
  BSR_TA_SWZ("v2", "v3", "v0", "v1", x)                                          \
  BSR_TA_SWZ("v6", "v7", "v4", "v5", x)                                          \
  BSR_TA_SWZ("v10", "v11", "v8", "v9", x)                                        \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_add_f64 v[0:1], v[0:1], v[2:3]\n\t"                                         \
  "v_add_f64 v[4:5], v[4:5], v[6:7]\n\t"                                         \
  "v_max_f64 v[8:9], v[8:9], v[10:11]\n\t"
#define BSR_TA_STEP2(x)                                                          \
  BSR_TA_SWZ("v2", "v3", "v0", "v1", x)                                          \
  BSR_TA_SWZ("v10", "v11", "v8", "v9", x)                                        \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_add_f64 v[0:1], v[0:1], v[2:3]\n\t"                                         \
  "v_max_f64 v[8:9], v[8:9], v[10:11]\n\t"
// the maximum's last steps: rows 16 apart, then the two halves (every lane ends with the wave's maximum, in v[8:9])
#define BSR_TA_MAX_TAIL                                                          \
  BSR_TA_SWZ("v10", "v11", "v8", "v9", "16")                                     \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_max_f64 v[8:9], v[8:9], v[10:11]\n\t"                                       \
  "s_nop 1\n\t"                                                                  \
  "v_mov_b32_e32 v10, v8\n\t"                                                    \
  "v_mov_b32_e32 v11, v9\n\t"                                                    \
  "s_nop 1\n\t"                                                                  \
  BSR_TA_SWAP32("v8", "v9", "v10", "v11")                                        \
  "s_nop 1\n\t"                                                                  \
  "v_max_f64 v[8:9], v[8:9], v[10:11]\n\t"
// the record: part + (p n_part + slice) 96 bytes (the caller's `part` already stands at the slice's record of tape 0)
#define BSR_TA_REC_ADDR                                                          \
  "s_mul_i32 s10, s17, %[np96]\n\t"                                              \
  "s_mul_hi_u32 s11, s17, %[np96]\n\t"                                           \
  "s_add_u32 s12, %[part], s10\n\t"                                              \
  "s_addc_u32 s13, %[parth], s11\n\t"
// the next tape's program, requested under the reduction (the registers of this one are done with)
#define BSR_TA_NEXT_PROG                                                         \
  "s_min_u32 s10, s75, %[nitems]\n\t"   /* (beyond the list: its one record of padding) */ \
  "s_lshl_b32 s10, s10, 6\n\t"                                                   \
  "s_load_dwordx16 s[16:31], %[progs], s10\n\t"
// max|z| and a zero behind it (words 10, 11) from lane 0, whose exec the caller has set; v87 = 0
#define BSR_TA_STORE_MAX                                                         \
  "v_mov_b32_e32 v10, 0\n\t"                                                     \
  "v_mov_b32_e32 v11, 0\n\t"                                                     \
  "global_store_dwordx4 v87, v[8:11], s[12:13] offset:80\n\t"
#define BSR_TA_ZERO4                                                             \
  "v_mov_b32_e32 v12, 0\n\t"                                                     \
  "v_mov_b32_e32 v13, 0\n\t"                                                     \
  "v_mov_b32_e32 v14, 0\n\t"                                                     \
  "v_mov_b32_e32 v15, 0\n\t"

// K = 3: (c0 c1) (c2 |sz|^2) through the swap network -> rows c0, c2, c1, |sz|^2 in v[0:1]; sz.y with itself -> v[4:5]
#define BSR_TA_REDUCE_K3                                                         \
  BSR_TA_REC_ADDR                                                                \
  BSR_TA_NEXT_PROG                                                               \
  BSR_TA_SWAP32("v72", "v73", "v74", "v75")                                      \
  BSR_TA_SWAP32("v76", "v77", "v80", "v81")                                      \
  "v_mov_b32_e32 v4, v82\n\t"                                                    \
  "v_mov_b32_e32 v5, v83\n\t"                                                    \
  "v_add_f64 v[0:1], v[72:73], v[74:75]\n\t"                                     \
  "v_add_f64 v[2:3], v[76:77], v[80:81]\n\t"                                     \
  BSR_TA_SWAP32("v82", "v83", "v4", "v5")                                        \
  BSR_TA_SWAP16("v0", "v1", "v2", "v3")                                          \
  "v_mov_b64_e32 v[8:9], v[84:85]\n\t"                                           \
  "v_add_f64 v[4:5], v[82:83], v[4:5]\n\t"                                       \
  "v_add_f64 v[0:1], v[0:1], v[2:3]\n\t"                                         \
  BSR_TA_SWZ("v6", "v7", "v4", "v5", "16")                                       \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_add_f64 v[4:5], v[4:5], v[6:7]\n\t"                                         \
  BSR_TA_STEP3("1") BSR_TA_STEP3("2") BSR_TA_STEP3("4") BSR_TA_STEP3("8")        \
  BSR_TA_MAX_TAIL                                                                \
  "s_mov_b64 exec, %[m4]\n\t"   /* lanes 0, 16, 32, 48: words 0, 2, 1, 8 */     \
  "s_nop 1\n\t"                                                                  \
  "global_store_dwordx2 %[so], v[0:1], s[12:13]\n\t"                             \
  "s_mov_b64 exec, 1\n\t"                                                        \
  "s_nop 1\n\t"                                                                  \
  "v_mov_b32_e32 v87, 0\n\t"                                                     \
  BSR_TA_ZERO4                                                                   \
  "global_store_dwordx4 v87, v[12:15], s[12:13] offset:24\n\t"                   \
  "global_store_dwordx4 v87, v[12:15], s[12:13] offset:40\n\t"                   \
  "global_store_dwordx2 v87, v[12:13], s[12:13] offset:56\n\t"                   \
  "global_store_dwordx2 v87, v[4:5], s[12:13] offset:72\n\t"                     \
  BSR_TA_STORE_MAX                                                               \
  "s_mov_b64 exec, -1\n\t"                                                       \
  "s_nop 1\n\t"
// K = 4: (c0 c1) (c2 c3) -> rows c0, c2, c1, c3 in v[0:1]; (|sz|^2 sz.y) -> lanes 0..31 / 32..63 of v[4:5]
#define BSR_TA_REDUCE_K4                                                         \
  BSR_TA_REC_ADDR                                                                \
  BSR_TA_NEXT_PROG                                                               \
  BSR_TA_SWAP32("v72", "v73", "v74", "v75")                                      \
  BSR_TA_SWAP32("v76", "v77", "v78", "v79")                                      \
  BSR_TA_SWAP32("v80", "v81", "v82", "v83")                                      \
  "v_add_f64 v[0:1], v[72:73], v[74:75]\n\t"                                     \
  "v_add_f64 v[2:3], v[76:77], v[78:79]\n\t"                                     \
  "v_add_f64 v[4:5], v[80:81], v[82:83]\n\t"                                     \
  "v_mov_b64_e32 v[8:9], v[84:85]\n\t"                                           \
  BSR_TA_SWAP16("v0", "v1", "v2", "v3")                                          \
  BSR_TA_SWZ("v6", "v7", "v4", "v5", "16")                                       \
  "v_add_f64 v[0:1], v[0:1], v[2:3]\n\t"                                         \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_add_f64 v[4:5], v[4:5], v[6:7]\n\t"                                         \
  BSR_TA_STEP3("1") BSR_TA_STEP3("2") BSR_TA_STEP3("4") BSR_TA_STEP3("8")        \
  BSR_TA_MAX_TAIL                                                                \
  "s_mov_b64 exec, %[m4]\n\t"   /* lanes 0, 16, 32, 48: words 0, 2, 1, 3 */     \
  "s_nop 1\n\t"                                                                  \
  "global_store_dwordx2 %[so], v[0:1], s[12:13]\n\t"                             \
  "s_mov_b64 exec, %[m2]\n\t"   /* lanes 0, 32: words 8, 9 */                   \
  "s_nop 1\n\t"                                                                  \
  "global_store_dwordx2 %[so2], v[4:5], s[12:13]\n\t"                            \
  "s_mov_b64 exec, 1\n\t"                                                        \
  "s_nop 1\n\t"                                                                  \
  "v_mov_b32_e32 v87, 0\n\t"                                                     \
  BSR_TA_ZERO4                                                                   \
  "global_store_dwordx4 v87, v[12:15], s[12:13] offset:32\n\t"                   \
  "global_store_dwordx4 v87, v[12:15], s[12:13] offset:48\n\t"                   \
  BSR_TA_STORE_MAX                                                               \
  "s_mov_b64 exec, -1\n\t"                                                       \
  "s_nop 1\n\t"
// K = 2: (c0 c1) (|sz|^2 sz.y) -> rows c0, |sz|^2, c1, sz.y in v[0:1]
#define BSR_TA_REDUCE_K2                                                         \
  BSR_TA_REC_ADDR                                                                \
  BSR_TA_NEXT_PROG                                                               \
  BSR_TA_SWAP32("v72", "v73", "v74", "v75")                                      \
  BSR_TA_SWAP32("v80", "v81", "v82", "v83")                                      \
  "v_add_f64 v[0:1], v[72:73], v[74:75]\n\t"                                     \
  "v_add_f64 v[2:3], v[80:81], v[82:83]\n\t"                                     \
  "v_mov_b64_e32 v[8:9], v[84:85]\n\t"                                           \
  "s_nop 1\n\t"                                                                  \
  BSR_TA_SWAP16("v0", "v1", "v2", "v3")                                          \
  "s_nop 1\n\t"                                                                  \
  "v_add_f64 v[0:1], v[0:1], v[2:3]\n\t"                                         \
  BSR_TA_STEP2("1") BSR_TA_STEP2("2") BSR_TA_STEP2("4") BSR_TA_STEP2("8")        \
  BSR_TA_MAX_TAIL                                                                \
  "s_mov_b64 exec, %[m4]\n\t"   /* lanes 0, 16, 32, 48: words 0, 8, 1, 9 */     \
  "s_nop 1\n\t"                                                                  \
  "global_store_dwordx2 %[so], v[0:1], s[12:13]\n\t"                             \
  "s_mov_b64 exec, 1\n\t"                                                        \
  "s_nop 1\n\t"                                                                  \
  "v_mov_b32_e32 v87, 0\n\t"                                                     \
  BSR_TA_ZERO4                                                                   \
  "global_store_dwordx4 v87, v[12:15], s[12:13] offset:16\n\t"                   \
  "global_store_dwordx4 v87, v[12:15], s[12:13] offset:32\n\t"                   \
  "global_store_dwordx4 v87, v[12:15], s[12:13] offset:48\n\t"                   \
  BSR_TA_STORE_MAX                                                               \
  "s_mov_b64 exec, -1\n\t"                                                       \
  "s_nop 1\n\t"
// K = 1: (c0 |sz|^2) (sz.y with itself) -> rows c0, sz.y, |sz|^2, sz.y in v[0:1]
#define BSR_TA_REDUCE_K1                                                         \
  BSR_TA_REC_ADDR                                                                \
  BSR_TA_NEXT_PROG                                                               \
  "v_mov_b32_e32 v4, v82\n\t"                                                    \
  "v_mov_b32_e32 v5, v83\n\t"                                                    \
  BSR_TA_SWAP32("v72", "v73", "v80", "v81")                                      \
  "s_nop 0\n\t"                                                                  \
  BSR_TA_SWAP32("v82", "v83", "v4", "v5")                                        \
  "v_add_f64 v[0:1], v[72:73], v[80:81]\n\t"                                     \
  "v_add_f64 v[2:3], v[82:83], v[4:5]\n\t"                                       \
  "v_mov_b64_e32 v[8:9], v[84:85]\n\t"                                           \
  "s_nop 1\n\t"                                                                  \
  BSR_TA_SWAP16("v0", "v1", "v2", "v3")                                          \
  "s_nop 1\n\t"                                                                  \
  "v_add_f64 v[0:1], v[0:1], v[2:3]\n\t"                                         \
  BSR_TA_STEP2("1") BSR_TA_STEP2("2") BSR_TA_STEP2("4") BSR_TA_STEP2("8")        \
  BSR_TA_MAX_TAIL                                                                \
  "s_mov_b64 exec, %[m4]\n\t"   /* lanes 0, 16, 32 (the mask leaves 48 out): words 0, 9, 8 */ \
  "s_nop 1\n\t"                                                                  \
  "global_store_dwordx2 %[so], v[0:1], s[12:13]\n\t"                             \
  "s_mov_b64 exec, 1\n\t"                                                        \
  "s_nop 1\n\t"                                                                  \
  "v_mov_b32_e32 v87, 0\n\t"                                                     \
  BSR_TA_ZERO4                                                                   \
  "global_store_dwordx4 v87, v[12:15], s[12:13] offset:8\n\t"                    \
  "global_store_dwordx4 v87, v[12:15], s[12:13] offset:24\n\t"                   \
  "global_store_dwordx4 v87, v[12:15], s[12:13] offset:40\n\t"                   \
  "global_store_dwordx2 v87, v[12:13], s[12:13] offset:56\n\t"                   \
  BSR_TA_STORE_MAX                                                               \
  "s_mov_b64 exec, -1\n\t"                                                       \
  "s_nop 1\n\t"

// ---- the block
#define BSR_TILE_TAPES_ASM_(ROW, RDX, RDY, QADDR, WAIT1, REDUCE)                 \
  "v_readfirstlane_b32 s74, %[idxv]\n\t"   /* (loop-carried state comes and goes in vector registers: tied scalar */ \
  "v_readfirstlane_b32 s76, %[pendv]\n\t"  /* operands around the caller's loop do not compile) */ \
  "s_mov_b32 s75, 0\n\t"                                                         \
  "s_getpc_b64 s[78:79]\n"                                                       \
  ".Lta_pc%=:\n\t"                                                               \
  "s_add_u32 s78, s78, .Lta_tab%=-.Lta_pc%=\n\t"                                 \
  "s_addc_u32 s79, s79, 0\n\t"                                                   \
  "s_mov_b32 s81, s79\n\t"                                                       \
  "s_min_u32 s10, s74, %[nitems]\n\t"                                            \
  "s_lshl_b32 s10, s10, 6\n\t"                                                   \
  "s_load_dwordx16 s[16:31], %[progs], s10\n"                                    \
  ".Lta_have%=:\n\t"   /* the program of tape idx is on its way into s[16:31] */ \
  "s_cmp_lt_i32 s74, %[nitems]\n\t"                                           \
  "s_cbranch_scc0 .Lta_done%=\n\t"                                               \
  /* the next list index, by lane 0 through the workgroup's counter: its round trip hides under this tape */ \
  "s_mov_b64 exec, 1\n\t"                                                        \
  "v_mov_b32_e32 v20, %[snext]\n\t"                                              \
  "v_mov_b32_e32 v87, 1\n\t"                                                     \
  "ds_add_rtn_u32 v87, v20, v87\n\t"                                             \
  "s_mov_b64 exec, -1\n\t"                                                       \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_readfirstlane_b32 s75, v87\n\t"                                          \
  "s_cmp_lt_i32 s17, 0\n\t"   /* padding behind the group's last tape */        \
  "s_cbranch_scc1 .Lta_skip%=\n\t"                                               \
  "s_cmp_lt_i32 s16, 0\n\t"   /* bit 31: a tape for this block */               \
  "s_cbranch_scc0 .Lta_generic%=\n\t"                                            \
  /* the list is in cost order: its first tapes run at raised priority (a heavy tape on an even share of its SIMD's */ \
  /* issue slots would end long after the list has drained) */                  \
  "s_cmp_lt_u32 s74, 4\n\t"                                                   \
  "s_cbranch_scc1 .Lta_prio3_%=\n\t"                                                 \
  "s_cmp_lt_u32 s74, 8\n\t"                                                   \
  "s_cbranch_scc1 .Lta_prio2_%=\n\t"                                                 \
  "s_cmp_lt_u32 s74, 16\n\t"                                                  \
  "s_cbranch_scc1 .Lta_prio1_%=\n\t"                                                 \
  "s_setprio 0\n\t"                                                              \
  "s_branch .Lta_priod_%=\n"                                                         \
  ".Lta_prio1_%=:\n\t"                                                               \
  "s_setprio 1\n\t"                                                              \
  "s_branch .Lta_priod_%=\n"                                                         \
  ".Lta_prio2_%=:\n\t"                                                               \
  "s_setprio 2\n\t"                                                              \
  "s_branch .Lta_priod_%=\n"                                                         \
  ".Lta_prio3_%=:\n\t"                                                               \
  "s_setprio 3\n"                                                                \
  ".Lta_priod_%=:\n\t"                                                               \
  "v_mov_b64_e32 v[72:73], 0\n\t"                                                \
  "v_mov_b64_e32 v[74:75], 0\n\t"                                                \
  "v_mov_b64_e32 v[76:77], 0\n\t"                                                \
  "v_mov_b64_e32 v[78:79], 0\n\t"                                                \
  "v_mov_b64_e32 v[80:81], 0\n\t"                                                \
  "v_mov_b64_e32 v[82:83], 0\n\t"                                                \
  "v_mov_b64_e32 v[84:85], 0\n\t"                                                \
  "s_mov_b32 s72, 0\n\t"                                                         \
  "v_mov_b32_e32 v86, %[lc]\n\t"                                                 \
  "s_cmp_eq_u32 %[bps], 0\n\t"   /* (a slice of no blocks at all -- fewer whole blocks than slices: a record of zeros) */ \
  "s_cbranch_scc1 .Lta_reduce%=\n"                                               \
  ".Lta_pass%=:\n\t"                                                             \
  "s_sub_u32 s73, %[bps], s72\n\t"                                               \
  "s_min_u32 s73, s73, 4\n\t"                                                    \
  "s_mov_b64 s[60:61], s[20:21]\n\t"                                             \
  "s_mov_b64 s[62:63], s[22:23]\n\t"                                             \
  "s_mov_b64 s[64:65], s[24:25]\n\t"                                             \
  "s_mov_b64 s[66:67], s[26:27]\n\t"                                             \
  "s_mov_b64 s[68:69], s[28:29]\n\t"                                             \
  "s_mov_b64 s[70:71], s[30:31]\n\t"                                             \
  BSR_TA_SLOT_ADDR                                                               \
  BSR_TA_LOAD4("v[40:43]", "v[44:47]", "v[48:51]", "v[52:55]", "a")              \
  BSR_TA_DISPATCH                                                                \
  /* ---- the operator table: sixteen 128-byte slots, 2 KB-aligned (the dispatch ORs a slot's offset into its address) */ \
  ".p2align 11\n"                                                                \
  ".Lta_tab%=:\n\t"                                                              \
  "s_branch .Lta_addup%=\n\t"   /* 0: end of the tape */                         \
  BSR_TA_SLOT("1") "s_branch .Lta_inv%=\n\t"                                     \
  BSR_TA_SLOT("2") "s_branch .Lta_ln%=\n\t"                                      \
  BSR_TA_SLOT("3") /* neg */                                                     \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_EACH(BSR_TA_NEG1)                                                       \
  BSR_TA_DISPATCH                                                                \
  BSR_TA_SLOT("4") "s_branch .Lta_sin%=\n\t"   /* + 128/pi, pi/128 in three parts, -1/5040, 1/120, -1/6, -1/720 */ \
  BSR_SC_QUAD("0x40445F306DC9C883", "0x3F9921FB54442D18", "0x3C31A62633145C07", "0xB8BF1976B7ED8FBC",                \
              "0xBF2A01A01A01A01A", "0x3F81111111111111", "0xBFC5555555555555", "0xBF56C16C16C16C17")                \
  BSR_TA_SLOT("5") "s_branch .Lta_cos%=\n\t"   /* + 1/24, 2^-26, 2^20 pi/2 (BSR_SINCOS_LIMIT) */ \
  BSR_SC_QUAD("0x3FA5555555555555", "0x3E50000000000000", "0x413921FB00000000", "0", "0", "0", "0", "0")             \
  BSR_TA_SLOT("6") "s_branch .Lta_exp%=\n\t"   /* + 64/ln2, ln2/64 in two parts, 1/720, 1/120, 1/24, 1/6, 710 */ \
  BSR_SC_QUAD("0x40571547652B82FE", "0x3F862E42FEFA39EF", "0x3C1ABC9E3B39803F", "0x3F56C16C16C16C17",                \
              "0x3F81111111111111", "0x3FA5555555555555", "0x3FC5555555555555", "0x4086300000000000")                \
  BSR_TA_SLOT("7") /* square */                                                  \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_EACH(BSR_TA_SQ1)                                                        \
  BSR_TA_DISPATCH                                                                \
  BSR_TA_SLOT("8") "s_branch .Lta_cube%=\n\t"   /* + -760, 200 */               \
  BSR_SC_QUAD("0xC087C00000000000", "0x4069000000000000", "0", "0", "0", "0", "0", "0")                              \
  BSR_TA_SLOT("9") /* saved + acc */                                             \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_EACHS(BSR_TA_SADD1)                                                     \
  BSR_TA_DISPATCH                                                                \
  BSR_TA_SLOT("10") /* saved * acc */                                            \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_EACHS(BSR_TA_SMUL1)                                                     \
  BSR_TA_DISPATCH                                                                \
  BSR_TA_SLOT("11") "s_branch .Lta_term%=\n\t"                                   \
  BSR_TA_SLOT("12") "s_branch .Lta_addt%=\n\t"                                   \
  BSR_TA_SLOT("13") "s_branch .Lta_mult%=\n\t"                                   \
  BSR_TA_SLOT("14") /* saved - acc */                                            \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_EACHS(BSR_TA_SSUB1)                                                     \
  BSR_TA_DISPATCH                                                                \
  BSR_TA_SLOT("15") "s_branch .Lta_div%=\n\t"                                    \
  ".p2align 7\n"                                                                 \
  ".Lta_term%=:\n\t"   /* a terminal: the accumulator becomes the saved value (the host packs tapes that hold one at most) */ \
  BSR_TA_SLOT_ADDR                                                               \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_EACHS(BSR_TA_SAVE1)                                                     \
  BSR_TA_LOAD4("v[40:43]", "v[44:47]", "v[48:51]", "v[52:55]", "t")              \
  BSR_TA_DISPATCH                                                                \
  ".Lta_div%=:\n\t"   /* saved / acc, protected like inv */                     \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_EACHS(BSR_TA_SDIV1)                                                     \
  BSR_TA_DISPATCH                                                                \
  ".Lta_addt%=:\n\t"                                                             \
  BSR_TA_BIN_T(BSR_TA_ADD2, "b")                                                 \
  ".Lta_mult%=:\n\t"                                                             \
  BSR_TA_BIN_T(BSR_TA_MUL2, "c")                                                 \
  ".Lta_ln%=:\n\t"   /* ln: a x + b, two roundings; the pair consumed moves out, the next one up */ \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_EACH(BSR_TA_LN1)                                                        \
  BSR_TA_EACH(BSR_TA_LN2)                                                        \
  "s_mov_b64 s[64:65], s[68:69]\n\t"                                             \
  "s_mov_b64 s[66:67], s[70:71]\n\t"                                             \
  BSR_TA_DISPATCH                                                                \
  ".Lta_inv%=:\n\t"                                                              \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_EACH(BSR_TA_INV1)                                                       \
  BSR_TA_DISPATCH                                                                \
  ".Lta_cube%=:\n\t"                                                             \
  "s_movk_i32 s10, 0x1f8\n\t"   /* the finite classes */                         \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_EACH(BSR_SC_CUBE)                                                       \
  BSR_TA_DISPATCH                                                                \
  ".Lta_sin%=:\n\t"                                                              \
  "s_load_dwordx16 s[36:51], s[78:79], 576\n\t"                                  \
  "s_load_dwordx8 s[52:59], s[78:79], 704\n\t"                                   \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_BIG_CHECK                                                               \
  BSR_TA_EACH(BSR_TA_SIN1)                                                       \
  BSR_TA_DISPATCH                                                                \
  ".Lta_cos%=:\n\t"                                                              \
  "s_load_dwordx16 s[36:51], s[78:79], 576\n\t"                                  \
  "s_load_dwordx8 s[52:59], s[78:79], 704\n\t"                                   \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_TA_BIG_CHECK                                                               \
  BSR_TA_EACH(BSR_TA_COS1)                                                       \
  BSR_TA_DISPATCH                                                                \
  ".Lta_exp%=:\n\t"                                                              \
  "s_load_dwordx16 s[36:51], s[78:79], 832\n\t"                                  \
  "s_load_dwordx4 s[52:55], s[78:79], 1088\n\t"                                  \
  "s_waitcnt lgkmcnt(0)\n\t"   /* (in front of the constant: no LDS read of this block is in flight into v[30:31] here, but the order costs nothing -- bsr_stream_chunk_asm.h had the race) */ \
  "v_mov_b32_e32 v30, 0x20000000\n\t"   /* 1e10: what the clipped exp returns beyond 200 (and for NaN) */ \
  "v_mov_b32_e32 v31, 0x4202a05f\n\t"                                            \
  BSR_TA_EACH(BSR_SC_EXP)                                                        \
  BSR_TA_DISPATCH                                                                \
  /* ---- end of the entries: the pass's rows into the sums, then the next pass or the record */ \
  ".Lta_addup%=:\n\t"                                                            \
  BSR_TA_ADDUP(ROW, RDX, RDY, QADDR, WAIT1)                                      \
  "s_add_u32 s72, s72, 4\n\t"                                                    \
  "s_cmp_lt_u32 s72, %[bps]\n\t"                                                 \
  "s_cbranch_scc0 .Lta_reduce%=\n\t"                                             \
  "v_add_u32_e32 v86, 0x1000, v86\n\t"                                           \
  "s_cmp_eq_u32 s76, 0\n\t"                                                  \
  "s_cbranch_scc1 .Lta_pass%=\n\t"                                               \
  /* the second half of the slice was still travelling when the wave began (bsr_tile_asm.hip): its copies, then everyone's */ \
  "s_waitcnt vmcnt(0)\n\t"                                                       \
  "s_barrier\n\t"                                                                \
  "s_mov_b32 s76, 0\n\t"                                                     \
  "s_branch .Lta_pass%=\n"                                                       \
  ".Lta_reduce%=:\n\t"                                                           \
  REDUCE                                                                         \
  "s_mov_b32 s74, s75\n\t"                                                 \
  "s_branch .Lta_have%=\n"                                                       \
  ".Lta_skip%=:\n\t"                                                             \
  "s_mov_b32 s74, s75\n\t"                                                       \
  "s_min_u32 s10, s74, %[nitems]\n\t"                                            \
  "s_lshl_b32 s10, s10, 6\n\t"                                                   \
  "s_load_dwordx16 s[16:31], %[progs], s10\n\t"                                  \
  "s_branch .Lta_have%=\n"                                                       \
  ".Lta_generic%=:\n\t"                                                          \
  "s_mov_b32 %[st], 1\n\t"                                                       \
  "s_branch .Lta_exit%=\n"                                                       \
  ".Lta_done%=:\n\t"                                                             \
  "s_mov_b32 %[st], 0\n"                                                         \
  ".Lta_exit%=:\n\t"                                                             \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "s_setprio 0\n\t"                                                              \
  "v_mov_b32_e32 %[idxv], s74\n\t"                                               \
  "v_mov_b32_e32 %[nxtv], s75\n\t"                                               \
  "v_mov_b32_e32 %[pendv], s76\n\t"

#define BSR_TILE_TAPES_ASM_K1 BSR_TILE_TAPES_ASM_(BSR_TA_ROW_K1, BSR_TA_RDX_K1, BSR_TA_RDY_K1, BSR_TA_QADDR_K1, BSR_TA_WAIT1_K1, BSR_TA_REDUCE_K1)
#define BSR_TILE_TAPES_ASM_K2 BSR_TILE_TAPES_ASM_(BSR_TA_ROW_K2, BSR_TA_RDX_K2, BSR_TA_RDY_K2, BSR_TA_QADDR_K2, BSR_TA_WAIT1_K2, BSR_TA_REDUCE_K2)
#define BSR_TILE_TAPES_ASM_K3 BSR_TILE_TAPES_ASM_(BSR_TA_ROW_K3, BSR_TA_RDX_K3, BSR_TA_RDY_K3, BSR_TA_QADDR_K3, BSR_TA_WAIT1_K3, BSR_TA_REDUCE_K3)
#define BSR_TILE_TAPES_ASM_K4 BSR_TILE_TAPES_ASM_(BSR_TA_ROW_K4, BSR_TA_RDX_K4, BSR_TA_RDY_K4, BSR_TA_QADDR_K4, BSR_TA_WAIT1_K4, BSR_TA_REDUCE_K4)

#define BSR_TILE_TAPES_CLOBBERS                                                                                        \
  "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17",   \
  "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", \
  "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", \
  "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", \
  "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", \
  "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101",     \
  "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",                                                                     \
  "s8", "s9", "s10", "s11", "s12", "s13", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26",  \
  "s27", "s28", "s29", "s30", "s31", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", \
  "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", \
  "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s78", "s79", "s80", "s81", "vcc", "scc", "memory"
// clang-format on
