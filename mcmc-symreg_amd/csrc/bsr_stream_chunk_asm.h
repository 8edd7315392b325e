// One chunk (128 rows) of the streaming row pass for ONE WAVE -- its four tapes: programs fetched, interpreted, rows added
// into the four sets of sums -- as one block of gfx950 assembly (K = 3: three basis columns; bsr_stream.hip: k_stream).
//
// bsr_stream_asm.h took the per-ENTRY scalar cost from 17 instructions to 5; what was left was the per-TAPE cost around
// it: 46 scalar instructions for a leaf tape in the compiler's loop (address arithmetic for the program's prefetch, two
// flag tests, copies of the eight program words, the select of the set of sums by a compare chain, the loop itself,
// entering and leaving the interpreter).  Here the four tapes are unrolled by the preprocessor: a program is ONE scalar
// load at a literal offset straight into the registers the interpreter shifts; the sums of tape q are operands named in
// the text; `end of tape` is a jump through a register to the tape's own add-up code.  Per tape: 14 scalar instructions.
//
// Layout of a program (StreamRec, bsr_internal.h) in s[16:23]: s16 meta (bit 31: the fast interpreter takes the tape,
// bit 6: there is a tape), s17 byte offset of the leading terminal's column in the chunk buffer, s[18:19] prescale,
// s[20:21] entries (4 bits each: operator + 1, 0 = end), s[22:23] slots of the terminals behind the first (8 bits each).
//
// Leaving and coming back: sin, cos, exp (C++ routines of bsr_device.h) and tapes for the general stack machine leave the
// block with `st` = code | tag << 4 (code 4, 5, 6: operator + 1, values in z0, z1, interpreter state in s0, s1, sv0..4;
// code 1: tape `tag` is the caller's to evaluate); the caller passes the same word back in `resume` with the new values
// in z0, z1.  st = 0: the chunk is done.
//
// Fixed registers, all caller-saved in the calling convention (the kernel's few calls leave the sums, which live in
// callee-saved registers, alone): v[0:3] accumulator, v[4:7] saved value, v[8:11] operand, v[12:19] temporaries, v20
// address, v[24:27] y, v[28:39] basis columns; s[8:9] scratch destination, s10, s12 temporaries, s13 ln pair address,
// s[14:15] prescale of the tape being added up (its program registers already receive the next tape's), s[16:23]
// program, s[24:25] operator table, s[26:27] jump target, s[28:29] where `end` goes.
#pragma once
#include "bsr_stream_asm.h"

// clang-format off
#define BSR_SC_DISPATCH                          \
  "s_lshl_b32 s10, s20, 7\n\t"                   \
  "s_and_b32 s10, s10, 0x780\n\t"                \
  "s_lshr_b64 s[20:21], s[20:21], 4\n\t"         \
  "s_or_b32 s26, s24, s10\n\t"                   \
  "s_setpc_b64 s[26:27]\n\t"

#define BSR_SC_SLOT_ADDR(SH)                     \
  "s_and_b32 s10, s22, 0xff\n\t"                 \
  "s_lshr_b64 s[22:23], s[22:23], 8\n\t"         \
  "v_lshl_add_u32 v20, s10, " SH ", %[lc]\n\t"

// The lane's two rows of a column of the staged chunk (address in v20) into the accumulator / the operand.  f64 storage:
// one ds_read_b128, left in flight (every operator starts by waiting for it).  f32 storage (round 6: columns kept as f32
// in HBM and LDS, the arithmetic stays f64): one ds_read_b64 into temporaries that are free wherever a terminal is read,
// waited for, converted -- v_cvt_f64_f32 is exact, so a row's value depends on its stored f32 bits alone.
#define BSR_SC_LDA_F64 "ds_read_b128 v[0:3], v20\n\t"
#define BSR_SC_LDO_F64 "ds_read_b128 v[8:11], v20\n\t" "s_waitcnt lgkmcnt(0)\n\t"
#define BSR_SC_LDA_F32 "ds_read_b64 v[12:13], v20\n\t" "s_waitcnt lgkmcnt(0)\n\t" \
  "v_cvt_f64_f32_e32 v[0:1], v12\n\t" "v_cvt_f64_f32_e32 v[2:3], v13\n\t"
#define BSR_SC_LDO_F32 "ds_read_b64 v[12:13], v20\n\t" "s_waitcnt lgkmcnt(0)\n\t" \
  "v_cvt_f64_f32_e32 v[8:9], v12\n\t" "v_cvt_f64_f32_e32 v[10:11], v13\n\t"
// (round 6: a SECOND value below the accumulator, v[40:43], where the block has them to spare -- K <= 3: a pushed terminal moves
// the saved value down first (PUSH2), a binary operator brings it back up behind itself (POP2); trees of Strahler number 3
// -- (a + b) * (c + d) below another binary operator -- no longer leave for the stack machine.  K >= 4: both empty.)
#define BSR_SC_PUSH2 "v_mov_b64_e32 v[40:41], v[4:5]\n\t" "v_mov_b64_e32 v[42:43], v[6:7]\n\t"
#define BSR_SC_POP2 "v_mov_b64_e32 v[4:5], v[40:41]\n\t" "v_mov_b64_e32 v[6:7], v[42:43]\n\t"
#define BSR_SC_NONE ""
#define BSR_SC_BIN(ins, neg, POP)                                                \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  ins " v[0:1], v[4:5], " neg "v[0:1]\n\t"                                       \
  ins " v[2:3], v[6:7], " neg "v[2:3]\n\t"                                       \
  POP                                                                            \
  BSR_SC_DISPATCH
#define BSR_SC_BIN_T(ins, SH, LDO)                                               \
  BSR_SC_SLOT_ADDR(SH)                                                             \
  LDO                                                                            \
  ins " v[0:1], v[0:1], v[8:9]\n\t"                                              \
  ins " v[2:3], v[2:3], v[10:11]\n\t"                                            \
  BSR_SC_DISPATCH
#define BSR_SC_LEAVE(code)                                                       \
  "s_movk_i32 s12, " code "\n\t"                                                 \
  "s_branch .Lsc_leave%=\n\t"
#define BSR_SC_SLOT(n) ".p2align 7\n.Lsc_op" n "_%=:\n\t"

// division and cube of bsr_stream_asm.h on this block's registers (s[8:9]: v_div_scale's unused scalar result; s10: the
// class mask)
#define BSR_SC_DIV(x, xlo, xhi, num)                                   \
  "v_div_scale_f64 v[12:13], s[8:9], " x ", " x ", " num "\n\t"        \
  "v_rcp_f64_e32 v[14:15], v[12:13]\n\t"                               \
  "v_div_scale_f64 v[16:17], vcc, " num ", " x ", " num "\n\t"         \
  "v_fma_f64 v[18:19], -v[12:13], v[14:15], 1.0\n\t"                   \
  "v_fmac_f64_e32 v[14:15], v[14:15], v[18:19]\n\t"                    \
  "v_fma_f64 v[18:19], -v[12:13], v[14:15], 1.0\n\t"                   \
  "v_fmac_f64_e32 v[14:15], v[14:15], v[18:19]\n\t"                    \
  "v_mul_f64 v[18:19], v[16:17], v[14:15]\n\t"                         \
  "v_fma_f64 v[12:13], -v[12:13], v[18:19], v[16:17]\n\t"              \
  "v_div_fmas_f64 v[12:13], v[12:13], v[14:15], v[18:19]\n\t"          \
  "v_div_fixup_f64 v[12:13], v[12:13], " x ", " num "\n\t"             \
  "v_cmp_neq_f64_e32 vcc, 0, " x "\n\t"                                \
  "s_nop 1\n\t"                                                        \
  "v_cndmask_b32_e32 " xhi ", 0, v13, vcc\n\t"                         \
  "v_cndmask_b32_e32 " xlo ", 0, v12, vcc\n\t"
#define BSR_SC_CUBE(x, xlo, xhi)                                       \
  "v_mul_f64 v[12:13], " x ", " x "\n\t"                               \
  "v_fma_f64 v[14:15], " x ", " x ", -v[12:13]\n\t"                    \
  "v_mul_f64 v[16:17], " x ", v[12:13]\n\t"                            \
  "v_fma_f64 v[18:19], v[12:13], " x ", -v[16:17]\n\t"                 \
  "v_mul_f64 v[14:15], " x ", v[14:15]\n\t"                            \
  "v_add_f64 v[14:15], v[18:19], v[14:15]\n\t"                         \
  "v_add_f64 v[14:15], v[16:17], v[14:15]\n\t"                         \
  "v_cmp_class_f64_e64 vcc, v[14:15], s10\n\t"                         \
  "s_nop 1\n\t"                                                        \
  "v_cndmask_b32_e32 " xhi ", v17, v15, vcc\n\t"                       \
  "v_cndmask_b32_e32 " xlo ", v16, v14, vcc\n\t"


// sin / cos / exp of bsr_fastmath.h (bsr_sincos, bsr_exp behind op_exp of bsr_device.h) for one value, operation by
// operation in the order of the C++ (every one a correctly rounded IEEE operation: the same bits).  Constants: read by
// scalar loads from the table itself -- the operator slots that only hold a branch carry 64 bytes of data behind it --
// into s[36:59]; an instruction takes ONE scalar operand, so the second constant of a fused multiply-add goes through
// a vector register first, as in the compiler's code.  Temporaries: v[8:19], v[22:39] -- the y and basis values are
// read again behind the operator.  %[tab]: LDS address of the tables (bsr_tables.h layout).
//   sincos constants: s[36:37] 128/pi, s[38:43] pi/128 in three parts, s[44:49] -1/5040, 1/120, -1/6, s[50:53] -1/720, 1/24,
//                     s[54:55] 2^-26, s[56:57] the limit of the fast path
#define BSR_SC_SINCOS(x, xlo, xhi, joff, tiny)                          \
  "v_mul_f64 v[8:9], " x ", s[36:37]\n\t"                               \
  "v_rndne_f64_e32 v[8:9], v[8:9]\n\t"                                  \
  "v_fma_f64 v[10:11], -v[8:9], s[38:39], " x "\n\t"                    \
  "v_mul_f64 v[12:13], v[8:9], s[40:41]\n\t"                            \
  "v_fma_f64 v[14:15], v[8:9], s[40:41], -v[12:13]\n\t"                 \
  "v_add_f64 v[16:17], v[10:11], -v[12:13]\n\t"       /* r */           \
  "v_add_f64 v[18:19], v[10:11], -v[16:17]\n\t"                         \
  "v_add_f64 v[18:19], v[18:19], -v[12:13]\n\t"       /* e */           \
  "v_add_f64 v[18:19], v[18:19], -v[14:15]\n\t"                         \
  "v_fma_f64 v[18:19], -v[8:9], s[42:43], v[18:19]\n\t" /* rl */        \
  "v_cvt_i32_f64_e32 v20, v[8:9]\n\t"                                   \
  joff                                                                  \
  "v_and_b32_e32 v20, 0xff, v20\n\t"                                    \
  "v_lshl_add_u32 v20, v20, 5, %[tab]\n\t"                              \
  "ds_read_b128 v[32:35], v20\n\t"                    /* S, Sl */       \
  "ds_read_b128 v[36:39], v20 offset:16\n\t"          /* C, Cl */       \
  "v_mul_f64 v[22:23], v[16:17], v[16:17]\n\t"        /* z */           \
  "v_mov_b64_e32 v[24:25], s[46:47]\n\t"                                \
  "v_fma_f64 v[24:25], v[22:23], s[44:45], v[24:25]\n\t"                \
  "v_fma_f64 v[24:25], v[22:23], v[24:25], s[48:49]\n\t"                \
  "v_mul_f64 v[24:25], v[22:23], v[24:25]\n\t"        /* ps */          \
  "v_mov_b64_e32 v[26:27], s[52:53]\n\t"                                \
  "v_fma_f64 v[26:27], v[22:23], s[50:51], v[26:27]\n\t"                \
  "v_fma_f64 v[26:27], v[22:23], v[26:27], -0.5\n\t"                    \
  "v_mul_f64 v[26:27], v[22:23], v[26:27]\n\t"        /* pc */          \
  "v_mul_f64 v[30:31], v[16:17], v[24:25]\n\t"        /* r ps */        \
  "s_waitcnt lgkmcnt(0)\n\t"                                            \
  "v_mul_f64 v[28:29], v[32:33], v[26:27]\n\t"        /* u = S pc */    \
  "v_fma_f64 v[28:29], v[36:37], v[30:31], v[28:29]\n\t"                \
  "v_add_f64 v[28:29], v[28:29], v[34:35]\n\t"                          \
  "v_fma_f64 v[28:29], v[38:39], v[16:17], v[28:29]\n\t"                \
  "v_fma_f64 v[28:29], v[36:37], v[18:19], v[28:29]\n\t"                \
  "v_fma_f64 v[28:29], v[36:37], v[16:17], v[28:29]\n\t"                \
  "v_add_f64 v[28:29], v[32:33], v[28:29]\n\t"                          \
  tiny
// (sin of a tiny argument is the argument itself)
#define BSR_SC_SIN_TINY(x, xlo, xhi)                                    \
  "v_cmp_lt_f64_e64 vcc, |" x "|, s[54:55]\n\t"                         \
  "s_nop 1\n\t"                                                         \
  "v_cndmask_b32_e32 " xlo ", v28, " xlo ", vcc\n\t"                    \
  "v_cndmask_b32_e32 " xhi ", v29, " xhi ", vcc\n\t"
#define BSR_SC_COS_MOVE(x)                                              \
  "v_mov_b64_e32 " x ", v[28:29]\n\t"
//   exp constants: s[36:37] 64/ln2, s[38:41] ln2/64 in two parts, s[42:49] 1/720, 1/120, 1/24, 1/6, s[50:51] 710,
//                  s[52:53] -760, s[54:55] 200
#define BSR_SC_EXP(x, xlo, xhi)                                         \
  "v_min_f64 v[10:11], " x ", s[50:51]\n\t"                             \
  "v_max_f64 v[10:11], v[10:11], s[52:53]\n\t"                          \
  "v_mul_f64 v[8:9], v[10:11], s[36:37]\n\t"                            \
  "v_rndne_f64_e32 v[8:9], v[8:9]\n\t"                                  \
  "v_fma_f64 v[10:11], -v[8:9], s[38:39], v[10:11]\n\t"                 \
  "v_fma_f64 v[10:11], -v[8:9], s[40:41], v[10:11]\n\t" /* r */         \
  "v_cvt_i32_f64_e32 v20, v[8:9]\n\t"                                   \
  "v_and_b32_e32 v22, 63, v20\n\t"                                      \
  "v_lshl_add_u32 v22, v22, 4, %[tab]\n\t"                              \
  "ds_read_b128 v[32:35], v22 offset:8192\n\t"        /* T, Tl */       \
  "v_mov_b64_e32 v[12:13], s[44:45]\n\t"                                \
  "v_fma_f64 v[12:13], v[10:11], s[42:43], v[12:13]\n\t"                \
  "v_fma_f64 v[12:13], v[10:11], v[12:13], s[46:47]\n\t"                \
  "v_fma_f64 v[12:13], v[10:11], v[12:13], s[48:49]\n\t"                \
  "v_fma_f64 v[12:13], v[10:11], v[12:13], 0.5\n\t"   /* q */           \
  "v_mul_f64 v[14:15], v[10:11], v[10:11]\n\t"                          \
  "v_fma_f64 v[14:15], v[14:15], v[12:13], v[10:11]\n\t" /* p */        \
  "v_ashrrev_i32_e32 v20, 6, v20\n\t"                                   \
  "s_waitcnt lgkmcnt(0)\n\t"                                            \
  "v_fma_f64 v[14:15], v[32:33], v[14:15], v[34:35]\n\t"                \
  "v_add_f64 v[14:15], v[32:33], v[14:15]\n\t"        /* m */           \
  "v_ldexp_f64 v[14:15], v[14:15], v20\n\t"                             \
  "v_cmp_le_f64_e64 vcc, " x ", s[54:55]\n\t"                           \
  "s_nop 1\n\t"                                                         \
  "v_cndmask_b32_e32 " xlo ", v30, v14, vcc\n\t"   /* (v[30:31]: 1e10 -- a literal next to vcc is one scalar operand too many) */ \
  "v_cndmask_b32_e32 " xhi ", v31, v15, vcc\n\t"
// y and the K basis columns of the lane's two rows (v[24:27], v[28:31] ...): read once per chunk for the four tapes, and
// again behind sin / cos / exp, which use their registers
#define BSR_SC_YQ_HEAD                                                  \
  "v_add_u32_e32 v20, %[yo], %[lc]\n\t"                                 \
  "ds_read_b128 v[24:27], v20\n\t"
#define BSR_SC_YQ1(O1) BSR_SC_YQ_HEAD "ds_read_b128 v[28:31], v20 offset:" O1 "\n\t"
#define BSR_SC_YQ2(O1, O2) BSR_SC_YQ1(O1) "ds_read_b128 v[32:35], v20 offset:" O2 "\n\t"
#define BSR_SC_YQ3(O1, O2, O3) BSR_SC_YQ2(O1, O2) "ds_read_b128 v[36:39], v20 offset:" O3 "\n\t"
#define BSR_SC_YQ4(O1, O2, O3, O4) BSR_SC_YQ3(O1, O2, O3) "ds_read_b128 v[40:43], v20 offset:" O4 "\n\t"
#define BSR_SC_YQ5(O1, O2, O3, O4, O5) BSR_SC_YQ4(O1, O2, O3, O4) "ds_read_b128 v[44:47], v20 offset:" O5 "\n\t"
#define BSR_SC_YQ6(O1, O2, O3, O4, O5, O6) BSR_SC_YQ5(O1, O2, O3, O4, O5) "ds_read_b128 v[48:51], v20 offset:" O6 "\n\t"
#define BSR_SC_YQ7(O1, O2, O3, O4, O5, O6, O7) BSR_SC_YQ6(O1, O2, O3, O4, O5, O6) "ds_read_b128 v[52:55], v20 offset:" O7 "\n\t"
#define BSR_SC_YQ8(O1, O2, O3, O4, O5, O6, O7, O8) BSR_SC_YQ7(O1, O2, O3, O4, O5, O6, O7) "ds_read_b128 v[56:59], v20 offset:" O8 "\n\t"
#define BSR_SC_QUAD(a, b, c, d, e, f, g, h)                             \
  ".p2align 6\n\t.quad " a ", " b ", " c ", " d ", " e ", " f ", " g ", " h "\n\t"

// where `end` goes for tape q (its add-up code, in front of the table), and the tape's ln pairs
#define BSR_SC_TAPE_REGS(q)                                                      \
  "s_sub_u32 s28, s24, .Lsc_tab%=-.Lsc_acc" #q "_%=\n\t"                         \
  "s_subb_u32 s29, s25, 0\n\t"                                                   \
  "s_add_u32 s13, %[ln], " #q "*48\n\t"   /* (BSR_STREAM_LN_PAIRS = 3 pairs of 16 bytes per tape) */

// Tape q of the wave: its program has been requested into s[16:23] (by the block's entry, or while tape q - 1 was added up).
// Its priority is 3 - q: a SIMD issues for its oldest wave first, so left alone its four waves finish a chunk one after
// the other and the youngest runs the end by itself, with nothing to hide its latencies behind; a wave that has moved on
// to its next tape now yields to the ones still on an earlier tape, and they reach the barrier together (C5: +4 %,
// interleaved A/B of two builds in one box).
// the sums' lines for basis columns 2..4 of row 0 / row 1 (K of them exist: BSR_SC_FNO where not)
#define BSR_SC_FB0(q) "v_fmac_f64_e32 %[cb" #q "], v[32:33], v[12:13]\n\t"
#define BSR_SC_FB1(q) "v_fmac_f64_e32 %[cb" #q "], v[34:35], v[16:17]\n\t"
#define BSR_SC_FC0(q) "v_fmac_f64_e32 %[cc" #q "], v[36:37], v[12:13]\n\t"
#define BSR_SC_FC1(q) "v_fmac_f64_e32 %[cc" #q "], v[38:39], v[16:17]\n\t"
#define BSR_SC_FD0(q) "v_fmac_f64_e32 %[cd" #q "], v[40:41], v[12:13]\n\t"
#define BSR_SC_FD1(q) "v_fmac_f64_e32 %[cd" #q "], v[42:43], v[16:17]\n\t"
#define BSR_SC_FE0(q) "v_fmac_f64_e32 %[ce" #q "], v[44:45], v[12:13]\n\t"
#define BSR_SC_FE1(q) "v_fmac_f64_e32 %[ce" #q "], v[46:47], v[16:17]\n\t"
#define BSR_SC_FF0(q) "v_fmac_f64_e32 %[cf" #q "], v[48:49], v[12:13]\n\t"
#define BSR_SC_FF1(q) "v_fmac_f64_e32 %[cf" #q "], v[50:51], v[16:17]\n\t"
#define BSR_SC_FG0(q) "v_fmac_f64_e32 %[cg" #q "], v[52:53], v[12:13]\n\t"
#define BSR_SC_FG1(q) "v_fmac_f64_e32 %[cg" #q "], v[54:55], v[16:17]\n\t"
#define BSR_SC_FH0(q) "v_fmac_f64_e32 %[ch" #q "], v[56:57], v[12:13]\n\t"
#define BSR_SC_FH1(q) "v_fmac_f64_e32 %[ch" #q "], v[58:59], v[16:17]\n\t"
// rows 0 and 1 of basis columns 2..K of tape q
#define BSR_SC_R0_1(q) ""
#define BSR_SC_R1_1(q) ""
#define BSR_SC_R0_2(q) BSR_SC_FB0(q)
#define BSR_SC_R1_2(q) BSR_SC_FB1(q)
#define BSR_SC_R0_3(q) BSR_SC_R0_2(q) BSR_SC_FC0(q)
#define BSR_SC_R1_3(q) BSR_SC_R1_2(q) BSR_SC_FC1(q)
#define BSR_SC_R0_4(q) BSR_SC_R0_3(q) BSR_SC_FD0(q)
#define BSR_SC_R1_4(q) BSR_SC_R1_3(q) BSR_SC_FD1(q)
#define BSR_SC_R0_5(q) BSR_SC_R0_4(q) BSR_SC_FE0(q)
#define BSR_SC_R1_5(q) BSR_SC_R1_4(q) BSR_SC_FE1(q)
#define BSR_SC_R0_6(q) BSR_SC_R0_5(q) BSR_SC_FF0(q)
#define BSR_SC_R1_6(q) BSR_SC_R1_5(q) BSR_SC_FF1(q)
#define BSR_SC_R0_7(q) BSR_SC_R0_6(q) BSR_SC_FG0(q)
#define BSR_SC_R1_7(q) BSR_SC_R1_6(q) BSR_SC_FG1(q)
#define BSR_SC_R0_8(q) BSR_SC_R0_7(q) BSR_SC_FH0(q)
#define BSR_SC_R1_8(q) BSR_SC_R1_7(q) BSR_SC_FH1(q)

#define BSR_SC_TAPE(q, qnext, R0, R1, LDA)                                       \
  "s_setprio 3-" #q "\n\t"   /* the waves that are behind go first (below) */   \
  BSR_SC_TAPE_REGS(q)                                                            \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "s_cmp_lt_i32 s16, 0\n\t"                                                      \
  "s_cbranch_scc0 .Lsc_slow" #q "_%=\n\t"                                        \
  ".Lsc_go" #q "_%=:\n\t"                                                        \
  "v_add_u32_e32 v20, s17, %[lc]\n\t"                                            \
  LDA                                                                            \
  BSR_SC_DISPATCH                                                                \
  ".Lsc_slow" #q "_%=:\n\t"   /* no tape in this set of sums, one for the stack machine -- or (bit 30, round 6) one the block takes \
                                 after all: a program of several words and / or a second value below the accumulator */ \
  "s_bitcmp1_b32 s16, 30\n\t"                                                    \
  "s_cbranch_scc0 .Lsc_nodeep" #q "_%=\n\t"                                      \
  "s_and_b32 s10, s16, 0xf0000\n\t"                                              \
  "s_cbranch_scc0 .Lsc_go" #q "_%=\n\t"   /* (one word: `end` is the tape's) */  \
  "s_mov_b64 s[14:15], s[28:29]\n\t"                                             \
  "s_add_u32 s28, s24, .Lsc_more%=-.Lsc_tab%=\n\t"                               \
  "s_addc_u32 s29, s25, 0\n\t"                                                   \
  "s_branch .Lsc_go" #q "_%=\n"                                                   \
  ".Lsc_nodeep" #q "_%=:\n\t"                                                    \
  "s_bitcmp1_b32 s16, 6\n\t"                                                     \
  "s_cbranch_scc1 .Lsc_gen" #q "_%=\n\t"                                         \
  "s_load_dwordx8 s[16:23], %[sr], " #qnext "*32\n\t"                            \
  "s_branch .Lsc_next" #q "_%=\n"                                                \
  ".Lsc_gen" #q "_%=:\n\t"                                                       \
  "s_movk_i32 %[st], " #q "*16+1\n\t"                                            \
  "s_branch .Lsc_exit%=\n"                                                       \
  ".Lsc_acc" #q "_%=:\n\t"   /* the tape's two rows of the chunk into its sums: the order of add_chunk (bsr_stream.hip) */ \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "s_mov_b64 s[14:15], s[18:19]\n\t"                                             \
  "s_load_dwordx8 s[16:23], %[sr], " #qnext "*32\n\t"                            \
  "v_mul_f64 v[12:13], v[0:1], s[14:15]\n\t"                                     \
  "v_mul_f64 v[16:17], v[2:3], s[14:15]\n\t"                                     \
  "v_max_f64 v[14:15], %[am" #q "], |v[0:1]|\n\t"                                \
  "v_fmac_f64_e32 %[sa" #q "], v[12:13], v[12:13]\n\t"                           \
  "v_fmac_f64_e32 %[sb" #q "], v[12:13], v[24:25]\n\t"                           \
  "v_fmac_f64_e32 %[ca" #q "], v[28:29], v[12:13]\n\t"                           \
  R0(q)                                                                          \
  "v_max_f64 %[am" #q "], v[14:15], |v[2:3]|\n\t"                                \
  "v_fmac_f64_e32 %[sa" #q "], v[16:17], v[16:17]\n\t"                           \
  "v_fmac_f64_e32 %[sb" #q "], v[16:17], v[26:27]\n\t"                           \
  "v_fmac_f64_e32 %[ca" #q "], v[30:31], v[16:17]\n\t"                           \
  R1(q)                                                                          \
  ".Lsc_next" #q "_%=:\n\t"

// coming back into tape q: its program again (prescale, flags), then either the caller's values straight to the sums
// (code 1) or the interpreter's state from the operands
#define BSR_SC_RESUME(q)                                                         \
  ".Lsc_res" #q "_%=:\n\t"                                                       \
  "s_load_dwordx8 s[16:23], %[sr], " #q "*32\n\t"                                \
  BSR_SC_TAPE_REGS(q)                                                            \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "s_lshr_b32 s10, %[resume], 8\n\t"   /* extension words left and the next one's place, as the leave found them */ \
  "s_lshl_b32 s10, s10, 16\n\t"                                                  \
  "s_and_b32 s16, s16, 0xc000ffff\n\t"                                           \
  "s_or_b32 s16, s16, s10\n\t"                                                   \
  "s_and_b32 s10, s16, 0xf0000\n\t"                                              \
  "s_cbranch_scc0 .Lsc_resp" #q "_%=\n\t"                                        \
  "s_mov_b64 s[14:15], s[28:29]\n\t"   /* (words left: `end` goes to .Lsc_more, the last one's to the add-up) */ \
  "s_add_u32 s28, s24, .Lsc_more%=-.Lsc_tab%=\n\t"                               \
  "s_addc_u32 s29, s25, 0\n"                                                      \
  ".Lsc_resp" #q "_%=:\n\t"                                                      \
  "v_mov_b64_e32 v[0:1], %[z0]\n\t"                                              \
  "v_mov_b64_e32 v[2:3], %[z1]\n\t"                                              \
  "s_cmp_eq_u32 s12, 1\n\t"                                                      \
  "s_cbranch_scc1 .Lsc_acc" #q "_%=\n\t"                                         \
  "s_branch .Lsc_state%=\n\t"

// -- shared pieces of the two blocks below --
#define BSR_SC_TABLE_BASE                                                        \
  "s_getpc_b64 s[24:25]\n"                                                       \
  ".Lsc_pc%=:\n\t"                                                               \
  "s_add_u32 s24, s24, .Lsc_tab%=-.Lsc_pc%=\n\t"                                 \
  "s_addc_u32 s25, s25, 0\n\t"                                                   \
  "s_mov_b32 s27, s25\n\t"
// four tapes per wave (K <= 4), or two (K >= 5: a set of sums is 8 + 3 registers pairs; the labels of tapes 2 and 3 that
// the shared resume / leave code names then stand behind the last tape and are never reached)
#define BSR_SC_TAPES4_(R0, R1, LDA) BSR_SC_TAPE(0, 1, R0, R1, LDA) BSR_SC_TAPE(1, 2, R0, R1, LDA) BSR_SC_TAPE(2, 3, R0, R1, LDA) BSR_SC_TAPE(3, 4, R0, R1, LDA)
#define BSR_SC_TAPES4(R0, R1) BSR_SC_TAPES4_(R0, R1, BSR_SC_LDA_F64)
#define BSR_SC_TAPES2(R0, R1) BSR_SC_TAPE(0, 1, R0, R1, BSR_SC_LDA_F64) BSR_SC_TAPE(1, 2, R0, R1, BSR_SC_LDA_F64) ".Lsc_acc2_%=:\n.Lsc_acc3_%=:\n\t"
#define BSR_SC_TAPES_1 BSR_SC_TAPES4(BSR_SC_R0_1, BSR_SC_R1_1)
#define BSR_SC_TAPES_2 BSR_SC_TAPES4(BSR_SC_R0_2, BSR_SC_R1_2)
#define BSR_SC_TAPES_3 BSR_SC_TAPES4(BSR_SC_R0_3, BSR_SC_R1_3)
#define BSR_SC_TAPES_4 BSR_SC_TAPES4(BSR_SC_R0_4, BSR_SC_R1_4)
#define BSR_SC_TAPES_5 BSR_SC_TAPES2(BSR_SC_R0_5, BSR_SC_R1_5)
#define BSR_SC_TAPES_6 BSR_SC_TAPES2(BSR_SC_R0_6, BSR_SC_R1_6)
#define BSR_SC_TAPES_7 BSR_SC_TAPES2(BSR_SC_R0_7, BSR_SC_R1_7)
#define BSR_SC_TAPES_8 BSR_SC_TAPES2(BSR_SC_R0_8, BSR_SC_R1_8)
#define BSR_SC_RESUME_PART BSR_SC_RESUME_PART_(BSR_SC_NONE)
#define BSR_SC_RESUME_PART_(STATE2)                                              \
  ".Lsc_resume%=:\n\t"                                                           \
  "s_bfe_u32 s10, %[resume], 0x40004\n\t"   /* (bits 8..22: where the tape's program stood, below) */ \
  "s_and_b32 s12, %[resume], 15\n\t"                                             \
  "s_cmp_eq_u32 s10, 0\n\t"                                                      \
  "s_cbranch_scc1 .Lsc_res0_%=\n\t"                                              \
  "s_cmp_eq_u32 s10, 1\n\t"                                                      \
  "s_cbranch_scc1 .Lsc_res1_%=\n\t"                                              \
  "s_cmp_eq_u32 s10, 2\n\t"                                                      \
  "s_cbranch_scc1 .Lsc_res2_%=\n\t"                                              \
  "s_branch .Lsc_res3_%=\n"                                                      \
  BSR_SC_RESUME(0) BSR_SC_RESUME(1) BSR_SC_RESUME(2) BSR_SC_RESUME(3)            \
  ".Lsc_state%=:\n\t"   /* the state a leave put into the operands: five words, the same in every lane */ \
  "v_readfirstlane_b32 s20, %[sv0]\n\t"                                          \
  "v_readfirstlane_b32 s21, %[sv1]\n\t"                                          \
  "v_readfirstlane_b32 s22, %[sv2]\n\t"                                          \
  "v_readfirstlane_b32 s23, %[sv3]\n\t"                                          \
  "v_readfirstlane_b32 s13, %[sv4]\n\t"                                          \
  "v_mov_b64_e32 v[4:5], %[s0]\n\t"                                              \
  "v_mov_b64_e32 v[6:7], %[s1]\n\t"                                              \
  STATE2                                                                         \
  BSR_SC_DISPATCH
#define BSR_SC_STATE2_IN "v_mov_b64_e32 v[40:41], %[s2]\n\t" "v_mov_b64_e32 v[42:43], %[s3]\n\t"
#define BSR_SC_STATE2_OUT "v_mov_b64_e32 %[s2], v[40:41]\n\t" "v_mov_b64_e32 %[s3], v[42:43]\n\t"
// the operator table (2 KB-aligned: the dispatch ORs a slot's offset into its address) and the operators too long for a slot
#define BSR_SC_TABLE_PART(SH, YQ) BSR_SC_TABLE_PART_(SH, YQ, BSR_SC_LDA_F64, BSR_SC_LDO_F64, BSR_SC_NONE, BSR_SC_NONE)
#define BSR_SC_TABLE_PART_(SH, YQ, LDA, LDO, PUSH, POP)                          \
  ".p2align 11\n"                                                                \
  ".Lsc_tab%=:\n\t"                                                              \
  "s_setpc_b64 s[28:29]\n\t"   /* 0: end of the program word: the tape's add-up code -- or .Lsc_more (round 6) */ \
  BSR_SC_SLOT("1") "s_branch .Lsc_inv%=\n\t"                                     \
  BSR_SC_SLOT("2") /* ln: a x + b, two roundings */                              \
  "v_mov_b32_e32 v20, s13\n\t"                                                   \
  "ds_read_b128 v[8:11], v20\n\t"                                                \
  "s_add_u32 s13, s13, 16\n\t"                                                   \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_mul_f64 v[0:1], v[8:9], v[0:1]\n\t"                                         \
  "v_mul_f64 v[2:3], v[8:9], v[2:3]\n\t"                                         \
  "v_add_f64 v[0:1], v[0:1], v[10:11]\n\t"                                       \
  "v_add_f64 v[2:3], v[2:3], v[10:11]\n\t"                                       \
  BSR_SC_DISPATCH                                                                \
  BSR_SC_SLOT("3") /* neg */                                                     \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_xor_b32_e32 v1, 0x80000000, v1\n\t"                                         \
  "v_xor_b32_e32 v3, 0x80000000, v3\n\t"                                         \
  BSR_SC_DISPATCH                                                                \
  BSR_SC_SLOT("4") "s_branch .Lsc_sin%=\n\t"   /* + 128/pi, pi/128 in three parts, -1/5040, 1/120, -1/6, -1/720 */ \
  BSR_SC_QUAD("0x40445F306DC9C883", "0x3F9921FB54442D18", "0x3C31A62633145C07", "0xB8BF1976B7ED8FBC",                \
              "0xBF2A01A01A01A01A", "0x3F81111111111111", "0xBFC5555555555555", "0xBF56C16C16C16C17")                \
  BSR_SC_SLOT("5") "s_branch .Lsc_cos%=\n\t"   /* + 1/24, 2^-26, 2^20 pi/2 (BSR_SINCOS_LIMIT) */ \
  BSR_SC_QUAD("0x3FA5555555555555", "0x3E50000000000000", "0x413921FB00000000", "0", "0", "0", "0", "0")             \
  BSR_SC_SLOT("6") "s_branch .Lsc_exp%=\n\t"   /* + 64/ln2, ln2/64 in two parts, 1/720, 1/120, 1/24, 1/6, 710 */ \
  BSR_SC_QUAD("0x40571547652B82FE", "0x3F862E42FEFA39EF", "0x3C1ABC9E3B39803F", "0x3F56C16C16C16C17",                \
              "0x3F81111111111111", "0x3FA5555555555555", "0x3FC5555555555555", "0x4086300000000000")                \
  BSR_SC_SLOT("7") /* square */                                                  \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_mul_f64 v[0:1], v[0:1], v[0:1]\n\t"                                         \
  "v_mul_f64 v[2:3], v[2:3], v[2:3]\n\t"                                         \
  BSR_SC_DISPATCH                                                                \
  BSR_SC_SLOT("8") "s_branch .Lsc_cube%=\n\t"   /* + -760, 200 */               \
  BSR_SC_QUAD("0xC087C00000000000", "0x4069000000000000", "0", "0", "0", "0", "0", "0")                              \
  BSR_SC_SLOT("9") BSR_SC_BIN("v_add_f64", "", POP)                              \
  BSR_SC_SLOT("10") BSR_SC_BIN("v_mul_f64", "", POP)                             \
  BSR_SC_SLOT("11") /* terminal: the accumulator becomes the saved value */      \
  BSR_SC_SLOT_ADDR(SH)                                                           \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  PUSH                                                                           \
  "v_mov_b64_e32 v[4:5], v[0:1]\n\t"                                             \
  "v_mov_b64_e32 v[6:7], v[2:3]\n\t"                                             \
  LDA                                                                            \
  BSR_SC_DISPATCH                                                                \
  BSR_SC_SLOT("12") BSR_SC_BIN_T("v_add_f64", SH, LDO)                            \
  BSR_SC_SLOT("13") BSR_SC_BIN_T("v_mul_f64", SH, LDO)                            \
  BSR_SC_SLOT("14") BSR_SC_BIN("v_add_f64", "-", POP) /* sub */                  \
  BSR_SC_SLOT("15") /* div, protected like inv */                                \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_SC_DIV("v[0:1]", "v0", "v1", "v[4:5]")                                     \
  BSR_SC_DIV("v[2:3]", "v2", "v3", "v[6:7]")                                     \
  POP                                                                            \
  BSR_SC_DISPATCH                                                                \
  ".Lsc_more%=:\n\t"   /* a program of several words (where `end` goes while words are left): s16 bits 16..19 extension words left, \
                           bits 20..29 the next one's place in 16-byte units behind %[sr]; s[14:15] where the tape's LAST end goes */ \
  "s_sub_u32 s16, s16, 0x10000\n\t"                                              \
  "s_bfe_u32 s10, s16, 0xa0014\n\t"                                              \
  "s_lshl_b32 s10, s10, 4\n\t"                                                   \
  "s_load_dwordx4 s[20:23], %[sr], s10\n\t"                                      \
  "s_add_u32 s16, s16, 0x100000\n\t"                                             \
  "s_and_b32 s10, s16, 0xf0000\n\t"                                              \
  "s_cbranch_scc1 .Lsc_more_go%=\n\t"                                            \
  "s_mov_b64 s[28:29], s[14:15]\n"   /* the last word: its end is the tape's */  \
  ".Lsc_more_go%=:\n\t"                                                          \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_SC_DISPATCH                                                                \
  ".Lsc_inv%=:\n\t"                                                              \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_SC_DIV("v[0:1]", "v0", "v1", "1.0")                                        \
  BSR_SC_DIV("v[2:3]", "v2", "v3", "1.0")                                        \
  BSR_SC_DISPATCH                                                                \
  ".Lsc_cube%=:\n\t"                                                             \
  "s_movk_i32 s10, 0x1f8\n\t"   /* the finite classes */                         \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_SC_CUBE("v[0:1]", "v0", "v1")                                              \
  BSR_SC_CUBE("v[2:3]", "v2", "v3")                                              \
  BSR_SC_DISPATCH                                                                \
  ".Lsc_sin%=:\n\t"                                                              \
  "s_movk_i32 s12, 4\n\t"                                                        \
  "s_load_dwordx16 s[36:51], s[24:25], 576\n\t"                                  \
  "s_load_dwordx8 s[52:59], s[24:25], 704\n\t"                                   \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_cmp_nlt_f64_e64 s[8:9], |v[0:1]|, s[56:57]\n\t"   /* huge, infinite, NaN: the caller's (library) routine */ \
  "v_cmp_nlt_f64_e64 vcc, |v[2:3]|, s[56:57]\n\t"                                \
  "s_or_b64 vcc, vcc, s[8:9]\n\t"                                                \
  "s_cbranch_vccnz .Lsc_leave%=\n\t"                                             \
  BSR_SC_SINCOS("v[0:1]", "v0", "v1", "", BSR_SC_SIN_TINY("v[0:1]", "v0", "v1")) \
  BSR_SC_SINCOS("v[2:3]", "v2", "v3", "", BSR_SC_SIN_TINY("v[2:3]", "v2", "v3")) \
  YQ                                                                             \
  BSR_SC_DISPATCH                                                                \
  ".Lsc_cos%=:\n\t"                                                              \
  "s_movk_i32 s12, 5\n\t"                                                        \
  "s_load_dwordx16 s[36:51], s[24:25], 576\n\t"                                  \
  "s_load_dwordx8 s[52:59], s[24:25], 704\n\t"                                   \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_cmp_nlt_f64_e64 s[8:9], |v[0:1]|, s[56:57]\n\t"                             \
  "v_cmp_nlt_f64_e64 vcc, |v[2:3]|, s[56:57]\n\t"                                \
  "s_or_b64 vcc, vcc, s[8:9]\n\t"                                                \
  "s_cbranch_vccnz .Lsc_leave%=\n\t"                                             \
  BSR_SC_SINCOS("v[0:1]", "v0", "v1", "v_add_u32_e32 v20, 64, v20\n\t", BSR_SC_COS_MOVE("v[0:1]")) \
  BSR_SC_SINCOS("v[2:3]", "v2", "v3", "v_add_u32_e32 v20, 64, v20\n\t", BSR_SC_COS_MOVE("v[2:3]")) \
  YQ                                                                             \
  BSR_SC_DISPATCH                                                                \
  ".Lsc_exp%=:\n\t"                                                              \
  "s_load_dwordx16 s[36:51], s[24:25], 832\n\t"                                  \
  "s_load_dwordx4 s[52:55], s[24:25], 1088\n\t"                                  \
  "s_waitcnt lgkmcnt(0)\n\t"   /* (FIRST: the chunk's y / basis reads may still be in flight, and v[28:31] is one of their destinations --  \
                                   round 6: written in front of this wait, the constant below was overwritten by the LDS data that landed behind \
                                   it, and exp of anything beyond 200 -- exp(exp(x)) -- returned a basis value instead of 1e10, run-dependent) */ \
  "v_mov_b32_e32 v30, 0x20000000\n\t"   /* 1e10: what the clipped exp returns beyond 200 (and for NaN) */ \
  "v_mov_b32_e32 v31, 0x4202a05f\n\t"                                            \
  BSR_SC_EXP("v[0:1]", "v0", "v1")                                               \
  BSR_SC_EXP("v[2:3]", "v2", "v3")                                               \
  YQ                                                                             \
  BSR_SC_DISPATCH
// leaving for the caller: sin / cos of huge arguments (the state out through the operands; which tape: by where `end` goes)
#define BSR_SC_LEAVE_PART BSR_SC_LEAVE_PART_(BSR_SC_NONE)
#define BSR_SC_LEAVE_PART_(STATE2)                                               \
  ".Lsc_leave%=:\n\t"   /* sin, cos of huge arguments: out with the state; which tape: the offset of its add-up code */ \
  "s_sub_u32 s10, s28, s24\n\t"                                                  \
  "s_cmp_eq_u32 s10, .Lsc_more%=-.Lsc_tab%=\n\t"   /* (a program with words left: its add-up code is in s[14:15]) */ \
  "s_cbranch_scc0 .Lsc_lv1%=\n\t"                                                \
  "s_sub_u32 s10, s14, s24\n"                                                     \
  ".Lsc_lv1%=:\n\t"                                                              \
  "v_mov_b32_e32 %[sv0], s20\n\t"                                                \
  "v_mov_b32_e32 %[sv1], s21\n\t"                                                \
  "v_mov_b32_e32 %[sv2], s22\n\t"                                                \
  "v_mov_b32_e32 %[sv3], s23\n\t"                                                \
  "v_mov_b32_e32 %[sv4], s13\n\t"                                                \
  "v_mov_b64_e32 %[s0], v[4:5]\n\t"                                              \
  "v_mov_b64_e32 %[s1], v[6:7]\n\t"                                              \
  STATE2                                                                         \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_mov_b64_e32 %[z0], v[0:1]\n\t"                                              \
  "v_mov_b64_e32 %[z1], v[2:3]\n\t"                                              \
  "s_mov_b32 %[st], s12\n\t"                                                     \
  "s_bfe_u32 s12, s16, 0xe0010\n\t"   /* extension words left and the next one's place: back in through `resume` */ \
  "s_lshl_b32 s12, s12, 8\n\t"                                                   \
  "s_or_b32 %[st], %[st], s12\n\t"                                               \
  "s_cmp_eq_u32 s10, .Lsc_acc0_%=-.Lsc_tab%=\n\t"                                \
  "s_cbranch_scc1 .Lsc_exit%=\n\t"                                               \
  "s_add_u32 %[st], %[st], 16\n\t"                                               \
  "s_cmp_eq_u32 s10, .Lsc_acc1_%=-.Lsc_tab%=\n\t"                                \
  "s_cbranch_scc1 .Lsc_exit%=\n\t"                                               \
  "s_add_u32 %[st], %[st], 16\n\t"                                               \
  "s_cmp_eq_u32 s10, .Lsc_acc2_%=-.Lsc_tab%=\n\t"                                \
  "s_cbranch_scc1 .Lsc_exit%=\n\t"                                               \
  "s_add_u32 %[st], %[st], 16\n"

// (DEEP: the second saved value and its moves, K <= 3; FLAT: none of it)
#define BSR_SC_DEEP BSR_SC_PUSH2, BSR_SC_POP2, BSR_SC_STATE2_IN, BSR_SC_STATE2_OUT
#define BSR_SC_FLAT BSR_SC_NONE, BSR_SC_NONE, BSR_SC_NONE, BSR_SC_NONE
#define BSR_STREAM_CHUNK_ASM_(SH, YQ, TP, ...) BSR_STREAM_CHUNK_ASM__(SH, YQ, TP, BSR_SC_LDA_F64, BSR_SC_LDO_F64, __VA_ARGS__)
#define BSR_STREAM_CHUNK_ASM_X(SH, YQ, TP, LDA, LDO, ...) BSR_STREAM_CHUNK_ASM__(SH, YQ, TP, LDA, LDO, __VA_ARGS__)
#define BSR_STREAM_CHUNK_ASM__(SH, YQ, TP, LDA, LDO, PUSH, POP, ST_IN, ST_OUT)   \
  "s_load_dwordx8 s[16:23], %[sr], 0x0\n\t"                                      \
  BSR_SC_TABLE_BASE                                                              \
  YQ   /* y and the basis columns of the lane's rows, once for the four tapes */ \
  "s_cmp_lg_u32 %[resume], 0\n\t"                                                \
  "s_cbranch_scc1 .Lsc_resume%=\n\t"                                             \
  TP                                                                             \
  "s_mov_b32 %[st], 0\n\t"                                                       \
  "s_branch .Lsc_exit%=\n"                                                       \
  BSR_SC_RESUME_PART_(ST_IN)                                                     \
  BSR_SC_TABLE_PART_(SH, YQ, LDA, LDO, PUSH, POP)                                \
  BSR_SC_LEAVE_PART_(ST_OUT)                                                     \
  ".Lsc_exit%=:\n\t"                                                             \
  "s_waitcnt lgkmcnt(0)\n\t"

// ---------------------------------------------------------------------------------------------------------------
// The whole pass of a wave over its slice in one block: for every chunk -- wait for its copies, barrier, request the
// chunk R - 1 ahead, the four tapes (above).  Why: after the barrier all sixteen waves of the workgroup run this
// hand-over at the same time on the CU's one scalar unit, and the compiler's version is 75 scalar instructions per
// wave (column bases re-read and clamped per piece, a compare chain to pick s_waitcnt's immediate, a compare per piece,
// spilled ring offsets): 0.6 us of every 3 us chunk.  Here: ~25.  One-block chunks only.
//
// More fixed registers: s[60:67] the column bases of the wave's (at most four) pieces, s68 chunk, s69 chunks of the
// slice, s70 where in the ring the next request goes (byte offset), s71 LDS address of the next chunk to compute on,
// s72 bytes of a buffer, s73 of the ring, s74 LDS address of the wave's first piece in buffer 0, s75 end of the ring,
// s76 first chunk of the tail (fewer copies in flight than in steady state: wait for all), s77 chunks that still
// request one, s78 next block to request, s79 LDS address of the chunk being computed on, s[80:81] the steady state's
// s_waitcnt, s[82:83] entry into the request code for the wave's number of pieces, s84 LDS address of the ring.
// A leave (st != 0) also puts s68, s70, s71, s78, s79 into sv5..sv9; the caller passes everything back as it got it.
#define BSR_SP_PIECE(k, base)                                                    \
  ".Lsp_is" #k "_%=:\n\t"                                                        \
  "s_add_u32 m0, s10, " #k "*16384\n\t"                                          \
  "s_nop 0\n\t"                                                                  \
  "global_load_lds_dwordx4 v20, " base "\n\t"
#define BSR_SP_WAIT(k) "s_waitcnt vmcnt(" #k ")\n\ts_branch .Lsp_waited%=\n\t"

#define BSR_STREAM_PASS_ASM_(YQ, TP, ...) BSR_STREAM_PASS_ASM__(YQ, TP, __VA_ARGS__)
#define BSR_STREAM_PASS_ASM__(YQ, TP, PUSH, POP, ST_IN, ST_OUT)                   \
  BSR_SC_TABLE_BASE                                                              \
  "s_mov_b64 s[60:61], %[ba0]\n\t"                                               \
  "s_mov_b64 s[62:63], %[ba1]\n\t"                                               \
  "s_mov_b64 s[64:65], %[ba2]\n\t"                                               \
  "s_mov_b64 s[66:67], %[ba3]\n\t"                                               \
  "s_mov_b32 s69, %[nch]\n\t"                                                    \
  "s_mov_b32 s72, %[bufb]\n\t"                                                   \
  "s_mul_i32 s73, s72, %[ring]\n\t"                                              \
  "s_mov_b32 s84, %[lds0]\n\t"                                                   \
  "s_add_u32 s75, s84, s73\n\t"                                                  \
  "s_lshl_b32 s10, %[wave], 10\n\t"                                              \
  "s_add_u32 s74, s84, s10\n\t"                                                  \
  "s_sub_u32 s10, %[ring], 2\n\t"                                                \
  "s_sub_i32 s76, s69, s10\n\t"                                                  \
  "s_sub_i32 s77, s76, 1\n\t"                                                    \
  "s_mul_i32 s10, s10, %[nmine]\n\t"   /* copies of later chunks in flight, steady state */ \
  "s_lshl_b32 s10, s10, 3\n\t"                                                   \
  "s_add_u32 s80, s24, .Lsp_waits%=-.Lsc_tab%=\n\t"                              \
  "s_addc_u32 s81, s25, 0\n\t"                                                   \
  "s_add_u32 s80, s80, s10\n\t"                                                  \
  "s_addc_u32 s81, s81, 0\n\t"                                                   \
  "s_mov_b32 s10, .Lsp_is_none%=-.Lsc_tab%=\n\t"                                 \
  "s_cmp_eq_u32 %[nmine], 1\n\t"                                                 \
  "s_cselect_b32 s10, .Lsp_is0_%=-.Lsc_tab%=, s10\n\t"                           \
  "s_cmp_eq_u32 %[nmine], 2\n\t"                                                 \
  "s_cselect_b32 s10, .Lsp_is1_%=-.Lsc_tab%=, s10\n\t"                           \
  "s_cmp_eq_u32 %[nmine], 3\n\t"                                                 \
  "s_cselect_b32 s10, .Lsp_is2_%=-.Lsc_tab%=, s10\n\t"                           \
  "s_cmp_eq_u32 %[nmine], 4\n\t"                                                 \
  "s_cselect_b32 s10, .Lsp_is3_%=-.Lsc_tab%=, s10\n\t"                           \
  "s_add_u32 s82, s24, s10\n\t"                                                  \
  "s_addc_u32 s83, s25, 0\n\t"                                                   \
  "s_cmp_lg_u32 %[resume], 0\n\t"                                                \
  "s_cbranch_scc1 .Lsp_resume%=\n\t"                                             \
  "s_mov_b32 s68, 0\n\t"                                                         \
  "s_mov_b32 s70, %[ioff]\n\t"                                                   \
  "s_mov_b32 s71, s84\n\t"                                                       \
  "s_add_u32 s78, %[b0], %[ring]\n\t"                                            \
  "s_sub_u32 s78, s78, 1\n"                                                      \
  ".Lsp_loop%=:\n\t"                                                             \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "s_cmp_ge_i32 s68, s76\n\t"                                                    \
  "s_cbranch_scc1 .Lsp_waitall_%=\n\t"                                              \
  "s_setpc_b64 s[80:81]\n"                                                       \
  ".Lsp_waitall_%=:\n\t"                                                            \
  "s_waitcnt vmcnt(0)\n"                                                         \
  ".Lsp_waited%=:\n\t"                                                           \
  "s_barrier\n\t"   /* chunk s68 has landed for everyone; everyone is done with the chunk before */ \
  "s_load_dwordx8 s[16:23], %[sr], 0x0\n\t"                                      \
  "s_cmp_lt_i32 s68, s77\n\t"                                                    \
  "s_cbranch_scc0 .Lsp_issued%=\n\t"                                             \
  "v_lshl_add_u32 v20, s78, 10, %[lane16]\n\t"                                   \
  "s_add_u32 s10, s74, s70\n\t"                                                  \
  "s_setpc_b64 s[82:83]\n"                                                       \
  ".Lsp_reissue_%=:\n\t"                                                          \
  "s_add_u32 s70, s70, s72\n\t"                                                  \
  "s_cmp_eq_u32 s70, s73\n\t"                                                    \
  "s_cselect_b32 s70, 0, s70\n\t"                                                \
  "s_add_u32 s78, s78, 1\n"                                                      \
  ".Lsp_issued%=:\n\t"                                                           \
  "s_mov_b32 s79, s71\n\t"                                                       \
  "v_add_u32_e32 %[lc], s71, %[lane16]\n\t"                                      \
  "s_add_u32 s71, s71, s72\n\t"                                                  \
  "s_cmp_eq_u32 s71, s75\n\t"                                                    \
  "s_cselect_b32 s71, s84, s71\n\t"                                              \
  YQ                                                                             \
  TP                                                                             \
  "s_add_u32 s68, s68, 1\n\t"                                                    \
  "s_cmp_lt_u32 s68, s69\n\t"                                                    \
  "s_cbranch_scc1 .Lsp_loop%=\n\t"                                               \
  "s_mov_b32 %[st], 0\n\t"                                                       \
  "s_branch .Lsc_exit%=\n"                                                       \
  ".Lsp_resume%=:\n\t"   /* back into the chunk the block left */               \
  "v_readfirstlane_b32 s68, %[sv5]\n\t"                                          \
  "v_readfirstlane_b32 s70, %[sv6]\n\t"                                          \
  "v_readfirstlane_b32 s71, %[sv7]\n\t"                                          \
  "v_readfirstlane_b32 s78, %[sv8]\n\t"                                          \
  "v_readfirstlane_b32 s79, %[sv9]\n\t"                                          \
  "s_nop 3\n\t"                                                                  \
  "v_add_u32_e32 %[lc], s79, %[lane16]\n\t"                                      \
  YQ                                                                             \
  BSR_SC_RESUME_PART_(ST_IN)                                                     \
  BSR_SC_TABLE_PART_("10", YQ, BSR_SC_LDA_F64, BSR_SC_LDO_F64, PUSH, POP)        \
  ".Lsp_waits%=:\n\t"   /* the steady state's wait: entry 8 n for n copies of later chunks in flight */ \
  BSR_SP_WAIT(0) BSR_SP_WAIT(1) BSR_SP_WAIT(2) BSR_SP_WAIT(3) BSR_SP_WAIT(4) BSR_SP_WAIT(5) BSR_SP_WAIT(6)     \
  BSR_SP_WAIT(7) BSR_SP_WAIT(8)                                                  \
  BSR_SP_PIECE(3, "s[66:67]") BSR_SP_PIECE(2, "s[64:65]") BSR_SP_PIECE(1, "s[62:63]") BSR_SP_PIECE(0, "s[60:61]") \
  ".Lsp_is_none%=:\n\t"                                                          \
  "s_branch .Lsp_reissue_%=\n\t"                                                  \
  BSR_SC_LEAVE_PART_(ST_OUT)                                                     \
  ".Lsc_exit%=:\n\t"                                                             \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_mov_b32_e32 %[sv5], s68\n\t"                                                \
  "v_mov_b32_e32 %[sv6], s70\n\t"                                                \
  "v_mov_b32_e32 %[sv7], s71\n\t"                                                \
  "v_mov_b32_e32 %[sv8], s78\n\t"                                                \
  "v_mov_b32_e32 %[sv9], s79\n\t"

#define BSR_STREAM_PASS_CLOBBERS_(C) C, "m0", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", \
  "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84"

// K = 1..4 basis columns; one-block chunks: a column of the buffer is 1024 bytes; two-block chunks: 2048 (the block's half
// picked by %[lc])
#define BSR_STREAM_CHUNK_ASM_K1 BSR_STREAM_CHUNK_ASM_("10", BSR_SC_YQ1("1024"), BSR_SC_TAPES_1, BSR_SC_FLAT)
#define BSR_STREAM_CHUNK_ASM_K2 BSR_STREAM_CHUNK_ASM_("10", BSR_SC_YQ2("1024", "2048"), BSR_SC_TAPES_2, BSR_SC_FLAT)
#define BSR_STREAM_CHUNK_ASM_K3 BSR_STREAM_CHUNK_ASM_("10", BSR_SC_YQ3("1024", "2048", "3072"), BSR_SC_TAPES_3, BSR_SC_FLAT)
#define BSR_STREAM_CHUNK_ASM_K4 BSR_STREAM_CHUNK_ASM_("10", BSR_SC_YQ4("1024", "2048", "3072", "4096"), BSR_SC_TAPES_4, BSR_SC_FLAT)
#define BSR_STREAM_CHUNK2_ASM_K1 BSR_STREAM_CHUNK_ASM_("11", BSR_SC_YQ1("2048"), BSR_SC_TAPES_1, BSR_SC_FLAT)
#define BSR_STREAM_CHUNK2_ASM_K2 BSR_STREAM_CHUNK_ASM_("11", BSR_SC_YQ2("2048", "4096"), BSR_SC_TAPES_2, BSR_SC_FLAT)
#define BSR_STREAM_CHUNK2_ASM_K3 BSR_STREAM_CHUNK_ASM_("11", BSR_SC_YQ3("2048", "4096", "6144"), BSR_SC_TAPES_3, BSR_SC_FLAT)
#define BSR_STREAM_CHUNK2_ASM_K4 BSR_STREAM_CHUNK_ASM_("11", BSR_SC_YQ4("2048", "4096", "6144", "8192"), BSR_SC_TAPES_4, BSR_SC_FLAT)
#define BSR_STREAM_PASS_ASM_K1 BSR_STREAM_PASS_ASM_(BSR_SC_YQ1("1024"), BSR_SC_TAPES_1, BSR_SC_FLAT)
#define BSR_STREAM_PASS_ASM_K1D BSR_STREAM_PASS_ASM_(BSR_SC_YQ1("1024"), BSR_SC_TAPES_1, BSR_SC_DEEP)   /* with the second value below the accumulator */
#define BSR_STREAM_PASS_ASM_K2 BSR_STREAM_PASS_ASM_(BSR_SC_YQ2("1024", "2048"), BSR_SC_TAPES_2, BSR_SC_FLAT)
#define BSR_STREAM_PASS_ASM_K2D BSR_STREAM_PASS_ASM_(BSR_SC_YQ2("1024", "2048"), BSR_SC_TAPES_2, BSR_SC_DEEP)   /* with the second value below the accumulator */
#define BSR_STREAM_PASS_ASM_K3 BSR_STREAM_PASS_ASM_(BSR_SC_YQ3("1024", "2048", "3072"), BSR_SC_TAPES_3, BSR_SC_FLAT)
#define BSR_STREAM_PASS_ASM_K3D BSR_STREAM_PASS_ASM_(BSR_SC_YQ3("1024", "2048", "3072"), BSR_SC_TAPES_3, BSR_SC_DEEP)   /* with the second value below the accumulator */
#define BSR_STREAM_PASS_ASM_K4 BSR_STREAM_PASS_ASM_(BSR_SC_YQ4("1024", "2048", "3072", "4096"), BSR_SC_TAPES_4, BSR_SC_FLAT)

// f32 storage (K <= 4): a column of the chunk buffer is 1 KiB = 256 rows of f32, evaluated as two blocks of 128 rows (%[lc]:
// the half's address + lane * 8); y and the basis through the temporaries v[8:17] (free at the head of a chunk and behind
// sin / cos / exp), waited for and converted at once
#define BSR_SCF_YQ_HEAD                                                 \
  "v_add_u32_e32 v20, %[yo], %[lc]\n\t"                                 \
  "ds_read_b64 v[8:9], v20\n\t"
#define BSR_SCF_CVT(d0, d1, s0, s1) "v_cvt_f64_f32_e32 " d0 ", " s0 "\n\t" "v_cvt_f64_f32_e32 " d1 ", " s1 "\n\t"
#define BSR_SCF_YQ1 BSR_SCF_YQ_HEAD "ds_read_b64 v[10:11], v20 offset:1024\n\t" "s_waitcnt lgkmcnt(0)\n\t" \
  BSR_SCF_CVT("v[24:25]", "v[26:27]", "v8", "v9") BSR_SCF_CVT("v[28:29]", "v[30:31]", "v10", "v11")
#define BSR_SCF_YQ2 BSR_SCF_YQ_HEAD "ds_read_b64 v[10:11], v20 offset:1024\n\t" "ds_read_b64 v[12:13], v20 offset:2048\n\t" \
  "s_waitcnt lgkmcnt(0)\n\t" BSR_SCF_CVT("v[24:25]", "v[26:27]", "v8", "v9") BSR_SCF_CVT("v[28:29]", "v[30:31]", "v10", "v11") \
  BSR_SCF_CVT("v[32:33]", "v[34:35]", "v12", "v13")
#define BSR_SCF_YQ3 BSR_SCF_YQ_HEAD "ds_read_b64 v[10:11], v20 offset:1024\n\t" "ds_read_b64 v[12:13], v20 offset:2048\n\t" \
  "ds_read_b64 v[14:15], v20 offset:3072\n\t" "s_waitcnt lgkmcnt(0)\n\t" \
  BSR_SCF_CVT("v[24:25]", "v[26:27]", "v8", "v9") BSR_SCF_CVT("v[28:29]", "v[30:31]", "v10", "v11") \
  BSR_SCF_CVT("v[32:33]", "v[34:35]", "v12", "v13") BSR_SCF_CVT("v[36:37]", "v[38:39]", "v14", "v15")
#define BSR_SCF_YQ4 BSR_SCF_YQ_HEAD "ds_read_b64 v[10:11], v20 offset:1024\n\t" "ds_read_b64 v[12:13], v20 offset:2048\n\t" \
  "ds_read_b64 v[14:15], v20 offset:3072\n\t" "ds_read_b64 v[16:17], v20 offset:4096\n\t" "s_waitcnt lgkmcnt(0)\n\t" \
  BSR_SCF_CVT("v[24:25]", "v[26:27]", "v8", "v9") BSR_SCF_CVT("v[28:29]", "v[30:31]", "v10", "v11") \
  BSR_SCF_CVT("v[32:33]", "v[34:35]", "v12", "v13") BSR_SCF_CVT("v[36:37]", "v[38:39]", "v14", "v15") \
  BSR_SCF_CVT("v[40:41]", "v[42:43]", "v16", "v17")
#define BSR_SCF_TAPES_1 BSR_SC_TAPES4_(BSR_SC_R0_1, BSR_SC_R1_1, BSR_SC_LDA_F32)
#define BSR_SCF_TAPES_2 BSR_SC_TAPES4_(BSR_SC_R0_2, BSR_SC_R1_2, BSR_SC_LDA_F32)
#define BSR_SCF_TAPES_3 BSR_SC_TAPES4_(BSR_SC_R0_3, BSR_SC_R1_3, BSR_SC_LDA_F32)
#define BSR_SCF_TAPES_4 BSR_SC_TAPES4_(BSR_SC_R0_4, BSR_SC_R1_4, BSR_SC_LDA_F32)
#define BSR_STREAM_CHUNKF_ASM_K1 BSR_STREAM_CHUNK_ASM_X("10", BSR_SCF_YQ1, BSR_SCF_TAPES_1, BSR_SC_LDA_F32, BSR_SC_LDO_F32, BSR_SC_FLAT)
#define BSR_STREAM_CHUNKF_ASM_K2 BSR_STREAM_CHUNK_ASM_X("10", BSR_SCF_YQ2, BSR_SCF_TAPES_2, BSR_SC_LDA_F32, BSR_SC_LDO_F32, BSR_SC_FLAT)
#define BSR_STREAM_CHUNKF_ASM_K3 BSR_STREAM_CHUNK_ASM_X("10", BSR_SCF_YQ3, BSR_SCF_TAPES_3, BSR_SC_LDA_F32, BSR_SC_LDO_F32, BSR_SC_FLAT)
#define BSR_STREAM_CHUNKF_ASM_K4 BSR_STREAM_CHUNK_ASM_X("10", BSR_SCF_YQ4, BSR_SCF_TAPES_4, BSR_SC_LDA_F32, BSR_SC_LDO_F32, BSR_SC_FLAT)

#define BSR_SC_YQ_A5 BSR_SC_YQ5("1024", "2048", "3072", "4096", "5120")
#define BSR_SC_YQ_A6 BSR_SC_YQ6("1024", "2048", "3072", "4096", "5120", "6144")
#define BSR_SC_YQ_A7 BSR_SC_YQ7("1024", "2048", "3072", "4096", "5120", "6144", "7168")
#define BSR_SC_YQ_A8 BSR_SC_YQ8("1024", "2048", "3072", "4096", "5120", "6144", "7168", "8192")
#define BSR_SC_YQ_B5 BSR_SC_YQ5("2048", "4096", "6144", "8192", "10240")
#define BSR_SC_YQ_B6 BSR_SC_YQ6("2048", "4096", "6144", "8192", "10240", "12288")
#define BSR_SC_YQ_B7 BSR_SC_YQ7("2048", "4096", "6144", "8192", "10240", "12288", "14336")
#define BSR_SC_YQ_B8 BSR_SC_YQ8("2048", "4096", "6144", "8192", "10240", "12288", "14336", "16384")
#define BSR_STREAM_CHUNK_ASM_K5 BSR_STREAM_CHUNK_ASM_("10", BSR_SC_YQ_A5, BSR_SC_TAPES_5, BSR_SC_FLAT)
#define BSR_STREAM_CHUNK_ASM_K6 BSR_STREAM_CHUNK_ASM_("10", BSR_SC_YQ_A6, BSR_SC_TAPES_6, BSR_SC_FLAT)
#define BSR_STREAM_CHUNK_ASM_K7 BSR_STREAM_CHUNK_ASM_("10", BSR_SC_YQ_A7, BSR_SC_TAPES_7, BSR_SC_FLAT)
#define BSR_STREAM_CHUNK_ASM_K8 BSR_STREAM_CHUNK_ASM_("10", BSR_SC_YQ_A8, BSR_SC_TAPES_8, BSR_SC_FLAT)
#define BSR_STREAM_CHUNK2_ASM_K5 BSR_STREAM_CHUNK_ASM_("11", BSR_SC_YQ_B5, BSR_SC_TAPES_5, BSR_SC_FLAT)
#define BSR_STREAM_CHUNK2_ASM_K6 BSR_STREAM_CHUNK_ASM_("11", BSR_SC_YQ_B6, BSR_SC_TAPES_6, BSR_SC_FLAT)
#define BSR_STREAM_CHUNK2_ASM_K7 BSR_STREAM_CHUNK_ASM_("11", BSR_SC_YQ_B7, BSR_SC_TAPES_7, BSR_SC_FLAT)
#define BSR_STREAM_CHUNK2_ASM_K8 BSR_STREAM_CHUNK_ASM_("11", BSR_SC_YQ_B8, BSR_SC_TAPES_8, BSR_SC_FLAT)
#define BSR_STREAM_PASS_ASM_K5 BSR_STREAM_PASS_ASM_(BSR_SC_YQ_A5, BSR_SC_TAPES_5, BSR_SC_FLAT)
#define BSR_STREAM_PASS_ASM_K6 BSR_STREAM_PASS_ASM_(BSR_SC_YQ_A6, BSR_SC_TAPES_6, BSR_SC_FLAT)
#define BSR_STREAM_PASS_ASM_K7 BSR_STREAM_PASS_ASM_(BSR_SC_YQ_A7, BSR_SC_TAPES_7, BSR_SC_FLAT)
#define BSR_STREAM_PASS_ASM_K8 BSR_STREAM_PASS_ASM_(BSR_SC_YQ_A8, BSR_SC_TAPES_8, BSR_SC_FLAT)

#define BSR_STREAM_CHUNK_CLOBBERS                                                                                      \
  "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17",   \
  "v18", "v19", "v20", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", \
  "v38", "v39", "s8", "s9", "s10", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", \
  "s24", "s25", "s26", "s27", "s28", "s29", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", \
  "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "vcc", "scc", "memory"
// (K = 4: the fourth basis column's values; K <= 3, round 6: the second value below the accumulator)
#define BSR_STREAM_CHUNK_CLOBBERS_K4 BSR_STREAM_CHUNK_CLOBBERS, "v40", "v41", "v42", "v43"
#define BSR_STREAM_PASS_CLOBBERS BSR_STREAM_PASS_CLOBBERS_(BSR_STREAM_CHUNK_CLOBBERS)
#define BSR_STREAM_PASS_CLOBBERS_K4 BSR_STREAM_PASS_CLOBBERS_(BSR_STREAM_CHUNK_CLOBBERS_K4)
// (K = 5..8: up to eight basis columns)
#define BSR_STREAM_CHUNK_CLOBBERS_K8 BSR_STREAM_CHUNK_CLOBBERS_K4, "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", \
  "v54", "v55", "v56", "v57", "v58", "v59"
#define BSR_STREAM_PASS_CLOBBERS_K8 BSR_STREAM_PASS_CLOBBERS_(BSR_STREAM_CHUNK_CLOBBERS_K8)
// clang-format on
