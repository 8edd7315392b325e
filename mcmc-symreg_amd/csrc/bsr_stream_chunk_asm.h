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

#define BSR_SC_SLOT_ADDR                         \
  "s_and_b32 s10, s22, 0xff\n\t"                 \
  "s_lshr_b64 s[22:23], s[22:23], 8\n\t"         \
  "v_lshl_add_u32 v20, s10, 10, %[lc]\n\t"

#define BSR_SC_BIN(ins, neg)                                                     \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  ins " v[0:1], v[4:5], " neg "v[0:1]\n\t"                                       \
  ins " v[2:3], v[6:7], " neg "v[2:3]\n\t"                                       \
  BSR_SC_DISPATCH
#define BSR_SC_BIN_T(ins)                                                        \
  BSR_SC_SLOT_ADDR                                                               \
  "ds_read_b128 v[8:11], v20\n\t"                                                \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
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

// where `end` goes for tape q (its add-up code, in front of the table), and the tape's ln pairs
#define BSR_SC_TAPE_REGS(q)                                                      \
  "s_sub_u32 s28, s24, .Lsc_tab%=-.Lsc_acc" #q "_%=\n\t"                         \
  "s_subb_u32 s29, s25, 0\n\t"                                                   \
  "s_add_u32 s13, %[ln], " #q "*48\n\t"

// Tape q of the wave: its program has been requested into s[16:23] (by the block's entry, or while tape q - 1 was added up)
#define BSR_SC_TAPE(q, qnext)                                                    \
  BSR_SC_TAPE_REGS(q)                                                            \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "s_cmp_lt_i32 s16, 0\n\t"                                                      \
  "s_cbranch_scc0 .Lsc_slow" #q "_%=\n\t"                                        \
  "v_add_u32_e32 v20, s17, %[lc]\n\t"                                            \
  "ds_read_b128 v[0:3], v20\n\t"                                                 \
  BSR_SC_DISPATCH                                                                \
  ".Lsc_slow" #q "_%=:\n\t"   /* no tape in this set of sums, or one for the stack machine */ \
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
  "v_fmac_f64_e32 %[cb" #q "], v[32:33], v[12:13]\n\t"                           \
  "v_fmac_f64_e32 %[cc" #q "], v[36:37], v[12:13]\n\t"                           \
  "v_max_f64 %[am" #q "], v[14:15], |v[2:3]|\n\t"                                \
  "v_fmac_f64_e32 %[sa" #q "], v[16:17], v[16:17]\n\t"                           \
  "v_fmac_f64_e32 %[sb" #q "], v[16:17], v[26:27]\n\t"                           \
  "v_fmac_f64_e32 %[ca" #q "], v[30:31], v[16:17]\n\t"                           \
  "v_fmac_f64_e32 %[cb" #q "], v[34:35], v[16:17]\n\t"                           \
  "v_fmac_f64_e32 %[cc" #q "], v[38:39], v[16:17]\n"                             \
  ".Lsc_next" #q "_%=:\n\t"

// coming back into tape q: its program again (prescale, flags), then either the caller's values straight to the sums
// (code 1) or the interpreter's state from the operands
#define BSR_SC_RESUME(q)                                                         \
  ".Lsc_res" #q "_%=:\n\t"                                                       \
  "s_load_dwordx8 s[16:23], %[sr], " #q "*32\n\t"                                \
  BSR_SC_TAPE_REGS(q)                                                            \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_mov_b64_e32 v[0:1], %[z0]\n\t"                                              \
  "v_mov_b64_e32 v[2:3], %[z1]\n\t"                                              \
  "s_cmp_eq_u32 s12, 1\n\t"                                                      \
  "s_cbranch_scc1 .Lsc_acc" #q "_%=\n\t"                                         \
  "s_branch .Lsc_state%=\n\t"

#define BSR_STREAM_CHUNK_ASM_K3                                                  \
  "s_load_dwordx8 s[16:23], %[sr], 0x0\n\t"                                      \
  "s_getpc_b64 s[24:25]\n"                                                       \
  ".Lsc_pc%=:\n\t"                                                               \
  "s_add_u32 s24, s24, .Lsc_tab%=-.Lsc_pc%=\n\t"                                 \
  "s_addc_u32 s25, s25, 0\n\t"                                                   \
  "s_mov_b32 s27, s25\n\t"                                                       \
  "v_add_u32_e32 v20, %[yo], %[lc]\n\t"   /* y and the basis columns of the lane's rows, once for the four tapes */ \
  "ds_read_b128 v[24:27], v20\n\t"                                               \
  "ds_read_b128 v[28:31], v20 offset:1024\n\t"                                   \
  "ds_read_b128 v[32:35], v20 offset:2048\n\t"                                   \
  "ds_read_b128 v[36:39], v20 offset:3072\n\t"                                   \
  "s_cmp_lg_u32 %[resume], 0\n\t"                                                \
  "s_cbranch_scc1 .Lsc_resume%=\n\t"                                             \
  BSR_SC_TAPE(0, 1) BSR_SC_TAPE(1, 2) BSR_SC_TAPE(2, 3) BSR_SC_TAPE(3, 4)        \
  "s_mov_b32 %[st], 0\n\t"                                                       \
  "s_branch .Lsc_exit%=\n"                                                       \
  ".Lsc_resume%=:\n\t"                                                           \
  "s_lshr_b32 s10, %[resume], 4\n\t"                                             \
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
  BSR_SC_DISPATCH                                                                \
  ".p2align 11\n"                                                                \
  ".Lsc_tab%=:\n\t"                                                              \
  "s_setpc_b64 s[28:29]\n\t"   /* 0: end of the tape */                          \
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
  BSR_SC_SLOT("4") BSR_SC_LEAVE("4") /* sin */                                   \
  BSR_SC_SLOT("5") BSR_SC_LEAVE("5") /* cos */                                   \
  BSR_SC_SLOT("6") BSR_SC_LEAVE("6") /* exp */                                   \
  BSR_SC_SLOT("7") /* square */                                                  \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_mul_f64 v[0:1], v[0:1], v[0:1]\n\t"                                         \
  "v_mul_f64 v[2:3], v[2:3], v[2:3]\n\t"                                         \
  BSR_SC_DISPATCH                                                                \
  BSR_SC_SLOT("8") "s_branch .Lsc_cube%=\n\t"                                    \
  BSR_SC_SLOT("9") BSR_SC_BIN("v_add_f64", "")                                   \
  BSR_SC_SLOT("10") BSR_SC_BIN("v_mul_f64", "")                                  \
  BSR_SC_SLOT("11") /* terminal: the accumulator becomes the saved value */      \
  BSR_SC_SLOT_ADDR                                                               \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_mov_b64_e32 v[4:5], v[0:1]\n\t"                                             \
  "v_mov_b64_e32 v[6:7], v[2:3]\n\t"                                             \
  "ds_read_b128 v[0:3], v20\n\t"                                                 \
  BSR_SC_DISPATCH                                                                \
  BSR_SC_SLOT("12") BSR_SC_BIN_T("v_add_f64")                                    \
  BSR_SC_SLOT("13") BSR_SC_BIN_T("v_mul_f64")                                    \
  BSR_SC_SLOT("14") BSR_SC_BIN("v_add_f64", "-") /* sub */                       \
  BSR_SC_SLOT("15") /* div, protected like inv */                                \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_SC_DIV("v[0:1]", "v0", "v1", "v[4:5]")                                     \
  BSR_SC_DIV("v[2:3]", "v2", "v3", "v[6:7]")                                     \
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
  ".Lsc_leave%=:\n\t"   /* sin, cos, exp: out with the state; which tape: the offset of its add-up code */ \
  "s_sub_u32 s10, s28, s24\n\t"                                                  \
  "v_mov_b32_e32 %[sv0], s20\n\t"                                                \
  "v_mov_b32_e32 %[sv1], s21\n\t"                                                \
  "v_mov_b32_e32 %[sv2], s22\n\t"                                                \
  "v_mov_b32_e32 %[sv3], s23\n\t"                                                \
  "v_mov_b32_e32 %[sv4], s13\n\t"                                                \
  "v_mov_b64_e32 %[s0], v[4:5]\n\t"                                              \
  "v_mov_b64_e32 %[s1], v[6:7]\n\t"                                              \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_mov_b64_e32 %[z0], v[0:1]\n\t"                                              \
  "v_mov_b64_e32 %[z1], v[2:3]\n\t"                                              \
  "s_mov_b32 %[st], s12\n\t"                                                     \
  "s_cmp_eq_u32 s10, .Lsc_acc0_%=-.Lsc_tab%=\n\t"                                \
  "s_cbranch_scc1 .Lsc_exit%=\n\t"                                               \
  "s_add_u32 %[st], %[st], 16\n\t"                                               \
  "s_cmp_eq_u32 s10, .Lsc_acc1_%=-.Lsc_tab%=\n\t"                                \
  "s_cbranch_scc1 .Lsc_exit%=\n\t"                                               \
  "s_add_u32 %[st], %[st], 16\n\t"                                               \
  "s_cmp_eq_u32 s10, .Lsc_acc2_%=-.Lsc_tab%=\n\t"                                \
  "s_cbranch_scc1 .Lsc_exit%=\n\t"                                               \
  "s_add_u32 %[st], %[st], 16\n"                                                 \
  ".Lsc_exit%=:\n\t"                                                             \
  "s_waitcnt lgkmcnt(0)\n\t"

#define BSR_STREAM_CHUNK_CLOBBERS                                                                                      \
  "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17",   \
  "v18", "v19", "v20", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", \
  "v38", "v39", "s8", "s9", "s10", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", \
  "s24", "s25", "s26", "s27", "s28", "s29", "vcc", "scc", "memory"
// clang-format on
