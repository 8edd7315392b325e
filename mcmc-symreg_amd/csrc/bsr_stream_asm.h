// The tape interpreter of the streaming row pass (bsr_stream.hip) for one-block chunks, in gfx950 assembly.
//
// Why assembly: the interpreter is a SCALAR program (extract an entry, branch to its operator, fetch a slot number) around
// two or three fp64 instructions per entry, and a CU has one scalar unit for its sixteen waves (1.17 cycles
// per scalar instruction and CU).  The compiler's loop spends 17 scalar instructions per entry (a compare tree, copies of
// the 64-bit code words, re-materialised constants); a jump into a table of 128-byte operator slots spends five:
//
//     s_lshl_b32 t, code.lo, 7 ; s_and_b32 t, t, 0x780 ; s_lshr_b64 code, code, 4 ; s_or_b32 pc.lo, table.lo, t ; s_setpc_b64 pc
//
// (the table is 2 KB-aligned: the OR cannot carry).  An entry's 4-bit code is operator + 1 (BSR_OP_* of include/bsr_hip.h,
// BSR_SOP_* of bsr_internal.h); 0 ends the tape -- what a right shift brings in behind the last entry.  `log` (operator
// 15) has no code here: the host gives such tapes to the stack machine.
//
// What runs here are the operators of one or two instructions per value, the protected divisions and the cube -- each the
// instruction sequence the compiler emits for the C++ of tape_fast / bsr_device.h (IEEE division by v_div_scale / v_rcp /
// two Newton steps / v_div_fmas / v_div_fixup; op_cube's compensated product), so a row's value is the same bits on
// either path (tests/test_gpu_stream.py compares them).  sin, cos and exp leave the block with their code in `st`, the
// interpreter's state in the operands; the caller runs the C++ routine on the two values and comes back (`resume`).
//
// Registers: the accumulator (two rows of the lane: v[0:3]), the saved value of a pending binary operator
// (v[4:7]), an operand (v[8:11]) and the division's temporaries (v[12:19], v20 for addresses) are fixed and
// declared clobbered -- an operand of the asm statement cannot be addressed by halves, and ds_read_b128 wants four
// consecutive registers; the same goes for s[16:27].  All of them are registers a callee may overwrite anyway (the
// calling convention's caller-saved ones): what the kernel keeps across its rare calls -- four sets of sums -- stays in
// the callee-saved registers.  Hazards the assembler does not pad for (gfx940 family): a
// transcendental's result needs one independent instruction before its first use; VCC written by a VALU compare needs two
// wait states before v_cndmask reads it and four before v_div_fmas.
#pragma once

// clang-format off
#define BSR_SA_DISPATCH                          \
  "s_lshl_b32 s24, s16, 7\n\t"                   \
  "s_and_b32 s24, s24, 0x780\n\t"                \
  "s_lshr_b64 s[16:17], s[16:17], 4\n\t"         \
  "s_or_b32 s22, s20, s24\n\t"                   \
  "s_setpc_b64 s[22:23]\n\t"

// the next terminal's LDS address into v103: slot number (low byte of the slot word) x 1024 bytes behind the lane's pair
#define BSR_SA_SLOT_ADDR                         \
  "s_and_b32 s24, s18, 0xff\n\t"                 \
  "s_lshr_b64 s[18:19], s[18:19], 8\n\t"         \
  "v_lshl_add_u32 v20, s24, 10, %[lc]\n\t"

// q = num / x by the compiler's expansion of an IEEE fp64 division, then x <- (x != 0) ? q : 0.
// x: the register pair of the value (hi, lo given apart for the selects); num: a register pair or 1.0
#define BSR_SA_DIV(x, xlo, xhi, num)                                   \
  "v_div_scale_f64 v[12:13], s[26:27], " x ", " x ", " num "\n\t"    \
  "v_rcp_f64_e32 v[14:15], v[12:13]\n\t"                           \
  "v_div_scale_f64 v[16:17], vcc, " num ", " x ", " num "\n\t"       \
  "v_fma_f64 v[18:19], -v[12:13], v[14:15], 1.0\n\t"             \
  "v_fmac_f64_e32 v[14:15], v[14:15], v[18:19]\n\t"              \
  "v_fma_f64 v[18:19], -v[12:13], v[14:15], 1.0\n\t"             \
  "v_fmac_f64_e32 v[14:15], v[14:15], v[18:19]\n\t"              \
  "v_mul_f64 v[18:19], v[16:17], v[14:15]\n\t"                   \
  "v_fma_f64 v[12:13], -v[12:13], v[18:19], v[16:17]\n\t"      \
  "v_div_fmas_f64 v[12:13], v[12:13], v[14:15], v[18:19]\n\t"  \
  "v_div_fixup_f64 v[12:13], v[12:13], " x ", " num "\n\t"         \
  "v_cmp_neq_f64_e32 vcc, 0, " x "\n\t"                                \
  "s_nop 1\n\t"                                                        \
  "v_cndmask_b32_e32 " xhi ", 0, v13, vcc\n\t"                        \
  "v_cndmask_b32_e32 " xlo ", 0, v12, vcc\n\t"

// x <- op_cube(x) (bsr_device.h): x2 = x x, e = fma(x, x, -x2), p = x2 x, pe = fma(x2, x, -p), r = p + (pe + e x);
// isfinite(r) ? r : p
#define BSR_SA_CUBE(x, xlo, xhi)                                       \
  "v_mul_f64 v[12:13], " x ", " x "\n\t"                             \
  "v_fma_f64 v[14:15], " x ", " x ", -v[12:13]\n\t"                \
  "v_mul_f64 v[16:17], " x ", v[12:13]\n\t"                        \
  "v_fma_f64 v[18:19], v[12:13], " x ", -v[16:17]\n\t"           \
  "v_mul_f64 v[14:15], " x ", v[14:15]\n\t"                        \
  "v_add_f64 v[14:15], v[18:19], v[14:15]\n\t"                   \
  "v_add_f64 v[14:15], v[16:17], v[14:15]\n\t"                   \
  "v_cmp_class_f64_e64 vcc, v[14:15], s24\n\t"                       \
  "s_nop 1\n\t"                                                        \
  "v_cndmask_b32_e32 " xhi ", v17, v15, vcc\n\t"                     \
  "v_cndmask_b32_e32 " xlo ", v16, v14, vcc\n\t"

#define BSR_SA_A0 "v[0:1]"
#define BSR_SA_A1 "v[2:3]"
#define BSR_SA_S0 "v[4:5]"
#define BSR_SA_S1 "v[6:7]"
#define BSR_SA_P0 "v[8:9]"
#define BSR_SA_P1 "v[10:11]"

// acc <- saved OP acc (a binary operator pops the saved value), acc <- acc OP operand (a fused terminal)
#define BSR_SA_BIN(ins, neg)                                                     \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  ins " " BSR_SA_A0 ", " BSR_SA_S0 ", " neg BSR_SA_A0 "\n\t"                     \
  ins " " BSR_SA_A1 ", " BSR_SA_S1 ", " neg BSR_SA_A1 "\n\t"                     \
  BSR_SA_DISPATCH
#define BSR_SA_BIN_T(ins)                                                        \
  BSR_SA_SLOT_ADDR                                                               \
  "ds_read_b128 v[8:11], v20\n\t"                                            \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  ins " " BSR_SA_A0 ", " BSR_SA_A0 ", " BSR_SA_P0 "\n\t"                         \
  ins " " BSR_SA_A1 ", " BSR_SA_A1 ", " BSR_SA_P1 "\n\t"                         \
  BSR_SA_DISPATCH
#define BSR_SA_LEAVE(code)                                                       \
  "s_mov_b32 %[st], " code "\n\t"                                                \
  "s_branch .Lsa_leave%=\n\t"

#define BSR_SA_SLOT(n) ".p2align 7\n.Lsa_op" n "_%=:\n\t"

#define BSR_STREAM_INTERP_ASM                                                    \
  "s_getpc_b64 s[20:21]\n"                                                       \
  ".Lsa_pc%=:\n\t"                                                               \
  "s_add_u32 s20, s20, .Lsa_tab%=-.Lsa_pc%=\n\t"                                 \
  "s_addc_u32 s21, s21, 0\n\t"                                                   \
  "s_mov_b32 s23, s21\n\t"                                                       \
  "s_cmp_eq_u32 %[resume], 0\n\t"                                                \
  "s_cbranch_scc0 .Lsa_resume%=\n\t"                                             \
  "v_add_u32_e32 v20, %[first], %[lc]\n\t"                                      \
  "ds_read_b128 v[0:3], v20\n\t"                                            \
  "s_mov_b64 s[16:17], %[code]\n\t"                                              \
  "s_mov_b64 s[18:19], %[sl]\n\t"                                                \
  "s_mov_b32 s25, %[ln]\n\t"                                                     \
  BSR_SA_DISPATCH                                                                \
  ".Lsa_resume%=:\n\t"   /* the state a leave put into %[stv]: five words, the same in every lane */ \
  "v_readfirstlane_b32 s16, %[sv0]\n\t"                                          \
  "v_readfirstlane_b32 s17, %[sv1]\n\t"                                          \
  "v_readfirstlane_b32 s18, %[sv2]\n\t"                                          \
  "v_readfirstlane_b32 s19, %[sv3]\n\t"                                          \
  "v_readfirstlane_b32 s25, %[sv4]\n\t"                                          \
  "v_mov_b64_e32 " BSR_SA_A0 ", %[z0]\n\t"                                       \
  "v_mov_b64_e32 " BSR_SA_A1 ", %[z1]\n\t"                                       \
  "v_mov_b64_e32 " BSR_SA_S0 ", %[s0]\n\t"                                       \
  "v_mov_b64_e32 " BSR_SA_S1 ", %[s1]\n\t"                                       \
  BSR_SA_DISPATCH                                                                \
  ".p2align 11\n"                                                                \
  ".Lsa_tab%=:\n\t"                                                              \
  /* 0: end of the tape */                                                       \
  "s_mov_b32 %[st], 0\n\t"                                                       \
  "s_branch .Lsa_out%=\n\t"                                                      \
  BSR_SA_SLOT("1") /* inv */                                                     \
  "s_branch .Lsa_inv%=\n\t"                                                      \
  BSR_SA_SLOT("2") /* ln: a x + b, two roundings; (a, b) from the wave's pairs in LDS (every lane the same 16 bytes) */ \
  "v_mov_b32_e32 v20, s25\n\t"                                                  \
  "ds_read_b128 v[8:11], v20\n\t"                                            \
  "s_add_u32 s25, s25, 16\n\t"                                                   \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_mul_f64 " BSR_SA_A0 ", " BSR_SA_P0 ", " BSR_SA_A0 "\n\t"                    \
  "v_mul_f64 " BSR_SA_A1 ", " BSR_SA_P0 ", " BSR_SA_A1 "\n\t"                    \
  "v_add_f64 " BSR_SA_A0 ", " BSR_SA_A0 ", " BSR_SA_P1 "\n\t"                    \
  "v_add_f64 " BSR_SA_A1 ", " BSR_SA_A1 ", " BSR_SA_P1 "\n\t"                    \
  BSR_SA_DISPATCH                                                                \
  BSR_SA_SLOT("3") /* neg */                                                     \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_xor_b32_e32 v1, 0x80000000, v1\n\t"                                     \
  "v_xor_b32_e32 v3, 0x80000000, v3\n\t"                                     \
  BSR_SA_DISPATCH                                                                \
  BSR_SA_SLOT("4") BSR_SA_LEAVE("4") /* sin */                                   \
  BSR_SA_SLOT("5") BSR_SA_LEAVE("5") /* cos */                                   \
  BSR_SA_SLOT("6") BSR_SA_LEAVE("6") /* exp */                                   \
  BSR_SA_SLOT("7") /* square */                                                  \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_mul_f64 " BSR_SA_A0 ", " BSR_SA_A0 ", " BSR_SA_A0 "\n\t"                    \
  "v_mul_f64 " BSR_SA_A1 ", " BSR_SA_A1 ", " BSR_SA_A1 "\n\t"                    \
  BSR_SA_DISPATCH                                                                \
  BSR_SA_SLOT("8") /* cubic */                                                   \
  "s_branch .Lsa_cube%=\n\t"                                                     \
  BSR_SA_SLOT("9") BSR_SA_BIN("v_add_f64", "")                                   \
  BSR_SA_SLOT("10") BSR_SA_BIN("v_mul_f64", "")                                  \
  BSR_SA_SLOT("11") /* terminal: the accumulator becomes the saved value */      \
  BSR_SA_SLOT_ADDR                                                               \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_mov_b64_e32 " BSR_SA_S0 ", " BSR_SA_A0 "\n\t"                               \
  "v_mov_b64_e32 " BSR_SA_S1 ", " BSR_SA_A1 "\n\t"                               \
  "ds_read_b128 v[0:3], v20\n\t"                                            \
  BSR_SA_DISPATCH                                                                \
  BSR_SA_SLOT("12") BSR_SA_BIN_T("v_add_f64")                                    \
  BSR_SA_SLOT("13") BSR_SA_BIN_T("v_mul_f64")                                    \
  BSR_SA_SLOT("14") BSR_SA_BIN("v_add_f64", "-") /* sub */                       \
  BSR_SA_SLOT("15") /* div, protected like inv */                                \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_SA_DIV(BSR_SA_A0, "v0", "v1", BSR_SA_S0)                               \
  BSR_SA_DIV(BSR_SA_A1, "v2", "v3", BSR_SA_S1)                               \
  BSR_SA_DISPATCH                                                                \
  ".Lsa_inv%=:\n\t"                                                              \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_SA_DIV(BSR_SA_A0, "v0", "v1", "1.0")                                   \
  BSR_SA_DIV(BSR_SA_A1, "v2", "v3", "1.0")                                   \
  BSR_SA_DISPATCH                                                                \
  ".Lsa_cube%=:\n\t"                                                             \
  "s_movk_i32 s24, 0x1f8\n\t"   /* the finite classes */                         \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  BSR_SA_CUBE(BSR_SA_A0, "v0", "v1")                                         \
  BSR_SA_CUBE(BSR_SA_A1, "v2", "v3")                                         \
  BSR_SA_DISPATCH                                                                \
  ".Lsa_leave%=:\n\t"                                                            \
  "v_mov_b32_e32 %[sv0], s16\n\t"                                                \
  "v_mov_b32_e32 %[sv1], s17\n\t"                                                \
  "v_mov_b32_e32 %[sv2], s18\n\t"                                                \
  "v_mov_b32_e32 %[sv3], s19\n\t"                                                \
  "v_mov_b32_e32 %[sv4], s25\n\t"                                                \
  "v_mov_b64_e32 %[s0], " BSR_SA_S0 "\n\t"                                       \
  "v_mov_b64_e32 %[s1], " BSR_SA_S1 "\n"                                         \
  ".Lsa_out%=:\n\t"                                                              \
  "s_waitcnt lgkmcnt(0)\n\t"                                                     \
  "v_mov_b64_e32 %[z0], " BSR_SA_A0 "\n\t"                                       \
  "v_mov_b64_e32 %[z1], " BSR_SA_A1 "\n\t"

#define BSR_STREAM_INTERP_CLOBBERS                                                                                     \
  "v20", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v0", "v1", \
  "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "s16", "s17", "s18", "s19", "s20",     \
  "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc", "scc", "memory"
// clang-format on
