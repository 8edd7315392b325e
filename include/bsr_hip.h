/*
 * bsr_hip.h -- C ABI of libbsr_hip.so: the MI355X (gfx950) implementation of the
 * MCMC-SymReg per-proposal likelihood hot path.
 *
 * The reference (ying531/MCMC-SymReg) is pure Python and has no FFI seam; the
 * seam is its Python surface.  The Python package `bsr` in
 * mcmc-symreg_amd/bsr binds these entry points with ctypes and keeps the
 * reference's BSR.fit/predict + Node/allcal/ylogLike/newProp surface on top.
 * Each entry point names the reference code it replaces (path:line under the
 * reference repository).
 *
 * Conventions
 *   - every function returns BSR_OK (0) or a negative BSR_E_* code; nothing
 *     throws across the boundary; bsr_last_error() gives the text
 *   - host buffers are caller-allocated and only touched during the call
 *   - a ctx owns its device memory, a main HIP stream, a stream per batch slot and (where the process has the
 *     CPUs) one or two submission threads that issue a batch's launches; the PUBLIC entry points are not
 *     thread-safe: one ctx per device, driven by one host thread at a time (ctypes drops the GIL during calls).
 *     The native sampler's worker threads go through slot-explicit internal entry points and a context lock.
 *   - all floating-point host buffers are IEEE binary64 whatever BSR_DTYPE_*
 *     the ctx computes in
 */
#ifndef BSR_HIP_H
#define BSR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSR_ABI_VERSION 1

/* limits */
#define BSR_MAX_K 8          /* trees per chain (reference default treeNum=3; paper uses up to 8) */
#define BSR_MAX_TAPE 16384   /* nodes per tape (depth<=12 binary tree has <= 8191 nodes) */
#define BSR_MAX_STACK 24     /* value-stack depth of the interpreter (incl. the accumulator) */

/* opcodes = index into the reference's operator table, codes/bsr_class.py:110-112
 * ['inv','ln','neg','sin','cos','exp','square','cubic','+','*'], then the terminal */
enum {
  BSR_OP_INV = 0, BSR_OP_LN = 1, BSR_OP_NEG = 2, BSR_OP_SIN = 3, BSR_OP_COS = 4,
  BSR_OP_EXP = 5, BSR_OP_SQUARE = 6, BSR_OP_CUBIC = 7, BSR_OP_ADD = 8, BSR_OP_MUL = 9,
  BSR_OP_TERMINAL = 10,
  /* 11, 12: reserved (fused entries of the device's opcode stream).
   * Extensions beyond the reference's table (semantics defined by oracle/bsr_oracle.py: allcal):
   *   sub  x - y                      (left - right, rows in tape order: left subtree first)
   *   div  y == 0 ? 0 : x / y         (protected like the reference's inv, codes/funcs.py:189-195)
   *   log  x == 0 ? 0 : log|x|        (natural logarithm; the reference's 'ln' is the affine map a*x+b) */
  BSR_OP_SUB = 13, BSR_OP_DIV = 14, BSR_OP_LOG = 15
};

enum { BSR_DTYPE_F64 = 0, BSR_DTYPE_F32 = 1 };

/* error codes */
enum {
  BSR_OK = 0,
  BSR_E_ARG = -1,       /* bad argument (null pointer, index out of range, ...) */
  BSR_E_HIP = -2,       /* HIP runtime error (text in bsr_last_error) */
  BSR_E_NODEVICE = -3,  /* no usable gfx950 device */
  BSR_E_TOOBIG = -4,    /* tape longer than BSR_MAX_TAPE / deeper than BSR_MAX_STACK / batch too large */
  BSR_E_TAPE = -5,      /* malformed tape (not a valid postfix program) */
  BSR_E_STATE = -6,     /* chain state not initialised for the requested operation */
  BSR_E_COMM = -7,      /* RCCL error */
  BSR_E_LINALG = -8     /* NaN reached the rank gate: the reference raises numpy.linalg.LinAlgError (codes/funcs.py:1226) */
};

/* One row of a postfix (RPN) tape = one node of codes/funcs.py:30-55 (Node).
 * Rows are in evaluation order; `left`/`right` are row indices of the children
 * inside the same tape (-1 = none) and keep the tree recoverable from the tape.
 * opcode 0..9: operator; `a`,`b` only meaningful for BSR_OP_LN (a*x+b).
 * opcode 10:   terminal, `feature` = column of X. */
typedef struct bsr_node {
  int32_t opcode;
  int32_t left;
  int32_t right;
  int32_t feature;
  double a;
  double b;
} bsr_node; /* 32 bytes */

/* per-proposal result flags */
#define BSR_F_INF 1u          /* candidate column (or a sibling column) holds +-inf: reference rank gate returns 0 */
#define BSR_F_NAN 2u          /* ... holds NaN: reference raises LinAlgError at codes/funcs.py:1226 */
#define BSR_F_RANKDEF 4u      /* rank(new_outputs) < K : rejected by the gate at codes/funcs.py:1226-1228 */
#define BSR_F_SCALE_RETRY 8u  /* K==1 only: column magnitude outside the accumulation range; rescored internally */
#define BSR_F_SV_BOUNDS 16u   /* the rank gate was settled by bounds far from its threshold (csrc/bsr_solve.h: solve_fast):
                               * smin is a lower bound of sigma_min (rank = K) or an upper bound (rank < K), smax an upper
                               * bound of sigma_max, each within sqrt(K); with rank < K, `rank` is itself a bound (< K) and
                               * loglik / sse / beta are NaN -- the reference returns before ylogLike there
                               * (codes/funcs.py:1226-1228).  Unset: singular values by one-sided Jacobi, everything exact. */

/* Data-side result of one Metropolis-Hastings proposal: everything
 * codes/funcs.py:1212-1235 (newProp) takes from the N rows. */
typedef struct bsr_score {
  double loglik;            /* ylogLike(y, new_outputs, new_sigma), codes/funcs.py:1147-1174 */
  double sse;               /* sum((y - XX@Beta)^2),               codes/funcs.py:1162 */
  double scale;             /* max|new_outputs|,                    codes/funcs.py:1149 */
  double maxabs;            /* max|candidate column| */
  double smin, smax;        /* extreme singular values of new_outputs (relative to an internal power-of-two scale);
                             * bounds when flags & BSR_F_SV_BOUNDS */
  double beta[BSR_MAX_K];   /* OLS weights on the scaled columns,   codes/funcs.py:1154-1155 */
  int32_t rank;             /* np.linalg.matrix_rank(new_outputs); 0 if inf present; -1 if NaN present; with
                             * BSR_F_SV_BOUNDS and rank < K: some value < K (the gate's verdict is what the path uses) */
  uint32_t flags;
} bsr_score;

/* State of one chain's K current trees ("old_outputs" of codes/funcs.py:1199-1224). */
typedef struct bsr_chain_info {
  double sse_old;           /* SSE of ylogLike(y, old_outputs, .), codes/funcs.py:1235; NaN if a column is non-finite */
  double scale_old;         /* max|old_outputs| */
  double maxabs[BSR_MAX_K]; /* per current column */
  double beta_old[BSR_MAX_K];
  uint32_t colflags[BSR_MAX_K]; /* BSR_F_INF / BSR_F_NAN per current column */
  int32_t rank_old;         /* matrix_rank(old_outputs) (informational) */
  int32_t pad;
} bsr_chain_info;

typedef struct bsr_ctx bsr_ctx;

/* ---- device / context ------------------------------------------------------ */

int bsr_abi_version(void);
int bsr_device_count(int* count);

/* Uploads the training data once.  X is row-major (N,d) as in a C-order numpy
 * array / DataFrame.values (the `indata` of codes/funcs.py:175); it is stored
 * feature-major on the device.  y may be NULL for an evaluate-only context
 * (allcal / predict).  K and n_chains size the chain caches (current columns and their orthonormal basis);
 * max_batch bounds the proposals / tapes of one call and sizes the per-batch descriptor, partial-sum and result
 * blocks (O(max_batch * N / 1024) doubles).  Candidate columns are not stored by the scoring pass; the
 * max_batch * N column buffer of bsr_eval_tapes is allocated on its first use.
 * CPU placement: the threads the LIBRARY starts (submission threads, the native sampler's workers) are placed on one L3
 * domain of the host (the one the creating thread is on; one per LOCAL_RANK when several ranks share the node) -- a
 * batch crosses three threads, and across sockets that costs up to 20 % of the pipelined rate.  The caller's own
 * affinity is never changed unless BSR_PIN=1 is set (then it is confined to the same CPUs, once per process);
 * BSR_PIN=0: no placement at all; BSR_PIN_CPUS gives the CPU list. */
int bsr_ctx_create(bsr_ctx** out, int device, int64_t N, int32_t d, const double* X_rowmajor,
                   const double* y, int32_t K, int32_t n_chains, int32_t max_batch, int32_t dtype);

/* bsr_ctx_create for a caller that knows what its batches look like (no reference counterpart): `typical_chains` distinct
 * chains and `typical_batch` proposals per batch at most, as a rule (0: the context's limits, = bsr_ctx_create).  The
 * row pass's geometry -- how long a slice of rows a workgroup keeps in LDS -- is fixed for the life of a context (a
 * proposal's sums must not depend on the batch it travels in) and has to leave room for a batch's columns: the features,
 * y, and K basis columns per chain IN THE BATCH.  The native sampler submits its chains in groups (a quarter of them per
 * batch): a context sized for every chain at once would cut the rows into three times as many, shorter slices than
 * those batches need.  A batch beyond the typical still scores, to the same bytes, through the chunked kernel. */
int bsr_ctx_create_tuned(bsr_ctx** out, int device, int64_t N, int32_t d, const double* X_rowmajor, const double* y,
                         int32_t K, int32_t n_chains, int32_t max_batch, int32_t dtype, int32_t typical_chains,
                         int32_t typical_batch);
int bsr_ctx_destroy(bsr_ctx* ctx);
const char* bsr_last_error(const bsr_ctx* ctx); /* ctx may be NULL: last error of a failed create */

/* ---- allcal: codes/funcs.py:175-220 ---------------------------------------- */

/* Evaluates n_tapes tapes over all N rows.  rows = concatenated tapes,
 * tape_off[n_tapes+1] = row offsets.  out_cols (n_tapes*N, tape-major) may be
 * NULL; maxabs[n_tapes] / flags[n_tapes] (BSR_F_INF|BSR_F_NAN) may be NULL.
 * n_tapes <= max_batch. */
int bsr_eval_tapes(bsr_ctx* ctx, const bsr_node* rows, const int32_t* tape_off, int32_t n_tapes,
                   double* out_cols, double* maxabs, uint32_t* flags);

/* ---- chain state: the K current columns of codes/funcs.py:1212-1224 -------- */

/* (Re)computes current column k of `chain` from its tape (initial trees, codes/bsr_class.py:128-151). */
int bsr_set_current(bsr_ctx* ctx, int32_t chain, int32_t k, const bsr_node* tape, int32_t len);
/* Adopts candidate `slot` of the LAST scored batch (bsr_score_batch / the last bsr_score_wait) as current column k
 * of `chain` (an accepted proposal, codes/bsr_class.py:200-204).  Candidate columns are not kept by the scoring
 * pass; the still-staged tape is re-run straight into the chain cache. */
int bsr_commit(bsr_ctx* ctx, int32_t chain, int32_t k, int32_t slot);
/* Rebuilds the chain's cached factors (ONE orthonormal basis of the chain's K current columns, old-state SSE). Must be
 * called after bsr_set_current/bsr_commit and before the next bsr_score_batch on that chain. */
int bsr_refresh(bsr_ctx* ctx, int32_t chain, bsr_chain_info* info);

/* ---- scoring: codes/funcs.py:1212-1235 + 1147-1174 ------------------------- */

/* Scores B proposals in one pass.  Proposal i replaces tree which_k[i] of chain[i] by tape i and is
 * scored with noise scale sigma[i] (new_sigma of codes/funcs.py:1195).  All proposals are scored
 * against the chains' CURRENT columns (speculative batches are exact because a rejected proposal
 * leaves the chain unchanged, codes/funcs.py:1300-1303).  B <= max_batch. */
int bsr_score_batch(bsr_ctx* ctx, const bsr_node* rows, const int32_t* tape_off, const int32_t* chain,
                    const int32_t* which_k, const double* sigma, int32_t B, bsr_score* out);

/* Asynchronous form of bsr_score_batch: submit enqueues the tape upload and the kernels (which write their results
 * into pinned host memory) and returns at once with a ticket; wait blocks until that batch is done and copies its B
 * results.  Up to BSR_MAX_INFLIGHT batches may be in
 * flight (tickets are handed out round-robin; wait for them in submission order), each on its own HIP stream, so the
 * host stages batch i+1 and the small per-proposal kernels of batch i overlap the row pass of batch i+1.  Chains of
 * a batch must not depend on accepts of a batch still in flight.  bsr_commit refers to the batch most recently
 * waited for and returns BSR_E_STATE once that batch's slot has been submitted to again. */
#define BSR_MAX_INFLIGHT 8
int bsr_score_submit(bsr_ctx* ctx, const bsr_node* rows, const int32_t* tape_off, const int32_t* chain,
                     const int32_t* which_k, const double* sigma, int32_t B, int32_t* ticket);
int bsr_score_wait(bsr_ctx* ctx, int32_t ticket, bsr_score* out);

/* ---- device-side Metropolis-Hastings step: codes/funcs.py:1226-1306 ----------------------
 * The scalar tail of newProp -- log-ratio assembly, the accept test and the search for the first proposal of a
 * speculative run that changes the chain -- evaluated on the device behind the scoring kernels, so that a batch
 * answers with one event per chain.  The host supplies, per proposal, the terms it knows before scoring
 * (terms[8*i + ...]):
 *   0 yll          ylogLike of the chain's current state at its current sigma              (funcs.py:1235)
 *   1 log_struc    structure-prior ratio, old - new, as funcs.py:1245 / :1289 assemble it
 *   2 log_q        log max(1e-5, Qinv/Q)                                                     (funcs.py:1249)
 *   3 log_h, 4 log_d   log max(1e-5, hratio), log max(1e-5, detjacob); used iff BSR_MH_JUMP   (funcs.py:1250-1251)
 *   5 lig_new, 6 lig_old   log invgamma.pdf(new_sigma; 4), log invgamma.pdf(sigma; 4)        (funcs.py:1253-1254)
 *   7 log_u        log of the accept uniform (funcs.py:1299); ignored with BSR_MH_NO_UNIFORM
 * and the device forms logR in the reference's order of additions (bit-identical to the host's), accepts iff
 * !(log_u >= min(logR, 0)) (funcs.py:1300-1304, NaN semantics included) and scans each chain's run.
 * flags[i]: BSR_MH_JUMP = 'shrinkage'/'expansion' move; BSR_MH_NO_UNIFORM = the host speculated this proposal as a
 * rank-gate rejection and drew no uniform behind it (codes/funcs.py:1226-1228).
 * span_off[n_spans + 1]: proposals span_off[j] .. span_off[j+1]-1 are consecutive proposals of one chain.        */
#define BSR_MH_JUMP 1
#define BSR_MH_NO_UNIFORM 2
enum { BSR_EV_NONE = 0,       /* every proposal of the run was rejected as speculated */
       BSR_EV_ACCEPT = 1,     /* proposal `index` is accepted */
       BSR_EV_GATE = 2,       /* proposal `index` is rejected by the rank gate but a uniform had been drawn behind it */
       BSR_EV_GATE_PASSED = 3 /* proposal `index` passes the gate although it was speculated as rejected */ };
typedef struct bsr_event {
  int32_t index;            /* position inside the span; span length when kind == BSR_EV_NONE */
  int32_t kind;
  double logR;              /* of proposal `index` (NaN when no logR was formed) */
} bsr_event;
int bsr_score_submit_mh(bsr_ctx* ctx, const bsr_node* rows, const int32_t* tape_off, const int32_t* chain,
                        const int32_t* which_k, const double* sigma, int32_t B, const double* terms8,
                        const int32_t* flags, const int32_t* span_off, int32_t n_spans, int32_t* ticket);
/* out may be NULL (the per-proposal scores stay available through bsr_score_wait on the same ticket). */
int bsr_score_wait_mh(bsr_ctx* ctx, int32_t ticket, bsr_score* out, bsr_event* events);

/* ---- on-accept / initial OLS with intercept: codes/bsr_class.py:147-163, 211-233 */

/* beta_out[K+1] = Beta/scale (intercept first), rmse_out = sqrt(mean((fitted-y)^2)). */
int bsr_fit_beta(bsr_ctx* ctx, int32_t chain, double* beta_out, double* rmse_out);

/* Copies the chain's K current columns to the host (K*N, column-major); for predict / tests. */
int bsr_get_current(bsr_ctx* ctx, int32_t chain, double* out_cols);

/* ---- ylogLike on caller-provided outputs: codes/funcs.py:1147-1174 --------- */

/* outputs is row-major (N,K).  skipna != 0 reproduces the pandas Series.sum(skipna=True) behaviour the
 * reference gets when y is a Series (codes/funcs.py:1162).  Runs on `device`; includes PCIe transfers. */
int bsr_yloglike_host(int device, int64_t N, int32_t K, const double* outputs_rowmajor, const double* y,
                      double sigma, int32_t skipna, double* loglik, double* sse, double* scale, double* beta,
                      int32_t* rank);

/* ---- timing hooks for bench.py (HIP events on the ctx stream) --------------- */

/* HIP-event timing of the LAST waited batch, in microseconds.  level 1: events around the row pass only
 * (us[0] = tree-eval + projection kernel; what bench.py uses in its timed region); level 2: events around every
 * kernel: us[1]=K x K solve, us[2]=residual pass, us[3]=finalise, us[4]=first launch to last (diagnostic: each
 * event adds a few microseconds of its own).  level 0 disables. */
int bsr_set_profiling(bsr_ctx* ctx, int32_t level);
int bsr_last_timing(bsr_ctx* ctx, double* us5);

/* What the context decided for this process and machine, for diagnostics and the multi-rank bench line (the reference
 * has no counterpart: it is one thread on one CPU, codes/bsr_class.py:99):
 *   info[0] submission threads   info[1] CPUs the library's threads are placed on (0: not placed)
 *   info[2] 1 if the caller was confined too (BSR_PIN=1)   info[3] CPU budget of this rank, x100
 *   info[4] tape groups T   info[5] row slices   info[6] blocks per slice
 *   info[7] the scoring row pass: 1 slices staged whole (k_tile1), 3 the same with the tape loop in assembly (k_tile1a),
 *           2 slices streamed through LDS (k_stream), 0 chunked k_tile / work-queue k_rows */
int bsr_ctx_info(const bsr_ctx* ctx, int32_t* info8);

/* How the LAST waited batch of `ticket` was scored, for the bench's depth-stress leg (no reference counterpart):
 *   stats[0] tapes   stats[1] tapes the row pass's assembly interpreter took (64-bit programs: at most 16-17 entries,
 *   8 terminals, one value below the accumulator)   stats[2] chain tapes (no value stack at all)
 *   stats[3] stream entries of the batch after the fusions (a `terminal, +|*` pair or a derived column is one entry) */
int bsr_batch_stats(const bsr_ctx* ctx, int32_t ticket, int32_t* stats4);

/* Diagnostics for tools/tile_stamps.py (no reference counterpart; off unless BSR_TILE_STAMPS is set when the context is
 * created): the shader-clock samples of the last tile-pass launch, out[workgroups][16 waves][8] uint64;
 * geom5 = {tape groups, row slices, blocks per slice, blocks, workgroups * 100 + 1}.  Returns the number of workgroups
 * copied (<= max_wgs), 0 when stamps are off, BSR_E_* (< 0) on failure. */
int bsr_debug_tile_stamps(bsr_ctx* ctx, unsigned long long* out, int32_t max_wgs, int32_t* geom5);
/* BSR_TILE_STAMPS=R keeps the stamps of the last R tile-pass launches (launch n in block n mod R), word 5 of a wave's
 * record = HW_ID | XCC_ID << 32 (which CU it ran on), words 7 / 6 = start / end on the 100 MHz clock: the busy intervals
 * of every CU over many pipelined launches (tools/cu_occupancy.py).  out[R][slots][16][8] uint64; info4 = {R, workgroup
 * slots per block, workgroups per launch, launches so far}.  Returns the blocks copied, 0 when stamps are off. */
int bsr_debug_tile_stamp_ring(bsr_ctx* ctx, unsigned long long* out, int32_t max_blocks, int32_t* info4);

/* How scoring batches reach the GPU (no reference counterpart):
 *   info[0] 1 if the context dispatches them itself (AQL packets into its own ROCr queues, csrc/bsr_aql.h), 0 if through
 *           HIP launches on the slots' streams (BSR_AQL=0, no large BAR, or after a failure)
 *   info[1] those queues   info[2] scoring batches dispatched directly so far   info[3] ... through a stream (batches
 *   timed kernel by kernel, a slot whose stream still had a commit or a copy in flight, every batch when info[0] is 0)
 *   info[4..7] reserved, 0 */
int bsr_dispatch_info(const bsr_ctx* ctx, int64_t* info8);

/* Where this process's library threads were placed (once per process, by the first bsr_ctx_create; csrc/bsr_place.h):
 *   info[0] 1 if a placement was made   info[1] its CPUs   info[2] NUMA node of the context's GPU as sysfs reports it
 *   (/sys/bus/pci/devices/<bdf>/numa_node; -1: unknown, -2: no placement)   info[3] 1 if the caller was confined too */
int bsr_place_info(int32_t* info4);

/* ---- multi-GPU: one process per GPU, one gather of accepted trees */

#define BSR_COMM_ID_BYTES 128
int bsr_comm_unique_id(void* id128);                       /* rank 0 creates, host side distributes */
int bsr_comm_init(bsr_ctx* ctx, int32_t nranks, int32_t rank, const void* id128);
/* all-gather of fixed-size records over RCCL: recv holds nranks*bytes_per_rank bytes, rank-major */
int bsr_comm_allgather(bsr_ctx* ctx, const void* send, void* recv, int64_t bytes_per_rank);
int bsr_comm_destroy(bsr_ctx* ctx);

/* ---- native sampler: Prop/auxProp/grow/fStruc/newProp + the chain loop of BSR.fit in C++ ----
 * Replaces, draw for draw, codes/funcs.py:74-119, 349-398, 406-1138, 1184-1306 and codes/bsr_class.py:99-273; the
 * data side goes through the entry points above.  Random streams are numpy legacy RandomState streams (MT19937,
 * polar normal with cached value, masked-rejection randint), one per chain. */
typedef struct bsr_engine bsr_engine;

typedef struct bsr_trace {   /* one consumed proposal (optional diagnostics / parity tests) */
  int32_t chain, count, action, change, rank, accepted, n_nodes, pad;
  double Q, Qinv, new_sigma, new_sa2, new_sb2, yllstar, yll, logR, u, rmse;
  uint64_t tree_hash;        /* FNV-1a over the proposed tree's pre-order (type, operator|100+feature) */
} bsr_trace;

int bsr_engine_create(bsr_engine** out, bsr_ctx* ctx, int32_t n_chains, int32_t K, int64_t N, int32_t n_feature,
                      double beta, int32_t val, int32_t y_is_series);
int bsr_engine_destroy(bsr_engine* e);
const char* bsr_engine_last_error(const bsr_engine* e);
/* reject = 0 (default): a NaN candidate aborts the run with BSR_E_LINALG, as the reference's matrix_rank call raises
 * (codes/funcs.py:1226); reject != 0: it is handled like a rank-gate rejection (documented divergence). */
int bsr_engine_set_nan_policy(bsr_engine* e, int32_t reject);
/* Operator table and prior weights (Ops / Op_weights / Op_type, hard-coded at codes/bsr_class.py:110-112): entry i has
 * opcode opcodes[i] (arity follows from the opcode) and weight weights[i]; at most 16 entries.  Default: the reference's
 * ten operators with weight 1/10.  Call before bsr_engine_init_chain. */
int bsr_engine_set_ops(bsr_engine* e, int32_t n_ops, const int32_t* opcodes, const double* weights);
int bsr_engine_seed(bsr_engine* e, int32_t chain, uint32_t seed);   /* == np.random.seed(seed) for that chain */
int bsr_engine_set_rng(bsr_engine* e, int32_t chain, const uint32_t* key624, int32_t pos, int32_t has_gauss,
                       double gauss);                                /* == np.random.set_state(...) */
int bsr_engine_get_rng(bsr_engine* e, int32_t chain, uint32_t* key624, int32_t* pos, int32_t* has_gauss,
                       double* gauss);
int bsr_engine_init_chain(bsr_engine* e, int32_t chain);            /* codes/bsr_class.py:116-163 */
int bsr_engine_run(bsr_engine* e, int32_t batch_per_chain, int64_t max_props, bsr_trace* trace, int64_t trace_cap,
                   int64_t* n_trace, int32_t max_batch);
/* tapes: K*tape_cap rows; current != 0: the chain's current trees, else the `Roots` list BSR.fit would append
 * (codes/bsr_class.py:270-273). counters[5] = proposals, accepts, rank-gate rejections, discarded, done. */
int bsr_engine_chain_result(bsr_engine* e, int32_t chain, bsr_node* tapes, int32_t tape_cap, int32_t* tape_len,
                            double* beta, double* errs, int32_t errs_cap, int32_t* n_errs, int64_t* counters,
                            double* sigma, int32_t current);
/* The sampler's score memo (no reference counterpart; BSR_ENGINE_MEMO=0 turns it off): a chain that proposes, in an
 * unchanged state, a (tree incl. ln parameters, slot k) it has had scored before takes rank and SSE from a table and
 * recomputes the log-likelihood for the proposal's own sigma (codes/funcs.py:1162-1173: SSE does not depend on sigma).
 * stats[0] = proposals answered from the table, stats[1] = lookups, over the chain's life. */
int bsr_engine_memo_stats(const bsr_engine* e, int32_t chain, int64_t* stats2);
int bsr_rng_selftest(uint32_t seed, int32_t n, const int32_t* kind, const int64_t* lo, const int64_t* hi, double* out);

#ifdef __cplusplus
}
#endif
#endif /* BSR_HIP_H */
