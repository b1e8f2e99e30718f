/*
 * ebnerd_hip.h -- C ABI of the MI355X (gfx950) NRMS / NRMSDocVec hot path.
 *
 * The reference (ebanalyse/ebnerd-benchmark) has no FFI: its hot path is Python
 * calling TensorFlow ops.  This header is the boundary a maintainer would bind
 * instead (ctypes stub in INTEGRATION.md).  Every entry point names the reference
 * interface it replaces (paths relative to src/ebrec/models/newsrec/).
 *
 * Conventions
 *   - all pointers are DEVICE pointers owned by the caller (e.g. torch tensors'
 *     data_ptr()); all matrices are row-major fp32 unless stated otherwise;
 *   - `stream` is a hipStream_t passed as void*; calls only ENQUEUE work on it and
 *     return immediately (no host sync, no allocation -> hipGraph-capturable);
 *   - return 0 on success, a positive hipError_t from the launch, or a negative
 *     EBN_ERR_* argument code (the host layer raises on non-zero);
 *   - no global mutable state: re-entrant, callable from several host threads on
 *     different streams;
 *   - step-dependent scalars (Adam step size, dropout keys) live in a DEVICE
 *     `ebn_step_state` so a captured graph replays with fresh values.
 */
#ifndef EBNERD_HIP_H
#define EBNERD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EBN_ABI_VERSION 1

#define EBN_OK 0
#define EBN_ERR_BAD_ARG (-1)     /* null pointer / negative size */
#define EBN_ERR_UNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define EBN_ERR_ALIGN (-3)       /* pointer or leading dimension not 16-byte friendly */

typedef void* ebn_stream_t;

/* Dropout call sites (keys in ebn_step_state.drop_key). nrms.py:136 / nrms.py:154 /
 * nrms_docvec.py:124 (site EBN_SITE_MLP0 + layer). */
#define EBN_SITE_NEWS_IN 0
#define EBN_SITE_NEWS_ATT 1
#define EBN_SITE_MLP0 8
#define EBN_N_SITES 12

/* 64-byte device-resident per-step state. */
typedef struct ebn_step_state {
  uint32_t step;       /* optimizer step t, 1-based after the first advance */
  uint32_t seed;       /* model seed (NRMSModel(seed=...), nrms.py:33) */
  float adam_alpha;    /* lr*sqrt(1-b2^t)/(1-b1^t), refreshed by ebn_step_advance */
  float lr;            /* current learning rate (ReduceLROnPlateau rewrites it) */
  uint32_t drop_key[EBN_N_SITES];
} ebn_step_state;

int ebn_abi_version(void);
const char* ebn_error_string(int code);
/* Kernel launches this library has enqueued in this process so far (captured ones included: a launch recorded into a hipGraph counts
 * once, at capture).  Diagnostics: bench.py reports `launches_per_step` as the difference around one eager step.  No reference
 * counterpart (Keras' train_function, nrms.py:92-99 compile + fit, hides its op launches).                                          */
int64_t ebn_launch_count(void);

/* step++ ; adam_alpha and the dropout keys for the new step. One tiny kernel. */
int ebn_step_advance(ebn_step_state* st, double beta1, double beta2, ebn_stream_t stream);

/* ---- a1  tf.keras.layers.Embedding (nrms.py:125-134) ---------------------------
 * out[r,:] = table[ids[r],:] (* inverted-dropout multiplier of nrms.py:136 when
 * drop_p > 0 and st != NULL).  ids outside [0,V) write a zero row and set *oob_flag
 * (may be NULL) to 1: the host layer turns that into an IndexError.               */
int ebn_gather_rows_f32(const int32_t* ids, const float* table, float* out, int64_t n_tok,
                        int32_t D, int64_t V, const ebn_step_state* st, int32_t site,
                        float drop_p, int32_t* oob_flag, ebn_stream_t stream);

/* ---- a13  device-side batch assembly (dataloader.py:169-179: lookup_article_matrix[article rows]) ----------
 * ids_out[r*T + t] = token_matrix[art_idx[r]*T + t]: the loader ships B*(H+C) article-row numbers per step
 * instead of B*(H+C)*T token ids; row 0 of the matrix is the "unknown / padded article" title.  Integer copy,
 * bit-exact.  art_idx outside [0, n_rows) writes zeros and sets *oob_flag (may be NULL).                       */
int ebn_expand_titles_i32(const int32_t* art_idx, const int32_t* token_matrix, int32_t* ids_out,
                          int64_t n_titles, int32_t T, int64_t n_rows, int32_t* oob_flag,
                          ebn_stream_t stream);

/* Backward of a1: dTable[ids[r],:] += dX[r,:] * dropout multiplier. dTable must be
 * zeroed by the caller (dense gradient, Keras-Adam dense semantics).   */
int ebn_embedding_grad_scatter_f32(const int32_t* ids, const float* dX, float* dTable,
                                   int64_t n_tok, int32_t D, int64_t V,
                                   const ebn_step_state* st, int32_t site, float drop_p,
                                   ebn_stream_t stream);

/* Deterministic form of the same backward: gradients are accumulated as 2^40-scaled 64-bit integers (integer
 * atomics are associative, so the result does not depend on the order in which duplicate tokens arrive -- hot rows
 * such as token 0 of padded history get the same bits every run), then ebn_fixed_to_f32 converts the accumulator
 * to the fp32 dense gradient and zeroes it for the next step.  Resolution 9e-13; |sum| must stay below 2^23 -- a single
 * term >= 2^21 (or NaN) or an accumulated |sum| >= 2^22 sets *range_flag (may be NULL) to 1 instead of wrapping silently:
 * the host layer raises FloatingPointError.                                                                       */
int ebn_embedding_grad_scatter_fixed(const int32_t* ids, const float* dX, int64_t* acc, int64_t n_tok,
                                     int32_t D, int64_t V, const ebn_step_state* st, int32_t site,
                                     float drop_p, int32_t* range_flag, ebn_stream_t stream);
/* ebn_embedding_grad_scatter_fixed combines the duplicates among each run of 64 consecutive tokens before they reach the
 * atomics (one atomic per distinct id, column and run: padded titles and Zipfian tokens hammer table row 0 -- reference
 * _behaviors.py:647-654, dataloader.py:43).  This is the plain form, one atomic per non-zero gradient element: same sums
 * bit for bit (wrapping integer addition is associative), kept as the validation / A-B form.                              */
int ebn_embedding_grad_scatter_fixed_atomic(const int32_t* ids, const float* dX, int64_t* acc, int64_t n_tok,
                                            int32_t D, int64_t V, const ebn_step_state* st, int32_t site,
                                            float drop_p, int32_t* range_flag, ebn_stream_t stream);
int ebn_fixed_to_f32(int64_t* acc, float* out, int64_t n, int32_t* range_flag, ebn_stream_t stream);
/* ebn_fixed_to_f32 + ebn_adam_keras_step_f32 in one pass over the table: the gradient is read from the accumulator
 * (which is re-zeroed), never materialised -- the single-GPU step of a trainable table (nrms.py:129 trainable=True with
 * Keras' dense moment decay).  Same arithmetic as the two calls in sequence.                           */
int ebn_adam_keras_step_fixed_f32(float* theta, int64_t* acc, float* m, float* v, int64_t n, const ebn_step_state* st,
                                  double beta1, double beta2, double eps, float grad_scale, int32_t* range_flag,
                                  ebn_stream_t stream);

/* ---- an OPT-IN second precision of the projection matmuls (layers.py:214-226 and their weight gradient): fp32-accurate
 * GEMM on the bf16 matrix pipe.  Each fp32 operand element is split exactly into three bf16 values (8 + 8 + 8 significand
 * bits) and the product keeps the six leading cross terms, each a bf16 MFMA with fp32 accumulation; the dropped terms are
 * below 2^-23 of |a.b| per product, i.e. below the rounding of the fp32 accumulation itself.  Same argument meaning as
 * ebn_gemm_f32 (all four layouts, alpha / beta, leading dimensions); `workspace`: ebn_gemm_split_workspace_bytes(M, N, K) bytes,
 * 16-byte aligned (the bf16 planes of both operands + deterministic split-K partials).  The exact-fp32 kernels stay the default
 * everywhere; ebn_gemm_f32_prec selects by `precision` (0 = exact fp32, 1 = bf16x6 split).                               */
int64_t ebn_gemm_split_workspace_bytes(int64_t M, int64_t N, int64_t K);
int ebn_gemm_f32_split(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
                       int64_t lda, const float* B, int64_t ldb, float beta, float* C, int64_t ldc, void* workspace,
                       int64_t workspace_bytes, ebn_stream_t stream);
/* The pieces ebn_gemm_f32_split is made of, for callers that keep an operand's planes across calls or have a producer write
 * them directly.  A plane set = the three bf16 planes of one operand as [plane][K/8][rows][8] (rows padded to 256, K to 16,
 * zero-filled: ebn_planes_bytes(rows, K) bytes, 16-byte aligned) -- `rows` is the operand's non-contracted extent (M or N).
 *   ebn_split_planes_f32         src fp32 row-major [rows][K] (trans = 0) or [K][rows] (trans = 1, transposed on the way)
 *   ebn_gather_split_planes_f32  the training step's Embedding + Dropout (nrms.py:125-136) in split precision: token r's row
 *                                table[ids[r]] with the dropout stream of ebn_gather_rows_f32 (same mask bit for bit), written as
 *                                planes in BOTH orientations -- planes_n (rows = tokens, K = D: the A operand of X.Wqkv) and
 *                                planes_t (rows = D, K = tokens: the A operand of X^T.dQKV) -- instead of the fp32 X
 *   ebn_gemm_planes_f32          C[M,N] = alpha * A.B^T + beta * C from two plane sets (A: rows M, B: rows N, same K);
 *                                workspace: ebn_gemm_planes_workspace_floats(M, N, K) floats (deterministic split-K partials) */
int64_t ebn_planes_bytes(int64_t rows, int64_t K);
int ebn_split_planes_f32(const float* src, int64_t ld, int64_t rows, int64_t K, int32_t trans, void* planes,
                         ebn_stream_t stream);
int ebn_gather_split_planes_f32(const int32_t* ids, const float* table, int64_t n_rows, int32_t D, int64_t V,
                                const ebn_step_state* st, int32_t site, float drop_p, int32_t* oob_flag, void* planes_n,
                                void* planes_t, ebn_stream_t stream);
int64_t ebn_gemm_planes_workspace_floats(int64_t M, int64_t N, int64_t K);
int ebn_gemm_planes_f32(const void* a_planes, int64_t M, const void* b_planes, int64_t N, int64_t K, float alpha, float beta,
                        float* C, int64_t ldc, float* workspace, int64_t workspace_floats, ebn_stream_t stream);
int64_t ebn_gemm_prec_workspace_bytes(int64_t M, int64_t N, int64_t K, int32_t precision);
int ebn_gemm_f32_prec(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
                      int64_t lda, const float* B, int64_t ldb, float beta, float* C, int64_t ldc, void* workspace,
                      int64_t workspace_bytes, int32_t precision, ebn_stream_t stream);

/* ---- row-sharded Embedding (BASELINE.json configs[4]; no reference counterpart: nrms.py:125-134 keeps one table on
 * one device) -- device-side plan of a lookup into a table whose rows are split over `world` ranks (rank o owns the
 * block [o*per, (o+1)*per), per = ceil(V/world); or, cyclic != 0, the ids = o mod world).  Dedups the n_tok local ids
 * and lays the distinct ones out in FIXED-CAPACITY per-owner request lists, so the two exchanges of a lookup are
 * equal-split all-to-alls with host-known sizes (no host sync, hipGraph-capturable):
 *   slot_rows[o*cap + j]  owner-local row number of the j-th distinct id wanted from owner o (ascending), -1 = padding
 *                         (what ebn_gather_rows_f32 / ebn_embedding_grad_scatter_f32 take as `ids` on the owner's side)
 *   inv[t]                o*cap + j of token t: its row in the received (world*cap, D) buffer; -1 for an id outside [0,V)
 *   counts[0..world)      distinct ids wanted from each owner;  counts[world] = 1 when a list overflowed `cap` (ids were
 *                         dropped: the caller must fail the step);  counts[world+1] = 1 when an id was out of range.
 *                         The two flag words are STICKY (only ever raised here; the caller zeroes them before the first
 *                         call and whenever it has read them), so one host read per epoch sees every step's overflow
 * `workspace`: ebn_shard_plan_workspace_ints(V, world) int32 of scratch.  Integer work only, deterministic.          */
int64_t ebn_shard_plan_workspace_ints(int64_t V, int32_t world);
int ebn_shard_plan_i32(const int32_t* ids, int64_t n_tok, int64_t V, int32_t world, int32_t cyclic, int64_t cap,
                       int32_t* workspace, int32_t* slot_rows, int32_t* inv, int32_t* counts, ebn_stream_t stream);

/* ---- K.dot / Dense matmuls (layers.py:65,214,220,226; nrms_docvec.py:116,130) ----
 * C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C, exact-fp32 MFMA
 * (v_mfma_f32_32x32x2_f32). transA=0: A is [M,K] (lda>=K); transA=1: A is [K,M].
 * transB=0: B is [K,N]; transB=1: B is [N,K].                                      */
int ebn_gemm_f32(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, float alpha,
                 const float* A, int64_t lda, const float* B, int64_t ldb, float beta, float* C,
                 int64_t ldc, ebn_stream_t stream);
/* Same, with a caller-owned scratch buffer that enables deterministic split-K for skinny
 * outputs (weight gradients: M,N ~ 1e3, K = all tokens of the batch).
 * ebn_gemm_workspace_floats() returns the size the planner can use for (M,N,K).
 * NOTE: which kernel family runs DEPENDS on the scratch given: with no / too little of it a skinny product runs unsplit on
 * the block tiles (or the 32 x 32 small-output tiles), with enough of it as split-K slices or -- transposed-A, small output,
 * K >= 4096 -- as the K-chunked 16 x 16-block kernel (csrc/ebn_gemm_direct.hip), whose slices ebn_gemm_workspace_floats()
 * already covers.  Results agree to fp32 summation order; callers that need bit-identical reruns keep the size fixed.  */
int64_t ebn_gemm_workspace_floats(int64_t M, int64_t N, int64_t K);
/* The plan the launcher will use for (M,N,K) with `workspace_floats` of scratch: block tile bm x bn (128x128, 64x64 or
 * the tall 256x64 that removes column padding / last-turn idling, e.g. N = 1200) and the split-K factor.  Diagnostic:
 * lets bench.py and profiler summaries name the kernel instantiation that runs.                                    */
int ebn_gemm_plan(int64_t M, int64_t N, int64_t K, int64_t workspace_floats, int32_t* bm, int32_t* bn, int32_t* splits);
int ebn_gemm_f32_ws(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, float alpha,
                    const float* A, int64_t lda, const float* B, int64_t ldb, float beta, float* C,
                    int64_t ldc, float* workspace, int64_t workspace_floats, ebn_stream_t stream);

/* The product WITHOUT its combining pass: alpha * op(A).op(B) is left in `workspace` as *n_parts dense [M][N] slices whose sum it
 * is -- the deterministic split-K partials of a weight-gradient GEMM (K.dot backward, layers.py:65,214-226), or the product
 * itself as one slice when the plan does not split K.  ebn_grad_finish_f32 (EBN_FINISH_SPLITK job) sums them, together with the
 * other small finishing passes of the step, in one launch.  workspace: ebn_gemm_partials_workspace_floats(M, N, K) floats
 * (with fewer, down to M * N, the plan settles for fewer slices); *n_parts is a HOST out-parameter, known when the call returns. */
int64_t ebn_gemm_partials_workspace_floats(int64_t M, int64_t N, int64_t K);
int ebn_gemm_f32_partials(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A,
                          int64_t lda, const float* B, int64_t ldb, float* workspace, int64_t workspace_floats,
                          int32_t* n_parts, ebn_stream_t stream);

/* ---- the finishing passes of a training step's backward as ONE launch ----------------------------------------------------
 * Second stages of deterministic reductions their producers left out (see ebn_gemm_f32_partials, ebn_attpool_bwd_dpre_f32,
 * ebn_user_head_train_f32); same summation order as the stand-alone passes, hence the same bits.  `jobs` is a HOST array (copied
 * into the launch).  kind EBN_FINISH_SPLITK: out0[r * ld + c] = sum_z partials[z][r][c] (+ beta * out0), z < n_parts, r < rows,
 * c < cols.  EBN_FINISH_COLRED: partials [n_parts][2][cols] -> out0[c] = scale * sum_p partials[p][0][c], out1 likewise with
 * [p][1][c] (accumulated into when beta != 0).  EBN_FINISH_HEAD: partials [rows][2][cols] -> out0 = d(q), out1 = d(b) summed
 * over the rows in a fixed order, loss_out[0] = sum(loss_rows[0 .. rows)).  Jobs get consecutive block ranges in list order and
 * blocks are dispatched in order: list the latency-bound jobs (COLRED, HEAD: a few blocks, long chains) before the bulk sums.   */
#define EBN_FINISH_SPLITK 0
#define EBN_FINISH_COLRED 1
#define EBN_FINISH_HEAD 2
#define EBN_FINISH_MAX_JOBS 6
typedef struct {
  int32_t kind;
  int32_t n_parts;
  int64_t rows, cols;
  const float* partials;
  float* out0;
  float* out1;
  int64_t ld;
  float beta, scale;
  const float* loss_rows;
  float* loss_out;
} ebn_finish_job;
int ebn_grad_finish_f32(const ebn_finish_job* jobs, int32_t n_jobs, ebn_stream_t stream);

/* ONE-RANK steps: the same finishing launch with the optimizer inside (nrms.py:69-80; with world > 1 the gradient all-reduce sits
 * between the gradients and Adam: ebn_grad_finish_f32 + ebn_adam_keras_step_f32 stay).  theta / grad / m / v: the flat parameter,
 * gradient and Adam-moment buffers (identical offsets, numel floats each); every out0 / out1 of `jobs` must point into `grad`.
 * Keras-form Adam (the arithmetic of ebn_adam_keras_step_f32, element by element) is applied to each gradient element a job finishes,
 * by the thread that has just written it, and -- in blocks behind the jobs' -- to the `rest` ranges (offset, length in floats) of the
 * flat buffers: the parameters whose gradients earlier launches of the step wrote complete.  Together the jobs' outputs and the rest
 * ranges must cover every parameter exactly once (the caller's contract; nothing checks it).                                      */
#define EBN_ADAM_FLAT_MAX_REST 12
typedef struct ebn_adam_flat {
  float* theta;
  const float* grad;
  float* m;
  float* v;
  int64_t numel;
  double beta1, beta2, eps;
  float grad_scale;
  int32_t n_rest;
  int64_t rest_off[EBN_ADAM_FLAT_MAX_REST];
  int64_t rest_len[EBN_ADAM_FLAT_MAX_REST];
} ebn_adam_flat;
int ebn_grad_finish_adam_f32(const ebn_finish_job* jobs, int32_t n_jobs, const ebn_adam_flat* adam, const ebn_step_state* st,
                             ebn_stream_t stream);

/* Same again; `site` is accepted and ignored (rounds 1-4 used it to label the kernel instantiation of the encoders' Q|K|V
 * projection for per-kernel profiler summaries; the instantiations it doubled are gone).                        */
int ebn_gemm_f32_site(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, float alpha,
                      const float* A, int64_t lda, const float* B, int64_t ldb, float beta, float* C,
                      int64_t ldc, float* workspace, int64_t workspace_floats, int32_t site,
                      ebn_stream_t stream);

/* ---- a3/a6  SelfAttention core (layers.py:231-252) -------------------------------
 * qkv [n_seq*L, ld_qkv] holds Q | K | V in column blocks [0,E) [E,2E) [2E,3E), E=h*d.
 * out[n,j,a*d+c] = sum_i softmax_j'(Q_i.K_j'/sqrt(d))[i,j] * V[i,c]   (P^T V, line 249)
 * times the dropout multiplier of nrms.py:154 when drop_p > 0.
 * Supported: L <= 256, d <= 32 (MFMA kernels for L <= 32 and d in {16, 20, 32}; VALU/LDS kernels otherwise);
 * EBN_ERR_UNSUPPORTED beyond.                                                        */
int ebn_attn_fwd_f32(const float* qkv, int64_t ld_qkv, float* out, int64_t ld_out, int64_t n_seq,
                     int32_t L, int32_t h, int32_t d, const ebn_step_state* st, int32_t site,
                     float drop_p, ebn_stream_t stream);
/* dqkv (same layout as qkv) from d(out); the dropout multiplier is re-derived.     */
int ebn_attn_bwd_f32(const float* qkv, int64_t ld_qkv, const float* dout, int64_t ld_dout,
                     float* dqkv, int64_t ld_dqkv, int64_t n_seq, int32_t L, int32_t h, int32_t d,
                     const ebn_step_state* st, int32_t site, float drop_p, ebn_stream_t stream);
/* The same backward when the attention output fed an AttLayer2 pooling (layers.py:79-81): the pooling term of d(Y),
 * pool_w[n*L + l] * pool_dout[n, :], is added to `dout` on the fly, so the GEMM that produces dout = dpre.W^T needs no
 * rank-1 epilogue.  MFMA path only (ebn_attn_bwd_pooled_supported(L, d), 16-byte aligned operands, ld % 4 == 0):
 * EBN_ERR_UNSUPPORTED otherwise -- callers then use ebn_gemm_f32_rank1 + ebn_attn_bwd_f32.                            */
int ebn_attn_bwd_pooled_f32(const float* qkv, int64_t ld_qkv, const float* dout, int64_t ld_dout, const float* pool_w,
                            const float* pool_dout, int64_t ld_pool, float* dqkv, int64_t ld_dqkv, int64_t n_seq,
                            int32_t L, int32_t h, int32_t d, const ebn_step_state* st, int32_t site, float drop_p,
                            ebn_stream_t stream);
int ebn_attn_bwd_pooled_supported(int32_t L, int32_t d);

/* C[M,N] = alpha * A[M,K] * B[N,K]^T + row_scale[m] * seq_rows[m / L, n]   (C overwritten; A row-major, B stored [N,K]).
 * The d(x) of AttLayer2 in one pass: dpre.W^T (backward of K.dot(x, W), layers.py:65) plus w[n,l]*dout[n,:] (backward of
 * the weighted sum, layers.py:79-81) -- the rank-1 term is added in the GEMM epilogue instead of being written by
 * ebn_attpool_bwd_pool_f32 and read back through beta = 1.  workspace as for ebn_gemm_f32_ws(0, 1, M, N, K).       */
int ebn_gemm_f32_rank1(int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const float* B,
                       int64_t ldb, float* C, int64_t ldc, const float* row_scale, const float* seq_rows,
                       int64_t ld_seq, int32_t L, float* workspace, int64_t workspace_floats, ebn_stream_t stream);

/* Both gradient GEMMs of Y[R,N_out] = X[R,K_in] . W[K_in,N_out] (K.dot at layers.py:65,214,220,226; Dense at
 * nrms_docvec.py:116,130 -- what tf.GradientTape emits for a MatMul):
 *   dW[K_in,N_out] = X^T . dY + beta_w * dW          dX[R,K_in] = dY . W^T          (all row-major)
 * They are independent of each other; when both are small-output shapes with 16-byte-aligned operands they run as ONE
 * launch (each alone leaves half of the chip idle and costs a launch of the step's dependent chain), otherwise as two
 * ebn_gemm_f32_ws calls.  dW, dX must not alias the inputs; workspace as for the larger of the two plain calls.        */
int ebn_dense_bwd_pair_f32(int64_t R, int64_t K_in, int64_t N_out, const float* X, int64_t ldx, const float* dY,
                           int64_t lddy, const float* W, int64_t ldw, float beta_w, float* dW, int64_t lddw, float* dX,
                           int64_t lddx, float* workspace, int64_t workspace_floats, ebn_stream_t stream);

/* C[M,N] = max(A[M,K] * B[K,N] + bias[n], 0): tf.keras.layers.Dense(units, activation="relu") forward
 * (nrms_docvec.py:116-119,130; nrms.py:143-146) in one pass -- bias and ReLU ride in the GEMM epilogue (or in its
 * split-K reduce) instead of a separate element-wise launch.  workspace as for ebn_gemm_f32_ws(0, 0, M, N, K).    */
int ebn_dense_relu_fwd_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                           const float* bias, float* C, int64_t ldc, float* workspace, int64_t workspace_floats,
                           ebn_stream_t stream);

/* ---- a4/a7  AttLayer2 (layers.py:55-81) after the x.W matmul ----------------------
 * fwd: U <- tanh(U + b) in place ([R,A], R = n_seq*L); e = U.q; a = exp(e);
 *      w = a/(sum_l a + 1e-7); out[n,:] = sum_l w[n,l] X[n,l,:].                    */
int ebn_attpool_fwd_f32(float* U, const float* b, const float* q, const float* X, float* out,
                        float* w, int64_t n_seq, int32_t L, int32_t E, int32_t A,
                        ebn_stream_t stream);
/* bwd step 1: dX[n,l,:] = w[n,l]*dout[n,:]; de[n,l] = w (dw - sum w dw), dw = dout.X.
 * dX may be NULL: only de is produced (the caller adds the dX term with ebn_gemm_f32_rank1). */
int ebn_attpool_bwd_pool_f32(const float* X, const float* w, const float* dout, float* dX,
                             float* de, int64_t n_seq, int32_t L, int32_t E,
                             ebn_stream_t stream);
/* bwd step 2: dq[k] = sum_r de[r] U[r,k]; U <- dpre = de*q*(1-U^2) in place;
 * db[k] = sum_r dpre[r,k].  `partials` is scratch of ebn_attpool_partials_len(R, A)
 * floats; dq/db are ACCUMULATED into when accumulate != 0 (else overwritten).
 * dq == db == NULL: only the row-block partials are written -- their sum is left to ebn_grad_finish_f32
 * (EBN_FINISH_COLRED job: partials, n_parts = ebn_attpool_partials_len(R, A) / (2 A) row blocks, cols = A).  */
int64_t ebn_attpool_partials_len(int64_t R, int32_t A);
int ebn_attpool_bwd_dpre_f32(float* U, const float* q, const float* de, float* dq, float* db,
                             float* partials, int64_t R, int32_t A, int32_t accumulate,
                             ebn_stream_t stream);
/* ---- stage level: SelfAttention + AttLayer2 over a batch of sequences -----------------
 * The news encoder after its embedding gather (nrms.py:137-156, L = title_size,
 * Din = word_emb_dim) and the user encoder after TimeDistributed(news encoder)
 * (nrms.py:108-111, L = history_size, Din = E) are the same stage:
 *   QKV = X.Wqkv ; Y = drop(P^T V) ; U = tanh(Y.W + b) ; w = exp(U.q)/(sum+1e-7) ; out = sum w Y
 * Wqkv is [Din, 3E] = WQ | WK | WV side by side (layers.py:155-172 have no bias).
 * All activation buffers are caller-allocated and are what the backward needs.          */
typedef struct ebn_encoder_dims {
  int64_t n_seq; /* sequences in this call */
  int32_t L;     /* sequence length */
  int32_t Din;   /* input feature width */
  int32_t h, d;  /* heads, head width; E = h*d */
  int32_t A;     /* attention_hidden_dim */
  int32_t drop_site; /* EBN_SITE_* of the dropout after self-attention, or -1 */
  float drop_p;
} ebn_encoder_dims;

typedef struct ebn_encoder_params {
  const float* Wqkv; /* [Din, 3E] */
  const float* W;    /* [E, A]  */
  const float* b;    /* [A]     */
  const float* q;    /* [A]     */
} ebn_encoder_params;

typedef struct ebn_encoder_acts {
  const float* X; /* [R, Din] input rows, R = n_seq*L */
  float* QKV;     /* [R, 3E] */
  float* Y;       /* [R, E]  (after dropout) */
  float* U;       /* [R, A]  tanh output */
  float* w;       /* [R]     attention weights */
  float* out;     /* [n_seq, E] */
} ebn_encoder_acts;

typedef struct ebn_encoder_grads {
  float* dWqkv; /* [Din, 3E] */
  float* dW;    /* [E, A] */
  float* db;    /* [A] */
  float* dq;    /* [A] */
} ebn_encoder_grads;

typedef struct ebn_encoder_scratch {
  float* dY;    /* [R, E]  */
  float* dQKV;  /* [R, 3E] */
  float* de;    /* [R]     */
  float* partials;        /* ebn_attpool_partials_len(R, A) floats */
  float* gemm_ws;         /* split-K scratch, may be NULL */
  int64_t gemm_ws_floats;
} ebn_encoder_scratch;

/* scratch may be NULL; when given, its gemm_ws lets the two projection GEMMs of a SMALL batch of sequences
 * (the user encoder: B*H rows) use split-K to fill the chip (only gemm_ws / gemm_ws_floats are read).    */
int ebn_encoder_fwd_f32(const ebn_encoder_dims* dims, const ebn_encoder_params* params,
                        const ebn_encoder_acts* acts, const ebn_encoder_scratch* scratch,
                        const ebn_step_state* st, ebn_stream_t stream);
/* ebn_encoder_fwd_f32 of the news encoder in INFERENCE mode with the embedding gather fused into the projection
 * (nrms.py:125-139 without the training-only Dropout of :136): row r of the projection's A operand is table[ids[r], :],
 * fetched table -> LDS -> MFMA; X (a->X) is neither read nor written.  ids outside [0, table_rows) read row 0 and set
 * *oob_flag (may be NULL).  EBN_ERR_UNSUPPORTED (nothing launched) when the shape is outside the fused kernel's
 * (fewer than 256 token rows, unaligned operands, a table of 4 GB or more): the caller gathers and calls ebn_encoder_fwd_f32. */
int ebn_encoder_fwd_gather_f32(const ebn_encoder_dims* dims, const ebn_encoder_params* params, const ebn_encoder_acts* acts,
                               const ebn_encoder_scratch* scratch, const int32_t* ids, const float* table, int64_t table_rows,
                               int32_t* oob_flag, ebn_stream_t stream);
/* The fused product by itself: C (M, N) = table[ids[0..M), :] (K columns, leading dimension ldt) . B (K, N).  Same contract. */
int ebn_gemm_f32_rowmap(const int32_t* ids, int64_t table_rows, int64_t M, int64_t N, int64_t K, const float* table,
                        int64_t ldt, const float* B, int64_t ldb, float* C, int64_t ldc, int32_t* oob_flag, ebn_stream_t stream);

/* dout [n_seq, E] -> parameter gradients (accumulated when accumulate != 0) and, when dX is
 * non-NULL, dX [R, Din] (overwritten).  acts->U is consumed (overwritten with d(pre-tanh)). */
int ebn_encoder_bwd_f32(const ebn_encoder_dims* dims, const ebn_encoder_params* params,
                        const ebn_encoder_acts* acts, const float* dout,
                        const ebn_encoder_grads* grads, const ebn_encoder_scratch* scratch,
                        float* dX, int32_t accumulate, const ebn_step_state* st,
                        ebn_stream_t stream);

/* The user encoder and the scorer of a TRAINING step in one call: ebn_encoder_fwd_f32 on the B sequences of L news vectors
 * (nrms.py:108-111), Dot + softmax + compiled loss (nrms.py:201-202, 56-67) against cand [B*C, E] / labels [B*C], and the
 * backward of both into dcand [B*C, E], dX [B*L, Din] (the history news vectors) and the user encoder's weight gradients
 * (overwritten).  When ebn_user_head_supported(L, C, E, A) and ebn_attn_bwd_pooled_supported(L, d) hold and
 * `head_partials` (ebn_user_head_partials_len(B, A) floats) is given, everything between the two GEMM groups is
 * ebn_user_head_train_f32 -- one launch instead of six; otherwise ebn_encoder_fwd_f32 + ebn_score_loss_train_f32 +
 * ebn_encoder_bwd_f32 run as they are.  a->out receives the user vectors, duser [B, E] their gradient.              */
int ebn_user_stage_train_f32(const ebn_encoder_dims* dims, const ebn_encoder_params* p, const ebn_encoder_acts* a,
                             const float* cand, const float* labels, float* scores, float* probs, float* loss_rows,
                             float* loss_out, float* dcand, float* duser, const ebn_encoder_grads* g,
                             const ebn_encoder_scratch* s, float* head_partials, float* dX, int32_t C, int32_t loss_kind,
                             float inv_batch, const ebn_step_state* st, ebn_stream_t stream);

/* ---- a8/a9  Dot + softmax/sigmoid + loss (nrms.py:201-205, 56-67) ------------------
 * scores[b,c] = cand[b,c,:].user[b,:]; probs = softmax_c (mode 0) or sigmoid (mode 1). */
int ebn_score_fwd_f32(const float* cand, const float* user, float* scores, float* probs,
                      int64_t B, int32_t C, int32_t E, int32_t mode, ebn_stream_t stream);
/* loss_kind 0: categorical CE on the softmax logits; 1: log_loss as sigmoid CE on the same logits
 * ([KERAS-SEMANTICS] _keras_logits path); 2: log_loss as binary CE on the softmax OUTPUTS clipped to
 * [1e-7, 1-1e-7], -(y log(p^+1e-7) + (1-y) log(1-p^+1e-7)) (SURVEY.md A.5's reading of nrms.py:54,61-62; Keras
 * binary_crossentropy(from_logits=False) on a tensor without cached logits).  Writes loss_rows[b] (already divided so that
 * sum_b loss_rows = batch loss), dscores, dcand[b,c,:], duser[b,:].                   */
int ebn_score_loss_bwd_f32(const float* cand, const float* user, const float* scores,
                           const float* labels, float* loss_rows, float* dcand, float* duser,
                           int64_t B, int32_t C, int32_t E, int32_t loss_kind, float inv_batch,
                           ebn_stream_t stream);
/* Training step: ebn_score_fwd_f32 (softmax mode) + ebn_score_loss_bwd_f32 as ONE launch (same arithmetic), followed by the
 * batch loss loss_out[0] = sum(loss_rows).  nrms.py:201-202 + nrms.py:56-67 and their backward.                      */
int ebn_score_loss_train_f32(const float* cand, const float* user, const float* labels, float* scores, float* probs,
                             float* loss_rows, float* loss_out, float* dcand, float* duser, int64_t B, int32_t C,
                             int32_t E, int32_t loss_kind, float inv_batch, ebn_stream_t stream);
/* The head of a TRAINING step in one launch (+ a small fixed-order reduction): everything between the user encoder's two
 * GEMM groups, all of it local to one impression --
 *   user AttLayer2 after its x.W matmul: U <- tanh(U + b), w = exp(U.q) / (sum + 1e-7), user = sum_l w_l X_l (layers.py:65-81)
 *   -> scores = cand . user, probs = softmax (nrms.py:201-202) -> compiled loss and d(scores) (nrms.py:56-67; loss_kind as above)
 *   -> dcand = ds (x) user, duser = sum_c ds_c cand_c -> AttLayer2 backward: de = w (dw - sum w dw), dw_l = duser . X_l,
 *      U <- d(pre-tanh) = de q (1 - tanh^2), dq = sum de tanh, db = sum d(pre-tanh)
 * i.e. ebn_attpool_fwd_f32 + ebn_score_loss_train_f32 + ebn_attpool_bwd_pool_f32 + ebn_attpool_bwd_dpre_f32 of the USER
 * encoder (same formulas; six links of the step's dependent launch chain become two).  One workgroup per impression keeps
 * its L x A, L x E and C x E rows in LDS: ebn_user_head_supported(L, C, E, A) says whether they fit (E, A multiples of 4;
 * history_size 50 at E = 400, A = 200 does).  U [B*L, A], X [B*L, E], cand / dcand [B*C, E], labels / scores / probs [B*C],
 * w / de [B*L], user / duser [B, E], loss_rows [B], loss_out [1] = sum(loss_rows), dq / db [A] (overwritten),
 * partials: ebn_user_head_partials_len(B, A) floats of scratch.  16-byte aligned pointers.
 * dq == db == NULL: the fixed-order reduction is left to ebn_grad_finish_f32 (EBN_FINISH_HEAD job: partials, rows = B, cols = A,
 * loss_rows -> loss_out); loss_out is not written by this call then.                                                       */
int ebn_user_head_supported(int32_t L, int32_t C, int32_t E, int32_t A);
int64_t ebn_user_head_partials_len(int64_t B, int32_t A);
int ebn_user_head_train_f32(float* U, const float* b, const float* q, const float* X, const float* cand, const float* labels,
                            float* w, float* user, float* scores, float* probs, float* loss_rows, float* loss_out,
                            float* dcand, float* duser, float* de, float* dq, float* db, float* partials, int64_t B,
                            int32_t L, int32_t C, int32_t E, int32_t A, int32_t loss_kind, float inv_batch,
                            ebn_stream_t stream);
/* Streaming AUC of compile(metrics=["AUC"]) (ebnerd_nrms.py:244-248; tf.keras.metrics.AUC defaults): every (label,
 * prediction) pair of a batch goes into pos_hist / neg_hist [n_thresholds + 1] at bucket = number of thresholds strictly
 * below the prediction (`thresholds`: ascending float64 on the device).  Integer atomics: order-independent.           */
int ebn_auc_hist_f32(const float* probs, const float* labels, int64_t n, const double* thresholds, int32_t n_thresholds,
                     int64_t* pos_hist, int64_t* neg_hist, ebn_stream_t stream);
/* ragged scoring for the eval path (dataloader.py:94-107 + nrms.py:204-205):
 * out[p] = act(user[u_idx[p],:] . news[n_idx[p],:]), act = sigmoid (mode 1) or id (0). */
int ebn_pair_score_f32(const float* user, const float* news, const int32_t* u_idx,
                       const int32_t* n_idx, float* out, int64_t n_pairs, int32_t E, int32_t mode,
                       ebn_stream_t stream);

/* ---- a10  tf.keras.optimizers.Adam (nrms.py:69-80), Keras update form -------------
 * m += (g-m)(1-b1); v += (g^2-v)(1-b2); theta -= alpha_t * m/(sqrt(v)+eps), dense over
 * n elements; g is multiplied by grad_scale first (1/world_size after an all-reduce sum).
 * Betas are doubles so that (1-beta) is formed as Keras forms it (Python floats), then cast. */
int ebn_adam_keras_step_f32(float* theta, const float* g, float* m, float* v, int64_t n,
                            const ebn_step_state* st, double beta1, double beta2, double eps,
                            float grad_scale, ebn_stream_t stream);

/* ---- small dense helpers used by the DocVec encoder (nrms_docvec.py:113-135) ------- */
/* Y = relu(X + bias) row-wise, in place allowed (Dense(relu)).                        */
int ebn_bias_relu_f32(const float* X, const float* bias, float* Y, int64_t R, int32_t Ccols,
                      ebn_stream_t stream);
/* dX = dY * (Y > 0); dbias[c] = sum_r dX[r,c] (partials scratch: ebn_colsum_partials_len). */
int64_t ebn_colsum_partials_len(int64_t R, int32_t Ccols);
int ebn_bias_relu_bwd_f32(const float* Y, const float* dY, float* dX, float* dbias,
                          float* partials, int64_t R, int32_t Ccols, int32_t accumulate,
                          ebn_stream_t stream);
/* BatchNormalization(axis=-1, momentum .99, eps 1e-3) over the R rows of one call site.
 * training != 0: batch statistics (biased var), saved to mean_out/istd_out, moving stats
 * updated in place; else moving statistics.  Optional fused dropout (site, drop_p) with
 * flat element offset elem_offset.                                                     */
int ebn_batchnorm_fwd_f32(const float* X, const float* gamma, const float* beta,
                          float* moving_mean, float* moving_var, float* Y, float* xhat,
                          float* mean_out, float* istd_out, float* partials, int64_t R,
                          int32_t Ccols, int32_t training, const ebn_step_state* st,
                          int32_t site, float drop_p, int64_t elem_offset,
                          ebn_stream_t stream);
int ebn_batchnorm_bwd_f32(const float* dY, const float* xhat, const float* gamma,
                          const float* istd, float* dX, float* dgamma, float* dbeta,
                          float* partials, int64_t R, int32_t Ccols, int32_t training,
                          int32_t accumulate, const ebn_step_state* st, int32_t site,
                          float drop_p, int64_t elem_offset, ebn_stream_t stream);
/* The two TimeDistributed call sites of a TRAINING step in one launch each way (rows [0,R0) = history block, [R0,R0+R1)
 * = candidate block of one row block; R0 + R1 <= 1024, else EBN_ERR_UNSUPPORTED and the caller uses the per-site entry
 * points): forward = two ebn_batchnorm_fwd_f32 (own batch statistics, two moving-average updates, history first; the
 * dropout stream is indexed by the element's position in the whole block); backward = two ebn_batchnorm_bwd_f32 with
 * dgamma/dbeta summed over the sites, followed by ebn_bias_relu_bwd_f32 of the Dense(relu) in front (relu_out = that
 * Dense's output): dX = d(pre-activation), dbias = its column sums.  nrms_docvec.py:116-124, nrms.py:143-152.          */
int ebn_batchnorm2_fwd_f32(const float* X, const float* gamma, const float* beta, float* moving_mean, float* moving_var,
                           float* Y, float* xhat, float* mean_out0, float* istd_out0, float* mean_out1, float* istd_out1,
                           int64_t R0, int64_t R1, int32_t Ccols, const ebn_step_state* st, int32_t site, float drop_p,
                           ebn_stream_t stream);
int ebn_batchnorm2_relu_bwd_f32(const float* dY, const float* xhat, const float* relu_out, const float* gamma,
                                const float* istd0, const float* istd1, float* dX, float* dgamma, float* dbeta,
                                float* dbias, int64_t R0, int64_t R1, int32_t Ccols, const ebn_step_state* st, int32_t site,
                                float drop_p, ebn_stream_t stream);

/* ---- a11 as ONE launch per Dense layer and direction (nrms_docvec.py:113-135, training step) ------------------------
 * x -> [Dense(u_l, relu, l2) -> BatchNormalization -> Dropout] x n_layers -> Dense(e_out, relu) over the rows of the two
 * TimeDistributed call sites (rows [0,n0) = history block, [n0,n0+n1) = candidate block: own batch statistics and one
 * moving-average update each, history first; nrms_docvec.py:88-90,176-178).  BatchNormalization, Dropout and the ReLU
 * backward never run as passes of their own: the matmul in front leaves per-tile column partials, the matmul behind
 * combines them and transforms its A operand on the way into LDS (csrc/ebn_docvec.hip).  Same arithmetic as
 * ebn_dense_relu_fwd_f32 + ebn_batchnorm2_fwd_f32 / ebn_batchnorm2_relu_bwd_f32 + ebn_bias_relu_bwd_f32 up to fp32
 * rounding (the column sums are order-independent 64-bit fixed-point accumulations of per-tile sums, the batch variance
 * is E[x^2] - mean^2 in float64 from them, not two-pass over the block): bitwise reproducible run to run.
 *
 * All pointers device, 16-byte aligned, matrices dense row-major.  W[l] is (d_l, u_l) with d_0 = din, W[n_layers] the
 * output kernel (u_last, e_out); b[l] likewise.  Forward writes R[l] = relu(Dense_l), Xn[l] = Dropout(BN(R[l])), NE, the
 * batch statistics (inside `stat`) and the moving statistics.  Backward reads dNE and writes dY[l] = d(Xn[l]) with the
 * dropout mask applied, dP[l] = d(pre-activation of Dense l) for l = 0..n_layers (the B operands of the weight gradients
 * W[l]' = Xn[l-1]^T . dP[l], bias gradient = column sums of dP[l]: ebn_gemm_tn_group_f32), ggamma / gbeta, and adds
 * l2 * sum_l sum(W[l]^2), l < n_layers, to loss[0] (loss may be NULL; the 2*l2*W term of the kernel gradients is the
 * l2_W option of ebn_gemm_tn_group_f32).
 * ebn_dvn_supported: 1 <= n_layers <= 4, widths multiples of 4, hidden widths <= 1024.
 * `stat`: ebn_dvn_stat_floats(args) floats, 16-byte aligned, ZERO before the first forward call, then owned by the two
 * calls: it must survive from the forward to the backward call of a step, and every forward call must be followed by the
 * backward call of the same step before the next forward (the accumulators inside are re-zeroed by the step's own
 * launches: the backward ones by the first forward launch, the forward ones by the last backward launch).  After a step
 * that ran only half way the caller zeroes `stat` again.                                                               */
#define EBN_DVN_MAX_LAYERS 4
typedef struct ebn_dvn_args {
  int32_t n_layers, din, e_out;
  int32_t units[EBN_DVN_MAX_LAYERS];
  int32_t n0, n1;
  float drop_p, l2;
  const float* W[EBN_DVN_MAX_LAYERS + 1];
  const float* b[EBN_DVN_MAX_LAYERS + 1];
  const float* gamma[EBN_DVN_MAX_LAYERS];
  const float* beta[EBN_DVN_MAX_LAYERS];
  float* moving_mean[EBN_DVN_MAX_LAYERS];
  float* moving_var[EBN_DVN_MAX_LAYERS];
  const float* X0;
  float* R[EBN_DVN_MAX_LAYERS];
  float* Xn[EBN_DVN_MAX_LAYERS];
  float* NE;
  float* stat;
  const float* dNE;
  float* dY[EBN_DVN_MAX_LAYERS];
  float* dP[EBN_DVN_MAX_LAYERS + 1];
  float* ggamma[EBN_DVN_MAX_LAYERS];
  float* gbeta[EBN_DVN_MAX_LAYERS];
  float* loss;
  int32_t* range_flag; /* may be NULL.  Set to 1 (never cleared) when a tile's column sum is not finite or leaves the range of the
                          fixed-point accumulators: the statistics of that step are invalid, the run has diverged               */
} ebn_dvn_args;
int ebn_dvn_supported(const ebn_dvn_args* args);
int64_t ebn_dvn_stat_floats(const ebn_dvn_args* args);
int ebn_dvn_fwd_train_f32(const ebn_dvn_args* args, const ebn_step_state* st, ebn_stream_t stream);
int ebn_dvn_bwd_f32(const ebn_dvn_args* args, const ebn_step_state* st, ebn_stream_t stream);

/* The prologue of an NRMSDocVec training step on a device-resident batch of article-row numbers (dataloader.py:169-179,
 * lookup_article_matrix[rows]) as ONE launch: ebn_step_advance (st may be NULL: no advance) + the label copy (n_labels floats) +
 * X0[r,:] = matrix[idx[r],:] for the n0 + n1 rows of the (up to two) index segments idx0 | idx1 -- what ebn_copy3_advance +
 * ebn_gather_rows_f32 do in two.  Rows outside [0, n_rows) write zeros and set *oob_flag (may be NULL).  din % 4 == 0.    */
int ebn_docvec_stage_gather_f32(const int32_t* idx0, int64_t n0, const int32_t* idx1, int64_t n1, const float* labels_src,
                                float* labels_dst, int64_t n_labels, const float* matrix, int64_t n_rows, int32_t din, float* X0,
                                int32_t* oob_flag, ebn_step_state* st, double beta1, double beta2, ebn_stream_t stream);

/* Up to EBN_TN_GROUP_MAX independent weight-gradient products C_i (M_i, N_i) = A_i^T . B_i (A_i (K_i, M_i), B_i (K_i, N_i),
 * all row-major) in ONE launch of 32x32 small-output tiles -- the Dense kernel gradients of a step (nrms_docvec.py:116-134
 * backward; the user encoder's K.dot gradients of layers.py:65,214-226), each too small to fill the chip alone.  Options per
 * problem: colsum (N_i floats) = column sums of B_i over K_i (the bias gradient of the same Dense layer, taken from the B
 * tiles as they pass); l2_W (M_i, N_i, leading dimension ldc) adds two_lambda * l2_W to C_i (kernel_regularizer=l2).
 * 16-byte aligned operands, extents and leading dimensions multiples of 4; EBN_ERR_UNSUPPORTED otherwise.              */
#define EBN_TN_GROUP_MAX 8
typedef struct ebn_tn_problem {
  int64_t M, N, K;
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  float* C;
  int64_t ldc;
  float* colsum;
  const float* l2_W;
  float two_lambda;
} ebn_tn_problem;
int ebn_gemm_tn_group_f32(const ebn_tn_problem* problems, int32_t n, ebn_stream_t stream);

/* The closing launch of a ONE-RANK NRMSDocVec training step (nrms_docvec.py:99-188 backward + nrms.py:69-80): the Dense weight
 * gradients of ebn_gemm_tn_group_f32 (`problems`: every C_i / colsum_i must lie inside the flat gradient buffer `grad`) and, dealt over
 * the same workgroups instead of launches of their own, (1) Keras-form Adam -- in the epilogue of the tile that has just produced a
 * gradient element, and element-wise over the `rest` ranges (offset, length in floats) of the flat buffers, i.e. every parameter no
 * tile owns; theta / grad / m / v are the flat parameter, gradient and moment buffers with identical offsets, `numel` floats each --
 * (2) the user head's finishing sums (ebn_user_head_train_f32 called with dq == db == NULL leaves `head_partials`, `loss_rows`):
 * d(q), d(b) over the B impressions and loss_out[0] = sum(loss_rows) + l2 * sum_l sum(W[l]^2) (ebn_dvn_bwd_f32 called with
 * args->loss == NULL leaves the L2 term to this call), then Adam on d(q) / d(b).  Same arithmetic, element by element, as
 * ebn_gemm_tn_group_f32 + ebn_user_head_train_f32's finishing pass + ebn_adam_keras_step_f32.  With world > 1 the gradient all-reduce
 * sits between the gradients and Adam: the separate calls stay.                                                                       */
#define EBN_DVN_FINALE_MAX_REST 12
typedef struct ebn_dvn_finale {
  float* theta;
  const float* grad;
  float* m;
  float* v;
  int64_t numel;
  double beta1, beta2, eps;
  float grad_scale;
  int32_t n_rest;
  int64_t rest_off[EBN_DVN_FINALE_MAX_REST];
  int64_t rest_len[EBN_DVN_FINALE_MAX_REST];
  const float* head_partials;
  int64_t B;
  int32_t A;
  float* dq;
  float* db;
  const float* loss_rows;
  float* loss_out;
} ebn_dvn_finale;
int ebn_dvn_finale_f32(const ebn_dvn_args* args, const ebn_tn_problem* problems, int32_t n, const ebn_dvn_finale* fin,
                       const ebn_step_state* st, ebn_stream_t stream);

/* Step prologue: copy up to three device buffers (history ids, candidate ids, labels of a batch handed over as device
 * tensors -- the inputs of nrms.py:170-176) into the step's static buffers with ONE launch; n_i in bytes, multiples
 * of 4; a NULL source or n_i = 0 skips that pair.                                                                   */
int ebn_copy3(const void* s0, void* d0, int64_t n0, const void* s1, void* d1, int64_t n1, const void* s2, void* d2,
              int64_t n2, ebn_stream_t stream);
/* The same copy with ebn_step_advance folded in (the staging launch of a training step on a device-resident batch). */
int ebn_copy3_advance(const void* s0, void* d0, int64_t n0, const void* s1, void* d1, int64_t n1, const void* s2,
                      void* d2, int64_t n2, ebn_step_state* st, double beta1, double beta2, ebn_stream_t stream);

/* y = a*x + y over n elements (L2 kernel-regulariser gradient, gradient accumulation). */
int ebn_axpy_f32(float a, const float* x, float* y, int64_t n, ebn_stream_t stream);
/* kernel_regularizer=l2(lambda) of one Dense kernel in a single pass (nrms_docvec.py:119-121, nrms.py:146-148):
 * gW += 2*lambda*W and loss[0] += lambda*sum(W^2); `partials` = scratch of at least 256 floats
 * (ebn_colsum_partials_len never returns less).                                                            */
int ebn_l2_reg_f32(const float* W, float* gW, int64_t n, float lambda, float* partials, float* loss,
                   ebn_stream_t stream);
/* The same for up to four Dense kernels with two launches in total (NULL W_i / n_i = 0 skips a slot): gW_i += 2*lambda*W_i,
 * loss[0] += lambda * sum_i sum(W_i^2); `partials` = scratch of at least 1024 floats.                                 */
int ebn_l2_reg4_f32(const float* W0, float* g0, int64_t n0, const float* W1, float* g1, int64_t n1, const float* W2,
                    float* g2, int64_t n2, const float* W3, float* g3, int64_t n3, float lambda, float* partials,
                    float* loss, ebn_stream_t stream);
/* out[0] (+)= scale * sum(x[0..n)) -- deterministic single-block reduction.            */
int ebn_sum_f32(const float* x, int64_t n, float scale, float* out, int32_t accumulate,
                ebn_stream_t stream);
/* out[0] (+)= scale * sum(x^2).                                                         */
int ebn_sumsq_f32(const float* x, int64_t n, float scale, float* out, int32_t accumulate,
                  ebn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EBNERD_HIP_H */
