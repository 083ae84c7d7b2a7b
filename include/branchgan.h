/* branchgan.h — C-ABI of libbranchgan_hip.so: the MI355X (gfx950) kernels behind Branch-GAN's
 * branch-sequence hot path.
 *
 * The reference (FreddeFrallan/Branch-GAN) is pure Python on PyTorch and has no FFI layer; the entry
 * points below are what a maintainer would bind (ctypes) to replace the chains of
 * ATen ops cited beside each one.  Conventions:
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - strides are in ELEMENTS of the tensor's dtype; the innermost (feature / vocabulary) dimension is
 *     contiguous;
 *   - dtype: BG_F32 (0) or BG_BF16 (1);  ids are int64 (the reference's type), lengths int32;
 *   - `stream` is a hipStream_t (NULL = default stream); nothing here synchronises or allocates: the
 *     caller owns every buffer, including the workspace whose size the *_workspace_bytes calls return;
 *   - return value: 0 = ok, otherwise an error code; bg_last_error() gives the message (thread-local).
 *     Shapes are validated on the host BEFORE any launch (a bad shape never reaches a kernel).
 */
#ifndef BRANCHGAN_H
#define BRANCHGAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BG_F32 0
#define BG_BF16 1

#define BG_OK 0
#define BG_ERR_ARG 1      /* invalid argument / unsupported shape */
#define BG_ERR_LAUNCH 2   /* HIP reported a launch error */
#define BG_ERR_WORKSPACE 3

#define BG_MAX_TOPN 64      /* samples per row */
#define BG_MAX_BLOCKED 8    /* blocked ids per row */
#define BG_MAX_SLOTS 64     /* branch KV slots (tree nodes incl. self) one call can see */
#define BG_MAX_RANKS 8      /* rank thresholds per bg_rank_stats call */

const char* bg_last_error(void);
int bg_version(void);
/* sha256 (first 32 hex digits) over the sources this library was built from (every .hip and .h under csrc, this
 * header, the Makefile): the host side compares it with the sources beside it and refuses a stale build. */
const char* bg_build_id(void);
/* gfx arch string of the current device ("gfx950"), into buf; returns BG_OK or BG_ERR_LAUNCH. */
int bg_device_arch(char* buf, size_t buf_len);

/* ------------------------------------------------------------------------------------------------
 * Sampling over the vocabulary.
 * Replaces BranchPredictions/SamplingStrategies.py:197-205 (_topKSampling: full argsort, slice [:n]),
 * :149-158 (_sampleWithoutReplacement: probs / noise, full argsort), :236-240 (inverse top-k) and the
 * label blocking scatter+cat of :258-265, all in ONE pass over the row.
 *
 *   key[r][j] = (j in blocked[r] ? blocked_value : scores[r][j]) / noise[r][j]     (noise may be NULL)
 *   out_ids[r][0..n) = the n largest keys, descending; equal keys in increasing j; NaN keys first
 *   out_keys (optional) = those keys.
 * The division is IEEE-correct fp32 (same bits as torch's probs / noise).
 * row_map (optional, int64 (rows)): output row r reads input row row_map[r] of scores (noise is indexed
 * by r): lets the caller sample only the K branch rows of a (B*S, V) tensor.
 * workspace: bg_topn_workspace_bytes(rows, V, n) bytes (may be 0 -> workspace may be NULL).
 */
size_t bg_topn_workspace_bytes(int64_t rows, int64_t V, int n);
int bg_topn_select(const void* scores, int dtype, int64_t rows, int64_t V, int64_t score_row_stride,
                   const int64_t* row_map,
                   const float* noise, int64_t noise_row_stride,
                   const int64_t* blocked, int n_blocked, float blocked_value,
                   int n, int64_t* out_ids, float* out_keys,
                   void* workspace, size_t workspace_bytes, void* stream);

/* The same selection for callers that cannot allocate device memory or bring a workspace (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  It has NO counterpart in the reference: it exists for those
 * callers only.  The per-(row, split) partial lists live in a file-scope device array of the library,
 * BG_TOPN_STATIC_ITEMS x 8 bytes = 1 MiB that every GPU the library runs on holds until the process ends; the call makes no
 * hipMalloc / hipFree and calls no allocating routine.  Arguments and results are those of bg_topn_select (bit-identical
 * ids and keys).  Limit: when rows < 2048 the row is split over `splits` = min(ceil(2048 / rows), V / 4096, 16)
 * workgroups and rows x splits x n must be <= BG_TOPN_STATIC_ITEMS (always true for n <= 32), otherwise
 * BG_ERR_WORKSPACE and a message with the numbers; rows >= 2048 needs no partial lists.  All calls on one GPU share that
 * GPU's array: calls that overlap on one GPU (two streams, two threads) race - serialise them.  The first call on each
 * device looks the array's address up (hipGetSymbolAddress); make that first call outside a graph capture.
 */
#define BG_TOPN_STATIC_ITEMS 131072
int bg_topn_select_static(const void* scores, int dtype, int64_t rows, int64_t V, int64_t score_row_stride,
                          const int64_t* row_map,
                          const float* noise, int64_t noise_row_stride,
                          const int64_t* blocked, int n_blocked, float blocked_value,
                          int n, int64_t* out_ids, float* out_keys, void* stream);

/* Top-p filter + exponential race (SamplingStrategies.py:161-187, top_p_sampling):
 * sort descending, cumulative sum, keep sorted position i iff cum[i+1] <= p (the reference's shift),
 * renormalise by the kept mass, keys = kept / noise, n largest.  A row that keeps nothing yields NaN
 * keys and therefore ids 0..n-1 (this library's tie rule).  probs fp32 only.
 */
size_t bg_topp_workspace_bytes(int64_t rows, int64_t V, int n);
int bg_topp_select(const float* probs, int64_t rows, int64_t V, int64_t row_stride,
                   const float* noise, int64_t noise_row_stride,
                   const int64_t* blocked, int n_blocked,
                   float top_p, int n, int64_t* out_ids, float* out_keys,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused softmax statistics over the vocabulary (never materialises probs).
 * Replaces Models/Generator.py:47 (softmax over V), BranchPredictions/MultiStepPrediction.py:238-240,
 * 257-261 (gathers of label / sampled-id probs) and Models/Losses/GeneratorLosses.py:133-137 (CE).
 *   lse[r]            = log sum_j exp(logits[r][j])
 *   out_logp[r][i]    = logits[r][ids[r][i]] - lse[r]         (ids[r][i] < 0 -> 0, no gradient)
 * backward:  dlogits[r][j] = softmax[r][j] * (g_lse[r] - sum_i g_logp[r][i]) + sum_i g_logp[r][i]*[j == ids[r][i]]
 * dlogits has the dtype and row stride of logits.
 */
int bg_softmax_stats_fwd(const void* logits, int dtype, int64_t rows, int64_t V, int64_t row_stride,
                         const int64_t* ids, int n_ids,
                         float* out_lse, float* out_logp, void* stream);
int bg_softmax_stats_bwd(const void* logits, int dtype, int64_t rows, int64_t V, int64_t row_stride,
                         const int64_t* ids, int n_ids, const float* lse,
                         const float* g_lse, const float* g_logp,
                         void* dlogits, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-row rank statistics over the vocabulary: the (rows, V) work of the evaluation metrics in one kernel, no sort.
 * Replaces Training/Metrics/GeneratorMetrics.py:24-26 (_calcRankProbMasses: full sort + prefix sums), :29-50
 * (_calculateInputIDsPlacementMetric: full argsort + (B,S,V) one-hot), :64-66 / :109-111 (label gather, row min / max)
 * and Training/Utils.py:20-21 (entropy).  x fp32 [rows][V] (row stride in elements), any values (NaNs order above +inf).
 *   label_value[r] = x[r][labels[r]]            (labels[r] outside [0, V): NaN, placement -1; labels may be NULL)
 *   placement[r]   = position of the label in a STABLE descending sort of the row
 *                  = #{j : x[r][j] > x[r][l]} + #{j < l : x[r][j] == x[r][l]}   (torch.argsort is not stable: on tied
 *                    values the reference's position is any of the tied ones)
 *   row_min, row_max;  entropy[r] = -sum_j x * log(x + entropy_eps)
 *   top_mass[r][i] = sum of the min(ranks[i], V) largest values of the row, exact (an MSD radix select finds the
 *                    ranks[i]-th largest value; ties at it are counted as often as they fit)
 * ranks: HOST array of n_ranks <= BG_MAX_RANKS values >= 1.  Every output pointer is optional (NULL: not written),
 * top_mass is required when n_ranks > 0.  Deterministic (integer histograms, fixed-order sums).
 */
int bg_rank_stats(const float* x, int64_t rows, int64_t V, int64_t row_stride, const int64_t* labels,
                  const int32_t* ranks, int n_ranks, float entropy_eps,
                  float* label_value, int64_t* placement, float* row_min, float* row_max, float* entropy,
                  float* top_mass, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Branch ("non-residual") attention, forward and backward, with the mask computed arithmetically.
 * Replaces Models/NeoXBranchAttention/NeoXNonResidualAttention.py:183-232 (_nonResidualTextAttention),
 * Models/GPT2BranchAttention/NonResidualAttention.py:119-172 (nonResidualTextAttention), the mask tensors
 * of BranchPredictions/MultiStepPrediction.py:92-127 and the KV concatenations of :129-152; with
 * n_slots == 0 and ctx_len[i] = i + 1 it is also the root causal attention (NeoX :234-290, GPT-2 :78-117).
 *
 * Per batch b and head h there are NQ query nodes x K branch points.  Query (n, k) attends to
 *   text keys   j < ctx_len[k]                         (ctx_len[k] = sparseIndex[k] + 1)
 *   branch keys (slot s, k) for s in path[n][0 .. path_len[n])   (its ancestors and itself, same k only)
 * softmax over that set with scores scale * <q, key>; out = sum w * value; lse = log-sum-exp (fp32).
 *
 * Layouts (element strides; feature dim Dh contiguous):
 *   q                : [b][h][n][k][Dh]  strides (q_sb, q_sh, q_sn, q_sk)
 *   out, dout        : same index space, strides (o_sb, o_sh, o_sn, o_sk);  dq: (dq_*) or, when those are 0, (o_*)
 *   k_text, v_text   : [b][h][s][Dh]     strides (t_sb, t_sh, t_ss)         S_text rows
 *   slots            : host arrays of n_slots device pointers, each [b][h][k][Dh] strides (s_sb, s_sh, s_sk)
 *   ctx_len          : int32 [K];  path: int32 [NQ][max_path] slot indices;  path_len: int32 [NQ]
 *   lse, delta       : fp32 [b][h][n][k] contiguous
 * Dh in {32, 64, 128, 256}; n_slots <= BG_MAX_SLOTS.
 * Backward writes dq, dk_text/dv_text (every text row, zeros where nothing attends) and each slot's
 * dk/dv; gradients are plain stores (deterministic, no atomics), fp32 or bf16 like the inputs.
 */
typedef struct {
  int dtype;
  int B, H, NQ, K, Dh, S_text;
  int n_slots, max_path;
  float scale;
  int64_t q_sb, q_sh, q_sn, q_sk;
  int64_t o_sb, o_sh, o_sn, o_sk;     /* out / dout (and dq unless dq_* are set) */
  int64_t t_sb, t_sh, t_ss;           /* k_text, v_text (and dk_text, dv_text) */
  int64_t s_sb, s_sh, s_sk;           /* every slot tensor (and its gradient) */
  int64_t dq_sb, dq_sh, dq_sn, dq_sk; /* backward only: strides of dq; all zero = the o_* strides.  Lets dq, dk, dv land in
                                         one packed dqkv buffer (the layout q, k, v were sliced from) with no gather */
  /* Dropout on the attention probabilities (GPT-2: NonResidualAttention.py:107-109,157-159, always on; NeoX:
   * NeoXNonResidualAttention.py:224,287): the weights are normalised over all visible keys, then each is kept with probability
   * 1 - p_drop and scaled by 1 / (1 - p_drop); lse is that of the undropped softmax.  p_drop = 0 (the zero-initialised
   * struct): no dropout, the fields below are ignored.  The mask is never stored: it is a counter-based function of
   * (drop_state[0] = seed, drop_state[1] = step counter, drop_call, batch*head, query row n*K + k, key column), regenerated by
   * the backward call that passes the same p_drop / drop_call / drop_state contents (bg_attn_dropout_mask writes it out).
   * drop_state is DEVICE memory read when the kernel runs: advance the counter (bg_dropout_advance) between the replays of
   * a captured graph to draw fresh masks; drop_call tells the attention calls of one step apart.  Dh <= 128 only.
   * drop_row0 is added to the query row before it enters the function: two calls that cover the same logical rows at different
   * positions (a branch step over ONE tree node, later a tree pass over all nodes) draw the same mask for them when they pass the
   * same drop_call and the row offsets that line the rows up. */
  float p_drop;
  uint32_t drop_call;
  const uint64_t* drop_state;
  uint32_t drop_row0;
  uint32_t drop_reserved;             /* 0 */
} bg_attn_shape;

int bg_branch_attn_fwd(const bg_attn_shape* shape,
                       const void* q, const void* k_text, const void* v_text,
                       const void* const* k_slots, const void* const* v_slots,   /* host arrays */
                       const int32_t* ctx_len, const int32_t* path, const int32_t* path_len,
                       void* out, float* lse, void* stream);

/* The same call with a caller workspace (same reference lines as bg_branch_attn_fwd).  With few query rows per (batch, head)
 * and a long key range - a generator branch step at seq 512: 64 queries against 512 keys - a launch has fewer workgroups than
 * the chip has CUs and each pulls the whole K / V of its head through one CU.  Given `workspace` (16-byte aligned, at least
 * bg_branch_attn_fwd_workspace_bytes(shape) bytes; that is 0 when the shape is not split) the key range of a query tile is
 * dealt to several workgroups whose partial (max, sum, output) states meet in the workspace and are combined by a second
 * kernel on the same stream.  Same results up to fp32 summation order; NULL / too small a workspace = bg_branch_attn_fwd. */
size_t bg_branch_attn_fwd_workspace_bytes(const bg_attn_shape* shape);
int bg_branch_attn_fwd_ws(const bg_attn_shape* shape,
                          const void* q, const void* k_text, const void* v_text,
                          const void* const* k_slots, const void* const* v_slots,   /* host arrays */
                          const int32_t* ctx_len, const int32_t* path, const int32_t* path_len,
                          void* out, float* lse, void* workspace, size_t workspace_bytes, void* stream);

int bg_branch_attn_bwd(const bg_attn_shape* shape,
                       const void* q, const void* k_text, const void* v_text,
                       const void* const* k_slots, const void* const* v_slots,   /* host arrays */
                       const int32_t* ctx_len, const int32_t* path, const int32_t* path_len,
                       const void* out, const float* lse, const void* dout,
                       void* dq, void* dk_text, void* dv_text,
                       const void* dk_text_in, const void* dv_text_in,           /* optional (NULL): gradients of the text keys /
                                                                                    values from their other consumers, same layout
                                                                                    as dk_text; dk_text = dK + dk_text_in (added in
                                                                                    fp32, rounded once); may alias dk_text / dv_text */
                       void* const* dk_slots, void* const* dv_slots,             /* host arrays */
                       float* delta /* fp32 scratch [b][h][n][k] */, void* stream);

/* The softmax weights on request (the reference returns them from _nonResidualTextAttention and its modules hand them out
 * under output_attentions=True: NeoXNonResidualAttention.py:224-232, NonResidualAttention.py:160-172).  The fused kernels never
 * form them; this rebuilds them from q, the keys and the forward's lse:
 *   weights fp32 [b][h][n][k][W], W >= S_text + (max_path - 1) * K + 1, in the reference's column order
 *   [text 0..S_text-1 | path entry 0: K columns, only column k set | ... | last column: the node's last path entry (itself)];
 *   invisible keys get weight 0.  A diagnostic path (one workgroup per query row), not a tuned kernel.
 */
int bg_branch_attn_weights(const bg_attn_shape* shape, const void* q, const void* k_text, const void* const* k_slots /* host array */,
                           const int32_t* ctx_len, const int32_t* path, const int32_t* path_len, const float* lse,
                           float* weights, int W, void* stream);

/* The keep mask of an attention call with shape->p_drop > 0, written out (tests, debugging; the attention kernels never store
 * it): keep uint8 [b][h][n*K + k][S_text + max_path], 1 = kept; column S_text + p belongs to entry p of the node's path.
 * Every position gets its decision, visible or not.  No counterpart in the reference (torch draws its mask from the global
 * generator inside F.dropout); the oracle applies this mask to its own softmax weights to check the kernels. */
int bg_attn_dropout_mask(const bg_attn_shape* shape, uint8_t* keep, void* stream);

/* drop_state[1] += 1 on the stream (one thread): the step counter of the dropout masks, see bg_attn_shape. */
int bg_dropout_advance(uint64_t* drop_state, void* stream);

/* Dropout over the rows of an activation, out = in * keep / (1 - p), with the mask of the attention dropout's generator instead
 * of torch's (GPT-2: F.dropout on the attention block's output NonResidualAttention.py:274-275 and on the embeddings
 * NonResidualDiscriminatorBase.py:231-232, always on; nn.Dropout of GPT2MLP / NeoXNonResidualBase.py:79-80,110-125 in training
 * mode): the keep decision of element (b, t, c) of a [batch][rows][cols] activation is a function of (drop_state, site, b,
 * logical row, c), logical row = t for t < text_rows, else row0 + (t - text_rows) - so a no-grad branch step over one tree node
 * and the later differentiated pass over all nodes draw the same mask for the same token (what lets the sampling phase run
 * without autograd under dropout), and the backward pass is the same call on the gradient.  in == out is allowed.
 * dtype BG_F32 / BG_BF16; row_stride in elements (cols contiguous); batch < 4096, logical rows < 2^20. */
int bg_dropout_rows(const void* in, void* out, int dtype, int batch, int rows, int cols, int64_t in_row_stride, int64_t out_row_stride,
                    int text_rows, uint32_t row0, float p, uint32_t site, const uint64_t* drop_state, void* stream);

/* ------------------------------------------------------------------------------------------------
 * NeoX prologue: split the fused QKV projection, move heads forward and apply partial rotary.
 * Replaces NeoXNonResidualAttention.py:94-115 (view/permute/slices) and :425-432 (apply_rotary_pos_emb
 * gathers).  qkv [B][T][H][3*Dh] contiguous -> q, k, v [B][H][T][Dh] contiguous; the first rot dims of
 * q and k are rotated with cos/sin tables fp32 [max_pos][rot] at position_ids int32 [B][T].
 * backward: dqkv from (dq, dk, dv).
 */
int bg_neox_qkv_rope_fwd(const void* qkv, int dtype, int B, int T, int H, int Dh, int rot,
                         const float* cos_tab, const float* sin_tab, int max_pos, const int32_t* position_ids,
                         void* q, void* k, void* v, void* stream);
int bg_neox_qkv_rope_bwd(const void* dq, const void* dk, const void* dv, int dtype, int B, int T, int H, int Dh, int rot,
                         const float* cos_tab, const float* sin_tab, int max_pos, const int32_t* position_ids,
                         void* dqkv, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row-wise glue of the transformer block, fused around the GEMMs (residual stream and statistics fp32,
 * activations fp32 or bf16).  Replaces, per block and per tree node, the ATen chains of
 * Models/GPT2BranchAttention/NonResidualDiscriminatorBase.py:46-90 (ln_1, residual add, ln_2) and
 * Models/NeoXBranchAttention/NeoXNonResidualBase.py:108-133, and in backward the separate cast, residual-add,
 * LayerNorm dgamma/dbeta and bias-gradient reductions autograd launches for them.
 *
 * bg_add_layernorm_fwd:  xn = x + delta [+ delta2] (delta may be NULL: xn = x, x_out untouched; delta2, optional, is the
 *   second update of a parallel-residual block, same dtype and shape as delta);  x_out = xn;
 *   y = (xn - mean) * rstd * gamma + beta, mean / rstd per row (biased variance, eps inside the root).
 *   x, x_out fp32 [rows][C]; delta [rows][C] of delta_dtype; y [rows][C] of y_dtype; mean, rstd fp32 [rows].
 * bg_layernorm_bwd:  dx = dres + LayerNorm'(dy) (dres may be NULL); dx_lp (optional) = dx rounded to lp_dtype
 *   (the gradient the preceding projection's GEMMs consume);  dgamma += sum_r dy * xhat;  dbeta += sum_r dy;
 *   dx_colsum (optional, fp32 [C]) += sum_r dx as that projection sees it (the rounded dx_lp when given): its bias
 *   gradient, so the projection needs no column-sum pass of its own.
 * bg_bias_act_bwd:  act 0: dbias += column sums of dy [rows][C] (row stride dy_row_stride);
 *   act 1 / 2: dx = dy * gelu'(pre) (tanh / erf form), dbias += column sums of dx.
 * C % 4 == 0; LayerNorm: C <= 4096.  The column reductions are deterministic (per-workgroup partials in the
 * workspace, added in a fixed order) and ACCUMULATE into the fp32 gradients.
 *
 * Deferred reductions: a backward pass of L blocks holds ~4L such column reductions, each a ~5 us launch that nothing
 * but the optimizer waits for.  With dgamma == dbeta == NULL (bg_layernorm_bwd; deferred_colsum != 0 asks for the
 * dx column sums too) or dbias == NULL (bg_bias_act_bwd) the call leaves its partials in the workspace -
 * [bg_layernorm_bwd_groups(rows)][2C or 3C] as dgamma | dbeta | colsum, [bg_bias_act_bwd_groups(rows)][C] - and the
 * caller reduces any number of them later with ONE bg_partial_reduce over a descriptor list (out[c] += sum over
 * g < groups of partial[g * stride + c], c < n; same summation order as the immediate form, bit-identical results;
 * descriptors whose `out` ranges overlap are applied one after another in list order, in separate launches).
 * The workspace of a deferred call must stay untouched until that reduction has run.
 */
typedef struct bg_reduce_desc {
  const float* partial;
  float* out;
  int32_t groups, stride, n, reserved;
} bg_reduce_desc;
int bg_partial_reduce(const bg_reduce_desc* descs, int count, void* stream);
/* Row-block gather for the batched weight-gradient GEMMs (no counterpart in the reference, whose autograd issues one dW GEMM per
 * nn.Linear / Conv1D: Models/GPT2BranchAttention/GPT2NonResidualAttention.py:75-80 c_attn / c_proj, transformers' GPT2MLP): every
 * descriptor copies `rows` rows of `row_bytes` bytes, `src_row_stride` bytes apart, to the contiguous `dst`; all of them in
 * launches of up to 48 descriptors.  Rows, strides and pointers are multiples of 16 bytes. */
typedef struct bg_stack_desc {
  const void* src;
  void* dst;
  int64_t rows, row_bytes, src_row_stride;
} bg_stack_desc;
int bg_stack_rows(const bg_stack_desc* descs, int count, void* stream);
int bg_layernorm_bwd_groups(int64_t rows);
int bg_bias_act_bwd_groups(int64_t rows);
int bg_add_layernorm_fwd(const float* x, const void* delta, const void* delta2, int delta_dtype, int64_t rows, int C,
                         const float* gamma, const float* beta, float eps,
                         float* x_out, void* y, int y_dtype, float* mean, float* rstd, void* stream);
/* The same with a SECOND LayerNorm of the same row: y_b = LayerNorm(x + delta + delta2; gamma_b, beta_b) beside y (one pass, one
 * mean / rstd).  For the parallel-residual GPT-NeoX block, whose attention and MLP branches each normalise the same residual
 * stream (NeoXNonResidualBase.py:117-126: input_layernorm and post_attention_layernorm of hidden_states).  gamma_b, beta_b, y_b
 * are given together or all NULL (= bg_add_layernorm_fwd). */
int bg_add_layernorm_pair_fwd(const float* x, const void* delta, const void* delta2, int delta_dtype, int64_t rows, int C,
                              const float* gamma, const float* beta, const float* gamma_b, const float* beta_b, float eps,
                              float* x_out, void* y, void* y_b, int y_dtype, float* mean, float* rstd, void* stream);
size_t bg_layernorm_bwd_workspace_bytes(int64_t rows, int C);
int bg_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd,
                     const float* gamma, const float* dres, int64_t rows, int C,
                     float* dx, void* dx_lp, int lp_dtype, float* dgamma, float* dbeta, float* dx_colsum,
                     int deferred_colsum, void* workspace, size_t workspace_bytes, void* stream);
/* The block's `x + dropout(projection)` inside the LayerNorm kernels (GPT-2: F.dropout(attn_output, resid_pdrop) before the
 * residual add, NonResidualAttention.py:274-275 + NonResidualDiscriminatorBase.py:46-60; the MLP's nn.Dropout, transformers'
 * GPT2MLP): forward  x_out = x + delta * keep / (1 - p), y = LayerNorm(x_out);  backward  dx as bg_layernorm_bwd, while dx_lp and
 * dx_colsum - what delta's projection and its bias receive - carry dx * keep / (1 - p).  The mask is bg_dropout_rows' function
 * of (drop_state, site, batch, logical row, column) with the rows laid out [batch][rows_per_batch] (rows = batch *
 * rows_per_batch): no dropout launch, no second copy of the activation, and the projection's bias gradient still comes out of
 * the LayerNorm backward.  delta / y (forward) and dy / dx_lp (backward) share a dtype (bf16 or fp32); dx_lp is required. */
typedef struct bg_row_dropout {
  float p;                    /* in (0, 1) */
  uint32_t site;
  uint32_t row0;              /* logical number of the first row that is not a text row */
  int32_t text_rows;          /* rows [0, text_rows) of every batch are text positions */
  int32_t rows_per_batch;
  int32_t reserved;
  const uint64_t* drop_state; /* device memory: [seed, step counter] */
} bg_row_dropout;
int bg_add_layernorm_dropout_fwd(const float* x, const void* delta, int delta_dtype, int64_t rows, int C,
                                 const float* gamma, const float* beta, float eps,
                                 float* x_out, void* y, int y_dtype, float* mean, float* rstd,
                                 const bg_row_dropout* drop, void* stream);
int bg_layernorm_dropout_bwd(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd,
                             const float* gamma, const float* dres, int64_t rows, int C,
                             float* dx, void* dx_lp, int lp_dtype, float* dgamma, float* dbeta, float* dx_colsum,
                             int deferred_colsum, void* workspace, size_t workspace_bytes,
                             const bg_row_dropout* drop, void* stream);
size_t bg_bias_act_bwd_workspace_bytes(int64_t rows, int C);
int bg_bias_act_bwd(const void* dy, int dtype, int64_t rows, int C, int64_t dy_row_stride,
                    int act, const void* pre, void* dx, float* dbias,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Token + position embedding of the GPT-2 family.  Replaces `wte(input_ids) + wpe(position_ids)` of the reference's
 * model forward (Models/GPT2BranchAttention/NonResidualDiscriminatorBase.py:300-330; two ATen gathers + an add) and the
 * two embedding_dense_backward calls + AccumulateGrad adds autograd runs for it.
 * bg_embed_fwd:  out[r] = wte[ids[r]] + wpe[pos[r % pos_rows]]   (pos may be NULL: token table only).
 *   ids int64 [rows]; pos int64 [pos_rows] (pos_rows divides the broadcast: [T] positions shared by B sequences);
 *   wte fp32 [vocab][C]; wpe fp32 [n_pos][C]; out fp32 [rows][C]; C % 4 == 0.
 *   A row whose index is outside its table comes back NaN (the reference raises IndexError; a kernel cannot).
 * bg_embed_bwd:  dW[idx[r % idx_rows]] += dy[r] for every row, DETERMINISTIC (no atomics: the rows that share an
 *   index are added in ascending row order by the workgroup of the lowest such row) and accumulating into the
 *   fp32 gradient.  dy fp32 [rows][C]; idx int64 [idx_rows]; dW fp32 [table_rows][C]; rows with an index outside the
 *   table are skipped.  workspace: bg_embed_bwd_workspace_bytes(rows), 4-byte aligned.
 */
int bg_embed_fwd(const int64_t* ids, int64_t rows, const int64_t* pos, int64_t pos_rows, const float* wte, int64_t vocab,
                 const float* wpe, int64_t n_pos, int C, float* out, void* stream);
size_t bg_embed_bwd_workspace_bytes(int64_t rows);
int bg_embed_bwd(const float* dy, const int64_t* idx, int64_t rows, int64_t idx_rows, int64_t table_rows, int C, float* dW,
                 void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Adam over flat arenas: one streaming launch per optimizer step.  Replaces, per model, the reference's
 * torch.optim.Adam step over every parameter tensor, the gradient zeroing that follows it
 * (Training/Optimizers.py:6-16 NAME_TO_OPTIM, :75-92 performUpdateStep: optimizer.step(); optimizer.zero_grad()) and
 * the fp32 -> bf16 weight casts autocast issues on the next forward.
 *   param, grad, exp_avg, exp_avg_sq: fp32 [n], 16-byte aligned, updated in place with torch's order of operations:
 *     g = grad * grad_scale (+ weight_decay * param);  exp_avg += (g - exp_avg) * (1 - beta1);
 *     exp_avg_sq = exp_avg_sq * beta2 + (1 - beta2) * g * g;
 *     param -= lr / (1 - beta1^step) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^step) + eps)
 *   param_lp (optional, bf16 [n]): the updated parameters rounded to nearest even; zero_grad != 0: grad = 0 afterwards.
 *   step counts from 1.  grad_scale folds the 1 / world of a summed all-reduce into the same pass.
 */
int bg_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, void* param_lp, int lp_dtype, int64_t n,
                 double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                 double grad_scale, int zero_grad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused discriminator losses.  Replaces Models/Losses/DiscriminatorLoss.py:4-74 (≈60 ATen launches).
 *   input_scores fp32 [B][S]; alt_scores fp32 [B][N][K] (the K real columns; the reference's zero padding
 *   to S is implicit); depth_mask fp32 [N][K]; txt_in, txt_alt fp32 [B][S]; alt_alt fp32 [N][B][K];
 *   root_links int32 [n_root]; child int32 [N][max_child] (-1 padded) with n_child int32 [N].
 *   n_update = min(numNegativeUpdateSamples, N).
 * out_losses fp32 [5] = {inputLoss, alterationLoss, valueHeadInputLoss, valueHeadAltLoss, valueHeadLoss}.
 * backward: given g_losses fp32 [3] = d(total)/d{inputLoss, alterationLoss, valueHeadLoss}, writes the
 * gradients of input_scores, alt_scores, txt_in, txt_alt, alt_alt (same shapes).
 */
typedef struct {
  int B, S, N, K, n_update, n_root, max_child;
} bg_disc_loss_shape;
int bg_disc_loss_fwd(const bg_disc_loss_shape* shape, const float* input_scores, const float* alt_scores,
                     const float* depth_mask, const float* txt_in, const float* txt_alt, const float* alt_alt,
                     const int32_t* root_links, const int32_t* child, const int32_t* n_child,
                     float* out_losses, void* stream);
int bg_disc_loss_bwd(const bg_disc_loss_shape* shape, const float* input_scores, const float* alt_scores,
                     const float* depth_mask, const float* txt_in, const float* txt_alt, const float* alt_alt,
                     const int32_t* root_links, const int32_t* child, const int32_t* n_child,
                     const float* g_losses,
                     float* d_input_scores, float* d_alt_scores, float* d_txt_in, float* d_txt_alt, float* d_alt_alt,
                     void* stream);

/* Fused generator REINFORCE loss.  Replaces Models/Losses/AdvantageCalculations.py:5-46,
 * ReinforceLoss.py:21-52 and GeneratorLosses.py:84-131.
 *   input_scores fp32 [B][S] (already replaced by ones when teacher forcing); alt_scores fp32 [B][N][K];
 *   in_head, alt_head fp32 [B][S] (text value heads); label_logp fp32 [B][S-1] (log prob of the true next
 *   token); alt_logp fp32 [B][N][K] (log prob of each sampled branch token); depth_mask fp32 [N][K].
 * out fp32 [2] = {loss-RL-input, loss-RL-alterations}.  backward: g fp32 [2] -> d_label_logp, d_alt_logp.
 * Probabilities enter as log-probs; the clamp to [1e-6, 1-1e-6] of ReinforceLoss.py:22,46 is applied to
 * exp(logp) and its zero gradient outside the clamp is reproduced.
 * out_diag (NULL: skipped; else one more launch) fp32 [BG_RL_DIAG_COUNT] = _calculateDebuggingStats (GeneratorLosses.py:34-81)
 * in the reference's dict order: Rl-Loss-{A,B,C}-Input, Rl-Loss-{A,B,C}-Alt, Rl-Advantage-{A,B,C}-Input,
 * Rl-Advantage-{A,B,C}-Alt, RL-Zone{A,B,C}-Input-Factor, RL-Zone{A,B,C}-Alt-Factor, RL-CorrectValueHeads-Factor,
 * RL-CorrectInput-Factor - means over the reference's padded (B, 1+N, S) extent (an empty zone reports 0); no gradient.
 */
#define BG_RL_DIAG_COUNT 20
typedef struct {
  int B, S, N, K;
  float wA, wB, wC;
  int ignore_input, ignore_alt;   /* IncorrectValueHeadsPolicy.IGNORE_SAMPLE for input / alterations */
  int use_depth_mask;             /* 0: reference's depthMask=None branch (mean over [:, :, :-1]) */
  int probs_input;                /* 0: label_logp / alt_logp hold log-probs;  1: they hold probabilities */
} bg_rl_loss_shape;
int bg_rl_loss_fwd(const bg_rl_loss_shape* shape, const float* input_scores, const float* alt_scores,
                   const float* in_head, const float* alt_head, const float* label_logp, const float* alt_logp,
                   const float* depth_mask, float* out_losses, float* out_diag, void* stream);
int bg_rl_loss_bwd(const bg_rl_loss_shape* shape, const float* input_scores, const float* alt_scores,
                   const float* in_head, const float* alt_head, const float* label_logp, const float* alt_logp,
                   const float* depth_mask, const float* g_losses,
                   float* d_label_logp, float* d_alt_logp, void* stream);

/* The element-wise pieces of the generator loss under the reference's own function boundaries (the train step itself runs the
 * fused bg_rl_loss_* above).
 * bg_dual_head_advantages = Models/Losses/AdvantageCalculations.py:5-46 (dualHeadAdvantages):
 *   input_scores fp32 [B][S]; alt_scores fp32 [B][N][S]; in_head, alt_head fp32 [B][S] (the text value heads, squeezed);
 *   scores[b][0][t] = input_scores[b][t+1] (0 at t = S-1), scores[b][1+n][t] = alt_scores[b][n][t];
 *   zone A: score >= in_head, zone C: score < alt_head, zone B: between (xor); zone B is scaled by
 *   (score - alt_head) / max(|in_head - alt_head| + epsilon, min_denominator);
 *   advantages fp32 [B][1+N][S] = A*wA + scaled*B*wB + C*wC, zeroed where alt_head >= in_head for row 0 (ignore_input) /
 *   rows 1.. (ignore_alt): IncorrectValueHeadsPolicy.IGNORE_SAMPLE.
 *   Optional outputs (NULL: skipped): zone_a/b/c fp32 [B][1+N][S]; heads_correct fp32 [B][S] = !(alt_head >= in_head);
 *   input_correct fp32 [B][S] = input_scores >= alt_head.
 * bg_reinforce_terms_* = Models/Losses/ReinforceLoss.py:21-52: per element, with pc = clamp(p, eps, 1 - eps),
 *   loss = (1 - m) * (-log(pc) * adv * f) + m * (-log(1 - pc) * adv);  m = mask_zone_c (NULL: 0 = reinforceLoss, else
 *   mirroredNegativeReinforceLoss);  f = calculateMinimumProbLimit's factor when use_min_prob_limit, else 1.
 *   backward: d_probs = g_loss * dloss/dp, zero where p lies outside [eps, 1 - eps] (the clamp's gradient).
 */
int bg_dual_head_advantages(const float* input_scores, const float* alt_scores, const float* in_head, const float* alt_head,
                            int B, int N, int S, float wA, float wB, float wC, int ignore_input, int ignore_alt,
                            float epsilon, float min_denominator, float* advantages, float* zone_a, float* zone_b,
                            float* zone_c, float* heads_correct, float* input_correct, void* stream);
int bg_reinforce_terms_fwd(const float* probs, const float* advantages, const float* mask_zone_c, int64_t n,
                           int use_min_prob_limit, float neg_push_prob_limit, float eps, float* loss, void* stream);
int bg_reinforce_terms_bwd(const float* probs, const float* advantages, const float* mask_zone_c, int64_t n,
                           int use_min_prob_limit, float neg_push_prob_limit, float eps, const float* g_loss,
                           float* d_probs, void* stream);

/* ------------------------------------------------------------------------------------------------
 * hipGraph hygiene for a captured train step.  No counterpart in the reference (it runs eagerly); these exist because
 * on ROCm 7.2.0 a MEMSET node of an instantiated graph (a captured hipMemsetAsync) clears its destination on the first
 * replay and fills it with garbage from the second on (sizes >= 16 B; benchmarks/probes/graph_anomalies.py), which breaks
 * every captured ATen multi-block reduction (their arrival counters are cleared by such a call).
 * `graph` is a hipGraph_t that has NOT been instantiated yet (host call, no stream).
 * bg_graph_node_census: node counts by type (any output pointer may be NULL).
 * bg_graph_replace_memset_nodes: replaces every memset node by a kernel node (a fill kernel of this library with the same
 *   destination, value, element size, width, height, pitch) wired to the same predecessors and successors.
 */
int bg_graph_node_census(void* graph, int* n_nodes, int* n_kernel, int* n_memset, int* n_memcpy);
int bg_graph_replace_memset_nodes(void* graph, int* n_replaced);

#ifdef __cplusplus
}
#endif
#endif /* BRANCHGAN_H */
