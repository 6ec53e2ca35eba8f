/*
 * uniter_hip.h — C ABI of libuniter_hip.so, the MI355X (gfx950 / CDNA4) implementation of the
 * UNITER encoder training hot path.
 *
 * The reference (ChenRocks/UNITER) has no FFI for this path: its seam is the Python nn.Module /
 * Optimizer surface.  Each entry point below therefore cites the reference Python
 * code (file:line, relative to the reference checkout) whose arithmetic it replaces.  The Python
 * side of the boundary (uniter_amd/model, uniter_amd/optim, uniter_amd/utils) keeps the reference
 * class names / signatures and binds these symbols with ctypes (uniter_amd/_lib.py).
 *
 * Conventions (all entry points):
 *   - return 0 on success, >0 = hipError_t, <0 = argument error (see uniter_hip_last_error());
 *   - never throw, never allocate or free caller memory, never synchronise the device;
 *   - every tensor argument is a raw DEVICE pointer owned by the caller (a torch tensor), contiguous,
 *     16-byte aligned, kept alive by the caller until `stream` has passed the call;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - activations / weights / gradients are bf16 (uint16 storage), statistics and optimizer state fp32;
 *   - nn.Linear weights keep the reference layout [out, in] row-major (model/layer.py:64-66);
 *   - dropout randomness is a counter-based Philox4x32-10 stream keyed by (seed, offset) supplied by
 *     the caller; p == 0 disables it (utils/misc.py:57-63 set_dropout).
 */
#ifndef UNITER_HIP_H
#define UNITER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNITER_HIP_ABI_VERSION 8

/* ------------------------------------------------------------------------------------------------
 * Library
 * ---------------------------------------------------------------------------------------------- */
int uniter_hip_abi_version(void);
/* Thread-local text of the last non-zero status returned on this thread ("" if none). */
const char* uniter_hip_last_error(void);
/* Writes device facts of the current HIP device: [0]=CU count, [1]=wavefront size, [2]=LDS bytes/CU,
 * [3]=gfx arch number (950).  Returns 0 / hipError_t. */
int uniter_hip_device_info(int32_t out[4]);
/* hipGraph support for dropout: when a device counter is registered, every dropout kernel adds *dev_counter to its
 * Philox offset at run time, so a captured graph draws fresh masks on each replay once the counter is advanced
 * (uniter_hip_counter_add, itself capturable).  NULL restores the purely host-supplied offsets. */
int uniter_hip_set_dropout_offset_ptr(const uint64_t* dev_counter);
int uniter_hip_counter_add(uint64_t* dev_counter, uint64_t inc, void* stream);

/* ---- optional per-launch timing (bench.py's roofline object) ------------------------------------------------
 * Between _begin and _end every GEMM / attention / LayerNorm / column-sum / AdamW entry point brackets its launches
 * with two HIP events on the stream it launches on.  _end synchronises the device and reports one record per
 * (kind, M, N, K): number of launches and their summed duration.  Not usable during stream capture.
 * kind: 0 gemm fwd +bias, 1 gemm fwd +bias+gelu, 2 gemm fwd +bias+dropout+residual, 3 gemm dgrad, 4 gemm dgrad x gelu',
 *       5 gemm wgrad (incl. split-K reduce), 6 attention fwd, 7 attention bwd, 8 layernorm fwd, 9 layernorm bwd,
 *       10 column sum, 11 AdamW, 12 layernorm bwd column sums (dgamma/dbeta/dbias; kind 9 is then the row half),
 *       13 grouped gemm wgrad (uniter_gemm_wgrad_group): there N = sum_i N_i*K_i (weight-gradient elements produced, so
 *       the launch's FLOP is 2*M*N) and K = the number of problems in the group.
 *       The reference has no counterpart (it profiles with nvprof / apex timers). */
typedef struct {
    int32_t kind;
    int32_t calls;
    int64_t M, N, K;
    double total_us;
} UniterTimingRecord;
int uniter_hip_timing_begin(void);
int uniter_hip_timing_end(UniterTimingRecord* out, int32_t cap, int32_t* n_out);

/* ------------------------------------------------------------------------------------------------
 * GEMM family — bf16 operands, fp32 MFMA accumulation, fused epilogues.
 * Replaces the cuBLAS calls behind nn.Linear in model/layer.py:76-78,112,140,153 (forward) and the
 * autograd-generated dgrad / wgrad GEMMs of the same layers.
 *   M = rows (tokens, B*L), K = in_features, N = out_features for the forward layer.
 *   Requirements: N % 64 == 0, K % 64 == 0 for fwd/dgrad shapes; wgrad needs N % 64 == 0 and K % 64 == 0.
 * ---------------------------------------------------------------------------------------------- */

/* Empirical tile selection (SYNCHRONOUS, allocates scratch: set-up time only): times every legal tile shape — and
 * split-K factor for wgrad — of one GEMM and caches the winner for (kind, M, N, K); later launches of that shape use
 * it.  kind 0 = forward, 1 = dgrad, 2 = wgrad, with M, N, K as in the corresponding call below.
 * uniter_gemm_tuned_choice reports the cached (tile index, splits) or (-1, -1); uniter_gemm_set_tuned installs a choice
 * saved from an earlier process (checked against the same legality rules the sweep uses), so that a job can skip the
 * sweep and run-to-run kernel selection is reproducible. */
int uniter_gemm_autotune(int kind, int64_t M, int64_t N, int64_t K, void* stream);
int uniter_gemm_set_tuned(int kind, int64_t M, int64_t N, int64_t K, int32_t cfg, int32_t splits);
/* Number of tile configurations the GEMM family was built with: saved tile choices (indices) are only valid for the
 * build that produced them. */
int uniter_gemm_tile_count(void);
int uniter_gemm_tuned_choice(int kind, int64_t M, int64_t N, int64_t K, int32_t out[2]);

/* y[M,N] = x[M,K] * w[N,K]^T + bias[N]        (bias may be NULL)                 layer.py:76-78 */
int uniter_gemm_bias_fwd(const void* x, const void* w, const void* bias, void* y,
                         int64_t M, int64_t N, int64_t K, void* stream);

/* Up to four such GEMMs over the same M rows and contraction K in ONE launch (ABI v7): y_q[M, N_q] = x_q w_q^T + bias_q with
 * per-problem row strides (0 = dense).  The two MultiheadAttention modules of the NLVR2 paired-attention head (model/nlvr2.py:
 * 170-189, model/attention.py:103-127) are four input projections and two output projections of 1 536 rows each.  Shapes no
 * grouped tile divides fall back to one launch per problem inside the call. */
int uniter_gemm_bias_fwd_group(int32_t n, const void* const* x, const int64_t* ldx, const void* const* w, const void* const* bias,
                               void* const* y, const int64_t* ldy, int64_t M, const int64_t* N, int64_t K, void* stream);
/* ... and their data gradients: dx_q[M, K] = dy_q[M, N_q] w_q[N_q, K] (+ resid_q[M, K]), row stride lddy_q on dy (0 = dense). */
int uniter_gemm_dgrad_group(int32_t n, const void* const* dy, const int64_t* lddy, const void* const* w, const void* const* resid,
                            void* const* dx, int64_t M, const int64_t* N, int64_t K, void* stream);

/* u = x*w^T + bias ; g = u*0.5*(1+erf(u/sqrt2))   writes both u (pre-activation, kept for backward)
 * and g.                                                                    layer.py:31-37,139-142 */
int uniter_gemm_bias_gelu_fwd(const void* x, const void* w, const void* bias, void* u, void* g,
                              int64_t M, int64_t N, int64_t K, void* stream);

/* z = dropout_p(x*w^T + bias) + resid     (the input of the following LayerNorm)
 *                                                                      layer.py:111-115,152-156 */
int uniter_gemm_bias_dropout_residual_fwd(const void* x, const void* w, const void* bias,
                                          const void* resid, void* z,
                                          int64_t M, int64_t N, int64_t K,
                                          float p_drop, uint64_t seed, uint64_t offset, void* stream);

/* dx[M,K] = dy[M,N] * w[N,K]  (+ resid[M,K] if resid != NULL)          autograd of layer.py:76-78 */
int uniter_gemm_dgrad(const void* dy, const void* w, const void* resid, void* dx,
                      int64_t M, int64_t N, int64_t K, void* stream);

/* dpre[M,K] = (dy[M,N] * w[N,K]) .* gelu'(u[M,K])                      autograd of layer.py:139-142 */
int uniter_gemm_dgrad_gelu(const void* dy, const void* w, const void* u, void* dpre,
                           int64_t M, int64_t N, int64_t K, void* stream);

/* dw[N,K] (+)= dy[M,N]^T * x[M,K] ; db[N] (+)= column sums of dy (db may be NULL).
 * accumulate != 0 adds to the existing contents (gradient accumulation, pretrain.py:298-312).
 * workspace: fp32 scratch of at least uniter_gemm_wgrad_workspace_bytes(M,N,K) bytes. */
size_t uniter_gemm_wgrad_workspace_bytes(int64_t M, int64_t N, int64_t K);
int uniter_gemm_wgrad(const void* dy, const void* x, void* dw, void* db,
                      int64_t M, int64_t N, int64_t K, int accumulate,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Up to four weight gradients over the same M tokens in ONE launch: dw_q[N_q,K_q] (+)= dy_q[M,N_q]^T * x_q[M,K_q],
 * q < n <= 4 (host arrays of device pointers / sizes).  The four nn.Linear weight gradients of a BertLayer
 * (model/layer.py:64-66,112,140,153) are independent of each other; one grid pays one launch / ramp / drain and the
 * tiles of the small problems fill the CUs the big ones leave idle.  No split-K; N_q % 64 == 0, K_q % 64 == 0.
 * db (host array, may be NULL; entries may be NULL): db_q[N_q] (+)= column sums of dy_q — the nn.Linear bias gradients.
 * They come out of the same launch: the tiles of the first tile column multiply the dy fragments they already hold by a
 * fragment of ones (one extra MFMA each), so no separate reduction kernel reads dy again. */
int uniter_gemm_wgrad_group(int32_t n, const void* const* dy, const int64_t* lddy, const void* const* x, const int64_t* ldx,
                            void* const* dw, void* const* db, int64_t M, const int64_t* N, const int64_t* K, int accumulate,
                            void* stream);
/* lddy / ldx (host arrays, may be NULL = contiguous; an entry of 0 = contiguous): row strides in elements when dy_q / x_q
 * are column slices of wider matrices (the q / k|v blocks of a packed projection buffer).
 * uniter_gemm_wgrad_group_autotune times the legal tiles of such a group once (synchronous, set-up time) and remembers
 * the winner under kind 3, (M, sum N, sum K). */
int uniter_gemm_wgrad_group_autotune(int32_t n, int64_t M, const int64_t* N, const int64_t* K, void* stream);
/* The same with a workspace and an explicit tile.  When every N_q and K_q is a multiple of 256 the group may run on the
 * 256 x 256 eight-phase tile (csrc/gemm8.cuh; tile index uniter_gemm_tile_count() - 1): with a workspace of
 * uniter_gemm_wgrad_group_workspace_bytes(n, N, K) bytes (one fp32 slab per output tile) every tile is computed as TWO
 * slices of the token contraction on two workgroups, which combine inside the launch — the first to finish parks its
 * accumulators in the slab, the other adds them and writes the gradient — and the bias gradients are summed by extra
 * workgroups of the same grid.  cfg < 0: the tuned choice (kind 3); splits: 0 = tuned, else 1 or 2. */
size_t uniter_gemm_wgrad_group_workspace_bytes(int32_t n, const int64_t* N, const int64_t* K);
int uniter_gemm_wgrad_group_ws(int32_t n, const void* const* dy, const int64_t* lddy, const void* const* x, const int64_t* ldx,
                               void* const* dw, void* const* db, int64_t M, const int64_t* N, const int64_t* K, int accumulate,
                               void* workspace, size_t workspace_bytes, int cfg, int splits, void* stream);

/* Strided-operand variants (row stride in elements; operands may be column slices of a wider row-major matrix —
 * e.g. the q / k|v column blocks of a packed [T,3H] projection buffer).  Used by the NLVR2 paired cross-attention
 * head (model/nlvr2.py:170-189 + model/attention.py:86-127: q projected from one sequence, k|v from its partner):
 *   y[M,N] (ldy)  = x[M,K] (ldx) * w[N,K]^T + bias
 *   dx[M,K]       = dy[M,N] (lddy) * w[N,K] (+ resid[M,K])
 *   dw[N,K] (+)=    dy[M,N]^T (lddy) * x[M,K] (ldx)
 * ld* must be multiples of 8 elements (16-byte rows) and >= the logical width. */
int uniter_gemm_bias_fwd_ld(const void* x, int64_t ldx, const void* w, const void* bias, void* y, int64_t ldy,
                            int64_t M, int64_t N, int64_t K, void* stream);
int uniter_gemm_dgrad_ld(const void* dy, int64_t lddy, const void* w, const void* resid, void* dx,
                         int64_t M, int64_t N, int64_t K, void* stream);
int uniter_gemm_wgrad_ld(const void* dy, int64_t lddy, const void* x, int64_t ldx, void* dw, void* db,
                         int64_t M, int64_t N, int64_t K, int accumulate,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused self-attention (scale + additive key mask + softmax + dropout + P·V), head_dim fixed at 64,
 * L <= 512 (max_position_embeddings of the shipped configs; pretrain.py:637-640).  Up to L = 256 one workgroup per
 * (example, head) keeps Q, K, V, dO in LDS for the backward pass; beyond, the backward pass is two launches (dQ, then
 * dK / dV) and needs the small workspace of uniter_attention_bwd_workspace_bytes — use uniter_attention_bwd_ws.
 * qkv is the fused projection output [B*L, 3H] = [Q | K | V], head h owns columns
 * h*64..h*64+63 of each third.  mask_bias[B,L] fp32 is the reference's extended_attention_mask
 * (1-m)*-10000 (model/model.py:342-345) without the two singleton dims.   model/layer.py:75-101
 * ---------------------------------------------------------------------------------------------- */
/* ctx[B*L,H] ; lse[B,heads,L] fp32 = log-sum-exp of the masked scaled scores (kept for backward). */
int uniter_attention_fwd(const void* qkv, const float* mask_bias, void* ctx, float* lse,
                         int64_t B, int64_t L, int64_t heads,
                         float p_drop, uint64_t seed, uint64_t offset, void* stream);
/* BertSelfAttention.forward in ONE launch (model/layer.py:75-101: the three nn.Linear(H, H) as one [3H, H] projection, then the
 * attention above): qkv[B*L, 3H] = x wqkv^T + bqkv (stored: the backward reads it), ctx, lse.  For dense batches of L == 96 tokens
 * (60 text + 36 regions, the shipped finetune / pretrain shape) and 64-wide heads tile (example, head) of a 96 x 192 GEMM tile keeps
 * its Q, K, V in LDS and runs the unit's attention in its epilogue; other shapes run the two launches inside the call.  Bit-identical
 * to uniter_gemm_bias_fwd + uniter_attention_fwd with the same (p_drop, seed, offset). */
int uniter_qkv_attention_fwd(const void* x, const void* wqkv, const void* bqkv, const float* mask_bias, void* qkv, void* ctx, float* lse,
                             int64_t B, int64_t L, int64_t heads, float p_drop, uint64_t seed, uint64_t offset, void* stream);
/* dqkv[B*L,3H] from dctx[B*L,H]; needs the forward's qkv, ctx, lse and the same (seed, offset). */
int uniter_attention_bwd(const void* qkv, const float* mask_bias, const void* ctx, const float* lse,
                         const void* dctx, void* dqkv,
                         int64_t B, int64_t L, int64_t heads,
                         float p_drop, uint64_t seed, uint64_t offset, void* stream);

/* Backward with a caller workspace (required for 256 < L <= 512, 0 bytes otherwise).  Dense: mask_bias != NULL,
 * cu_seqlens == NULL; packed: cu_seqlens != NULL (mask_bias ignored) and L = the longest example. */
size_t uniter_attention_bwd_workspace_bytes(int64_t B, int64_t L, int64_t heads);
int uniter_attention_bwd_ws(const void* qkv, const float* mask_bias, const int32_t* cu_seqlens, const void* ctx, const float* lse,
                            const void* dctx, void* dqkv, int64_t B, int64_t L, int64_t heads,
                            float p_drop, uint64_t seed, uint64_t offset, void* workspace, size_t workspace_bytes, void* stream);

/* Packed ("varlen") forms: qkv [total, 3H], ctx / dctx [total, H], dqkv [total, 3H]; example b owns rows
 * cu_seqlens[b] .. cu_seqlens[b+1]-1 (device int32 [B+1]); max_len = longest example (<= 256 for the backward entry below, <= 512 through uniter_attention_bwd_ws); lse [B, heads, max_len].
 * No key mask: only real tokens exist. */
int uniter_attention_fwd_packed(const void* qkv, const int32_t* cu_seqlens, void* ctx, float* lse,
                                int64_t B, int64_t max_len, int64_t heads,
                                float p_drop, uint64_t seed, uint64_t offset, void* stream);
int uniter_attention_bwd_packed(const void* qkv, const int32_t* cu_seqlens, const void* ctx, const float* lse,
                                const void* dctx, void* dqkv, int64_t B, int64_t max_len, int64_t heads,
                                float p_drop, uint64_t seed, uint64_t offset, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (apex FusedLayerNorm semantics: biased variance, eps inside the sqrt, fp32 statistics).
 * Call sites: model/layer.py:108,149 ; model/model.py:229,252,253,258.
 * ---------------------------------------------------------------------------------------------- */
/* y = (z-mean)*rstd*gamma + beta ; mean/rstd [rows] fp32 are written for backward (may be NULL).
 * If p_drop > 0 the dropout of the embedding blocks (model/model.py:230,259) is applied to y. */
int uniter_layernorm_fwd(const void* z, const void* gamma, const void* beta, void* y,
                         float* mean, float* rstd, int64_t rows, int64_t H, float eps,
                         float p_drop, uint64_t seed, uint64_t offset, void* stream);

/* Backward of  z = dropout_p(d) + r ; y = LN(z):
 *   dz  [rows,H]  = dL/dz                       (gradient of the residual branch r)
 *   dd  [rows,H]  = dz .* keep/(1-p)            (gradient of the dense output d; NULL -> not written; with p == 0 it
 *                                                is a plain copy of dz)
 *   dgamma, dbeta [H] (+)= ...                  (bf16, accumulate flag)
 *   dbias [H] (+)= column sums of dd            (bias gradient of the dense layer feeding the LN; may be NULL)
 * dy_extra (may be NULL) is added to dy first (lets a caller fold a second incoming gradient in).
 * drop_on_output != 0: the dropout was applied to the LN OUTPUT (embedding blocks, model/model.py:230,259):
 *   dy is multiplied by the keep mask on load, dd is not produced.
 * workspace: fp32 scratch >= uniter_layernorm_bwd_workspace_bytes(rows,H). */
size_t uniter_layernorm_bwd_workspace_bytes(int64_t rows, int64_t H);
int uniter_layernorm_bwd(const void* dy, const void* dy_extra, const void* z, const float* mean,
                         const float* rstd, const void* gamma,
                         void* dz, void* dd, void* dgamma, void* dbeta, void* dbias,
                         int64_t rows, int64_t H, int accumulate,
                         float p_drop, uint64_t seed, uint64_t offset, int drop_on_output,
                         void* workspace, size_t workspace_bytes, void* stream);

/* out[N] (+)= column sums of a[rows,N] (bias gradients of QKV / FFN1).  autograd of layer.py:76-78,140 */
/* Split-K input gradient dx[M,K] = dy[M,N] * w[N,K] for a short M against a long contraction N (the MLM decoder:
 * a few hundred masked rows x 28996 classes, model/layer.py:205-222): slices of the contraction fill the chip, fp32
 * partials in `workspace` (>= uniter_gemm_dgrad_splitk_workspace_bytes) are then summed.  N % 64 == 0, K % 64 == 0. */
size_t uniter_gemm_dgrad_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K);
int uniter_gemm_dgrad_splitk(const void* dy, int64_t lddy, const void* w, void* dx, int64_t M, int64_t N, int64_t K,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pre-training output heads          model/layer.py:188-222, model/pretrain.py:36-47
 * The heads' GEMMs are the entry points above (strided variants over a padded logits buffer); these are the
 * HBM-bound pieces around them.
 * uniter_ce_fwd: F.cross_entropy(logits.float(), labels, ignore_index=-1, reduction='none') (model/pretrain.py:129-133)
 *   over bf16 logits [n, V] with row stride ld (elements, ld % 8 == 0 for the vector path): loss[n], lse[n] (saved).
 *   Rows whose label is < 0 get loss 0.
 * uniter_ce_bwd: overwrites the logits with d loss / d logits = (softmax - one_hot(label)) * gout[row] (0 for
 *   ignored rows); columns >= V of the buffer are left untouched.
 * uniter_gelu_bwd: dx = dy .* gelu'(u), the element-wise backward of the transform's GELU (model/layer.py:188-203). */
int uniter_ce_fwd(const void* logits, int64_t ld, const int64_t* labels, float* loss, float* lse,
                  int64_t n, int64_t V, void* stream);
int uniter_ce_bwd(void* logits, int64_t ld, const int64_t* labels, const float* lse, const float* gout,
                  int64_t n, int64_t V, void* stream);
int uniter_gelu_bwd(const void* dy, const void* u, void* dx, int64_t numel, void* stream);

/* The whole head in one call each way (all launches issued from C++, like the encoder):
 *   t = LayerNorm(gelu(x * dense_w^T + dense_b)) ; logits = t * proj_w^T + proj_b ; loss = cross_entropy(logits, labels)
 * = BertLMPredictionHead + F.cross_entropy(reduction='none') for MLM (model/layer.py:188-222, model/pretrain.py:129-133;
 * proj_w is then the word-embedding table, V = 28996) and RegionClassification + hard-label cross entropy for MRC
 * (model/pretrain.py:36-47,222-229; V = 1601).  x [n, H] bf16, labels [n] (negative = ignored row), loss [n] fp32.
 * `save` (>= uniter_head_ce_save_bytes) carries the activations and the bf16 logits to the backward call, which turns the
 * logits into d loss / d logits in place, accumulates every parameter gradient into g_* (all required except g_proj_b,
 * which may be NULL together with proj_b) and writes dx [n, H] (may be NULL).  V need not be a multiple of the GEMM tile:
 * the last V % 64 classes are handled exactly by sliver kernels.  A saved forward can be back-propagated once. */
typedef struct {
    const void *dense_w, *dense_b, *ln_g, *ln_b, *proj_w, *proj_b;
    void *g_dense_w, *g_dense_b, *g_ln_g, *g_ln_b, *g_proj_w, *g_proj_b;
} UniterHeadParams;
size_t uniter_head_ce_save_bytes(int64_t n, int64_t H, int64_t V);
size_t uniter_head_ce_workspace_bytes(int64_t n, int64_t H, int64_t V);
int uniter_head_ce_fwd(const UniterHeadParams* p, const void* x, const int64_t* labels, float* loss, void* save,
                       int64_t n, int64_t H, int64_t V, float eps, void* stream);
int uniter_head_ce_bwd(const UniterHeadParams* p, const void* x, const int64_t* labels, const float* gloss, void* dx,
                       void* save, void* workspace, size_t workspace_bytes, int64_t n, int64_t H, int64_t V, void* stream);
/* Same head with the MRC-KL loss (model/pretrain.py:217-221): loss[n, V] fp32 = F.kl_div(log_softmax(logits), target,
 * reduction='none') element-wise against fp32 soft labels target[n, V] (contiguous); the backward call takes the
 * element-wise gradient gloss[n, V].  save / workspace sizes as for the cross-entropy pair. */
int uniter_head_kl_fwd(const UniterHeadParams* p, const void* x, const float* target, float* loss, void* save,
                       int64_t n, int64_t H, int64_t V, float eps, void* stream);
int uniter_head_kl_bwd(const UniterHeadParams* p, const void* x, const float* target, const float* gloss, void* dx,
                       void* save, void* workspace, size_t workspace_bytes, int64_t n, int64_t H, int64_t V, void* stream);

size_t uniter_colsum_workspace_bytes(int64_t rows, int64_t N);
int uniter_colsum(const void* a, void* out, int64_t rows, int64_t N, int accumulate,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Embeddings                                                         model/model.py:217-272,321-334
 * ---------------------------------------------------------------------------------------------- */
/* fp32 scratch size (bytes) sufficient for every uniter_embed_*_bwd call on a [rows, N] gradient. */
size_t uniter_embed_ws_bytes(int64_t rows, int64_t N);

/* z[b,t,:] = word[ids[b,t]] + pos[position_ids[t]] + type[type_ids[b,t] or 0]   (pre-LayerNorm sum, bf16)
 * ids/type_ids int64 [B,Lt]; position_ids int64 [Lt] (broadcast over the batch, data/mlm.py:115-116);
 * type_ids may be NULL (= zeros, model/model.py:233-234).  The LayerNorm + dropout that follow are
 * uniter_layernorm_fwd. */
int uniter_embed_txt_fwd(const int64_t* ids, const int64_t* position_ids, const int64_t* type_ids,
                         const void* word, const void* pos, const void* type, void* z,
                         int64_t B, int64_t Lt, int64_t H, int64_t vocab, int64_t max_pos,
                         int64_t n_types, void* stream);
/* Scatter dz[B*Lt,H] into the three tables' gradients (always accumulating; deterministic: one owner
 * per distinct table row, fp32 summation). */
int uniter_embed_txt_bwd(const int64_t* ids, const int64_t* position_ids, const int64_t* type_ids,
                         const void* dz, void* dword, void* dpos, void* dtype,
                         int64_t B, int64_t Lt, int64_t H, int64_t vocab, int64_t max_pos,
                         int64_t n_types, void* stream);

/* f_out[rows,D] (bf16) = img_feat[rows,D] (fp32 or bf16, see feat_is_fp32) + (img_masks[row] ? mask_row[D] : 0)
 *                                                                        model/model.py:262-265 */
int uniter_embed_img_prep(const void* img_feat, int feat_is_fp32, const uint8_t* img_masks,
                          const void* mask_row, void* f_out, int64_t rows, int64_t D, void* stream);

/* pos_lin[rows,H] = pos_feat[rows,7] * wpos[H,7]^T + bpos   (K=7 is too thin for MFMA)  model.py:268 */
int uniter_embed_pos_linear_fwd(const void* pos_feat, int feat_is_fp32, const void* wpos,
                                const void* bpos, void* out, int64_t rows, int64_t H, void* stream);
/* dwpos[H,7] (+)= d^T * pos_feat ; dbpos[H] (+)= colsum(d) */
int uniter_embed_pos_linear_bwd(const void* pos_feat, int feat_is_fp32, const void* d,
                                void* dwpos, void* dbpos, int64_t rows, int64_t H,
                                void* workspace, size_t workspace_bytes, void* stream);

/* z[r,:] = a[r,:] + b[r,:] + type[type_ids[r] (or 1 if NULL)]   (input of the last image LayerNorm)
 *                                                                model/model.py:269 ; :313-316 */
int uniter_embed_img_combine_fwd(const void* a, const void* b, const int64_t* type_ids, const void* type,
                                 void* z, int64_t rows, int64_t H, int64_t n_types, void* stream);
/* dtype_table[n_types,H] += per-type column sums of dz[rows,H] (type_ids NULL -> all rows are type 1). */
int uniter_embed_type_bwd(const void* dz, const int64_t* type_ids, void* dtype_table,
                          int64_t rows, int64_t H, int64_t n_types, int default_type,
                          void* workspace, size_t workspace_bytes, void* stream);
/* dmask_row[D] += sum over rows with img_masks[row] != 0 of df[row,:]   (mask_embedding.weight[1]) */
int uniter_embed_mask_bwd(const void* df, const uint8_t* img_masks, void* dmask_row,
                          int64_t rows, int64_t D, void* workspace, size_t workspace_bytes, void* stream);

/* out[b,j,:] = cat(txt[b], img[b])[gather_index[b,j], :]                  model/model.py:330-333 */
int uniter_embed_gather_fwd(const void* txt, const void* img, const int64_t* gather_index, void* out,
                            int64_t B, int64_t Lt, int64_t Li, int64_t Lout, int64_t H, void* stream);
/* dtxt / dimg = scatter-add of dout through gather_index (deterministic, overwrite semantics). */
int uniter_embed_gather_bwd(const void* dout, const int64_t* gather_index, void* dtxt, void* dimg,
                            int64_t B, int64_t Lt, int64_t Li, int64_t Lout, int64_t H, void* stream);

/* mask_bias[B,L] fp32 = (1 - attn_masks[B,L]) * -10000                    model/model.py:342-345 */
int uniter_mask_bias(const int64_t* attn_masks, float* mask_bias, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-encoder entry points (one C call launches every kernel of layers [layer_begin, layer_end)).
 * Replaces UniterEncoder.forward (model/model.py:282-292) -> BertLayer.forward (model/layer.py:166-170)
 * and its autograd backward.
 * ---------------------------------------------------------------------------------------------- */
typedef struct UniterLayerParams {
    /* parameters, bf16.  wqkv = [query.weight; key.weight; value.weight] stacked to [3H,H]
     * (three separate nn.Linear in the reference, model/layer.py:64-66; the Python side keeps
     * them as three named Parameters that are views of one buffer). */
    const void *wqkv, *bqkv;          /* [3H,H], [3H]                attention.self.{query,key,value} */
    const void *wo, *bo;              /* [H,H],  [H]                 attention.output.dense           */
    const void *ln1_g, *ln1_b;        /* [H]                         attention.output.LayerNorm       */
    const void *w1, *b1;              /* [I,H],  [I]                 intermediate.dense               */
    const void *w2, *b2;              /* [H,I],  [H]                 output.dense                     */
    const void *ln2_g, *ln2_b;        /* [H]                         output.LayerNorm                 */
    /* gradients, bf16, same shapes; accumulated into (+=).  Unused by forward. */
    void *g_wqkv, *g_bqkv, *g_wo, *g_bo, *g_ln1_g, *g_ln1_b, *g_w1, *g_b1, *g_w2, *g_b2, *g_ln2_g, *g_ln2_b;
} UniterLayerParams;

typedef struct UniterEncoderShape {
    int64_t B, L, H, heads, I;
    float p_hidden;      /* hidden_dropout_prob           (config/uniter-base.json:5) */
    float p_attn;        /* attention_probs_dropout_prob  (config/uniter-base.json:2) */
    float ln_eps;        /* 1e-12 (model/layer.py:108)    */
    int32_t training;    /* 0: inference (no dropout, activations not kept) */
    /* Padding-free ("packed") execution, SURVEY.md §8 f-3.  total_tokens == 0: dense [B, L, H] rows (every example
     * has L rows, padded keys masked through mask_bias).  total_tokens > 0: x / y / dy / dx are [total_tokens, H], example
     * b owns rows cu_seqlens[b] .. cu_seqlens[b+1]-1 (device int32 [B+1], cu_seqlens[B] == total_tokens), L is the
     * longest example; mask_bias is ignored (may be NULL) — only real tokens exist.  Every GEMM / LayerNorm then runs
     * over total_tokens rows and attention over each example's own length (data/data.py:271-279 builds exactly this
     * compact order; data/sampler.py:31-57 batches by token count). */
    int64_t total_tokens;
    const int32_t* cu_seqlens;
    int32_t hidden_act;  /* config.hidden_act (model/layer.py:44 ACT2FN): 0 = gelu (exact erf form), 1 = relu, 2 = swish */
} UniterEncoderShape;

/* Bytes of saved activations per layer / of shared scratch, for the caller to allocate. */
size_t uniter_encoder_layer_act_bytes(const UniterEncoderShape* s);
size_t uniter_encoder_scratch_bytes(const UniterEncoderShape* s);
/* Byte offset (inside one layer's activation block) of the layer OUTPUT y [B*L,H] bf16. */
size_t uniter_encoder_layer_out_offset(const UniterEncoderShape* s);

/* Forward of layers [layer_begin, layer_end).  x_in [B*L,H] is the input of layer_begin.
 * acts: n_layers * uniter_encoder_layer_act_bytes() bytes (block l belongs to layer l);
 * the output of layer l is at acts + l*act_bytes + out_offset.  scratch: uniter_encoder_scratch_bytes().
 * Dropout stream for layer l uses offset + l*8 + site. */
int uniter_encoder_forward(const UniterEncoderShape* s, const UniterLayerParams* layers,
                           int32_t layer_begin, int32_t layer_end,
                           const void* x_in, const float* mask_bias,
                           void* acts, void* scratch, uint64_t seed, uint64_t offset, void* stream);

/* Backward of layers [layer_begin, layer_end), run from layer_end-1 down to layer_begin.
 * dy [B*L,H]: gradient w.r.t. the output of layer layer_end-1 (read only).
 * dx [B*L,H]: receives the gradient w.r.t. x_in of layer_begin.
 * Parameter gradients are ACCUMULATED into layers[l].g_*.  Needs the forward's acts and (seed, offset). */
int uniter_encoder_backward(const UniterEncoderShape* s, const UniterLayerParams* layers,
                            int32_t layer_begin, int32_t layer_end,
                            const void* x_in, const float* mask_bias, const void* dy, void* dx,
                            void* acts, void* scratch, uint64_t seed, uint64_t offset, void* stream);

/* Deferred weight gradients.  Nothing downstream of backward needs a weight gradient before the optimizer (or the gradient
 * bucket's allreduce), and one layer's four weight gradients fill less than half of an MI355X.  When the calling thread has
 * registered a stage of at least uniter_encoder_wgrad_stage_bytes(s, layer_end - layer_begin) bytes, uniter_encoder_backward
 * keeps the dy operands of EVERY layer of the call there (one set per layer: the gradients w.r.t. the outputs of the four
 * nn.Linear of model/layer.py:64-66,112,140,153 and the inputs of the two LayerNorm backward passes) and computes all their weight
 * + bias gradients and the LayerNorm weight / bias gradients in ONE launch at the end of the call (which also reads the caller's
 * dy: keep it valid until the weight-gradient stream has been joined) (hidden and intermediate sizes must be multiples of 256, the token count a multiple of 64; otherwise — and during
 * stream capture — the per-layer grouped launches run as before).  A stage of twice that size
 * lets consecutive calls (layer ranges of one backward) alternate halves instead of waiting for each other's launch.  The
 * registration is per calling thread (autograd runs backward on its own thread) and is read at the start of each call;
 * buf = NULL unregisters. */
size_t uniter_encoder_wgrad_stage_bytes(const UniterEncoderShape* s, int32_t n_layers);
int uniter_encoder_set_wgrad_stage(void* buf, size_t bytes);

/* Backward keeps the weight-gradient work on an internal side stream and, at the end of every call, makes `stream` wait for
 * it.  A caller that runs the stack as several layer ranges (one per gradient bucket of a data-parallel reducer) can avoid
 * serialising the two streams at every range boundary: uniter_encoder_defer_side_join(1) makes the following calls of THIS
 * thread return without that wait — the parameter gradients of those ranges are then only complete on a stream that has
 * called uniter_encoder_side_join(stream) after them (typically the communication stream of the bucket) — and
 * uniter_encoder_defer_side_join(0) before the last range restores the default, whose end-of-call wait covers everything.
 * (model/model.py:282-292 has no counterpart: the reference's autograd runs layer by layer on one stream.) */
int uniter_encoder_defer_side_join(int enable);

/* Gradient overwrite (round 6; torch's zero_grad(set_to_none=True) semantics for the encoder's parameter gradients).
 * uniter_encoder_set_grad_overwrite(1) makes the NEXT uniter_encoder_backward call of this thread REPLACE the contents of g_w*, g_b*,
 * g_ln* of the layers it covers instead of adding to them (the flag is consumed by that call): the deferred launch neither reads the old
 * gradients nor needs them zeroed, so a fused optimizer step can leave them alone (uniter_adamw_plan_keep_grads).  Flows without the
 * deferred launch zero the tensors first and accumulate — same result, no saving.  Gradient accumulation over several backward calls
 * (pretrain.py:298-312): state it for the first call of an optimizer step only. */
int uniter_encoder_set_grad_overwrite(int32_t enable);

/* Gradient-norm partials (round 6).  After uniter_encoder_set_grad_sq(1) (per thread, sticky) a uniter_encoder_backward call whose
 * parameter gradients go out as the one deferred launch — and not as data-parallel buckets, whose gradients are reduced across ranks
 * before their norm is taken — also leaves one float per 256 x 256 weight-gradient tile: the sum of squares of the bf16 values it stored
 * (old gradient included when it accumulated).  uniter_encoder_last_grad_sq returns that device array (library-owned, valid until the
 * next backward call of this thread; written on the weight-gradient stream: join it first) and its length, or NULL / 0 when the last
 * call produced none.  Together the floats are sum g^2 over the wqkv, wo, w1, w2 gradients of the layers of that call. */
int uniter_encoder_set_grad_sq(int32_t enable);
int uniter_encoder_last_grad_sq(void** partials_out, int32_t* n_out);
int uniter_encoder_side_join(void* stream);
/* The same from ANY thread: makes `stream` wait for the weight-gradient streams of every thread of this process that left a
 * backward call un-joined on the current device (autograd runs backward on its own thread, the optimizer runs on the caller's).
 * A training loop that defers the join of its one backward call lets the embedding backward overlap the deferred weight-gradient
 * launch, and calls this before anything reads a weight gradient (clip_grad_norm_, optimizer.step, zero_grad). */
int uniter_encoder_side_join_all(void* stream);
/* Gradient buckets of a data-parallel step (ABI v7; utils/distributed.py:16-43 and pretrain.py:298-312 allreduce all gradients
 * after backward; this lets the reduction of the top layers' gradients start while the rest are still being computed WITHOUT
 * cutting the backward call into per-bucket ranges).  uniter_encoder_set_grad_buckets(L) — per thread, 0 = off — groups the
 * layers of the following uniter_encoder_backward calls into buckets of L layers, top layers first; the deferred launch then
 * completes the buckets in that order and raises one flag per bucket.  uniter_encoder_grad_bucket_count reports how many buckets
 * the thread's last backward call completed that way (0: it ran without a stage / without buckets — use
 * uniter_encoder_side_join instead), and uniter_encoder_bucket_wait(k, stream) makes `stream` (a communication stream) wait
 * for bucket k of that call (hipStreamWaitValue32 on a word of signal memory): everything enqueued on `stream` afterwards
 * sees the weight, bias and LayerNorm-parameter gradients of the bucket's layers complete.  Epoch counter wraps at 2^32 calls. */
int uniter_encoder_set_grad_buckets(int32_t layers_per_bucket);
int uniter_encoder_grad_bucket_count(int32_t* n_out);
int uniter_encoder_bucket_wait(int32_t bucket, void* stream);
/* The same wait in two halves, for a caller that learns about the buckets on one thread (autograd's, inside backward) and
 * enqueues the collectives later from another: _token returns the flag's address and the value bucket k of the thread's last
 * backward call will write; uniter_hip_stream_wait_value32 — callable from any thread — makes `stream` wait until the word at
 * `flag` (signal memory) is >= value.  (Issuing the waits and collectives only after backward() has returned keeps the host
 * from delaying the embedding backward, which should start while the deferred launch is young.) */
int uniter_encoder_bucket_token(int32_t bucket, void** flag_out, uint32_t* value_out);
int uniter_hip_stream_wait_value32(void* stream, void* flag, uint32_t value);

/* The calling thread's weight-gradient stream (created on first use), as a raw hipStream_t.  A deferred launch reads the
 * call's activations / input / dy after uniter_encoder_backward has returned; a caller whose allocator recycles memory per
 * stream (PyTorch: Tensor.record_stream on an ExternalStream of this handle) uses it to keep those buffers from being handed
 * out again before the launch is through, instead of holding references until the join.  (ABI v7) */
int uniter_encoder_side_stream(void** stream_out);

/* Autotune the 12 GEMM shapes (4 forward, 4 dgrad, 4 wgrad) of one BertLayer for this (B, L, H, I): synchronous,
 * call once per shape at set-up time (the Python side does it on the first forward of a new shape). */
int uniter_encoder_autotune(const UniterEncoderShape* s, void* stream);

/* Overlapped kernel chains (ABI v7).  With a scratch buffer, uniter_encoder_forward and (in the deferred-weight-gradient flow)
 * uniter_encoder_backward dispatch the dependent kernels of all their layers WITHOUT the queue barrier between them
 * (hipExtLaunchKernel, hipExtAnyOrderLaunch) and order them through per-32-row flags in the scratch buffer instead: a consumer
 * tile starts when the row block it reads is complete, not when the slowest tile of the producer has drained
 * (csrc/common.cuh "Overlapped kernel chains", EXPERIMENTS.md section 10).  Same kernels, same arithmetic: results are bit-identical
 * to the in-order launches.  Chains are OFF by default (measured neutral at 32 x 96 tokens, EXPERIMENTS.md section 10.2); the
 * switch is a test hook (include/uniter_hip_test.h).  A flag wait that does not complete within 50 ms gives up and sets a status word
 * instead of hanging; uniter_encoder_chain_status reads it (synchronising the device): 0 = clean.  (model/model.py:282-292 has no
 * counterpart.) */
int uniter_encoder_chain_status(const UniterEncoderShape* s, const void* scratch, int32_t* status_out);

/* ------------------------------------------------------------------------------------------------
 * Fused multi-tensor AdamW + global gradient norm / clipping.                 optim/adamw.py:40-103
 * ---------------------------------------------------------------------------------------------- */
typedef struct UniterAdamTensor {
    void*    param;      /* bf16 or fp32 (see param_is_bf16), updated in place                         */
    const void* grad;    /* same dtype as param                                                       */
    float*   master;     /* fp32 master copy (required when param is bf16, else NULL)                  */
    float*   exp_avg;    /* fp32                                                                      */
    float*   exp_avg_sq; /* fp32                                                                      */
    int64_t  numel;
    int32_t  group;      /* index into the per-group hyper-parameter arrays                           */
    int32_t  param_is_bf16;
} UniterAdamTensor;

typedef struct UniterAdamGroup {
    float lr, beta1, beta2, eps, weight_decay;
    int32_t correct_bias;
    int32_t step;        /* value of state['step'] AFTER this update (>= 1)                           */
} UniterAdamGroup;

/* Opaque plan: the tensor table lives on the device; build once, reuse every step. */
int uniter_adamw_plan_create(const UniterAdamTensor* tensors, int64_t n_tensors, void** plan_out);
int uniter_adamw_plan_destroy(void* plan);
/* Per-tensor flags of a plan (one byte per tensor, in the order of uniter_adamw_plan_create's table; all zero = the default):
 *   UNITER_ADAM_KEEP_GRAD  uniter_adamw_step_zero / _step_async leave this tensor's gradient as it is — its producer overwrites it in
 *                          the next backward pass (uniter_encoder_set_grad_overwrite): 2 bytes per parameter less written by the update;
 *   UNITER_ADAM_SKIP_NORM  uniter_adamw_grad_norm_ex does not read this tensor's gradient: the caller passes sums of squares that stand
 *                          for it (uniter_encoder_last_grad_sq). */
#define UNITER_ADAM_KEEP_GRAD 1
#define UNITER_ADAM_SKIP_NORM 2
int uniter_adamw_plan_set_flags(void* plan, const uint8_t* flags, int64_t n_tensors);

/* norm_out[0] = sqrt(sum g^2) * grad_scale ; norm_out[1] = clip coefficient
 *   coef = grad_scale * min(1, max_norm / (norm + 1e-6))     (max_norm <= 0: coef = grad_scale)
 * = torch.nn.utils.clip_grad_norm_ as called at pretrain.py:329-331, with the 1/world averaging of
 * the allreduce (utils/distributed.py:35) foldable into grad_scale.  No host synchronisation. */
int uniter_adamw_grad_norm(void* plan, float grad_scale, float max_norm, float* norm_out, void* stream);
/* The same with `n_extra` partial sums of squares (device floats) added to the total in place of the tensors flagged
 * UNITER_ADAM_SKIP_NORM, whose gradients are then not read: the weights' share of pretrain.py:329-331's norm comes out of the launch
 * that produced those gradients (uniter_encoder_last_grad_sq) instead of a second pass over 170 MB.  n_extra == 0: uniter_adamw_grad_norm. */
int uniter_adamw_grad_norm_ex(void* plan, float grad_scale, float max_norm, float* norm_out, const float* extra, int32_t n_extra,
                              void* stream);

/* One AdamW update of every tensor of the plan.  clip_coef (device pointer, may be NULL = 1.0)
 * multiplies every gradient element on the fly (fused clipping). */
int uniter_adamw_step(void* plan, const UniterAdamGroup* groups, int32_t n_groups,
                      const float* clip_coef, void* stream);

/* uniter_adamw_step that also zeroes every gradient element once it has been read: optimizer.zero_grad()
 * (pretrain.py:334) folded into the update — one pass over the gradients and one launch fewer. */
int uniter_adamw_step_zero(void* plan, const UniterAdamGroup* groups, int32_t n_groups,
                           const float* clip_coef, void* stream);

/* The same update, asynchronous and segmented, so that the next forward pass overlaps it: the work is cut at the given
 * parameter addresses (ascending, e.g. the first parameter of every BertLayer of a flat arena) and the segments run in
 * ascending address order on an internal stream that first waits for everything enqueued on `stream`.  With zero_grads
 * != 0 each gradient element is zeroed once read (optimizer.zero_grad(), pretrain.py:334, folded in).  Afterwards
 *   uniter_params_wait(addr, s)   makes stream s wait until the segment holding parameter address `addr` is updated
 *                                 (no-op for other addresses / when nothing is pending); uniter_encoder_forward calls it
 *                                 for every layer, the embedding entry points for their tables;
 *   uniter_params_wait_all(s)     makes s wait for the whole step (call before gradients are written again or any
 *                                 parameter is read by code that does not use the per-address form). */
int uniter_adamw_step_async(void* plan, const UniterAdamGroup* groups, int32_t n_groups, const float* clip_coef,
                            const void* const* bounds, int32_t n_bounds, int32_t zero_grads, void* stream);
int uniter_params_wait(const void* addr, void* stream);
int uniter_params_wait_all(void* stream);

/* Same update with the per-group hyper-parameters read from DEVICE memory: dev_hyper = n_groups x 6 floats
 * {lr, beta1, beta2, eps, weight_decay, step_size}, step_size = lr*sqrt(1-b2^t)/(1-b1^t) (or lr without bias
 * correction).  Lets a captured hipGraph follow the LR schedule: the host refreshes a pinned buffer, a captured
 * H2D copy moves it before the kernel. */
int uniter_adamw_step_dev(void* plan, const float* dev_hyper, int32_t n_groups, const float* clip_coef, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense pieces of the NLVR2 paired-attention head — model/nlvr2.py:150-153,196-204.
 *   uniter_gemm_bias_relu_dropout_fwd   y = dropout(relu(x w^T + b))  (self.fc = Linear(2H,H) + ReLU + Dropout), one GEMM
 *   uniter_relu_dropout_bwd             dpre = dy * scale where y > 0 else 0  (y is the saved output: it is positive
 *                                       exactly where the unit is active and kept)
 *   uniter_cls_ce_fwd / _bwd            Linear(D, C <= 8) + F.cross_entropy(reduction='none') over n rows; logits are
 *                                       rounded to bf16 like the module's; probs [n,C] fp32 saved for backward; bwd writes
 *                                       dx [n,D] and ACCUMULATES into gw [C,D] / gb [C] (bf16).  targets int64 [n]. */
/* uniter_nlvr2_pair_masks: the masks of the paired head from attn_masks [2 n_pairs rows in (pair, side) order, L] in one launch
 * (model/nlvr2.py:172-176,183-186): pad [2n, L] uint8 = (mask == 0) with rows regrouped [left block; right block], and
 * partner_bias [2n, L] fp32 = (1 - m) * -10000 of the OTHER image of the row's pair (the key mask of the cross attention). */
int uniter_nlvr2_pair_masks(const int64_t* attn_masks, uint8_t* pad, float* partner_bias, int64_t n_pairs, int64_t L, void* stream);
int uniter_gemm_bias_relu_dropout_fwd(const void* x, const void* w, const void* bias, void* y, int64_t M, int64_t N, int64_t K,
                                      float p_drop, uint64_t seed, uint64_t offset, void* stream);
int uniter_relu_dropout_bwd(const void* dy, const void* out, void* dpre, int64_t numel, float p_drop, void* stream);
int uniter_cls_ce_fwd(const void* x, const void* w, const void* b, const int64_t* target, float* loss, float* probs, float* logits,
                      int64_t n, int64_t D, int64_t C, void* stream);
int uniter_cls_ce_bwd(const void* x, const void* w, const float* probs, const int64_t* target, const float* gloss, void* dx,
                      void* gw, void* gb, int64_t n, int64_t D, int64_t C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Attention pooling of the NLVR2 paired-attention head — model/nlvr2.py:110-125 (AttentionPool):
 *   score_t = relu(x_t . w + b) - 1e4 * pad_t ; p = dropout(softmax_t(score)) ; out[B,H] = sum_t p_t x_t
 * x [B,L,H] bf16, pad [B,L] uint8/bool (1 = padded, may be NULL), w [H] bf16 (= fc.0.weight[0]), b [1] bf16.
 * Saved for backward (fp32 [B,L] each): raw = x_t.w + b, sm = softmax, pw = softmax * dropout multiplier.
 * Backward: dx [B,L,H] ; dw [H], db [1] are accumulated into (+=, bf16; either may be NULL).  L <= 256, H <= 1024.
 * ---------------------------------------------------------------------------------------------- */
size_t uniter_attn_pool_workspace_bytes(int64_t B, int64_t H);
int uniter_attn_pool_fwd(const void* x, const uint8_t* pad, const void* w, const void* b, void* out,
                         float* raw, float* sm, float* pw, int64_t B, int64_t L, int64_t H,
                         float p_drop, uint64_t seed, uint64_t offset, void* stream);
int uniter_attn_pool_bwd(const void* x, const void* w, const float* raw, const float* sm, const float* pw,
                         const void* dout, void* dx, void* dw, void* db, int64_t B, int64_t L, int64_t H,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Word-region alignment loss: IPOT optimal-transport distance.
 * Replaces model/ot.py:11-85 (cost_matrix_cosine, ipot, trace) and the un-compaction scatter of
 * model/pretrain.py:166-181 — one workgroup per example instead of ~8 PyTorch launches x 50 iterations.
 *   seq [B,L,H] bf16: the encoder output (compact [txt_i ; img_i ; pad] rows);
 *   scatter [B,L] int64: destination slot of every row (data/itm.py:128-135: text slots 0..tl-1, image slots
 *     tl..tl+il-1; other destinations are ignored);  txt_pad [B,tl], img_pad [B,il]: uint8 / bool, 1 = padded slot;
 *   dist [B] fp32 = trace(C T) with C the cosine cost (0 at padded pairs) and T the IPOT plan after `iterations`
 *     outer and `k` inner iterations with temperature beta (reference defaults 0.5, 50, 1);
 *   plan [B,il,tl] fp32 = T (no gradient flows through it, as in the reference), kept for backward.
 * Backward: dseq [B,L,H] bf16 = d(sum_b gdist[b] * dist[b]) / d seq, every row written (zeros where the row is not a
 * non-padded text / image slot).  Needs H % 8 == 0, H <= 1024 and (2 tl il + 6 tl + 3 il) * 4 bytes <= 160 KiB of LDS.
 * ---------------------------------------------------------------------------------------------- */
int uniter_ot_fwd(const void* seq, const int64_t* scatter, const uint8_t* txt_pad, const uint8_t* img_pad,
                  float* dist, float* plan, int64_t B, int64_t L, int64_t H, int64_t tl, int64_t il,
                  float beta, int32_t iterations, int32_t k, void* stream);
int uniter_ot_bwd(const void* seq, const int64_t* scatter, const uint8_t* txt_pad, const uint8_t* img_pad,
                  const float* plan, const float* gdist, void* dseq,
                  int64_t B, int64_t L, int64_t H, int64_t tl, int64_t il, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RCCL communicator (one process per GPU).                              utils/distributed.py:16-209
 * The Python launcher exchanges the 128-byte unique id (rank 0 creates it, broadcasts it through the
 * torch.distributed store) and each rank calls uniter_comm_init.
 * ---------------------------------------------------------------------------------------------- */
int uniter_comm_unique_id(uint8_t id_out[128]);
int uniter_comm_init(const uint8_t id[128], int32_t rank, int32_t world, void** comm_out);
int uniter_comm_destroy(void* comm);
/* In-place sum-allreduce of a bf16 (dtype=0) or fp32 (dtype=1) buffer, then multiply by `scale`
 * (1/world for Horovod's average, utils/distributed.py:35-36) as part of the consumer (see adamw). */
int uniter_comm_allreduce(void* comm, void* buf, int64_t count, int32_t dtype, void* stream);
int uniter_comm_broadcast(void* comm, void* buf, int64_t bytes, int32_t root, void* stream);
int uniter_comm_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNITER_HIP_H */
