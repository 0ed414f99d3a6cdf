/*
 * s2t_hip.h -- C ABI of libs2t_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * speech-translation forward/backward hot path of FBK-fairseq-ST (`conv_transformer`).
 *
 * The reference has NO native interface on this path: its boundary is Python (fairseq's
 * register_model / register_task / register_criterion plug-in surface) and every
 * FLOP is an ATen call.  Each entry point below therefore names the reference *call site* whose
 * arithmetic it replaces (file:line under the reference tree).  The Python side of the boundary
 * (fbk_fairseq_st_amd/*.py) mirrors the reference's module interface and binds these symbols with
 * ctypes; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless noted; no ownership
 *     transfer; outputs are caller-allocated; calls are stream-ordered on `stream` (a hipStream_t,
 *     NULL = default stream) and re-entrant per stream; no call synchronises the device.  Internal scratch
 *     (workgroup partial sums of s2t_lsce / s2t_kd_loss / s2t_grad_norm_clip / s2t_conv1_bwd_bn, the cached work
 *     tables of s2t_wgrad_group) is keyed by (current device, stream) under a mutex: concurrent calls of one entry
 *     point on different streams, devices or host threads do not share it.
 *   - dtype: S2T_F32 = 0 (parity path, exact-f32 MFMA), S2T_BF16 = 1 (bf16 storage, f32 accumulate).
 *   - return 0 on success; -22 (EINVAL) bad argument; -95 (ENOTSUP) unsupported shape/dtype;
 *     -(1000 + hipError_t) when the HIP runtime reported an error at launch.
 *   - time-major activations: X[t][b][:] row index t*B + b (the reference's T x B x C).
 */
#ifndef S2T_HIP_H
#define S2T_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define S2T_F32 0
#define S2T_BF16 1

/* epilogue selector of s2t_gemm */
#define S2T_ACT_NONE 0
#define S2T_ACT_RELU 1      /* fairseq/utils.py:390-408 ("relu") */
#define S2T_ACT_GELU 2      /* fairseq/modules/gelu.py:24-25 (erf form, computed in f32); pre-activation -> aux_out */
#define S2T_ACT_RELU_BWD 3  /* out = acc * (aux > 0)            (aux = forward post-activation) */
#define S2T_ACT_GELU_BWD 4  /* out = acc * gelu'(aux)           (aux = forward pre-activation)  */
/* ReLU with a 1-bit record instead of a re-read of the activations in the backward pass (the FFN of transformer_layer.py:171-176: the data
 * gradient of fc2 needs only "was the stored, post-dropout activation > 0").  S2T_ACT_RELU_MASK = S2T_ACT_RELU that also writes the
 * record to aux_out; S2T_ACT_RELU_BWD_MASK = S2T_ACT_RELU_BWD with aux = that record.  The record is s2t_gemm_relu_mask_bytes(M, N, K)
 * bytes in the tile / lane order of the 256-wide kernel (opaque: valid only between two products of the same M, N); products that kernel
 * does not take get S2T_ENOTSUP with these codes -- ask s2t_gemm_relu_mask_bytes first (0 = use S2T_ACT_RELU / S2T_ACT_RELU_BWD). */
#define S2T_ACT_RELU_MASK 5
#define S2T_ACT_RELU_BWD_MASK 6

/* ---- library info ------------------------------------------------------------------------- */
/* Weight gradient of the second subsampling convolution (nn.Conv2d(C, C, 3, stride 2, padding 1), conv_transformer.py:348-354), all
 * nine taps in one pass: gw[co][tap*C + ci] += sum_{t4,b,f4} dpre[t4][b][f4][co] * y1n[b][2 t4+kh-1][2 f4+kw-1][ci].
 * bf16, C = 64 (S2T_ENOTSUP otherwise: use the gathered s2t_gemm_gather products). */
int s2t_conv2_wgrad(int dtype, const void* dpre, const void* y1n, float* gw, int B, int T2, int F2, int C, void* stream);
/* Forward of the same convolution as a direct kernel (input rows staged once in LDS, no row maps):
 * z2[t4][b][f4][co] = act(bias[co] + sum y1n[b][2 t4+kh-1][2 f4+kw-1][ci] * w2p[co][(kh*3+kw)*C + ci]), act = S2T_ACT_RELU | S2T_ACT_GELU (GELU also
 * writes the pre-activation to `pre`).  w2p = s2t_permute_conv_w(mode 0).  bf16, C = 64, F2 <= 48 (S2T_ENOTSUP otherwise: s2t_gemm_gather). */
int s2t_conv2_fwd(int dtype, const void* y1n, const void* w2p, const float* bias, void* z2, void* pre, int B, int T2, int F2, int C,
                  int act, void* stream);
/* Data gradient of the same convolution, all four pixel-parity classes in one launch, with the dropout mask of y1n on the way out:
 * dy1n[b][t2][f2][ci] = dropout(sum_{taps reaching (t2, f2)} sum_co dpre[t4][b][f4][co] * w[co][ci][kh][kw], p_drop, seed) (mask indexed by the
 * element's position in dy1n).  w2q = s2t_permute_conv_w(mode 1).  bf16, C = 64, F2 <= 47 (S2T_ENOTSUP otherwise). */
int s2t_conv2_dgrad(int dtype, const void* dpre, const void* w2q, void* dy1n, int B, int T2, int F2, int C, float p_drop,
                    unsigned long long seed, void* stream);
/* The K largest logits of every row, descending, with their columns (scripts/generate_topk.py:64-66: the teacher dump of word-level
 * knowledge distillation).  x [rows][V] with row stride ld; vals f32 [rows][K], idx i32 [rows][K]. */
int s2t_topk(int dtype, const void* x, float* vals, int* idx, long rows, int V, int ld, int K, void* stream);
/* TimeStretch + SpecAugment in one pass (examples/speech_recognition/modules/time_stretch.py:18-57, specaugment.py:44-112; applied by
 * tasks/speech_recognition.py:254-258): out[b][t][:] = x[b][row_map[b][t]][:] (row_map NULL = identity, -1 = zero row), zeroed inside
 * tmask[b][i] = (t0, width) and fmask[b][i] = (f0, width).  x [B][T][F] f32, out [B][To][F] f32 (out != x). */
int s2t_augment(const float* x, float* out, const int* row_map, const int* fmask, const int* tmask, int B, int T, int To, int F,
                int nF, int nT, void* stream);
/* Host: frame-budget batching of utterance indices (fairseq/data/data_utils_fast.pyx:16-68 batch_by_size_fast; caller
 * fairseq/data/data_utils.py:200-234).  lens[idx] = frames of utterance idx; out_flat[n], out_offsets[n + 1]. */
int s2t_host_batch_by_size(const long long* indices, long long n, const long long* lens, long long max_tokens,
                           long long max_sentences, int bsz_mult, long long* out_flat, long long* out_offsets,
                           long long* n_batches);
int s2t_abi_version(void);                       /* bumps when a signature OR a workspace contract changes.  8 (round 5): s2t_ctc_loss's la / lb
                                                  * workspaces are B*T*S2T_CTC_ROW(Lmax) floats holding log2 values with per-step offsets
                                                  * (a phase-1 workspace of a version-7 library is unusable by a version-8 phase 2);
                                                  * the "gemm4w" option of s2t_set_option is gone.  9 (round 6): the s2t_decode_* entry points */
const char* s2t_build_info(void);                /* "gfx950 <date> ..." */

/* ---- GEMM with fused epilogue (MFMA) --------------------------------------------------------
 * C[M,N] = act( alpha * op(A)[M,K] . op(B)[K,N] + bias[N] ) + residual[M,N]      (accumulate: C += ...)
 *   trans_a = 0: A is [M][K];  1: A is [K][M]        trans_b = 0: B is [N][K] (nn.Linear weight);  1: B is [K][N]
 *   in_dtype: A, B.   out_dtype: C, residual, aux, aux_out.   bias is always f32.
 *   splitk > 1: K is split over gridDim.z and partial products are added with f32 atomics into C
 *               (C must be f32 and pre-initialised; no bias/act/residual).
 * Replaces: F.linear at fairseq/modules/multihead_attention.py:190-208,356 (q/k/v/out projections),
 *   fairseq/modules/transformer_layer.py:132-134,358-360 (fc1 + activation, fc2 + residual),
 *   examples/speech_recognition/models/conv_transformer.py:227 (fc3 + activation), :279 (ctc_fc),
 *   fairseq/models/transformer.py:784-788 (output projection), and their autograd backward
 *   (dX = dY W: trans_b = 1;  dW = dY^T X: trans_a = trans_b = 1). */
int s2t_gemm(int in_dtype, int out_dtype, int trans_a, int trans_b, int M, int N, int K,
             const void* A, int lda, const void* B, int ldb, void* C, int ldc,
             const float* bias, const void* residual, int ldr, const void* aux, void* aux_out, int ldaux,
             int act, int accumulate, int splitk, float alpha, void* stream);

/* Same kernel with row gather / scatter, used to run Conv2d(64->64, 3x3, stride 2, pad 1)
 * (conv_transformer.py:203-206, 2nd iteration) and its backward as implicit GEMMs over channels-last
 * activations:  mapA[(k/periodA)*M + r] = source row of A for output row r and k-block (tap) k/periodA
 * (-1 = zero padding);  mapB[k] = source row of a [K][N] B operand;  mapC[r] = destination row.
 * p_drop > 0: dropout (Philox(seed, row*N+col)) on the activated value before the residual add
 * (transformer_layer.py:123-124,133-136: x = residual + dropout(linear(...))). */
int s2t_gemm_gather(int in_dtype, int out_dtype, int trans_a, int trans_b, int M, int N, int K,
                    const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                    const float* bias, const void* residual, int ldr, const void* aux, void* aux_out, int ldaux,
                    int act, int accumulate, int splitk, float alpha,
                    const int* mapA, int periodA, const int* mapB, const int* mapC,
                    float p_drop, unsigned long long seed, void* stream);
/* Size in bytes of the 1-bit ReLU record of an [M][N] bf16 product with reduction length K (S2T_ACT_RELU_MASK / S2T_ACT_RELU_BWD_MASK
 * above), or 0 when products of that shape do not run on the kernel that keeps such records. */
size_t s2t_gemm_relu_mask_bytes(int M, int N, int K);

/* Weight (and bias) gradients of MANY Linears in one launch (bf16 operands, f32 gradients; csrc/wgrad_group.hip):
 *     dW_p[n_out][n_in] += dY_p[tokens][n_out]^T X_p[tokens][n_in]      db_p[n_out] += column sums of dY_p   (db may be NULL)
 * Replaces, for the Transformer blocks, the per-Linear autograd products of F.linear (fairseq/modules/multihead_attention.py:190-208,
 * fairseq/modules/transformer_layer.py:132-134): nothing reads a weight gradient before the optimizer / the gradient all-reduce, so
 * the caller may queue the (dY, X) pairs of several layers during backward and submit them together.  Every 256 x 256 tile of every
 * dW is owned by one workgroup for the whole token range: no split-K, no atomics
 * (when the tile count does not fill the last round of one tile per CU, the tiles of that round are cut along the token range and
 * meet in f32 atomics).  Requirements: ldy, ldx multiples of 8 elements and >= the column count rounded up to 8, 16-byte aligned
 * operand bases; a dW must not appear twice in one call.  `probs` is a HOST array (uploaded stream-ordered with the work list). */
typedef struct S2TWgradProblem {
    const void* dY; const void* X; float* dW; float* db;
    int n_out, n_in, tokens, ldy, ldx, ldw;
} S2TWgradProblem;
int s2t_wgrad_group(int n, const S2TWgradProblem* probs, void* stream);
/* The same for f32 operands (the parity mode; csrc/wgrad_f32.hip): exact-f32 MFMA, every 128 x 128 tile of every dW owned by one workgroup
 * over all its tokens (no atomics: results do not depend on scheduling), all tiles of all products dealt to two workgroups per CU,
 * longest reductions first.  Requirements: ldy, ldx multiples of 4 elements and >= the column count rounded up to 4, 16-byte aligned
 * operand bases; a dW must not appear twice in one call. */
int s2t_wgrad_group_f32(int n, const S2TWgradProblem* probs, void* stream);

/* out[n] += sum_m X[m][n]  (bias gradients of every nn.Linear above; f32 atomics) */
/* Parameter gradients of a Linear in one pass over dY (autograd of F.linear: fairseq/modules/multihead_attention.py:190-208,
 * fairseq/modules/transformer_layer.py:132-134): dW[n_out][n_in] += dY[tokens][n_out]^T X[tokens][n_in] (f32) and, if db is not
 * NULL, db[n_out] += column sums of dY. */
int s2t_linear_wgrad(int in_dtype, int n_out, int n_in, int tokens, const void* dY, int ldy, const void* X, int ldx,
                     float* dW, int ldw, float* db, int splitk, void* stream);
int s2t_colsum(int dtype, const void* X, int ld, int M, int N, float* out, void* stream);

/* ---- fused multi-head attention ---------------------------------------------------------------
 * Replaces fairseq/modules/multihead_attention.py:155-177 (F.multi_head_attention_forward) and
 * :316-355 (own bmm/softmax/bmm path): scores, key-padding (-inf), causal mask
 * (fairseq/models/transformer.py:798-810), fp32 softmax, dropout on P, P.V.  Scores never reach HBM.
 * X[t][b][h][j] = X + t*x_st + b*x_sb + h*head_dim + j (element strides);  head_dim in {32, 64}.
 * klen[b] = number of valid (non-padded) keys of batch b, NULL = all Tk (padding is a suffix on this
 * path: conv_transformer.py:293-300, transformer.py:739-741).  LSE [B][H][Tq] f32 (saved for backward). */
int s2t_attn_fwd(int dtype, int head_dim, int B, int H, int Tq, int Tk,
                 const void* Q, long q_st, long q_sb, const void* K, long k_st, long k_sb,
                 const void* V, long v_st, long v_sb, void* O, long o_st, long o_sb, float* LSE,
                 const int* klen, int causal, int dist_penalty, float scale, float p_drop, unsigned long long seed, void* stream);
/* Head-averaged attention probabilities of ONE layer, recomputed from its q and k (the fused kernels never materialise P):
 * out[b][tq][tk] = mean over the first heads_used heads of softmax_tk(scale * q . k), 0 past klen[b]; f32 [B][Tq][Tk], Tk <= 2048.
 * Replaces the need_attn / need_head_weights return of fairseq/modules/multihead_attention.py:342-355 as consumed by
 * fairseq/models/transformer.py:756-782 (alignment_layer / alignment_heads; generate.py --print-alignment, ensemble averaging). */
int s2t_attn_probs_avg(int dtype, int head_dim, int B, int H, int Tq, int Tk, const void* Q, long q_st, long q_sb,
                       const void* K, long k_st, long k_sb, const int* klen, int heads_used, float scale, float* out, void* stream);
/* Backward of s2t_attn_fwd (flash-style recomputation; Delta [B][H][Tq] f32 is workspace). */
int s2t_attn_bwd(int dtype, int head_dim, int B, int H, int Tq, int Tk,
                 const void* Q, long q_st, long q_sb, const void* K, long k_st, long k_sb,
                 const void* V, long v_st, long v_sb, const void* O, long o_st, long o_sb,
                 const void* dO, long do_st, long do_sb, const float* LSE, float* Delta,
                 void* dQ, long dq_st, long dq_sb, void* dK, long dk_st, long dk_sb, void* dV, long dv_st, long dv_sb,
                 const int* klen, int causal, int dist_penalty, float scale, float p_drop, unsigned long long seed, void* stream);

/* ---- LayerNorm (fairseq/modules/layer_norm.py:29-32; eps 1e-5; rows of D <= 1024) ------------------ */
int s2t_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y,
                      float* mean, float* rstd, int M, int D, float eps, void* stream);
/* dx = LN'(dy) + dres (dres optional: the residual branch of a pre-LN block); dgamma/dbeta += (f32).
 * dx_drop (optional): a second output = s2t_dropout(dx, p_drop, seed), i.e. the gradient that enters the
 * residual-branch dropout of the block that consumes dx, written in the same pass (bit-identical to the
 * separate call). */
int s2t_layernorm_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* rstd,
                      const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta,
                      int M, int D, void* dx_drop, float p_drop, unsigned long long seed, void* stream);

/* ---- one Transformer layer per call ------------------------------------------------------------------
 * fairseq/modules/transformer_layer.py:87-139 (encoder layer) and :243-377 (decoder layer), pre-LN form, bf16 path: the kernels of
 * a layer in the order the host engine issues them (LayerNorm, projections with fused bias / residual / dropout epilogues, flash
 * attention, FFN with the 1-bit ReLU record or the GELU pre-activation), behind ONE entry point per direction.  Nothing new is
 * computed here -- every launch is one of the entry points above -- but a small batch (8 utterances per GPU: SURVEY.md 8-d) is
 * bound by the HOST's launch rate, and a layer is 7-13 launches forward and 9-20 backward.
 * S2TLayerDesc: shapes, rates and parameter pointers of one layer (fixed per model and batch shape); S2TLayerCall: the operands of
 * one call.  Activations are time-major rows (t*B + b) of D elements; `ws` keeps what the backward needs (s2t_layer_ws_bytes),
 * `tmp` is scratch of the backward (s2t_layer_bwd_tmp_bytes; it holds the dY operands of the layer's weight-gradient products,
 * which s2t_layer_bwd does NOT launch: it appends them to `items` for a later s2t_wgrad_group, so `ws`, `tmp`, `x` and `enc` must
 * stay alive until then).  Dropout seeds are per site, exactly the ones the per-kernel path passes. */
typedef struct S2TLayerDesc {
    int dtype, decoder;                 /* S2T_BF16; 0 = self-attention + FFN, 1 = + encoder attention in between */
    int T, B, D, heads, ffn, Ts;        /* Ts: source positions (decoder) */
    int gelu, causal, dist_penalty;     /* activation (0 ReLU); mask / distance penalty of the self-attention */
    float ln_eps, p_drop, p_attn, p_act;
    const void *w_qkv, *w_o, *w_xq, *w_xkv, *w_xo, *w_fc1, *w_fc2;                  /* [out][in], compute dtype */
    const float *b_qkv, *b_o, *b_xq, *b_xkv, *b_xo, *b_fc1, *b_fc2;
    const float *ln1_g, *ln1_b, *lnx_g, *lnx_b, *ln2_g, *ln2_b;
    float *g_w_qkv, *g_w_o, *g_w_xq, *g_w_xkv, *g_w_xo, *g_w_fc1, *g_w_fc2;         /* f32 gradient slices (backward) */
    float *g_b_qkv, *g_b_o, *g_b_xq, *g_b_xkv, *g_b_xo, *g_b_fc1, *g_b_fc2;
    float *g_ln1_g, *g_ln1_b, *g_lnx_g, *g_lnx_b, *g_ln2_g, *g_ln2_b;
} S2TLayerDesc;
typedef struct S2TLayerCall {
    int training;
    const int* self_klen; const int* enc_klen;            /* key lengths [B] or NULL */
    unsigned long long seed_sa_attn, seed_sa_out, seed_xa_attn, seed_xa_out, seed_ffn_act, seed_ffn_out;
    const void* x; const void* enc; void* y; void* ws;    /* forward: y = layer(x [T*B][D], enc [Ts*B][D]) */
    /* backward: dy = gradient w.r.t. y; dy_drop = dropout(dy) with the FFN's output mask when the producer of dy already wrote it
     * (else NULL); dx, and when nxt_p > 0 also dx_drop = dropout(dx, nxt_p, nxt_seed) for the consumer of dx; the encoder-output
     * gradient is written (denc_accumulate = 0) or added to denc [Ts*B][D] */
    const void* dy; const void* dy_drop; void* dx; void* dx_drop; float nxt_p; unsigned long long nxt_seed;
    void* denc; int denc_accumulate; void* tmp;
    S2TWgradProblem* items; int max_items; int n_items;   /* host array; n_items is advanced by the products appended */
} S2TLayerCall;
size_t s2t_layer_ws_bytes(const S2TLayerDesc* L, int training);
size_t s2t_layer_bwd_tmp_bytes(const S2TLayerDesc* L);
int s2t_layer_fwd(const S2TLayerDesc* L, S2TLayerCall* c, void* stream);
int s2t_layer_bwd(const S2TLayerDesc* L, S2TLayerCall* c, void* stream);

/* ---- convolutional subsampler (conv_transformer.py:202-232) ----------------------------------------
 * conv1: x [B][T][F] f32 -> y [B][T2][F2][C] (channels-last) = act(conv3x3 s2 p1 + bias), act = S2T_ACT_RELU | S2T_ACT_GELU
 * (--activation-fn, conv_transformer.py:140-142,212); with GELU the pre-activation goes to `pre` (same shape, needed by the
 * backward; NULL for ReLU).  Also the BatchNorm sums: sums[c] += y, sums[C+c] += y^2 (double, caller zeroes).  C in {32, 64, 128}. */
int s2t_conv1_fwd(int dtype, const float* x, const float* w, const float* bias, void* y, void* pre, double* sums,
                  int B, int T, int F, int C, int act, void* stream);
/* dw[c][3][3] += , db[c] += from dpre [B][T2][F2][C] (gradient after the ReLU mask) */
int s2t_conv1_bwd(int dtype, const float* x, const void* dpre, float* dw, float* db, int B, int T, int F, int C, void* stream);
/* s2t_bn_bwd_apply + s2t_conv1_bwd in one pass (the gradient w.r.t. the convolution's output is never written): dw / db += as
 * s2t_conv1_bwd would from dpre = act'(.) * BatchNorm'(dyn) (arguments as s2t_bn_bwd_apply: y = BatchNorm input, pre = pre-activation
 * for GELU or NULL, sums from s2t_chan_sums(mode 1)); dgamma / dbeta += */
int s2t_conv1_bwd_bn(int dtype, const float* x, const void* dyn, const void* y, const void* pre, const float* mean, const float* rstd,
                     const float* gamma, const double* sums, float* dw, float* db, float* dgamma, float* dbeta, int B, int T, int F,
                     int C, double count, int training, void* stream);
/* per-channel double sums over a channels-last [P][C] tensor: mode 0 (y, y^2); mode 1 (dyn, dyn*xhat) */
int s2t_chan_sums(int dtype, const void* y, const void* dyn, const float* mean, const float* rstd,
                  double* sums, long P, int C, int mode, void* stream);
/* nn.BatchNorm2d statistics (conv_transformer.py:212,364-368): training -> batch stats from sums, running
 * stats momentum update (unbiased var), num_batches += 1; eval -> running stats.  scale/shift for s2t_bn_apply. */
int s2t_bn_finalize(const double* sums, const float* gamma, const float* beta, float* run_mean, float* run_var,
                    long long* num_batches, float* mean, float* rstd, float* scale, float* shift,
                    double count, int C, int training, float momentum, float eps, void* stream);
/* yn = dropout(y*scale + shift, p_drop, seed): the normalisation and the dropout that follows it (conv_transformer.py:212-214)
 * in one pass; the mask / rounding are those of s2t_dropout on the normalised tensor */
int s2t_bn_apply(int dtype, const void* y, const float* scale, const float* shift, void* yn, long n, int C,
                 float p_drop, unsigned long long seed, void* stream);
/* BatchNorm backward fused with the derivative of the activation in front of it: the ReLU mask (y > 0) when pre == NULL, else
 * gelu'(pre); sums from s2t_chan_sums(mode 1); dgamma/dbeta += */
int s2t_bn_bwd_apply(int dtype, const void* dyn, const void* y, const void* pre, const float* mean, const float* rstd,
                     const float* gamma, const double* sums, void* dpre, float* dgamma, float* dbeta,
                     long n, int C, double count, int training, void* stream);
/* fc3 weight [N][C*F] (k = c*F+f, conv_transformer.py:225-226) <-> channels-last k' = f*C+c */
int s2t_permute_cf(int dst_dtype, const float* src, void* dst, int N, int C, int F, int mode, void* stream);
/* conv2 weight [Co][Ci][3][3] <-> implicit-GEMM operand layouts (see subsample.hip): 0 forward, 1 data gradient by parity
 * class, 2 weight gradient back to the master layout */
int s2t_permute_conv_w(int dst_dtype, const float* src, void* dst, int Co, int Ci, int mode, void* stream);
/* dst[t][b][:] = dropout(src[t][b][:] + sinusoid[(t < len[b]) ? t+1 : 0][:])  (positional_embedding_audio.py:21-27; the add and the
 * F.dropout of conv_transformer.py:229-232 in one pass; src may equal dst; p_drop = 0: no dropout; mask = s2t_dropout's on the flat index) */
int s2t_add_pos(int dtype, const void* src, void* dst, const float* table, const int* len, int T, int B, int D,
                float p_drop, unsigned long long seed, void* stream);

/* ---- CTC compression (conv_transformer.py:278-291, 385-426) -----------------------------------------
 * pred[b][t] = first arg-max of softmax(logits[t][b][:]) (bit-exact integer path), pmax = its probability.
 * Logit rows have a stride of ld >= V elements (here and in the two loss kernels): the producer GEMM pads the
 * row stride to a multiple of 8 so that every GEMM touching the logits can use 16-byte loads. */
/* lse (optional, [T*B] f32): the rows' log-sum-exps, for s2t_ctc_loss over the same logits (phase | 4) */
int s2t_ctc_argmax(int dtype, const void* logits, int* pred, float* pmax, float* lse, int T, int B, int V, int ld, void* stream);
/* run-length collapse inside len[b]; seg/run_start/run_len [B][T] int32, new_len [B] int64, w [B][T] f32
 * strategy 0 avg, 1 weighted, 2 softmax */
int s2t_ctc_rle(const int* pred, const float* pmax, const long long* len, int* seg, int* run_start, int* run_len,
                long long* new_len, float* w, int T, int B, int strategy, void* stream);
/* out[j][b][:] = sum_{t in run j} w[b][t] x[t][b][:]  (zeros for new_len[b] <= j < Tout) */
int s2t_ctc_compress_fwd(int dtype, const void* x, const float* w, const int* run_start, const int* run_len,
                         const long long* new_len, void* out, int T, int B, int D, int Tout, void* stream);
int s2t_ctc_compress_bwd(int dtype, const void* dout, const float* w, const int* seg, void* dx, int T, int B, int D,
                         int accumulate, void* stream);

/* ---- losses ------------------------------------------------------------------------------------
 * F.log_softmax + F.ctc_loss(reduction="sum", zero_infinity=True) (CTC_loss.py:143-151) and its gradient
 * w.r.t. the logits [T][B][V].  Workspaces: lse [T*B], la/lb [B*T*S2T_CTC_ROW(Lmax)], nll [B] (all f32; la/lb rows are padded
 * to the positions of the recursion's wave and hold log2 values shifted per frame: private to the two passes).
 * loss_sum[0] += sum_b nll_b (caller zeroes). grad is multiplied by grad_scale and, if given, by the device scalar
 * grad_scale_dev[0] (the upstream gradient autograd hands to backward).  phase 0: loss and gradient in one call;
 * phase 1: loss only (workspaces kept by the caller); phase 2: the gradient from the workspaces of a phase-1 call;
 * phase | 4: `lse` already holds the row log-sum-exps of these logits (written by s2t_ctc_argmax), skip that pass.
 * Limits (the reference's F.ctc_loss has none): transcripts of at most S2T_CTC_MAX_TARGET units (Lmax, the padded width of
 * `targets`) and vocabularies of at most S2T_CTC_MAX_VOCAB entries; beyond them the call returns -95 (ENOTSUP) and the host side
 * (criterions.py) refuses the batch with the limit in the message. */
#define S2T_CTC_MAX_TARGET 511
#define S2T_CTC_MAX_VOCAB 40704
/* row stride of the la / lb workspaces: the 2*Lmax+1 extended-target positions rounded up to 64 x {1, 2, 4, 8, 16} */
#define S2T_CTC_ROW(Lmax) (2 * (Lmax) + 1 <= 64 ? 64 : 2 * (Lmax) + 1 <= 128 ? 128 : 2 * (Lmax) + 1 <= 256 ? 256 : 2 * (Lmax) + 1 <= 512 ? 512 : 1024)
int s2t_ctc_loss(int dtype, const void* logits, const long long* targets, const long long* tgt_len, const int* in_len,
                 float* lse, float* la, float* lb, float* nll, void* grad, float* loss_sum,
                 int T, int B, int V, int ld, int Lmax, int blank, float grad_scale, int phase, const float* grad_scale_dev,
                 void* stream);
/* label_smoothed_nll_loss over log_softmax(logits.float()) (label_smoothed_cross_entropy.py:12-29), fused
 * with its gradient: sums2[0] += loss, sums2[1] += nll (caller zeroes); dlogits may be NULL. */
int s2t_lsce(int dtype, const void* logits, const long long* target, void* dlogits, float* sums2,
             long rows, int V, int ld, float eps, int pad, float grad_scale, void* stream);

/* word-level knowledge distillation from stored teacher top-K logits (fairseq/criterions/knowledge_distillation.py:44-96):
 * sum1[0] += sum_rows (1-lambda)*NLL + lambda*KD(tau); teacher_idx [rows][Kt] int64, teacher_logits [rows][Kt] f32, Kt <= 64 */
int s2t_kd_loss(int dtype, const void* logits, const long long* target, const long long* teacher_idx,
                const float* teacher_logits, void* dlogits, float* sum1, long rows, int V, int ld, int Kt,
                float lambda, float tau, int pad, float grad_scale, void* stream);

/* ---- decoder embedding (fairseq/models/transformer.py:720-737) -------------------------------------- */
int s2t_embed_fwd(int dtype, const long long* tokens, const void* W, const float* table, void* out,
                  int B, int L, int D, float scale, int pad, int pos_offset, void* stream);
/* f32 log-probabilities of one decoding step (fairseq/sequence_generator.py:711-768) */
int s2t_log_softmax(int dtype, const void* logits, float* out, long rows, int V, int ld, float inv_temperature, void* stream);
/* The model's get_normalized_probs for criteria that work on (log-)probabilities themselves (fairseq/models/fairseq_decoder.py:58-79,
 * fairseq/models/fairseq_model.py:46-74: utils.log_softmax / utils.softmax of logits.float(); the three criteria of the S2T recipes use
 * the fused loss kernels instead).  s2t_softmax_probs: out = softmax(logits * inv_temperature) in f32.  s2t_softmax_bwd: the gradient of
 * either form w.r.t. the logits, from the saved OUTPUT `out` and its gradient `dout` (both f32 [rows][V], dense):
 *   log_probs = 1:  dlogits = it * (dout - exp(out) * sum_v dout)        log_probs = 0:  dlogits = it * out * (dout - sum_v dout * out)
 * written in `dtype` with row stride ld. */
int s2t_softmax_probs(int dtype, const void* logits, float* out, long rows, int V, int ld, float inv_temperature, void* stream);
int s2t_softmax_bwd(int dtype, const float* out, const float* dout, void* dlogits, long rows, int V, int ld, float inv_temperature,
                    int log_probs, void* stream);
/* ensemble of n <= 8 models (fairseq/sequence_generator.py:757-768 EnsembleModel.forward_decoder): out = logsumexp_j lprobs[j] - log n,
 * element-wise over numel f32 values; `lprobs` is a HOST array of n device pointers */
int s2t_ensemble_lse(int n, const float* const* lprobs, float* out, size_t numel, void* stream);
int s2t_embed_bwd(int dtype, const long long* tokens, const void* dout, float* dW, int B, int L, int D,
                  float scale, int pad, void* stream);
/* out = dropout(dy) * act'(y): act 1 = relu (y = post-activation), act 2 = gelu (y = pre-activation); the dropout (p_drop > 0, mask
 * of s2t_dropout on the flat index) is the backward of one that followed the activation in the forward pass */
int s2t_act_bwd(int dtype, const void* dy, const void* y, void* out, size_t n, int act, float p_drop, unsigned long long seed,
                void* stream);
/* y += x (merging the gradients of two consumers of one activation) */
int s2t_add_inplace(int dtype, const void* x, void* y, size_t n, void* stream);
/* y = x * keep/(1-p), mask from Philox(seed, index); the backward pass calls it again on the gradient */
int s2t_dropout(int dtype, const void* x, void* y, size_t n, float p, unsigned long long seed, void* stream);

/* ---- ConvAttention2D blocks of the front end (examples/speech_recognition/modules/conv_attention_2d.py:46-135,
 * conv_transformer.py:216-222).  Channels-last tensors over pixel rows r = (t*B + b)*F + f:
 * x [M][C], qkv [M][16] (q 0-3 | k 4-7 | v 8-11 | 4 zero channels), cat [M][8] (time 0-3 | frequency 4-7).
 * The two 3x3 convolutions are s2t_gemm_gather calls on the packed weights of s2t_a2d_pack_w. ------------------------- */
/* channel moments (mode 0: sum z', z'^2 with z' = prescale[c]*z; mode 1: sum dyn, dyn*xhat with dyn = dy*[BN(z') > 0]),
 * accumulated (double) in groups of Cg channels laid out [Cg | Cg] so that each group feeds s2t_bn_finalize */
int s2t_a2d_chan_stats(int dtype, const void* z, const void* dy, const float* prescale, const float* mean, const float* rstd,
                       const float* scale, const float* shift, double* sums, long M, int C, int ld_z, int ld_dy, int Cg,
                       int mode, void* stream);
/* y = relu(prescale*z*scale + shift) [+ res]   (nn.BatchNorm2d then ReLU, conv_attention_2d.py:92-94,121) */
int s2t_a2d_bn_act(int dtype, const void* z, const float* prescale, const float* scale, const float* shift, const void* res,
                   void* y, long M, int C, int ld_z, int ld_y, void* stream);
/* gradient w.r.t. z of the above given the mode-1 sums (training: batch statistics; eval: running statistics) */
int s2t_a2d_bn_bwd(int dtype, const void* dy, const void* z, const float* prescale, const float* mean, const float* rstd,
                   const float* scale, const float* shift, const double* sums, void* dz, long M, int C, int ld_dy, int ld_z,
                   int Cg, double count, int training, void* stream);
/* dbeta += sums[0..Cg), dgamma += sums[Cg..2Cg) of one group of mode-1 sums */
int s2t_a2d_param_grads(const double* sums, float* dgamma, float* dbeta, int Cg, void* stream);
/* nn.Conv2d weights [Co][Ci][3][3] (f32) <-> gathered-GEMM operands, row stride ld, CP = padded channels per tap:
 * mode 0: dst[co][j*CP+ci] = W[co][ci][j];  mode 1: dst[ci][j*CP+co] = W[co][ci][8-j];  mode 2: grad[co][ci][j] += src[co][j*CP+ci] */
int s2t_a2d_pack_w(int dst_dtype, const float* src, void* dst, float* grad, int Co, int Ci, int CP, int ld, int mode, void* stream);
/* weight gradient of a 3x3 / pad 1 convolution over the pixel rows, all nine taps in one pass, added to the master layout
 * dW [CO][CI][3][3] (f32): (CO <= 16 with ld_dy >= 16, CI = 64) or (CO = 64, CI = 8); other shapes: S2T_ENOTSUP (callers fall
 * back to nine gathered s2t_gemm_gather products).  ws: S2T_A2D_WGRAD_GROUPS x max(CO,16) x CI x 9 floats of workspace
 * (per-workgroup partial sums, reduced by a second kernel: 16 instead of 512 atomics per weight) */
#define S2T_A2D_WGRAD_GROUPS 512
int s2t_a2d_conv_wgrad(int dtype, const void* dY, int ld_dy, const void* X, int ld_x, float* dW, float* ws, int CO, int CI,
                       int B, int T, int F, void* stream);
/* channels-last pixel rows [M][ld] <-> head-major planes [G][T][B][4*32] (plane of head h of group g = channel ch0 + 4g + h,
 * columns 32h .. 32h+F-1, zero padded): dir 0 gathers the planes, dir 1 scatters them back.  The time attention of a block then
 * runs on s2t_attn_fwd / s2t_attn_bwd at head_dim 32 */
int s2t_a2d_planes(int dtype, void* chl, void* planes, int G, int ch0, int ld, int B, int T, int F, int dir, void* stream);
/* time attention of every (batch, head) plane: cat[.., h] = dropout(softmax_t'(q k^T)) v ; lse [B*4][T] for the backward */
int s2t_a2d_time_fwd(int dtype, const void* qkv, void* cat, float* lse, int B, int T, int F, float p_drop,
                     unsigned long long seed, void* stream);
/* writes dq, dk, dv of the time attention into dqkv (delta [B*4][T] is workspace) */
int s2t_a2d_time_bwd(int dtype, const void* qkv, const void* cat, const void* dcat, const float* lse, float* delta, void* dqkv,
                     int B, int T, int F, float p_drop, unsigned long long seed, void* stream);
/* frequency attention: cat[.., 4+h] = (dropout(softmax_f'(q^T k)) v^T)^T ; A [B*4][F][F] = the probabilities before dropout */
int s2t_a2d_freq_fwd(int dtype, const void* qkv, void* cat, float* A, int B, int T, int F, float p_drop,
                     unsigned long long seed, void* stream);
/* ADDS dq, dk, dv of the frequency attention to dqkv */
int s2t_a2d_freq_bwd(int dtype, const void* qkv, const void* dcat, const float* A, void* dqkv, int B, int T, int F,
                     float p_drop, unsigned long long seed, void* stream);

/* ---- optimizer (fairseq/trainer.py:416-443, fairseq/utils.py:253-277, fairseq/optim/adam.py:147-202) ----
 * out2[0] = gnorm = scale*||g||_2 ; out2[1] = scale * min(1, max_norm/(gnorm+1e-6)) (all on device) */
int s2t_grad_norm_clip(const float* g, size_t n, double* acc_ws, float scale, float max_norm, float* out2, void* stream);
/* the same with scale / max(*divisor_dev, 1) in place of scale: divisor_dev = the sample size summed over the data-parallel ranks, still
 * on the device after its all-reduce (fairseq/trainer.py:416-430 reads it on the host; here no host synchronisation precedes Adam) */
int s2t_grad_norm_clip_div(const float* g, size_t n, double* acc_ws, float scale, const double* divisor_dev, float max_norm,
                           float* out2, void* stream);
/* Adam over a flat arena; gradients are multiplied by mult2[1] (NULL = 1); optional bf16 shadow refresh */
int s2t_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, size_t n, const float* mult2,
                  float lr, float beta1, float beta2, float eps, float wd, int step, void* stream);
int s2t_cast(int src_dtype, int dst_dtype, const void* src, void* dst, size_t n, void* stream);
int s2t_scale_by_device_scalar(int dtype, void* x, size_t n, const float* scalar, void* stream);

/* ---- incremental decoding + beam search as ONE stream-ordered sequence of launches per step ----------------------------------------
 * Replaces, per decoding step, fairseq/sequence_generator.py:243-447 (the loop body of SequenceGenerator._generate: forward_decoder,
 * log-softmax, the pad / unk / min-len / max-len rules, BeamSearch.step = fairseq/search.py:55-83, EOS finalisation bookkeeping :502-600,
 * next-beam selection :417-446, reorder_incremental_state :430-447 / fairseq/modules/multihead_attention.py:246-283,407-420) and the
 * incremental TransformerDecoder forward (fairseq/models/transformer.py:674-782, transformer_layer.py:243-377).
 *
 * N = B * beam hypothesis slots, slot n = s * beam + j of sentence s.  One step is 3 * layers + 4 launches:
 *   per layer  S: LayerNorm -> q|k|v of one head (k, v written to the cache row of this step) -> attention over the cached rows of the
 *                 hypothesis' ancestors -> this head's share of the output projection           grid (B, heads)
 *              C: LayerNorm -> q of one head -> attention over the sentence's encoder rows -> share of the output projection
 *              F: LayerNorm -> one slice of fc1 + activation -> that slice's share of fc2        grid (B, ffn_slices)
 *   then       final LayerNorm; output projection over all N rows; per row: log-softmax, score rules, + cumulative score, 2*beam best;
 *              per sentence: merge, finalise EOS candidates, choose the next beam, record it, embed its tokens for the next step.
 * The shares (per head / per slice, in the compute type T) are summed in f32, in a fixed order, by the launch that consumes them, together with the residual and the
 * bias: no atomics, results do not depend on scheduling.  The K/V cache is never re-ordered: anc[n][p] names the slot whose row at
 * position p belongs to hypothesis n's history (the index indirection that replaces reorder_incremental_state's index_select copies),
 * the encoder-side K/V exist once per sentence.  The step index lives in device memory (steps[s]), so one recorded sequence (a hipGraph
 * captured around s2t_decode_step) serves every step.  Nothing synchronises; the host polls `finished` when it chooses to.
 * Limits (S2T_ENOTSUP otherwise): head size 64, D = 256, 512 or 1024, beam <= 16, B * beam <= 128, ffn / ffn_slices = 64, 128 or 256,
 * max_len + 1 <= 1024 positions, Tsp a multiple of 128, and the LDS plan of every launch within 152 KiB (s2t_decode_lds_bytes). */
typedef struct S2TDecodeLayer {
    const void *ln1_g, *ln1_b;       /* f32 [D]: self_attn_layer_norm */
    /* every w_* below is the nn.Linear weight in FRAGMENT-MAJOR order (s2t_decode_pack_weight): [rows / 16][K / ks][64 lanes][16 bytes] */
    const void *w_qkv, *b_qkv;       /* T [3D][D] packed, f32 [3D] */
    const void *w_o, *b_o;           /* T [D][D] packed, f32 [D] */
    const void *lnx_g, *lnx_b;       /* encoder_attn_layer_norm */
    const void *w_xq, *b_xq, *w_xo, *b_xo;
    const void *ln2_g, *ln2_b;       /* final_layer_norm */
    const void *w_fc1, *b_fc1;       /* T [ffn][D] packed, f32 [ffn] */
    const void *w_fc2, *b_fc2;       /* T [D][ffn] packed, f32 [D] */
    const void* kv_enc;              /* T [B][heads][Tsp / 16][64 / ks][64][16 B]: this layer's encoder-side keys (static_kv), fragment-major, one
                                      * copy per SENTENCE, zero beyond Ts (s2t_decode_prepare_enc); Tsp % 128 == 0 */
    const void* vt_enc;              /* T [B][heads][4][Tsp / ks][64][16 B]: the values, transposed and fragment-major (same call) */
    void* kv_cache;                  /* T [max_len + 1][N][2D]: k | v rows written by step t at position t */
} S2TDecodeLayer;

typedef struct S2TDecodeDesc {
    int dtype, B, beam, D, heads, ffn, layers, V, ldv, Ts, Tsp, max_len, min_len, ffn_slices, gelu;
    int pad, unk, eos, step0_all_slots;      /* step0_all_slots: HierarchicalBeamSearch (twophase_sequence_generator.py:22-49) */
    float ln_eps, embed_scale, unk_penalty, inv_temperature;
    const S2TDecodeLayer* layer;             /* HOST array [layers] */
    const void *lnf_g, *lnf_b;               /* decoder.layer_norm */
    const void* w_out;                       /* T [V][D] output projection (the embedding when shared), packed like the layers' weights */
    const void* embed;                       /* T [V][D] */
    const float* pos_table;                  /* f32 [>= pad + 2 + max_len][D] sinusoidal table, row `pad` zero */
    const int* enc_klen;                     /* i32 [B] valid encoder rows per sentence, or NULL (no padding) */
    const float* init_scores;                /* f32 [N] starting score of every slot (step0_all_slots), or NULL */
    /* state, all device memory owned by the caller */
    float *x0, *x1;                          /* f32 [N][D] residual stream (ping-pong) */
    void *part0, *part1;                     /* T [max(heads, ffn_slices)][N][D] shares (ping-pong) */
    void* xn;                                /* T [N][D] final LayerNorm output */
    float* logits;                           /* f32 [N][ldv] */
    int* steps;                              /* i32 [B] */
    int* anc;                                /* i32 [N][max_len + 1] */
    float* cand_val; int* cand_idx;          /* [N][2 * beam] per-row candidates */
    int* tok_hist; int* par_hist; float* cum_hist;   /* [max_len + 2][N]: arrangement i = the beam after i selections */
    int* blacklist;                          /* i32 [N] */
    int* nfin; int* finished;                /* i32 [B] */
    int* fin_step; int* fin_row; float* fin_score;   /* [B][beam]: finalised hypotheses in the order the reference appends them */
} S2TDecodeDesc;

/* Fragment-major copies for the MFMA B operand: lane l of fragment (tile, step) holds rows 16 tile + (l & 15), columns ks step + per (l >> 4) ..
 * (ks = 32, per = 8 for bf16; 16, 4 for f32), so a wave loads a fragment as one contiguous KiB.
 * s2t_decode_pack_weight: W [N][K] (row stride ldw elements, K % ks == 0) -> Wp [ceil(N / 16)][K / ks][64][per], rows past N zero.
 * s2t_decode_prepare_enc: kv [Ts][B][2D] (K | V rows of the encoder output under one layer's encoder_attn.kv) -> the layer's kv_enc and
 * vt_enc as S2TDecodeLayer describes them. */
int s2t_decode_pack_weight(int dtype, const void* W, int ldw, int N, int K, void* Wp, void* stream);
int s2t_decode_prepare_enc(int dtype, const void* kv, void* kv_enc, void* vt_enc, int Ts, int Tsp, int B, int D, int heads, void* stream);
/* resets the state for a new search: steps, blacklist, nfin, finished = 0; arrangement 0 = `bos` in every slot; x0 = its embedding */
int s2t_decode_begin(const S2TDecodeDesc* d, int bos, void* stream);
/* the launches of one step (see above).  `d` and d->layer are HOST memory read during the call only. */
int s2t_decode_step(const S2TDecodeDesc* d, void* stream);
/* dynamic LDS bytes the S / C / F launches of `d` need (0 when `d` is outside the limits): the caller may compare with 160 KiB */
size_t s2t_decode_lds_bytes(const S2TDecodeDesc* d);
/* n_steps (1..64) consecutive steps recorded as ONE hipGraph: `create` captures n_steps x s2t_decode_step(d) on a private stream (nothing
 * executes) and instantiates it, `launch` replays it on `stream` (the kernels read the step index from d->steps, so the same recording
 * serves every step, and every kernel returns at once for a sentence whose step index has passed max_len: replaying past the end is
 * harmless; `d`'s device buffers must stay where they are), `destroy` frees it after the caller has synchronised with the last replay.
 * *graph_exec is HOST. */
int s2t_decode_graph_create(const S2TDecodeDesc* d, int n_steps, void** graph_exec);
int s2t_decode_graph_launch(void* graph_exec, void* stream);
int s2t_decode_graph_destroy(void* graph_exec);

/* ---- measurement aid: what a collective costs the kernels beside it, on ONE GPU (bench.py data_parallel.dry_run) --------------------
 * `workgroups` workgroups stay resident on `stream` for the time a ring all-reduce of `bytes` over `ranks` ranks takes at `bus_gbps`
 * (2 (ranks - 1) / ranks * bytes / bus) and copy src -> dst twice meanwhile, evenly paced.  Stands in for RCCL's kernels of
 * fairseq/legacy_distributed_data_parallel.py:96-170's all-reduce as a competitor for CUs and HBM; moves nothing between GPUs. */
int s2t_comm_standin(const void* src, void* dst, size_t bytes, int workgroups, int ranks, float bus_gbps, void* stream);

/* ---- in-library timing of kernel families with HIP events (bench.py roofline) ------------------------
 * While enabled, every launch of the named family on `stream` is bracketed by hipEvents; s2t_prof_read
 * synchronises those events and returns accumulated milliseconds, launches and algorithmic flops/bytes. */
int s2t_prof_enable(int on);
/* ---- kernel-route options (tests cover every route; nothing here changes results beyond rounding) --------------------
 * key = "gemm256": 1 (default) sends the big bf16 NT / NN products to the 256x256x64 LDS-DMA kernel, 0 keeps them on the
 *                  128x128 register-staged kernels;
 *       "attn_v1": 1 forces the first-generation attention kernels (f32 / d 32 / short sequences use them anyway), default 0;
 *       "attn_v2_min_tq": shortest query block taken by the second-generation attention kernels (default 16);
 *       "gemm256_min_tiles": fewest 192 / 256-row output tiles a product must have for the 256-wide kernel (0 = default = 160;
 *                  tools/gemm_gate_probe.py measures both sides of it);
 *       "gemm256_sched": 1 selects gemm256's second K-loop schedule (diagnostic twins only: -95 in the product library);
 *       "gemm4w": 1 sends gemm256's NT products to the four-wave partition of the same tile (gemm4w.hip: an experiment,
 *                  bit-identical results, default 0);
 *       "reserve_cus": 0..128 (default 0): the persistent one-workgroup-per-CU kernels (gemm256, wgrad_group) launch 256 - value
 *                  workgroups and plan their rounds for that many CUs -- what a data-parallel run sets while RCCL's kernels share
 *                  the chip with backward (trainer: --reserve-cus);
 *       "gemm_f32_small_nt" / "gemm_f32_small_kt" (defaults 1024 / 512): f32 products with fewer 128 x 128 tiles than this take the 64 x 64
 *                  form (NT / NN and TN layouts): the exact-f32 MFMA is 1/16 of the bf16 rate, so what counts is that every CU has
 *                  tiles (tools/f32_tile_sweep.py: 22.96 -> 20.03 ms per update of configs[1]); "gemm_f32_narrow" (default 0): below
 *                  that many the 128 x 64 form;
 *       "gemm_small_nt" / "gemm_small_kt" (defaults 192 / 40): the same thresholds for bf16 products (tools/small_m_sweep.py: at M = 3,000 the
 *                  64 x 64 form loses on every NN product and the whole update moves within its noise: the defaults stay);
 *       "gemm_deep" (default 1): the 64 x 64 four-wave bf16 GEMM form requests 8 (NT) / 4 (NN, TN) k-tiles before its first MFMA instead
 *                  of keeping two in flight; bit-identical results (tools/gemm_deep_check.py); 0 restores the two-set loop;
 *       "ln_small" (default 1): the bf16, D = 512 LayerNorm backward of activations below 8,192 rows requests a wave's rows three at a
 *                  time instead of one ahead (same formulas; results agree with the other kernel to bf16 rounding); 0 restores it;
 *       "attn_bwd_fused" (default 0): 1 sends the bf16, d = 64, plain-softmax attention backward with 128 <= Tk <= 384 and Tq >= 128 to
 *                  the one-kernel form (attention.hip: attn_bwd_fused_kernel; same results within bf16 rounding, measured no faster:
 *                  profiles/r06_attn_bwd_fused.txt);
 *       "decode_stop_after": diagnostic, ends s2t_decode_step after that many launches (0 = off);
 * returns the previous value, or S2T_EINVAL (-22) for an unknown key or a value out of range. */
int s2t_set_option(const char* key, int value);
int s2t_prof_read(const char* family, double* ms, long long* launches, double* flops, double* bytes);
int s2t_prof_reset(void);

/* ---- host-side helpers (HOST pointers) -------------------------------------------------------------
 * greedy CTC decode + edit-distance alignment error count (compute_ctc_uer, CTC_loss.py:31-74;
 * examples/speech_recognition/utils/wer_utils.py:71-203) -- the reference runs this in pure Python
 * on the critical path of every step. */
int s2t_host_ctc_uer(const int* pred, const long long* input_len, int B, int T, const long long* targets,
                     const long long* target_len, int L, int blank, double* errors, double* total);

#ifdef __cplusplus
}
#endif
#endif
