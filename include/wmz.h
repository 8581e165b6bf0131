/* libwmz_hip.so -- C ABI of the MI355X (gfx950) denoiser hot path.
 *
 * The reference (world-modelz/world-modelz, vq-video-diffusion/) has no FFI: its boundary for this path is
 * the Python nn.Module surface.  These entry points are what the drop-in modules in
 * world_modelz_amd/ call through ctypes; each one names the reference code it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless named host_*; tensors are row-major / channels-last,
 *    exactly the layouts the reference holds at that point ([B,S,H,W,C] token grids, [N,E] latents);
 *  - `dtype` selects the element type of activations: WMZ_F32 or WMZ_BF16 (accumulation is always fp32; WMZ_F16 where an
 *    entry point says so);
 *    parameters marked `float*` are always fp32, indices are int64 like the reference's LongTensors;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); kernels are only enqueued;
 *  - no allocation, no host sync, no global state: safe to capture in a hipGraph, re-entrant per stream (the library also
 *    exports a few wmz_debug_* development probes -- process-wide timing switches declared in the private header
 *    world_modelz_amd/csrc/wmz_debug.h, not part of this interface);
 *  - return 0 on success, WMZ_ERR_* otherwise; wmz_last_error() gives the message (thread-local).
 */
#ifndef WMZ_H_
#define WMZ_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WMZ_VERSION 113

enum { WMZ_F32 = 0, WMZ_BF16 = 1,
       WMZ_F16 = 2 /* IEEE half activations / MFMA operands: the PRECISE fused inference mode (wmz_local3d_attn_fwd* on the
                      16- / 8-wide-plane fast path and the *_f16 per-token entry points below); other entry points refuse it */ };
enum { WMZ_OK = 0, WMZ_ERR_ARG = 1, WMZ_ERR_HIP = 2, WMZ_ERR_UNSUPPORTED = 3 };

/* epilogue / prologue flags of wmz_linear_fwd */
enum {
  WMZ_LIN_GELU = 1,    /* exact erf GELU on (A W^T + b) */
  WMZ_LIN_GELU_IN = 2, /* exact erf GELU applied to A while it is staged (A = saved pre-activation) */
  WMZ_LIN_DGELU = 4    /* backward: `residual` holds the pre-activation z, C = (A W^T + b) * gelu'(z) */
};

int wmz_version(void);
const char* wmz_last_error(void);

/* ---- Local3dAttention.local_attention (local_3d_attention.py:78-99; pad :57-63, unfold :65-69, mask :71-76)
 * q,k,v: [B,S,H,W,heads*dh] (row strides ldq/ldk/ldv in elements, head-major channels), out same shape
 * (row stride ldo).  Softmax runs over the in-grid neighbours of the (2eS+1)(2eH+1)(2eW+1) window, which
 * equals the reference's zero-pad + -1e9 masking.  lse (optional, fp32 [B*S*H*W, heads]) receives
 * log-sum-exp of the scaled logits for the backward.  logits_dbg (optional, fp32 [N, heads, K], caller
 * pre-fills with -1e9) receives the scaled logits in the reference's (i j k) slot order: parity probe only. */
int wmz_local3d_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, float* logits_dbg,
                         int B, int S, int H, int W, int heads, int dh, int eS, int eH, int eW,
                         long ldq, long ldk, long ldv, long ldo, int dtype, void* stream);

/* Same contract, always on the general kernel (any W, fp32 / bf16) even where wmz_local3d_attn_fwd would pick the
 * 16-wide-plane fast path: the parity tests hold the two against each other. */
int wmz_local3d_attn_fwd_general(const void* q, const void* k, const void* v, void* out, float* lse, float* logits_dbg,
                                 int B, int S, int H, int W, int heads, int dh, int eS, int eH, int eW,
                                 long ldq, long ldk, long ldv, long ldo, int dtype, void* stream);

/* Backward of the above given the saved lse: dq, dk, dv.  Gather form, no atomics (the window relation is
 * symmetric): one query-owner pass (dq; it leaves delta = rowsum(dout*out) and -lse / scale per token in delta_ws) and
 * one key-owner pass (dk, dv) that starts its accumulators from those.  delta_ws: caller-allocated scratch of
 * 2 * N * heads floats (N = B*S*H*W), contents unspecified on return.  Replaces the checkpoint re-run + autograd of
 * local_3d_attention.py:110-111. */
int wmz_local3d_attn_bwd(const void* q, const void* k, const void* v, const void* out, const float* lse,
                         const void* dout, void* dq, void* dk, void* dv, float* delta_ws,
                         int B, int S, int H, int W, int heads, int dh, int eS, int eH, int eW,
                         long ldq, long ldk, long ldv, long ldo, long lddo, long lddq, long lddk, long lddv,
                         int dtype, void* stream);

/* ---- nn.Linear family (local_3d_attention.py:46-53 to_q/to_k/to_v/to_out, :23-29 FeedForward, main.py:31
 * logit_proj), optionally fused with the PreNorm LayerNorm in front (:14-17) and the residual add behind
 * (:160-161):   C[M,N] = act( LN?(A)[M,K] @ Wt[N,K]^T + bias[N] ) + residual[M,N]
 * A, C, residual: activation dtype (lda/ldc/ldr row strides in elements); Wt: activation dtype, row-major
 * [N,K] (nn.Linear.weight layout); bias, ln_gamma, ln_beta: fp32 or NULL; ln_gamma != NULL enables the
 * LayerNorm prologue over the K axis (eps = ln_eps).  out_f32 != 0 stores C as fp32 regardless of dtype
 * (logits).  Requires K % 8 == 0. */
int wmz_linear_fwd(const void* A, long lda, const void* Wt, const float* bias, const void* residual, long ldr,
                   void* C, long ldc, int M, int N, int K, const float* ln_gamma, const float* ln_beta,
                   float ln_eps, int flags, int out_f32, int dtype, void* stream);

/* The same with the LayerNorm statistics supplied (mean / rstd [M] from wmz_layernorm_stats, or NULL): the training
 * forward computes them once, the prologue skips its two passes over A, and the backward reuses them. */
int wmz_linear_fwd_stats(const void* A, long lda, const void* Wt, const float* bias, const void* residual, long ldr,
                         void* C, long ldc, int M, int N, int K, const float* ln_gamma, const float* ln_beta,
                         const float* ln_mean, const float* ln_rstd, float ln_eps, int flags, int out_f32, int dtype,
                         void* stream);

/* FeedForward's first GEMM as the training forward wants it (local_3d_attention.py:24-25: nn.Linear -> nn.GELU): the
 * pre-activation Z = LN?(A) Wt^T + bias (what gelu' needs in the backward) AND the activation H = GELU(Z), both [M, N] in the
 * activation dtype (ldz / ldh row strides), from one accumulator.  The second GEMM and its weight gradient read H as it is
 * instead of re-evaluating the erf of their operand panel once per output column tile. */
int wmz_linear_fwd_gelu_pair(const void* A, long lda, const void* Wt, const float* bias, void* Z, long ldz, void* H, long ldh,
                             int M, int N, int K, const float* ln_gamma, const float* ln_beta, const float* ln_mean,
                             const float* ln_rstd, float ln_eps, int dtype, void* stream);
/* The training forward's PreNorm GEMM in full: C = LN(A) Wt^T + bias; H (optional) = GELU(C); An (optional) = LN(A) [M, K] as the
 * GEMM consumed it, i.e. the operand of the layer's weight gradient -- which then needs no LayerNorm prologue
 * (wmz_linear_wgrad_batch*'s 256-wide tiles take plain operands only).  ln_mean / ln_rstd: supplied or NULL (computed). */
int wmz_linear_fwd_train(const void* A, long lda, const void* Wt, const float* bias, void* C, long ldc, void* H, long ldh,
                         void* An, long ldan, int M, int N, int K, const float* ln_gamma, const float* ln_beta,
                         const float* ln_mean, const float* ln_rstd, float ln_eps, int dtype, void* stream);

/* logit_proj on the LAST FRAME of every clip, read in place (main.py:35-36: x[:, -1] -> nn.Linear): C[M,N] = A' Wt^T + bias
 * where row m of A' is A + (m / rows_per_block) * block_stride + (m % rows_per_block) * lda (elements); rows_per_block =
 * H*W, block_stride = S*H*W*D selects the last plane of each clip of a [B,S,H,W,D] stream without a gather copy. */
int wmz_linear_fwd_blocked(const void* A, long lda, int rows_per_block, long block_stride, const void* Wt,
                           const float* bias, void* C, long ldc, int M, int N, int K, int out_f32, int dtype, void* stream);

/* Weight / bias gradient of the family above: dW[N,K] += dC[M,N]^T . A'[M,K], dbias[N] += colsum(dC), where
 * A' = A, LayerNorm(A) (ln_* non-NULL; mean/rstd from wmz_layernorm_stats) or GELU(A) (gelu_in).  dW / dbias are fp32
 * and ACCUMULATED with float atomics (split over M): zero them first, or pass .grad buffers to accumulate. */
int wmz_linear_wgrad(const void* dC, long ldc, const void* A, long lda, float* dW, float* dbias, int M, int N, int K,
                     const float* ln_gamma, const float* ln_beta, const float* ln_mean, const float* ln_rstd,
                     int gelu_in, int dtype, void* stream);

/* The same with a two-stage reduction instead of float atomics: the slices of M leave their 128 x 128 partial tiles in
 * `workspace` (fp32, at least wmz_linear_wgrad_workspace_floats(M, N, K, dtype) floats, contents undefined afterwards) and
 * a second launch adds their sum to dW / dbias (overwrite != 0: stores it instead) -- deterministic summation order, no
 * same-address atomics.  The caller owns the workspace (the library never allocates); one workspace can serve every wgrad
 * on a stream. */
long wmz_linear_wgrad_workspace_floats(int M, int N, int K, int dtype);
int wmz_linear_wgrad_ws(const void* dC, long ldc, const void* A, long lda, float* dW, float* dbias, int M, int N, int K,
                        const float* ln_gamma, const float* ln_beta, const float* ln_mean, const float* ln_rstd,
                        int gelu_in, int overwrite, float* workspace, long workspace_floats, int dtype, void* stream);
/* n <= 6 independent weight gradients (no prologue) by ONE launch pair: HOST tables of n entries each, the arguments of
 * wmz_linear_wgrad_ws per problem; the workspace must hold the sum of the problems' workspace sizes.  (The fused backward
 * of a transformer layer has five such GEMMs whose launches are each one wave of workgroups: batched they share the ramp
 * and the tail.)  a_tiled: NULL, or a table whose non-zero entries say that problem i's A is the fused path's TILED stream
 * (WMZ_FUSED_X_OUT_TILED: bf16, K = 256, M a multiple of 32; lda is ignored) -- the to_q weight gradient reads the raw
 * stream where the forward left it. */
int wmz_linear_wgrad_batch(int n, const void* const* dC, const long* ldc, const void* const* A, const long* lda,
                           float* const* dW, float* const* dbias, const int* M, const int* N, const int* K,
                           const int* overwrite, const int* a_tiled, float* workspace, long workspace_floats, int dtype,
                           void* stream);
/* The same with an optional LayerNorm prologue per problem (tables of n pointers; a NULL ln_gamma[i] = plain problem, else
 * ln_beta[i], ln_mean[i], ln_rstd[i] as for wmz_linear_wgrad_ws): the weight gradients of one transformer layer on the
 * op-by-op path as one launch pair. */
int wmz_linear_wgrad_batch_ln(int n, const void* const* dC, const long* ldc, const void* const* A, const long* lda,
                              float* const* dW, float* const* dbias, const int* M, const int* N, const int* K,
                              const int* overwrite, const float* const* ln_gamma, const float* const* ln_beta,
                              const float* const* ln_mean, const float* const* ln_rstd, float* workspace, long workspace_floats,
                              int dtype, void* stream);
/* nn.LayerNorm statistics (PreNorm, local_3d_attention.py:14): mean[M], rstd[M] over the K axis. */
int wmz_layernorm_stats(const void* x, long ldx, float* mean, float* rstd, int M, int K, float eps, int dtype,
                        void* stream);
/* nn.LayerNorm backward: dx = dLN(x)^T dyhat + skip + skip2 (both optional: the residual gradient and the gradient of the
 * un-normalised q path, which land on the same tensor in x = attn(LN(x), q = x) + x),
 * dgamma[K] += sum_m dyhat*xhat, dbeta[K] += sum_m dyhat (fp32, atomics). */
int wmz_layernorm_bwd(const void* x, long ldx, const void* dyhat, long lddy, const void* skip, long ldskip,
                      const void* skip2, long ldskip2, const float* gamma, void* dx, long lddx, float* dgamma,
                      float* dbeta, int M, int K, float eps, int dtype, void* stream);

/* ---- Local3dAttentionTransformer embedding (local_3d_attention.py:140-157):
 * x[b,s,h,w,:] = emb[z[b,s,h,w]] + ((pos_s[s] + pos_h[h]) + pos_w[w]); tables fp32, x in `dtype`. */
int wmz_embed_pos3d_fwd(const int64_t* z, const float* emb, const float* pos_s, const float* pos_h,
                        const float* pos_w, void* x, int B, int S, int H, int W, int D, int num_classes,
                        int dtype, void* stream);
/* its backward: scatter-add dx into the four fp32 tables (accumulated; zero them first). */
int wmz_embed_pos3d_bwd(const int64_t* z, const void* dx, float* demb, float* dpos_s, float* dpos_h, float* dpos_w,
                        int B, int S, int H, int W, int D, int num_classes, int dtype, void* stream);
/* The same gradients WITHOUT a float atomic per element (W = 16, D = 256, bf16, <= 12 288 classes; WMZ_ERR_UNSUPPORTED
 * otherwise): the tokens are counting-sorted by class (histogram -> offsets -> list, int atomics only) and the embedding
 * table's gradient is gathered, one wave per 64 list entries (csrc/embed_bwd.hip).  `workspace`: at least
 * wmz_embed_pos3d_bwd_workspace_ints(...) ints, ZERO-FILLED by the caller before the first call; every call leaves the
 * counters zeroed again (so a captured hipGraph can replay it), the rest is scratch. */
long wmz_embed_pos3d_bwd_workspace_ints(int B, int S, int H, int W, int num_classes);
int wmz_embed_pos3d_bwd_sorted(const int64_t* z, const void* dx, float* demb, float* dpos_s, float* dpos_h, float* dpos_w,
                               int B, int S, int H, int W, int D, int num_classes, int* workspace, long workspace_ints,
                               int dtype, void* stream);

/* config 5, VqSparseDiffusionModel (minecraft/sparse_diffusion.py:91-111): ntok tokens at arbitrary flat grid positions:
 * x[t,:] = emb[tok[t]] + ((pos_s[p/(H*W)] + pos_h[(p/W)%H]) + pos_w[p%W]), p = pos[t]; and its backward (atomics, fp32). */
int wmz_embed_indexed_fwd(const int64_t* tok, const int64_t* pos, const float* emb, const float* pos_s,
                          const float* pos_h, const float* pos_w, void* x, long ntok, int S, int H, int W, int D,
                          int num_classes, int dtype, void* stream);
int wmz_embed_indexed_bwd(const int64_t* tok, const int64_t* pos, const void* dx, float* demb, float* dpos_s,
                          float* dpos_h, float* dpos_w, long ntok, int S, int H, int W, int D, int num_classes, int dtype,
                          void* stream);

/* ---- VectorQuantizerEMA (vq.py) ----
 * wmz_vq_argmin: encode / codebook_distance+argmin (vq.py:77-87, :30-33).  x [N,E] fp32 (row stride ldx),
 * codebook [C,E] fp32.  dist = sum_e (x-e)^2 evaluated in the exact fp32 order ATen uses on x86
 * (8 lanes x 4 accumulators, no FMA contraction), ties -> lowest index: indices are bit-identical to the
 * reference.  idx int64 [N]; dist_min (optional) fp32 [N]. */
int wmz_vq_argmin(const float* x, long ldx, const float* codebook, int64_t* idx, float* dist_min,
                  int N, int C, int E, void* stream);

/*
 * wmz_vq_argmin_screened: the same result as wmz_vq_argmin (indices AND minimum distances bit-identical: the reference's fp32
 * summation order decides), computed by screening on the matrix cores + exact re-check (csrc/vq_screen.hip): bf16 head / tail
 * split products with a proven error bound pick each row's code; rows whose two best codes are closer than twice the bound
 * (equal codes, near ties) are re-scanned over the whole codebook in the pinned arithmetic.  Built for embedding_dim 64 and a
 * multiple of 64 codes: wmz_vq_argmin_screened_workspace_bytes returns 0 for any other shape (use wmz_vq_argmin).
 * workspace: caller-allocated device scratch of that many bytes (contents irrelevant; rewritten by every call).
 * Replaces: vq.py:30-33, :77-87 (codebook_distance + argmin), as wmz_vq_argmin does.
 */
long wmz_vq_argmin_screened_workspace_bytes(int N, int C, int E);
int wmz_vq_argmin_screened(const float* x, long ldx, const float* codebook, int64_t* idx, float* dist_min, int N, int C, int E,
                           void* workspace, long workspace_bytes, void* stream);
/* decode (vq.py:89-94): out[n,:] = codebook[idx[n],:]; out in `dtype` (row stride ldo). */
int wmz_vq_gather(const int64_t* idx, const float* codebook, void* out, long ldo, int N, int C, int E,
                  int dtype, void* stream);
/* The tail of VectorQuantizerEMA.forward (vq.py:67 commitment loss, :70 straight-through estimator, :72-73 perplexity) as two
 * launches: st [N, Ep] = x + (q - x) in out_dtype (columns E..Ep zero: the decoder's first conv wants channels % 8 == 0),
 * loss[0] = mean (q - x)^2, perplexity[0] = exp(-sum p log(p + 1e-10)), p = counts / N.  x, q fp32 [N, E] contiguous, counts fp32
 * [C]; partial: wmz_loss_partials_workspace_floats() floats of workspace (per-workgroup sums, added in a fixed order: deterministic).
 * wmz_vq_tail_bwd: d_x [N, E] (in_dtype) = d_st[:, :E] (st_dtype, row stride Ep; NULL: 0) + g_loss[0] * 2 / (N E) * (x - q)
 * (g_loss: device scalar, NULL: 0). */
long wmz_loss_partials_workspace_floats(void);
int wmz_vq_tail_fwd(const float* x, const float* q, const float* counts, void* st, float* partial, float* loss,
                    float* perplexity, long N, int E, int Ep, int C, int out_dtype, void* stream);
int wmz_vq_tail_bwd(const void* d_st, const float* x, const float* q, const float* g_loss, void* d_x, long N, int E, int Ep,
                    int in_dtype, int st_dtype, void* stream);
/* The reconstruction loss of train_vqae.py:139-150 / :264-271 (kind 0 SmoothL1 with beta 1, 1 MSE, 2 L1; reduction 'mean') read
 * from the decoder's NHWC output in place: y [B, HW, Cp] in `dtype` (channels C..Cp are padding), target [B, C, HW] fp32 (NCHW
 * frames); loss[0] fp32.  wmz_recon_loss_bwd: d_y [B, HW, Cp] = g_loss[0] (device scalar, NULL: 1) / (B HW C) * loss'(y - target),
 * zero in the padding channels -- the operand of the last conv's gradients, no NCHW copy either way. */
int wmz_recon_loss_fwd(const void* y, const float* target, float* partial, float* loss, long B, long HW, int C, int Cp,
                       int kind, int dtype, void* stream);
int wmz_recon_loss_bwd(const void* y, const float* target, const float* g_loss, void* d_y, long B, long HW, int C, int Cp,
                       int kind, int dtype, void* stream);
/* statistics of forward (vq.py:35-36, :43-46): counts[c] += #{n: idx[n]==c}, dw[c,:] += sum x[n,:],
 * sqerr[c] += sum |codebook[c]-x[n]|^2.  Caller zeroes counts/dw (sqerr accumulates into accumulated_error). */
int wmz_vq_ema_stats(const float* x, long ldx, const int64_t* idx, const float* codebook, float* counts,
                     float* dw, float* sqerr, int N, int C, int E, void* stream);
/* The same accumulations as a counting sort by code + gather (<= 12 288 codes, E <= 256; csrc/class_sort.h, csrc/vq.hip):
 * ~2 flushes per 64 rows instead of one atomic per element -- the scatter form serialises on sqerr[c] / dw[c] (1.8 ms at
 * N = 65 536, C = 1 024, E = 64; 35 us here, whatever the code distribution).  `workspace`: >=
 * wmz_vq_ema_stats_workspace_ints(N, C) ints, ZERO-FILLED by the caller before the first call; calls leave the counters zeroed. */
long wmz_vq_ema_stats_workspace_ints(int N, int C);
int wmz_vq_ema_stats_sorted(const float* x, long ldx, const int64_t* idx, const float* codebook, float* counts, float* dw,
                            float* sqerr, int N, int C, int E, int* workspace, long workspace_ints, void* stream);
/* EMA update (vq.py:53-65): cluster_size = g*cs + (1-g)*counts; n = sum(cs);
 * embedding = g*embedding + (1-g) * dw / ((cs+eps)/(n+C*eps)*n).   activation_count += counts (vq.py:44). */
int wmz_vq_ema_update(float* embedding, float* cluster_size, float* activation_count, const float* counts,
                      const float* dw, int C, int E, double decay, double eps, void* stream);

/* ---- everything per-token of a transformer layer boundary in ONE launch (bf16 speed path):
 * head (has_head): x1 = o Wout^T + bout + x ; x_out = W2 GELU(W1 LN2(x1) + b1) + b2 + x1
 *                  (to_out + residual, PreNorm(FeedForward) + residual: local_3d_attention.py:50-53, :20-31, :160-161)
 * tail (has_tail): q = Wq' x_out ; k|v = Wkv' LN1'(x_out) + bkv'   (the NEXT layer's to_q / to_k / to_v, :46-48, :106-108;
 *                  without a head the tail reads x directly: the first layer after the embedding)
 * o [ntok, I], x / x_out [ntok, D], q [ntok, I], kv [2, ntok, I] (the k rows, then the v rows), all bf16 contiguous.
 * wpack: the stage weights as bf16, pre-packed in consumption order -- Wout, then for c = 0..7: W1' rows 32c..32c+31,
 * W2 columns 32c..32c+31, then Wq, Wk', Wv' -- as 1 KB MFMA 32x32x16 A operands in (16-deep k-step, 32-feature block)
 * order, followed by 64 KB of padding; the LayerNorm affines are folded in (W1' = W1 diag(g2), Wk' = Wk diag(g1), ..).
 * vec: 2048 fp32: bout[D] b1'[M] b2[D] bk'[I] bv'[I] (b1' = b1 + W1 be2, bk' = Wk be1, bv' = bv + Wv be1), zero padded.
 * world_modelz_amd/fused.py (_pack_w, _layer_pack) builds both and documents the exact element order.  Built for
 * D = 256, I = 128, M = 256; other widths return WMZ_ERR_UNSUPPORTED and callers use the per-op entry points above. */
int wmz_layer_fused_fwd(const void* o, const void* x, void* x_out, void* q_out, void* kv_out, const void* wpack,
                        const float* vec, int ntok, int D, int I, int M, int has_head, int has_tail, float eps,
                        void* stream);

/* First layer: the token + 3-axis position embedding (local_3d_attention.py:140-157) fused with the tail above:
 * x_out = embedding (bf16), q / k|v of layer 0 from it.  wpack / vec as for a tail-only call. */
int wmz_embed_qkv_fused_fwd(const int64_t* z, const float* emb, const float* pos_s, const float* pos_h,
                            const float* pos_w, void* x_out, void* q_out, void* kv_out, const void* wpack,
                            const float* vec, int B, int S, int H, int W, int D, int I, int M, int num_classes, float eps,
                            void* stream);

/* ---- trailing-planes variants (inference).  main.py:37 keeps only the LAST frame's logits (x[:, -1]), so layer l of a
 * depth-L stack only has to produce the last min(S, 1 + (L-1-l) * eS) planes -- the dependence cone of the last frame
 * through the temporal windows (local_3d_attention.py:95-104).  Same arithmetic per token as the full-grid calls (results
 * are bit-identical on the planes that are computed); the dead planes are simply never launched.
 *   attention: q / k / v cover all S planes, queries are taken from planes [q_plane0, q_plane0 + q_planes) only and
 *              out / lse are compact [B, q_planes, H, W, ..];
 *   fused layer: x holds planes_in planes per clip, o and the outputs hold its last planes_out planes;
 *   embedding:  z is the full [B, S, H, W] token grid, the outputs hold the last planes_out planes of each clip;
 *   xflags:     WMZ_FUSED_X_IN_TILED / WMZ_FUSED_X_OUT_TILED -- x / x_out in the fused path's private stream layout (per
 *               32-token tile: [16 chunks][2 halves][32 tokens][8 features], every access of the kernel one contiguous
 *               KB) instead of row-major; needs whole 32-token tiles per clip.  Layer -> layer only: the last layer
 *               writes row-major. */
#define WMZ_FUSED_X_IN_TILED 1
#define WMZ_FUSED_X_OUT_TILED 2
/* wmz_layer_fused_fwd_train only: x1_out receives the NORMALISED rows (x1 - mean) rstd of the feed-forward block's input
 * instead of x1 itself -- everything the fused backward needs of x1 (wmz_ff_fused_bwd with xhat_out = NULL). */
#define WMZ_FUSED_X1_NORMALISED 4
/* wmz_layer_fused_fwd_train / wmz_embed_qkv_fused_fwd_train with a tail and a TILED x_out: x_out_rowmajor receives the
 * NORMALISED rows (x - mean) rstd of the stream (the next layer's to_k | to_v input without the affine) instead of a copy of x:
 * what wmz_qkv_fused_bwd (xhat_out = NULL) and the to_k | to_v weight gradient read; the raw stream -- the to_q weight
 * gradient's operand -- is read from the tiled x_out (wmz_linear_wgrad_batch: a_tiled). */
#define WMZ_FUSED_XRM_NORMALISED 8
int wmz_local3d_attn_fwd_planes(const void* q, const void* k, const void* v, void* out, float* lse, int B, int S, int H,
                                int W, int heads, int dh, int eS, int eH, int eW, long ldq, long ldk, long ldv, long ldo,
                                int q_plane0, int q_planes, int dtype, void* stream);
int wmz_layer_fused_fwd_planes(const void* o, const void* x, void* x_out, void* q_out, void* kv_out, const void* wpack,
                               const float* vec, int B, int planes_out, int planes_in, int HW, int D, int I, int M,
                               int has_head, int has_tail, int xflags, float eps, void* stream);
/*
 * The same fusion for widths the default-width kernel cannot hold in registers (csrc/layer_chain.hip: 16-token waves, MFMA
 * 16x16x32, fp32 residual stream in registers, activations chained lane-locally, weights by an LDS-DMA ring): the width triples of
 * csrc/chain_widths.h -- the reference's published runs (dim 96 / mlp 256, dim 384 / mlp 512; results/README.md), its own test()
 * geometry (local_3d_attention.py:166-174: dim 128, 3 heads of 64, mlp 256) and the neighbours of the argparse defaults.
 * wmz_layer_chain_supported: 1 when (D, I, M) is instantiated, *mc = the hidden-chunk size the weight packer must use.
 * Tensors as wmz_layer_fused_fwd_planes (row-major only); wpack / vec in the order of world_modelz_amd/fused.py::_chain_pack,
 * wpack followed by two slabs of readable padding.  Replaces: local_3d_attention.py:11-31, :46-53, :106-108, :159-161.
 */
int wmz_layer_chain_supported(int D, int I, int M, int* mc);
int wmz_layer_chain_slab_pieces(void);     /* pieces (KB) per weight slab: the packer pads every stage to a multiple of it */
int wmz_layer_chain_fwd_planes(const void* o, const void* x, void* x_out, void* q_out, void* kv_out, const void* wpack,
                               const float* vec, int B, int n_q, int n_in, int HW, int D, int I, int M, int head, int tail,
                               float eps, void* stream);

/* The TRAINING forward of the same launch (whole grids of ntok tokens); wpack / vec exactly as for wmz_layer_chain_fwd_planes
 * (LayerNorm affines folded).  Besides x_out [ntok, D], q_out [ntok, I] and kv_out [ntok, 2 I] (k | v per row) it writes what the
 * step's backward reads: x1 [ntok, D] the feed-forward block's raw input (or NULL), xn_ff [ntok, D] its NORMALISED rows (they are
 * the packed operand of the GEMM behind the norm), z / h [ntok, M] the pre-activation and GELU of it, st_ff [2, ntok] mean | rstd
 * (head != 0); xn_attn [ntok, D] normalised rows and st_attn [2, ntok] for the next layer's to_k / to_v (tail != 0).
 * main.py:216-287 (the step), local_3d_attention.py:11-31. */
int wmz_layer_chain_fwd_train(const void* o, const void* x, void* x_out, void* q_out, void* kv_out, const void* wpack,
                              const float* vec, void* x1, void* xn_ff, void* z, void* h, float* st_ff, void* xn_attn,
                              float* st_attn, long ntok, int D, int I, int M, int head, int tail, float eps, void* stream);

/* Backward of the same per-token work (csrc/layer_chain_bwd.hip), two launches per layer around the attention backward
 * (main.py:278 through local_3d_attention.py:11-31, :46-53, :160-161):
 *   wmz_chain_ff_bwd:  dy [ntok, D] -> dz = (W2^T dy) gelu'(z) [ntok, M] (operand of dW1), dx1 = dy + LayerNorm'(W1'^T dz; xhat, rstd)
 *                      [ntok, D], dout = Wout^T dx1 [ntok, I].  z / xhat / rstd: wmz_layer_chain_fwd_train's z, xn_ff, st_ff[1].
 *                      wpack: per hidden chunk (wmz_layer_chain_supported's mc) the pieces of W2[:, chunk]^T then of W1'[chunk]^T
 *                      (W1' = W1 diag(gamma_ff)), then Wout^T padded to whole slabs, + 2 slabs of readable padding.
 *   wmz_chain_qkv_bwd: dx [ntok, D] = dx1 + Wq^T dq + LayerNorm'(Wk'^T dk + Wv'^T dv; xhat, rstd); dq [ntok, I], dkv [ntok, 2 I];
 *                      wpack: pieces of Wk'^T, Wv'^T (gamma_attn folded; each padded to whole slabs), Wq^T, + 2 slabs.
 * The weight gradients behind a norm are then plain GEMMs against xhat, converted by wmz_ln_affine_grads. */
int wmz_chain_ff_bwd(const void* dy, const void* z, const void* xhat, const float* rstd, void* dz, void* dx1, void* dout,
                     const void* wpack, long ntok, int D, int I, int M, void* stream);
int wmz_chain_qkv_bwd(const void* dq, const void* dkv, const void* xhat, const float* rstd, const void* dx1, void* dx,
                      const void* wpack, long ntok, int D, int I, void* stream);

int wmz_embed_qkv_fused_fwd_planes(const int64_t* z, const float* emb, const float* pos_s, const float* pos_h,
                                   const float* pos_w, void* x_out, void* q_out, void* kv_out, const void* wpack,
                                   const float* vec, int B, int S, int H, int W, int planes_out, int D, int I, int M,
                                   int num_classes, int xflags, float eps, void* stream);

/* ---- the PRECISE fused inference mode: the same kernels with IEEE-half MFMA operands and a half residual stream between the
 * layers (same matrix rate as bf16, 11 significand bits instead of 8).  The reference computes in fp32
 * (local_3d_attention.py:78-118); BASELINE.json asks for attention logits within 1e-3 relative of it, which bf16 operands miss
 * end to end (4e-3 on the default model) and fp32 operands pay 11x for (the exact-f32 MFMA runs at 1/16 of the rate): half
 * operands give 5e-4 at the bf16 speed.  Contracts as the entry points without the suffix, every `bf16` there read as `half`;
 * wmz_local3d_attn_fwd / _fwd_planes take dtype = WMZ_F16 for the same tensors (dim_head 32 / 64 / 128, planes 16 wide or 8
 * wide with an even number of rows; anything else returns WMZ_ERR_UNSUPPORTED).  Values beyond +-65504 become infinities. */
int wmz_layer_fused_fwd_f16(const void* o, const void* x, void* x_out, void* q_out, void* kv_out, const void* wpack,
                            const float* vec, int ntok, int D, int I, int M, int has_head, int has_tail, float eps,
                            void* stream);
int wmz_embed_qkv_fused_fwd_f16(const int64_t* z, const float* emb, const float* pos_s, const float* pos_h,
                                const float* pos_w, void* x_out, void* q_out, void* kv_out, const void* wpack,
                                const float* vec, int B, int S, int H, int W, int D, int I, int M, int num_classes, float eps,
                                void* stream);
int wmz_layer_fused_fwd_planes_f16(const void* o, const void* x, void* x_out, void* q_out, void* kv_out, const void* wpack,
                                   const float* vec, int B, int planes_out, int planes_in, int HW, int D, int I, int M,
                                   int has_head, int has_tail, int xflags, float eps, void* stream);
int wmz_embed_qkv_fused_fwd_planes_f16(const int64_t* z, const float* emb, const float* pos_s, const float* pos_h,
                                       const float* pos_w, void* x_out, void* q_out, void* kv_out, const void* wpack,
                                       const float* vec, int B, int S, int H, int W, int planes_out, int D, int I, int M,
                                       int num_classes, int xflags, float eps, void* stream);
int wmz_layer_fused_pack_f16(const float* wout, const float* bout, const float* g2, const float* be2, const float* w1,
                             const float* b1, const float* w2, const float* b2, const float* g1, const float* be1,
                             const float* wq, const float* wk, const float* wv, const float* bv, void* wpack, float* vec,
                             int D, int I, int M, void* stream);
int wmz_fused_pack_table_f16(const void* block_rows, int nblk, long total8, const void* vec_jobs, int nvec, int D, int I, int M,
                             void* stream);
/* ... the same for the chain kernels' widths (wmz_layer_chain_supported, csrc/chain_widths.h): wmz_layer_chain_fwd_planes
 * with half tensors and a half weight stream (world_modelz_amd/fused.py::_chain_pack packs either format). */
int wmz_layer_chain_fwd_planes_f16(const void* o, const void* x, void* x_out, void* q_out, void* kv_out, const void* wpack,
                                   const float* vec, int B, int n_q, int n_in, int HW, int D, int I, int M, int head, int tail,
                                   float eps, void* stream);
/* ... and the linear behind it (VqVideoDiffusionModel.logit_proj on the last frame, main.py:33-36): wmz_linear_fwd /
 * wmz_linear_fwd_stats / wmz_linear_fwd_blocked with dtype = WMZ_F16 (A and Wt half, bias / LayerNorm parameters fp32, C half or,
 * with out_f32, fp32). */
int wmz_linear_fwd_f16(const void* A, long lda, const void* Wt, const float* bias, const void* residual, long ldr, void* C,
                       long ldc, int M, int N, int K, const float* ln_gamma, const float* ln_beta, float ln_eps, int flags,
                       int out_f32, int dtype, void* stream);
int wmz_linear_fwd_stats_f16(const void* A, long lda, const void* Wt, const float* bias, const void* residual, long ldr, void* C,
                             long ldc, int M, int N, int K, const float* ln_gamma, const float* ln_beta, const float* ln_mean,
                             const float* ln_rstd, float ln_eps, int flags, int out_f32, int dtype, void* stream);
int wmz_linear_fwd_blocked_f16(const void* A, long lda, int rows_per_block, long block_stride, const void* Wt, const float* bias,
                               void* C, long ldc, int M, int N, int K, int out_f32, int dtype, void* stream);
/* All MFMA-operand copies of the fp32 parameters in one launch (after wmz_adamw_step has rewritten the weights): entry i
 * turns the logical matrix [rows0 + rows1, cols] = (src0 ; src1) -- src1 optional (row concatenation, e.g. to_k over
 * to_v), src0 NULL = zero rows -- into dst[i], row-major or transposed (WMZ_OPERAND_TRANSPOSE: what the dgrad GEMMs
 * read), bf16 or fp32 (WMZ_OPERAND_F32).  The tables are HOST arrays of n <= 64 entries. */
#define WMZ_OPERAND_TRANSPOSE 1
#define WMZ_OPERAND_F32 2
int wmz_operands_refresh(const void* const* src0, const void* const* src1, const int* rows0, const int* rows1,
                         const int* cols, void* const* dst, const int* flags, int n, void* stream);
/* The conv encoder / decoder's GEMM operands from nn.Conv2d's fp32 weights [Co, Ci, KH, KW] (kk = KH*KW), every layer in one
 * launch (n <= 48 entries): mode 0 = the forward / weight-gradient layout [Co, kk * Ci8] (tap-major, channels fastest,
 * zero-padded to a multiple of 8), mode 1 = the data-gradient layout [Ci8, kk * Co8] with the taps flipped
 * (autoencoder.py:_w_op / _wT_op); dst in `dtype`. */
int wmz_conv_operands_refresh(const void* const* weight, void* const* dst, const int* co, const int* ci, const int* kk,
                              const int* mode, int n, int dtype, void* stream);
/* The same with, per entry, an optional fragment-order destination instead of the row-major one (bf16): pack[i] = 1: the weight
 * stream of wmz_conv3x3_direct_fwd for that operand (what wmz_conv3x3_direct_pack makes of it), 2: wmz_conv_point_fwd's
 * (wmz_conv_point_pack), 0: row-major as above; pack == NULL: all row-major. */
int wmz_conv_operands_refresh_packed(const void* const* weight, void* const* dst, const int* co, const int* ci, const int* kk,
                                     const int* mode, const int* pack, int n, int dtype, void* stream);

/* Builds the packed weight stream and the vector block of wmz_layer_fused_fwd* from the layer's fp32 parameters in one
 * launch (the LayerNorm affines g2/be2 -- the feed-forward's norm -- and g1/be1 -- the NEXT layer's attention norm -- are
 * folded in).  Head parameters (wout .. b2) NULL: tail-only stream; tail parameters (g1 .. bv) NULL: head-only.
 * wpack: (weights + 32 768) bf16, vec: 2048 fp32. */
int wmz_layer_fused_pack(const float* wout, const float* bout, const float* g2, const float* be2, const float* w1,
                         const float* b1, const float* w2, const float* b2, const float* g1, const float* be1,
                         const float* wq, const float* wk, const float* wv, const float* bv, void* wpack, float* vec,
                         int D, int I, int M, void* stream);

/* Every weight stream / vector block of a model in two launches (training repacks them after each optimizer step).
 * block_rows: DEVICE array of nblk + 1 rows of eleven 64-bit fields { w, rs, ks, N, K, gn, gk, gamma, rgamma, dst, start8 }:
 * block element (f, k) = w[f * rs + k * ks] (* gamma[k]) (* rgamma[f]) for N output features x K contraction indices,
 * ownership groups gn / gk (forward streams: gn = N, gk = K; backward streams: see wmz_layer_fused_bwd_pack), written as
 * bf16 to dst in the kernels' piece order; start8 = index of the block's first 8-element group in the launch (row nblk:
 * total8).  vec_jobs: DEVICE array of nvec rows of ten pointers { bout, b1, w1, be2, b2, wk, wv, be1, bv, vec } (the
 * inputs and output of wmz_layer_fused_pack's vector block; absent parts NULL).  The streams' zero padding is the caller's
 * (allocate the buffers zeroed once).  world_modelz_amd/fused.py::PackSet builds the tables. */
int wmz_fused_pack_table(const void* block_rows, int nblk, long total8, const void* vec_jobs, int nvec, int D, int I, int M,
                         void* stream);

/* Training forward on the same kernels (replaces the five per-op GEMM launches per layer of the training forward).
 * Besides the inference outputs they write what the backward (wmz_linear_wgrad, wmz_layernorm_bwd, wmz_local3d_attn_bwd
 * ..) reads, row-major: x1_out [ntok, D] = the feed-forward block's input (x + to_out(o)), x_out_rowmajor [ntok, D] = a
 * row-major copy of x_out when x_out itself is tiled (NULL otherwise), and kv_out as ONE [ntok, 2I] buffer (k | v column
 * halves).  z_tiled_out (optional, needs the head and ntok % 32 == 0): the feed-forward pre-activation W1 LN(x1) + b1 as
 * bf16 in the private tiled layout wmz_ff_fused_bwd reads (per 32-token tile [M/32 chunks][2][64 lanes][8]); NULL: not
 * exported, the per-op backward recomputes it with one LayerNorm-GEMM.
 * ln_ff_stats / ln_attn_stats (optional, fp32 [2, ntok]: means then reciprocal standard deviations): the statistics of the
 * two LayerNorms the kernel applies -- in front of the feed-forward, and in front of the next layer's k | v -- for the
 * backward (wmz_linear_wgrad's LayerNorm prologue, wmz_linear_fwd_stats), which otherwise spends a pass per LayerNorm. */
int wmz_layer_fused_fwd_train(const void* o, const void* x, void* x_out, void* x_out_rowmajor, void* x1_out, void* q_out,
                              void* kv_out, float* ln_ff_stats, float* ln_attn_stats, void* z_tiled_out, const void* wpack,
                              const float* vec, int ntok, int D, int I, int M, int has_head, int has_tail, int xflags,
                              float eps, void* stream);
int wmz_embed_qkv_fused_fwd_train(const int64_t* z, const float* emb, const float* pos_s, const float* pos_h,
                                  const float* pos_w, void* x_out, void* x_out_rowmajor, void* q_out, void* kv_out,
                                  float* ln_attn_stats, const void* wpack, const float* vec, int B, int S, int H, int W,
                                  int D, int I, int M, int num_classes, int xflags, float eps, void* stream);

/* Fused per-token BACKWARD (bf16, dim 256 / inner 128 / mlp 256; csrc/layer_fused_bwd.hip): the counterpart of
 * wmz_layer_fused_fwd_train.  Per layer, in backward order:
 *   wmz_ff_fused_bwd    dy [ntok, D] (gradient w.r.t. the feed-forward block's output), the tiled pre-activation, the
 *                       block's input x1 and its LayerNorm statistics ->
 *                         g_out  = GELU(z)                 [ntok, M]   operand of dW2 = dy^T g
 *                         dz_out = (dy W2) GELU'(z)        [ntok, M]   operand of dW1
 *                         xhat_out = (x1 - mean) rstd      [ntok, D]   operand of dW1; NULL: `x1` already HOLDS these rows
 *                                                                      (forward flag WMZ_FUSED_X1_NORMALISED) -- nothing is
 *                                                                      recomputed or written, the forward's tensor is the operand
 *                         dx1_out = dy + LNbwd(dz W1')     [ntok, D]   gradient w.r.t. x1 (= to_out's output + residual)
 *                         do_out  = dx1 Wout               [ntok, I]   gradient w.r.t. the attention output
 *   wmz_local3d_attn_bwd  (do -> dq, dk | dv)
 *   wmz_qkv_fused_bwd   dq [ntok, I], dk | dv [ntok, 2I], the layer's input x + statistics, res = dx1 ->
 *                         dx = res + dq Wq + LNbwd(dk Wk' + dv Wv')      gradient w.r.t. the layer's input
 *                         xhat_out = (x - mean) rstd                      operand of the to_k | to_v weight gradient; NULL: `x`
 *                                                                         already HOLDS these rows (WMZ_FUSED_XRM_NORMALISED)
 * wmz_ff_fused_bwd with dy_last_planes = S > 0 (the LAST layer under a last-frame loss, main.py:37): dy holds only the
 * clips' last planes, [ntok / S, D] with dy_plane_tokens rows per clip; every other token's gradient is zero and is read
 * from zero_row (D bf16 zeros) -- no [ntok, D] tensor of zeros is written or read.  0 / 0 / NULL: dy is [ntok, D].
 * wpack: the TRANSPOSED weight streams of wmz_layer_fused_bwd_pack (wpack_ff: 163 840 + 32 768 bf16, wpack_qkv: 98 304 +
 * 32 768 bf16; the LayerNorm gammas are folded in).  Replaces wmz_linear_fwd x5 (dgrad / recompute) and wmz_layernorm_bwd
 * x2 per layer; the weight gradients stay wmz_linear_wgrad calls on the operands written here, computed against the
 * NORMALISED inputs, and wmz_ln_affine_grads turns such a raw gradient G[N, K] = dC^T xhat and s[N] = column sums of dC
 * (ntok must be a multiple of 32 for both kernels) into the parameter gradients: dW += G diag(gamma) + s beta^T, dbias[n - bias_from] += s[n] (n >= bias_from; NULL: no
 * bias), dgamma[k] += sum_n W[n,k] G[n,k], dbeta[k] += sum_n W[n,k] s[n]  (all fp32, accumulating). */
int wmz_layer_fused_bwd_pack(const float* wq, const float* wk, const float* wv, const float* g1, const float* wout,
                             const float* w1, const float* g2, const float* w2, void* wpack_qkv, void* wpack_ff, int D, int I,
                             int M, void* stream);
int wmz_ff_fused_bwd(const void* dy, const void* z_tiled, const void* x1, const float* ln_stats, void* g_out, void* dz_out,
                     void* xhat_out, void* dx1_out, void* do_out, const void* wpack, int ntok, int D, int I, int M,
                     int dy_last_planes, int dy_plane_tokens, const void* zero_row, void* stream);
int wmz_qkv_fused_bwd(const void* dq, long lddq, const void* dkv, long lddkv, const void* x, const float* ln_stats,
                      const void* res, void* dx, void* xhat_out, const void* wpack, int ntok, int D, int I, void* stream);
int wmz_ln_affine_grads(const float* G, const float* s, const float* W, const float* gamma, const float* beta, float* dW,
                        float* dbias, float* dgamma, float* dbeta, int N, int K, int bias_from, void* stream);
/* n <= 4 such conversions by one launch (HOST tables of n entries; dbias[i] may be NULL): a layer's two -- the feed-forward's
 * W1 and the k | v projection -- are each a grid of a few latency-bound workgroups. */
int wmz_ln_affine_grads_batch(int n, const float* const* G, const float* const* s, const float* const* W,
                              const float* const* gamma, const float* const* beta, float* const* dW, float* const* dbias,
                              float* const* dgamma, float* const* dbeta, const int* N, const int* K, const int* bias_from,
                              void* stream);

/* ---- conv encoder / decoder (autoencoder.py:8-152), NHWC, implicit GEMM on MFMA ----
 * out[b,ho,wo,co] = act( (conv(x, w)[..] + bias[co]) * scale[co] + shift[co] + residual ), w as [Cout, KH, KW, Cin]
 * (nn.Conv2d weight permuted (0,2,3,1)), Cin % 8 == 0 (zero-pad), act = LeakyReLU(slope) if leaky.  scale/shift carry a
 * folded eval-mode BatchNorm (autoencoder.py:21-25).  stat_sum / stat_sq (optional, fp32 [WMZ_STAT_REPLICAS][Cout],
 * accumulated; zero them first) receive per-channel sum and sum of squares of the stored output: the batch statistics of a
 * training-mode BatchNorm.  REPLICATED: a workgroup adds into replica (its index mod WMZ_STAT_REPLICAS) and the consumer
 * (wmz_bn_finalize) sums the replicas -- thousands of workgroups adding into ONE row is a chain of same-address atomics that
 * complete ~15 ns apart, 60-120 us behind a 70-200 us convolution. */
#define WMZ_STAT_REPLICAS 8
int wmz_conv2d_nhwc_fwd(const void* x, const void* w, void* out, const float* bias, const float* scale,
                        const float* shift, const void* residual, float* stat_sum, float* stat_sq, int B, int Hi, int Wi,
                        int Cin, int Cout, int KH, int KW, int stride, int pad, int leaky, float slope, int dtype,
                        void* stream);
/* The same with an INPUT prologue for 1x1 convolutions (KH = KW = 1, pad = 0): the A operand is LeakyReLU(x * in_scale[c]
 * + in_shift[c]) (slope in_slope), applied while the slab is staged -- the training-mode BatchNorm + activation in front
 * of the conv (autoencoder.py:21-25 Residual: conv3x3 -> BN -> LeakyReLU -> conv1x1) without a pass of its own. */
int wmz_conv2d_nhwc_fwd_pre(const void* x, const void* w, void* out, const float* bias, const float* scale,
                            const float* shift, const void* residual, float* stat_sum, float* stat_sq,
                            const float* in_scale, const float* in_shift, float in_slope, int B, int Hi, int Wi, int Cin,
                            int Cout, int KH, int KW, int stride, int pad, int leaky, float slope, int dtype, void* stream);
/* Direct 3x3 / stride 1 / pad 1 convolution in bf16 (csrc/conv_direct.hip; autoencoder.py:8-10 conv3x3 inside Residual :18-42,
 * UpscaleResidual :89-131 and the decoder's first / last convolutions :134-152): the same result as wmz_conv2d_nhwc_fwd on
 * these shapes (same k order, same epilogue arithmetic) with the haloed input patch and the weight stream staged by LDS-DMA.
 * wpack: the GEMM operand [Cout, 9 * Cin] re-ordered by wmz_conv3x3_direct_pack (wmz_conv3x3_direct_pack_elems(Cin, Cout) bf16
 * elements).  Shapes: wmz_conv3x3_direct_supported(H, W, Cin, Cout) != 0 -- Cin % 64 == 0, Cout % 8 == 0 and <= 128, W a
 * multiple of 32 with H % 8 == 0, or W = 16 with H % 16 == 0. */
int wmz_conv3x3_direct_supported(int H, int W, int Cin, int Cout);
long wmz_conv3x3_direct_pack_elems(int Cin, int Cout);
int wmz_conv3x3_direct_pack(const void* w_op, void* wpack, int Cin, int Cout, void* stream);
int wmz_conv3x3_direct_fwd(const void* x, const void* wpack, void* out, const float* bias, const float* scale,
                           const float* shift, const void* residual, float* stat_sum, float* stat_sq, int B, int H, int W,
                           int Cin, int Cout, int leaky, float slope, void* stream);
/* The same at stride 2 (pad 1; autoencoder.py:27-33, the first convolution of the down-sampling Residual): H, W the (even) INPUT
 * plane, the output H / 2 x W / 2; Cout = 128, output planes of 8 k x 16 m pixels; same packed weight stream.  stride == 1 is
 * wmz_conv3x3_direct_fwd. */
int wmz_conv3x3_direct_supported_strided(int H, int W, int Cin, int Cout, int stride);
int wmz_conv3x3_direct_fwd_strided(const void* x, const void* wpack, void* out, const float* bias, const float* scale,
                                   const float* shift, const void* residual, float* stat_sum, float* stat_sq, int B, int H, int W,
                                   int Cin, int Cout, int stride, int leaky, float slope, void* stream);
/* Small-K convolutions in bf16 (csrc/conv_point.hip; K = KH KW Cin <= 256: the 1x1 convolutions of Residual :18-42 and
 * UpscaleResidual :89-131, the 2x2 / stride 2 down-sampling convolution :29-33, the 3-channel conv_1 :60-86): the same result as
 * wmz_conv2d_nhwc_fwd_pre without a residual (same k order and epilogue arithmetic), as a persistent streaming kernel -- weights
 * resident in LDS, every wave on its own runs of 64 output pixels.  wpack: the GEMM operand [Cout, K] re-ordered by
 * wmz_conv_point_pack (wmz_conv_point_pack_elems(K, Cout) bf16 elements).  Shapes: wmz_conv_point_supported(...) != 0 -- Cin % 8
 * == 0, Cout % 8 == 0 and <= 128, pad <= 1, B Ho Wo a multiple of 64. */
int wmz_conv_point_supported(int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad);
long wmz_conv_point_pack_elems(int K, int Cout);
int wmz_conv_point_pack(const void* w_op, void* wpack, int K, int Cout, void* stream);
int wmz_conv_point_fwd(const void* x, const void* wpack, void* out, const float* bias, const float* scale, const float* shift,
                       float* stat_sum, float* stat_sq, const float* in_scale, const float* in_shift, float in_slope, int B, int Hi,
                       int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad, int leaky, float slope, void* stream);
/* A training-mode nn.BatchNorm2d (autoencoder.py:21-25) handed to the kernel that APPLIES it as its raw batch statistics: that kernel
 * does what wmz_bn_finalize does -- same arithmetic -- while it sets up its per-channel constants, and one of its workgroups moves the
 * running statistics (momentum, unbiased variance), increments *num_batches_tracked and writes the optional outputs (scale / shift /
 * mean / rstd of this batch, for the backward pass): no wmz_bn_finalize launch between a convolution and its consumer.  A HOST
 * struct of DEVICE pointers, read at call time.  sum / sq: [WMZ_STAT_REPLICAS][C] as the producer left them; count = B Ho Wo. */
typedef struct wmz_bn_stats {
  const float* sum; const float* sq;
  const float* gamma; const float* beta;            /* NULL: 1 / 0 */
  float* running_mean; float* running_var;          /* both or neither */
  int64_t* num_batches_tracked;                     /* or NULL */
  float* scale; float* shift;                       /* out [C], both or neither */
  float* mean; float* rstd;                         /* out [C], both or neither */
  double count, momentum, eps;
} wmz_bn_stats;
/* wmz_conv_point_fwd with the input prologue's BatchNorm given as raw statistics (in_bn; then in_scale = in_shift = NULL). */
int wmz_conv_point_fwd_bn(const void* x, const void* wpack, void* out, const float* bias, const float* scale, const float* shift,
                          float* stat_sum, float* stat_sq, const float* in_scale, const float* in_shift, const wmz_bn_stats* in_bn,
                          float in_slope, int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad, int leaky,
                          float slope, void* stream);
/* logical NCHW frames (contiguous; in_dtype) -> NHWC with the channels zero-padded to a multiple of 8 (out_dtype): the layout
 * flip in front of the encoder's first convolution (autoencoder.py:83; the reference's modules take NCHW) as one pass. */
int wmz_nchw_to_nhwc8(const void* x, void* y, int B, int C, int H, int W, int in_dtype, int out_dtype, void* stream);
/* per-channel sum / sum of squares of an NHWC tensor viewed as [M, C] (accumulated into fp32 [WMZ_STAT_REPLICAS][C]). */
int wmz_channel_stats_nhwc(const void* x, long M, int C, float* sum, float* sq, int dtype, void* stream);
/* nn.BatchNorm2d bookkeeping: training != 0: batch mean / biased var from (sum, sq: [WMZ_STAT_REPLICAS][C], summed here; count), running stats updated with
 * `momentum` and the unbiased variance, *num_batches_tracked (optional, int64) incremented; training == 0: running stats.
 * Emits scale = gamma*rstd, shift = beta - mean*scale. */
int wmz_bn_finalize(const float* sum, const float* sq, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, double momentum, double eps, int training, float* scale,
                    float* shift, float* mean_out, float* rstd_out, int C, int64_t* num_batches_tracked, void* stream);
/* y = act(a*sa + ta (+ b*sb + tb)) per channel on NHWC [M, C] (BatchNorm apply, skip add, LeakyReLU). */
int wmz_affine_act_nhwc(const void* a, const float* sa, const float* ta, const void* b, const float* sb, const float* tb,
                        void* y, long M, int C, int leaky, float slope, int dtype, void* stream);
/* The same with either operand's affine given as a training-mode BatchNorm's raw statistics (wmz_bn_stats; bna replaces sa / ta,
 * bnb replaces sb / tb; NULL: the plain call).  Needs the 16-byte kernel: wmz_affine_act_bn_supported(C, dtype) != 0 and 16-byte
 * aligned tensors. */
int wmz_affine_act_bn_supported(int C, int dtype);
int wmz_affine_act_nhwc_bn(const void* a, const float* sa, const float* ta, const wmz_bn_stats* bna, const void* b, const float* sb,
                           const float* tb, const wmz_bn_stats* bnb, void* y, long M, int C, int leaky, float slope, int dtype,
                           void* stream);
/* dz [B, Hz, Wz, C] = dy [B, Ho, Wo, C] placed at every `stride`-th pixel (dz[b, s y, s x] = dy[b, y, x]), zero elsewhere: the
 * zero-inserted plane on which the data gradient of a strided nn.Conv2d (autoencoder.py:27-33) runs as a stride-1 convolution.
 * C a multiple of 8 (bf16) / 4 (fp32), 16-byte aligned tensors. */
int wmz_dilate_nhwc(const void* dy, void* dz, int B, int Ho, int Wo, int C, int Hz, int Wz, int stride, int dtype, void* stream);
/* F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) (autoencoder.py:138) on NHWC. */
int wmz_bilinear2x_nhwc(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);

/* ---- backward of the conv path (VQ-AE training, train_vqae.py:125-192; the reference gets these from autograd) ----
 * data gradient: wmz_conv2d_nhwc_fwd on the (zero-dilated for stride 2) output gradient with flipped, transposed weights.
 * weight gradient: dW[Cout, KH*KW*Cin] += dy^T . im2col(x) (implicit), dbias[Cout] += colsum(dy); fp32, accumulated. */
int wmz_conv2d_nhwc_wgrad(const void* x, const void* dy, float* dW, float* dbias, int B, int Hi, int Wi, int Cin, int Cout,
                          int KH, int KW, int stride, int pad, int dtype, void* stream);
/* ... by the two-stage reduction of wmz_linear_wgrad_ws (caller-owned workspace of at least
 * wmz_conv2d_nhwc_wgrad_workspace_floats(...) floats; deterministic, no float atomics; overwrite != 0: dW / dbias are stored,
 * not accumulated -- no zero fill needed).  conv_layout_co > 0: dW is nn.Conv2d's own weight (gradient) tensor
 * [conv_layout_co, conv_layout_ci, KH, KW] -- the channel padding of the operands cropped, the taps transposed -- and dbias has
 * conv_layout_co entries: the gradient lands in the parameter's .grad without a permute / copy / add pass. */
long wmz_conv2d_nhwc_wgrad_workspace_floats(int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                            int dtype);
int wmz_conv2d_nhwc_wgrad_ws(const void* x, const void* dy, float* dW, float* dbias, int B, int Hi, int Wi, int Cin, int Cout,
                             int KH, int KW, int stride, int pad, int overwrite, int conv_layout_co, int conv_layout_ci,
                             float* workspace, long workspace_floats, int dtype, void* stream);
/* n <= 6 such weight gradients by ONE launch pair (HOST tables of n entries; dbias[i] may be NULL; workspace: the sum of the
 * problems' wmz_conv2d_nhwc_wgrad_workspace_floats suffices): the small layers of a VQ-AE training step, whose launches sit on a
 * side branch of the step's hipGraph.  wmz_conv2d_nhwc_wgrad_is_direct(...) != 0: the layer has a kernel of its own (3x3 / stride 1
 * on the direct weight-gradient kernel) and is better launched alone through wmz_conv2d_nhwc_wgrad_ws. */
int wmz_conv2d_nhwc_wgrad_batch(int n, const void* const* x, const void* const* dy, float* const* dW, float* const* dbias,
                                const int* B, const int* Hi, const int* Wi, const int* Cin, const int* Cout, const int* KH,
                                const int* KW, const int* stride, const int* pad, const int* overwrite, const int* conv_layout_co,
                                const int* conv_layout_ci, float* workspace, long workspace_floats, int dtype, void* stream);
int wmz_conv2d_nhwc_wgrad_is_direct(int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad, int dtype);
/* training-mode BatchNorm + LeakyReLU backward, pass 1: g = dy * act'(y) (optional g_out), sum_g[C] += g,
 * sum_gx[C] += g * (x - mean) * rstd;  pass 2: dx = gamma*rstd*(g - sum_g/M - xhat*sum_gx/M)  (dgamma = sum_gx, dbeta = sum_g). */
int wmz_bn_act_bwd_reduce(const void* x, const void* y, const void* dy, const float* mean, const float* rstd, void* g_out,
                          float* sum_g, float* sum_gx, long M, int C, int leaky, float slope, int dtype, void* stream);
int wmz_bn_bwd_apply(const void* x, const void* g, const float* mean, const float* rstd, const float* gamma,
                     const float* sum_g, const float* sum_gx, void* dx, long M, int C, int dtype, void* stream);
/* Both of the above for y = LeakyReLU(BatchNorm_train(x)) WITHOUT a skip input (autoencoder.py:21-25 Residual's first
 * normalisation, :100-118 UpscaleResidual's two): the LeakyReLU mask is recomputed from x and the forward's (scale, shift) -- the
 * sign of fmaf(x, scale, shift), what wmz_affine_act_nhwc evaluated -- so the stored output is not read and g = dy * act'(y) is not
 * written: 5 tensor passes instead of 7.  dy: the gradient behind the activation; sum_g / sum_gx: fp32 [C], ZERO on entry,
 * receive dbeta / dgamma; dx = dL/dx.  Shapes: wmz_bn_leaky_bwd_supported(C, dtype) != 0 (the 16-byte kernels). */
int wmz_bn_leaky_bwd_supported(int C, int dtype);
int wmz_bn_leaky_bwd(const void* x, const void* dy, const float* scale, const float* shift, const float* mean,
                     const float* rstd, const float* gamma, float* sum_g, float* sum_gx, const void* add, void* dx, long M, int C,
                     float slope, int dtype, void* stream);
/* wmz_bn_bwd_apply with dx += add (optional, [M, C] in the activations' dtype): the gradient x receives from its other consumer --
 * a residual block's skip path (autoencoder.py:35-42, :119-131) -- summed by this pass instead of one of its own (the reference's
 * autograd runs an add kernel there); wmz_bn_leaky_bwd's `add` is the same. */
int wmz_bn_bwd_apply_add(const void* x, const void* g, const float* mean, const float* rstd, const float* gamma,
                         const float* sum_g, const float* sum_gx, const void* add, void* dx, long M, int C, int dtype,
                         void* stream);
/* adjoint of wmz_bilinear2x_nhwc (gather form, deterministic): dy [B,2H,2W,C] -> dx [B,H,W,C]. */
int wmz_bilinear2x_nhwc_bwd(const void* dy, void* dx, int B, int H, int W, int C, int dtype, void* stream);

/* ---- the steps either side of the model in the training loop (main.py:246-274) ----
 * corruption of the last latent frame: out[b, p] = C (mask token) with probability r[b]; else a uniformly redrawn code with
 * probability 0.1 r[b]; else z_last[b, p]  (== multinomial(lerp(one_hot, 1/C, 0.1 r)) then mask: same law, in-kernel
 * Philox4x32-10 keyed by (seed, stream_id)).  z_last / out: int64, clip strides in elements; target (optional) = z_last copy. */
int wmz_corrupt_tokens(const int64_t* z_last, long clip_stride, const float* r, int64_t* out, long out_stride,
                       int64_t* target, int B, int HW, int C, unsigned long long seed, unsigned long long stream_id,
                       void* stream);
/* The same with the low 40 bits of the stream id read from device memory when the kernel runs (`counter`, one uint64 the
 * caller advances between launches; stream_hi supplies the bits above, e.g. the data-parallel rank << 40): the launch
 * can be captured in a hipGraph and still draw a fresh corruption on every replay. */
int wmz_corrupt_tokens_dev(const int64_t* z_last, long clip_stride, const float* r, int64_t* out, long out_stride,
                           int64_t* target, int B, int HW, int C, unsigned long long seed, unsigned long long stream_hi,
                           const unsigned long long* counter, void* stream);

/* Config 5's step prologue as one launch (minecraft/sparse_diffusion.py:44-72 sample_time_dependent, :437 gather, :440-449
 * perturbation & masking): per clip b a frame window whose width grows with the noise level r[b] (placed by o[b] in [0, 1), or
 * uniformly by the in-kernel generator when o is NULL), n distinct positions uniform inside it (random order), the clip's tokens
 * there and their corruption by the law of wmz_corrupt_tokens with redraw probability r * p_uniform (training: p_uniform = 0.1,
 * sparse_diffusion.py:447; the sampler, :185-187, masks only: 0).  z [B, S * HW] int64 (clip stride clip_stride); indices / tokens /
 * target [B, n] int64.  Philox keyed by (seed, stream_id); with counter != NULL the low 40 bits of the stream id are read from
 * device memory at run time (hipGraph replays).  wmz_sparse_draw_context_supported: 1 when S * HW <= 65536 positions, n <= 512 and
 * 2 ceil(n / HW) <= S (the reference's precondition: its narrowest window holds n positions); else the host draws with torch ops (world_modelz_amd/sparse_diffusion.py). */
int wmz_sparse_draw_context_supported(int S, int HW, int n);
int wmz_sparse_draw_context(const int64_t* z, long clip_stride, const float* r, const float* o, int64_t* indices,
                            int64_t* tokens, int64_t* target, int B, int S, int HW, int n, int C, float p_uniform,
                            unsigned long long seed, unsigned long long stream_id, const unsigned long long* counter,
                            void* stream);
/* The draw of config 5's sampler (sparse_diffusion.py:190-197: softmax -> multinomial -> scatter_ into the clip): one class per
 * row of fp32 logits [R, C] (row stride ld) drawn from softmax(logits) by inverse CDF in class order; with z != NULL written to
 * z[row / rows_per_clip][indices[row]] (clip stride clip_stride), with samples != NULL also to samples[row].  Philox keyed as
 * wmz_sparse_draw_context (its own domain of the stream id). */
int wmz_categorical_scatter(const float* logits, long ld, long R, int C, const int64_t* indices, int64_t* z, long clip_stride,
                            long rows_per_clip, int64_t* samples, unsigned long long seed, unsigned long long stream_id,
                            const unsigned long long* counter, void* stream);
/* One step of the sampler loop between two forward passes (main.py:76-104): per row of fp32 logits [R, C <= 2048] keep the
 * top_k largest (<= 0: all; ties with the k-th kept), softmax, draw a class by the inverse CDF from one in-kernel uniform,
 * re-mask with a second one: the position gets mask_token where u2 > alpha (and, with `last_mask` [R] bytes, in / out, only
 * where the previous call masked: consistent masking).  denoised[R] receives the draws, out_tokens[(r / rows_per_block) *
 * block_stride + r % rows_per_block] the (re-masked) tokens -- the last frame of a [B, S, H, W] grid in place.
 * alpha = alphas[*counter % n_alpha] and the Philox stream id = *counter are read from DEVICE memory (the caller advances
 * the counter): the call sits in a hipGraph with the forward pass it feeds. */
int wmz_sample_tokens_dev(const float* logits, long ld, int R, int C, int top_k, const float* alphas, int n_alpha,
                          int64_t mask_token, int64_t* out_tokens, long rows_per_block, long block_stride, int64_t* denoised,
                          unsigned char* last_mask, unsigned long long seed, const long long* counter, void* stream);
/* CrossEntropyLoss(reduction='none') over fp32 logits [R, C] (row stride ld): loss[R], lse[R]; and its gradient
 * dlogits[r,c] = (softmax - one_hot) * grad_rows[r], written in `dtype` (the GEMM operand type of the backward). */
int wmz_ce_fwd(const float* logits, long ld, const int64_t* target, float* loss, float* lse, long R, int C, void* stream);
int wmz_ce_bwd(const float* logits, long ld, const int64_t* target, const float* lse, const float* grad_rows, void* dlogits,
               long R, int C, int dtype, void* stream);
/* Both in one pass over the logits (a wave keeps its row in registers; C <= 8192, else the two launches above): what the
 * training step's fused linear + cross-entropy calls. */
int wmz_ce_fwd_bwd(const float* logits, long ld, const int64_t* target, float* loss, float* lse, const float* grad_rows,
                   void* dlogits, long R, int C, int dtype, void* stream);

/* ---- training-step tail over flat fp32 arenas (one launch each) ----
 * grad_norm (main.py:188-193): out[0] += scale^2 * sum g^2 (caller zeroes out[0]; no host sync). */
int wmz_grad_sqnorm(const float* g, long n, float scale, float* out, void* stream);
/* torch.optim.AdamW step as configured at main.py:433 (decoupled weight decay, amsgrad off), gradients read as
 * grad_scale * g (1/world after a SUM all-reduce); `step` is the 1-based step count for the bias corrections. */
int wmz_adamw_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                   double eps, double weight_decay, long step, double grad_scale, void* stream);
/* The same with the per-step scalars in device memory, hyper = [lr, 1 - beta1^t, sqrt(1 - beta2^t)] (fp32): a captured
 * training step is replayed with a new learning rate / bias correction each time.  sqnorm_out (optional): += the squared
 * gradient norm sum (grad_scale g)^2, computed in the same pass (zero it first). */
int wmz_adamw_step_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, double beta1, double beta2,
                       double eps, double weight_decay, double grad_scale, float* sqnorm_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WMZ_H_ */
