/* avt_hip.h -- C ABI of libavt_hip.so: the MI355X (gfx950) kernels behind the AVT training hot path.
 *
 * This is the drop-in boundary.  The reference (facebookresearch/AVT) is pure Python: its "FFI" for this path is
 * the set of torch.nn calls its modules make.  Each entry point below names the reference call it replaces
 * (file:line in the upstream checkout; [timm]/[hf] = the un-vendored timm==0.4.12 / transformers==4.2.2 code those
 * lines dispatch into).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain C: raw device pointers, explicit dims / leading dimensions (in ELEMENTS), scalar hyper-parameters and an
 *     explicit hipStream_t (passed as void*).  No torch types.  Nothing is allocated or retained by the library;
 *     kernels are enqueued on `stream` and the call returns immediately (never synchronises).
 *   - activations and activation gradients are bf16 (raw uint16 bits); statistics, losses, parameters' master copies,
 *     parameter gradients and optimizer state are fp32.
 *   - return 0 on success; non-zero on failure, in which case avt_last_error() (thread-local) describes it.
 *     Shape / alignment violations are rejected on the host before any launch.
 *   - re-entrant, no dependence on the thread's "current device/stream" (forward runs on the Python main thread,
 *     backward on the autograd thread -- SURVEY 8b).
 *   - parameter-gradient outputs ACCUMULATE (fp32 atomics): the caller keeps the gradient buffer zeroed between
 *     steps (avt_sgd_step can re-zero it while it consumes it).
 */
#ifndef AVT_HIP_H_
#define AVT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVT_ABI_VERSION 9

const char* avt_last_error(void);
int avt_abi_version(void);

/* ---- GEMM -------------------------------------------------------------------------------------------------------
 * C[M,N] = epilogue( sum_k opA[m,k] * opB[n,k] ), bf16 inputs, fp32 accumulate (v_mfma_f32_32x32x16_bf16).
 *   a_kmajor = 1: A stored [M][K] (row stride lda);  0: A stored [K][M] (the reduction index is the row index).
 *   b_kmajor = 1: B stored [N][K] (row stride ldb);  0: B stored [K][N].
 * Replaces every torch.nn.Linear / HF Conv1D matmul of the step and their autograd backward:
 *   ViT qkv/proj/fc1/fc2 [timm] (models/video_classification.py:224), patch-embed Conv2d as a GEMM over avt_im2col
 *   rows, AVT-h encoder/decoder (models/future_prediction.py:80-81,163,190), GPT-2 c_attn/c_proj/c_fc [hf]
 *   (models/future_prediction.py:178-181), classifier (models/base_model.py:203-216, 222-238).
 * out_mode 0: bf16 C, 1: fp32 C, with the fused epilogue, applied in this order:
 *     v = acc + bias[n];  act 3: v *= aux[m,n] (backward of an activation whose derivative was saved);
 *     act 1|2: C2[m,n] = gelu_erf'|gelu_tanh'(v) (optional, saved for backward), v = gelu_erf|gelu_tanh(v);
 *     act 0 with C2: C2[m,n] = v;  dropout(drop_p, drop_seed, element index m*N+n);
 *     v += res[(res_period ? m % res_period : m), n];  colsum[n] += v (over the bf16-rounded values when C is bf16; see "partials" below);  C[m,n] = v
 * out_mode 2: C (fp32) += acc with atomics, no epilogue; splitk > 1 splits the reduction over workgroups
 *             (splitk <= 0 picks a factor that fills the chip).  Used for weight gradients.
 * tile: 0 = choose; 32 (M <= 64) | 64 | 128 | 256 | 808 (256x256 tile, 8-phase schedule, one tile per workgroup) | 809 (the same schedule as a persistent
 * kernel: one workgroup per CU draws tiles from a queue and fetches the next tile's first K tile under its epilogue; bit-identical to 808;
 * k-major operands, N % 256 == 0, K % 128 == 0, >= 512 tiles, bf16 output, bias | GELU | bias + residual | saved-derivative epilogue --
 * an error otherwise; the automatic choice takes it for K <= 4096) force a kernel.  The automatic choice walks the
 * tiles of an activation GEMM whose B operand exceeds an XCD's 4-MB L2 (N*K*2 > 4 MB, e.g. the fc1 weight) in column strips, so that
 * the strip of B stays L2-resident (results do not depend on the tile order).  Small outputs (fewer than 200 tiles of 256 x 256; the reference's own
 * 3 clips per GPU, expts/01_ek100_avt.txt:5): all-k-major contractions take the 8-phase kernel from 96 such tiles and 64 x 64 tiles with a 3-deep ring
 * below that when K <= 3072; everything else 128 x 128 tiles (>= 192 of them) or 64 x 64; at most 64 output rows of k-major A rows (round 6; the head at
 * 3 clips per GPU: 30 rows at T = 10, 45 at T = 15): tile 32, the skinny kernel -- N / 32 workgroups, each one ordered MFMA chain fed by an 18-stage LDS-DMA ring (a weight stream: HBM-bound;
 * same bits as every other tile).  Requirements: 16-B aligned pointers, lda/ldb % 8 == 0,
 * K % 8 == 0 when an operand is k-major, N % 4 == 0 and ldc % 4 == 0 for out_mode 0/1. */
int avt_gemm_bf16(const void* A, int a_kmajor, int lda, const void* B, int b_kmajor, int ldb,
                  void* C, int ldc, int M, int N, int K,
                  const float* bias, int act, const void* aux, int ldaux,
                  void* C2, int ldc2, const void* res, int ldres, int res_period,
                  float drop_p, uint64_t drop_seed, float* colsum,
                  int out_mode, int splitk, int tile, float* partials, size_t partials_bytes, void* stream);
size_t avt_gemm_colsum_workspace_bytes(int M, int N, int tile);
/* Fragment-major private tensors (ABI 7).  The derivative that fc1 forward saves (C2 of act 1) is read exactly once, by the fc2 data gradient (aux of
 * act 3) -- a launch with the same M and N on the same kernel: [timm] Mlp.fc1 + GELU and its autograd backward (models/video_classification.py:224).
 * Such a tensor needs no row-major form.  ldc2 == 0 (with C2 != NULL, act 1) writes it, and ldaux == 0 (with aux != NULL, act 3) reads it, in the
 * persistent kernel's own order: per (128-row strip, 64-column group) four 4-KB blocks of [4 stores][64 lanes][16 bytes] in the MFMA accumulator
 * layout (csrc/gemm_persist.hip; avt_amd/ops.py::gemm_frag_unpack restates it for the tests).  The buffer holds avt_gemm_frag_bytes(M, N) bytes
 * (rows padded to 128); the values are bit-identical to those of the row-major form.  Only the persistent 8-phase kernel knows the layout:
 * avt_gemm_frag_ok(M, N, K) says whether a call of this shape (contiguous k-major operands, tile 0) lands there; a call that does not is an error. */
size_t avt_gemm_frag_bytes(int M, int N);
int avt_gemm_frag_ok(int M, int N, int K);

/* ---- "partials": run-to-run identical parameter gradients ----------------------------------------------------------
 * Every entry point that folds many workgroups into one fp32 vector (colsum of avt_gemm_bf16, dgamma / dbeta / colsum of
 * avt_layernorm_bwd, dbias of avt_vit_attn_bwd, avt_colsum_bf16, avt_patch_embed_bwd_reduce) takes a `partials` workspace
 * (16-byte aligned fp32, `partials_bytes` long; the matching *_workspace_bytes query gives an upper bound).  NULL: the
 * workgroups merge with fp32 atomics, so the last bits of the result depend on arrival order.  Non-NULL: every workgroup stores
 * its partial vector, and a second kernel adds them in an order that depends on the problem shape only, then accumulates into
 * the destination -- two runs on the same inputs give the same bits.  The workspace is scratch: it may be shared by all calls
 * on one stream.  (PyTorch's own autograd uses atomics for none of these reductions either.) */

/* Deterministic weight-gradient accumulate: C[M,N] (fp32) += sum_k A[k,m] * B[k,n] with BOTH operands stored reduction-index-
 * major (dW = dy^T x of a Linear, x^T dy of an HF Conv1D).  Same kernels as out_mode 2, but every (split, tile) workgroup
 * writes its partial tile to its own slab of `workspace` and a second kernel adds the slabs in split order into C: no atomics,
 * bit-reproducible, and cheaper than the atomics (66 MB of full-line stores + one pass instead of 16 M fp32 atomics for fc1).
 * avt_gemm_accum_workspace_bytes(M, N, K) = bytes the automatic tile / split choice (tile = 0, splitk <= 0) needs.
 * tile: 0 = choose (256x256 output tiles: the 4-wave kernel with 128x128 wave tiles and AGPR accumulators, else 128x128 tiles);
 * 808 = the 8-phase kernel, 2565 = the 4-wave kernel, 128 = small tiles.  Kernels differ in where the reduction is split, so their
 * results differ in the last bits; each kernel is bit-reproducible. */
int avt_gemm_accum_bf16(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                        int splitk, int tile, void* workspace, size_t workspace_bytes, void* stream);
size_t avt_gemm_accum_workspace_bytes(int M, int N, int K);
/* The same product WRITTEN into C (ABI 9): for a caller that knows C holds zeros -- the reference's optimizer.zero_grad() (func/train.py:221), here
 * avt_sgd_step's zero_grad -- the accumulate form reads 4 bytes per weight back for nothing (394 M weights: 1.6 GB per step, which at the reference's
 * 3 clips per GPU is 2 % of it).  Same kernels, same bits as accumulating into zeros; the one-split kernel skips its loads of C, the slab reduce stores
 * instead of adding.  avt_amd/ops.py::gemm_accum takes this form for the first write into a region of a gradient buffer since avt_amd/optim.py::FusedSGD re-zeroed it. */
int avt_gemm_assign_bf16(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                         int splitk, int tile, void* workspace, size_t workspace_bytes, void* stream);

/* ---- LayerNorm ---------------------------------------------------------------------------------------------------
 * [timm] Block.norm1/norm2/VisionTransformer.norm (eps 1e-6); [hf] GPT2 ln_1/ln_2/ln_f (eps 1e-5).
 * fwd: y = (x-mean)*rstd*gamma+beta, rows of D (% 8 == 0, <= 4096) bf16, strided rows allowed; mean/rstd may be NULL.
 * bwd: dx = LN'(dy) [+ dres];  dgamma/dbeta/colsum (fp32 [D], may be NULL) accumulate; colsum = column sums of dx. */
int avt_layernorm_fwd(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy,
                      float* mean, float* rstd, int rows, int D, float eps, void* stream);
int avt_layernorm_bwd(const void* dy, int lddy, const void* x, int ldx, const float* mean, const float* rstd,
                      const float* gamma, const void* dres, int lddres, void* dx, int lddx,
                      float* dgamma, float* dbeta, float* colsum, int rows, int D,
                      float* partials, size_t partials_bytes, void* stream);
size_t avt_layernorm_bwd_workspace_bytes(int rows, int D);

/* ---- LayerNorm folded into the GEMMs around it (round 5) ----------------------------------------------------------------
 * [timm] Block.forward: x + attn(norm1(x)), x + mlp(norm2(x)) (models/video_classification.py:224).  The normalised copy of the
 * residual stream is never written (one read + one write of [tokens, D] per LayerNorm saved):
 *   forward   LN(x) W^T + b = rstd o (x G^T) - (rstd o mean) c^T + b'   with  G = gamma o W (bf16),  c = G 1,  b' = b + W beta
 *   backward  dY' = rstd o dY;  d xhat' = dY' G;  dx = d xhat' - mean_k(d xhat') - xhat o mean_k(d xhat' o xhat) (+ dres);
 *             T = dY'^T x;  dG = T - rowmean_k(T);  dW = gamma o dG + db beta^T;  dgamma = colsum(W o dG);  dbeta = W^T db;  db = colsum(dY)
 * avt_gemm_ln_bf16 = avt_gemm_bf16 (same arguments, same kernels) plus three optional pointers -- both operands k-major, out_mode 0 | 1:
 *   ln_c != NULL  ("fold"; act 0 | 1, no residual / column sums / dropout): A = the UN-normalised rows, B = G, bias = b', ln_c = c [N],
 *                 ln_stat [M][2] = {rstd, -mean * rstd}:  v = rstd[m] * acc + (bias[n] - mean[m] rstd[m] c[n]), then the activation as usual;
 *   ln_c == NULL, ln_stat != NULL ("scale"; act 3 only): ln_stat [M][2] = {rstd, 1 / rstd}: the output rows leave multiplied by rstd[m]
 *                 (dY' for the folded backward) and `colsum` is taken over the UNscaled values (= db);
 *   stat_part != NULL (act 0, bf16 output, bias (+ residual) epilogue, N % 32 == 0): [ceil(N / 64)][M][2][2] fp32, the rows' partial (sum, sum of squares)
 *                 over each 32-column slot of the output (slot s at [s / 2][m][s % 2]) -- the statistics of the NEXT LayerNorm, taken where its input is produced.
 * avt_ln_stats_finalize: stat_part -> stat_fwd [rows][2] = {rstd, -mean * rstd} and stat_bwd [rows][2] = {rstd, 1 / rstd} (either may be NULL).
 * avt_ln_fold_weights: G [N][K] (bf16, row stride ldg), c [N], b' [N] from the fp32 master W [N][K], gamma / beta [K], bias [N] (may be NULL).
 * avt_layernorm_bwd_folded: dx from dy = d xhat' (lddy), x, stat_fwd; dres / colsum as in avt_layernorm_bwd (colsum = column sums of dx).
 * avt_ln_fold_wgrad: T [N][K] fp32 (the raw gradient accumulated by avt_gemm_accum_bf16 into a ZEROED scratch; re-zeroed here), dbias_tmp [N]
 *   (this backward's colsum(dY); re-zeroed here) -> dW [N][K] += , dgamma [K] += , dbeta [K] += , dbias [N] += (may be NULL).  K % 4 == 0, K <= 2048. */
int avt_gemm_ln_bf16(const void* A, int a_kmajor, int lda, const void* B, int b_kmajor, int ldb,
                     void* C, int ldc, int M, int N, int K,
                     const float* bias, int act, const void* aux, int ldaux,
                     void* C2, int ldc2, const void* res, int ldres, int res_period,
                     float drop_p, uint64_t drop_seed, float* colsum,
                     int out_mode, int splitk, int tile, float* partials, size_t partials_bytes,
                     const float* ln_stat, const float* ln_c, float* stat_part, void* stream);
int avt_ln_stats_finalize(const float* stat_part, int nslots, int rows, int D, float eps, float* stat_fwd, float* stat_bwd, void* stream);
int avt_ln_fold_weights(const float* W, int ldw, const float* gamma, const float* beta, const float* bias, void* G, int ldg,
                        float* c, float* b2, int N, int K, void* stream);
int avt_layernorm_bwd_folded(const void* dy, int lddy, const void* x, int ldx, const float* stat_fwd, const void* dres, int lddres,
                             void* dx, int lddx, float* colsum, int rows, int D, float* partials, size_t partials_bytes, void* stream);
size_t avt_layernorm_bwd_folded_workspace_bytes(int rows, int D);
int avt_ln_fold_wgrad(float* T, int ldt, const float* W, int ldw, const float* gamma, const float* beta, float* dbias_tmp,
                      float* dW, int lddw, float* dgamma, float* dbeta, float* dbias, int N, int K,
                      float* partials, size_t partials_bytes, void* stream);
size_t avt_ln_fold_wgrad_workspace_bytes(int N, int K);

/* ---- ViT spatial attention core ----------------------------------------------------------------------------------
 * [timm] Attention.forward: softmax(q k^T * scale) v per (frame, head); qkv [frames*S, 3*H*64] with columns [q|k|v]
 * head-major; out [frames*S, H*64]; lse fp32 [frames, H, S].  S <= 208, head_dim == 64.
 * bwd writes dqkv (same layout) and accumulates dbias[3*H*64] += column sums of dqkv (may be NULL). */
int avt_vit_attn_fwd(const void* qkv, void* out, float* lse, int frames, int S, int H, int head_dim, float scale, void* stream);
int avt_vit_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias,
                     int frames, int S, int H, int head_dim, float scale, float* partials, size_t partials_bytes, void* stream);
size_t avt_vit_attn_bwd_workspace_bytes(int frames, int S, int H);
/* avt_vit_attn_bwd with row r of dqkv multiplied by row_stat[2 r] on its way out (row_stat [frames*S][2] = {rstd, 1 / rstd} of the LayerNorm folded
 * into the qkv projection: dY' = rstd o dY, see "LayerNorm folded into the GEMMs"); dbias stays the column sums of the UNscaled gradient. */
int avt_vit_attn_bwd_scaled(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias,
                            int frames, int S, int H, int head_dim, float scale, float* partials, size_t partials_bytes,
                            const float* row_stat, void* stream);

/* ---- single-query attention ---------------------------------------------------------------------------------------
 * avt_cls_attn_*: the LAST ViT block's attention for the CLS query only.  [timm] VisionTransformer.forward_features returns
 * x[:, 0] after the final norm (reached through models/video_classification.py:224 -> models/base_model.py:157), so of the last
 * block only token 0's attention / proj / MLP output is ever consumed, in forward and in backward.
 *   q [frames, H*64] (row stride ldq), kv [frames*S, 2*H*64] columns [k | v] head-major, out [frames, H*64], probs fp32
 *   [frames, H, S] (saved for backward).  bwd writes dq [frames, H*64] and ALL rows of dkv; bias gradients are left to the
 *   caller (colsum(dk) == 0, colsum(dv) == colsum(dout), colsum(dq) over the compact tensor).  S <= 256, head_dim == 64.
 * avt_causal_attn_decode: [hf] GPT2Attention with `past_key_values` (models/future_prediction.py:168-202, roll-out for
 *   output_len > 1): the new token's qkv [B, 3*H*hd]; its k / v are appended to kcache / vcache [B, tmax, H*hd] at row
 *   `pos`, then out [B, H*hd] = softmax(q k[0..pos]^T * scale) v[0..pos].  No dropout (eval path). */
int avt_cls_attn_fwd(const void* q, int ldq, const void* kv, int ldkv, void* out, int ldo, float* probs,
                     int frames, int S, int H, int head_dim, float scale, void* stream);
int avt_cls_attn_bwd(const void* q, int ldq, const void* kv, int ldkv, const float* probs, const void* dout, int lddo,
                     void* dq, int lddq, void* dkv, int lddkv, int frames, int S, int H, int head_dim, float scale, void* stream);
int avt_causal_attn_decode(const void* qkv, void* kcache, void* vcache, void* out, int B, int H, int head_dim, int pos,
                           int tmax, float scale, void* stream);

/* ---- AVT-h causal attention core ---------------------------------------------------------------------------------
 * [hf] GPT2Attention._attn via models/future_prediction.py:178-181: softmax(causal(q k^T * scale)), attention dropout,
 * times v.  qkv [B*T, 3*H*hd]; probs fp32 [B,H,T,T] = pre-dropout probabilities saved for backward.  T <= 32. */
int avt_causal_attn_fwd(const void* qkv, void* out, float* probs, int B, int T, int H, int head_dim, float scale,
                        float drop_p, uint64_t seed, void* stream);
int avt_causal_attn_bwd(const void* qkv, const float* probs, const void* dout, void* dqkv, int B, int T, int H,
                        int head_dim, float scale, float drop_p, uint64_t seed, void* stream);
/* The same kernels with the mask as an argument: causal = 0 is torch.nn.MultiheadAttention inside the TransformerEncoder
 * aggregator (models/temporal_aggregation.py:86-90): every query sees all T keys. */
int avt_head_attn_fwd(const void* qkv, void* out, float* probs, int B, int T, int H, int head_dim, float scale,
                      float drop_p, uint64_t seed, int causal, void* stream);
int avt_head_attn_bwd(const void* qkv, const float* probs, const void* dout, void* dqkv, int B, int T, int H,
                      int head_dim, float scale, float drop_p, uint64_t seed, int causal, void* stream);

/* ---- patch embedding helpers ([timm] PatchEmbed + cls_token/pos_embed, via models/video_classification.py:213-227) --
 * avt_im2col_patch16: video fp32 [N,3,H,W] -> bf16 rows [N*(P+1), 768], row n*(P+1) (CLS slot) zero, k = c*256+ky*16+kx.
 * avt_posres_prep:    R[s,:] = pos[s,:] + (s == 0 ? cls : conv_bias)  (bf16 [S,D]) = row-periodic residual of the GEMM.
 * avt_patch_embed_bwd_reduce: dx0 bf16 [N,S,D] -> dpos[S,D] += sum_n; dcls[D] += row 0; dbias[D] += rows >= 1. */
int avt_im2col_patch16(const float* video, void* patches, int N, int Himg, int Wimg, void* stream);
int avt_posres_prep(const float* pos, const float* cls, const float* bias, void* R, int S, int D, void* stream);
int avt_patch_embed_bwd_reduce(const void* dx, float* dpos, float* dcls, float* dbias, int N, int S, int D,
                               float* partials, size_t partials_bytes, void* stream);
size_t avt_patch_embed_bwd_reduce_workspace_bytes(int N, int S, int D);

/* ---- elementwise ---------------------------------------------------------------------------------------------------
 * casts (bf16 shadow of fp32 parameters), dropout (nn.Dropout, models/base_model.py:81,204,215; also its own backward),
 * GPT-2 position embedding + embd dropout ([hf] GPT2Model.forward: inputs_embeds + wpe(position_ids), drop) and its
 * backward (denc = dh*mask, dwpe[t] += sum_b), column sums (bias gradients), shifted MSE
 * (models/future_prediction.py:207-215 with future_pred_loss = torch.nn.MSELoss(reduction='none'):
 * loss = (decoded[:, :T-1] - feats[:, 1:T])^2, fp32 [B,T-1,F]; bwd writes ddec and dfeats, both fp32 [B,T,F], rows without a
 * term zero), fp32 -> bf16 cast with zero column padding (classifier gradient, models/base_model.py:222-238), strided row add
 * (residual gradient of the CLS rows, see avt_cls_attn_*). */
int avt_cast_f32_to_bf16(const float* src, void* dst, long n, void* stream);
int avt_cast_bf16_to_f32(const void* src, float* dst, long n, void* stream);
int avt_dropout_bf16(const void* x, void* y, long n, float p, uint64_t seed, void* stream);
int avt_embed_pos_fwd(const void* enc, const float* wpe, void* h, int B, int T, int E, float p, uint64_t seed, void* stream);
int avt_embed_pos_bwd(const void* dh, void* denc, float* dwpe, int B, int T, int E, float p, uint64_t seed, void* stream);
int avt_colsum_bf16(const void* x, int ld, float* out, int M, int N, float* partials, size_t partials_bytes, void* stream);
size_t avt_colsum_workspace_bytes(int M, int N);
int avt_mse_shift_fwd(const float* dec, const float* x, float* loss, int B, int T, int F, void* stream);
int avt_mse_shift_bwd(const float* dec, const float* x, const float* gloss, float* ddec, float* dx, int B, int T, int F,
                      void* stream);
int avt_pad_cast_f32_to_bf16(const float* src, int lds, void* dst, int ldd, int rows, int cols, void* stream);
int avt_add_rows_bf16(void* dst, long ldd, const void* src, long lds, int rows, int D, void* stream);
/* dst[c][r] = src[r][c] (bf16): transposed shadow of a Linear weight (out,in) -> (in,out), so that the data gradient
 * dx = dy W of torch.nn.Linear's backward is a k-major x k-major GEMM like the forward (5-9 % faster for K or N >= 2304). */
int avt_transpose_bf16(const void* src, long ld_src, void* dst, long ld_dst, int rows, int cols, void* stream);
/* The same for `njobs` matrices in one launch.  jobs: DEVICE memory, njobs records of 40 bytes
 * { const void* src; void* dst; int64 ld_src; int64 ld_dst; int32 rows; int32 cols; } (the caller builds the table once: the
 * arena's weights do not move); max_tiles >= ceil(rows/64) * ceil(cols/64) of every job. */
int avt_transpose_batch_bf16(const void* jobs, int njobs, int max_tiles, void* stream);
/* ReLU + derivative mask (bf16 0/1) -- nn.TransformerEncoderLayer's activation (models/temporal_aggregation.py:87). */
int avt_relu_bf16(const void* x, void* y, void* mask, long n, void* stream);

/* ---- GPU input pipeline ----------------------------------------------------------------------------------
 * The reference's per-clip CPU transform chain (func/train.py:550-569; common/transforms.py:60-91 resize, :124-146 to_tensor,
 * :149-164 normalize, :167-175 hflip, RandomCropVideo / CenterCropVideo): uint8 frames (B,T,H,W,3) -> /255 -> bilinear resize
 * (align_corners = False) to (new_h, new_w) -> optional horizontal flip -> x scale_pix -> optional channel reversal ->
 * (v - mean) / std -> crop (OH, OW) at (crop_i, crop_j), written as fp32 (B,T,3,1,OH,OW).  params: int32 [B][6] =
 * {new_h, new_w, flip, crop_i, crop_j, source clip} per OUTPUT clip (device memory; the random draws stay with the caller; the
 * evaluation MultiCropVideo, common/transforms.py:254-296, is several output clips reading one source clip). mean3 / std3: host.
 * quantize_u8 != 0: the resized pixels are cut to 8 bits (floor(v * 255) / 255) before scaling -- what the training chain's
 * ColorJitterVideo (common/transforms.py:399-421, strengths 0 in every AVT experiment) does through its float -> PIL -> float
 * round trip (torchvision 0.8.2 to_pil_image: pic.mul(255).byte(); to_tensor: / 255).
 * patches (ABI 8; may be NULL): the SAME pixels written as the patch-embedding GEMM's bf16 rows [B T (P + 1)][768] -- row n (P + 1) the zero CLS
 * slot, row n (P + 1) + 1 + p patch p flattened as k = c 256 + ky 16 + kx, i.e. exactly avt_im2col_patch16 of the fp32 frames (timm PatchEmbed via
 * models/video_classification.py:213-227) -- so that the 0.95 MB of fp32 per frame and the im2col pass leave the step; OH, OW multiples of 16 then.
 * dst may be NULL when patches is given (either or both outputs). */
int avt_video_preproc_u8(const void* src, float* dst, void* patches, const int* params, int B, int T, int H, int W, int OH, int OW,
                         float scale_pix, const float* mean3, const float* std3, int reverse_channels, int quantize_u8,
                         void* stream);

/* The same chain with NON-ZERO colour jitter (ColorJitterVideo, common/transforms.py:399-421, around torchvision 0.8.2's ColorJitter on a
 * PIL image): resize + flip -> 8-bit clip (the wrapper's to_pil_image) -> up to four in-place Pillow operations per clip, in the order
 * and with the factors the caller drew -> / 255 -> scale -> (reverse) -> normalise -> crop.  jitter_ops: int32 [B][4], application order,
 * 0 brightness | 1 contrast | 2 saturation (ImageEnhance = Image.blend with black | the clip's mean grey | the pixel's grey) | 3 hue
 * (8-bit HSV round trip) | -1 none; jitter_factors: fp32 [B][4], the blend factor, or for hue the 8-bit shift (int)(hue_factor * 255) & 255.
 * max_h / max_w >= every clip's new_h / new_w: pitch of the 8-bit `scratch` (avt_video_jitter_scratch_bytes); luma_sums: B x uint64 scratch.
 * slot_mask: what the host knows about the device-side op table -- bit s (0..3) = some clip has an operation in slot s (other slots
 * are skipped), bit 4 + s = some clip's operation in slot s is contrast (only then is the clip's mean luma computed); 0xff = unknown.
 * Bit-exact against Pillow (tests/golden/g11_color_jitter.npz, generated through the reference's own wrapper; the HSV round trip
 * exhaustively over all 2^24 colours, tests/test_ops_gpu.py). */
int avt_video_preproc_jitter_u8(const void* src, float* dst, void* patches, const int* params, const int* jitter_ops, const float* jitter_factors,
                                int B, int T, int H, int W, int OH, int OW, int max_h, int max_w, float scale_pix,
                                const float* mean3, const float* std3, int reverse_channels, int slot_mask,
                                void* scratch, size_t scratch_bytes, unsigned long long* luma_sums, void* stream);
size_t avt_video_jitter_scratch_bytes(int B, int T, int max_h, int max_w);

/* ---- softmax cross-entropy -------------------------------------------------------------------------------------------
 * loss_fn/multidim_xentropy.py:11-25 (CrossEntropyLoss(ignore_index=-1, reduction='none')) + common/utils.py:17-44.
 * logits fp32 [R, ld], C valid columns; target int64 [R]; loss/lse fp32 [R]; rank int32 [R] (#logits > target logit,
 * -1 for ignored rows; may be NULL).  bwd: dlogits bf16 [R, ldd] = (softmax - onehot) * gout[r], padding columns zero. */
int avt_xent_fwd(const float* logits, int ld, const long* target, float* loss, float* lse, int* rank, int R, int C,
                 long ignore_index, void* stream);
int avt_xent_bwd(const float* logits, int ld, const long* target, const float* lse, const float* gout, void* dlogits,
                 int ldd, int R, int C, long ignore_index, void* stream);

/* ---- classifier + softmax cross-entropy as one operator -------------------------------------------------------
 * Replaces torch.nn.Linear (models/base_model.py:203-216, conf/model/classifier/linear.yaml:3) followed by MultiDimCrossEntropy
 * (loss_fn/multidim_xentropy.py:11-25: ignore_index, reduction 'none') and, in backward, their two autograd nodes.
 * fwd: logits[R, ldl >= Cpad] (fp32) = x[R,K] W[Cpad,K]^T + bias[Cpad];  loss[r] = lse[r] - logits[r, target[r]] (0 where target ==
 *      ignore_index), lse[r], rank[r] = number of logits above the target's (-1 where ignored; may be NULL).  C valid classes, the
 *      weight / bias rows C..Cpad-1 are zero padding (Cpad % 8 == 0).
 * bwd: dlogits = (softmax - onehot) * gloss[r] (+ glogits[r, :C], fp32 [R, ldg], a gradient that reached the logits from another
 *      consumer; may be NULL) written ONCE as bf16 [R, Cpad] (caller scratch, zero padding columns) and consumed
 *      in place by  dw[Cpad,K] (fp32) += dlogits^T x  (deterministic split-K: `workspace` of avt_gemm_accum_workspace_bytes(Cpad,K,R)),
 *      dbias[Cpad] += column sums (`partials`: see above),  dx[R,K] = dlogits W  (bf16, or fp32 when dx_f32), passed through the mask
 *      of the dropout that preceded the classifier when dx_drop_p > 0 (keep(seed, r*K + k), the mask avt_dropout_bf16 draws; the
 *      reference's nn.Dropout in front of the classifier, models/base_model.py:87-97).  dw / dbias / dx may be NULL.
 * The drop-in model runs exactly this pair as ONE autograd node when the training operator hands it the targets
 * (avt_amd/models/classifiers.py::HipLinear.forward_with_loss, avt_amd/func/train_eval_ops.py::Basic). */
int avt_linear_softmax_xent_fwd(const void* x, int ldx, const void* w, int ldw, const float* bias, const long* target,
                                float* logits, int ldl, float* loss, float* lse, int* rank,
                                int R, int C, int Cpad, int K, long ignore_index, void* stream);
int avt_linear_softmax_xent_bwd(const float* logits, int ldl, const long* target, const float* lse, const float* gloss,
                                const float* glogits, int ldg,
                                const void* x, int ldx, const void* w, int ldw, void* dlogits_bf16,
                                float* dw, int lddw, float* dbias, void* dx, int lddx, int dx_f32,
                                float dx_drop_p, uint64_t dx_drop_seed,
                                int R, int C, int Cpad, int K, long ignore_index,
                                void* workspace, size_t workspace_bytes, float* partials, size_t partials_bytes, void* stream);

/* ---- optimizer -------------------------------------------------------------------------------------------------------
 * torch.optim.SGD(momentum, nesterov, weight_decay) over a flat fp32 range (func/train.py:233, conf/opt/optimizer/sgd.yaml):
 * g = grad*grad_scale + wd*p; buf = first ? g : mom*buf + g; p -= lr*(nesterov ? g + mom*buf : buf).
 * Also writes the bf16 shadow (may be NULL) and re-zeroes grad when zero_grad != 0. */
int avt_sgd_step(float* param, float* grad, float* momentum_buf, void* shadow_bf16, long n, float lr, float momentum,
                 float weight_decay, float grad_scale, int nesterov, int first_step, int zero_grad, void* stream);

/* ---- captured steps (ABI 9) ---------------------------------------------------------------------------------------------
 * The reference's train_one_epoch body (func/train.py:203-239) is ~640 kernel launches at its own 3 clips per GPU, and the host needs as long to issue
 * them as the device to run them.  Every entry point here enqueues on the caller's stream and none allocates or synchronises, so the whole step can be
 * recorded once into a hipGraph (stream capture) and replayed -- provided the two kinds of scalars that change from step to step do not sit in the
 * launches' arguments:
 *   * dropout seeds: every `seed` / `drop_seed` / `dx_drop_seed` argument of this header may be INDIRECT -- bit 63 set, bits 0..47 = the device address
 *     of a uint64 holding a base seed, bits 48..62 = an offset added to it (the modules derive a layer's seeds as base + small constants).  The host
 *     rewrites the base seed between replays.  Plain seeds must keep bit 63 clear;
 *   * the learning rate: avt_sgd_step_dev = avt_sgd_step with the rate read from device memory (lr_dev: one float, 4-byte aligned).
 * avt_amd/func/graph.py::CapturedStep does exactly that on top of torch.cuda.CUDAGraph; its replays equal the eager steps bit for bit.
 * Warm the library up on the capturing stream first (the persistent GEMM's ticket block of a new stream is allocated at its first launch). */
int avt_sgd_step_dev(float* param, float* grad, float* momentum_buf, void* shadow_bf16, long n, const float* lr_dev, float momentum,
                     float weight_decay, float grad_scale, int nesterov, int first_step, int zero_grad, void* stream);

/* ---- gradient exchange over RCCL (ABI 8) ------------------------------------------------------------------------------
 * What the reference gets from torch.nn.parallel.DistributedDataParallel(model, device_ids=[gpu]) (func/train.py:771-778) after
 * torch.distributed.init_process_group(backend='nccl') (common/utils.py:145-148): the SUM of a bucket of the flat gradient buffer over the
 * ranks of one node (xGMI), in place, enqueued on `stream` (use a side stream and order it after the producing kernels with an event: the
 * exchange then overlaps the rest of backward -- avt_amd/ddp.py::GradReducer does exactly that, with either torch.distributed or these
 * entry points underneath).  One process per GPU.  Rank 0 draws a 128-byte id (avt_comm_unique_id) and the host ships it to the other
 * ranks by any side channel; every rank then calls avt_comm_init_rank(nranks, rank, device, id) -- collective, blocks until all ranks arrive.
 * dtype: 0 = fp32, 1 = bf16.  avt_reduce_scatter_bucket + avt_allgather_bucket are the same sum as two collectives (every rank
 * reduces 1 / nranks of the bucket, then the shards are exchanged): shard r = elements [r n / nranks, (r + 1) n / nranks) of `buf`, in place;
 * n must split into 16-byte aligned shards.  avt_broadcast_bucket: DDP's constructor broadcast of rank `root`'s parameters.
 * The 1 / nranks average is NOT applied here: avt_sgd_step's grad_scale carries it.  RCCL is loaded at first use (dlopen of librccl.so.1),
 * so a process that never calls these does not need it; errors carry RCCL's own message in avt_last_error(). */
#define AVT_COMM_ID_BYTES 128
int avt_comm_unique_id(void* id128);
int avt_comm_init_rank(void** comm, int nranks, int rank, int device, const void* id128);
int avt_comm_destroy(void* comm);
int avt_comm_size(void* comm, int* nranks, int* rank);
int avt_allreduce_bucket(void* comm, void* buf, size_t n, int dtype, void* stream);
int avt_reduce_scatter_bucket(void* comm, void* buf, size_t n, int dtype, void* stream);
int avt_allgather_bucket(void* comm, void* buf, size_t n, int dtype, void* stream);
int avt_broadcast_bucket(void* comm, void* buf, size_t n, int dtype, int root, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AVT_HIP_H_ */
