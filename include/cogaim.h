/* C ABI of libcogaim.so: gfx950 (MI355X) kernels for the Cognitive-Aim depth
 * forward pass.
 *
 * The reference (yenjane-dot/cognitive-aim-depth-estimation) is pure Python on
 * stock ATen ops: it has no plugin / operator / FFI layer, so each entry point
 * below names the reference ARITHMETIC it replaces (file:line; "hf:" =
 * transformers/models/dinov2/modeling_dinov2.py, the un-vendored DINOv2
 * dependency).  The Python side that keeps the reference's nn.Module API
 * (create_model / forward / forward_with_guidance) binds these with ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless stated; the caller owns all memory;
 *  - `stream` is a hipStream_t (NULL = default stream); nothing here allocates,
 *    synchronises or throws; launches are asynchronous;
 *  - return 0 on success, -22 for an unsupported shape/argument, -1000-<hipError>
 *    when the launch itself failed;
 *  - `dtype` selects the 16-bit operand type: 0 = bf16, 1 = fp16 (accumulation is fp32);
 *  - matrices are row-major; "16-bit [M,K]" operands are K-contiguous with a
 *    leading dimension in ELEMENTS that is a multiple of 8 and at least K; GEMM operand and
 *    output base pointers are 16-byte aligned, output leading dimensions at least N (else -22);
 *  - `range_flag` (GEMMs that store 16-bit results; may be NULL): with dtype 1 (fp16) the word it
 *    points to is OR-ed with 1 when a result exceeds the largest finite fp16 value, 65504 (the
 *    stored value is then inf); never written with dtype 0.  The caller zeroes it.
 */
#ifndef COGAIM_H
#define COGAIM_H

#ifdef __cplusplus
extern "C" {
#endif

int cogaim_version(void);

/* demo.py:162-166 (applied at demo.py:319): torchvision Resize -> ToTensor -> Normalize of one decoded RGB image,
 * on the device.  Resize of a PIL image is Pillow's Image.resize(BILINEAR): per axis, integer bilinear weights
 * (22 fractional bits, support widened by the shrink factor), horizontal pass into an 8-bit intermediate, then
 * vertical pass, each (2^21 + sum) >> 22 clipped to 0..255 -- reproduced bit for bit.  The caller supplies the
 * per-axis tables ([out, 2] = first source index and tap count; [out, k] weights) and a [3, 256] fp32 table of
 * (v / 255 - mean) / std.  src uint8 [src_h, src_w, src_pixel_bytes] (row pitch src_row_bytes; 3 bytes per pixel =
 * packed RGB, 4 = RGBX, the layout Pillow keeps an RGB image in, fourth byte ignored), tmp uint8 [src_h, out_w, 3]
 * scratch, dst fp32 [3, out_h, out_w]. */
int cogaim_preprocess_rgb8(const void* src, int src_h, int src_w, long long src_row_bytes, int src_pixel_bytes,
                           const int* xbounds,
                           const int* xcoefs, int xk, const int* ybounds, const int* ycoefs, int yk, void* tmp,
                           const float* lut, float* dst, int out_h, int out_w, void* stream);

/* The same for a batch of n decoded images in two launches (the demo.py loop at demo.py:406-432 preprocesses file by
 * file).  items: n records of 88 bytes in DEVICE memory, one per image:
 *   { u64 src, i64 src_row_bytes, u64 xbounds, u64 xcoefs, u64 ybounds, u64 ycoefs, u64 tmp, u64 dst,
 *     i32 src_h, i32 src_w, i32 src_pixel_bytes, i32 xk, i32 yk, i32 0 }
 * with the meaning of the arguments of cogaim_preprocess_rgb8 (every image its own tmp); max_src_h = largest src_h. */
int cogaim_preprocess_rgb8_batch(const void* items, int n, int max_src_h, const float* lut, int out_h, int out_w,
                                 void* stream);

/* The GEMM entry points take an optional `workspace` (device memory, `workspace_bytes` long;
 * NULL/0 allowed).  With at least cogaim_gemm_workspace_bytes() bytes they split the tiles of
 * a partial last round along K over the idle CUs (fp32 partials, summed in fixed order: results
 * stay bit-reproducible).  No counterpart in the reference. */
long long cogaim_gemm_workspace_bytes(void);
#ifdef COGAIM_TOOLS
/* Builds with -DCOGAIM_TOOLS only (tools/wt_build.sh, for same-box A/B timing); the shipped library does not export
 * it, because it is process-wide and not synchronised.  0 = 128x128-tile kernel for every problem, 1 = 256x256 kernel
 * (one tile per workgroup), 2 (default) = persistent 256x256 kernel; 1 and 2 apply to large problems only. */
int cogaim_set_gemm_algo(int algo);
#endif

/* ---- DINOv2 backbone --------------------------------------------------- */

/* hf:139-149 Dinov2PatchEmbeddings (Conv2d k=14,s=14 as an im2col GEMM), step 1:
 * images fp32 [B,3,H,W] -> cols 16-bit [B*(H/p)*(W/p), Kp], column = c*p*p + ky*p + kx,
 * zero-padded from 3*p*p to Kp (multiple of 64). */
int cogaim_im2col(const float* images, void* cols, int B, int H, int W, int patch, int Kp, int dtype, void* stream);

/* hf:97-116 Dinov2Embeddings, step 2: x[b, 1+p, :] = cols[b*P+p] . W^T + bias + pos[1+p]
 * (x fp32 [B, P+1, D], W 16-bit [D, Kp], pos fp32 [P+1, D] already interpolated, hf:57-95). */
int cogaim_gemm_patch_embed(const void* cols, const void* W, const float* bias, const float* pos, float* x, int B,
                            int P, int D, int K, int dtype, void* workspace, long long workspace_bytes, void* stream);

/* hf:108-112: x[b, 0, :] = cls_token + pos[0]. */
int cogaim_cls_row(const float* cls, const float* pos, float* x, int B, int T, int D, void* stream);

/* hf:348,353,441 nn.LayerNorm(D, eps): out = LN(x) * gamma + beta.  x fp32 [M,D];
 * out_dtype 0 bf16, 1 fp16, 2 fp32. */
int cogaim_layernorm(const float* x, const float* gamma, const float* beta, void* out, int M, int D, float eps,
                     int out_dtype, void* stream);

/* hf:199-213 query/key Linear (one fused GEMM): out16[M,N] = A . W^T + bias, and columns
 * [0, scale_cols) times `scale` (the softmax scale hf:196 folded into Q).  Also the focal
 * stream's query_proj/key_proj (src/model.py:192-193, scale 1/sqrt(D//8) of :69,197). */
int cogaim_gemm_bias_half(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int lda,
                          int ldw, int ldc, int scale_cols, float scale, int dtype, void* workspace,
                          long long workspace_bytes, int* range_flag, void* stream);

/* hf:153-178 softmax(Q K^T * dh^-0.5) V for all heads (dh = 64), flash style, on the fused projection buffer
 * (hf:203-234 computes query, key and value from the same hidden state: one cogaim_gemm_bias_half with the
 * three weights stacked): qkv 16-bit [B*T, ld], Q (pre-multiplied by dh^-0.5*log2(e)) in cols [0,D), K in
 * [D,2D), V in [2D,3D), all row-major (no transposed copy of V); ctx 16-bit [B*T, D]. */
int cogaim_mhsa_qkv_fwd(const void* qkv, void* ctx, int B, int T, int H, int ld, int dtype, void* stream);

/* hf:246-252 output dense + hf:277-278 LayerScale + hf:370 residual, and the same for
 * fc2 (hf:296, 375-378):  x[M,N] += (A . W^T + bias) * layerscale, in place, x fp32. */
int cogaim_gemm_residual_f32(const void* A, const void* W, const float* bias, const float* layerscale, float* x, int M,
                             int N, int K, int lda, int ldw, int ldc, int dtype, void* workspace,
                             long long workspace_bytes, void* stream);

/* hf:246-252 output dense + hf:277-278 LayerScale + hf:370 residual as above, and the LayerNorm that reads the new
 * residual stream (hf:372 norm2) in the same kernel, for N = 768 only (other widths: -22, use the two calls):
 *   x[M,768] += (A . W^T + bias) * layerscale  (in place, fp32);  h16[M,768] = LayerNorm(x) * gamma + beta.
 * K a multiple of 32; lda, ldw multiples of 8, ldx, ldh multiples of 4 (elements). */
int cogaim_gemm_residual_layernorm(const void* A, const void* W, const float* bias, const float* layerscale, float* x,
                                   const float* gamma, const float* beta, void* h, int M, int N, int K, int lda, int ldw,
                                   int ldx, int ldh, float eps, int dtype, void* stream);

/* hf:293-295 fc1 + exact (erf) GELU: out16[M,N] = gelu(A . W^T + bias). */
int cogaim_gemm_bias_gelu_half(const void* A, const void* W, const float* bias, void* out, int M, int N, int K, int lda,
                               int ldw, int ldc, int dtype, void* workspace, long long workspace_bytes, int* range_flag,
                               void* stream);

/* ---- focal stream (src/model.py:128-313, 391-455) ----------------------- */

/* :184 patch_tokens + pos_encoding, with the iteration rescale of :426 folded in:
 * out16[b*N+n] = tok[b][n] * scale[b][n] + pe[n]   (scale may be NULL). */
int cogaim_focal_prep(const float* tok, long long tok_stride_b, const float* scale, const float* pe, void* out, int B,
                      int N, int D, int dtype, void* stream);

/* :197 attention_scores = Q K^T / scale, per image: out32[z] = A[z] . W[z]^T (strides in elements). */
int cogaim_gemm_batched_f32(const void* A, const void* W, float* out, int M, int N, int K, int lda, int ldw, int ldc,
                            long long strideA, long long strideW, long long strideC, int batch, int dtype,
                            void* stream);

/* rows per partial block of the two column-sum kernels below */
int cogaim_focal_colsum_rows(void);

/* :200 softmax row statistics (scores in the log2 domain) and :234 attention_weights.mean(dim=1):
 * rmax/rinv [B,N]; part [B, nblk, N] = per-row-block column sums of the softmax. */
int cogaim_focal_softmax_stats_colsum(const float* S, int ldS, long long strideS, float* rmax, float* rinv, float* part,
                                      int B, int N, void* stream);

/* :234-249 the three deterministic summaries of the attention matrix the reference chooses between, each with the
 * centre bias added: cand[0] = column mean (:234-239), cand[1] = diagonal (:243-244), cand[2] = row maximum
 * (:248-249); cand fp32 [3][B][N], stats fp64 [B][6] = per-image {sum, sum of squares} of each.
 * part [B, nblk, N] column-sum partials; nblk <= 0 = ceil(N / cogaim_focal_colsum_rows()).  S (scores, log2 domain,
 * leading dimension ldS, strideS between images) may be NULL when no score matrix is kept: cand[1] repeats cand[0]. */
int cogaim_focal_attention_candidates(const float* part, int nblk, const float* center_bias, const float* S, int ldS,
                                      long long strideS, const float* rmax, const float* rinv, float* cand, double* stats,
                                      int B, int N, void* stream);

/* :242-282 the cascade `if patch_attention.var() < 1e-6` (unbiased variance over the whole batch) over the candidates,
 * then L1 normalise, clamp, renormalise -> a[B,N]; also :426 next-iteration token scale.  flags[slot] = 1 when the
 * last deterministic candidate is degenerate as well (the reference then draws noise, :252-257: not reproduced) or
 * the diagonal is needed and have_diag is 0; else 0. */
int cogaim_focal_attention_vector(const float* cand, const double* stats, int have_diag, float* a, const float* scale_in,
                                  float* scale_out, float focus_strength, int B, int N, int* flags, int slot, void* stream);

/* The same :197-204, :234 for large grids, with no score matrix in memory (N x N fp32 is 120 MB per image at
 * 1036 pixels): S = Y X^T is formed tile by tile three times and reduced in registers --
 *   rowstats: max and sum of exp2(S - max) per row and 64-column strip -> pmax / psum [batch][N][parts],
 *             parts = cogaim_focal_rowstats_parts(N); combine: -> rmax / rinv [B*N];
 *   colsum:   part[z][blk][c] = sum over the 128 rows n of block blk of roww[n] exp2(S[n][c] - rmax[n]) rinv[n]
 *             (roww NULL = 1), part [batch][cogaim_focal_scores_colsum_blocks(N)][N] -- once with roww NULL for the
 *             column mean of :234, once with roww = a for (a^T P) of :204/:308.
 * Y, X 16-bit [batch][N, D] with leading dimension ld and `stride` elements between images. */
int cogaim_focal_rowstats_parts(int N);
int cogaim_focal_scores_colsum_blocks(int N);
int cogaim_focal_scores_rowstats(const void* Y, const void* X, float* pmax, float* psum, int N, int D, int ld,
                                 long long stride, int batch, int dtype, void* stream);
int cogaim_focal_rowstats_combine(const float* pmax, const float* psum, int parts, float* rmax, float* rinv, int B, int N,
                                  void* stream);
int cogaim_focal_scores_colsum(const void* Y, const void* X, const float* rmax, const float* rinv, const float* roww,
                               float* part, int N, int D, int ld, long long stride, int batch, int dtype, void* stream);

/* :204,:308  sum_n a_n (P V)_n = (a^T P) V:  part[b][blk][c] = sum_{n in blk} roww[b][n] P[b][n][c]. */
int cogaim_focal_weighted_colsum(const float* S, int ldS, long long strideS, const float* rmax, const float* rinv,
                                 const float* roww, float* part, int B, int N, void* stream);
int cogaim_focal_reduce_u(const float* part, float* u, float* usum, int B, int N, int nblk, void* stream);

/* out[b][:] = sum_n w[b][n] * (tok[b][n][:] * scale[b][n] + pe[n][:]);  scale/pe NULL = 1/0.
 * Used for (a^T P) Xt above and for the guided token sum of src/model.py:1412-1414.
 * workspace: fp32 [B * ceil(N / cogaim_weighted_tokens_rows()) * D]. */
int cogaim_weighted_tokens_rows(void);
int cogaim_weighted_tokens(const float* tok, long long tok_stride_b, const float* scale, const float* pe, const float* w,
                           float* out, float* workspace, int B, int N, int D, void* stream);
/* The same with w[b][n] = sum_blk cpart[b][blk][n] (the partial column sums u = a^T P of cogaim_focal_weighted_colsum /
 * cogaim_focal_scores_colsum, cnblk blocks; <= 0 = ceil(N / cogaim_focal_colsum_rows())) reduced inside the token pass;
 * wsum_part [B][ceil(N / cogaim_weighted_tokens_rows())] (NULL: skipped) receives sum_n w[b][n] per chunk of rows
 * (the factor of the value-projection bias, src/model.py:194 under :308). */
int cogaim_weighted_tokens_parts(const float* tok, long long tok_stride_b, const float* scale, const float* pe,
                                 const float* cpart, int cnblk, float* out, float* workspace, float* wsum_part, int B,
                                 int N, int D, void* stream);

/* No counterpart in the reference: cogaim_weighted_tokens for callers that cannot allocate device memory or
 * pass a workspace (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
 * Allocates nothing: the partial sums live in a static device array of the library, 8 MiB per GPU, held
 * until the process ends; B * ceil(N / cogaim_weighted_tokens_rows()) * D must not exceed
 * COGAIM_WT_STATIC_MAX_FLOATS (else -22, nothing launched).  The first call on a device looks the array up
 * (make it outside a stream capture).  All calls on one GPU share the array: calls that overlap in time on
 * one GPU race.  Results equal cogaim_weighted_tokens bit for bit. */
#define COGAIM_WT_STATIC_MAX_FLOATS (2 * 1024 * 1024)
int cogaim_weighted_tokens_static(const float* tok, long long tok_stride_b, const float* scale, const float* pe,
                                  const float* w, float* out, int B, int N, int D, void* stream);

/* src/model.py:1401-1409: out = softmax((alpha*mask + (1-alpha)*a) / temperature). */
int cogaim_guided_attention(const float* a, const float* mask, float* out, int B, int N, float alpha, float temperature,
                            void* stream);

/* The same for K instructions at once (the base attention of src/model.py:1257 does not depend on the
 * instruction, only the mask of :1262-1398 does): masks [K,N], a [B,N] -> out [K,B,N]; slice k is bit-identical
 * to cogaim_guided_attention with masks[k].  Then the guided token sums of :1412-1414 for all K in one pass:
 * w [K,B,N] -> out [K,B,D] (no scale / position table), workspace fp32 [K*B*ceil(N/rows)*D]. */
int cogaim_guided_attention_multi(const float* a, const float* masks, float* out, int B, int N, int K, float alpha,
                                  float temperature, void* stream);
int cogaim_weighted_tokens_multi(const float* tok, long long tok_stride_b, const float* w, float* out, float* workspace,
                                 int B, int N, int D, int K, void* stream);

/* ---- per-image MLPs and heads ------------------------------------------- */

/* nn.Linear (+ReLU) in fp32 for the ambient MLP (src/model.py:37-44), focal projection (:82-87),
 * focal fusion (:342-346), EXIF MLPs (:468-480), fusion (:908-912):
 * out[b*ldo+o] = act(x[b*ldx+:K] . W[o] + bias[o] + bias2[o] * s_b),  s_b = sum_{p < scale_parts} bias2_scale[b*scale_parts+p];
 * act 0 none, 1 ReLU; bias, bias2 and bias2_scale may be NULL (s_b = 1).  One wave per output: the form for layers wide
 * enough to fill the chip (768 inputs); narrower chains go through cogaim_mlp_chain. */
int cogaim_linear_f32(const float* x, int ldx, const float* W, const float* bias, const float* bias2,
                      const float* bias2_scale, int scale_parts, float* out, int ldo, int B, int O, int K, int act,
                      void* stream);

/* src/model.py:491-514: camera embedding gather and [focal_length, aperture, log(iso+1)].
 * A camera_idx outside [0, num_embeddings) sets *err_flag = 1 (if not NULL) and fills that image's
 * embedding with NaN, so every value derived from it is NaN (the reference raises IndexError). */
int cogaim_exif_prep(const long long* camera_idx, const float* focal_length, const float* aperture, const float* iso,
                     const float* embedding, int num_embeddings, int emb_dim, float* feat, int ldf, float* cont, int B,
                     int* err_flag, void* stream);

/* src/model.py:925-941,1120-1123: depth = Softplus(Linear), conf = Sigmoid(Linear(1,1)(ReLU(Linear))). */
int cogaim_heads(const float* fused, int K, const float* wd, const float* bd, const float* wc, const float* bc,
                 const float* w2, const float* b2, float* depth, float* conf, int B, void* stream);

/* Several small Linear layers per image in ONE launch: ambient MLP (src/model.py:37-44), a FocalStream's value
 * projection + projection MLP (:194, :82-87, :308-311), focal fusion (:342-346, :430), EXIF prior (:465-480, :491-519).
 * One workgroup per image walks n_layers <= 8 layers; activations stay on chip in an array of 4096 floats that the
 * offsets below index (narrow layers only: a layer of 768 inputs keeps one workgroup busy for ~20 us, so such layers go
 * through cogaim_linear_f32, which spreads them over the chip).
 *   layer l:  buf[out_off + o] = act(W[o, :K] . buf[in_off : in_off + K] + b[o] + b2[o] * scale[image]),  o < O
 * (b, b2 may be NULL; act 0 none, 1 ReLU; a layer's input and output regions must not overlap).  Before layer 0:
 * buf[in0_off + i] = sum_{p < in0_parts} in0[image * in0_ld + p * in0_part_stride + i] for i < in0_len, and, when in1
 * is not NULL, buf[in1_off + i] = in1[image * in1_ld + i] for i < in1_len.  Afterwards
 * out[image * out_ld + i] = buf[out_off + i] for i < out_len.  W matrices are fp32 [O, K] row-major.
 * The descriptor is read on the host during the call. */
typedef struct {
  const float* W; const float* b; const float* b2;
  int K, O, act, in_off, out_off, reserved;
} cogaim_mlp_layer;
typedef struct {
  cogaim_mlp_layer layer[8];
  int n_layers, in0_len, in0_off, in0_parts;
  const float* in0; long long in0_ld, in0_part_stride;
  const float* in1; long long in1_ld; int in1_len, in1_off;
  const float* scale;
  float* out; long long out_ld; int out_len, out_off;
} cogaim_mlp_chain_desc;
int cogaim_mlp_chain(const cogaim_mlp_chain_desc* desc, int B, void* stream);

/* src/model.py:908-912 + 925-941 (applied at :1045, :1120-1123) in one launch: fused [B,F] = ReLU(Linear(K,F)(cat)),
 * depth = Softplus(Linear(F,1)(fused)), conf = Sigmoid(Linear(1,1)(ReLU(Linear(F,1)(fused)))).  K, F <= 1024. */
int cogaim_fusion_heads(const float* cat, int ldcat, int K, const float* wf, const float* bf, int F, const float* wd,
                        const float* bd, const float* wc, const float* bc, const float* w2, const float* b2, float* fused,
                        float* depth, float* conf, int B, void* stream);

/* No counterpart in the reference (torch returns fresh tensors there): one launch that copies a forward's results
 * out of fixed workspace buffers into the caller's tensors -- depth / conf [B], fused [B,F] (dst NULL: skipped),
 * attention [B,N] (dst NULL: skipped) -- and n_flags condition words into `flags_host`, a HOST pointer to page-locked
 * mapped memory for which cogaim_host_pointer_ok returned 0 (NULL: skipped). */
int cogaim_export_outputs(const float* depth_src, const float* conf_src, const float* fused_src, int F,
                          const float* attn_src, int N, float* depth, float* conf, float* fused, float* attn,
                          const int* flags, int* flags_host, int n_flags, int B, void* stream);
int cogaim_host_pointer_ok(void* host_ptr);

/* No counterpart in the reference (it runs one image at a time on one device): the gather step of image-sharded
 * inference.  Every rank contributes `count` floats (its [B_local, 2] depth / confidence rows), recvbuf receives
 * world * count floats in rank order, over the caller's ncclComm_t (RCCL; xGMI inside a node) on `stream`.
 * RCCL is looked up at call time among the libraries already loaded into the process: -38 when there is none.
 * The bundled Python paths (bench.py, inference.py) reach the same collective through torch.distributed, which
 * owns its communicator; this entry point is for hosts that hold their own. */
int cogaim_allgather_f32(const float* sendbuf, float* recvbuf, long long count, void* comm, void* stream);

/* Curiosity module (src/model.py:586-773), the parts that are not plain Linear layers.  Its score never
 * reaches depth or confidence; it is evaluated only in the opt-in faithful mode, for the state it leaves behind
 * (exploration_history ring buffer, history_pointer) and the random draws it consumes.
 * sample: z = mu + eps * exp(logvar / 2) [B,L] (:608-610) and noisy = cls + noise [B,D] (:743-745); the draws
 * eps, noise are the caller's.  finish: reconstruction error ("robust", :620-623), KL term (:635), softplus
 * uncertainty head (:638-643), local term with its sensitivity (:738-758), geometric term 0.5 (no EXIF, :698-700),
 * softmax(weights3) mix (:666-671) -> reward [B]; then history[(pointer + i) % hist_len] = reward[i] in image
 * order and pointer += B mod hist_len (:769-773), unless history / pointer are NULL. */
int cogaim_curiosity_sample(const float* mu, const float* logvar, const float* eps, float* z, const float* cls,
                            long long ld_cls, const float* noise, float* noisy, int B, int L, int D, void* stream);
int cogaim_curiosity_finish(const float* rec, const float* cls, long long ld_cls, const float* mu, const float* logvar,
                            const float* unc_pre, const float* lb_pre, const float* ln_pre, const float* weights3,
                            float* reward, float* history, long long* pointer, int B, int L, int hist_len, void* stream);

/* demo.py:533-552 (the attention map as it is drawn): per image, weights cubed, those at or below their 70th
 * percentile (numpy's default linear interpolation) multiplied by 0.3, min-max rescaled with the reference's
 * 1e-8 in the denominator, laid out on the G x G grid (G = ceil(sqrt(N)), zero padded when N is not a square).
 * attn fp32 [B,N], out fp32 [B,G*G]; N <= 6144. */
int cogaim_heatmap_postprocess(const float* attn, float* out, int B, int N, int G, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COGAIM_H */
