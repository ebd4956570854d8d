/* tvae_hip.h — C ABI of libtvae_hip.so: the latent bottleneck + MMD hot path of
 * Transformer-VAE on MI355X (gfx950).
 *
 * The reference (Fraser-Greenlee/transformer-vae) is pure Python and has no FFI layer;
 * the seam this library plugs into is the module interface of
 * transformer_vae/autoencoders.py.  Each entry point below names the reference lines
 * whose arithmetic it replaces.  The Python host (transformer_vae_amd/) binds these
 * symbols with ctypes and wraps them in torch.autograd.Functions; a reference
 * maintainer would bind them the same way.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated otherwise; sizes are element
 *     counts (int64_t); matrices are row-major with the innermost dimension contiguous;
 *   - `dtype` is the element type of activations AND parameters of that call
 *     (TVAE_F32 or TVAE_BF16); accumulation is always fp32 (fp32-input MFMA), the
 *     MMD reductions are fp64;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Kernels are
 *     enqueued on it; no entry point synchronises, allocates or frees device memory —
 *     scratch comes from the caller as `workspace`, sized by the matching
 *     *_workspace_bytes() query (a host-only function);
 *   - return value: 0 on success, a TVAE_ERR_* code otherwise; tvae_last_error()
 *     returns a thread-local, human-readable message for the last failure;
 *   - concurrent calls from different host threads on different streams (nn.DataParallel
 *     replicas) are safe: the only state the library keeps is per-device and write-once
 *     (kernel attribute bits, the address of tvae_mmd_fwd_bwd_static's scratch array) plus the
 *     thread-local error message and measurement hook.
 */
#ifndef TVAE_HIP_H
#define TVAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TVAE_F32 0
#define TVAE_BF16 1
/* tvae_mmd_* only: fp32 inputs with every product on the fp32-input matrix instruction (v_mfma_f32_16x16x4_f32).
 * Plain TVAE_F32 gives the MMD the same accuracy class (1e-5 of the fp32 reference on loss and gradient, fp32
 * accumulation, fp64 reductions) but, for latent_size <= 1024 (latent_size <= 192: from 1024 points on), forms each
 * fp32 product from two (Gram: row operand hi + lo, column operand hi) or three (gradient) fp16 matrix-core products of
 * scaled hi/lo halves, at several times the speed (measured error against fp64: 1e-7 .. 5e-7).  For
 * 192 < latent_size <= 1024 the gradient's part linear in the squared distances is closed form (moments and three small
 * fp32 GEMMs inside the call) and the remainder takes one fp16 product; a device-side switch keeps the
 * three-product form when the points' norms are large against latent_size. */
#define TVAE_F32_STRICT 2

#define TVAE_OK 0
#define TVAE_ERR_ARG 1         /* bad shape / null pointer / unsupported size */
#define TVAE_ERR_HIP 2         /* a HIP runtime call failed */
#define TVAE_ERR_WORKSPACE 3   /* workspace too small */

/* Library ABI version (major*1000 + minor). */
int tvae_version(void);
/* Thread-local message describing the last error returned on this host thread. */
const char* tvae_last_error(void);

/* ---------------------------------------------------------------------------------
 * Latent encoder.  Replaces LatentEncoderNTokens.forward, autoencoders.py:21-22:
 *     latent = tanh(Linear(d_model -> L)(h))[:, :n_tokens, :]
 * computed slice-first: only the first n_eff = min(n_tokens, S) tokens of every
 * sequence are projected (bit-identical values, S/n times less work).
 *   h      (B, S, d) with element strides (h_stride_b, h_stride_s, 1)
 *   We     (L, d), be (L)            nn.Linear weight / bias
 *   latent (B, n_eff, L) contiguous  output
 *   workspace: optional split-K scratch (may be NULL / 0: the product is then not split)
 * ------------------------------------------------------------------------------- */
int64_t tvae_encode_fwd_workspace_bytes(int64_t B, int64_t n_eff, int64_t d, int64_t L);
int tvae_encode_fwd(const void* h, int64_t h_stride_b, int64_t h_stride_s,
                    const void* We, const void* be, void* latent,
                    int64_t B, int64_t n_eff, int64_t d, int64_t L,
                    int dtype, void* workspace, int64_t workspace_bytes, void* stream);

/* Backward of the encoder (autograd of autoencoders.py:22).
 *   d_latent (B, n_eff, L)  total cotangent of `latent`
 *   latent   (B, n_eff, L)  saved forward output (tanh'(x) = 1 - latent^2)
 *   dh       (B, S, d) contiguous, nullable: rows [0,n_eff) of every sequence receive
 *            d_pre @ We, rows [n_eff, S) are zero-filled by this call (the reference's
 *            slice backward yields exact zeros there)
 *   dWe (L, d), dbe (L)     nullable (skipped when null) */
int64_t tvae_encode_bwd_workspace_bytes(int64_t B, int64_t n_eff, int64_t d, int64_t L);
int tvae_encode_bwd(const void* d_latent, const void* latent,
                    const void* h, int64_t h_stride_b, int64_t h_stride_s,
                    const void* We, void* dh, void* dWe, void* dbe,
                    int64_t B, int64_t S, int64_t n_eff, int64_t d, int64_t L,
                    int dtype, void* workspace, int64_t workspace_bytes, void* stream);

/* tvae_encode_bwd with the latent cotangent given in two parts (saves the caller a pass to add them):
 *     cotangent of `latent` = d_latent + (*extra_scale) * d_latent_extra
 * In the model this is the decoder's share plus the regulariser's: the gradient of `loss = ce + reg * w`
 * (transformer_vae/model.py:305-306) reaches the latent through both, and the regulariser's gradient is
 * saved unscaled by tvae_mmd_fwd_bwd while its upstream factor (w times whatever the trainer applies,
 * trainer.py:258-281) only exists as a device scalar during backward.
 *   d_latent_extra (B, n_eff, L) in `dtype`;  extra_scale: DEVICE pointer to one float.
 * Workspace: tvae_encode_bwd_workspace_bytes. */
int tvae_encode_bwd_sum(const void* d_latent, const void* d_latent_extra, const float* extra_scale,
                        const void* latent, const void* h, int64_t h_stride_b, int64_t h_stride_s,
                        const void* We, void* dh, void* dWe, void* dbe,
                        int64_t B, int64_t S, int64_t n_eff, int64_t d, int64_t L,
                        int dtype, void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------
 * Latent decoder.  Replaces LatentDecoderNTokens.forward (autoencoders.py:38-39) and,
 * when gamma != NULL, LatentDecoderFunnelNorm.forward (autoencoders.py:66-67):
 *     r = latent @ Wd^T + bd          (Wd == NULL: identity, the L == d_model case
 *                                      of autoencoders.py:32-34; then L must equal d)
 *     recon = LayerNorm_d(r; gamma, beta, eps)   or   recon = r
 *   latent (N, L)  N = B * n_eff
 *   Wd (d, L), bd (d), gamma (d), beta (d)
 *   recon (N, d) output
 *   pre_norm (N, d): r, saved for backward; required iff gamma != NULL and Wd != NULL
 *            (with Wd == NULL the caller's `latent` is r)
 *   mean, rstd (N) fp32: LayerNorm statistics, required iff gamma != NULL */
int64_t tvae_decode_fwd_workspace_bytes(int64_t N, int64_t L, int64_t d);
int tvae_decode_fwd(const void* latent, const void* Wd, const void* bd,
                    const void* gamma, const void* beta, float eps,
                    void* recon, void* pre_norm, float* mean, float* rstd,
                    int64_t N, int64_t L, int64_t d, int dtype,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* Backward of the decoder.
 *   d_recon (N, d) cotangent;  pre_norm/mean/rstd as saved by tvae_decode_fwd
 *   d_latent (N, L) output (nullable);  dWd (d, L), dbd (d), dgamma (d), dbeta (d) nullable */
int64_t tvae_decode_bwd_workspace_bytes(int64_t N, int64_t L, int64_t d);
int tvae_decode_bwd(const void* d_recon, const void* latent, const void* Wd,
                    const void* gamma, const void* pre_norm, const float* mean, const float* rstd,
                    void* d_latent, void* dWd, void* dbd, void* dgamma, void* dbeta,
                    int64_t N, int64_t L, int64_t d, int dtype,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------
 * MMD regulariser, forward and gradient in one pass.  Replaces
 * EncoderDecoderVAE._compute_kernel/_compute_mmd/_regularliser_loss
 * (autoencoders.py:117-136) and their autograd:
 *     K(a,b) = exp(-||a-b||^2 / D^2);  MMD = mean Kyy + mean Kzz - 2 mean Kyz
 * evaluated as tiled Gram products on the matrix cores with E = K - 1 = expm1(-t)
 * accumulated instead of K (the '+1's cancel exactly), never materialising N x N.
 *
 * Row-sharded form: this call handles rows [row_offset, row_offset + n_rows) of the
 * kernel matrices against ALL n_total columns (single GPU: row_offset 0, n_rows =
 * n_total).  z_all / y_all hold all n_total latent / prior rows (after the
 * all-gather), both (n_total, D), same dtype.
 *   sums3  double[3] out: this call's share of the sums of E over (z rows x all z), (z rows x all y),
 *          (y rows x all y).  Adding sums3 over a partition of the rows into EQUAL blocks (every call
 *          the same n_rows, row_offset a multiple of it; or one call for everything) gives the global
 *          sums:  MMD = (S_yy + S_zz - 2 S_zy) / n_total^2.  (The y-y share of a call is not simply
 *          "its rows x all columns": symmetric pairs of row blocks are split between the calls that own
 *          them, and all calls of a partition must decide alike whether to -- equal blocks do.)
 *   loss   float out, nullable: loss_scale * (S_yy + S_zz - 2 S_zy) of THIS call's rows
 *          (single GPU: pass loss_scale = n_tokens / (B * n_total^2) to get reg_loss)
 *   d_z    (n_rows, D) out, nullable: grad_scale * sum-form gradient for the local rows,
 *          complete (needs no exchange):  pass grad_scale = 4 n / (B N^2 D^2) for
 *          d reg_loss / d latent  (SURVEY.md 8a).  Written in `dtype`.
 * Any D: fused tile kernels up to 1024.  TVAE_F32 inputs run on the 16-bit matrix cores as centred, power-of-two-scaled fp16
 * pairs hi + lo: ONE fp16 product per fp32 product for 192 < D <= 768 (the kernels sum the second-order part Q = E + t and the
 * part linear in t is closed form; a device flag falls back to two / three products when t can leave the polynomial's range), the
 * same for 768 < D <= 1024 on the 32-row kernels, three for D <= 192 from 1024 points on.  TVAE_BF16 inputs wider than 192: one
 * bf16 product.  TVAE_F32_STRICT and everything else: fp32-input MFMA.  D > 1024: strip-wise through the projection GEMM.
 * n_total < 2^30. */
int64_t tvae_mmd_workspace_bytes(int64_t n_total, int64_t n_rows, int64_t D, int want_grad);
/* Same, for the dtype the call will be made with (TVAE_F32 / TVAE_BF16).  tvae_mmd_workspace_bytes covers
 * either; bf16 calls with large column sets use one gradient slab per ~24 MB column chunk (up to 32 GiB in all),
 * which an fp32 caller need not pay for.  Like the query above it has no counterpart in the reference, which
 * materialises (N, N, D) temporaries instead (autoencoders.py:117-126). */
int64_t tvae_mmd_workspace_bytes_for(int64_t n_total, int64_t n_rows, int64_t D, int want_grad, int dtype);
int tvae_mmd_fwd_bwd(const void* z_all, const void* y_all,
                     int64_t n_total, int64_t row_offset, int64_t n_rows, int64_t D, int dtype,
                     double* sums3, float* loss, double loss_scale,
                     void* d_z, double grad_scale,
                     void* workspace, int64_t workspace_bytes, void* stream);

/* The same evaluation cut into three stages, for a GLOBAL batch whose rows are sharded over R processes (one per GPU):
 * what `EncoderDecoderVAE._regularliser_loss` (autoencoders.py:128-136) computes when the batch it is given is the
 * concatenation of every rank's rows -- the reference itself has no collective (under DDP each rank regularises its own
 * rows).  No rank touches another rank's rows before the last stage, so the O(N D) preparation and the O(N D^2)
 * closed-form products shrink with R like the tile work does, and what travels is the fp16 image (half the bytes of the
 * fp32 latents).  Every rank passes the same n_total, n_rows (equal row counts), D, dtype and want_grad, its own
 * row_offset = rank * n_rows, its own rows z_loc / y_loc (n_rows, D), and ONE workspace of
 * tvae_mmd_workspace_bytes_for() bytes that lives across the three calls.  Between the calls the caller moves the
 * regions tvae_mmd_shard_layout names (collectives are the caller's: RCCL in distributed.py):
 *     tvae_mmd_stats_local   -> rec_a: SUM over ranks (its max |value| word: MAX)
 *     tvae_mmd_prep_local    -> image rows and norms of the own rows: ALL-GATHER in rank order;
 *                               rec_b_local: SUM over ranks into rec_b (its max-norm word: MAX); gram: SUM over ranks
 *     tvae_mmd_tiles         -> sums3 / loss / d_z of the own rows as tvae_mmd_fwd_bwd returns them
 * tvae_mmd_tiles(phase): 1 = the part that needs only the prior's image, norms and rec_b (with a gradient: the y-y
 * sum-only launch), 2 = the rest, 3 = both -- a caller runs phase 1 while the latent images are still on the wire.
 * tvae_mmd_fwd_bwd is these three stages back to back over all rows.  Inputs are centred on the pooled mean of both
 * sets in the split modes (the MMD is translation invariant, fp16 pairs are not).  D <= 1024.
 * layout out[16]: [0] workspace bytes; [1] [2] offsets of the z / y image; [3] bytes per image row; [4] bytes from an
 * image to its lo image (0: none); [5] [6] offsets of the z / y squared norms (fp32 per row); [7] offset of rec_a
 * (-1: the mode has no stage A), [8] its doubles (max |value| is the last but one); [9] [10] offsets of rec_b and
 * rec_b_local, [11] their doubles (the max-norm word is the last but one); [12] offset of gram (-1: none), [13] its
 * floats; [14] 1: other ranks' lo image rows are read only when 4 max-norm / D^2 >= 1/16 (the caller may skip their
 * gather otherwise); [15] padded D. */
int tvae_mmd_shard_layout(int64_t n_total, int64_t n_rows, int64_t D, int want_grad, int dtype, int64_t* out16);
int tvae_mmd_stats_local(const void* z_loc, const void* y_loc, int64_t n_total, int64_t row_offset, int64_t n_rows,
                         int64_t D, int dtype, int want_grad, void* workspace, int64_t workspace_bytes, void* stream);
int tvae_mmd_prep_local(const void* z_loc, const void* y_loc, int64_t n_total, int64_t row_offset, int64_t n_rows,
                        int64_t D, int dtype, int want_grad, void* workspace, int64_t workspace_bytes, void* stream);
int tvae_mmd_tiles(const void* z_loc, int64_t n_total, int64_t row_offset, int64_t n_rows, int64_t D, int dtype,
                   double* sums3, float* loss, double loss_scale, void* d_z, double grad_scale, int phase,
                   void* workspace, int64_t workspace_bytes, void* stream);

/* Allocation-free variant of tvae_mmd_fwd_bwd.  NO counterpart in the reference: it exists for callers that
 * cannot allocate device memory around the call (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  Same arguments and results as tvae_mmd_fwd_bwd minus the workspace pair: the scratch
 * (padded copies, norms, partial sums, gradient slabs, rendezvous counters) lives in a device array inside the
 * library, TVAE_MMD_STATIC_SCRATCH_BYTES per GPU the library runs on, held until the process ends.  Problems
 * whose tvae_mmd_workspace_bytes_for() exceeds that are rejected with TVAE_ERR_WORKSPACE (the reference's
 * shipped shapes up to BASELINE cfg3, N = 5120 x D = 77, fit).  Calls no hipMalloc/hipFree.  All calls on one GPU
 * share that GPU's array: calls that overlap on one GPU (two streams, two host threads) race; serialise them.
 * The device of `stream` must be current.  The first call on a device looks the array's address up
 * (hipGetSymbolAddress): make it once outside graph capture. */
#define TVAE_MMD_STATIC_SCRATCH_BYTES (64ll << 20)
int tvae_mmd_fwd_bwd_static(const void* z_all, const void* y_all,
                            int64_t n_total, int64_t row_offset, int64_t n_rows, int64_t D, int dtype,
                            double* sums3, float* loss, double loss_scale,
                            void* d_z, double grad_scale, void* stream);

/* ---------------------------------------------------------------------------------
 * Small-batch whole step (fp32, stock decoder = Linear + LayerNorm): the path of
 * EncoderDecoderVAE.forward (autoencoders.py:94-115: encode :21-22, regulariser :110-112 and
 * :117-136, decode :38-39 and :66-67) and of its autograd, in TWO launches each, for the
 * reference's own configurations (N = B * n_eff <= 512 latent rows, latent_size <= 128,
 * d_model <= 2048 and a multiple of 16).  Same arguments and results as the sequence
 * tvae_encode_fwd -> tvae_mmd_fwd_bwd -> tvae_decode_fwd, resp. tvae_decode_bwd ->
 * tvae_encode_bwd_sum (to fp32 rounding: the sums run in a different order).
 *   tvae_small_step_supported  1 when the shape is inside the path
 *   fwd: prior (N, L) fp32; d_latent_mmd (N, L) out, nullable (no gradient wanted);
 *        loss_scale / grad_scale as in tvae_mmd_fwd_bwd
 *   bwd: d_reg device scalar (upstream gradient of reg_loss), nullable together with
 *        d_latent_mmd; every output pointer is nullable
 * --------------------------------------------------------------------------------- */
int tvae_small_step_supported(int64_t B, int64_t n_eff, int64_t d, int64_t L);
int64_t tvae_small_step_fwd_workspace_bytes(int64_t B, int64_t n_eff, int64_t d, int64_t L);
int64_t tvae_small_step_bwd_workspace_bytes(int64_t B, int64_t n_eff, int64_t d, int64_t L);
int tvae_small_step_fwd(const void* h, int64_t h_stride_b, int64_t h_stride_s, const void* We, const void* be,
                        const void* Wd, const void* bd, const void* gamma, const void* beta, float eps,
                        const void* prior, int64_t B, int64_t n_eff, int64_t d, int64_t L, int dtype,
                        void* latent, void* recon, void* pre_norm, float* mean, float* rstd, void* d_latent_mmd,
                        double* sums3, float* loss, double loss_scale, double grad_scale,
                        void* workspace, int64_t workspace_bytes, void* stream);
int tvae_small_step_bwd(const void* d_recon, const float* d_reg, const void* latent, const void* pre_norm,
                        const float* mean, const float* rstd, const void* gamma, const void* Wd, const void* We,
                        const void* h, int64_t h_stride_b, int64_t h_stride_s, const void* d_latent_mmd,
                        void* dh, void* dWe, void* dbe, void* dWd, void* dbd, void* dgamma, void* dbeta,
                        int64_t B, int64_t S, int64_t n_eff, int64_t d, int64_t L, int dtype,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------
 * Next to the path.
 *
 * Upsample of reconstructed_encoding into decoder memory.  Replaces
 * transformers.models.funnel.modeling_funnel.upsample as called at transformer_vae/model.py:265-271:
 *     stride == 1: identity;  otherwise repeat every token `stride` times along the sequence
 *     (the first token is kept single when separate_cls; truncate_seq appends stride-1 zero rows)
 *     and cut to target_len.
 *   x (B, n, d) contiguous -> out (B, out_len, d), out_len = tvae_upsample_out_len(...) (host-only).
 * Backward: d_x (B, n, d) = sum of d_out over the positions that copied each row. */
int64_t tvae_upsample_out_len(int64_t n, int64_t stride, int64_t target_len, int separate_cls, int truncate_seq);
int tvae_upsample_fwd(const void* x, void* out, int64_t B, int64_t n, int64_t d, int64_t stride,
                      int64_t target_len, int separate_cls, int truncate_seq, int dtype, void* stream);
int tvae_upsample_bwd(const void* d_out, void* d_x, int64_t B, int64_t n, int64_t d, int64_t stride,
                      int64_t target_len, int separate_cls, int truncate_seq, int dtype, void* stream);

/* Spherical interpolation of latent rows.  Replaces slerp, transformer_vae/utils.py:17-34 (callers:
 * trainer.py:290,294,300):  out = sin((1-r)w)/sin(w) * t1 + sin(r w)/sin(w) * t2 with
 * w = acos(<t1/|t1|, t2/|t2|>) per row.  t1, t2, out (rows, dim); `ratios` = per-row r on the device
 * (fp32), or NULL to use the scalar `ratio`.  Forward only (its callers work on detached latents). */
int tvae_slerp(const void* t1, const void* t2, const float* ratios, float ratio, void* out,
               int64_t rows, int64_t dim, int dtype, void* stream);

/* The kernel matrix itself.  Replaces EncoderDecoderVAE._compute_kernel, autoencoders.py:117-126, for callers that want the
 * (nx, ny) matrix (subclasses, notebooks):  K[i][j] = exp(-(sum_c (x[i][c] - y[j][c])^2 / D) / D), direct-difference form in
 * fp32 like the reference.  The training path never materialises it (tvae_mmd_fwd_bwd).  x (nx, D), y (ny, D), K (nx, ny),
 * all fp32.  Backward: gK (nx, ny) cotangent of K -> dx (nx, D), dy (ny, D), either nullable. */
int tvae_rbf_kernel_fwd(const float* x, const float* y, int64_t nx, int64_t ny, int64_t D, float* K, void* stream);
int tvae_rbf_kernel_bwd(const float* x, const float* y, const float* K, const float* gK, int64_t nx, int64_t ny, int64_t D,
                        float* dx, float* dy, void* stream);

/* Measurement hook — no counterpart in the reference.  Arms the NEXT tvae_mmd_fwd_bwd call made
 * on this host thread: hipEventRecord(ev_start) / (ev_stop) are enqueued on that call's stream
 * immediately before / after the dominant kernel (mmd_tile_kernel), so bench.py can time that
 * kernel live with HIP events.  One-shot; pass NULLs to disarm.  Events are hipEvent_t as void*. */
int tvae_profile_next_mmd(void* ev_start, void* ev_stop);

#ifdef __cplusplus
}
#endif
#endif /* TVAE_HIP_H */
