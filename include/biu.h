/*
 * biu.h -- C ABI of the MI355X-native U-Net hot path (libbiu_hip.so).
 *
 * The reference (danihae/bio-image-unet) has no FFI: its seam is class injection of an nn.Module whose
 * forward/backward dispatch to torch.nn primitives.  Each entry point below replaces one of
 * those primitive call sites with a hand-written gfx950 kernel; the citation on every declaration names the
 * reference line whose arithmetic it takes over.  Paths are relative to /root/reference/bio_image_unet.
 *
 * Conventions
 *   - Activations live in HBM channels-last: element (n,d,h,w,c) of a biu_act is at
 *         p + (((n*D + d)*H + h)*W + w) * pitch + c          (2-D tensors use D = 1)
 *     `pitch` >= c lets several tensors share one buffer as channel slices, which is how the
 *     skip-concat (unet/unet.py:62-67) costs zero bytes: producers write straight into their slice.
 *   - dtype: BIU_F32 or BIU_BF16 for activations / activation gradients.  Parameters, parameter
 *     gradients and all BatchNorm vectors are always fp32.
 *   - A biu_xform is the per-channel epilogue of the *producer* applied by the *consumer* while loading:
 *         T(v) = max(t, slope[c]*t),  t = scale[c]*v + shift[c]        (0 <= slope <= 1)
 *     i.e. BatchNorm-affine followed by LeakyReLU.  NULL members mean scale=1 / shift=0 / slope=1.
 *     Spatial zero padding is applied AFTER T (padding pads the activated tensor).
 *   - Every function only enqueues work on `stream`; no allocation, no host synchronisation.  Outputs and
 *     workspaces are caller-owned (PyTorch's caching allocator in practice).
 *   - Return value: 0 = BIU_OK, otherwise a negative biu_status; biu_last_error() gives a message.
 *     Nothing throws across this boundary.
 */
#ifndef BIU_H
#define BIU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* biu_stream;          /* hipStream_t */

enum { BIU_F32 = 0, BIU_BF16 = 1 };

typedef enum {
    BIU_OK = 0,
    BIU_ERR_SHAPE = -1,            /* operand shapes inconsistent with each other             */
    BIU_ERR_UNSUPPORTED = -2,      /* valid request that no kernel covers                     */
    BIU_ERR_ALIGN = -3,            /* pointer / pitch alignment requirement violated          */
    BIU_ERR_WORKSPACE = -4,        /* workspace too small                                     */
    BIU_ERR_LAUNCH = -5            /* hipLaunchKernel reported an error                       */
} biu_status;

typedef struct {
    void*   p;                     /* first element of channel 0 of the slice                 */
    int32_t n, d, h, w;            /* batch and spatial extent                                */
    int32_t c;                     /* channels in this slice                                  */
    int32_t pitch;                 /* elements between consecutive voxels                     */
} biu_act;

typedef struct {
    const float* scale;
    const float* shift;
    const float* slope;
} biu_xform;

const char* biu_last_error(void);
int  biu_version(void);

/* How fp32 tensors are multiplied by the 2-D 3x3 convolution and ConvTranspose kernels (forward, data gradient, weight gradient).
 *   2 "bf16x6" (default): every operand is split hi + mid + lo in bf16 while it is staged (24 significant bits: the split is exact up to
 *                2^-24), a product is the six terms of order >= 2^-16 -- hi*hi' + hi*mid' + mid*hi' + mid*mid' + hi*lo' + lo*hi' -- on the
 *                bf16 matrix pipe with fp32 accumulation: <= 2^-23 relative per product, i.e. as good as an fp32 product's own rounding, at
 *                2.7 x the matrix-pipe rate of the fp32 MFMA.  Tensors, accumulators and every other kernel stay fp32.
 *   0 "exact"  : v_mfma_f32_32x32x2_f32 -- IEEE fp32 products, fp32 accumulation.
 *   1 "bf16x3" : hi + lo, three terms: <= 2^-15 relative per product (32 x tighter than the TF32 products torch.backends.cudnn
 *                .allow_tf32 = True -- the reference's default on Ampere and later -- uses), twice the rate of mode 2.
 * Process-wide; must be chosen before the first fp32 convolution or weight packing call (the packed weights differ), else BIU_ERR_UNSUPPORTED.
 * BIU_FP32_PRODUCTS=exact|bf16x3|bf16x6 in the environment selects the mode when this function was never called.
 * Replaces: torch.backends.cudnn.allow_tf32 / torch.set_float32_matmul_precision as used around unet/train.py:70-139. */
int  biu_set_fp32_products(int mode);

/* ------------------------------------------------------------------------------------------------
 * 3x3 / 3x3x3 "same" convolution, stride 1, padding = dilation            [K1, K2 of SURVEY 2b]
 * replaces nn.Conv2d / nn.Conv3d inside the conv block: unet/unet.py:56, unet3d/unet3d.py:54,
 * siam_unet/siam_unet.py:61, multi_output_unet3d/multi_output_unet3d.py:86-87
 * ---------------------------------------------------------------------------------------------- */

/* Weight packing for the MFMA implicit-GEMM kernels.  `kind`: 0 = forward operand, 1 = data-gradient
 * operand (spatially flipped, Cin<->Cout swapped).  w is the PyTorch tensor (Cout,Cin,kd,kh,kw) fp32,
 * kd = 1 for 2-D.  biu_conv_packed_bytes returns 0 when the shape is served by the direct kernels and
 * no packing is needed. */
size_t biu_conv_packed_bytes(int kind, int cin, int cout, int kd, int kh, int kw, int dilation, int dtype);
int    biu_conv_pack(int kind, const float* w, int cin, int cout, int kd, int kh, int kw, int dtype,
                     void* packed, biu_stream stream);

/* All packings of a network in one launch.  jobs is DEVICE memory (n entries); each job packs one weight tensor exactly as
 * biu_conv_pack (transposed = 0: PyTorch (Cout, Cin, kd, kh, kw)) or biu_convt_pack (transposed = 1: (Cin, Cout, [2,] 2, 2),
 * kd = 1 | 2) would into `packed` (sized by the matching *_packed_bytes query).                                          */
typedef struct biu_pack_job {
    const void* w;
    void* packed;
    int32_t transposed, kind, cin, cout, kd, kh, kw, reserved;
} biu_pack_job;
int biu_pack_batch(const biu_pack_job* jobs, int n, int dtype, biu_stream stream);

/* Split workspace (caller-owned, like every other byte the library touches).  An fp32 3x3(x3) launch whose bricks x channel
 * tiles would fill under half of the CUs (U-Net bottlenecks) splits its INPUT channels over workgroups: every split writes a
 * partial result into its slice of `ws`, a second kernel sums the slices into the output.  biu_conv_split_workspace returns the
 * bytes such a launch writing y (and y1: the second output of a two-tensor data gradient, else NULL) from `cin` input channels
 * needs; 0 = that launch is never split.  The calls below that take (ws, ws_bytes) split only when ws_bytes covers the query;
 * with ws == NULL (or too small) they run the same arithmetic unsplit.  The library keeps no scratch of its own: nothing is
 * allocated, cached or freed behind the caller's back, so a launch captured in a hipGraph refers to caller memory only.       */
size_t biu_conv_split_workspace(int cin, const biu_act* y, const biu_act* y1, int kd, int kh, int kw, int dilation, int dtype);

/* y = conv(T(x), w) + bias.  w: PyTorch layout fp32; packed: result of biu_conv_pack(kind 0), or NULL
 * when biu_conv_packed_bytes() returned 0 for this shape.  ws: biu_conv_split_workspace(x->c, y, NULL, ...) bytes or NULL.  */
int biu_conv_fwd(const biu_act* x, const biu_xform* xf, const float* w, const void* packed,
                 const float* bias, int kd, int kh, int kw, int dilation,
                 const biu_act* y, void* ws, size_t ws_bytes, int dtype, biu_stream stream);

/* Same, and additionally emits the BatchNorm statistics partials of y: float[nblk][cout][2] = (sum, sum of squares)
 * per block.  On the MFMA path they come out of the convolution's own epilogue (no extra pass over y); otherwise a
 * separate reduction runs.  bn_partial must hold biu_conv_fwd_stats_floats(y, kd) floats; *bn_nblk receives nblk.      */
size_t biu_conv_fwd_stats_floats(const biu_act* y, int kd);
int biu_conv_fwd_stats(const biu_act* x, const biu_xform* xf, const float* w, const void* packed,
                       const float* bias, int kd, int kh, int kw, int dilation, const biu_act* y,
                       float* bn_partial, size_t bn_partial_floats, int* bn_nblk, void* ws, size_t ws_bytes, int dtype,
                       biu_stream stream);

/* dx = conv_transpose_of_the_above(dy): dx[v,ci] = sum_{tap,co} dy[v - off(tap), co] * w[co,ci,tap].
 * packed: result of biu_conv_pack(kind 1) or NULL.  accumulate != 0 adds into dx.
 * ws: biu_conv_split_workspace(dy->c, dx, NULL, ...) bytes or NULL.                                      */
int biu_conv_bwd_data(const biu_act* dy, const float* w, const void* packed,
                      int kd, int kh, int kw, int dilation,
                      const biu_act* dx, int accumulate, void* ws, size_t ws_bytes, int dtype, biu_stream stream);

/* Data gradient that also emits the BatchNorm-backward partial sums (biu_bn_bwd_reduce's output) of the conv block that
 * PRODUCED the tensor whose gradient dx is: y_up is that block's raw conv output, (scale, shift, slope, mean, invstd) its
 * transform / saved statistics.  Use only when this call writes the complete gradient of that tensor.  `partial` holds
 * biu_bwd_data_bnred_floats(dx, kd, transposed) floats; *nblk receives the number of partial rows.                    */
size_t biu_bwd_data_bnred_floats(const biu_act* dx, int kd, int transposed);
int biu_conv_bwd_data_bnred(const biu_act* dy, const float* w, const void* packed, int kd, int kh, int kw, int dilation,
                            const biu_act* dx, const biu_act* y_up, const float* scale, const float* shift,
                            const float* slope, const float* mean, const float* invstd, float* partial,
                            size_t partial_floats, int* nblk, void* ws, size_t ws_bytes, int dtype, biu_stream stream);
int biu_convt_bwd_data_bnred(const biu_act* dy, const float* w, const void* packed, int kd, const biu_act* dx,
                             const biu_act* y_up, const float* scale, const float* shift, const float* slope,
                             const float* mean, const float* invstd, float* partial, size_t partial_floats, int* nblk,
                             int dtype, biu_stream stream);

/* dw[co,ci,tap] = sum_v T(x)[v + off(tap), ci] * dy[v, co]  (PyTorch layout fp32, overwritten);
 * dbias[co] = sum_v dy[v,co] (may be NULL).  ws: biu_conv_bwd_weight_workspace() bytes.                 */
size_t biu_conv_bwd_weight_workspace(int cin, int cout, int kd, int kh, int kw, int dtype);
int biu_conv_bwd_weight(const biu_act* x, const biu_xform* xf, const biu_act* dy,
                        int kd, int kh, int kw, int dilation,
                        float* dw, float* dbias, void* ws, size_t ws_bytes, int dtype, biu_stream stream);

/* Weight gradient with the BatchNorm+LeakyReLU backward of the conv's own output fused into its tile loader:
 * on entry `da` = d loss / d a (a = T(y)); on return it holds d loss / d y (what biu_bn_bwd_apply would write), and
 * dw is the weight gradient w.r.t. that dy.  coefA/B/C come from biu_bn_bwd_finalize.  (The conv bias gradient is
 * identically zero in front of a train-mode BatchNorm and is not produced.)                                          */
int biu_conv_bwd_weight_bn(const biu_act* x, const biu_xform* xf, const biu_act* da, const biu_act* y,
                           const float* scale, const float* shift, const float* slope, const float* coefA,
                           const float* coefB, const float* coefC, int kd, int kh, int kw, int dilation,
                           float* dw, void* ws, size_t ws_bytes, int dtype, biu_stream stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm (training: batch statistics) + LeakyReLU(0.1)                          [K3, K4]
 * replaces nn.BatchNorm2d/3d + nn.LeakyReLU: unet/unet.py:57-58, unet3d/unet3d.py:55-56
 * ---------------------------------------------------------------------------------------------- */
#define BIU_BN_MAX_PARTIALS 1024
/* partial: float[nblk][c][2]; returns nblk actually written through *nblk_out (<= BIU_BN_MAX_PARTIALS). */
int biu_bn_stats(const biu_act* y, float* partial, int* nblk_out, int dtype, biu_stream stream);

/* Training-mode finalize: mean/biased var from the partials (fp64 merge), running stats updated with the
 * unbiased variance and `momentum`, and the consumer transform (scale, shift) = (g*r, b - mean*g*r).
 * save_mean / save_invstd are kept for the backward pass.                                                */
int biu_bn_finalize(const float* partial, int nblk, int c, double count,
                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, float* scale, float* shift,
                    float* save_mean, float* save_invstd, biu_stream stream);

/* Eval-mode transform from the running statistics.                                                       */
int biu_bn_eval_affine(int c, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, biu_stream stream);

/* out = T(x) (materialises BatchNorm-affine + LeakyReLU; also plain dtype-preserving copies).            */
int biu_xform_apply(const biu_act* x, const biu_xform* xf, const biu_act* out, int dtype, biu_stream stream);

/* Backward of a = T(y), T = lrelu_slope(scale*y+shift) with scale/shift from batch statistics.
 *   reduce:   partial[nblk][c][2] = (sum dz, sum dz*yhat),  dz = da * T'(.)
 *   finalize: dgamma, dbeta and the three vectors of   dy = A*dz + B*y + C
 *   apply:    dy (may alias da)                                                                           */
int biu_bn_bwd_reduce(const biu_act* da, const biu_act* y, const float* scale, const float* shift,
                      const float* slope, const float* save_mean, const float* save_invstd,
                      float* partial, int* nblk_out, int dtype, biu_stream stream);
int biu_bn_bwd_finalize(const float* partial, int nblk, int c, double count,
                        const float* scale, const float* save_mean, const float* save_invstd,
                        float* dgamma, float* dbeta, float* coefA, float* coefB, float* coefC,
                        biu_stream stream);
/* The same for a BatchNorm in EVAL mode (running statistics are constants: model.eval(); loss.backward(), the frozen-BatchNorm
 * fine-tuning the reference's plain nn.BatchNorm allows, unet/unet.py:54-60).  `partial` from biu_bn_bwd_reduce called with
 * save_mean = running_mean and save_invstd = 1 / sqrt(running_var + eps):  dgamma = sum dz*yhat, dbeta = sum dz, dy = scale * dz, i.e.
 * (A, B, C) = (scale, 0, 0), and the conv bias gradient dbias = sum_v dy = scale * sum dz is no longer zero (dbias may be NULL).          */
int biu_bn_bwd_finalize_eval(const float* partial, int nblk, int c, const float* scale,
                             float* dgamma, float* dbeta, float* dbias, float* coefA, float* coefB, float* coefC,
                             biu_stream stream);
int biu_bn_bwd_apply(const biu_act* da, const biu_act* y, const float* scale, const float* shift,
                     const float* slope, const float* coefA, const float* coefB, const float* coefC,
                     const biu_act* dy, int dtype, biu_stream stream);

/* ------------------------------------------------------------------------------------------------
 * 2x2 / 2x2x2 max-pool, stride 2                                                        [K6]
 * replaces nn.MaxPool2d/3d: unet/unet.py:22-31, unet3d/unet3d.py:26-32
 * bwd routes the gradient to the FIRST maximum in (d,h,w) scan order (PyTorch tie rule).
 * nearest: F.interpolate(scale 0.5 / 2, 'nearest'), multi_output_unet3d.py:112-156     [K11]
 * ---------------------------------------------------------------------------------------------- */
int biu_maxpool_fwd(const biu_act* x, const biu_xform* xf, const biu_act* out, int dtype, biu_stream stream);
int biu_maxpool_bwd(const biu_act* x, const biu_xform* xf, const biu_act* dout, const biu_act* dx,
                    int accumulate, int dtype, biu_stream stream);
/* biu_maxpool_bwd that also emits biu_bn_bwd_reduce's partial sums for the conv block that produced x (xf = that block's
 * BatchNorm transform, mean / invstd its saved statistics).  Use when this call completes the gradient of x.
 * partial holds >= BIU_BN_MAX_PARTIALS * C * 2 floats (more lets the kernel use more workgroups); *nblk = rows written. */
int biu_maxpool_bwd_bnred(const biu_act* x, const biu_xform* xf, const biu_act* dout, const biu_act* dx, int accumulate,
                          const float* mean, const float* invstd, float* partial, size_t partial_floats, int* nblk,
                          int dtype, biu_stream stream);
int biu_nearest_down_fwd(const biu_act* x, const biu_xform* xf, const biu_act* out, int dtype, biu_stream stream);
int biu_nearest_down_bwd(const biu_act* dout, const biu_act* dx, int accumulate, int dtype, biu_stream stream);
int biu_nearest_up_fwd(const biu_act* x, const biu_xform* xf, const biu_act* out, int dtype, biu_stream stream);
int biu_nearest_up_bwd(const biu_act* dout, const biu_act* dx, int accumulate, int dtype, biu_stream stream);

/* Nearest-neighbour up-sampling (x2 per axis) FOLDED into the 3x3x3 convolution behind it -- forward only.
 * replaces F.interpolate(scale_factor=2, mode='nearest') + self.upN_conv's nn.Conv3d:
 * multi_output_unet3d/multi_output_unet3d.py:138-139,147-148,156-157 (conv3d block: :86-87).
 *   y[2v + p] = bias + sum_{t in {0,1}^3} W'[p][t] . T(x)[v + t - 1 + p]      p = output parity per axis, x = the coarse tensor (D,H,W),
 *   W'[p][t] = the sum of the conv's taps that read the same coarse voxel (8 x 8 tap groups instead of 27 taps: 0.30 x the FLOPs),
 * identical to the convolution of the up-sampled tensor up to fp32 summation order (coarse zero padding = fine zero padding).
 * y is (2D, 2H, 2W).  biu_upconv_bwd_data is the matching data gradient straight onto the coarse tensor (replaces biu_conv_bwd_data +
 * biu_nearest_up_bwd):   dx[u] (+)= sum_p sum_{s in {0,1}^3} W'[p][1 - s]^T . dy[2 (u - p + s) + p].
 * biu_upconv_bwd_weight_bn is the weight gradient, with the block's BatchNorm+LeakyReLU backward in its loader exactly as
 * biu_conv_bwd_weight_bn has it (da -> dy in place; y = NULL: da already is dy): per parity class
 *   G[p][t] = sum_v dy[2v + p] (x) T(x)[v + t - 1 + p],   dw[k] = sum_p G[p][t_p(k)]   (t_p(k): the coarse tap fine tap k reads under parity p);
 * with all three the up-sampled tensor is never materialised (no biu_nearest_up_fwd / _bwd).
 * biu_upconv_ok: shapes / channels the folded kernels serve (else: up-sample + biu_conv_*).  biu_upconv_pack folds and packs
 * w (Cout, Cin, 3, 3, 3) fp32 into `packed` (biu_upconv_packed_bytes; kind 0 = forward image, 1 = data-gradient image).
 * bn_partial may be NULL (no statistics); with it, *bn_nblk partial rows of [Cout][2] (sum, sum of squares) are written, as
 * biu_conv_fwd_stats does (biu_upconv_fwd_stats_floats sizes the buffer). */
int    biu_upconv_ok(const biu_act* x, const biu_act* y, int dtype);
size_t biu_upconv_packed_bytes(int kind, int cin, int cout, int dtype);
int    biu_upconv_pack(int kind, const float* w, int cin, int cout, int dtype, void* packed, biu_stream stream);
int    biu_upconv_bwd_data(const biu_act* dy, const void* packed, const biu_act* dx, int accumulate, int dtype, biu_stream stream);
size_t biu_upconv_bwd_weight_workspace(int cin, int cout, int dtype);
int    biu_upconv_bwd_weight_bn(const biu_act* x, const biu_xform* xf, const biu_act* da, const biu_act* y, const float* scale,
                                const float* shift, const float* slope, const float* coefA, const float* coefB, const float* coefC,
                                float* dw, void* ws, size_t ws_bytes, int dtype, biu_stream stream);
size_t biu_upconv_fwd_stats_floats(const biu_act* x, const biu_act* y);
int    biu_upconv_fwd(const biu_act* x, const biu_xform* xf, const void* packed, const float* bias, const biu_act* y,
                      float* bn_partial, size_t bn_partial_floats, int* bn_nblk, int dtype, biu_stream stream);

/* ------------------------------------------------------------------------------------------------
 * ConvTranspose k=2, stride=2 (non-overlapping)                                         [K7]
 * replaces nn.ConvTranspose2d/3d: unet/unet.py:38-47, unet3d/unet3d.py:40-42
 * w: PyTorch layout (Cin, Cout, kd, 2, 2) fp32 with kd = 2 (3-D) or 1 (2-D).
 * ---------------------------------------------------------------------------------------------- */
/* BCEDiceLoss (unet/losses.py:78-112) over fp32 NC[D]HW logits / targets as the heads emit them, one pass each way.
 * fwd: partial[n][biu_bce_dice_blocks(per_sample)][4] = per-block sums of (bce(l,t), p, t, p*t), p = sigmoid(l).
 * bwd: dlogits_i (+)= coef[n][0]*(p_i - t_i) + (coef[n][1] + coef[n][2]*t_i) * p_i*(1 - p_i).                        */
int biu_bce_dice_blocks(long long per_sample);
int biu_bce_dice_fwd(const float* logits, const float* target, int n, long long per_sample, float* partial, biu_stream stream);
int biu_bce_dice_bwd(const float* logits, const float* target, int n, long long per_sample, const float* coef, float* dlogits,
                     int accumulate, biu_stream stream);

/* SmoothL1 (beta 1, mean) between neighbouring BATCH entries of fp32 logits -- nn.SmoothL1Loss()(y_logits[1:], y_logits[:-1]),
 * the "time" term of the 3-D trainer (unet3d/train.py:140-145).  fwd: partial[biu_pair_smooth_l1_blocks(pairs)] block sums of
 * h(l[i + per_sample] - l[i]), pairs = (n - 1) * per_sample.  bwd: dlogits (+)= coef[0] * d sum / d logits (coef on the device).      */
int biu_pair_smooth_l1_blocks(long long pairs);
int biu_pair_smooth_l1_fwd(const float* logits, int n, long long per_sample, float* partial, biu_stream stream);
int biu_pair_smooth_l1_bwd(const float* logits, int n, long long per_sample, const float* coef, float* dlogits, int accumulate,
                           biu_stream stream);
/* The scalar arithmetic of the fused segmentation losses in two one-block launches (instead of ~20 element-wise torch kernels on [n, 4]
 * tensors per step).  biu_seg_loss_finish: partial = biu_bce_dice_fwd's [n][nb][4] rows, time_partial = biu_pair_smooth_l1_fwd's nbt
 * sums or NULL -> saved[1 + 4n + n + 3] = { loss, sums[n][4] (BCE, P, T, P.T), den[n] = P + T + smooth, tp, tden, Tversky } with
 *   loss = a_bce * sum BCE / (n per) + a_dice * (1 - mean_n 2 (PT_n + smooth) / den_n)                    (unet/losses.py:78-112)
 *        + [has_tversky] (1 - Tv) or log cosh(1 - Tv),  Tv = (TP + s) / (TP + alpha FP + beta FN + s)     (unet/losses.py:145-239)
 *        + w_time * sum SmoothL1 / ((n - 1) per)                                                          (unet3d/train.py:140-145)
 * biu_seg_loss_coef: g = d / d loss (device scalar) -> coef[n][3] for biu_bce_dice_bwd, time_coef[1] for biu_pair_smooth_l1_bwd.      */
int biu_seg_loss_finish(const float* partial, int n, int nb, long long per_sample, const float* time_partial, int nbt,
                        float a_bce, float a_dice, float smooth, int has_tversky, float tv_alpha, float tv_beta, float tv_smooth,
                        int logcosh, float w_time, float* saved, biu_stream stream);
int biu_seg_loss_coef(const float* g, const float* saved, int n, long long per_sample, float a_bce, float a_dice, float smooth,
                      int has_tversky, float tv_alpha, float tv_beta, float tv_smooth, int logcosh, float w_time, float* coef,
                      float* time_coef, biu_stream stream);
/* d loss / d logits of one head from the caller's gradients w.r.t. its logits and / or its activated output (act as in
 * biu_head_fwd; multi_output_unet3d.py:97-104), written to channels [dst_c0, dst_c0 + ch) of a fp32 [n, dst_channels, spatial]
 * tensor -- the stacked operand of one biu_head_bwd over all heads that share a trunk.                                    */
int biu_head_dlogits(const float* g_logits, const float* g_act, const float* activated, int act, int n, int ch, long long spatial,
                     float* dst, int dst_channels, int dst_c0, biu_stream stream);

/* Trilinear x2 up-sampling, align_corners = False (F.interpolate(scale_factor=2, mode='trilinear'),
 * unet3d/unet3d.py:82,89,96): out = interp(T(x)); depth is doubled when out->d == 2 * x->d, kept when equal.
 * bwd: dx (+)= adjoint(dout).                                                                                        */
int biu_trilinear_up_fwd(const biu_act* x, const biu_xform* xf, const biu_act* out, int dtype, biu_stream stream);
int biu_trilinear_up_bwd(const biu_act* dout, const biu_act* dx, int accumulate, int dtype, biu_stream stream);

/* Depth-wise cross-correlation of two equal 2-D maps, padding='same' (Siam_UNet.depthwise_xcorr,
 * siam_unet/siam_unet.py:75-83): out[n,y,x,c] = sum_ij T(cur)[n,y+i-ph,x+j-pw,c] * T(prev)[n,i,j,c], ph=(H-1)/2, pw=(W-1)/2.
 * bwd needs identity transforms (materialised operands): dcur, dprev (+)= gradients.                                  */
int biu_xcorr_fwd(const biu_act* cur, const biu_xform* xc, const biu_act* prev, const biu_xform* xp, const biu_act* out,
                  int dtype, biu_stream stream);
int biu_xcorr_bwd(const biu_act* cur, const biu_xform* xc, const biu_act* prev, const biu_xform* xp, const biu_act* dout,
                  const biu_act* dcur, const biu_act* dprev, int accumulate, int dtype, biu_stream stream);

/* The same conv block on a channel CONCATENATION (x0 | x1) whose two parts stay in separate dense buffers -- the decoder's
 * torch.cat (unet/unet.py:62-67, unet3d/unet3d.py:60-61) without a concat buffer.  w / packed / dw have Cin = x0->c + x1->c in
 * that order.  Only shapes the MFMA kernels serve: biu_conv_cat_ok returns 1 when all three calls will succeed (channel counts
 * that are multiples of 32 -- x0's of 64 when the total is an even number of 32-tiles --, dilation 1, aligned dense rows).
 * fwd: bn_partial may be NULL (no statistics).  bwd_weight: y == NULL gives the plain weight gradient (da is dy), otherwise the
 * BatchNorm backward is fused exactly as in biu_conv_bwd_weight_bn.  bwd_data writes dx0 and dx1 (each with its own
 * accumulate flag).  (ws, ws_bytes) of fwd / bwd_data: biu_conv_split_workspace(x0->c + x1->c, y, NULL, ...) resp.
 * (dy->c, dx0, dx1, ...) bytes, or NULL.                                                                                  */
int biu_conv_cat_ok(const biu_act* x0, const biu_act* x1, const biu_act* y, int kd, int kh, int kw, int dilation, int dtype);
int biu_conv_fwd_cat(const biu_act* x0, const biu_xform* xf0, const biu_act* x1, const biu_xform* xf1, const float* w,
                     const void* packed, const float* bias, int kd, int kh, int kw, int dilation, const biu_act* y,
                     float* bn_partial, size_t bn_partial_floats, int* bn_nblk, void* ws, size_t ws_bytes, int dtype,
                     biu_stream stream);
int biu_conv_bwd_data_cat(const biu_act* dy, const float* w, const void* packed, int kd, int kh, int kw, int dilation,
                          const biu_act* dx0, int accumulate0, const biu_act* dx1, int accumulate1, void* ws, size_t ws_bytes,
                          int dtype, biu_stream stream);
int biu_conv_bwd_weight_cat(const biu_act* x0, const biu_xform* xf0, const biu_act* x1, const biu_xform* xf1,
                            const biu_act* da, const biu_act* y, const float* scale, const float* shift, const float* slope,
                            const float* coefA, const float* coefB, const float* coefC, int kd, int kh, int kw, int dilation,
                            float* dw, void* ws, size_t ws_bytes, int dtype, biu_stream stream);

/* MFMA operand packing, as for the 3x3 kernels.  kind 0 = forward operand, 1 = data-gradient operand.          */
size_t biu_convt_packed_bytes(int kind, int cin, int cout, int kd, int dtype);
int    biu_convt_pack(int kind, const float* w, int cin, int cout, int kd, int dtype, void* packed, biu_stream stream);
/* y[2v + a, co] = bias[co] + sum_ci T(x)[v, ci] * w[ci, co, a]                                                 */
int biu_convt_fwd(const biu_act* x, const biu_xform* xf, const float* w, const void* packed, const float* bias,
                  int kd, const biu_act* y, int dtype, biu_stream stream);
/* dx[v, ci] (+)= sum_{a, co} dy[2v + a, co] * w[ci, co, a]                                                      */
int biu_convt_bwd_data(const biu_act* dy, const float* w, const void* packed, int kd, const biu_act* dx,
                       int accumulate, int dtype, biu_stream stream);
/* dw[ci, co, a] = sum_v T(x)[v, ci] * dy[2v + a, co]; dbias[co] = sum dy (may be NULL)                          */
size_t biu_convt_bwd_weight_workspace(int cin, int cout, int kd, int dtype);
int biu_convt_bwd_weight(const biu_act* x, const biu_xform* xf, const biu_act* dy, int kd, float* dw, float* dbias,
                         void* ws, size_t ws_bytes, int dtype, biu_stream stream);

/* ConvTranspose(k2, s2) + concat + 3x3x3 convolution of a decoder level as ONE op, the up half folded onto the coarse tensor.
 * replaces self.upN (nn.ConvTranspose3d) + torch.cat + the first conv of the decode block: unet3d/unet3d.py:40-42,84-90
 * (multi_output_unet3d/multi_output_unet3d.py with use_interpolation=False likewise); no non-linearity sits between the two.
 *   y = conv(concat(convT(T(x_low)) + b_T, T(skip))) + b_conv
 *     = conv_skip(T(skip)) + b_conv + sum_{taps k inside} Wb[k] + sum_{t in {0,1}^3} W'[p][t] . T(x_low)[v + t - 1 + p]        at y[2v + p],
 *   W'[p][t][ci][co] = sum_{k in class(p,t)} sum_c W_conv[co][c][k] W_T[ci][c][q(p,k)],  Wb[k][co] = sum_c W_conv[co][c][k] b_T[c]:
 * the same function of the four parameter tensors (the up-sampled tensor is not materialised, 8 x 8 instead of 27 taps on its channels).
 * w_conv (Cout, cup + cskip, 3,3,3) with the concat order (up | skip); w_t (Cin_low, cup, 2,2,2); packed: biu_foldt_packed_bytes, refreshed
 * by biu_foldt_pack whenever one of the four tensors changes.  biu_foldt_ok: the kernels serve the level and it is large enough to pay for the
 * weight-space work of every step (3 x 216 small GEMMs of Cout x cup x Cin_low; BIU_FOLDT=always in the environment drops the size test).  bn_partial as in biu_upconv_fwd (biu_foldt_fwd_stats_floats). */
int    biu_foldt_ok(const biu_act* x_low, const biu_act* skip, const biu_act* y, int dtype);
size_t biu_foldt_packed_bytes(int cin_low, int cskip, int cout, int dtype);
int    biu_foldt_pack(const float* w_conv, const float* b_conv, const float* w_t, const float* b_t, int cin_low, int cup, int cskip, int cout,
                      int dtype, void* packed, biu_stream stream);
size_t biu_foldt_fwd_stats_floats(const biu_act* x_low, const biu_act* y);
/* which form biu_foldt_fwd takes: 0 = brick kernels (the skip half + biases stored, the border shell corrected, the fold accumulated on top);
 * 1 = rolling-window kernels for 64 -> 32 | 32-channel levels (the fold + border-state bias stored first, the skip half accumulated onto it and
 * rounded once).  Same function; the storage roundings of the intermediate differ (a bit-level checker needs to know: tests/insitu.py). */
int    biu_foldt_fwd_form(const biu_act* x_low, const biu_act* skip, const biu_act* y, int dtype);
int    biu_foldt_fwd(const biu_act* x_low, const biu_xform* xf_low, const biu_act* skip, const biu_xform* xf_skip, const void* packed,
                     const biu_act* y, float* bn_partial, size_t bn_partial_floats, int* bn_nblk, int dtype, biu_stream stream);
/* backward of the same op.  biu_foldt_bwd_data: d skip (the conv's data gradient restricted to the skip channels) and d x_low (the composed fold
 * transposed: replaces biu_conv_bwd_data_cat's up half + biu_convt_bwd_data); y_low != NULL additionally emits the BatchNorm-backward sums of
 * x_low's producer as biu_convt_bwd_data_bnred does (acc_low must be 0; biu_foldt_bwd_data_bnred_floats sizes `partial`).
 * biu_foldt_bwd_weight_bn: BatchNorm+LeakyReLU backward of the block in the loader (da -> dy in place; y = NULL: da already is dy), then
 *   dw_conv (Cout, cup + cskip, 27) in full, dw_t (Cin_low, cup, 8) and db_t (cup; may be NULL) by the chain rule from
 *   G[p][t] = sum_v dy[2v + p] (x) T(x_low)[v + t - 1 + p]:  dw_conv[.., c < cup, k] = sum_p W_T[., c, q(p,k)] . G[p][t_p(k)] + b_T[c] S_k,
 *   dw_t[ci, c, q] = sum_{(p,k): q(p,k) = q} W_conv[., c, k] . G[p][t_p(k)][., ci],  db_t[c] = sum_k W_conv[., c, k] . S_k with S_k the sum of dy over
 *   the voxels whose tap k stays inside the tensor = dy_sum - (border sums).  dy_sum (Cout floats) = sum_v dy per channel; NULL = identically
 *   zero, which holds behind a train-mode BatchNorm only: a plain dy (y = NULL) or an eval-mode BatchNorm (coefB = coefC = 0, see
 *   biu_bn_bwd_finalize_eval) must pass it (BIU_ERR_UNSUPPORTED otherwise when the ConvT has a bias). */
size_t biu_foldt_bwd_data_bnred_floats(const biu_act* dx_low);
int    biu_foldt_bwd_data(const biu_act* dy, const void* packed, const biu_act* dx_low, int acc_low, const biu_act* dskip, int acc_skip,
                          const biu_act* y_low, const float* scale, const float* shift, const float* slope, const float* mean,
                          const float* invstd, float* partial, size_t partial_floats, int* nblk, void* ws, size_t ws_bytes, int dtype,
                          biu_stream stream);
size_t biu_foldt_bwd_weight_workspace(int cin_low, int cskip, int cout, int dtype);
int    biu_foldt_bwd_weight_bn(const biu_act* x_low, const biu_xform* xf_low, const biu_act* skip, const biu_xform* xf_skip, const biu_act* da,
                               const biu_act* y, const float* scale, const float* shift, const float* slope, const float* coefA,
                               const float* coefB, const float* coefC, const float* dy_sum, const float* w_conv, const float* w_t, const float* b_t,
                               int cup, float* dw_conv, float* dw_t, float* db_t, void* ws, size_t ws_bytes, int dtype, biu_stream stream);
/* the same call in parts, for a caller that overlaps everything only the optimizer waits for with the rest of its backward.  `phases` is a
 * mask: 1 = the skip half (da -> dy in place by the fused BatchNorm backward, the skip slice of dw_conv); 4 = G on the finished dy into ws;
 * 2 = the border sums of dy and the chain rule on the tables (the up slice of dw_conv, dw_t, db_t).  Parts 4 and 2 read dy (= da after part
 * 1), x_low, ws, dy_sum and the parameters: they may be enqueued on ANOTHER stream once part 1 is complete there (the caller orders them
 * with an event, part 4 before part 2, and leaves dy and ws alone in between; reading dy meanwhile is fine); 7 = biu_foldt_bwd_weight_bn. */
int    biu_foldt_bwd_weight_bn_phase(const biu_act* x_low, const biu_xform* xf_low, const biu_act* skip, const biu_xform* xf_skip, const biu_act* da,
                                     const biu_act* y, const float* scale, const float* shift, const float* slope, const float* coefA,
                                     const float* coefB, const float* coefC, const float* dy_sum, const float* w_conv, const float* w_t,
                                     const float* b_t, int cup, float* dw_conv, float* dw_t, float* db_t, void* ws, size_t ws_bytes, int dtype,
                                     int phases, biu_stream stream);

/* ------------------------------------------------------------------------------------------------
 * 1x1(x1) head + activation                                                             [K9]
 * replaces final Conv + torch.sigmoid: unet/unet.py:51,103-104, unet3d/unet3d.py:50,98-99,
 * multi_output_unet3d.py:80-82,164-168.  Outputs are fp32 NCDHW (what the caller's loss consumes).
 * act: 0 none, 1 sigmoid, 2 tanh, 3 relu.  logits / activated may each be NULL.
 * ---------------------------------------------------------------------------------------------- */
int biu_head_fwd(const biu_act* x, const biu_xform* xf, const float* w, const float* bias, int cout,
                 int act, float* logits, float* activated, int dtype, biu_stream stream);
/* dlogits: fp32 NCDHW.  dx = W^T dlogits (activation-gradient layout); dw (cout,cin), dbias (cout) fp32,
 * overwritten; each of dx / dw / dbias may be NULL.  ws: biu_head_bwd_workspace(cin) bytes (for dw).      */
size_t biu_head_bwd_workspace(int cin);
int biu_head_bwd(const biu_act* x, const biu_xform* xf, const float* w, int cout, const float* dlogits,
                 const biu_act* dx, float* dw, float* dbias, void* ws, size_t ws_bytes, int dtype,
                 biu_stream stream);
/* biu_head_bwd that also emits biu_bn_bwd_reduce's partial sums of the conv block that produced x (xf = its BatchNorm transform,
 * mean / invstd its saved statistics); valid when the head is x's only reader.  partial: >= BIU_BN_MAX_PARTIALS * C * 2 floats. */
int biu_head_bwd_bnred(const biu_act* x, const biu_xform* xf, const float* w, int cout, const float* dlogits, const biu_act* dx,
                       float* dw, float* dbias, void* ws, size_t ws_bytes, const float* mean, const float* invstd,
                       float* partial, size_t partial_floats, int* nblk, int dtype, biu_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise helpers on activation slices
 * ---------------------------------------------------------------------------------------------- */
/* out = max(T1(a), T2(b))  -- Siam 'max' join, siam_unet/siam_unet.py:117; bwd: the winner takes dout, an exact tie
 * splits it evenly (torch.maximum's backward).                                                              */
int biu_max_join_fwd(const biu_act* a, const biu_xform* xa, const biu_act* b, const biu_xform* xb,
                     const biu_act* out, int dtype, biu_stream stream);
int biu_max_join_bwd(const biu_act* a, const biu_xform* xa, const biu_act* b, const biu_xform* xb,
                     const biu_act* dout, const biu_act* da, const biu_act* db, int accumulate,
                     int dtype, biu_stream stream);
/* dst (+)= src, both activation slices of equal shape.                                                   */
int biu_act_add(const biu_act* src, const biu_act* dst, int accumulate, int dtype, biu_stream stream);
/* NC[D]HW fp32 <-> channels-last activation (network input / gradient at the module boundary).           */
int biu_from_nchw(const float* src, const biu_act* dst, int dtype, biu_stream stream);
int biu_to_nchw(const biu_act* src, const biu_xform* xf, float* dst, int dtype, biu_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Attention gate of AttentionUnet (unet/attention_unet.py:112-181; the 1x1 conv + BatchNorm stages are biu_conv_* calls
 * with kernel size 1 and slope 1):
 *   add_relu: out = relu(T(a) + T(b))                               (:177, psi = self.relu(g1 + x1))
 *             bwd: da, db (+)= dout where the stored out > 0
 *   gate:     out[v, c] = T(e)[v, c] * sigmoid(T(psi)[v, 0])         (:178-179, psi has ONE channel)
 *             bwd: de[v, c] (+)= dout[v, c] * s ;  dpsi[v] = s (1 - s) * sum_c dout[v, c] * T(e)[v, c]   (gradient w.r.t. T(psi));
 *             de may be NULL.
 * ---------------------------------------------------------------------------------------------- */
int biu_add_relu_fwd(const biu_act* a, const biu_xform* xa, const biu_act* b, const biu_xform* xb, const biu_act* out, int dtype,
                     biu_stream stream);
int biu_add_relu_bwd(const biu_act* out, const biu_act* dout, const biu_act* da, const biu_act* db, int accumulate, int dtype,
                     biu_stream stream);
int biu_gate_fwd(const biu_act* e, const biu_xform* xe, const biu_act* psi, const biu_xform* xpsi, const biu_act* out, int dtype,
                 biu_stream stream);
int biu_gate_bwd(const biu_act* e, const biu_xform* xe, const biu_act* psi, const biu_xform* xpsi, const biu_act* dout,
                 const biu_act* de, int accumulate_e, const biu_act* dpsi, int dtype, biu_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Data formats either side of the network, kept on the device
 *   biu_from_nchw_u8 : uint8 NC[D]HW batch * scale -> channels-last activation; replaces `batch / 255` + `.to(device)` of float32
 *                      (unet/data.py:253-266 items, unet/predict.py:192-196 patches): a quarter of the H2D bytes, no float copy
 *   biu_u8_to_f32    : uint8 * scale -> fp32 (targets: masks are stored 0 / 255)
 *   biu_quantize_u8  : (p * scale) truncated to uint8 -- `(res * 255).astype('uint8')`, unet/predict.py:200
 *   biu_stitch_add   : one patch [channels, pd, ph, pw] (uint8 or fp32) times an optional weight [pd, ph, pw] added into (set != 0:
 *                      written over) acc [channels, D, H, W] / wsum [D, H, W] at origin (z0, y0, x0); 2-D: D = pd = 1
 *   biu_stitch_finish: out = acc / wsum where wsum > 0 else 0 (fp32: the ramp blend of multi_output_unet3d/predict.py:300-303), or
 *                      floor(sum_layers acc / sum_layers wsum) as uint8 (the nan-mean of overlapping uint8 tiles cast to uint8,
 *                      unet/predict.py:204-229; `layers` = 3 reproduces unet3d/predict.py:173-195)
 * ---------------------------------------------------------------------------------------------- */
int biu_from_nchw_u8(const uint8_t* src, float scale, const biu_act* dst, int dtype, biu_stream stream);
int biu_u8_to_f32(const uint8_t* src, float scale, float* dst, long long n, biu_stream stream);
int biu_quantize_u8(const float* src, float scale, uint8_t* dst, long long n, biu_stream stream);
int biu_stitch_add(const void* patch, int patch_is_u8, const float* weight, int channels, int pd, int ph, int pw, float* acc, float* wsum,
                   int D, int H, int W, int z0, int y0, int x0, int set, biu_stream stream);
int biu_stitch_finish(const float* acc, const float* wsum, int layers, int channels, long long spatial, void* out, int out_is_u8,
                      biu_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Fused multi-tensor Adam                                                               [K14]
 * replaces torch.optim.Adam(lr) step: unet/train.py:102,139 (betas 0.9/0.999, eps 1e-8, no decay).
 * One launch updates `n` parameter tensors; ptrs are device arrays of device pointers.
 * ---------------------------------------------------------------------------------------------- */
int biu_adam_step(int n, float* const* params, const float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const int64_t* numel, float lr, float beta1, float beta2,
                  float eps, int step, float grad_scale, biu_stream stream);

/* The same update with its scalars in device memory, for a step captured in a hipGraph (bio_image_unet_amd/graph.py): the graph holds
 * biu_adam_step_hyper, and biu_adam_set_hyper -- launched in front of every replay, arguments by value -- writes
 * hyper[6] = {lr, beta1, beta2, eps, grad_scale, step} for the step being replayed (ReduceLROnPlateau of unet/train.py:103 keeps working). */
int biu_adam_set_hyper(float* hyper, float lr, float beta1, float beta2, float eps, int step, float grad_scale, biu_stream stream);
int biu_adam_step_hyper(int n, float* const* params, const float* const* grads, float* const* exp_avg,
                        float* const* exp_avg_sq, const int64_t* numel, const float* hyper, biu_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Gradient-norm clipping over a table of fp32 tensors                                     [K14b]
 * replaces torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0): multi_output_unet3d/train.py:201.
 *   total = sqrt(sum_i |g_i|^2), coef = min(1, max_norm / (total + 1e-6)), g_i *= coef in place (three launches, sums in a fixed order).
 * grads / numel: device arrays as for biu_adam_step; scratch: biu_grad_clip_scratch_floats(n) floats of device memory;
 * total_norm: one device float receiving the norm before clipping, or NULL.
 * ---------------------------------------------------------------------------------------------- */
size_t biu_grad_clip_scratch_floats(int n);
int biu_grad_clip(int n, float* const* grads, const int64_t* numel, float max_norm, float* scratch, size_t scratch_floats, float* total_norm,
                  biu_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* BIU_H */
