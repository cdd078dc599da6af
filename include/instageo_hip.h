/* instageo_hip.h -- C ABI of libinstageo_hip.so: the MI355X (gfx950) hot path of InstaGeo's Prithvi
 * segmentation model (instageo/model).  Plain pointers and sizes only; every pointer is a DEVICE pointer
 * unless stated otherwise; `stream` is a hipStream_t (NULL = default stream).  All functions return 0 on
 * success or a negative IG_ERR_* code and set a thread-local message readable through ig_last_error().
 *
 * The reference has no native interface: the seam is the Python symbol instageo.model.base.PrithviSeg
 * (base.py:28,69-77).  Each entry point below replaces the ATen op(s) that the reference module calls at the
 * cited file:line (paths relative to the reference root); INTEGRATION.md shows the ctypes binding.
 *
 * bf16 tensors: raw uint16 storage.  Every bf16 tensor argument is a pair (x_hi, x_lo): x_lo == NULL selects
 * plain bf16; non-NULL selects the split "bf16x3" precision mode (value = hi + lo, products hi*hi+hi*lo+lo*hi).
 * Either all bf16 operands of a call are split or none.  Activations in the decode head are NHWC.
 * Placement of a split pair (performance only, results are the same up to fp32 summation order): when x_lo lies ABOVE x_hi, 16-byte aligned
 * and less than 4 GiB minus the tensor away -- e.g. both halves of one allocation, which is how the Python host allocates them -- the
 * linear and wide-convolution engines fetch hi and lo with ONE LDS-DMA stream ("paired" K-tiles, 5-35 % faster); any other placement runs
 * the three-pass form of the same products.
 * Conv weights (3x3 and transposed) are stored Wc[Cout][9][Cin], tap = ky*3+kx.
 * Dropout is a counter-based hash of (drop_seed + *drop_seed_dev, element index): backward regenerates the mask;
 * drop_seed_dev (device uint32, may be NULL) lets a captured graph advance the seed without new host arguments.
 */
#ifndef INSTAGEO_HIP_H
#define INSTAGEO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define IG_OK 0
#define IG_ERR_ARG (-1)
#define IG_ERR_HIP (-2)
#define IG_ERR_UNSUPPORTED (-3)

/* ---- runtime ---------------------------------------------------------------------------------------- */
const char* ig_last_error(void);
/* name of the (last) kernel the most recent MFMA entry point of this thread launched, as rocprofv3 prints it minus
 * "(anonymous namespace)::" and blanks -- bench.py keys its per-kernel roofline table by it */
const char* ig_last_kernel(void);
int ig_note_reset(void); /* forget the name (an entry point without a named kernel then reports "") */
int ig_last_grid(void);  /* workgroups of the last persistent GEMM launch of this thread (see ig_set_reserved_cus) */
int ig_version(void);
/* first 32 bits of the MD5 of this header as the library was built against it: the host mirror refuses a library whose
 * entry points were compiled from a different revision of the declarations (stale .so next to a newer header) */
int ig_header_stamp(void);
int ig_device_info(int device, char* name, int name_len, int* cu_count, int* lds_per_block, long* hbm_bytes);
/* Compute units the persistent kernels (one workgroup per CU) leave free.  Data-parallel training (Lightning DDP in the
 * reference, pipeline_utils.py:368-374) launches RCCL all-reduce kernels beside the backward GEMMs: a grid that pins all
 * 256 CUs serialises them behind a whole GEMM.  Default 0, or the IG_RESERVED_CUS environment variable. */
int ig_set_reserved_cus(int n);
int ig_get_reserved_cus(void);
/* Run-to-run deterministic reductions (the reference's Trainer runs with deterministic=True, pipeline_utils.py:373; float atomics
 * make the bias / norm / head gradients depend on arrival order).  shadow: zeroed int64[n] on the device, paralleling the flat fp32
 * gradient buffer grad_base[n]; while registered, every multi-contributor reduction into that buffer is an INTEGER add of the
 * 2^44-scaled contribution into the shadow (order-independent; one gradient element must stay below 5.2e5 in magnitude), and the
 * BatchNorm statistics are folded from per-workgroup partials in index order; call ig_det_fold(lo, hi) once the gradients of flat
 * range [lo, hi) are complete to add the shadow into them (and clear it).  shadow = NULL switches the mode off.  Not
 * stream-concurrent: call with no kernel of the library in flight, outside captures.  The cross-entropy statistics of ig_ce_loss
 * and the weight gradients of the linears (ordered split-K folds) are order-independent in both modes. */
int ig_set_deterministic(void* shadow, const void* grad_base, long n, void* stream);
int ig_get_deterministic(void);
int ig_det_fold(long lo, long hi, void* stream);
/* the same over n ranges in ONE launch: ranges_dev = DEVICE int64 [n][2] (lo, hi), longest = the longest range (sizes the grid) */
int ig_det_fold_ranges(int n, const long* ranges_dev, long longest, void* stream);

/* ---- dataset side: normalise + layout (instageo/model/dataloader.py:495-524, 707-750) ---------------- */
/* src (B, T*C, H, W) band = t*C+c, src_dtype 0=int16 1=float32 -> dst (B, C, T, H, W) f32 = (src*mult-mean_c)/std_c */
int ig_normalize_chips(const void* src, int src_dtype, const float* mean, const float* stdv, double mult, int mult_enabled,
                       float* dst, int B, int T, int C, int H, int W, void* stream);
/* On-device input pipeline.
 * RandomCrop(im) + hflip/vflip + per-band normalise in one pass (dataloader.py:58-141, 495-585): src (B, T*C, Hs, Ws)
 * int16|f32, params[b] = {top, left, hflip, vflip} (host-drawn, so the reference's RNG stream can be replayed);
 * labels (optional, f32 (B, Hs, Ws) -> (B, im, im)) get the same crop and flips. */
int ig_crop_flip_normalize(const void* src, int src_dtype, const float* mean, const float* stdv, double mult, int mult_enabled,
                           const int* params, float* dst, const float* labels_in, float* labels_out, int B, int T, int C, int Hs,
                           int Ws, int im, void* stream);
/* Sliding-window gather + normalise (process_test / crop_array, dataloader.py:588-669; chip_inference over a tile,
 * BASELINE configs[3]): tile (T*C, Hs, Ws) int16|f32 and origins[i] = {top, left} -> dst (n, C, T, crop, crop) f32 normalised,
 * optionally the same windows of a label tile (Hs, Ws) f32 -> (n, crop, crop).  One launch for all n windows. */
int ig_normalize_windows(const void* tile, int src_dtype, const float* mean, const float* stdv, double mult, int mult_enabled,
                         const int* origins, float* dst, const float* labels_tile, float* labels_out, int n, int T, int C, int Hs,
                         int Ws, int crop, void* stream);
/* Photometric / resampling augmentations of the training pipeline (dataloader.py:144-386) on a raw-domain float32 batch
 * (B, T*C, S, S) -- the output of ig_crop_flip_normalize with identity statistics -- one launch per augmentation and batch.
 * Every random decision is drawn on the host, per chip, like the crop origins.
 * ig_aug_rotate: RandomRotation (dataloader.py:144-187) = Pillow's nearest-neighbour Image.rotate with constant fill;
 *   params[b] = {apply, a0, a1, a2, a3, a4, a5, 0}: the 16.16 fixed-point inverse affine of the drawn angle (host mirror:
 *   dataloader.rotate_coeffs); labels (B, S, S) follow with their own fill.  src != dst.
 * ig_aug_brightness_contrast: RandomBrightnessContrast (dataloader.py:190-260), in place; params[b] = {apply, bright, contrast, 0}.
 * ig_aug_blur: RandomGaussianBlur (dataloader.py:263-333): clip/scale to [0,1], ksize x ksize kernel2d (outer product of the
 *   two normalised 1-D Gaussians) with reflect padding, clamp, * max_pixel, truncate to uint16; apply[b] in {0,1}.  src != dst.
 * ig_aug_noise: RandomGaussianNoise (dataloader.py:336-386), in place; params[b] = {apply, seed}; noise = optional standard-normal
 *   field of buf's shape (else a counter hash + Box-Muller seeded per chip). */
int ig_aug_rotate(const float* src, float* dst, const float* labels_in, float* labels_out, const int* params, float fill,
                  float label_fill, int B, int CT, int S, void* stream);
int ig_aug_brightness_contrast(float* buf, const float* params, float max_pixel, int B, int CT, int S, void* stream);
int ig_aug_blur(const float* src, float* dst, const int* apply, const float* kernel2d, int ksize, float max_pixel, int B, int CT,
                int S, void* stream);
int ig_aug_noise(float* buf, const int* params, const float* noise, float noise_std, float max_pixel, int B, int CT, int S,
                 void* stream);
/* mode=stats reduction (pipeline_utils.py:207-254): sums[c] += mean_bc, sums[C+c] += biased var_bc over (T,H,W) for every
 * chip b of x (B, C, T, H, W) f32; counts[v - lo] += 1 per label value (counts[nbins] = everything else) */
int ig_chip_stats(const float* x, double* sums, int B, int C, long n_per_channel, void* stream);
int ig_label_hist(const float* labels, unsigned long long* counts, long n, int lo, int nbins, void* stream);

/* ---- encoder (instageo/model/pritvhi.py) ------------------------------------------------------------- */
/* Conv3d(k=s=(1,p,p)) im2col: img (B,C,T,H,W) f32 -> patches [B*T*gh*gw][C*p*p], token order (t,row,col)  :243-268 */
int ig_patchify(const float* img, void* out_hi, void* out_lo, int B, int C, int T, int H, int W, int p, void* stream);
/* x[b][0][:] = cls_token + pos_embed[0]                                                        :520-522 */
int ig_cls_rows(float* x, const float* cls, const float* pos, int B, int ntok, int D, void* stream);
/* x[b][1+tp][:] = patches[b*TP+tp] @ w^T + bias + pos_embed[1+tp]                               :266-268,513-517 */
int ig_patch_embed_fwd(const void* p_hi, const void* p_lo, const void* w_hi, const void* w_lo, const float* bias,
                       const float* pos, float* x, int batch, int tokens_per_chip, int D, int K, void* stream);
/* nn.LayerNorm(eps) over D; feat_T>=1 writes the model.py:406-413 feature-image layout [B][G][D*T] (c = d*T+t) */
int ig_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* out_hi, void* out_lo, float* mean,
                     float* rstd, int M, int D, float eps, int feat_T, int feat_G, int ntok, void* stream);
int ig_layernorm_bwd(const void* dy_hi, const void* dy_lo, const float* x, const float* mean, const float* rstd,
                     const float* gamma, float* dx, int accumulate, void* dxb_hi, void* dxb_lo, float* dgamma, float* dbeta,
                     float* dcol, int M, int D, int feat_T, int feat_G, int ntok, void* stream);
/* timm Block linears (qkv / fc1 [+GELU]) : y = act(x @ w^T + b), act 0 none, 1 exact GELU            :446-456
 * act 1 with dact != NULL (training) also stores dact = gelu'(x @ w^T + b), the factor ig_linear_dgrad mode 1 applies */
int ig_linear_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias, void* y_hi,
                  void* y_lo, void* dact_hi, void* dact_lo, int M, int N, int K, int act, void* stream);
/* timm Block residual linears (proj / fc2): out = resid + x @ w^T + b (fp32 residual stream)          :446-456 */
int ig_linear_residual_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                           const float* resid, float* out, int M, int N, int K, void* stream);
/* dx = dy @ w  (mode 1: * dact elementwise, the saved gelu'); optional dx_colsum[k] += sum_m dx[m][k] (bias grad of the producing layer);
 * dw += dy^T @ x (fp32 atomics) */
int ig_linear_dgrad(const void* dy_hi, const void* dy_lo, const void* w_hi, const void* w_lo, void* dx_hi, void* dx_lo,
                    const void* dact_hi, const void* dact_lo, float* dx_colsum, int M, int N, int K, int mode, void* stream);
int ig_linear_wgrad(const void* dy_hi, const void* dy_lo, const void* x_hi, const void* x_lo, float* dw, int M, int N, int K,
                    void* stream);
/* n weight gradients sharing the token count M in ONE launch (the four linears of a timm Block, pritvhi.py:446-456:
 * autograd's grad_weight of F.linear for qkv / proj / fc1 / fc2): dw[g] += dy[g]^T @ x[g].  All array arguments are HOST arrays
 * of n entries; dy_lo / x_lo may be NULL (plain bf16).  Bit-reproducible (ordered split-K fold).  overwrite != 0: dw[g] = dy[g]^T @
 * x[g] -- the first backward of a step then needs neither a zeroed dw nor reads its old contents (4 + 4 bytes per weight less). */
int ig_linear_wgrad_group(int n, const void* const* dy_hi, const void* const* dy_lo, const void* const* x_hi,
                          const void* const* x_lo, float* const* dw, const int* N, const int* K, int M, int overwrite, void* stream);
/* base[lo .. hi) = 0 for n flat ranges in ONE launch: ranges_dev = DEVICE int64 [n][2], longest = the longest range */
int ig_zero_ranges(float* base, int n, const long* ranges_dev, long longest, void* stream);
/* ig_linear_dgrad with the weight handed over TRANSPOSED (wt [K][N] = w^T, see ig_transpose_bf16): same result, but both
 * operands are contiguous in the reduce dimension, the form of the forward linears (dx = dy @ w is autograd's grad_input of
 * F.linear, pritvhi.py:446-456) */
int ig_linear_dgrad_wt(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, void* dx_hi, void* dx_lo,
                       const void* dact_hi, const void* dact_lo, float* dx_colsum, int M, int N, int K, int mode, void* stream);
/* dst[b][c][r] = src[b][r][c] for b < batch: bf16 matrix transposes (R, C multiples of 64; strides in elements).  The engine
 * keeps a transposed operand copy of the Block linears' weights next to the bf16 shadow and refreshes it once per step. */
int ig_transpose_bf16(const void* src_hi, const void* src_lo, void* dst_hi, void* dst_lo, int R, int C, int batch, long src_stride,
                      long dst_stride, void* stream);
/* F.scaled_dot_product_attention of timm Attention: qkv [B][N][3][H][hd] -> out [B][N][H*hd], lse [B][H][N]; head_dim 64 or 80.
 * ig_attention_bwd: dqkv_colsum (optional, fp32 [3*H*hd]) += column sums of dqkv over the B*N tokens = the bias gradient of the
 * fused qkv Linear (pritvhi.py:446-456 -> timm Attention.qkv); fused into the single-pass backward kernel where that runs. */
int ig_attention_fwd(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, float* lse, int B, int N, int H,
                     int head_dim, void* stream);
int ig_attention_bwd(const void* qkv_hi, const void* qkv_lo, const void* out_hi, const void* out_lo, const void* dout_hi,
                     const void* dout_lo, const float* lse, float* delta, void* dqkv_hi, void* dqkv_lo, float* dqkv_colsum, int B,
                     int N, int H, int head_dim, void* stream);
/* gradient plumbing: column sums (bias grads), patch-embed grad prep (cls_token / conv bias grads) */
int ig_colsum(const void* hi, const void* lo, float* out, long M, int C, void* stream);
int ig_patch_grad_prep(const float* dx, void* hi, void* lo, float* dcls, float* dbias, int B, int ntok, int D, void* stream);
int ig_split_bf16(const float* src, void* hi, void* lo, long n, void* stream);
int ig_merge_bf16(const void* hi, const void* lo, float* dst, long n, void* stream);

/* ---- decode head (instageo/model/model.py:349-390) --------------------------------------------------- */
/* nn.ConvTranspose2d(k=3,s=2,p=1,op=1) + nn.Dropout(p): x (B,H,W,Cin) -> y (B,2H,2W,Cout)             :361-369 */
int ig_convT_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias, void* y_hi,
                 void* y_lo, int B, int H, int W, int Cin, int Cout, unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p,
                 void* stream);
int ig_convT_dgrad(const void* dy_hi, const void* dy_lo, const void* w_hi, const void* w_lo, void* dx_hi, void* dx_lo, int B,
                   int H, int W, int Cin, int Cout, void* stream);
/* dbias (optional): dbias[co] += sum over the output pixels of dy (the ConvTranspose2d bias gradient); fused into the direct
 * kernel of the 96 -> 48 stage, one column-sum pass otherwise */
int ig_convT_wgrad(const void* dy_hi, const void* dy_lo, const void* x_hi, const void* x_lo, float* dw, float* dbias, int B, int H,
                   int W, int Cin, int Cout, void* stream);
/* nn.Conv2d(k=3,padding=1)                                                                           :370-375 */
/* bn_scale/bn_shift (may be NULL): eval-mode BatchNorm2d + ReLU folded into the epilogue, y = relu((conv+bias)*s + t) */
int ig_conv3x3_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                   const float* bn_scale, const float* bn_shift, void* y_hi, void* y_lo, int B, int H, int W, int Cin, int Cout,
                   void* stream);
/* the same in front of a training-mode nn.BatchNorm2d (:376): where a direct kernel runs (48 / 96 channels), the convolution also leaves
 * sums[2 Cout] (per-channel sum / sum of squares of the stored outputs) and sets *fused = 1 (HOST int); else *fused = 0 */
int ig_conv3x3_fwd_stats(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias, void* y_hi,
                         void* y_lo, double* sums, int* fused, int B, int H, int W, int Cin, int Cout, void* stream);
/* inference tail: the last nn.Conv2d(k=3, padding=1) (:370-375, + eval-mode BatchNorm + ReLU) and the nn.Conv2d(k=1) classifier (:389;
 * dropout is the identity in eval mode) in ONE kernel where the direct 48-channel kernel runs and ncls <= 2: *fused = 1 (HOST int), y_hi
 * may be NULL (the activation is then not stored).  *fused = 0: nothing was computed, run ig_conv3x3_fwd + ig_classifier_fwd. */
int ig_conv3x3_cls_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias, const float* bn_scale,
                       const float* bn_shift, void* y_hi, const float* cls_w, const float* cls_b, float* logits, int* fused, int B, int H,
                       int W, int Cin, int Cout, int ncls, void* stream);
/* batch statistics -> scale / shift / mean / rstd (+ running update) from sums a producer filled (finalize step of ig_bn_relu_fwd) */
int ig_bn_finalize(const double* sums, const float* gamma, const float* beta, float* running_mean, float* running_var, float* scale,
                   float* shift, float* mean, float* rstd, long M, int C, float eps, float momentum, int update_running, void* stream);
int ig_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float* scale,
                      float* shift, int C, float eps, void* stream);
int ig_conv3x3_dgrad(const void* dy_hi, const void* dy_lo, const void* w_hi, const void* w_lo, void* dx_hi, void* dx_lo, int B,
                     int H, int W, int Cin, int Cout, unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p, void* stream);
/* dbias (optional): dbias[co] += sum_pixels dy[p][co] (the Conv2d bias gradient); fused into the direct kernels (48 / 96 / 192
 * input channels: an all-ones MFMA operand against the dy fragments), one column-sum pass otherwise */
int ig_conv3x3_wgrad(const void* dy_hi, const void* dy_lo, const void* x_hi, const void* x_lo, float* dw, float* dbias, int B, int H,
                     int W, int Cin, int Cout, void* stream);
/* nn.Conv2d(kernel_size=KS, padding=1) for odd KS in 3..9: the 5 x 5 / 7 x 7 convolutions of the 600M variants' decode head
 * (model.py:169-177 seg_head_kernel_sizes, :370-375).  NHWC, weights Wc[Cout][KS*KS][Cin]; x is (B,H,W,Cin), y / dy are
 * (B,Ho,Wo,Cout) with Ho = H + 3 - KS, Wo = W + 3 - KS.  bn_scale / bn_shift as in ig_conv3x3_fwd. */
int ig_convk_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias, const float* bn_scale,
                 const float* bn_shift, void* y_hi, void* y_lo, int B, int H, int W, int Cin, int Cout, int KS, void* stream);
int ig_convk_dgrad(const void* dy_hi, const void* dy_lo, const void* w_hi, const void* w_lo, void* dx_hi, void* dx_lo, int B,
                   int H, int W, int Cin, int Cout, int KS, unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p,
                   void* stream);
int ig_convk_wgrad(const void* dy_hi, const void* dy_lo, const void* x_hi, const void* x_lo, float* dw, float* dbias, int B, int H,
                   int W, int Cin, int Cout, int KS, void* stream);
/* nn.BatchNorm2d + nn.ReLU on [M][C] (M = B*H*W)                                                      :376-377 */
int ig_bn_relu_fwd(const void* x_hi, const void* x_lo, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, void* y_hi, void* y_lo, float* scale, float* shift, float* mean, float* rstd,
                   double* sums, long M, int C, float eps, float momentum, int training, int update_running, void* stream);
int ig_bn_relu_bwd(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, const float* scale,
                   const float* shift, const float* mean, const float* rstd, void* dx_hi, void* dx_lo, float* dgamma,
                   float* dbeta, double* sums, long M, int C, void* stream);
/* y_hi == NULL: batch statistics + scale / shift / mean / rstd (+ running update) only, no apply pass (ig_classifier_bn_fwd applies);
 * training == 2: batch statistics that a producer has already left in sums (ig_conv3x3_fwd_stats with *fused == 1): no statistics pass */
/* nn.Dropout(p) + nn.Conv2d(k=1): f (B,HW,C) -> logits (B,ncls,HW) f32                                 :388-389 */
int ig_classifier_fwd(const void* f_hi, const void* f_lo, const float* w, const float* bias, float* logits, int B, long HW, int C,
                      int ncls, unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p, void* stream);
int ig_classifier_bwd(const float* dlogits, const void* f_hi, const void* f_lo, const float* w, void* df_hi, void* df_lo,
                      float* dw, float* db, const double* count, int B, long HW, int C, int ncls, unsigned drop_seed,
                      const unsigned* drop_seed_dev, float drop_p, void* stream);
/* training-mode tail of the head in fused passes: the last stage's nn.BatchNorm2d + nn.ReLU (:376-377) applied inside the
 * nn.Dropout + nn.Conv2d(k=1) kernels (:388-389); x = that stage's Conv2d output.  The activation between them and its gradient
 * are recomputed, never stored.  Backward: classifier dW / db + the BatchNorm sums in one pass over x, dx + dgamma / dbeta in a second. */
int ig_classifier_bn_fwd(const void* x_hi, const void* x_lo, const float* scale, const float* shift, const float* w, const float* bias,
                         float* logits, int B, long HW, int C, int ncls, unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p,
                         void* stream);
int ig_classifier_bn_bwd(const float* dlogits, const void* x_hi, const void* x_lo, const float* scale, const float* shift,
                         const float* mean, const float* rstd, const float* w, void* dx_hi, void* dx_lo, float* dw, float* db,
                         float* dgamma, float* dbeta, double* sums, const double* count, int B, long HW, int C, int ncls,
                         unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p, void* stream);

/* ---- task module (instageo/model/segmentation.py, metrics.py, infer_utils.py, base.py) --------------- */
/* CE(weight, ignore_index,'none') + masked mean pieces, argmax, int64 confusion matrix   segmentation.py:85-87,117-151 */
int ig_ce_loss(const float* logits, const void* labels, int label_dtype, const float* class_weights, long ignore_index,
               double* stats, float* dlogits, long long* preds, signed char* preds_i8, unsigned long long* confusion, int B,
               long HW, int ncls, void* stream);
/* torch.argmax(dim=1) -> int8                                                           infer_utils.py:99-101 */
int ig_argmax_i8(const float* logits, signed char* out, int B, long HW, int ncls, void* stream);
/* RunningConfusionMatrix.update                                                         metrics.py:86-108 */
/* knowledge distillation (segmentation.py:352-378): KLDivLoss(batchmean)(log_softmax(student),
 * softmax(teacher)) over the valid pixels: *kl_sum += KL sum, dlogits += softmax(student) - softmax(teacher) (un-normalised,
 * on top of the cross-entropy gradient written by ig_ce_loss) */
int ig_kd_loss(const float* student_logits, const float* teacher_logits, const void* labels, int label_dtype, long ignore_index,
               double* kl_sum, float* dlogits, int B, long HW, int ncls, void* stream);
/* regression head (regression.py:141-191, metrics.py:330-352): masked MSE (+ log1p label scale) of the
 * single-channel output: stats double[2] += (sum sq. err, #valid), dpred = 2 (pred - label') un-normalised, msums double[9] =
 * streaming sums of RunningRegressionMetrics on the de-scaled values {n, Sx, Sy, Sxy, Sxx, Syy, S|e|, See, #within EE} */
int ig_mse_loss(const float* pred, const float* labels, float ignore_value, int use_log_scale, double* stats, float* dpred,
                double* msums, float ee_bias, float ee_coef, int include_ee, long n, void* stream);
/* distillation of the regression task (regression.py:345-534): sum[0] += sum over valid pixels (labels != ignore_value) of
 * (pred - teacher')^2, teacher' = log1p(teacher) under use_log_scale (regression.py:527-529); dpred += 2 (pred - teacher') on
 * top of the gradient ig_mse_loss wrote (both terms are means over the same valid count, regression.py:496-503) */
int ig_kd_mse_loss(const float* pred, const float* teacher, const float* labels, float ignore_value, int use_log_scale, double* sum,
                   float* dpred, long n, void* stream);
/* test-time metrics on the device: RunningAUC histograms of softmax(logits) (metrics.py:214-256 via
 * segmentation.py:153-156; hist = uint64 [2][ncls][nbins], 0 = positives / 1 = negatives of each class, ignored pixels
 * skipped) and predict_step's softmax(logits, 1)[:, cls] (segmentation.py:202-213) */
int ig_auc_update(const float* logits, const void* labels, int label_dtype, long ignore_index, unsigned long long* hist, int B,
                  long HW, int ncls, int nbins, float min_score, float max_score, void* stream);
int ig_softmax_prob(const float* logits, float* out, int B, long HW, int ncls, int cls, void* stream);
int ig_confusion_update(const long long* y_true, const long long* y_pred, unsigned long long* confusion, long n, int k,
                        long ignore_index, int has_ignore, void* stream);
/* torch.optim.AdamW step on a flat buffer (+ clip_weights, + bf16 shadow refresh)        base.py:103-126 */
int ig_adamw_advance(float* hyper, void* stream);
int ig_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_hi, void* shadow_lo, const float* hyper, long n,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INSTAGEO_HIP_H */
