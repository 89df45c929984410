/* dedark_yolo.h -- C-ABI of the MI355X-native Dedark-YOLO hot path (libdedark_yolo.so).
 *
 * Drop-in boundary: the reference is 100 % Python on PyTorch; every entry point below replaces the
 * ATen op sequence issued by one reference function (cited as U/<file>:<lines>, U = ultralytics/ in the reference
 * tree).  Signatures are plain pointers and sizes -- no torch types.  All pointers are DEVICE pointers unless
 * named host_*.  Every function enqueues on `stream` (a hipStream_t passed as void*), never synchronises, never
 * allocates, and returns 0 on success; on failure it returns non-zero and dy_last_error() describes it.
 *
 * Layouts: activations are NHWC ("pixel-major"); a tensor view is (pointer to element [n=0,h=0,w=0,c=c0], ld) where
 * `ld` is the pixel stride in ELEMENTS, so a channel slice of a wider concat buffer is a view with ld = total width.
 * dtype: DY_F32 (parity path, exact-f32 MFMA 32x32x2), DY_BF16 (throughput path, MFMA 32x32x16 / 16x16x32, f32 accumulate) or
 * DY_F16 (IEEE half: the same kernels on the f16 MFMA; the reference's AMP dtype, BASELINE configs[4]).
 * Weights: f32 OIHW master (the reference state_dict layout) is packed per step into [Cout][KH][KW][Cin] ("KRSC").
 */
#ifndef DEDARK_YOLO_H
#define DEDARK_YOLO_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DY_F32 0
#define DY_BF16 1
#define DY_F16 2
#define DY_ACT_NONE 0
#define DY_ACT_SILU 1
#define DY_ACT_LEAKY 2 /* LeakyReLU(0.1) */
#define DY_STATS_REPLICAS 64 /* copies of the BN batch-statistics accumulators filled by dy_conv2d_fwd */
#define DY_BN_BWD_REPLICAS 8 /* copies of the (sum g, sum g*zhat) accumulators of dy_bn_act_bwd_reduce */

const char* dy_last_error(void);
int dy_version(void);
/* Profiling aid: symbol of the GPU kernel launched by the last call on this thread ("" if the entry reports none); bench.py
 * attaches its per-kernel roofline to it.  dy_clear_last_kernel resets it. */
const char* dy_last_kernel(void);
void dy_clear_last_kernel(void);

/* ------------------------------------------------------------------------------------------------ convolution
 * Replaces nn.Conv2d inside Conv (U/nn/modules/conv.py:38-55), add_conv (U/nn/modules/block.py:24-45),
 * ConvBlock (U/nn/modules/common.py:9-23), Detect's 1x1 heads (U/nn/modules/head.py:40-46), RFBblock
 * (block.py:703-734) and nn.Linear of the extractor (common.py:65-66).  Implicit GEMM on MFMA. */
typedef struct {
  const void* src;   /* source activations view */
  int64_t src_ld;
  int N, Hs, Ws, Cs; /* source geometry; Cs = channels consumed per tap (multiple of 4 for f32, 8 for bf16) */
  const void* w;     /* packed weights [Cd][KH][KW][Cs], compute dtype */
  void* dst;         /* destination view */
  int64_t dst_ld;
  int Hd, Wd, Cd;    /* destination geometry; Cd = channels produced */
  int KH, KW, stride, pad, dil;
  const float* scale; /* optional per-Cd affine applied to the accumulator: v = acc*scale + shift (NULL = 1) */
  const float* shift; /* optional (bias / folded BN shift) (NULL = 0) */
  int act;            /* DY_ACT_* applied after the affine */
  double* stats;      /* optional [DY_STATS_REPLICAS][2*Cd] zeroed accumulators: (sum, sum of squares) of the RAW conv
                         output over all pixels, spread over replicas to avoid atomic contention (BN batch stats) */
  int accumulate;     /* 1: dst += result */
  int dtype;          /* DY_F32 | DY_BF16 | DY_F16 (src, w, dst) */
  /* optional extensions, all zero = dense destination / full window (a zero-initialised descriptor keeps the old meaning) */
  int64_t dst_row_stride; /* elements between destination rows    (0: Wd * dst_ld) */
  int64_t dst_img_stride; /* elements between destination images  (0: Hd * row stride) */
  int KHf, KWf;           /* tap subset: `w` is a [Cd][KHf][KWf][Cs] pack and window tap (th, tw) of the KH x KW window uses */
  int kh0, kh_step;       /* weight tap (kh0 + kh_step*th, kw0 + kw_step*tw).  KHf == 0: no subset.  dy_conv2d_dgrad uses   */
  int kw0, kw_step;       /* this internally to run a stride-2 data gradient as 4 dense stride-1 problems (one per parity)   */
  int dst_valid_channels; /* hint: only the first k destination channels can be non-zero (the rest is channel padding whose
                             weights are zero); 0 = unknown.  Lets the stem kernels skip the padding. */
  void* dst_planar;       /* dy_conv2d_dgrad only, optional: write dx as PLANAR [N, dst_valid_channels, Hd, Wd] (compute dtype)
                             instead of the NHWC view `dst` (which may then be NULL).  Only the direct stem kernel (3x3,
                             stride 2, pad 1, <= 8 padded input channels, bf16) supports it; otherwise the call fails. */
  const void* add_src;    /* dy_conv2d_dgrad only, optional: a [N,Hd,Wd,Cd] view (pixel stride add_src_ld, compute dtype) added to */
  int64_t add_src_ld;     /* the result: dst = [dst +] dx + add_src.  Bottleneck's shortcut gradient (U/nn/modules/block.py:565:
                             x + cv2(cv1(x))) joins the data gradient of cv1 this way instead of through a separate pass; the
                             large-tile kernels add it in their epilogue, other routes run dy_copy2d(accumulate) afterwards. */
} dy_conv_desc;

/* forward: dst[n,ho,wo,:] = epilogue( sum_{kh,kw,c} src[n, ho*stride-pad+kh*dil, wo*stride-pad+kw*dil, c] * w[:,kh,kw,c] ) */
int dy_conv2d_fwd(const dy_conv_desc* d, void* stream);
/* data gradient: src = dz [N,Hs,Ws,Cs=Cout], w = transposed pack [Cd=Cin][KH][KW][Cs=Cout], dst = dx [N,Hd,Wd,Cd];
 * dx[n,h,w,:] = sum over taps with (h+pad-kh*dil) % stride == 0 of dz[n,(h+pad-kh*dil)/stride, ...,:] * w */
int dy_conv2d_dgrad(const dy_conv_desc* d, void* stream);
/* weight gradient: g_oihw[Cout][Cin][KH][KW] (f32, overwritten) = sum_pixels dz^T * gather(x).
 * x view [N,Hi,Wi,Cin_pad], dz view [N,Ho,Wo,Cout_pad] (zero-padded channels).  The pixel reduction is split over
 * thread blocks that store partial tiles into `scratch` (any f32 workspace of scratch_elems floats, >= one tile set;
 * more allows more splits); a second kernel sums the slabs in a fixed order (deterministic, no atomics) and writes OIHW. */
int dy_conv2d_wgrad(const void* x, int64_t x_ld, int N, int Hi, int Wi, int Cin_pad, const void* dz, int64_t dz_ld, int Ho,
                    int Wo, int Cout_pad, int KH, int KW, int stride, int pad, int dil, int Cout, int Cin, float* scratch,
                    int64_t scratch_elems, float* g_oihw, int dtype, void* stream);
/* f32 OIHW [Cout][Cin][KH][KW] -> packed [Cout_pad][KH][KW][Cin_pad] (transposed=0) or [Cin_pad][KH][KW][Cout_pad]
 * (transposed=1) in `dtype`; padded input/output channels are written as zero. */
int dy_pack_weight(const float* w_oihw, void* packed, int Cout, int Cout_pad, int Cin, int Cin_pad, int KH, int KW,
                   int transposed, int dtype, void* stream);
/* The same for every weight of a model in ONE launch (after each optimizer step): `items_dev` is a DEVICE array sorted by
 * first_block; item i owns thread blocks [first_block, first_block + dy_pack_item_blocks(...)); n_blocks = their sum. */
typedef struct {
  const float* w;   /* f32 OIHW master weight */
  void* packed;     /* destination, `dtype` elements */
  int Cout, Cout_pad, Cin, Cin_pad, KH, KW, transposed, dtype;
  int64_t first_block;
} dy_pack_item;
int64_t dy_pack_item_blocks(int Cout_pad, int Cin_pad, int KH, int KW);
int dy_pack_weights_multi(const dy_pack_item* items_dev, int n_items, int64_t n_blocks, void* stream);
/* packed f32 grad [Cout][KH][KW][Cin_pad] -> OIHW f32 (overwrite) */
int dy_unpack_wgrad(const float* dw_packed, float* g_oihw, int Cout, int Cin, int Cin_pad, int KH, int KW, void* stream);

/* ------------------------------------------------------------------------------------ BatchNorm + activation
 * Replaces nn.BatchNorm2d (train: batch statistics, eps 1e-3, momentum 0.03, U/utils/torch_utils.py:263-265) followed by
 * SiLU / LeakyReLU(0.1) (conv.py:40,51; block.py:42) and the Bottleneck residual add (block.py:565). */
/* stats[DY_STATS_REPLICAS][2C] (from dy_conv2d_fwd) -> scale/shift (y = z*scale + shift), saved mean/invstd, running buffers
 * update. */
int dy_bn_finalize(const double* stats, int64_t count, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float momentum, float eps, float* scale, float* shift, float* mean, float* invstd,
                   int C, void* stream);
/* eval-mode fold: scale = gamma/sqrt(var+eps), shift = beta - mean*scale */
int dy_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                    float* scale, float* shift, int C, void* stream);
/* y = act(z*scale + shift) (+ residual); views over `pixels` pixels x C channels */
int dy_bn_act_fwd(const void* z, int64_t z_ld, const float* scale, const float* shift, int act, const void* residual,
                  int64_t res_ld, void* y, int64_t y_ld, int64_t pixels, int C, int dtype, void* stream);
/* backward pass 1: sums[0:C] = sum g, sums[C:2C] = sum g*zhat with g = dy*act'(u), u = z*scale+shift,
 * zhat = (z-mean)*invstd (has_bn) ; without BN only sum g (bias gradient).
 * `sums` is [DY_BN_BWD_REPLICAS][2C] doubles, zeroed by the caller; thread blocks spread their atomics over the replicas
 * and dy_bn_act_bwd_apply adds them up. */
int dy_bn_act_bwd_reduce(const void* dy, int64_t dy_ld, const void* z, int64_t z_ld, const float* scale,
                         const float* shift, const float* mean, const float* invstd, int act, int has_bn, double* sums,
                         int64_t pixels, int C, int dtype, void* stream);
/* backward pass 2: dz = gamma*invstd*(g - sum_g/M - zhat*sum_gz/M)  (has_bn) or dz = g; also dgamma = sum_gz,
 * dbeta = sum_g (written when the pointers are non-NULL, by block 0). */
int dy_bn_act_bwd_apply(const void* dy, int64_t dy_ld, const void* z, int64_t z_ld, const float* scale, const float* shift,
                        const float* mean, const float* invstd, const float* gamma, int act, int has_bn,
                        const double* sums, void* dz, int64_t dz_ld, float* dgamma, float* dbeta, int64_t pixels, int C,
                        int dtype, void* stream);

/* ----------------------------------------------------------------------------- pooling / resampling / concat
 * SPPF's 3 chained MaxPool2d(5,1,2) (block.py:331-338), ASFF's MaxPool2d(2,2) and max_pool2d(3,2,1) (block.py:58,85-86),
 * nn.Upsample / F.interpolate nearest (yolov8.yaml head; block.py:91,97,99), torch.cat (conv.py:473), x + y. */
/* argmax (optional) stores the window offset kh*k+kw of the FIRST maximum in scan order, one byte per output element */
int dy_maxpool_fwd(const void* x, int64_t x_ld, void* y, int64_t y_ld, uint8_t* argmax, int N, int H, int W, int C, int k,
                   int stride, int pad, int Ho, int Wo, int dtype, void* stream);
/* dx (+)= adjoint of the pooling through argmax (gather form, no atomics) */
int dy_maxpool_bwd(const void* dy, int64_t dy_ld, const uint8_t* argmax, void* dx, int64_t dx_ld, int N, int H, int W, int C,
                   int k, int stride, int pad, int Ho, int Wo, int accumulate, int dtype, void* stream);
int dy_upsample_nearest_fwd(const void* x, int64_t x_ld, void* y, int64_t y_ld, int N, int H, int W, int C, int scale,
                            int dtype, void* stream);
/* dx (+)= sum over the scale x scale children of dy */
int dy_upsample_nearest_bwd(const void* dy, int64_t dy_ld, void* dx, int64_t dx_ld, int N, int H, int W, int C, int scale,
                            int accumulate, int dtype, void* stream);
/* strided copy / add of `pixels` x C channel slabs (concat writes, chunk reads, gradient accumulation) */
int dy_copy2d(const void* src, int64_t src_ld, void* dst, int64_t dst_ld, int64_t pixels, int C, int accumulate, int dtype,
              void* stream);
int dy_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
/* ASFF blend (block.py:103-111): w = softmax(logits[.,3]); out = sum_i w_i * x_i */
int dy_asff_fuse_fwd(const void* x0, int64_t ld0, const void* x1, int64_t ld1, const void* x2, int64_t ld2,
                     const void* logits, int64_t ldl, void* out, int64_t ldo, int64_t pixels, int C, int dtype, void* stream);
int dy_asff_fuse_bwd(const void* dout, int64_t lddo, const void* x0, int64_t ld0, const void* x1, int64_t ld1,
                     const void* x2, int64_t ld2, const void* logits, int64_t ldl, void* dx0, int64_t ldd0, void* dx1,
                     int64_t ldd1, void* dx2, int64_t ldd2, void* dlogits, int64_t lddl, int64_t pixels, int C,
                     int acc0, int acc1, int acc2, int dtype, void* stream);

/* ------------------------------------------------------------------------------------- low-light front-end
 * lowlight_recovery.forward (U/nn/modules/llie.py:17-54) and the five filters of U/nn/modules/filtersB.py. */
/* NCHW f32 [B,3,H,W] -> NHWC with 8 channels (3 used, 5 zero) in `dtype`; optional bilinear resize to (Ho,Wo)
 * with align_corners=False (llie.py:43). Ho==H && Wo==W is a plain relayout. */
int dy_image_to_nhwc8(const float* x, int B, int H, int W, void* y, int Ho, int Wo, int dtype, void* stream);
/* adjoint of the bilinear resize: dx[B,3,H,W] f32 += resize^T(dy NHWC8 f32) */
int dy_resize_bwd(const float* dy_nhwc, int dy_ld, int B, int H, int W, int Ho, int Wo, float* dx, void* stream);
/* feat[B, feat_ld >= 15] -> params[B,8] = (omega, s_r, s_g, s_b, gamma, alpha, lambda, 0)  (filtersB.py regressors);
 * the backward writes all feat_ld columns of dfeat (zeros beyond the used slots) */
int dy_filter_params_fwd(const float* feat, int feat_ld, float* params, int B, void* stream);
/* dparams: f64 [B,8], zeroed by the caller and accumulated by dy_usm_bwd / dy_filters_pointwise_bwd with f64 atomics (a sum of f32
 * block partials that does not depend on their arrival order: the regressor's gradients repeat from run to run) */
int dy_filter_params_bwd(const float* feat, int feat_ld, const double* dparams, float* dfeat, int B, void* stream);
/* DeDark -> WB -> Gamma -> Contrast pointwise chain (filtersB.py:190-303) x -> s4 ; A [B,3] or NULL (0.8);
 * IcA [B,H,W] or NULL (0.5).  fast_math = 0: libm powf / division (f32 parity mode, comparable with torch.pow to ~1 ulp);
 * fast_math = 1: v_log_f32 / v_exp_f32 / v_rcp_f32 (~2e-6 relative; the throughput mode, 10x faster backward) */
int dy_filters_pointwise_fwd(const float* x, const float* params, const float* A, const float* IcA, float* s4, int B,
                             int H, int W, int fast_math, void* stream);
/* USM (filtersB.py:153-175) as a separable 25-tap gaussian with reflect halo: out = (s4 - blur)*lambda + s4.
 * Writes out NCHW f32 (optional), the NHWC8 copy in `dtype` for the stem conv (optional), and hp = s4 - blur (optional). */
int dy_usm_fwd(const float* s4, const float* params, float* out_nchw, void* out_nhwc8, float* hp, int B, int H, int W,
               int dtype, void* stream);
/* USM backward: ds4 = dout*(1+lambda) - lambda*blur^T(dout); dparams[b,6] += sum dout*hp.  dout is either NCHW f32
 * (dout_nchw) or, in `dtype`, a padded NHWC view with pixel stride dout_ld >= 3 / a planar [B,3,H,W] tensor when
 * dout_ld == 0 (dout_nhwc); exactly one of the two pointers is non-NULL. */
int dy_usm_bwd(const float* dout_nchw, const void* dout_nhwc, int dout_ld, const float* hp, const float* params, float* ds4,
               double* dparams, int B, int H, int W, int dtype, void* stream);
/* pointwise backward: recomputes the chain from x, consumes ds4, writes dx (overwrite or +=; NULL = not needed) and
 * accumulates dparams[b, 0..5] */
int dy_filters_pointwise_bwd(const float* x, const float* params, const float* A, const float* IcA, const float* ds4,
                             float* dx, double* dparams, int B, int H, int W, int accumulate, int fast_math, void* stream);

/* ------------------------------------------------------------------------------------------ detection loss
 * v8DetectionLoss / RcoveryDetectionLoss (U/utils/loss.py:103-193,388-416), TaskAlignedAssigner (U/utils/tal.py),
 * bbox_iou CIoU (U/utils/metrics.py:75-128), make_anchors / dist2bbox (tal.py:246-271). */
typedef struct {
  const void* map[3]; /* Detect train outputs, NHWC views [B, h_l*w_l, no], no = 64+nc */
  int64_t map_ld[3];
  int h[3], w[3];
  float stride[3];
  int B, nc, n_levels, dtype;
} dy_det_maps;

/* group targets per image: rows (batch_idx, cls, cx, cy, w, h normalised) -> gt[B][n_max][5] = (cls, x1,y1,x2,y2 px),
 * counts[B]; order within an image is preserved (loss.py:124-139). */
int dy_loss_prepare_targets(const float* batch_idx, const float* cls, const float* bboxes, int n_targets, int B,
                            int n_max, float img_w, float img_h, float* gt, int32_t* counts, void* stream);
/* decode (loss.py:141-146): pred_boxes[B,A,4] xyxy in grid units, via softmax over 16 bins . arange */
int dy_loss_decode(const dy_det_maps* m, float* pred_boxes, void* stream);
/* task-aligned assignment (tal.py:83-127; topk 10, alpha .5, beta 6).  Scratch: work_f [2*R + 2*B*n_max] floats,
 * work_i [R] int32, work_b [R] bytes with R = B*n_max*A.  Outputs: target_gt_idx[B,A] i32, fg_mask[B,A] u8,
 * norm[B,A] f32 (= target_scores at the assigned label, 0 elsewhere), target_label[B,A] i32, target_box[B,A,4] f32 (px).
 * top-10 ties follow std::partial_sort as torch.topk does on CPU for A >= 640. */
int dy_tal_assign(const dy_det_maps* m, const float* pred_boxes, const float* gt, const int32_t* counts, int n_max,
                  float* work_f, int32_t* work_i, uint8_t* work_b, int32_t* target_gt_idx, uint8_t* fg_mask, float* norm,
                  int32_t* target_label, float* target_box, void* stream);
/* The same assigner on TaskAlignedAssigner.forward's own arguments (U/utils/tal.py:84-132): class probabilities pd_scores
 * [B,A,nc] f32, decoded boxes pd_bboxes [B,A,4] f32 in pixels, anchor points [A,2] in pixels; gt rows = (label, x1,y1,x2,y2) with
 * masked-out rows zeroed, counts[b] = rows of image b to consider.  Outputs as dy_tal_assign. */
int dy_tal_assign_decoded(const float* pd_scores, const float* pd_bboxes, const float* anc_points, const float* gt,
                          const int32_t* counts, int B, int A, int nc, int n_max, float* work_f, int32_t* work_i, uint8_t* work_b,
                          int32_t* target_gt_idx, uint8_t* fg_mask, float* norm, int32_t* target_label, float* target_box,
                          void* stream);
/* ---- SCConv pieces of MFRU (reference ultralytics/nn/modules/conv.py:323-440, block.py:164-217); NHWC views, HW = pixels per image ----
 * dy_chan_moments: out[n][c][2] += (sum x, sum x^2) over the pixels of image n (out zeroed by the caller): the group statistics
 * of GroupBatchnorm2d (conv.py:337-343) and the AdaptiveAvgPool2d(1) of CRU (conv.py:403,415). */
int dy_chan_moments(const void* x, int64_t ld, int N, int64_t HW, int C, double* out, int dtype, void* stream);
/* SRU.forward (conv.py:360-378): group norm over `groups` channel groups per image (torch.std: unbiased; eps added to std),
 * gate sigmoid(gn * gamma / sum(gamma)) >= 0.5, cross reconstruction y[c] = m[c] gn[c] + (1 - m[c']) gn[c'], c' = c +- C/2.
 * moments = dy_chan_moments(x). */
int dy_sru_fwd(const void* x, int64_t x_ld, void* y, int64_t y_ld, int N, int64_t HW, int C, int groups, const double* moments,
               const float* gamma, const float* beta, float eps, int dtype, void* stream);
/* its backward: dx (written), red[n][c][2] += (sum dgn, sum dgn * xhat) (zeroed by the caller) from which the caller takes
 * d gamma[c] = sum_n red[n][c][1], d beta[c] = sum_n red[n][c][0] (the gate has no gradient). */
int dy_sru_bwd(const void* x, int64_t x_ld, const void* dy, int64_t dy_ld, void* dx, int64_t dx_ld, int N, int64_t HW, int C, int groups,
               const double* moments, const float* gamma, const float* beta, float eps, double* red, int dtype, void* stream);
/* CRU tail (conv.py:413-417): o [.., 2C] = cat(Y1, Y2); s = softmax over the 2C channels of mean_pixels(o) per image;
 * res[c] = o[c] s[c] + o[c + C] s[c + C].  moments = dy_chan_moments(o) with 2C channels. */
int dy_cru_fuse_fwd(const void* o, int64_t o_ld, void* res, int64_t r_ld, int N, int64_t HW, int C, const double* moments, int dtype,
                    void* stream);
/* its backward: dout [.., 2C] (written); ds[n][k][2] scratch (zeroed by the caller; slot 0 = sum_pixels dres[c(k)] o[k]). */
int dy_cru_fuse_bwd(const void* o, int64_t o_ld, const void* dres, int64_t d_ld, void* dout, int64_t do_ld, int N, int64_t HW, int C,
                    const double* moments, double* ds, int dtype, void* stream);
/* CIoU of n box pairs, xyxy f32 (reference ultralytics/utils/metrics.py:75-128 bbox_iou(b1, b2, xywh=False, CIoU=True): eps added to
 * h only, alpha constant in the backward); grad_b1 (nullable) [n,4] = d out[i] / d b1[i]. */
int dy_bbox_ciou(const float* b1, const float* b2, int64_t n, float* out, float* grad_b1, void* stream);
/* Every mode of the same function (metrics.py:75-128): kind 0 IoU, 1 GIoU, 2 DIoU, 3 CIoU; xywh != 0: (cx, cy, w, h) boxes, w / h used
 * as given (:95-99), else xyxy with eps added to h only (:100-104).  grad_b1 (nullable) [n,4] = d out[i] / d b1[i] in b1's own
 * coordinates (b2 is a target; CIoU's alpha constant as under the reference's no_grad). */
int dy_bbox_iou(const float* b1, const float* b2, int64_t n, int xywh, int kind, float eps, float* out, float* grad_b1, void* stream);
/* Distribution focal loss (reference ultralytics/utils/loss.py:75-84 BboxLoss._df_loss): pred_dist [n_boxes*4, 16] logits,
 * target [n_boxes, 4] in [0, 15); out [n_boxes] = mean over the 4 sides; grad (nullable) = d sum(out) / d pred_dist. */
int dy_dfl_loss(const float* pred_dist, const float* target, int64_t n_boxes, float* out, float* grad, void* stream);
/* loss sums: acc[0]=sum target_scores, acc[1]=BCE sum, acc[2]=sum (1-ciou)*w, acc[3]=sum dfl*w (acc zeroed first) */
int dy_loss_fwd(const dy_det_maps* m, const float* pred_boxes, const uint8_t* fg_mask, const float* norm,
                const int32_t* target_label, const float* target_box, double* acc, void* stream);
/* finish: loss_out[0] = (box*hb + cls*hc + dfl*hd)*B + lrl*rec ; items[3] = (box*hb, cls*hc + lrl*rec, dfl*hd) */
int dy_loss_finish(const double* acc, const float* recovery, float hyp_box, float hyp_cls, float hyp_dfl, float lrl,
                   int B, float* loss_out, float* items, void* stream);
/* gradient wrt the three maps (written in `dtype`, every element), scaled by *grad_out (device scalar) */
int dy_loss_bwd(const dy_det_maps* m, void* const dmap[3], const int64_t dmap_ld[3], const float* pred_boxes,
                const uint8_t* fg_mask, const float* norm, const int32_t* target_label, const float* target_box,
                const double* acc, const float* grad_out, float hyp_box, float hyp_cls, float hyp_dfl, void* stream);
/* Detect eval decode (head.py:66-93): y[B, 4+nc, A] f32 = cat(xywh*stride, sigmoid(cls)) */
int dy_detect_decode(const dy_det_maps* m, float* y, void* stream);
/* ------------------------------------------------------------------------------------------------ NMS
 * non_max_suppression (U/utils/ops.py:144-278; called from U/models/yolo/detect/val.py:62-70) for the whole batch.
 * pred [B, 4+nc, A] f32 = Detect's eval output (xywh px + class scores).  Three stages:
 *  dy_nms_candidates: keys[b, slot] = (~bits(score) << 32) | (anchor*nc + cls) for every candidate (score > conf_thres; every
 *                     (anchor, cls) pair when multi_label && nc > 1 (ops.py:244-246), else the anchor's best class (:248-249));
 *                     counts[b] = number of candidates (may exceed cap = slots per image; the excess is dropped).
 *  dy_nms_sort:       per-image ascending key sort == stable descending-score order of the reference's candidate list.
 *                     workspace == NULL: only *workspace_bytes is written (size query).
 *  dy_nms_greedy:     first min(count, max_nms) candidates (ops.py:255-256), boxes + cls*max_wh unless agnostic (:258-259),
 *                     greedy suppression IoU > iou_thres (torchvision.ops.nms semantics, :261), first max_det kept (:262).
 *                     out [B, max_det, 6] = (x1,y1,x2,y2,conf,cls); keep_idx [B, max_det] = anchor*nc + cls of each kept row;
 *                     out_counts[B].  boxes_ws: B*max_nms*4 floats (16-byte aligned), dead_ws: B*max_nms bytes.
 * The wall-clock break of ops.py:274-276 is deliberately not reproduced. */
int dy_nms_candidates(const float* pred, int B, int nc, int A, float conf_thres, int multi_label, uint64_t* keys, int* counts,
                      int64_t cap, void* stream);
int dy_nms_sort(const uint64_t* keys, uint64_t* keys_sorted, const int* counts, int B, int64_t cap, void* workspace,
                size_t* workspace_bytes, void* stream);
int dy_nms_greedy(const float* pred, const uint64_t* keys_sorted, const int* counts, int B, int nc, int A, int64_t cap,
                  double iou_thres, int max_nms, int max_det, float max_wh, int agnostic, float* boxes_ws, uint8_t* dead_ws,
                  float* out, int64_t* keep_idx, int* out_counts, void* stream);

/* preprocess_batch tensor part (U/models/yolo/detect/train.py:70-111): u8 NCHW -> f32 /255 (^gamma), mse accumulators */
int dy_preprocess_batch(const uint8_t* img, float* img_out, float* clean_out, float dark_param, int lowlight, int dedark,
                        double* mse_acc, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------ optimizer
 * BaseTrainer.optimizer_step (U/engine/trainer.py:459-467): clip_grad_norm_(10.0) + SGD(nesterov) / AdamW step +
 * ModelEMA.update (U/utils/torch_utils.py:360-371) on flat f32 parameter ranges. */
int dy_sumsq(const float* g, int64_t n, double* acc, void* stream);
/* One flat buffer holds every trainable parameter; group_id[i] in {0,1,2} selects (lr, weight_decay) of the reference's three
 * parameter groups (decayed weights / BN weights / biases; NULL = group 0).  g is multiplied by grad_scale and by the clip
 * coefficient min(1, max_norm / (sqrt(*sumsq) + 1e-6)) when sumsq != NULL; ema may be NULL. */
int dy_sgd_step(float* p, const float* g, float* mom_buf, float* ema, const uint8_t* group_id, float lr0, float lr1, float lr2,
                float wd0, float wd1, float wd2, float momentum, int nesterov, float ema_decay, const double* sumsq,
                float max_norm, float grad_scale, int64_t n, void* stream);
int dy_adamw_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* ema, const uint8_t* group_id, float lr0,
                  float lr1, float lr2, float wd0, float wd1, float wd2, float beta1, float beta2, float eps, int step,
                  float ema_decay, const double* sumsq, float max_norm, float grad_scale, int64_t n, void* stream);
/* fp16 training (reference AMP: torch.cuda.amp.GradScaler, U/engine/trainer.py:221,330,340,459-467).  The loss gradient is
 * multiplied by loss_scale[0] before the backward pass; the *_scaled steps divide it out again (clip on the TRUE norm), and leave
 * parameters and optimizer state untouched when *sumsq is inf / NaN (the EMA is still updated, as trainer.optimizer_step does).
 * dy_loss_scale_update then applies GradScaler.update to state = {scale, consecutive finite steps, overflowed steps in total} (three
 * floats): scale *= backoff after an overflow, scale *= growth after `interval` finite steps.  dy_adamw_step_scaled takes its bias
 * correction at step - state[2]: torch's Adam does not advance on the steps GradScaler skips.  loss_scale == NULL: exactly
 * dy_sgd_step / dy_adamw_step. */
int dy_sgd_step_scaled(float* p, const float* g, float* mom_buf, float* ema, const uint8_t* group_id, float lr0, float lr1, float lr2,
                       float wd0, float wd1, float wd2, float momentum, int nesterov, float ema_decay, const double* sumsq,
                       float max_norm, float grad_scale, const float* loss_scale, int64_t n, void* stream);
int dy_adamw_step_scaled(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* ema, const uint8_t* group_id, float lr0,
                         float lr1, float lr2, float wd0, float wd1, float wd2, float beta1, float beta2, float eps, int step,
                         float ema_decay, const double* sumsq, float max_norm, float grad_scale, const float* loss_scale, int64_t n,
                         void* stream);
int dy_loss_scale_update(float* state, const double* sumsq, float growth, float backoff, int interval, void* stream);
/* ema = decay*ema + (1-decay)*src (EMA of the BatchNorm running buffers) */
int dy_ema_lerp(float* ema, const float* src, float decay, int64_t n, void* stream);
/* acc += g: gradient accumulation over `accumulate` batches (nbs / batch, U/engine/trainer.py:248,340-342) */
int dy_grad_accumulate(float* acc, const float* g, int64_t n, void* stream);
/* Merged entries: exactly the launches of the separate calls, behind ONE foreign-function call (the training step of BASELINE
 * configs[1] is ~700 launches issued from Python and had become host-bound).
 *   dy_conv2d_bn_act_fwd  = dy_conv2d_fwd(d: dst = raw z, stats) + dy_bn_finalize + dy_bn_act_fwd(z -> y); aff = 4*Cd floats
 *                           [scale | shift | mean | invstd] (kept for the backward pass)
 *   dy_bn_act_bwd         = dy_bn_act_bwd_reduce + dy_bn_act_bwd_apply with the same aff buffer
 *   dy_conv2d_wgrad_forked = dy_stream_fork(wait_for, stream) + dy_conv2d_wgrad on `stream` */
int dy_conv2d_bn_act_fwd(const dy_conv_desc* d, int64_t count, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float momentum, float eps, float* aff, int act, const void* residual, int64_t res_ld,
                         void* y, int64_t y_ld, void* stream);
int dy_bn_act_bwd(const void* dy, int64_t dy_ld, const void* z, int64_t z_ld, const float* aff, const float* gamma, int act,
                  double* sums, void* dz, int64_t dz_ld, float* dgamma, float* dbeta, int64_t pixels, int C, int dtype, void* stream);
int dy_conv2d_wgrad_forked(void* wait_for, const void* x, int64_t x_ld, int N, int Hi, int Wi, int Cin_pad, const void* dz,
                           int64_t dz_ld, int Ho, int Wo, int Cout_pad, int KH, int KW, int stride, int pad, int dil, int Cout,
                           int Cin, float* scratch, int64_t scratch_elems, float* g_oihw, int dtype, void* stream);

/* Stream plumbing of the backward pass (no counterpart in the single-stream reference): `to` waits for everything issued so far
 * on `from` (hipEventRecord on an internal ring of timing-free events + hipStreamWaitEvent; no host synchronisation).  Used to
 * run the weight gradients of a conv on a second HIP stream next to its dgrad -> BatchNorm-backward chain. */
int dy_stream_fork(void* from, void* to);
int dy_frontend_init(void); /* uploads the gaussian taps (call once per process, outside graph capture) */

/* ---- device-side input pipeline ----------------------------------------------------------------------------------
 * The pixel work of the reference's dataloader workers (ultralytics/data/augment.py:118-603,745-751, data/base.py:142-169) and of the
 * trainer's dark-channel loop (models/yolo/detect/train.py:42-68).  Images are uint8 HWC BGR (cv2.imread layout) in device memory. */
typedef struct dy_aug_sample {
  const uint8_t* src[4];   /* the (up to) four images of a mosaic, at their load_image size; one image for the letterbox case */
  int32_t sh[4], sw[4];
  int64_t pitch[4];        /* bytes per row */
  int32_t rect[4][6];      /* x1a, y1a, x2a, y2a on the canvas, x1b, y1b in the image (Mosaic._mosaic4, augment.py:166-188) */
  int32_t n_src, canvas_h, canvas_w;
  int32_t hsv, flipud, fliplr;
  double minv[6];          /* inverse of RandomPerspective's 2x3 matrix as cv::warpAffine inverts it (output -> canvas) */
  uint8_t lut[3][256];     /* RandomHSV's hue / saturation / value tables (augment.py:493-497) */
} dy_aug_sample;
/* cv2.resize(INTER_LINEAR) of load_image (base.py:152-157). */
int dy_aug_resize_u8(const uint8_t* src, int sh, int sw, int64_t src_pitch, uint8_t* dst, int dh, int dw, int64_t dst_pitch, void* stream);
/* LetterBox + Format (augment.py:559-591,745-751): resize to new_h x new_w, border 114 (top / left given), out = uint8 [3, out_h, out_w] RGB. */
int dy_aug_letterbox(const uint8_t* src, int sh, int sw, int64_t src_pitch, int new_h, int new_w, int top, int left, int out_h, int out_w,
                     uint8_t* out, void* stream);
/* Mosaic canvas -> cv2.warpAffine(borderValue 114) -> RandomHSV -> RandomFlip x2 -> Format for B samples in one launch (augment.py:158-195,
 * 323-345,486-499,527-532,745-751); samples: DEVICE array of B descriptors; out = uint8 [B, 3, out_h, out_w] RGB = batch['img']. */
int dy_aug_mosaic_warp(const dy_aug_sample* samples, int B, int out_h, int out_w, uint8_t* out, void* stream);
/* DarkChannel / AtmLight / DarkIcA of preprocess_batch (train.py:42-68,81-96) without the device->host copy and the Python loop:
 * img f32 [B,3,H,W] in [0,1] (the darkened batch) -> A [B,3] (0..255 scale, as train.py:95), ica [B,1,H,W].  Deterministic: ties of the
 * reference's unstable argsort go by pixel index, the rows DarkIcA leaves uninitialised use the per-channel formula. */
int dy_dark_channel_prior(const float* img, int B, int H, int W, float* A, float* ica, void* stream);

#ifdef __cplusplus
}
#endif
#endif
