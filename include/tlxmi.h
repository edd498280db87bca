/*
 * tlxmi.h — C-ABI of libtlxmi.so, the MI355X (gfx950) forward-pass engine that sits under the
 * TensorLayerX-compatible layer surface of tlxcv_amd.
 *
 * The reference (tensorlayer/TLXCV) has NO native interface: every op on the hot path is a
 * `tlx.nn.*` layer call that TensorLayerX forwards to a backend library.  Each entry point below
 * therefore names the reference *call site* it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain C, POD descriptors, raw device pointers, sizes in elements unless stated otherwise;
 *   - the caller owns every buffer; nothing is retained after the call returns;
 *   - every launch is asynchronous on the `stream` argument (a hipStream_t passed as void*);
 *   - no allocation, no synchronisation inside a call (hipGraph-capturable);
 *   - returns TLXMI_OK (0) or a negative tlxmi_status; tlxmi_last_error() gives a thread-local
 *     message.  Nothing throws across this boundary.
 *   - activations are NHWC ("pixel-major"): element (n,h,w,c) of a tensor with pixel stride `ld`
 *     lives at ((n*H + h)*W + w)*ld + c.  A token matrix [rows][features] is the H=W=1 case.
 *   - dtype: TLXMI_F16 = IEEE binary16 storage, fp32 accumulate; TLXMI_F32 = fp32 storage, exact
 *     fp32 MFMA/FMA accumulate (the parity mode: 1e-4 vs the CPU oracle).
 *   - one process drives ONE device (one rank per GPU): the library binds to the device that is current at its first
 *     launch and returns TLXMI_ERR_UNSUPPORTED for launches under another current device.
 *   - buffer offsets are 32-bit: a single tensor handed to conv2d / linear / the GEMM family must stay under 2 GiB
 *     (256 images of ResNet-50's largest map are 411 MB); larger ones return TLXMI_ERR_UNSUPPORTED — split the batch.
 *   - the product library reads no environment variable; tuning knobs exist only in the -DTLXMI_TUNING flavour.
 */
#ifndef TLXMI_H
#define TLXMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 101 (round 5): + tlxmi_linear_stats / tlxmi_linear_ln / tlxmi_linear_ln_supported, tlxmi_attention_windows,
 * tlxmi_mlp_seam(_supported), tlxmi_softmax_rows, tlxmi_patch_embed4_pos, tlxmi_multiclass_nms_index.  100 (round 4) had removed
 * tlxmi_row_stats / tlxmi_linear_ln(11 args) / tlxmi_layernorm_linear of the round-3 header without a version step: a caller built
 * against that header must check tlxmi_version() >= 101 and use the signatures below. */
#define TLXMI_VERSION 101

typedef enum tlxmi_status {
    TLXMI_OK = 0,
    TLXMI_ERR_BAD_ARG = -1,      /* null pointer, non-positive extent, inconsistent descriptor */
    TLXMI_ERR_UNSUPPORTED = -2,  /* legal request this build has no kernel for */
    TLXMI_ERR_ALIGNMENT = -3,    /* pointer / stride does not meet the documented alignment */
    TLXMI_ERR_LAUNCH = -4,       /* HIP reported an error at launch */
    TLXMI_ERR_NO_DEVICE = -5     /* no gfx950 device visible */
} tlxmi_status;

typedef enum tlxmi_dtype { TLXMI_F16 = 0, TLXMI_F32 = 1 } tlxmi_dtype;

/* Activations fused into epilogues.  Reference layers: nn.ReLU (resnet.py:50), nn.ReLU6
 * (mobilenetv2.py), nn.LeakyReLU(0.1) (darknet.py:50), nn.Hardswish / nn.HardSigmoid
 * (mobilenetv3.py), tlx.ops.GeLU exact-erf (vision_transformer.py:70), nn.Sigmoid, swish. */
typedef enum tlxmi_act {
    TLXMI_ACT_NONE = 0,
    TLXMI_ACT_RELU = 1,
    TLXMI_ACT_RELU6 = 2,
    TLXMI_ACT_LEAKY = 3,       /* x>=0 ? x : act_param*x */
    TLXMI_ACT_HARDSWISH = 4,   /* x*relu6(x+3)/6 */
    TLXMI_ACT_HARDSIGMOID = 5, /* relu6(x+3)/6 */
    TLXMI_ACT_GELU = 6,        /* 0.5x(1+erf(x/sqrt2)) */
    TLXMI_ACT_SIGMOID = 7,
    TLXMI_ACT_SILU = 8
} tlxmi_act;

/* epilogue flag bits */
#define TLXMI_EPI_RES_AFTER_ACT 1u /* y = act(a*scale+shift) + res   (darknet.py:155-159)      */
                                   /* default: y = act(a*scale+shift+res) (resnet.py:154-155)  */
#define TLXMI_EPI_MAXPOOL_3S2P1 4u /* y = maxpool(k 3, stride 2, pad 1) of the conv + epilogue result, written as
                                     [N][Ho/2][Wo/2][y_ld]; the conv map itself is never stored (resnet.py:287-290).
                                     fp16 stem geometry only: ask tlxmi_conv2d_maxpool_supported() first */
#define TLXMI_EPI_RES_BCAST_N 2u   /* res has no batch axis (pos_embed, vision_transformer.py:323) */
/* planning hints (same `flags` word; they choose tiles, never change results beyond a tile shape's summation order): the
 * launch will SHARE the device with launches of another stream (a caller running two half batches on two streams).
 * _HALF: workgroup rounds are priced for half the device's CUs (fewer, larger tiles; the other stream fills the rest);
 * _FULL: priced for the whole device.  Either one: a short last round is not split off into a launch of its own.  The
 * hint travels with the call: the library keeps no planning state, so concurrent host threads cannot disturb each other.
 * (The reference has no counterpart: TensorLayerX's backends pick their own launch shapes.) */
#define TLXMI_PLAN_SHARED_HALF 0x100u
#define TLXMI_PLAN_SHARED_FULL 0x200u

/* ------------------------------------------------------------------------------------------
 * Library / device
 * ---------------------------------------------------------------------------------------- */
int tlxmi_version(void);
const char* tlxmi_last_error(void);
/* Number of visible HIP devices whose gcnArchName starts with "gfx950"; <0 on HIP error. */
int tlxmi_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Layout conversion at the model boundary.
 * Replaces the implicit NCHW handling of tlx.nn.GroupConv2d(data_format='channels_first')
 * (resnet.py:199-207, vision_transformer.py:197-204): the engine is NHWC inside.
 *   src: [N][C][H][W] contiguous, src_dtype;  dst: [N][H][W][Cpad] dst_dtype, channels C..Cpad-1 = 0.
 * ---------------------------------------------------------------------------------------- */
int tlxmi_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int C, int H,
                       int W, int Cpad, void* stream);
/* Same with a b x b space-to-depth fold: dst [N][H/b][W/b][Cpad], channel (ph*b + pw)*C + c holds
 * src[n][c][h2*b+ph][w2*b+pw].  Lets the 3-channel stem / patch-embedding convs (resnet.py:199-207 7x7/2,
 * vision_transformer.py:197-204 16x16/16, swin_transformer.py:490 4x4/4) run as dense-K implicit GEMMs
 * (K 448 -> 256, 2048 -> 768, 128 -> 48 halves) with filters re-indexed once on the host side. */
int tlxmi_nchw_to_nhwc_s2d(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int C, int H,
                           int W, int b, int Cpad, void* stream);
/* Patch rows for a patch-embedding conv run as a Linear (vision_transformer.py:197-204 proj + :321-323 cat / pos_embed):
 * dst [N][lead + (H/ps)(W/ps)][C*ps*ps], element (c*ps + ky)*ps + kx of patch (py, px) = src[n][c][py*ps+ky][px*ps+kx] (the
 * conv filter [Cout][C][ps][ps] flattened is the Linear weight); the `lead` rows in front of each image's patches (the cls
 * token's slot) are zero.  ps a multiple of 8; src / dst 16-byte aligned. */
int tlxmi_patchify(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int C, int H, int W, int ps,
                   int lead, void* stream);
/* Swin's patch embedding in one pass (swin_transformer.py:471-505: Conv2d(3 -> D, kernel 4, stride 4) -> flatten -> transpose ->
 * LayerNorm(D)).  x [N][3][H][W] fp32 / fp16; w [D][64] fp16 with k = 16 c + 4 ky + kx for k < 48 and zeros above (the conv filter
 * [D][3][4][4] flattened and padded); bias / gamma / beta fp32 [D] or null (gamma == beta == null: no LayerNorm); y [N*H/4*W/4][D]
 * fp16.  D in {96, 128, 192, 256}; H, W multiples of 4; all buffers 16-byte aligned. */
int tlxmi_patch_embed4(const void* x, int x_dtype, const void* w, const float* bias, const float* gamma,
                       const float* beta, void* y, int N, int H, int W, int D, float eps, void* stream);
/* The same + SwinTransformer(ape=True)'s absolute position embedding (swin_transformer.py:561-565, 603-604): pos fp32 [H/4 * W/4][D]
 * is added to every image's tokens after the LayerNorm. */
int tlxmi_patch_embed4_pos(const void* x, int x_dtype, const void* w, const float* bias, const float* gamma, const float* beta,
                           const float* pos, void* y, int N, int H, int W, int D, float eps, void* stream);
/* dst: [N][C][H][W] contiguous; src: NHWC with pixel stride ld (>= C). */
int tlxmi_nhwc_to_nchw(const void* src, int src_dtype, int ld, void* dst, int dst_dtype, int N, int C,
                       int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * Filter packing (once, at set_eval()).
 *   src: OIHW fp32 [Cout][Cin/groups][R][S] (the layout TensorLayerX's torch backend keeps)
 *   dst: [Cout_pad][Kpad] dtype, K index = (r*S + s)*Cin_pad + c, zero padded;
 *        Cin_pad*sizeof(dtype) % 16 == 0, Cout_pad % 128 == 0, Kpad*sizeof(dtype) % 128 == 0.
 * ---------------------------------------------------------------------------------------- */
size_t tlxmi_packed_filter_bytes(int Cout, int Cin, int R, int S, int dtype);
int tlxmi_pack_filter(const float* src_oihw, void* dst, int Cout, int Cin, int R, int S, int dtype,
                      void* stream);

/* Fold eval-mode BatchNorm into per-channel (scale, shift):
 *   scale = gamma / sqrt(var + eps);  shift = beta - mean*scale + conv_bias*scale
 * Reference: nn.BatchNorm2d after every GroupConv2d (resnet.py:107,122,134; darknet.py:48;
 * mobilenetv1.py:61).  Any of gamma/beta/mean/var/conv_bias may be NULL (1/0/0/1/0). */
int tlxmi_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var,
                  const float* conv_bias, float eps, int C, float* scale, float* shift, void* stream);

/* ------------------------------------------------------------------------------------------
 * Conv2d as implicit GEMM, groups == 1.
 * Replaces nn.GroupConv2d(+BatchNorm2d +activation +residual add): resnet.py:142-156,
 * vision_transformer.py:197-220 (patch embed), darknet.py:54-58, yolov3.py:313-322;
 * with R=S=1,H=W=1 it is also nn.Linear (+bias +GELU +residual): vision_transformer.py:81-87,
 * 112-123, resnet.py:234-237.
 * ---------------------------------------------------------------------------------------- */
typedef struct tlxmi_conv2d_desc {
    int32_t dtype;           /* tlxmi_dtype of x, w, y, res */
    int32_t N, H, W, C;      /* input extent; C = padded input channels (C*elt % 16 == 0) */
    int32_t Cout;            /* true output channels */
    int32_t R, S;            /* filter height, width */
    int32_t stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;
    int32_t Ho, Wo;          /* output extent.  Below the full-correlation extent of (H,W,pad,R,S,stride,dil) it crops
                              * (asymmetric padding of the space-to-depth stem); above it, the last windows run past the
                              * bottom / right edge and read zeros (one-sided end padding: padding='SAME' at stride 2,
                              * efficientnet.py:92-125); every window must start inside the image or its leading pad */
    int32_t x_ld, y_ld, res_ld; /* pixel strides in elements (x_ld >= C, y_ld >= Cout) */
    int32_t y_nstride;       /* elements between images of y; 0 = dense (Ho*Wo*y_ld).  Lets the   */
    int32_t res_nstride;     /* patch-embed conv write rows 1.. of a [B][1+P][D] token matrix     */
                             /* (vision_transformer.py:321-323).  res: 0 = dense, or see BCAST.   */
    int32_t act;             /* tlxmi_act */
    float act_param;         /* LEAKY slope */
    uint32_t flags;          /* TLXMI_EPI_* */
} tlxmi_conv2d_desc;

int tlxmi_conv2d(const tlxmi_conv2d_desc* d, const void* x, const void* w_packed,
                 const float* scale /* [Cout] or NULL=1 */, const float* shift /* [Cout] or NULL=0 */,
                 const void* res /* or NULL */, void* y, void* stream);
/* 1 when tlxmi_conv2d takes this descriptor with TLXMI_EPI_MAXPOOL_3S2P1 set (a stride-1 4x4 conv on 16 fp16 channels
 * — the 7x7/2 ResNet stem after the 2x2 space-to-depth fold — with 112-pixel output rows), else 0: the caller then
 * runs tlxmi_conv2d and tlxmi_maxpool2d as two launches. */
int tlxmi_conv2d_maxpool_supported(const tlxmi_conv2d_desc* d);

/* ------------------------------------------------------------------------------------------
 * YOLOv3 post-processing (tlxcv/models/detection/yolov3.py:541-579, utils/ops.py:255-329).
 * tlxmi_yolo_box: one head map x ([N][A*(5+C)][H][W], or [N][H][W][A*(5+C)] with channels_last) -> boxes
 *   [N][Mtot][4] (x1, y1, x2, y2 in image pixels) and scores [N][Mtot][C] fp32, written at box offset m_offset ..
 *   m_offset + A*H*W (the heads of one image append).  img_size [N][2] = (height, width) int32; anchors [A][2].
 *   The published algorithm of Paddle's yolo_box (the op the reference calls and only has on the Paddle backend).
 * tlxmi_multiclass_nms: tlx_multiclass_nms — best class per box, score >= score_threshold, class-aware greedy NMS
 *   (IoU > nms_threshold, descending score), the first keep_top_k survivors as rows (class, score, x1, y1, x2, y2) of
 *   detections [N][keep_top_k][6] (zero filled), counts [N].  M <= 65536 boxes per image; workspace:
 *   tlxmi_multiclass_nms_workspace_bytes(N, M).
 * ---------------------------------------------------------------------------------------- */
/* IoU-aware objectness of the YOLOv3 head (yolov3.py:355-376): x [pixels][A*(6+C)] NHWC (A IoU channels first) ->
 * y [pixels][A*(5+C)] with obj' = de_sigmoid(sigmoid(obj)^(1 - factor) * sigmoid(iou)^factor); box / class entries copied. */
int tlxmi_yolo_iou_aware(const void* x, void* y, int dtype, int64_t pixels, int A, int C, float factor, void* stream);
int tlxmi_yolo_box(const void* x, int dtype, int N, int A, int C, int H, int W, int channels_last, const int32_t* img_size,
                   const float* anchors, float conf_thresh, int downsample_ratio, int clip_bbox, float scale_x_y,
                   float* boxes, float* scores, int Mtot, int m_offset, void* stream);
size_t tlxmi_multiclass_nms_workspace_bytes(int N, int M);
int tlxmi_multiclass_nms(const float* boxes, const float* scores, int N, int M, int C, float score_threshold, float nms_threshold,
                         int keep_top_k, void* workspace, float* detections, int32_t* counts, void* stream);
/* The same with the kept boxes' indices (the index-returning NMS the for_mot branch of YOLOv3.forward asks of its post-process,
 * yolov3.py:70-78; Paddle's multiclass_nms(return_index=True), utils/ops.py:189-229): keep_index [N][keep_top_k] = position
 * of each detection row's box among the M boxes of its image, -1 beyond counts[n]. */
int tlxmi_multiclass_nms_index(const float* boxes, const float* scores, int N, int M, int C, float score_threshold,
                               float nms_threshold, int keep_top_k, void* workspace, float* detections, int32_t* counts,
                               int32_t* keep_index, void* stream);

/* ------------------------------------------------------------------------------------------
 * Image pre-processing on the device (the host pipeline of demo/image_classification/predict.py:22-29,
 * Compose([Resize((h, w)), Normalize(mean, std), ToTensor(data_format)])): uint8 HWC images in, network input out.
 *   images:  [N][H][W][C] uint8, C <= 4
 *   xbounds / xk, ybounds / yk: the resampler tables of the two passes — bounds[o] = (first input sample, count),
 *            k[o][t] integer weights with 22 fractional bits, row pitch kw / kh — built by the caller exactly as Pillow's
 *            ImagingResample does (tlxcv_amd/tlx/vision/transforms/resample.py), which makes the result bit-identical to
 *            PIL.Image.resize on the host; an axis that keeps its size passes the identity table (count 1, weight 1 << 22)
 *   mean / std: [C] fp32 when `normalize`, else the value is scaled by 1/255 (ToTensor on uint8)
 *   workspace: tlxmi_preprocess_u8_workspace_bytes(d) bytes (the uint8 image after the horizontal pass)
 *   out:     layout 0 [N][C][out_h][out_w] ("CHW"), 1 [N][out_h][out_w][C] ("HWC"), 2 the fold_b x fold_b space-to-depth
 *            NHWC image of tlxmi_nchw_to_nhwc_s2d ([N][out_h/b][out_w/b][cpad], channel (ph*b + pw)*C + c, zero padded)
 * ---------------------------------------------------------------------------------------- */
typedef struct tlxmi_preproc_desc {
    int32_t N, H, W, C;
    int32_t out_h, out_w;
    int32_t kh, kw;
    int32_t out_dtype;
    int32_t layout;
    int32_t fold_b, cpad;
    int32_t normalize;
} tlxmi_preproc_desc;
size_t tlxmi_preprocess_u8_workspace_bytes(const tlxmi_preproc_desc* d);
int tlxmi_preprocess_u8(const tlxmi_preproc_desc* d, const void* images, const int32_t* xbounds, const int32_t* xk,
                        const int32_t* ybounds, const int32_t* yk, const float* mean, const float* std_, void* workspace,
                        void* out, void* stream);

/* The detection demo's pipeline (demo/object_detection/transforms.py:96-246, predict-YOLOv3.py:54-61):
 * Resize(size, max_size, auto_divide) = cv2.resize(image, (ow, oh), INTER_LINEAR) on the uint8 image, then Normalize
 * ((v / 255 - mean) / std).  OpenCV's 8-bit bilinear resize restated: two taps per axis, 11-bit fixed-point weights.
 *   images: [N][H][W][C] uint8 (one size per call);  xidx / yidx: [out][2] source index of the two taps (already clamped
 *   to the image);  xcoef / ycoef: [out][2] weights (sum 2048), built by the caller as OpenCV's coefficient loop does
 *   (tlxcv_amd/tlx/vision/transforms/detection.py);  d->kh / kw / fold_b / cpad unused;  layout 0 ("CHW") or 1 ("HWC");
 *   without `normalize` the value is v / 255.  UNPINNED: OpenCV is not available to check against. */
int tlxmi_preprocess_linear_u8(const tlxmi_preproc_desc* d, const void* images, const int32_t* xidx, const int32_t* xcoef,
                               const int32_t* yidx, const int32_t* ycoef, const float* mean, const float* std_, void* out,
                               void* stream);

/* ------------------------------------------------------------------------------------------
 * The seam between two ResNet bottleneck blocks in one launch (resnet.py:142-156 of block b, :143-145 of block b + 1):
 *     y  = relu( (t2 . W3^T) * scale3 + shift3 + skip )      conv3 + bn3 + residual add + relu of block b
 *     t1 = relu( (y  . W1^T) * scale1 + shift1 )             conv1 + bn1 + relu of block b + 1
 * t2: [rows][t2_ld] (K1 channels), skip / y: [rows][..] (N1 channels), t1: [rows][t1_ld] (N2 channels); w3_packed /
 * w1_packed: tlxmi_pack_filter images of the two 1x1 filters ([N1][K1] and [N2][N1]).  The wide map y is written once
 * and never re-read by the second convolution.  fp16 only, ReLU only (`act`), channel triples with a compiled kernel:
 * ask tlxmi_bottleneck_seam_supported(dtype, K1, N1, N2) first; otherwise two tlxmi_conv2d launches do the same.
 * ---------------------------------------------------------------------------------------- */
typedef struct tlxmi_seam_desc {
    int32_t dtype;
    int64_t rows;                        /* pixels: N * H * W */
    int32_t K1, N1, N2;
    int32_t t2_ld, skip_ld, y_ld, t1_ld; /* elements between rows */
    int32_t act;                         /* TLXMI_ACT_RELU */
} tlxmi_seam_desc;
int tlxmi_bottleneck_seam_supported(int dtype, int K1, int N1, int N2);
/* tlxmi_bottleneck_seam_proj: the seam of a block WITH a projection shortcut (resnet.py:246-261, layer1.0: 1x1 conv + BN on
 * the block's input at stride 1): skip = (x . Wd^T) * scale_d + shift_d is computed in the launch instead of being read
 * from a stored map.  x: [rows][skip_ld], K1 channels.  fp16, K1 = N2 = 64 only (TLXMI_ERR_UNSUPPORTED otherwise). */
int tlxmi_bottleneck_seam_proj(const tlxmi_seam_desc* d, const void* t2, const void* w3_packed, const float* scale3,
                               const float* shift3, const void* x, const void* wd_packed, const float* scale_d,
                               const float* shift_d, void* y, const void* w1_packed, const float* scale1,
                               const float* shift1, void* t1, void* stream);
int tlxmi_bottleneck_seam(const tlxmi_seam_desc* d, const void* t2, const void* w3_packed, const float* scale3,
                          const float* shift3, const void* skip, void* y, const void* w1_packed, const float* scale1,
                          const float* shift1, void* t1, void* stream);

/* ------------------------------------------------------------------------------------------
 * Grouped convolution, 1 < groups < C: nn.GroupConv2d(n_group=cardinality) of the ResNeXt bottleneck,
 * resnext.py:30-40 (constructed at :83-91 with groups = 32 / 64), + BatchNorm(act='relu') :46-52.
 * d->C / d->Cout are the TOTAL input / output channels (both divisible by groups), x_ld / y_ld the pixel
 * strides; everything else as tlxmi_conv2d.  m consecutive groups are merged into one launch chunk with a
 * block-diagonal filter (tlxmi_group_conv_chunks() chunks of C/chunks -> Cout/chunks channels; 0 = the
 * channel counts cannot be merged into 16-byte aligned chunks -> TLXMI_ERR_UNSUPPORTED).
 *   src: OIHW fp32 [Cout][Cin/groups][R][S]; dst: `chunks` packed filters of tlxmi_pack_filter's layout.
 * groups == 1 forwards to tlxmi_conv2d / tlxmi_pack_filter; groups == C is tlxmi_dwconv2d's job.
 * ---------------------------------------------------------------------------------------- */
int tlxmi_group_conv_chunks(int Cin, int Cout, int groups, int dtype);
size_t tlxmi_packed_group_filter_bytes(int Cout, int Cin, int R, int S, int groups, int dtype);
int tlxmi_pack_group_filter(const float* src_oihw, void* dst, int Cout, int Cin, int R, int S, int groups,
                            int dtype, void* stream);
int tlxmi_group_conv2d(const tlxmi_conv2d_desc* d, int groups, const void* x, const void* w_packed,
                       const float* scale, const float* shift, const void* res, void* y, void* stream);
/* 1 when the small-block MFMA kernel (group_conv.hip: v_mfma_f32_4x4x4_16b_f16, sixteen 4x4x4 products per
 * instruction, so no MFMA work on the zero blocks of the block-diagonal filter) can run this layer (without a
 * residual): fp16, C == Cout a multiple of 64, 4 / 8 / 16 / 32 channels per group, 3x3, padding 1, stride 1 or 2,
 * dilation 1, dense output, W <= 104.  tlxmi_group_conv2d takes it where it measured faster (4 and 8 channels per
 * group, 16 at stride 1).  Same operands and results (to fp32 summation order) as the general path. */
int tlxmi_group_conv2d_small_supported(const tlxmi_conv2d_desc* d, int groups);

/* ------------------------------------------------------------------------------------------
 * nn.Linear with few rows and a large filter — the classifier heads: resnet.py:234-237, vgg.py:42-50 (25088 -> 4096),
 * alexnet.py:162-168, vision_transformer.py:333.  K is cut into `splits` slices that run side by side (K % splits == 0,
 * (K/splits)*elt % 128 == 0); `partials` is a caller-owned scratch of splits*rows*Cout FLOATS (fp32 for both dtypes: the
 * fp32 accumulators of a slice are stored unrounded); the partial sums are added in slice order (deterministic, fp32), then
 * y = act((sum)*scale + shift (+res)) as tlxmi_conv2d.
 * w_packed: tlxmi_pack_filter of the [Cout][K] weight (1x1).
 * ---------------------------------------------------------------------------------------- */
int tlxmi_linear_splitk(int dtype, int64_t rows, int K, int Cout, int x_ld, const void* x, const void* w_packed,
                        int splits, void* partials, const float* scale, const float* shift, const void* res,
                        int res_ld, int act, float act_param, uint32_t flags, void* y, int y_ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * Convolution with few output pixels and a long K — the 3x3 convs of ResNet's 7 x 7 stage (resnet.py:111-121: N*49 pixels,
 * K = 9 * 512) and the strided 1x1 / 3x3 convs entering it: K is cut into `splits` slices that run side by side, each
 * storing its fp32 accumulators into `partials` (caller-owned, splits * N*Ho*Wo * Cout floats); the partial sums are added
 * in slice order (deterministic), then y = act(sum*scale + shift (+res)) exactly as tlxmi_conv2d.  Same descriptor and packed
 * filter as tlxmi_conv2d.  Supported: what tlxmi_conv2d_splitk_supported() reports (R x 3 filters or strided 1x1, dilation 1,
 * C * elt a power-of-two multiple of 128 bytes, Cout >= 128 and % 8, dense y / res, >= 4 K tiles of 128 bytes per slice).
 * ---------------------------------------------------------------------------------------- */
int tlxmi_conv2d_splitk_supported(const tlxmi_conv2d_desc* d, int splits);
int tlxmi_conv2d_splitk(const tlxmi_conv2d_desc* d, int splits, const void* x, const void* w_packed, void* partials,
                        const float* scale, const float* shift, const void* res, void* y, void* stream);

/* ------------------------------------------------------------------------------------------
 * Depthwise conv (groups == C == Cout), HBM-bound, no MFMA.
 * Replaces nn.GroupConv2d(n_group=C)+BN+act: mobilenetv1.py:79-88, mobilenetv2.py:30,
 * mobilenetv3.py (k3/k5).   w: [R][S][C] dtype.
 * ---------------------------------------------------------------------------------------- */
typedef struct tlxmi_dwconv2d_desc {
    int32_t dtype;
    int32_t N, H, W, C;
    int32_t R, S, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;
    int32_t Ho, Wo;
    int32_t x_ld, y_ld;
    int32_t act;
    float act_param;
} tlxmi_dwconv2d_desc;
int tlxmi_dwconv2d(const tlxmi_dwconv2d_desc* d, const void* x, const void* w_rsc, const float* scale,
                   const float* shift, void* y, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pooling.  nn.MaxPool2d(3,2,padding=1) resnet.py:213-218 (padding value -inf);
 * nn.AdaptiveAvgPool2d((1,1)) resnet.py:228-231 / mobilenetv1.py:246; AdaptiveAvgPool1d(1) over
 * tokens swin_transformer.py:609.
 * ---------------------------------------------------------------------------------------- */
int tlxmi_maxpool2d(const void* x, void* y, int dtype, int N, int H, int W, int C, int x_ld, int y_ld,
                    int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int Ho, int Wo,
                    void* stream);
/* nn.AvgPool2d(kernel, stride, padding) with zero padding counted in the divisor (always R*S): the anti-aliasing
 * down-sampling of ResNeSt, resnest.py:212-218, 250-256 (3x3, stride s, pad 1) and :271-286 (s x s, stride s, pad 0). */
int tlxmi_avgpool2d(const void* x, void* y, int dtype, int N, int H, int W, int C, int x_ld, int y_ld,
                    int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int Ho, int Wo, void* stream);

/* Split attention of ResNeSt's SplatConv (resnest.py:147-166; rSoftmax :53-82).  x: [N][HW][x_ld >= radix*C], split r =
 * channels [r*C, (r+1)*C);  g: [N][g_ld >= C];  logit: [N][l_ld >= radix*C] in the channel order conv3 (:157) produces;
 *   tlxmi_radix_gap:        g[n][c] = mean_p sum_r x[n][p][r*C + c]                               (:150-155)
 *   tlxmi_split_attention:  y[n][p][c] = sum_r a_r(n,c) * x[n][p][r*C + c]                        (:158-165)
 * with, for radix > 1, a_r(n,c) = softmax over r of logit[n][(k*radix + r)*cpg + c'] (c = k*cpg + c', cpg = C/cardinality:
 * rSoftmax's reshape-transpose-softmax(axis=1)-reshape), and for radix == 1, a = sigmoid(logit[n][c]). */
int tlxmi_radix_gap(const void* x, void* g, int dtype, int N, int HW, int C, int radix, int x_ld, int g_ld, void* stream);
/* att_ws: caller-owned fp32 scratch of N * radix * C values (the attention weights in split order; two launches). */
int tlxmi_split_attention(const void* x, const void* logit, float* att_ws, void* y, int dtype, int N, int HW, int C,
                          int radix, int cardinality, int x_ld, int l_ld, int y_ld, void* stream);
/* y[n][c] = mean over H*W of x[n][.][.][c];  y pixel stride y_ld */
int tlxmi_global_avgpool(const void* x, void* y, int dtype, int N, int HW, int C, int x_ld, int y_ld,
                         void* stream);
/* nn.AdaptiveAvgPool2d((OH, OW)), vgg.py:36-39: y[n][oh][ow][c] = mean of x over rows [floor(oh*H/OH),
 * ceil((oh+1)*H/OH)) and the matching columns.  x: [N][H][W][x_ld], y: [N][OH][OW][y_ld]. */
int tlxmi_adaptive_avgpool2d(const void* x, void* y, int dtype, int N, int H, int W, int C, int OH, int OW,
                             int x_ld, int y_ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise: y = act(x*scale[c] + shift[c]) (+ res).  Used when a BatchNorm / activation / add
 * cannot be fused into a producer (stand-alone nn.BatchNorm2d, nn.ReLU, `out += identity`).
 * ---------------------------------------------------------------------------------------- */
int tlxmi_affine_act(const void* x, const float* scale, const float* shift, const void* res, void* y,
                     int dtype, int64_t rows, int C, int x_ld, int res_ld, int y_ld, int act,
                     float act_param, uint32_t flags, void* stream);

/* Squeeze-Excitation gating (mobilenetv3.py:54-56 `scale * input`): y[n][p][c] = x[n][p][c] * s[n][c],
 * s is the (N, C) output of the SE bottleneck, same dtype as x. */
int tlxmi_scale_channels(const void* x, const void* s, void* y, int dtype, int N, int HW, int C, int x_ld,
                         int s_ld, int y_ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dimension (biased variance), nn.LayerNorm(dim, epsilon)
 * vision_transformer.py:144,159,283; swin_transformer.py:258,279,371,495,591.
 * ---------------------------------------------------------------------------------------- */
int tlxmi_layernorm(const void* x, const float* gamma, const float* beta, void* y, int dtype,
                    int64_t rows, int C, int x_ld, int y_ld, float eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm fused with Swin's window plumbing, SwinTransformerBlock.forward swin_transformer.py:315-335:
 *   layernorm_window_partition: win = window_partition(roll(norm1(x), -shift))          (:315-324)
 *   window_reverse_layernorm:   sum = res + roll(window_reverse(win), +shift); y = norm2(sum)   (:327-335)
 * x / res / sum / y: [B][H][W][C] dense; win: [B*nW][ws*ws][C]; same index map as tlxmi_window_partition.
 * ---------------------------------------------------------------------------------------- */
int tlxmi_layernorm_window_partition(const void* x, const float* gamma, const float* beta, void* win, int dtype,
                                     int B, int H, int W, int C, int ws, int shift, float eps, void* stream);
int tlxmi_window_reverse_layernorm(const void* win, const void* res, const float* gamma, const float* beta,
                                   void* sum, void* y, int dtype, int B, int H, int W, int C, int ws, int shift,
                                   float eps, void* stream);

/* Row softmax over the last axis (tlx.ops.softmax / nn.Softmax of a reference forward run layer by layer: detr.py:1011-1043,
 * vision_transformer.py:118): y[r][c] = exp(x[r][c] - max_r) / sum_c exp(...), fp32 arithmetic, dtype of x out; any C, strides in elements. */
int tlxmi_softmax_rows(const void* x, void* y, int dtype, int64_t rows, int C, int64_t x_ld, int64_t y_ld, void* stream);

/* A transformer MLP as one launch (swin_transformer.py:62-82, :335): out = fc2(gelu(fc1(x) + b1)) + b2 + res; the hidden activations
 * (rows x hidden) never leave the CU.  fp16; x [rows][x_ld] (K channels), res / out [rows][ld] (N channels); w1 / w2 packed by
 * tlxmi_pack_filter (1 x 1) as [hidden][K] and [N][hidden].  Compiled for K = N = 128, hidden a multiple of 64 up to 2048 (Swin-B
 * stage 1); tlxmi_mlp_seam_supported answers 1 when a shape is taken. */
int tlxmi_mlp_seam_supported(int dtype, int K, int hidden, int N);
int tlxmi_mlp_seam(int dtype, int64_t rows, int K, int hidden, int N, const void* x, int x_ld, const void* w1_packed, const float* bias1,
                   const void* w2_packed, const float* bias2, const void* res, int res_ld, void* out, int out_ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm folded AROUND the Linear layers of a transformer block (fp16; vision_transformer.py:144-175: norm1 -> attn.qkv,
 * norm2 -> mlp.fc1; swin_transformer.py:310-337).  The Linear that PRODUCES the residual stream (proj, fc2, the patch embedding)
 * also emits the row statistics of its output, the Linear that CONSUMES a LayerNorm applies it in its epilogue; the normalised
 * activations are never written or read:
 *   tlxmi_linear_stats   y[m][n] = sum_k x[m][k] W[n][k] + bias[n] (+ res[m][n]);  partials[m][p] = (sum, sum of squares) of
 *                        y[m][256 p .. 256 p + 255] (the fp32 values before rounding) for p < ceil(Cout / 256) <= 4 — `partials` is
 *                        rows x 4 x 2 floats (32 bytes a row, 16-byte aligned); pairs past ceil(Cout / 256) are not written
 *   tlxmi_linear_ln      per row m: mean = sum_p partials[m][p][0] / K, var = sum_p partials[m][p][1] / K - mean^2 (biased, as
 *                        nn.LayerNorm), rstd = 1 / sqrt(var + eps);  y[m][n] = act(rstd * sum_k x[m][k] Wg[n][k] - mean * rstd * c1[n]
 *                        + c2[n]) on the RAW rows x, with Wg = W * gamma packed by tlxmi_pack_filter (1 x 1), c1[n] = sum_k Wg[n][k]
 *                        (of the fp16 values as packed), c2[n] = bias[n] + sum_k W[n][k] * beta[k];  act: TLXMI_ACT_NONE or
 *                        TLXMI_ACT_GELU;  `partials`: what a tlxmi_linear_stats launch with Cout = K left for these rows (its first
 *                        ceil(K / 256) pairs of a row are read) — no launch in between
 * Both run on the 256 x 256 GEMM kernels (the persistent one; a residual with K < 704 and launches of few tiles on its
 * one-tile-per-workgroup form): fp16, Cout % 32 == 0, Cout >= 256, K >= 128, K <= 1024 for the consumer;
 * tlxmi_linear_ln_supported(dtype, rows, K, Cout, act, with_res) — with_res 0: the consumer, 1: a producer with a residual, 2: a producer
 * without (a producer's Cout, the consumer's K <= 1024: the LayerNorm's width) — answers 1 when the shape is taken, otherwise the calls return
 * TLXMI_ERR_UNSUPPORTED and the caller keeps tlxmi_layernorm + tlxmi_conv2d.  `flags`: TLXMI_PLAN_SHARED_* or 0.
 * ---------------------------------------------------------------------------------------- */
int tlxmi_linear_ln_supported(int dtype, int64_t rows, int K, int Cout, int act, int with_res);
int tlxmi_linear_stats(int dtype, int64_t rows, int K, int Cout, int x_ld, int y_ld, const void* x, const void* w_packed,
                       const float* bias, const void* res, int res_ld, void* y, float* partials, unsigned flags, void* stream);
int tlxmi_linear_ln(int dtype, int64_t rows, int K, int Cout, int x_ld, int y_ld, const void* x, const void* w_packed,
                    const float* c1, const float* c2, const float* partials, float eps, int act, void* y, unsigned flags, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused multi-head self attention on a packed qkv matrix (the output of the qkv Linear):
 *   qkv: [B][Ntok][3][heads][hd]   (vision_transformer.py:112-116; swin_transformer.py:194-200)
 *   out: [B][Ntok][heads*hd]       softmax(scale * q k^T + bias + mask) v, heads re-interleaved
 *   bias: optional [heads][Ntok][Ntok] fp32 (Swin relative position bias, :205-215)
 *   mask: optional [nW][Ntok][Ntok] fp32, window index = b % nW (Swin shift mask, :216-220)
 * hd <= 128.  Ntok <= 256 runs the tuned kernels (fp16: MFMA, hd in {32, 64, 96}); longer sequences (ViT at
 * 384 x 384: 577 tokens) run a generic online-softmax kernel.
 * ---------------------------------------------------------------------------------------- */
typedef struct tlxmi_attn_desc {
    int32_t dtype;
    int32_t B, Ntok, heads, hd;
    float scale;
    int32_t nW;              /* mask windows (0 = no mask) */
} tlxmi_attn_desc;
int tlxmi_attention(const tlxmi_attn_desc* d, const void* qkv, const float* bias, const float* mask,
                    void* out, void* stream);
/* Same, with bias + mask pre-summed and padded by the caller (once per layer instead of once per forward,
 * swin_transformer.py:205-220): comb[w][h][i][j] = bias[h][i][j] + mask[w][i][j] for i, j < Ntok, 0 elsewhere,
 * shape [max(nW,1)][heads][NP][NP] fp32 with NP = 32 * ceil(Ntok / 32).  fp16, hd in {32, 64, 96}, Ntok <= 256;
 * TLXMI_ERR_UNSUPPORTED otherwise (use tlxmi_attention). */
int tlxmi_attention_comb(const tlxmi_attn_desc* d, const void* qkv, const float* comb, void* out, void* stream);

/* Swin's windowed attention on IMAGE-order token matrices (swin_transformer.py:316-333): qkv [B / wpi][H * W][3][heads][hd] (the qkv
 * Linear applied to the rows of the residual stream), out [B / wpi][H * W][heads * hd]; the cyclic shift, window_partition and
 * window_reverse are row arithmetic inside the kernel.  d->B = number of windows (images * wpi, wpi = (H / ws) * (W / ws)),
 * d->Ntok = ws * ws <= 64, d->nW = 0 (no shift mask in `comb`) or wpi; comb as tlxmi_attention_comb.  fp16. */
int tlxmi_attention_windows(const tlxmi_attn_desc* d, const void* qkv, const float* comb, void* out, int H, int W, int ws, int shift,
                            void* stream);


/* ------------------------------------------------------------------------------------------
 * General multi-head attention on separate, strided Q / K / V — the core of tlx.nn.MultiheadAttention and of DETR's
 * MultiHeadAttention.forward (tlxcv/models/detection/detr.py:1003-1062): query and key lengths may differ, layouts
 * are given by element strides (sequence-first [L][B][D]: batch stride D, row stride B*D; batch-first: batch stride
 * L*D, row stride D; a packed qkv matrix: three pointers into it with row stride 3*D), head h of a row is the hd
 * values at offset h*hd.
 *   out[b][i][h*hd + :] = sum_j softmax_j( (scale * q_i) . k_j + mask[..][i][j] ) v_j
 *   mask_mode 0: none; 1: one fp32 [Lq][Lk] mask for every (batch, head); 2: [B*heads][Lq][Lk] (detr.py:1038-1039)
 *   avg_weights: optional fp32 [B][Lq][Lk] output, the softmax weights averaged over the heads (detr.py:1054-1060);
 *                NULL = not wanted.
 * hd <= 128.  A coverage kernel (any lengths), not a tuned one: self attention over a packed qkv of <= 256 tokens
 * belongs on tlxmi_attention.
 * ---------------------------------------------------------------------------------------- */
typedef struct tlxmi_mha_desc {
    int32_t dtype;
    int32_t B, Lq, Lk, heads, hd;
    float scale;
    int32_t mask_mode;
    int64_t q_batch_stride, q_row_stride, k_batch_stride, k_row_stride, v_batch_stride, v_row_stride,
            out_batch_stride, out_row_stride;       /* in elements */
} tlxmi_mha_desc;
int tlxmi_mha(const tlxmi_mha_desc* d, const void* q, const void* k, const void* v, const float* mask,
              void* out, float* avg_weights, void* stream);

/* ------------------------------------------------------------------------------------------
 * Swin window plumbing folded into index math (swin_transformer.py:85-116, 317-333):
 *   partition: x[B][H][W][C] --roll(-shift)--> windows [B*nW][ws*ws][C]
 *   reverse:   windows --> x (+roll(+shift)), optionally y = res + reverse(windows)
 * ---------------------------------------------------------------------------------------- */
int tlxmi_window_partition(const void* x, void* win, int dtype, int B, int H, int W, int C, int ws,
                           int shift, void* stream);
int tlxmi_window_reverse(const void* win, const void* res, void* y, int dtype, int B, int H, int W,
                         int C, int ws, int shift, void* stream);
/* PatchMerging gather (swin_transformer.py:373-388): x[B][H][W][C] -> y[B][H/2][W/2][4C],
 * channel blocks in the order (0,0),(1,0),(0,1),(1,1). */
int tlxmi_patch_merge_gather(const void* x, void* y, int dtype, int B, int H, int W, int C,
                             void* stream);
/* the same gather + LayerNorm over the 4C merged channels in one pass (swin_transformer.py:381-388: PatchMerging's
 * cat + self.norm): y[B*(H/2)*(W/2)][4C]; gamma / beta fp32 [4C], 16-byte aligned. */
int tlxmi_patch_merge_layernorm(const void* x, const float* gamma, const float* beta, void* y, int dtype, int B, int H,
                                int W, int C, float eps, void* stream);

/* nearest x2 upsample written at a channel offset of a wider NHWC buffer (yolov3.py:250-256:
 * interpolate(scale_factor=2) + concat).  y has pixel stride y_ld; channels [c_off, c_off+C). */
int tlxmi_upsample2x_nearest(const void* x, void* y, int dtype, int N, int H, int W, int C, int x_ld,
                             int y_ld, int c_off, void* stream);
/* strided copy of an NHWC tensor into a channel window of another (tlx.concat along channels);
 * x_ld == 0 broadcasts one source row to every destination row (cls token, vision_transformer.py:321) */
int tlxmi_copy_channels(const void* x, void* y, int dtype, int64_t rows, int C, int x_ld, int y_ld,
                        void* stream);

/* argmax over the last dimension -> int64 (tasks/image_classification.py:23) */
int tlxmi_argmax_lastdim(const void* x, int dtype, int64_t rows, int C, int x_ld, int64_t* out,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TLXMI_H */
