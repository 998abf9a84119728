/*
 * libxdet_hip.so -- C-ABI of the MI355X-native Light-Head R-CNN forward path.
 *
 * Every entry point replaces one interface of HiKapok/X-Detector's eval path
 * (reference file:line cited per function).  Plain pointers and sizes only; all tensor
 * pointers are DEVICE pointers unless the name ends in _host.  Every function returns
 * 0 on success or a negative code (XDET_ERR_*); xdet_last_error() gives the message.
 * Nothing here synchronises the stream unless stated; `stream` is a hipStream_t (NULL =
 * default stream).  The caller owns every buffer; a net handle owns only its weights and
 * a fixed workspace sized at xdet_net_build().
 *
 * Activation layout inside the library is NHWC with the channel count padded to a
 * multiple of 32 ("ld"); the public PsRoiAlign op also accepts the reference's NCHW.
 * Boxes are (ymin, xmin, ymax, xmax) normalised to [0,1]; ROIs to PsRoiAlign (cy,cx,h,w).
 */
#ifndef XDET_H_
#define XDET_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XDET_OK 0
#define XDET_ERR_INVALID_ARG (-1) /* the op's OP_REQUIRES -> InvalidArgument, ps_roi_align_op.cc:209-226 */
#define XDET_ERR_HIP (-2)         /* HIP failure; reference: fprintf + exit(-1), ps_roi_align_op.cu:150-155 */
#define XDET_ERR_STATE (-3)
#define XDET_ERR_UNSUPPORTED (-4)

const char* xdet_last_error(void);
int xdet_version(void);
int xdet_device_count(int* n);
int xdet_set_device(int dev);
/* "0000:c1:00.0"-style PCI bus id of HIP device `dev` (what identifies a physical GPU across ranks: the launcher
 * binds a rank to the NUMA node of its GPU with it, bench.py gathers it from every rank); buflen >= 16 */
int xdet_device_pci_bus_id(int dev, char* buf, int buflen);
/* hipMalloc + hipIpcGetMemHandle + hipFree on the current device: 0 if this process can export device memory to a
 * peer process (what RCCL's intra-node transport needs), XDET_ERR_HIP otherwise.  Whether it works depends on
 * HSA_ENABLE_IPC_MODE_LEGACY, which the HSA runtime reads when it starts -- so xdet.launch runs this in a fresh
 * probe process before it decides what to export to the ranks. */
int xdet_probe_ipc(void);
/* Arithmetic of the conv / dense contractions for layers and nets created AFTER the call:
 *   0 = f32 MFMA (v_mfma_f32_32x32x2_f32, exact f32 FMA chains; default)
 *   1 = f16x3: operands split into f16 hi+lo parts, three v_mfma_f32_32x32x16_f16 per product
 *       block, f32 accumulate (~2^-21 relative per product: f32-class accuracy on the 2.5 PFLOP/s pipe)
 *   2 = plain f16 operands (speed mode; drifts beyond 1e-3 through the 40-layer stack) */
int xdet_set_default_precision(int mode);
int xdet_get_default_precision(void);

/* ---- memory / streams (what TF's allocator and stream executor did for the op) ---------- */
int xdet_malloc(void** dptr, size_t bytes);
int xdet_free(void* dptr);
int xdet_memset(void* dptr, int value, size_t bytes, void* stream);
int xdet_memcpy_h2d(void* dst, const void* src_host, size_t bytes, void* stream);
int xdet_memcpy_d2h(void* dst_host, const void* src, size_t bytes, void* stream);
int xdet_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
int xdet_stream_create(void** stream);
int xdet_stream_destroy(void* stream);
int xdet_stream_sync(void* stream);
/* hipEvent timing on `stream` (bench.py's roofline leg) */
int xdet_event_create(void** ev);
int xdet_event_destroy(void* ev);
int xdet_event_record(void* ev, void* stream);
int xdet_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms); /* syncs on ev_stop */

/* ---- A9: PsRoiAlign forward --------------------------------------------------------------
 * Replaces op_module.ps_roi_align(inputs, rois, grid_dim_width, grid_dim_height, pool_method)
 * (light_head_rfcn_eval.py:143-155; REGISTER_OP ps_roi_align_op.cc:38-76; CPU functor
 * :81-201; CUDA kernel ps_roi_align_op.cu:36-132).
 *   feat   f32 [N,C,H,W] (feat_layout 0) or [N,H,W,ldc] (feat_layout 1, first C channels used)
 *   rois   f32 [N,R,4] (cy,cx,h,w) -- or corner boxes when rois_are_corners != 0, in which case
 *          _point2center (net/xception_body.py:215-218) is applied on the fly
 *   pooled f32 [N,R,out_ld] (first C = gh*gw*bank entries per ROI, = [N,R,gh*gw,bank] when out_ld == C)
 *   index  i32 same shape, may be NULL.  Degenerate ROI: pooled 0 and index 0.
 * Errors: same argument checks as PSROIAlignOp::Compute (:209-226) -> XDET_ERR_INVALID_ARG. */
int xdet_psroialign_fwd(const float* feat, const float* rois, float* pooled, int32_t* index, int N, int C, int H,
                        int W, int R, int grid_w, int grid_h, int use_max, int feat_layout, int ldc, int out_ld,
                        int rois_are_corners, void* stream);

/* ---- F2: PsRoiAlignGrad (training backward of A9) ---------------------------
 * Replaces op_module.ps_roi_align_grad(inputs, rois, pooled_features_grad, pooled_index,
 * grid_dim_width, grid_dim_height, pool_method) (REGISTER_OP ps_roi_align_grad_op.cc:39-57; CUDA
 * kernel ps_roi_align_grad_op.cu:36-171; python gradient registration test_op.py:93-104).
 *   rois         f32 [N,R,4] (cy,cx,h,w)
 *   grad_pooled  f32 [N,R,C]   (= [N,R,gh*gw,bank])
 *   pooled_index i32 [N,R,C]   the forward's argmax sample ids ('max'; may be NULL for 'mean')
 *   grad_feat    f32 [N,C,H,W] (feat_layout 0) or [N,H,W,ldc] (feat_layout 1); zero-filled, then
 *                accumulated with float atomics exactly as the reference does -- summation order is
 *                therefore unspecified and results agree with a sequential evaluation to rounding. */
int xdet_psroialign_grad(const float* rois, const float* grad_pooled, const int32_t* pooled_index, float* grad_feat,
                         int N, int C, int H, int W, int R, int grid_w, int grid_h, int use_max, int feat_layout,
                         int ldc, void* stream);

/* ---- layer objects: the tf.layers.* kernels the graph builders call ---------------------
 * xdet_conv_create: tf.layers.conv2d / dense (+ folded inference BN / bias, + ReLU)
 * (net/xception_body.py:243-265,381-400,450-475,540-558; net/resnet_v2.py:89-100).
 *   kernel_hwio_host f32 [kh,kw,cin,cout]; scale_host/shift_host f32 [cout] (y = conv*scale+shift),
 *   either may be NULL (1 / 0).  pad_mode 0 = VALID, 1 = SAME (TF rule), 2 = explicit pad_t/pad_l.
 * xdet_conv_forward: in NHWC [N,H,W,ld_in] -> out NHWC [N,Ho,Wo,ld_out]; residual (may be NULL)
 *   has the output shape.  ld_in must be round_up(cin,32) (4 when cin <= 4), ld_out = round_up(cout,32).
 *   relu_in applies ReLU to the input on the fly (the `relu -> conv` edges of the graph). */
int xdet_conv_create(void** layer, int kh, int kw, int cin, int cout, int stride, int dilation, int pad_mode,
                     int pad_t, int pad_l, const float* kernel_hwio_host, const float* scale_host,
                     const float* shift_host, int relu_out);
int xdet_conv_forward(void* layer, const float* in, int N, int H, int W, int ld_in, float* out, int ld_out,
                      const float* residual, int relu_in, void* stream);
int xdet_conv_out_shape(void* layer, int H, int W, int* Ho, int* Wo);
/* Fixed split of the reduction (csrc/conv_mfma_ksplit.hip; layers fed with planes only): the K steps are cut into
 * `ksplit` equal ranges and the result is the left fold ((p_0 + p_1) + ...) of the per-range sums -- a constant of the
 * layer, so results do not depend on the batch.  The net builders choose it from the layer geometry (ResNet-50 stages
 * 3-4, net/resnet_v2.py:142-184; the RPN conv / head GEMMs of a single image); this entry sets it on a stand-alone layer.
 * ksplit 0 = back to the plain kernels, 1 = the split-K kernel with one range (bit-identical to the plain kernels).
 * mode 0 = ranges in parallel when tiles * ksplit <= 448 else one workgroup per tile, 1 / 2 = force either (the two are
 * bit-identical); max_parallel_tiles sizes the scratch slabs of the parallel mode. */
int xdet_conv_set_ksplit(void* layer, int ksplit, int mode, int max_parallel_tiles);
/* Split-precision operand planes (modes 1/2): x = hi + lo, both f16, blocked
 * [ceil(n_pix/16)][ld/32][16][32] with n_pix = N*H*W (16 pixels x 32 channels = one contiguous 1 KB
 * block: what one LDS-DMA instruction of a conv tile ingests); a plane holds ceil(n_pix/16)*16*ld halves.  xdet_split_f32 (x NHWC [n_pix][ld], ld % 32 == 0) writes them element-wise (optionally through a ReLU); xdet_conv_forward_planes runs a
 * layer whose A operand already lives as planes (LDS-DMA kernel, no register staging).  Inside a
 * net the depthwise kernels and split passes produce the planes; these two entry points expose the
 * same kernels for tests. */
int xdet_split_f32(const float* in, uint16_t* hi, uint16_t* lo, int64_t n_pix, int ld, int relu, void* stream);
int xdet_conv_forward_planes(void* layer, const uint16_t* in_hi, const uint16_t* in_lo, int N, int H, int W,
                             int ld_in, float* out, int ld_out, const float* residual, void* stream);
/* The "x8" form of the planes for a pointwise (1x1, stride 1) layer of mode 1 (csrc/conv_params.h): the two cross terms of the
 * split-precision product, a_hi*w_lo + a_lo*w_hi (2^-11 of the result), are computed from fp8 (e4m3, OCP) copies of the
 * operands by one block-scaled MFMA per 32 input channels instead of four f16 MFMAs (the GEMM is 21-26 % faster, one
 * contraction accurate to ~1e-5 instead of ~5e-7 relative: profiles/NOTES_r04.md).  `lo8` has the size and blocking of a
 * `lo` plane; per (pixel, 32-channel block) it holds 32 bytes fp8(hi * 2^-x8_exp) followed by 32 bytes
 * fp8(lo * 2^(11 - x8_exp)).  x8_exp: a power-of-two scale of the tensor such that its largest magnitude * 2^-x8_exp lands in
 * (128, 256] (values beyond 448 * 2^x8_exp saturate in the fp8 copies only).  Inside a net (option "cross" = "fp8") the
 * depthwise kernels write this form and the calibration pass chooses the exponents; these two entries expose the kernels for
 * tests. */
int xdet_split_f32_x8(const float* in, uint16_t* hi, uint16_t* lo8, int64_t n_pix, int ld, int relu, int x8_exp, void* stream);
int xdet_conv_forward_planes_x8(void* layer, const uint16_t* in_hi, const uint16_t* in_lo8, int N, int H, int W, int ld_in,
                                float* out, int ld_out, const float* residual, int x8_exp, void* stream);
int xdet_layer_destroy(void* layer);
/* depthwise 3x3 SAME stride 1 (the depthwise half of tf.layers.separable_conv2d,
 * net/xception_body.py:224-231); dw_kernel_host f32 [3,3,C,1]; in/out NHWC with stride ld. */
int xdet_depthwise_create(void** layer, int C, int dilation, const float* dw_kernel_host);
int xdet_depthwise_forward(void* layer, const float* in, int N, int H, int W, int ld, float* out, int relu_in,
                           void* stream);
/* relu_separable_bn_block as ONE kernel (net/xception_body.py:220-234: (ReLU ->) depthwise 3x3 -> pointwise 1x1 ->
 * BN, "nothing in between"): the depthwise result stays on the CU instead of crossing HBM as split planes.
 * dw_layer from xdet_depthwise_create (dilation 1), pw_layer a 1x1 stride-1 layer from xdet_conv_create in a
 * split-precision mode (its scale/shift = the folded BN, relu_out as created); <= 256 input channels
 * (multiple of 32) and 128 or 256 outputs; in/out NHWC f32.  Bit-identical to xdet_depthwise_forward ->
 * xdet_split_f32 -> xdet_conv_forward_planes.  Inside a net the entry-flow blocks use it
 * (option "sepconv" = "fused" (default) | "split"). */
int xdet_sepconv_fused_forward(void* dw_layer, void* pw_layer, const float* in, int N, int H, int W, int ld_in, float* out,
                               int ld_out, int relu_in, void* stream);
/* tf.layers.conv2d(3x3, VALID, stride 1, use_bias=False) -> BN (-> ReLU) over 32 input channels and <= 64 outputs
 * (block1_conv2, net/xception_body.py:252-259) with the input tile staged ONCE in LDS: in_hi/in_lo are the split
 * planes of xdet_split_f32 (ld 32), `layer` a matching xdet_conv_create layer in a split-precision mode; out is
 * NHWC f32 [N][H-2][W-2][ld_out].  Bit-identical to xdet_conv_forward_planes on the same layer.  Inside a net:
 * option "conv3x3" = "patch" (default) | "gemm". */
int xdet_conv3x3_patch_forward(void* layer, const uint16_t* in_hi, const uint16_t* in_lo, int N, int H, int W, float* out,
                               int ld_out, void* stream);
/* One ResNet v2 bottleneck block with an identity shortcut (net/resnet_v2.py:142-184 with projection_shortcut = None,
 * strides = 1: shortcut = inputs; BN-ReLU -> conv 1x1 -> BN-ReLU -> conv 3x3 -> BN-ReLU -> conv 1x1; + shortcut) as ONE
 * kernel (csrc/resnet_bneck.hip): the block input is read once per use, the pre-activation and the two narrow
 * intermediates never leave the CU.  conv_a / conv_b / conv_c: layers from xdet_conv_create in mode 1 -- 1x1 Cin -> Cmid
 * with the folded BN that follows it and relu 1; 3x3 SAME stride 1 Cmid -> Cmid likewise; 1x1 Cmid -> Cin, relu 0
 * (Cin 256, Cmid 64: stage 1 of ResNet-50).  x: the block input, NHWC f32 [N][H][W][Cin]; pre_scale / pre_shift
 * (device, [Cin]): the block's first BN folded, i.e. the pre-activation is relu(x * pre_scale + pre_shift) (one fused
 * multiply-add per element, then hi = f16(.), lo = f16(. - hi): what xdet_split_f32 makes of that tensor);
 * out = x + branch.  With out_hi / out_lo given, relu(out * next_scale + next_shift) -- the NEXT block's pre-activation --
 * is also written as planes (xdet_split_f32 layout) for a successor that runs layer by layer.  Bit-identical to
 * xdet_split_f32 + three xdet_conv_forward_planes calls.  Inside xdet_resnet_*: the identity blocks of stage 1
 * (xdet_resnet_set_option "bneck" = "off": three launches per block). */
int xdet_resnet_bneck_forward(void* conv_a, void* conv_b, void* conv_c, const float* pre_scale, const float* pre_shift,
                              const float* x, int N, int H, int W, float* out, const float* next_scale,
                              const float* next_shift, uint16_t* out_hi, uint16_t* out_lo, void* stream);
/* The entry-flow tail "separable block -> max_pooling2d(3, 2, 'same') -> tf.add(residual)" (net/xception_body.py:268-286)
 * with the pool split between the two kernels: xdet_sepconv_fused_hpool_forward is xdet_sepconv_fused_forward whose
 * epilogue writes the 3-column / stride-2 maximum of every output row (out_hpooled: NHWC f32 [N][H][(W+1)/2][ld_out]),
 * xdet_maxpool_v3s2_add the 3-row / stride-2 maximum of that (+ residual) -> [N][(H+1)/2][(W+1)/2][ld].  Together
 * bit-identical to xdet_sepconv_fused_forward -> xdet_maxpool3x3s2_add; the full-resolution tensor crosses HBM at half
 * size.  Inside a net: option "pool" = "split" (default: the 237 x 237 block) | "whole" | "split_all". */
int xdet_sepconv_fused_hpool_forward(void* dw_layer, void* pw_layer, const float* in, int N, int H, int W, int ld_in,
                                     float* out_hpooled, int ld_out, int relu_in, void* stream);
int xdet_maxpool_v3s2_add(const float* in_hpooled, const float* residual, float* out, int N, int H, int Wo, int C, int ld,
                          void* stream);
/* tf.layers.max_pooling2d(3,2,'same') + tf.add(residual) (net/xception_body.py:281-286) */
int xdet_maxpool3x3s2_add(const float* in, const float* residual, float* out, int N, int H, int W, int C, int ld,
                          void* stream);
int xdet_nchw_to_nhwc4(const float* in_nchw, float* out_nhwc4, int N, int C, int H, int W, void* stream);
/* F1: light_head_preprocess_for_eval / _for_test (preprocessing/common_preprocessing.py:383-458), fused:
 * uint8 [H,W,3] (device) -> whitened f32, TF-legacy bilinear warp to out_size x out_size, CHW
 * (one image of the [N,3,S,S] network input); bbox_img is the constant [0,0,1,1]. */
int xdet_preprocess_eval(const uint8_t* image_hwc, int H, int W, float* out_chw, int out_size, void* stream);

/* ---- A4+A6: RPN glue + AnchorEncoder.decode_all_anchors ----------------------------------
 * (light_head_rfcn_eval.py:389-397; preprocessing/anchor_manipulator.py:641-669,698-757)
 *   rpn_out NHWC [N,Hh,Ww,ld]: cls logits at channels [cls_off, cls_off+2A), box deltas at
 *   [box_off, box_off+4A); anchors_yx [Hh*Ww,2] centres, anchors_hw [A,2] sizes.
 *   -> objectness [N,Hh*Ww*A], boxes [N,Hh*Ww*A,4] */
int xdet_rpn_decode(const float* rpn_out, int ld, int cls_off, int box_off, int N, int Hh, int Ww, int A,
                    const float* anchors_yx, const float* anchors_hw, float* objectness, float* boxes, void* stream);

/* ---- A7: get_proposals (net/xception_body.py:402-448) ------------------------------------
 *   objectness [N,n_anchor], boxes [N,n_anchor,4] -> rois [N,post_n,4]; workspace from
 *   xdet_proposals_workspace_bytes().  counts_out (may be NULL) i32 [N,4] device:
 *   {n_valid, n_candidates, n_kept_by_nms, 0}.  nms_thr >= 0; post_n <= 6144 (the kept list lives in LDS).
 *   The NMS is tf.image.non_max_suppression's greedy walk (visit in score order, IoU > thr against boxes kept before,
 *   stop at post_n) computed in panels of 256 / 512 candidates; with N <= 64 an image's panels are spread over a cluster
 *   of up to 16 workgroups that wait for each other inside the kernel (every per-call control word is cleared by the
 *   call itself; a cluster whose members never meet -- it cannot happen on an otherwise working GPU -- gives up after
 *   ~1 s and marks its image, whose detection scores then come out NaN through xdet_net_forward). */
size_t xdet_proposals_workspace_bytes(int N, int n_anchor, int pre_n, int post_n);
int xdet_get_proposals(const float* objectness, const float* boxes, int N, int n_anchor, int pre_n, int post_n,
                       float nms_thr, float min_size, void* workspace, float* rois, int* counts_out, void* stream);

/* ---- A11: AnchorEncoder.ext_decode_rois (anchor_manipulator.py:671-683) ------------------ */
int xdet_ext_decode_rois(const float* rois, const float* reg, int ld_reg, int64_t n, float* out, void* stream);

/* ---- A12: bboxes_eval detection part (light_head_rfcn_eval.py:263-287) -------------------
 *   cls logits [N,R,ld_cls], boxes [N,R,4], image_shapes i32 [N,2] (H,W of the raw image),
 *   bbox_img f32 [N,4] -> det_scores [N,num_classes-1,nms_topk], det_boxes [..,4], zero padded. */
int xdet_bboxes_eval(const float* cls, int ld_cls, const float* boxes, int N, int R, int num_classes,
                     const int* image_shapes, const float* bbox_img, int net_h, int net_w, float select_thr,
                     float nms_thr, int nms_topk, float* det_scores, float* det_boxes, void* stream);

/* ---- the model: lighr_head_model_fn in eval mode (light_head_rfcn_eval.py:364-433) -------
 * Weights enter by TF variable name (scope prefix stripped), TF layouts (HWIO / [in,out]). */
typedef struct {
  int image_size;         /* train_image_size, 480 */
  int max_batch;          /* workspace is sized for this many images per call */
  int num_classes;        /* 21 */
  int num_anchors;        /* 22 */
  int rpn_pre_nms_top_n;  /* 5000 */
  int rpn_post_nms_top_n; /* 1000 (300 in BASELINE config 3) */
  float rpn_nms_thres;    /* 0.7 */
  float rpn_min_size;     /* 16/480 */
  float select_threshold; /* 0.01 */
  float nms_threshold;    /* 0.3 */
  int nms_topk;           /* 200 */
  int grid;               /* 7 */
  int bank;               /* 10 */
} xdet_lighthead_config;

int xdet_net_create(void** net, const xdet_lighthead_config* cfg);
int xdet_net_set_weight(void* net, const char* name, const float* data_host, int ndim, const int64_t* dims);
/* options, before xdet_net_build:
 *   "large_sep" = "auto" | "direct" | "spectral": arithmetic form of large_sep_kernel (net/xception_body.py:450-475).
 *   direct = the (15,1)/(1,15) convs as implicit GEMMs; spectral = the same linear maps evaluated in the DFT
 *   domain of the convolved axis (one GEMM per frequency bin, ~5x fewer MFMA FLOPs; needs a split-precision
 *   mode and a 16/30/50 feature map); auto = spectral whenever those hold, else direct.
 *   The choice is per net, never per call: results do not depend on the batch an image arrives in.
 *   "sepconv" = "fused" | "split": entry-flow separable blocks as one kernel (default) or depthwise + pointwise.
 *   "rpn_stream" = "side" | "main": the RPN / proposal branch forks onto a side stream under the large-separable
 *   convs (default) or stays on the caller's stream (per-kernel profiles without cross-stream sharing).
 *   "conv3x3" = "patch" | "gemm": block1_conv2 on the staged-tile kernel (default) or the implicit-GEMM kernel.
 *   "pool" = "split" | "whole" | "split_all": the horizontal half of the block2 / block3 max-pools in the producing
 *   block's epilogue (default) or the whole pool as its own kernel.
 *   "ksplit" = "on" | "off" | "all": the 2048 -> 25 head GEMM (a single image: 3 tiles against 64 K steps) on the fixed
 *   split-K kernel (on, the default; a layer constant: results identical at every batch size), nothing (off), or the RPN
 *   3x3 conv as well (all: 32 tiles against 207 K steps; faster for one image, 0.9 % slower at bench-size batches).
 *   "workspace" = "reuse" | "ssa" | "poison": see xdet_net_memory.
 *   "pool_sub" = "on" | "off": the vertical pool pass of blocks 2-3 also writes the split planes of the raw subsampled sum (the
 *   next block's 1x1 / stride-2 projection reads those) and stores the sum as relu(sum) (its first separable conv reads that)
 *   -- on (default) -- or leaves both to their own passes (off; the same values either way).
 *   "check_range" = "off" | "on": after each forward validate everything that is turned into f16 against the f16 range --
 *   every split plane (no inf / NaN in the hi plane; the planes hold x * 2^-e after xdet_net_calibrate), the f32 input
 *   of a register-split conv (|x| <= 65504) and of a fused separable block (relu?(x) * sum|taps| * 2^-e <= 65504) -- and
 *   every other activation tensor for NaN / inf; a violation marks the image's detection scores NaN (slot 0 of every
 *   class), as a non-finite RPN score or head logit always does.  A diagnostic for new checkpoints: the pass re-reads
 *   all activations. */
int xdet_net_set_option(void* net, const char* key, const char* value);
int xdet_net_build(void* net);     /* folds BN, transposes/pads weights, allocates the workspace */
int xdet_net_destroy(void* net);
/* named workspace buffers (views, owned by the net): "mid","out","rpn_out","feat","objectness",
 * "rpn_boxes","proposals","pooled","fc","cls_reg","head_boxes","prop_counts" */
int xdet_net_buffer(void* net, const char* name, void** dptr, int64_t dims[4], int* ld);
/* stage entry points = the reference's graph-builder functions (net/xception_body.py) */
int xdet_net_xception_body(void* net, const float* images_nchw, int N, void* stream);  /* :236 -> "mid","out" */
int xdet_net_get_rpn(void* net, int N, void* stream);                                  /* :381 -> "rpn_out" */
int xdet_net_large_sep(void* net, int N, void* stream);                                /* :450 -> "feat" */
int xdet_net_rpn_decode(void* net, int N, void* stream);                               /* -> "objectness","rpn_boxes" */
int xdet_net_get_proposals(void* net, int N, void* stream);                            /* :402 -> "proposals" */
int xdet_net_get_head(void* net, int N, void* stream);                                 /* :477 -> "cls_reg" */
int xdet_net_head_decode(void* net, int N, void* stream);                              /* -> "head_boxes" */
int xdet_net_bboxes_eval(void* net, int N, const int* image_shapes, const float* bbox_img, float* det_scores,
                         float* det_boxes, void* stream);
/* whole forward: images f32 [N,3,S,S] -> det_scores [N,20,topk], det_boxes [N,20,topk,4].
 * image_shapes / bbox_img may be NULL (S x S, [0,0,1,1]).  use_graph != 0 replays a hipGraph
 * of the whole forward.  A graph bakes in every pointer it was captured with, so graphs are cached
 * per argument tuple (N, images, image_shapes, bbox_img, det_scores, det_boxes): the first call with a
 * new tuple captures (and runs) a new graph, later calls with the same tuple replay it; at most 8
 * graphs are kept per net (oldest evicted).  The CONTENTS of the buffers may change between replays,
 * the buffers themselves must stay allocated while their graph is cached.
 * xdet_net_graph_count: number of graphs currently cached (tests). */
int xdet_net_forward(void* net, const float* images_nchw, int N, const int* image_shapes, const float* bbox_img,
                     float* det_scores, float* det_boxes, int use_graph, void* stream);
/* Activation pre-scale of the split-precision operands (modes 1 / 2).  An f16 hi part overflows beyond 65504 while the
 * reference computes in f32 everywhere and has BN-less edges (net/xception_body.py:381-400,450-475).  Every tensor that
 * is split into f16 planes carries a power-of-two exponent e: the planes hold x * 2^-e and the consuming contraction
 * folds 2^e back into its epilogue scale -- both exact.  xdet_net_calibrate runs the forward on `images` (device,
 * [N,3,S,S]) until no operand exceeds 4096 (16x headroom) and reports how many tensors got e != 0; a net whose
 * activations are small keeps every exponent at 0 and its results bit for bit.  Cached graphs are dropped.
 * xdet_net_plane_scales / _name: the exponents and what they belong to (exps may be NULL to query the count). */
int xdet_net_calibrate(void* net, const float* images_nchw, int N, int* n_scaled, void* stream);
int xdet_net_plane_scales(void* net, int max_n, int* n_out, int* exps);
int xdet_net_plane_scale_name(void* net, int idx, char* buf, int buflen);
/* option "cross" = "fp8": how many split-precision tensors are in the x8 form (fp8 copies for the cross terms; see
 * xdet_conv_forward_planes_x8) -- 0 until xdet_net_calibrate has measured them, then the 28 depthwise -> pointwise edges of
 * the light-head net (net/xception_body.py:220-234, blocks 5-14). */
int xdet_net_x8_planes(void* net, int* n_on);
int xdet_net_graph_count(void* net, int* count);
/* Device memory of the net's workspace and weights in bytes (everything the plan allocated), and how many bytes of
 * tensors were placed into blocks recycled from dead tensors (option "workspace" = "reuse", the default: a builder hands a
 * tensor's block back once its last consumer is planned and a later tensor takes it over -- a planes tensor only a block
 * of exactly its own size, an f32 tensor the best-fitting free block of at most twice its size (NOT re-zeroed: what lies
 * behind the tensor, its 128-float loader slack included, is the predecessor's bytes; no consumer may use them, which
 * "poison" proves) -- the middle flow's 24 x 2 tensors live in a handful of blocks; "ssa": one block per tensor, which
 * option check_range selects by itself because its validation pass reads every tensor after the forward; "poison"
 * (tests): "reuse", with everything behind a recycled f32 tensor filled with NaN bits in front of its producer on every
 * forward -- the detections must still be those of "ssa", bit for bit). */
int xdet_net_memory(void* net, size_t* allocated_bytes, size_t* recycled_bytes);
/* per-kernel accounting of the last build: total dense FLOPs (2*MAC, unpadded) of one image */
int xdet_net_flops_per_image(void* net, double* backbone, double* rpn, double* large_sep, double* head);

/* ---- per-op HIP-event timing (bench.py's roofline leg; the reference's counterpart is the
 * commented-out tf.train.ProfilerHook, light_head_rfcn_train.py:463,522) ---------------------
 * net_kind 0 = light-head net, 1 = resnet trunk.  While enabled every planned op (conv,
 * depthwise, pool ...) is bracketed by an event pair on the launch stream; xdet_profile_read
 * waits for them and returns per-op totals since the last read: ms, launch count and the
 * algorithmic dense FLOPs of one image (0 for non-MFMA ops). */
int xdet_profile_enable(void* net, int net_kind, int enable);
int xdet_profile_read(void* net, int net_kind, int max_ops, int* n_ops, double* ms, int* launches, double* flops);
int xdet_profile_op_name(void* net, int net_kind, int op, char* buf, int buflen);
/* per planned op: FLOPs it EXECUTES on the matrix cores per image (every split-precision product counted; less than
 * 3 x algorithmic for the spectral GEMMs, 0 for VALU ops).  A negative `flops` from xdet_profile_read marks an
 * auxiliary pass (DFT) of the contraction in front of / behind it. */
int xdet_profile_mfma_flops(void* net, int net_kind, int max_ops, int* n_ops, double* issued_flops_per_image);

/* ---- A13: ResNet-50 v2 trunk (net/resnet_v2.py:311-345), BASELINE config 2 --------------- */
int xdet_resnet_create(void** net, int image_size, int max_batch);
int xdet_resnet_set_weight(void* net, const char* name, const float* data_host, int ndim, const int64_t* dims);
/* Before xdet_resnet_build; every key takes "on" (default) | "off".  The off forms are the layer-by-layer plans the fused
 * kernels are tested against (tests/test_gpu_resnet_bneck.py: same bits, or f32 rounding for "projcat"):
 *   "bneck"     stage 1's identity blocks as one kernel (csrc/resnet_bneck.hip) | three launches per block
 *   "preconv"   a block's opening 1x1 conv makes its own pre-activation (csrc/resnet_preconv.hip) | a split pass + the conv
 *   "projcat"   a projection block's shortcut folded into its closing GEMM as extra K | projection GEMM + add
 *   "stem7"     the 7x7 / stride-2 stem conv from the NCHW image (csrc/resnet_stem.hip) | the implicit-GEMM kernel
 *   "stem_pool" the stem's max-pool writes the first block's pre-activation planes | pool, then a split pass
 *   "ksplit"    fixed split-K / one-range ring kernels for stages 2-4 at small batches | the plain kernels */
int xdet_resnet_set_option(void* net, const char* key, const char* value);
int xdet_resnet_build(void* net);
int xdet_resnet_forward(void* net, const float* images_nchw, int N, float* out_nhwc, void* stream);
/* the same forward as a replayed hipGraph (captured on the first call with a given (N, images, out) tuple; needs an
 * explicit stream; falls back to the eager form while per-op profiling is enabled) */
int xdet_resnet_forward_graph(void* net, const float* images_nchw, int N, float* out_nhwc, void* stream);
/* activation pre-scale of the trunk's split-precision operands, as xdet_net_calibrate (xdet_net_plane_scales / _name accept
 * a trunk handle too).  Handles are checked: an entry point given the other net type's handle returns XDET_ERR_INVALID_ARG. */
int xdet_resnet_calibrate(void* net, const float* images_nchw, int N, int* n_scaled, void* stream);
int xdet_resnet_out_shape(void* net, int* Ho, int* Wo, int* C);
int xdet_resnet_flops_per_image(void* net, double* flops);
int xdet_resnet_destroy(void* net);

/* ---- (e) multi-GPU: image-sharded ranks + ONE RCCL all-gather of detections per step ------
 * SURVEY.md 8e.  The reference has nothing to replace here (one tf.estimator session at batch 1,
 * light_head_rfcn_eval.py:212,466-499); BASELINE config 4 asks for this layer.  One process per GPU.
 * xdet_comm_init: call after xdet_set_device.  Rank 0 creates the ncclUniqueId and publishes it at
 *   unique_id_path (temporary name + rename); the other ranks poll for that file for up to timeout_s
 *   seconds (<= 0: 120), then every rank enters ncclCommInitRank.  world == 1 needs no path.
 *   librccl.so is bound by dlopen here, not at library load.
 * xdet_comm_allgather_detections: det_scores [B,C,K] + det_boxes [B,C,K,4] of this rank are packed
 *   into packed_local [B,C,K,5] (score | ymin xmin ymax xmax) and all-gathered into gathered
 *   [world*B,C,K,5] (rank-major).  Runs on the communicator's own stream: it first waits for the
 *   n_producers streams that write the det buffers, and makes those streams wait for the pack, so the
 *   caller may enqueue the next forward immediately -- the gather overlaps it.  No host sync.
 *   det_double_buffered != 0: the caller alternates between two det buffer pairs from call to call;
 *   the producers then only wait for the PREVIOUS call's pack (the one that read the pair they write
 *   next), which removes the once-per-step join of the producer streams.
 * xdet_comm_wait: stream != NULL -> that stream waits for the last gather; NULL -> the host does.
 * xdet_comm_allreduce_max / xdet_comm_barrier: scalar collectives for bench timing (host-synchronous).
 * xdet_comm_allgather_bytes: `bytes` (<= 1 MiB) of host memory from every rank -> world * bytes in rank order,
 *   moved by ncclAllGather on the communicator's stream (rank / device / PCI-bus-id records, per-rank rates).
 * Watchdog: every HOST wait on the communicator's stream (comm_wait(NULL), the scalar collectives, allgather_bytes,
 *   destroy) polls instead of blocking; when RCCL reports an asynchronous error or nothing completes for the
 *   timeout (XDET_COMM_TIMEOUT_S or xdet_comm_set_timeout, default 300 s: the length of ONE wait, so a caller whose
 *   peers legitimately take longer between two collectives raises it) the communicator is aborted
 *   (ncclCommAbort) and the call returns XDET_ERR_STATE, as does every later call -- a rank whose peer died exits
 *   with an error instead of hanging in hipStreamSynchronize.  Host buffers of the scalar / byte collectives are
 *   staged through pinned memory owned by the communicator, so no copy can block the host outside the watchdog.
 * Rank 0 removes a stale id file before it creates the id and removes its own once ncclCommInitRank has returned. */
int xdet_comm_init(void** comm, int rank, int world, const char* unique_id_path, int timeout_s);
int xdet_comm_destroy(void* comm);
int xdet_comm_info(void* comm, int* rank, int* world, int* device, int* rccl_version);
int xdet_pack_detections(const float* det_scores, const float* det_boxes, int64_t n_slots, float* packed, void* stream);
int xdet_comm_allgather_detections(void* comm, const float* det_scores, const float* det_boxes, int n_images,
                                   int n_fg_classes, int topk, float* packed_local, float* gathered,
                                   void* const* producer_streams, int n_producers, int det_double_buffered);
int xdet_comm_wait(void* comm, void* stream);
int xdet_comm_allreduce_max(void* comm, double* value_host);
int xdet_comm_barrier(void* comm);
int xdet_comm_allgather_bytes(void* comm, const void* send_host, void* recv_host, size_t bytes);
int xdet_comm_set_timeout(void* comm, double seconds);
/* which shared object the collective entry points were bound from (realpath of dladdr(ncclAllGather)), and whether it
 * was named by XDET_RCCL_LIB.  XDET_RCCL_LIB is honoured only together with XDET_ALLOW_RCCL_OVERRIDE=1 (the test double of
 * tests/fake_rccl, another RCCL build); otherwise xdet_comm_init fails with XDET_ERR_STATE: a bench never runs silently
 * on a stand-in.  Binds the library if no communicator exists yet. */
int xdet_comm_library(char* path_buf, int buflen, int* overridden);

#ifdef __cplusplus
}
#endif
#endif /* XDET_H_ */
