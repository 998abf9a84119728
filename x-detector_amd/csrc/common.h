// Shared host-side helpers for libxdet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/xdet.h"   // XDET_OK / XDET_ERR_* codes
#include "conv_params.h"   // struct ConvParams: what every MFMA conv kernel is launched with
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace xdet {

void set_last_error(const std::string& s);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define XDET_HIP(expr)                                                          \
  do {                                                                          \
    hipError_t _e = (expr);                                                     \
    if (_e != hipSuccess) return ::xdet::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

#define XDET_LAUNCH_CHECK() XDET_HIP(hipGetLastError())

#define XDET_REQUIRE(cond, msg)                                   \
  do {                                                            \
    if (!(cond)) {                                                \
      ::xdet::set_last_error(std::string("invalid argument: ") + (msg)); \
      return XDET_ERR_INVALID_ARG;                        \
    }                                                             \
  } while (0)

#define XDET_TRY(expr)            \
  do {                            \
    int _rc = (expr);             \
    if (_rc != 0) return _rc;     \
  } while (0)

// hipFuncSetAttribute is a per-DEVICE setting: apply it once per (kernel, device) -- one bit per device
// ordinal; two threads racing on the first launch just repeat an idempotent call.
struct DeviceOnce {
  std::atomic<unsigned long long> done{0};
};
static inline int ensure_dynamic_lds(DeviceOnce& once, const void* kern, int bytes) {
  int dev = 0;
  XDET_HIP(hipGetDevice(&dev));
  const unsigned long long bit = 1ull << (dev & 63);
  if (once.done.load(std::memory_order_acquire) & bit) return XDET_OK;
  XDET_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  once.done.fetch_or(bit, std::memory_order_release);
  return XDET_OK;
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- implicit-GEMM convolutions on the MFMA pipe (conv_mfma*.hip; struct ConvParams: conv_params.h) ----
int launch_conv_mfma_f32(const ConvParams& p, bool small_cin, int n_tile, hipStream_t s);
// nsplit 3 = f16x3 (hi/lo f16 operands, f32-class accuracy), 1 = plain f16 operands
int launch_conv_mfma_split(const ConvParams& p, bool small_cin, int n_tile, int nsplit, hipStream_t s);
int launch_conv_mfma_dma(const ConvParams& p, int n_tile, int nsplit, hipStream_t s);
// fixed split-K (conv_mfma_ksplit.hip): mode 0 = parallel ranges when the grid is small, else one workgroup per tile
// walking all ranges; 1 / 2 force either (tests).  scratch_tiles = tiles the layer's scratch was sized for.
bool conv_dma_fold_applicable(const ConvParams& p, int n_tile, int nsplit);   // conv_mfma_dma.hip: the fold in the 256 x 128 kernel
int launch_conv_mfma_dma_fold(const ConvParams& p, hipStream_t s);
int64_t conv_ksplit_tiles(int64_t M, int cout_pad, int n_tile);
bool conv_ksplit_supported(int kh, int kw, int64_t n_pix_in, int ld_in, int cin_p, int cout_pad);
int launch_conv_mfma_ksplit(const ConvParams& p, int n_tile, int nsplit, int mode, int64_t scratch_tiles, hipStream_t s);
enum { PREC_F32 = 0, PREC_F16X3 = 1, PREC_F16 = 2 };

// ---- element-wise / window kernels (elementwise.hip) ---------------------------------
int launch_nchw_to_nhwc4(const float* in, float* out, int N, int C, int H, int W, int ldo, hipStream_t s);
int launch_depthwise3x3(const float* in, const float* w9c, float* out, int N, int H, int W, int C, int ld,
                        int dil, int relu_in, hipStream_t s);
int launch_maxpool3x3s2_add(const float* in, const float* res, float* out, int N, int H, int W, int C, int ld,
                            int Ho, int Wo, int pad_t, int pad_l, hipStream_t s);
// pool -> bn + ReLU -> split planes only (the ResNet stem tail in front of the first block's pre-activation)
int launch_maxpool3x3s2_bn_planes(const float* in, const float* scale, const float* shift, unsigned short* hi, unsigned short* lo,
                                  int N, int H, int W, int C, int ld, int Ho, int Wo, int pad_t, int pad_l, float mul,
                                  hipStream_t s, unsigned short* hi2 = nullptr, unsigned short* lo2 = nullptr, int c32_2 = 0,
                                  float mul2 = 1.f);
// vertical half only, over rows the producer already pooled horizontally (sepconv_fused.hip HPOOL)
int launch_maxpool_v3s2_add(const float* in_hpooled, const float* res, float* out, int N, int H, int Wo, int C, int ld,
                            int Ho, int pad_t, hipStream_t s, unsigned short* sub_hi = nullptr, unsigned short* sub_lo = nullptr,
                            float sub_mul = 1.f);
// diagnostic: bad_per_image[n] = 1 if any element of image n is NaN or beyond +-limit
// groups > 1: a stack of `groups` blocks of group_elems floats / group_pix pixels (the frequency bins of the spectral
// large-separable convs); the image of an element is its position inside its block / per_image, rows past N are padding
// relu != 0: the tensor is consumed through a ReLU, only values above +limit count (NaN / inf always do)
int launch_range_check(const float* x, int N, size_t per_image, float limit, int* bad_per_image, hipStream_t s,
                       int groups = 1, size_t group_elems = 0, int relu = 0);
int launch_range_check_planes(const unsigned short* hi, int N, int64_t pix_per_image, int ld, int* bad_per_image,
                              hipStream_t s, int groups = 1, int64_t group_pix = 0);
int launch_relu_copy(const float* in, float* out, int64_t n, hipStream_t s);
int launch_stem_conv3x3s2(const float* in_nchw, const float* w27x32, const float* scale, const float* shift,
                          unsigned short* hi, unsigned short* lo, int N, int S, hipStream_t s);
// mul: the planes' activation pre-scale 2^-e (a power of two; 1 = none): hi + lo = x * mul
int launch_split_f32(const float* in, unsigned short* hi, unsigned short* lo, int64_t n_pix, int ld, int relu,
                     hipStream_t s, float mul = 1.f, int x8 = 0, int x8_exp = 0);
int launch_planes_copy_blocks(const unsigned short* shi, const unsigned short* slo, unsigned short* dhi, unsigned short* dlo,
                              int64_t n_pix, int ld_src, int c32_dst, float r, hipStream_t s);
int launch_split_f32_subsample2(const float* in, unsigned short* hi, unsigned short* lo, int N, int H, int W, int ld,
                                hipStream_t s, const float* scale = nullptr, const float* shift = nullptr, float mul = 1.f,
                                int c32_dst = 0);
// range calibration: largest |hi| of a plane (f16 bits) / largest |x| or max(x, 0) of an f32 tensor (f32 bits), atomicMax'ed into *out
int launch_absmax_planes(const unsigned short* hi, int64_t n_halves, unsigned* out, hipStream_t s);
int launch_absmax_f32(const float* x, int64_t n, int relu, unsigned* out, hipStream_t s);
// x8: the planes in the x8 form (conv_params.h) with the tensor's fp8 exponent x8_exp
int launch_depthwise3x3_split(const float* in, const float* w9c, unsigned short* hi, unsigned short* lo, int N,
                              int H, int W, int C, int ld, int dil, int relu_in, hipStream_t s, int x8 = 0, int x8_exp = 0);

// ---- spectral large-separable conv: DFT passes around the grouped GEMM (spectral.hip) ----
bool spectral_supported(int F);
int spectral_points(int F);
void spectral_tables(int F, std::vector<float>* fwd, std::vector<float>* inv);
void spectral_weights(const float* w, int T, int cin, int cout, int cin_ld, int cout_ld, int F, std::vector<float>* out);
int launch_dft_fwd(const float* in, int F, int ld, int axis, int N, int m_pad, const float* tab, unsigned short* hi,
                   unsigned short* lo, hipStream_t s);
int launch_dft_inv(const float* Y, int F, int ldn, int C_ld, int N, int m_pad, const float* tab, const float* scale,
                   const float* shift, int relu, float* out, int ldo, int axis, hipStream_t s);

// ---- fused separable block of the entry flow (sepconv_fused.hip) ----
bool sepconv_fused_supported(int cin_ld, int cout_pad, int dil);
int launch_sepconv_fused(const float* in, const float* w9c, const unsigned short* wt_hi_blocked,
                         const unsigned short* wt_lo_blocked, const float* scale, const float* shift, float* out, int N,
                         int H, int W, int ld, int ldo, int cout_pad, int relu_in, int relu_out, hipStream_t s,
                         int pool_pad_l = -1);

// ---- 3x3 / stride 1 / VALID conv over 32 channels with the input tile staged once in LDS (conv3x3_patch.hip) ----
bool conv3x3_patch_supported(int kh, int kw, int cin, int cout_pad, int stride, int dil, int pad_mode);
int launch_conv3x3_patch(const unsigned short* in_hi, const unsigned short* in_lo, const unsigned short* wt_hi_blocked,
                         const unsigned short* wt_lo_blocked, const float* scale, const float* shift, float* out, int N, int H,
                         int W, int ldo, int relu, hipStream_t s);

// ---- ResNet v2 stem conv 7x7 / stride 2 / 3 -> 64 from the NCHW input, patch staged in LDS (resnet_stem.hip) ----
bool resnet_stem7x7_supported(int kh, int kw, int cin, int cout, int stride, int pad_mode, int pad, int S);
int launch_resnet_stem7x7(const float* img_nchw, const unsigned short* wt_hi, const unsigned short* wt_lo, const float* scale,
                          const float* shift, float* out, int N, int S, hipStream_t s);

// ---- a bottleneck block's opening 1x1 conv with the pre-activation (bn + ReLU + split of the raw input) made on the CU
//      (resnet_preconv.hip): out = planes of relu(bn_b(conv1x1(relu(bn_a(x))))) ----
bool resnet_preconv_supported(int cin, int cmid, int64_t M);
int launch_resnet_preconv(const float* x, const float* pre_sc, const float* pre_sh, const unsigned short* w_hi,
                          const unsigned short* w_lo, const float* sc, const float* sh, unsigned short* out_hi,
                          unsigned short* out_lo, int64_t M, int cin, int cmid, hipStream_t s);

// ---- one ResNet v2 identity bottleneck block as a single kernel (resnet_bneck.hip) ----
struct BneckLaunch {
  const float* x;                          // the block input (identity shortcut and, through bn + ReLU, the first conv's operand)
  const float *pre_sc, *pre_sh;            // the block's pre-activation BN, folded: relu(x * pre_sc + pre_sh)
  const unsigned short *wa_hi, *wa_lo, *wb_hi, *wb_lo, *wc_hi, *wc_lo;   // K-blocked weights of the three convs
  const float *sc_a, *sh_a, *sc_b, *sh_b, *sc_c, *sh_c;                   // their epilogue scale / shift
  const float *pl_sc, *pl_sh;              // the planes copy of the output: relu(out * pl_sc + pl_sh)
  float* out;
  unsigned short *out_hi, *out_lo;         // (NULL: no planes copy)
  int H, W, cin, cmid, cout;
};
bool resnet_bneck_supported(int cin, int cmid, int cout, int H, int W, int N);
int launch_resnet_bneck(const BneckLaunch& a, int N, hipStream_t s);

// ---- F1 pre-processing (preprocess.hip) ----------------------------------------------
int launch_preprocess_eval(const unsigned char* img, int H, int W, float* out_chw, int S, hipStream_t s);

// ---- PsRoiAlign (psroialign.hip) -------------------------------------------------------
int launch_psroialign(const float* feat, const float* rois, float* pooled, int32_t* index, int N, int C, int H,
                      int W, int R, int gw, int gh, int use_max, int layout, int ldc, int out_ld,
                      int rois_are_corners, hipStream_t s);
int launch_psroialign_grad(const float* rois, const float* grad_pooled, const int32_t* pooled_index, float* grad_out,
                           int N, int C, int H, int W, int R, int gw, int gh, int use_max, int layout, int ldc,
                           hipStream_t s);

// ---- RPN tail / proposals (proposals.hip) ----------------------------------------------
struct ProposalWorkspace {
  // all per image, sized for n_anchor / pre_n
  unsigned long long* keys;   // [N][n_anchor]
  float* cboxes;              // [N][n_anchor][4] clipped boxes
  int* ranks;                 // [N][n_anchor]
  int* counts;                // [N][4]: n_valid, n_cand, n_keep, candidate-list length
  float* sboxes;              // [N][pre_n][4]
  float* sscores;             // [N][pre_n]
  unsigned long long* cand;   // [N][n_anchor] keys that can still reach the top pre_n
  int* hist;                  // [N][16384] histogram of the keys' top 16 bits
  int* bad;                   // [N] set when an objectness score / box of the image is not finite (read by bboxes_eval)
  int* tbin;                  // [N] threshold bin
  int* kept;                  // [N][post_n]
  unsigned* nms_ctl;          // [N][4] cluster barrier of the proposal NMS: arrival counter, give-up flag (zeroed with hist)
  unsigned long long* nms_sup;  // [N][ceil(pre_n / 256)][8] per panel: candidates suppressed by the kept list (zeroed with hist)
  unsigned long long* nms_col;  // [min(N, 64)][3][36][64] a cluster's exchange of the panel's own IoU bits
};
size_t proposal_workspace_bytes(int N, int n_anchor, int pre_n, int post_n);
void proposal_workspace_carve(void* base, int N, int n_anchor, int pre_n, int post_n, ProposalWorkspace* ws);
int launch_rpn_decode(const float* rpn_out, int ld, int cls_off, int box_off, int N, int Hh, int Ww, int A,
                      const float* anchors_yx, const float* anchors_hw, float* objectness, float* boxes,
                      hipStream_t s);
int launch_get_proposals(const float* objectness, const float* boxes, int N, int n_anchor, int pre_n, int post_n,
                         float nms_thr, float min_size, const ProposalWorkspace& ws, float* rois, hipStream_t s);

// ---- detection post-processing (detect.hip) --------------------------------------------
int launch_ext_decode_rois(const float* rois, const float* reg, int ld_reg, int64_t n, float* out, hipStream_t s);
int launch_bboxes_eval(const float* cls, int ld_cls, const float* boxes, int N, int R, int num_classes,
                       const int* image_shapes, const float* bbox_img, int net_h, int net_w, float select_thr,
                       float nms_thr, int nms_topk, float* det_scores, float* det_boxes, hipStream_t s,
                       const int* bad_per_image = nullptr);

// the whole forward's form: A11 + the head's class probabilities (class-major [N][num_classes][R]) in one pass over the ROIs,
// then A12 reading its class's column instead of redoing the softmax in each of the 20 class workgroups
int launch_head_decode_probs(const float* rois, const float* cls_reg, int ld, int num_classes, int R, int64_t n, float* out,
                             float* probs, int* bad, hipStream_t s);
int launch_bboxes_eval_probs(const float* probs, const float* boxes, int N, int R, int num_classes, const int* image_shapes,
                             const float* bbox_img, int net_h, int net_w, float select_thr, float nms_thr, int nms_topk,
                             float* det_scores, float* det_boxes, hipStream_t s, const int* bad_per_image = nullptr);

}  // namespace xdet
