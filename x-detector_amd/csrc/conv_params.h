// Launch parameters of the MFMA convolution kernels (conv_mfma*.hip).  Its own header so that the profile
// bookkeeping (bench.py CONV_KERNEL_FILES) can tell a change of the conv kernels' interface from any other edit of common.h.
#pragma once

namespace xdet {

struct ConvParams {
  const float* in;   // NHWC, channel stride ldi (padded channels are zero)
  const float* wt;   // [Cout_pad][Kp], K contiguous, k = tap*Cin_p + ci   (f32 path)
  const unsigned short* wt_hi;   // f16 planes of the (per-channel power-of-two scaled) matrix: split path
  const unsigned short* wt_lo;
  // A operand already split into f16 planes (conv_mfma_dma.hip), blocked [pixels/16][ldi/32][16][32]
  // (pixel = n*H*W + y*W + x): 16 pixels x 32 channels are one contiguous 1 KB block
  const unsigned short* in_hi;
  const unsigned short* in_lo;
  const unsigned short* zeros;   // >= 16 B of zeros: the source of out-of-image taps for the LDS DMA
  float* out;        // NHWC, channel stride ldo; may be NULL when only the planes below are wanted
  // optional second copy of the output as split planes [pix/16][ldo/32][16][32] for a consumer on the
  // LDS-DMA path (saves its split pass); planes_relu: the planes hold max(out, 0) (a `relu -> conv` edge)
  unsigned short* out_hi;
  unsigned short* out_lo;
  int planes_relu;
  // channel blocks per 16-pixel group of the planes destination when that is WIDER than this conv's output: the conv fills
  // blocks [0, ldo/32) of a concatenated operand [pix/16][pl_c32][16][32] whose other blocks another producer writes (ResNet:
  // the closing 1x1 of a projection block reads [3x3 output | block input] against [w_c ; w_proj]); 0 = ldo / 32
  int pl_c32 = 0;
  // optional per-channel affine applied to the planes copy before its ReLU (a following inference BN:
  // planes = relu(out * pl_scale + pl_shift)); NULL = none
  const float* pl_scale;
  const float* pl_shift;
  const float* scale;   // [Cout_pad] folded BN scale (1 for plain bias)
  const float* shift;   // [Cout_pad] folded BN shift / bias
  const float* res;     // optional residual, same N,Ho,Wo, channel stride ldr
  int N, H, W, ldi;
  int Ho, Wo, ldo, ldr;
  int Cin_p;         // channels walked per tap (multiple of 32; 4 in small-cin mode)
  int Kp;            // padded reduction length (multiple of 32)
  int Cout_pad;      // multiple of the N tile
  int KH, KW, stride, dil, pad_t, pad_l;
  int M;             // N*Ho*Wo
  int relu_in, relu_out;
  // grouped GEMM (conv_mfma_dma.hip only; the frequency bins of the spectral large-separable conv): rows
  // [g*group_rows, (g+1)*group_rows) of the M dimension use weight matrix g (wt_* + g*group_wt_stride halves)
  // and scale/shift row g (+ g*Cout_pad).  group_rows is a multiple of every M tile; 0 = one group.
  int group_rows;
  long long group_wt_stride;
  // rows [group_live_rows, group_rows) of every group are padding whose results nobody reads (a bin holds N * F real rows, padded
  // to whole M tiles): an M tile that lies wholly in them is not computed.  0 = every row is live.
  int group_live_rows = 0;
  // fixed split of the reduction (conv_mfma_ksplit.hip only): the K steps are cut into `ksplit` equal ranges and the
  // result is the left fold of the per-range sums; ks_partial = scratch slabs [tiles][ksplit][128 x N tile] f32 of
  // the mode that runs the ranges in parallel
  // "x8" form of a pointwise contraction on the LDS-DMA path (conv_mfma_dma.hip): the two cross terms of the split-precision
  // product (a_hi*w_lo + a_lo*w_hi, 2^-11 of the result) are computed from fp8 (e4m3) copies of the operands by ONE block-scaled
  // MFMA per 32 real K instead of four f16 MFMAs.  Then `in_lo` / `wt_lo` hold, per 32-channel block of a row, 32 bytes of
  // hi8 = fp8(hi * 2^-x8_exp) followed by 32 bytes of lo8 = fp8(lo * 2^(11 - x8_exp)) (weights: fp8(w_lo * 2^9) followed by
  // fp8(w_hi * 2^-2)) in the place of the 32 f16 `lo` values -- same bytes, same DMA.  x8_exp: the activation tensor's
  // power-of-two fp8 scale, chosen by the calibration pass so that its largest magnitude lands in (128, 256].
  int x8 = 0;
  int x8_exp = 0;
  int skip_dead = 1;   // conv_dma_f16_kernel: waves skip 32-column blocks that are pure padding of the N tile (0: A/B runs)
  int ksplit = 1;
  float* ks_partial = nullptr;
  // parallel mode with an in-kernel fold: ks_ticket[tile] counts the ranges of a tile that have parked their slab; the LAST one
  // to arrive folds all of them (in range order) and runs the epilogue, then puts the ticket back to 0.  >= 4096 ints, zero
  // between launches.  NULL: the fold is a second launch (conv_ksplit_fold_kernel).
  int* ks_ticket = nullptr;
};

}  // namespace xdet
