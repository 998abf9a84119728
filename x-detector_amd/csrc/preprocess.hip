// F1: the step right before the hot path, fused into one kernel --
// light_head_preprocess_for_eval / _for_test (preprocessing/common_preprocessing.py:383-458):
//   uint8 HWC image -> tf.image.convert_image_dtype(float32) * 2 - [R,G,B mean]/127.5
//   -> tf.image.resize_images(BILINEAR, align_corners=False) warp to SxS (tf_image.py:307-319)
//   -> HWC -> CHW (data_format 'NCHW').
// TF1 legacy bilinear: src = dst * (in/out) (no half-pixel centre), lower = trunc(src),
// upper = min(lower+1, in-1), lerp = src - lower; top/bottom lerp in x, then lerp in y.
// Compiled with -ffp-contract=off so every product/sum rounds as TF's separate f32 ops do.
#include "common.h"

namespace xdet {

__device__ __forceinline__ float whiten(unsigned char u, float mean) {
  return ((float)u * (1.0f / 255.0f)) * 2.0f - mean;
}

__global__ void preprocess_eval_kernel(const unsigned char* __restrict__ img, int H, int W, float* __restrict__ out,
                                       int S, float hscale, float wscale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * S) return;
  const int oy = i / S, ox = i - oy * S;
  const float fy = (float)oy * hscale, fx = (float)ox * wscale;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float means[3] = {123.68f / 127.5f, 116.78f / 127.5f, 103.94f / 127.5f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float tl = whiten(img[((size_t)y0 * W + x0) * 3 + c], means[c]);
    const float tr = whiten(img[((size_t)y0 * W + x1) * 3 + c], means[c]);
    const float bl = whiten(img[((size_t)y1 * W + x0) * 3 + c], means[c]);
    const float br = whiten(img[((size_t)y1 * W + x1) * 3 + c], means[c]);
    const float top = tl + (tr - tl) * lx;
    const float bot = bl + (br - bl) * lx;
    out[(size_t)c * S * S + i] = top + (bot - top) * ly;
  }
}

int launch_preprocess_eval(const unsigned char* img, int H, int W, float* out_chw, int S, hipStream_t s) {
  XDET_REQUIRE(img && out_chw && H > 0 && W > 0 && S > 0, "preprocess: bad arguments");
  const float hscale = (float)H / (float)S, wscale = (float)W / (float)S;
  hipLaunchKernelGGL(preprocess_eval_kernel, dim3((unsigned)cdiv((int64_t)S * S, 256)), dim3(256), 0, s, img, H, W,
                     out_chw, S, hscale, wscale);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

}  // namespace xdet
