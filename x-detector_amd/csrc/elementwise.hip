// HBM-bound window / element-wise kernels of the backbone (NHWC, 16 B per lane):
//   * NCHW image -> NHWC4 (network entry, light_head_rfcn_eval.py:85 feeds channels_first)
//   * depthwise 3x3 (dilation 1|2, SAME) with optional leading ReLU: the depthwise half of
//     tf.layers.separable_conv2d (net/xception_body.py:220-234,268,354,366)
//   * max-pool 3x3/2 SAME + residual add (net/xception_body.py:281-286,302-307,321-326)
#include "common.h"

namespace xdet {

__global__ void nchw_to_nhwc4_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C, int H,
                                     int W, int ldo) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / ((int64_t)H * W);
    const int64_t px = i - n * H * W;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C && c < 4; ++c) v[c] = in[(n * C + c) * H * W + px];
    *reinterpret_cast<float4*>(out + i * ldo) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

int launch_nchw_to_nhwc4(const float* in, float* out, int N, int C, int H, int W, int ldo, hipStream_t s) {
  XDET_REQUIRE(C <= 4 && ldo == 4, "nchw_to_nhwc4: at most 4 channels");
  const int64_t total = (int64_t)N * H * W;
  const int blocks = (int)std::min<int64_t>(cdiv(total, 256), 4096);
  hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(blocks), dim3(256), 0, s, in, out, N, C, H, W, ldo);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// one thread = 4 channels of one output pixel; w9c is [9][ld] (tap-major, zero in padded channels)
__global__ void depthwise3x3_kernel(const float* __restrict__ in, const float* __restrict__ w9c,
                                    float* __restrict__ out, int N, int H, int W, int ld, int dil, int relu_in) {
  const int c4n = ld >> 2;
  const int64_t total = (int64_t)N * H * W * c4n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    const int64_t px = i / c4n;
    const int x = (int)(px % W);
    const int y = (int)((px / W) % H);
    const int64_t nb = (px / ((int64_t)W * H)) * H * W;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y + (ky - 1) * dil;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = x + (kx - 1) * dil;
        if ((unsigned)ix >= (unsigned)W) continue;
        float4 v = *reinterpret_cast<const float4*>(in + (nb + (int64_t)iy * W + ix) * ld + c4 * 4);
        if (relu_in) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        const float4 w = *reinterpret_cast<const float4*>(w9c + (ky * 3 + kx) * ld + c4 * 4);
        acc.x = fmaf(v.x, w.x, acc.x); acc.y = fmaf(v.y, w.y, acc.y);
        acc.z = fmaf(v.z, w.z, acc.z); acc.w = fmaf(v.w, w.w, acc.w);
      }
    }
    *reinterpret_cast<float4*>(out + px * ld + c4 * 4) = acc;
  }
}

int launch_depthwise3x3(const float* in, const float* w9c, float* out, int N, int H, int W, int C, int ld, int dil,
                        int relu_in, hipStream_t s) {
  XDET_REQUIRE(ld % 4 == 0 && ld >= C, "depthwise: channel stride must be a multiple of 4");
  const int64_t total = (int64_t)N * H * W * (ld / 4);
  const int blocks = (int)std::min<int64_t>(cdiv(total, 256), 256 * 32);
  hipLaunchKernelGGL(depthwise3x3_kernel, dim3(blocks), dim3(256), 0, s, in, w9c, out, N, H, W, ld, dil, relu_in);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// ---- split-precision planes: x = hi + lo, both f16 (the A operand format of conv_mfma_dma.hip) ----
typedef _Float16 f16x4e __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_store(const float4 v, unsigned short* hi, unsigned short* lo, int64_t o) {
  const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y, h2 = (_Float16)v.z, h3 = (_Float16)v.w;
  f16x4e hv = {h0, h1, h2, h3};
  f16x4e lv = {(_Float16)(v.x - (float)h0), (_Float16)(v.y - (float)h1), (_Float16)(v.z - (float)h2),
               (_Float16)(v.w - (float)h3)};
  *reinterpret_cast<uint2*>(hi + o) = *reinterpret_cast<uint2*>(&hv);
  *reinterpret_cast<uint2*>(lo + o) = *reinterpret_cast<uint2*>(&lv);
}

__global__ void split_f32_kernel(const float4* __restrict__ in, unsigned short* __restrict__ hi,
                                 unsigned short* __restrict__ lo, int64_t n4, int relu) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = in[i];
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    split_store(v, hi, lo, i * 4);
  }
}

int launch_split_f32(const float* in, unsigned short* hi, unsigned short* lo, int64_t n, int relu, hipStream_t s) {
  XDET_REQUIRE(n % 4 == 0, "split: length must be a multiple of 4");
  if (n == 0) return XDET_OK;
  const int blocks = (int)std::min<int64_t>(cdiv(n / 4, 256), 256 * 32);
  hipLaunchKernelGGL(split_f32_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(in), hi, lo,
                     n / 4, relu);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// depthwise 3x3 whose only consumer is a pointwise conv on the split path: emit the planes directly
__global__ void depthwise3x3_split_kernel(const float* __restrict__ in, const float* __restrict__ w9c,
                                          unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, int N,
                                          int H, int W, int ld, int dil, int relu_in) {
  const int c4n = ld >> 2;
  const int64_t total = (int64_t)N * H * W * c4n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    const int64_t px = i / c4n;
    const int x = (int)(px % W);
    const int y = (int)((px / W) % H);
    const int64_t nb = (px / ((int64_t)W * H)) * H * W;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y + (ky - 1) * dil;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = x + (kx - 1) * dil;
        if ((unsigned)ix >= (unsigned)W) continue;
        float4 v = *reinterpret_cast<const float4*>(in + (nb + (int64_t)iy * W + ix) * ld + c4 * 4);
        if (relu_in) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        const float4 w = *reinterpret_cast<const float4*>(w9c + (ky * 3 + kx) * ld + c4 * 4);
        acc.x = fmaf(v.x, w.x, acc.x); acc.y = fmaf(v.y, w.y, acc.y);
        acc.z = fmaf(v.z, w.z, acc.z); acc.w = fmaf(v.w, w.w, acc.w);
      }
    }
    split_store(acc, hi, lo, px * ld + c4 * 4);
  }
}

int launch_depthwise3x3_split(const float* in, const float* w9c, unsigned short* hi, unsigned short* lo, int N,
                              int H, int W, int C, int ld, int dil, int relu_in, hipStream_t s) {
  XDET_REQUIRE(ld % 4 == 0 && ld >= C, "depthwise: channel stride must be a multiple of 4");
  const int64_t total = (int64_t)N * H * W * (ld / 4);
  const int blocks = (int)std::min<int64_t>(cdiv(total, 256), 256 * 32);
  hipLaunchKernelGGL(depthwise3x3_split_kernel, dim3(blocks), dim3(256), 0, s, in, w9c, hi, lo, N, H, W, ld, dil,
                     relu_in);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

__global__ void maxpool3x3s2_add_kernel(const float* __restrict__ in, const float* __restrict__ res,
                                        float* __restrict__ out, int N, int H, int W, int ld, int Ho, int Wo,
                                        int pad_t, int pad_l) {
  const int c4n = ld >> 2;
  const int64_t total = (int64_t)N * Ho * Wo * c4n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    const int64_t px = i / c4n;
    const int ox = (int)(px % Wo);
    const int oy = (int)((px / Wo) % Ho);
    const int64_t n = px / ((int64_t)Wo * Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - pad_t + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - pad_l + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const float4 v = *reinterpret_cast<const float4*>(in + ((n * H + iy) * W + ix) * ld + c4 * 4);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    if (res) {
      const float4 r = *reinterpret_cast<const float4*>(res + px * ld + c4 * 4);
      m.x += r.x; m.y += r.y; m.z += r.z; m.w += r.w;
    }
    *reinterpret_cast<float4*>(out + px * ld + c4 * 4) = m;
  }
}

int launch_maxpool3x3s2_add(const float* in, const float* res, float* out, int N, int H, int W, int C, int ld,
                            int Ho, int Wo, int pad_t, int pad_l, hipStream_t s) {
  XDET_REQUIRE(ld % 4 == 0 && ld >= C, "maxpool: channel stride must be a multiple of 4");
  const int64_t total = (int64_t)N * Ho * Wo * (ld / 4);
  const int blocks = (int)std::min<int64_t>(cdiv(total, 256), 256 * 32);
  hipLaunchKernelGGL(maxpool3x3s2_add_kernel, dim3(blocks), dim3(256), 0, s, in, res, out, N, H, W, ld, Ho, Wo,
                     pad_t, pad_l);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

__global__ void relu_copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = in[i];
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    out[i] = v;
  }
}

int launch_relu_copy(const float* in, float* out, int64_t n, hipStream_t s) {
  XDET_REQUIRE(n % 4 == 0, "relu_copy: length must be a multiple of 4");
  const int blocks = (int)std::min<int64_t>(cdiv(n / 4, 256), 256 * 16);
  hipLaunchKernelGGL(relu_copy_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(in),
                     reinterpret_cast<float4*>(out), n / 4);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

}  // namespace xdet
