// ResNet v2 stem: conv2d_fixed_padding(7 x 7, 64 filters, stride 2) on the 3-channel image (net/resnet_v2.py:311-320,
// fixed_padding :41-50: explicit pad 3 / 3, then VALID) straight from the NCHW input, with the INPUT PATCH STAGED IN LDS.
//
// On the generic small-cin kernel (conv_mfma_split.hip) the layer is a K = 7 * 7 * 4 = 196 (-> 224) GEMM whose A operand
// is gathered from an NHWC4 copy of the image, one 16-byte load per (output pixel, tap): 93 us behind a 19 us layout pass
// at batch 8 where its output (118 MB) takes 26 us to write.  Here a persistent workgroup (8 waves) takes tiles of
// 8 x 32 output pixels:
//
//   patch   21 x 69 input pixels x {R, G, B, 0} f32 (24 KB) per tile, read from the three NCHW planes with row-contiguous
//           loads, double-buffered (the next tile's pixels travel in registers under this tile's MFMAs)
//   A       tap (ky, kx) of output pixel (oy, ox) = patch[2 oy + ky][2 ox + kx]: two ds_read_b128 per lane and 16-deep K
//           half (2 taps x 4 channels), split into f16 hi / lo in registers (the arithmetic of conv_mfma_split.hip's split4)
//   B       the whole filter, [64][224] f16 hi and lo, lives in LDS for the life of the workgroup (58 KB, row pitch 464 B:
//           16 consecutive rows of a ds_read_b128 lane group hit 16 different 4-bank windows)
//   84 x v_mfma_f32_32x32x16_f16 per wave and tile (wave = one row of 32 output pixels x 64 channels), epilogue straight
//   from the accumulators (a lane holds one channel of 16 pixels: each store instruction writes whole 128-byte lines)
//
// K order (tap * 4 + channel, ascending, taps 49..55 zero weights), product order (lo*hi, hi*lo, hi*hi per 16-deep half)
// and the epilogue's fma are those of conv_mfma_f16_kernel<128, 64, 4, 1, true, 3>: bit-identical (tests/test_gpu_resnet_bneck.py).
#include "common.h"
#include <algorithm>

namespace xdet {

typedef float rs_f32x16 __attribute__((ext_vector_type(16)));
typedef float rs_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 rs_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int RS_TR = 8, RS_TC = 32;                 // output rows x columns of a tile
constexpr int RS_PR = 2 * RS_TR + 5, RS_PC = 2 * RS_TC + 5;   // patch rows x columns (21 x 69)
constexpr int RS_PP = 72;                            // patch row pitch in pixels
constexpr int RS_PATCH_B = RS_PR * RS_PP * 16;       // bytes per patch buffer (24,192)
constexpr int RS_KP = 224, RS_WP = 464;              // padded K; bytes per filter row in LDS
constexpr int RS_W_B = 64 * RS_WP;                   // bytes per filter plane
constexpr int RS_NPIX = (RS_PR * RS_PC + 511) / 512; // patch pixels per thread (3)
constexpr int RS_LDS = 2 * RS_PATCH_B + 2 * RS_W_B;

typedef unsigned rs_u4 __attribute__((ext_vector_type(4)));

// (one asm block: gfx950 wants a wait state between a half-register write and a VALU read of the register, and inline asm is
//  opaque to the hazard recogniser -- see sf_split4)
__device__ __forceinline__ void rs_split4(const float4 a, uint2* h, uint2* l) {
  unsigned h0, h1, l0, l1;
  asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
      "v_cvt_pk_f16_f32 %1, %6, %7\n\t"
      "v_fma_mixlo_f16 %2, %4, 1.0, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %3, %6, 1.0, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %5, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %3, %7, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "s_nop 0"
      : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1)
      : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w));
  *h = make_uint2(h0, h1);
  *l = make_uint2(l0, l1);
}

struct ResnetStemParams {
  const float* img;          // NCHW f32 [N][3][S][S]
  const u16* wt_hi; const u16* wt_lo;   // [64][224] f16, k = tap * 4 + channel
  const float* scale; const float* shift;
  float* out;                // NHWC f32 [N][Ho][Wo][64]
  int N, S, Ho, Wo, TY, TX, ntiles;
};

__global__ __launch_bounds__(512) void resnet_stem7x7_kernel(ResnetStemParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fh = lane >> 5;
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, GW = gridDim.x >> 3;
  const int per_xcd = (p.ntiles + 7) >> 3;
  const int t_begin = xcd * per_xcd + wg;
  const int t_end = min(p.ntiles, (xcd + 1) * per_xcd);
  if (t_begin >= t_end) return;

  // ---- the filter: global [64][224] -> LDS rows of 464 bytes ----
  for (int i = tid; i < 64 * (RS_KP / 8); i += 512) {          // 16-byte chunks
    const int n = i / (RS_KP / 8), c = i - n * (RS_KP / 8);
    *reinterpret_cast<uint4*>(smem + 2 * RS_PATCH_B + n * RS_WP + c * 16) = *reinterpret_cast<const uint4*>(p.wt_hi + n * RS_KP + c * 8);
    *reinterpret_cast<uint4*>(smem + 2 * RS_PATCH_B + RS_W_B + n * RS_WP + c * 16) = *reinterpret_cast<const uint4*>(p.wt_lo + n * RS_KP + c * 8);
  }

  struct Coord { int n, ty, tx; };
  auto decode = [&](int q) {
    Coord c;
    c.tx = q % p.TX; q /= p.TX;
    c.ty = q % p.TY;
    c.n = q / p.TY;
    return c;
  };
  // patch pixel i of this thread: (row, column) = (i / 69, i % 69) for i = tid, tid + 512, tid + 1024
  int prow[RS_NPIX], pcol[RS_NPIX];
#pragma unroll
  for (int k = 0; k < RS_NPIX; ++k) {
    const int i = tid + 512 * k;
    prow[k] = i / RS_PC;
    pcol[k] = i - prow[k] * RS_PC;
  }
  float px[RS_NPIX][3];
  auto load_patch = [&](const Coord& c) {
    const int y0 = 2 * c.ty * RS_TR - 3, x0 = 2 * c.tx * RS_TC - 3;
    const float* base = p.img + (size_t)c.n * 3 * p.S * p.S;
#pragma unroll
    for (int k = 0; k < RS_NPIX; ++k) {
      const int y = y0 + prow[k], x = x0 + pcol[k];
      const bool ok = prow[k] < RS_PR && (unsigned)y < (unsigned)p.S && (unsigned)x < (unsigned)p.S;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) px[k][ch] = ok ? base[((size_t)ch * p.S + y) * p.S + x] : 0.f;
    }
  };
  auto store_patch = [&](int buf) {
#pragma unroll
    for (int k = 0; k < RS_NPIX; ++k)
      if (prow[k] < RS_PR)
        *reinterpret_cast<float4*>(smem + buf * RS_PATCH_B + (prow[k] * RS_PP + pcol[k]) * 16) = make_float4(px[k][0], px[k][1], px[k][2], 0.f);
  };

  // B fragment rows of this lane: channel j * 32 + frow
  const unsigned char* w_base = smem + 2 * RS_PATCH_B + frow * RS_WP + fh * 16;
  const float esc[2] = {p.scale[frow], p.scale[32 + frow]}, esh[2] = {p.shift[frow], p.shift[32 + frow]};

  Coord cur = decode(t_begin);
  load_patch(cur);
  store_patch(0);
  __syncthreads();
  int buf = 0;
  for (int t = t_begin; t < t_end; t += GW, buf ^= 1) {
    const bool more = t + GW < t_end;
    const Coord nxt = decode(min(t + GW, p.ntiles - 1));
    if (more) load_patch(nxt);                     // the next tile's pixels travel under this tile's MFMAs
    rs_f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // patch address of this lane's output pixel (row 2 wave, column 2 frow), taps added below
    const unsigned char* a_base = smem + buf * RS_PATCH_B + ((2 * wave) * RS_PP + 2 * frow) * 16;
#pragma unroll
    for (int kk = 0; kk < RS_KP / 16; ++kk) {      // 16-deep halves: taps 4 kk .. 4 kk + 3, this lane half's two: 4 kk + 2 fh, + 1
      // (taps past the 49th carry zero weights: their A values only have to be finite -- they read the last real tap's pixel)
      auto tap_off = [](int tap) {
        const int tt = tap < 49 ? tap : 48;
        return ((tt / 7) * RS_PP + (tt % 7)) * 16;
      };
      const unsigned o0 = fh ? (unsigned)tap_off(4 * kk + 2) : (unsigned)tap_off(4 * kk);
      const unsigned o1 = fh ? (unsigned)tap_off(4 * kk + 3) : (unsigned)tap_off(4 * kk + 1);
      const float4 v0 = *reinterpret_cast<const float4*>(a_base + o0);
      const float4 v1 = *reinterpret_cast<const float4*>(a_base + o1);
      // hi = f16(v), lo = f16(v - float(hi)): v_cvt_pk_f16_f32 + v_fma_mixlo / mixhi_f16 (sepconv_fused.hip sf_split4: the same bits
      // as convert / convert back / subtract / convert -- the difference has at most 13 significant bits -- in 6 instructions per 4 values)
      uint2 h0, l0, h1, l1;
      rs_split4(v0, &h0, &l0);
      rs_split4(v1, &h1, &l1);
      const rs_u4 hq = {h0.x, h0.y, h1.x, h1.y}, lq = {l0.x, l0.y, l1.x, l1.y};
      const rs_f16x8 ah = __builtin_bit_cast(rs_f16x8, hq), al = __builtin_bit_cast(rs_f16x8, lq);
      rs_f16x8 bh[2], bl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = *reinterpret_cast<const rs_f16x8*>(w_base + j * 32 * RS_WP + kk * 32);
        bl[j] = *reinterpret_cast<const rs_f16x8*>(w_base + RS_W_B + j * 32 * RS_WP + kk * 32);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[j], 0, 0, 0);
    }
    // ---- epilogue: lane = channel j * 32 + frow of pixels (r & 3) + 8 (r >> 2) + 4 fh of output row wave ----
    {
      const int oy = cur.ty * RS_TR + wave, ox0 = cur.tx * RS_TC;
      if (oy < p.Ho) {
        float* orow = p.out + (((size_t)cur.n * p.Ho + oy) * p.Wo + ox0) * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * fh;
            if (ox0 + c < p.Wo) orow[c * 64 + j * 32 + frow] = fmaf(acc[j][r], esc[j], esh[j]);
          }
      }
    }
    if (more) store_patch(buf ^ 1);
    __syncthreads();                               // the next patch is complete; everyone is done with this one
    cur = nxt;
  }
}

bool resnet_stem7x7_supported(int kh, int kw, int cin, int cout, int stride, int pad_mode, int pad, int S) {
  return kh == 7 && kw == 7 && cin == 3 && cout == 64 && stride == 2 && pad_mode == 2 && pad == 3 && S >= 16;
}

int launch_resnet_stem7x7(const float* img_nchw, const unsigned short* wt_hi, const unsigned short* wt_lo, const float* scale,
                          const float* shift, float* out, int N, int S, hipStream_t s) {
  XDET_REQUIRE(img_nchw && wt_hi && wt_lo && scale && shift && out, "resnet_stem: NULL argument");
  if (N <= 0) return XDET_OK;
  ResnetStemParams p;
  p.img = img_nchw; p.wt_hi = wt_hi; p.wt_lo = wt_lo; p.scale = scale; p.shift = shift; p.out = out;
  p.N = N; p.S = S;
  p.Ho = (S + 6 - 7) / 2 + 1; p.Wo = p.Ho;
  p.TY = (int)cdiv(p.Ho, RS_TR); p.TX = (int)cdiv(p.Wo, RS_TC);
  const int64_t nt = (int64_t)N * p.TY * p.TX;
  p.ntiles = (int)nt;
  static DeviceOnce once;
  XDET_TRY(ensure_dynamic_lds(once, reinterpret_cast<const void*>(resnet_stem7x7_kernel), RS_LDS));
  const dim3 g((unsigned)std::min<int64_t>(256, cdiv(nt, 8) * 8));
  hipLaunchKernelGGL(resnet_stem7x7_kernel, g, dim3(512), RS_LDS, s, p);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

}  // namespace xdet
