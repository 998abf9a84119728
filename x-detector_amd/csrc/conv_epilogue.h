// Shared epilogue of the 16-bit MFMA conv kernels (conv_mfma_dma.hip, conv_mfma_split.hip).
#pragma once
#include "common.h"

namespace xdet {

typedef float ep_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 ep_f16x4 __attribute__((ext_vector_type(4)));

// ReLU that keeps NaN: fmaxf(NaN, 0) = 0 would launder an overflowed split-precision operand (hi = inf -> NaN out
// of the MFMAs) into a clean zero; !(v <= 0) is true for v > 0 and for NaN.  Same result as fmaxf for every other input.
__device__ __forceinline__ float ep_relu(float v) { return !(v <= 0.f) ? v : 0.f; }

// The MFMA accumulator layout gives each lane one
// column and 16 scattered rows, i.e. 4-byte global stores (and residual loads).  Bounce each 32-row slab
// of the wave tile through the (now idle) operand LDS so a lane owns 4 consecutive channels of a row:
// 16-B coalesced residual loads and stores, 4x fewer memory instructions.
template <int WM, int WN, int TM, int TN, int NW, int LDS_BYTES>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, ep_f32x16 (&acc)[TM][TN], unsigned short* smem16, int wave,
                                              int lane, int wm, int wn, int m0, int n0) {
  const int frow = lane & 31;
  const int fh = lane >> 5;
  constexpr int EP_LD = WN + 4;                  // floats per staged row (+4: keeps float4 rows 16-B aligned)
  constexpr int C4N = WN / 4;                    // float4 columns per row
  static_assert(NW * 32 * EP_LD * 4 <= LDS_BYTES, "epilogue staging must fit the operand LDS");
  __syncthreads();                               // every wave is done reading its operands
  float* ep = reinterpret_cast<float*>(smem16) + wave * (32 * EP_LD);
  const int c4 = lane % C4N;                     // fixed per lane: its 4 output channels
  const int co4 = n0 + wn * WN + c4 * 4;
  const bool col_ok = co4 < p.ldo;
  float4 sc4 = make_float4(0.f, 0.f, 0.f, 0.f), sh4 = sc4;
  if (col_ok) {
    sc4 = *reinterpret_cast<const float4*>(p.scale + co4);
    sh4 = *reinterpret_cast<const float4*>(p.shift + co4);
  }
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.out_hi && p.pl_scale && col_ok) {
    psc = *reinterpret_cast<const float4*>(p.pl_scale + co4);
    psh = *reinterpret_cast<const float4*>(p.pl_shift + co4);
  }
  constexpr int NQ = (32 * C4N) / 64;            // float4 rows a lane handles per 32-row slab
  // residual rows of slab i are requested before slab i-1 is stored, so their HBM latency hides
  // behind the LDS bounce instead of being paid once per slab (one workgroup per CU: nothing else
  // would cover it)
  float4 rr[NQ];
  auto load_res = [&](int i) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int m = m0 + wm * WM + i * 32 + (q * 64 + lane) / C4N;
      rr[q] = (col_ok && m < p.M) ? *reinterpret_cast<const float4*>(p.res + (size_t)m * p.ldr + co4)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (p.res) load_res(0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ep[((r & 3) + 8 * (r >> 2) + 4 * fh) * EP_LD + j * 32 + frow] = acc[i][j][r];
    float4 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(ep + ((q * 64 + lane) / C4N) * EP_LD + c4 * 4);
      v[q] = make_float4(fmaf(a.x, sc4.x, sh4.x), fmaf(a.y, sc4.y, sh4.y), fmaf(a.z, sc4.z, sh4.z),
                         fmaf(a.w, sc4.w, sh4.w));
      if (p.res) { v[q].x += rr[q].x; v[q].y += rr[q].y; v[q].z += rr[q].z; v[q].w += rr[q].w; }
      if (p.relu_out) {
        v[q].x = ep_relu(v[q].x); v[q].y = ep_relu(v[q].y); v[q].z = ep_relu(v[q].z); v[q].w = ep_relu(v[q].w);
      }
    }
    if (p.res && i + 1 < TM) load_res(i + 1);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int m = m0 + wm * WM + i * 32 + (q * 64 + lane) / C4N;
      if (col_ok && m < p.M) {
        if (p.out) *reinterpret_cast<float4*>(p.out + (size_t)m * p.ldo + co4) = v[q];
        if (p.out_hi) {   // second copy as split planes for a consumer on the LDS-DMA path
          float4 t = v[q];
          if (p.pl_scale) {
            t.x = fmaf(t.x, psc.x, psh.x); t.y = fmaf(t.y, psc.y, psh.y);
            t.z = fmaf(t.z, psc.z, psh.z); t.w = fmaf(t.w, psc.w, psh.w);
          }
          if (p.planes_relu) { t.x = ep_relu(t.x); t.y = ep_relu(t.y); t.z = ep_relu(t.z); t.w = ep_relu(t.w); }
          const _Float16 h0 = (_Float16)t.x, h1 = (_Float16)t.y, h2 = (_Float16)t.z, h3 = (_Float16)t.w;
          ep_f16x4 hv = {h0, h1, h2, h3};
          ep_f16x4 lv = {(_Float16)(t.x - (float)h0), (_Float16)(t.y - (float)h1), (_Float16)(t.z - (float)h2),
                      (_Float16)(t.w - (float)h3)};
          const size_t o = ((((size_t)m >> 4) * (size_t)(p.ldo >> 5) + (size_t)(co4 >> 5)) << 9) + ((m & 15) << 5) + (co4 & 31);
          *reinterpret_cast<uint2*>(p.out_hi + o) = *reinterpret_cast<uint2*>(&hv);
          *reinterpret_cast<uint2*>(p.out_lo + o) = *reinterpret_cast<uint2*>(&lv);
        }
      }
    }
  }
}

}  // namespace xdet
