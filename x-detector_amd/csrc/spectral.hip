// Spectral form of the large-separable convolutions (net/xception_body.py:450-475): a (15,1) / (1,15)
// SAME convolution is a 1-D correlation along one image axis, i.e. a pointwise product in the DFT
// domain of that axis.  With L = F + 14 >= F + 15 - 1 points (F = feature-map side) the circular
// convolution equals the zero-padded linear one, so
//
//     y = IDFT_L( DFT_L(x) . DFT_L(g) ),   g[(7 - t) mod L] = w[t]
//
// exactly (in exact arithmetic), and the 15 x Cin x Cout MACs per output pixel of the direct form become
// L/2 complex bins x 4 x Cin x Cout / F per pixel: 5.2x fewer at F = 30 (22 bins), 3.7x at F = 50.
// The per-bin contractions stay on the split-precision MFMA kernel (conv_mfma_dma.hip, grouped GEMM:
// bin b = rows [b*m_pad, (b+1)*m_pad) with its own [2Cin x 2Cout] real block matrix [[Gr, Gi], [-Gi, Gr]]);
// this file holds the two transforms around them:
//
//   dft_fwd_kernel: f32 NHWC [N,F,F,ld] --(real DFT along y or x)--> split f16 planes
//                   [bin][m = n*F + other][re: ld | im: ld], the A operand of the grouped GEMM
//   dft_inv_kernel: f32 [bin][m][re: C | im: C] --(inverse real DFT, per-channel scale/shift, ReLU)-->
//                   f32 NHWC [N,F,F,ldo]
//
// A real signal's DFT is conjugate-symmetric: bins 1..L/2-1 are kept as complex bins, and the two real
// bins (DC and Nyquist) share "bin 0" (re slot = DC, im slot = Nyquist) with a block-diagonal weight
// matrix.  The transforms are dense F x L real matrices applied with VALU FMAs (F <= 50: an FFT
// butterfly network would save little and cost registers); the twiddle tables are wave-uniform and
// come through the scalar cache.  One thread owns 4 channels of one line; a wave covers 8 lines x 32
// channels, so every load/store instruction moves 8 x 128 B (f32) or one contiguous 512 B (planes).
#include "common.h"

#include <cmath>
#include <vector>

namespace xdet {

typedef _Float16 sp_f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

__device__ __forceinline__ size_t sp_blocked_off(size_t pix, int c, int c32n) {
  return (((pix >> 4) * (size_t)c32n + (size_t)(c >> 5)) << 9) + ((pix & 15) << 5) + (size_t)(c & 31);
}

__device__ __forceinline__ void sp_store_split(u16* __restrict__ hi, u16* __restrict__ lo, size_t off, const float4 v) {
  const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y, h2 = (_Float16)v.z, h3 = (_Float16)v.w;
  sp_f16x4 hv = {h0, h1, h2, h3};
  sp_f16x4 lv = {(_Float16)(v.x - (float)h0), (_Float16)(v.y - (float)h1), (_Float16)(v.z - (float)h2),
                 (_Float16)(v.w - (float)h3)};
  *reinterpret_cast<uint2*>(hi + off) = *reinterpret_cast<uint2*>(&hv);
  *reinterpret_cast<uint2*>(lo + off) = *reinterpret_cast<uint2*>(&lv);
}

// tab: [L/2][2][F]: row (b,0) = coefficients of the re slot, (b,1) = of the im slot
template <int F, int L>
__global__ __launch_bounds__(256) void dft_fwd_kernel(const float* __restrict__ in, int ld, int axis, int M, int m_pad,
                                                      const float* __restrict__ tab, u16* __restrict__ hi,
                                                      u16* __restrict__ lo) {
  constexpr int NB = L / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cblk = blockIdx.y * 4 + wave;            // 32-channel block of this wave
  if (cblk * 32 >= ld) return;
  const int c = cblk * 32 + (lane & 7) * 4;
  const int m = blockIdx.x * 8 + (lane >> 3);
  const bool ok = m < M;
  const int n = m / F, o = m - n * F;
  // axis 0: transform along y (samples F*ld floats apart), the line is (n, x = o);
  // axis 1: transform along x (samples ld floats apart), the line is (n, y = o)
  const float* base = in + ((size_t)n * F * F + (axis == 0 ? o : o * F)) * ld + c;
  const size_t step = (size_t)(axis == 0 ? F : 1) * ld;
  float4 v[F];
#pragma unroll
  for (int i = 0; i < F; ++i) v[i] = ok ? *reinterpret_cast<const float4*>(base + i * step) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int c32n = (2 * ld) >> 5;
  // gridDim.z workgroups share a line's bins (every bin is computed from the line alone: how the bins are dealt out
  // changes nothing in the results).  One workgroup per line walks all L/2 bins in ~30 us -- fine when thousands of lines
  // fill the chip, the critical path of a single image's large-separable branch otherwise (4 x 16 workgroups).
  const int nbz = (NB + (int)gridDim.z - 1) / (int)gridDim.z;
  const int b_lo = (int)blockIdx.z * nbz, b_hi = min(NB, b_lo + nbz);
#pragma unroll 1
  for (int b = b_lo; b < b_hi; ++b) {
    const float* __restrict__ tr = tab + (size_t)b * 2 * F;
    const float* __restrict__ ti = tr + F;
    float4 re = make_float4(0.f, 0.f, 0.f, 0.f), im = re;
#pragma unroll
    for (int i = 0; i < F; ++i) {
      const float cr = tr[i], ci = ti[i];
      re.x = fmaf(v[i].x, cr, re.x); re.y = fmaf(v[i].y, cr, re.y); re.z = fmaf(v[i].z, cr, re.z); re.w = fmaf(v[i].w, cr, re.w);
      im.x = fmaf(v[i].x, ci, im.x); im.y = fmaf(v[i].y, ci, im.y); im.z = fmaf(v[i].z, ci, im.z); im.w = fmaf(v[i].w, ci, im.w);
    }
    if (ok) {
      const size_t row = (size_t)b * m_pad + m;
      sp_store_split(hi, lo, sp_blocked_off(row, c, c32n), re);
      sp_store_split(hi, lo, sp_blocked_off(row, ld + c, c32n), im);
    }
  }
}

// Y: [L/2 * m_pad][ldn] f32, re at channel c, im at channel C_ld + c.  tab: [L/2][2][F] (inverse coefficients,
// 1/L and the factor 2 of the conjugate half folded in).  out = relu?(idft * scale + shift), NHWC stride ldo.
// ZI: workgroups per line along the output positions (each output is its own sum over the bins: dealing the positions
// out changes nothing in the results); 2 for the few lines of one or two images, else 1.
template <int F, int L, int ZI>
__global__ __launch_bounds__(256) void dft_inv_kernel(const float* __restrict__ Y, int ldn, int C_ld, int M, int m_pad,
                                                      const float* __restrict__ tab, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int relu, float* __restrict__ out,
                                                      int ldo, int axis) {
  constexpr int NB = L / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cblk = blockIdx.y * 4 + wave;
  if (cblk * 32 >= C_ld) return;
  const int c = cblk * 32 + (lane & 7) * 4;
  const int m = blockIdx.x * 8 + (lane >> 3);
  if (m >= M) return;
  const int n = m / F, o = m - n * F;
  static_assert(F % ZI == 0, "output positions must divide evenly");
  constexpr int G = F / ZI;                          // output positions of this workgroup: [i0, i0 + G)
  const int i0 = (int)blockIdx.z * G;
  float4 acc[G];
#pragma unroll
  for (int i = 0; i < G; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
  for (int b = 0; b < NB; ++b) {
    const float* row = Y + ((size_t)b * m_pad + m) * ldn + c;
    const float4 yr = *reinterpret_cast<const float4*>(row);
    const float4 yi = *reinterpret_cast<const float4*>(row + C_ld);
    const float* __restrict__ tr = tab + (size_t)b * 2 * F + i0;
    const float* __restrict__ ti = tr + F;
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const float cr = tr[i], ci = ti[i];
      acc[i].x = fmaf(yr.x, cr, acc[i].x); acc[i].y = fmaf(yr.y, cr, acc[i].y);
      acc[i].z = fmaf(yr.z, cr, acc[i].z); acc[i].w = fmaf(yr.w, cr, acc[i].w);
      acc[i].x = fmaf(yi.x, ci, acc[i].x); acc[i].y = fmaf(yi.y, ci, acc[i].y);
      acc[i].z = fmaf(yi.z, ci, acc[i].z); acc[i].w = fmaf(yi.w, ci, acc[i].w);
    }
  }
  const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
  const size_t step = (size_t)(axis == 0 ? F : 1) * ldo;
  float* base = out + ((size_t)n * F * F + (axis == 0 ? o : o * F)) * ldo + c + (size_t)i0 * step;
#pragma unroll
  for (int i = 0; i < G; ++i) {
    float4 t = make_float4(fmaf(acc[i].x, sc.x, sh.x), fmaf(acc[i].y, sc.y, sh.y), fmaf(acc[i].z, sc.z, sh.z),
                           fmaf(acc[i].w, sc.w, sh.w));
    // NaN-propagating ReLU: fmaxf(NaN, 0) = 0 would turn an overflowed bin (hi = inf in the planes -> NaN out of the
    // GEMM) into silent zeros in `feat`; kept as NaN it reaches the head logits and the always-on guard of bboxes_eval
    if (relu) {
      t.x = (t.x > 0.f || t.x != t.x) ? t.x : 0.f; t.y = (t.y > 0.f || t.y != t.y) ? t.y : 0.f;
      t.z = (t.z > 0.f || t.z != t.z) ? t.z : 0.f; t.w = (t.w > 0.f || t.w != t.w) ? t.w : 0.f;
    }
    *reinterpret_cast<float4*>(base + i * step) = t;
  }
}

bool spectral_supported(int F) { return F == 16 || F == 30 || F == 50; }
int spectral_points(int F) { return F + 14; }   // even for every supported F

// host: the DFT tables of one axis length.  fwd/inv: [L/2][2][F] as the kernels read them.
void spectral_tables(int F, std::vector<float>* fwd, std::vector<float>* inv) {
  const int L = spectral_points(F), NB = L / 2;
  fwd->assign((size_t)NB * 2 * F, 0.f);
  inv->assign((size_t)NB * 2 * F, 0.f);
  const double w = 2.0 * M_PI / L;
  for (int i = 0; i < F; ++i) {
    // bin 0: re slot = DC, im slot = Nyquist (both real)
    (*fwd)[(size_t)0 * 2 * F + i] = 1.f;
    (*fwd)[(size_t)0 * 2 * F + F + i] = (i & 1) ? -1.f : 1.f;
    (*inv)[(size_t)0 * 2 * F + i] = (float)(1.0 / L);
    (*inv)[(size_t)0 * 2 * F + F + i] = (float)(((i & 1) ? -1.0 : 1.0) / L);
    for (int b = 1; b < NB; ++b) {
      const double th = w * (double)(((long long)b * i) % L);
      (*fwd)[((size_t)b * 2) * F + i] = (float)std::cos(th);          // X[b] = sum x[i] e^{-i th}
      (*fwd)[((size_t)b * 2 + 1) * F + i] = (float)(-std::sin(th));
      (*inv)[((size_t)b * 2) * F + i] = (float)(2.0 * std::cos(th) / L);   // x[i] = 1/L sum_b' Y[b'] e^{+i th}
      (*inv)[((size_t)b * 2 + 1) * F + i] = (float)(-2.0 * std::sin(th) / L);
    }
  }
}

// host: the per-bin real block matrices of a T-tap kernel w[T][cin][cout] (pad = T/2 leading taps, TF SAME),
// laid out as a grouped 1x1 weight [L/2][2*cin_ld][2*cout_ld] (rows: re | im of the input, cols: re | im of the output)
void spectral_weights(const float* w, int T, int cin, int cout, int cin_ld, int cout_ld, int F, std::vector<float>* out) {
  const int L = spectral_points(F), NB = L / 2, pad = T / 2;
  const size_t K2 = 2 * (size_t)cin_ld, N2 = 2 * (size_t)cout_ld;
  out->assign((size_t)NB * K2 * N2, 0.f);
  const double om = 2.0 * M_PI / L;
  std::vector<double> cr(T), ci(T);
  std::vector<double> gr(cout), gi(cout);
  for (int b = 0; b < NB; ++b) {
    float* B = out->data() + (size_t)b * K2 * N2;
    if (b == 0) {
      // DC: G[0] = sum_t w[t]; Nyquist: G[L/2] = sum_t w[t] (-1)^{(pad - t) mod L}
      for (int k = 0; k < cin; ++k) {
        for (int n = 0; n < cout; ++n) { gr[n] = 0; gi[n] = 0; }
        for (int t = 0; t < T; ++t) {
          const int j = ((pad - t) % L + L) % L;
          const double sgn = (j & 1) ? -1.0 : 1.0;
          const float* src = w + ((size_t)t * cin + k) * cout;
          for (int n = 0; n < cout; ++n) { gr[n] += src[n]; gi[n] += sgn * src[n]; }
        }
        float* r0 = B + (size_t)k * N2;                  // DC input row -> DC output columns
        float* r1 = B + ((size_t)cin_ld + k) * N2;       // Nyquist input row -> Nyquist output columns
        for (int n = 0; n < cout; ++n) { r0[n] = (float)gr[n]; r1[cout_ld + n] = (float)gi[n]; }
      }
      continue;
    }
    for (int t = 0; t < T; ++t) {
      const int j = ((pad - t) % L + L) % L;
      const double th = om * (double)(((long long)b * j) % L);
      cr[t] = std::cos(th);
      ci[t] = -std::sin(th);
    }
    for (int k = 0; k < cin; ++k) {
      for (int n = 0; n < cout; ++n) { gr[n] = 0; gi[n] = 0; }
      for (int t = 0; t < T; ++t) {
        const float* src = w + ((size_t)t * cin + k) * cout;
        const double a = cr[t], bb = ci[t];
        for (int n = 0; n < cout; ++n) { gr[n] += a * src[n]; gi[n] += bb * src[n]; }
      }
      // [Xr Xi] x [[Gr, Gi], [-Gi, Gr]] = [Xr Gr - Xi Gi, Xr Gi + Xi Gr]
      float* r0 = B + (size_t)k * N2;
      float* r1 = B + ((size_t)cin_ld + k) * N2;
      for (int n = 0; n < cout; ++n) {
        r0[n] = (float)gr[n]; r0[cout_ld + n] = (float)gi[n];
        r1[n] = (float)(-gi[n]); r1[cout_ld + n] = (float)gr[n];
      }
    }
  }
}

template <int F>
static int launch_fwd_t(const float* in, int ld, int axis, int N, int m_pad, const float* tab, u16* hi, u16* lo, hipStream_t s) {
  const int M = N * F;
  // few lines (one or two images): deal the bins of a line to up to 6 workgroups so that the launch covers the chip
  const int64_t wgs = cdiv(M, 8) * cdiv(ld / 32, 4);
  const unsigned z = wgs >= 512 ? 1u : wgs >= 256 ? 2u : wgs >= 128 ? 4u : 6u;
  dim3 grid((unsigned)cdiv(M, 8), (unsigned)cdiv(ld / 32, 4), z);
  hipLaunchKernelGGL((dft_fwd_kernel<F, F + 14>), grid, dim3(256), 0, s, in, ld, axis, M, m_pad, tab, hi, lo);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}
template <int F>
static int launch_inv_t(const float* Y, int ldn, int C_ld, int N, int m_pad, const float* tab, const float* scale,
                        const float* shift, int relu, float* out, int ldo, int axis, hipStream_t s) {
  const int M = N * F;
  if (cdiv(M, 8) * cdiv(C_ld / 32, 4) < 256) {        // few lines: two workgroups per line
    dim3 grid((unsigned)cdiv(M, 8), (unsigned)cdiv(C_ld / 32, 4), 2);
    hipLaunchKernelGGL((dft_inv_kernel<F, F + 14, 2>), grid, dim3(256), 0, s, Y, ldn, C_ld, M, m_pad, tab, scale, shift, relu,
                       out, ldo, axis);
  } else {
    dim3 grid((unsigned)cdiv(M, 8), (unsigned)cdiv(C_ld / 32, 4));
    hipLaunchKernelGGL((dft_inv_kernel<F, F + 14, 1>), grid, dim3(256), 0, s, Y, ldn, C_ld, M, m_pad, tab, scale, shift, relu,
                       out, ldo, axis);
  }
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

int launch_dft_fwd(const float* in, int F, int ld, int axis, int N, int m_pad, const float* tab, unsigned short* hi,
                   unsigned short* lo, hipStream_t s) {
  XDET_REQUIRE(ld % 32 == 0 && (axis == 0 || axis == 1) && N > 0 && m_pad >= N * F, "dft_fwd: bad arguments");
  switch (F) {
    case 16: return launch_fwd_t<16>(in, ld, axis, N, m_pad, tab, hi, lo, s);
    case 30: return launch_fwd_t<30>(in, ld, axis, N, m_pad, tab, hi, lo, s);
    case 50: return launch_fwd_t<50>(in, ld, axis, N, m_pad, tab, hi, lo, s);
  }
  set_last_error("dft_fwd: unsupported feature-map side");
  return XDET_ERR_UNSUPPORTED;
}

int launch_dft_inv(const float* Y, int F, int ldn, int C_ld, int N, int m_pad, const float* tab, const float* scale,
                   const float* shift, int relu, float* out, int ldo, int axis, hipStream_t s) {
  XDET_REQUIRE(C_ld % 32 == 0 && ldn >= 2 * C_ld && ldo >= C_ld && (axis == 0 || axis == 1) && N > 0 && m_pad >= N * F,
               "dft_inv: bad arguments");
  switch (F) {
    case 16: return launch_inv_t<16>(Y, ldn, C_ld, N, m_pad, tab, scale, shift, relu, out, ldo, axis, s);
    case 30: return launch_inv_t<30>(Y, ldn, C_ld, N, m_pad, tab, scale, shift, relu, out, ldo, axis, s);
    case 50: return launch_inv_t<50>(Y, ldn, C_ld, N, m_pad, tab, scale, shift, relu, out, ldo, axis, s);
  }
  set_last_error("dft_inv: unsupported feature-map side");
  return XDET_ERR_UNSUPPORTED;
}

}  // namespace xdet
