// Implicit-GEMM convolution / dense layer on the gfx950 f32 MFMA pipe.
//
//   out[m, co] = epilogue( sum_k A[m, k] * Wt[co, k] ),   m = (n, oy, ox),  k = (tap, ci)
//
// A is never materialised (im2col-free): each workgroup gathers its BM x 32 slice of the
// virtual im2col matrix straight from the NHWC activation (zero outside the image) into
// LDS; weights are pre-transposed on the host to [Cout][K] so both operands sit in LDS as
// [row][k] with a 36-float row stride -> every MFMA operand is one conflict-free
// ds_read_b128 of 4 consecutive k.  v_mfma_f32_32x32x2_f32 sums k-slots {h*4+t} over the
// two half-waves h, so four back-to-back MFMAs consume one b128 read of A and of B.
//
// Covers: Xception stem convs (small-cin mode), 1x1 projections and pointwise convs,
// RPN 3x3, large-separable 15x1 / 1x15, dense layers (H=W=1), ResNet 7x7/3x3/1x1.
// Epilogue: y = acc*scale[co] + shift[co] (+ residual) (ReLU): folded inference BN
// (net/xception_body.py:232, net/resnet_v2.py:41-50) or bias.
#include "common.h"

namespace xdet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int LDS_LD = 36;   // floats per LDS row: 32 + 4 pad (144 B, keeps b128 reads conflict-free)

template <int BM, int BN, int WAVES_M, int WAVES_N, bool SMALL_CIN>
__global__ __launch_bounds__(256) void conv_mfma_f32_kernel(ConvParams p) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;   // wave tile
  constexpr int TM = WM / 32, TN = WN / 32;             // 32x32 MFMA tiles per wave
  constexpr int A_IT = BM / 32, B_IT = BN / 32;         // float4 loads per thread per K step

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                          // [2][BM][LDS_LD]
  float* Bs = smem + 2 * BM * LDS_LD;        // [2][BN][LDS_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  const int lrow = tid >> 3;   // 0..31
  const int kq = tid & 7;      // float4 index within the 32-wide K slice

  // per-thread im2col row descriptors
  int iy0[A_IT], ix0[A_IT], pbase[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + lrow + 32 * i;
    if (m < p.M) {
      const int hw = p.Ho * p.Wo;
      const int n = m / hw;
      const int rem = m - n * hw;
      const int oy = rem / p.Wo;
      const int ox = rem - oy * p.Wo;
      iy0[i] = oy * p.stride - p.pad_t;
      ix0[i] = ox * p.stride - p.pad_l;
      pbase[i] = n * p.H * p.W;
    } else {
      iy0[i] = -(1 << 28);
      ix0[i] = 0;
      pbase[i] = 0;
    }
  }

  float4 ra[A_IT], rb[B_IT];
  const int nk = p.Kp / BK;

  auto load_global = [&](int kt) {
    int dy, dx, coff;
    bool tap_ok = true;
    if (SMALL_CIN) {
      const int tap = kt * 8 + kq;           // Cin_p == 4: one float4 per tap
      tap_ok = tap < p.KH * p.KW;
      const int ky = tap / p.KW;
      dy = ky * p.dil;
      dx = (tap - ky * p.KW) * p.dil;
      coff = 0;
    } else {
      const int k0 = kt * BK;
      const int tap = k0 / p.Cin_p;          // block-uniform
      const int ky = tap / p.KW;
      dy = ky * p.dil;
      dx = (tap - ky * p.KW) * p.dil;
      coff = k0 - tap * p.Cin_p + 4 * kq;
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int iy = iy0[i] + dy, ix = ix0[i] + dx;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tap_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
        const float* src = p.in + ((size_t)(pbase[i] + iy * p.W + ix) * p.ldi + coff);
        v = *reinterpret_cast<const float4*>(src);
        if (p.relu_in) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const float* src = p.wt + ((size_t)(n0 + lrow + 32 * i) * p.Kp + kt * BK + 4 * kq);
      rb[i] = *reinterpret_cast<const float4*>(src);
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      *reinterpret_cast<float4*>(&As[(buf * BM + lrow + 32 * i) * LDS_LD + 4 * kq]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      *reinterpret_cast<float4*>(&Bs[(buf * BN + lrow + 32 * i) * LDS_LD + 4 * kq]) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_global(0);
  store_lds(0);
  __syncthreads();

  const int frow = lane & 31;
  const int fh = lane >> 5;

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_global(kt + 1);
    const float* Ab = As + (buf * BM + wm * WM + frow) * LDS_LD + fh * 4;
    const float* Bb = Bs + (buf * BN + wn * WN + frow) * LDS_LD + fh * 4;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) store_lds(buf ^ 1);
    __syncthreads();
  }

  // epilogue: D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = n0 + wn * WN + j * 32 + frow;
    if (co >= p.ldo) continue;
    const float sc = p.scale[co], sh = p.shift[co];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (m < p.M) {
          float v = fmaf(acc[i][j][r], sc, sh);
          if (p.res) v += p.res[(size_t)m * p.ldr + co];
          if (p.relu_out) v = fmaxf(v, 0.f);
          p.out[(size_t)m * p.ldo + co] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool SMALL_CIN>
static int launch_t(const ConvParams& p, hipStream_t s) {
  constexpr size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
  auto kern = conv_mfma_f32_kernel<BM, BN, WAVES_M, WAVES_N, SMALL_CIN>;
  static DeviceOnce once;
  XDET_TRY(ensure_dynamic_lds(once, reinterpret_cast<const void*>(kern), (int)lds));
  dim3 grid((unsigned)cdiv(p.M, BM), (unsigned)(p.Cout_pad / BN));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

int launch_conv_mfma_f32(const ConvParams& p, bool small_cin, int n_tile, hipStream_t s) {
  XDET_REQUIRE(p.Kp % BK == 0, "conv: Kp must be a multiple of 32");
  XDET_REQUIRE(p.Cout_pad % n_tile == 0, "conv: Cout_pad must be a multiple of the N tile");
  XDET_REQUIRE(small_cin || (p.Cin_p % BK == 0 && p.ldi >= p.Cin_p), "conv: Cin_p must be a multiple of 32 and <= ldi");
  XDET_REQUIRE(!small_cin || (p.Cin_p == 4 && p.ldi == 4), "conv: small-cin mode needs 4-channel input");
  XDET_REQUIRE(p.ldi % 4 == 0, "conv: ldi must be a multiple of 4");
  if (p.M <= 0) return XDET_OK;
  if (n_tile == 128) {
    return small_cin ? launch_t<128, 128, 2, 2, true>(p, s) : launch_t<128, 128, 2, 2, false>(p, s);
  } else if (n_tile == 64) {
    return small_cin ? launch_t<128, 64, 4, 1, true>(p, s) : launch_t<128, 64, 4, 1, false>(p, s);
  }
  set_last_error("conv: unsupported N tile");
  return XDET_ERR_UNSUPPORTED;
}

}  // namespace xdet
