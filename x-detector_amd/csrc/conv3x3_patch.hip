// 3x3 / stride 1 / VALID convolution over 32 input channels with the INPUT TILE STAGED IN LDS ONCE: block1_conv2 of
// the Xception entry (net/xception_body.py:252-259: 32 -> 64 channels on a 239 x 239 map, BN + ReLU).
//
// On the generic implicit-GEMM kernel this layer is a K = 288 GEMM with a 64-wide N tile: every one of its nine
// K steps re-fetches a shifted 128-pixel view of the same activation rows through the LDS DMA (144 KB of DMA per
// 128 output pixels, 9x the unique bytes) against only 12 MFMAs per wave -- 0.57 ms per 64 images where the HBM
// bound is 0.28.  Here a workgroup stages the (4 + 2) x 32-pixel patch of the split f16 planes once (24 KB), and the
// nine taps are nine shifted fragment reads of that patch:
//
//   patch  (6 rows x 32 px x 32 ch, hi and lo planes)   --buffer_load ... lds, 1 KB pieces, chunk-permuted-->  LDS
//   tap (ky, kx), 16-deep half ks:  A fragment of output pixel px = patch[(row + ky) * 32 + px + kx]  (ds_read_b128)
//   B fragments: the whole 3 x 3 x 32 x 64 filter lives in REGISTERS for the lifetime of the (persistent) workgroup
//   108 x v_mfma_f32_32x32x16_f16 per wave and tile (f16x3: lo*hi, hi*lo, hi*hi per half, taps ascending) -> BN + ReLU
//
// Tile = 4 output rows x 30 pixels (a 128-row GEMM tile; 30 + 2 = 32 patch columns = two 1 KB DMA pieces per row
// and plane), 4 waves as 2 x 2 over 128 pixels x 64 channels, two patch buffers (the next tile's patch is in flight
// during the MFMAs; fragment reads behind the DMA are inline asm, see sepconv_fused.hip), XCD-band interleaved tile
// order.  K order (tap ascending, two halves) and product order are conv_dma_f16_kernel's: bit-identical.
#include "common.h"

namespace xdet {

typedef float cp_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 cp_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int CP_R = 4, CP_X = 30, CP_P = 32, CP_ROWS = CP_R + 2;
constexpr int CP_PLANE_H = CP_ROWS * CP_P * 32 + 3 * 32;     // halves per plane per buffer (+3 pixels of slack: px 30, 31 + kx)
constexpr int CP_NJ = CP_ROWS * 2 * 2 / 4;                   // DMA pieces per wave per tile (rows x 2 pieces x 2 planes / 4 waves)

struct Conv3x3PatchParams {
  const u16* in_hi; const u16* in_lo;    // planes [N*H*W][32]
  const u16* wt_hi; const u16* wt_lo;    // K-blocked [9][64][32]
  const float* scale; const float* shift;
  float* out;                            // NHWC f32 [N][Ho][Wo][ldo]
  int N, H, W, Ho, Wo, ldo, relu, TY, TX, ntiles;
};

__device__ __forceinline__ unsigned cp_lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) void*)(p);
}
template <int OFF>
__device__ __forceinline__ cp_f16x8 cp_ds_read_b128(unsigned addr) {
  cp_f16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}

template <bool SPLIT3>
__global__ __launch_bounds__(256, 2) void conv3x3_patch_kernel(Conv3x3PatchParams p) {
  __shared__ __attribute__((aligned(16))) u16 s_patch[2][2][CP_PLANE_H];     // [buffer][hi | lo]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, G = gridDim.x >> 3;
  const int per_xcd = (p.ntiles + 7) >> 3;
  const int t_begin = xcd * per_xcd + wg;
  const int t_end = min(p.ntiles, (xcd + 1) * per_xcd);
  if (t_begin >= t_end) return;

  struct Coord { int ty, tx, n; };
  auto decode = [&](int q) {
    Coord c;
    c.ty = q % p.TY; q /= p.TY;
    c.tx = q % p.TX;
    c.n = q / p.TX;
    return c;
  };
  Coord cur = decode(t_begin);

  const size_t plane_bytes = (size_t)p.N * p.H * p.W * 32 * 2;
  const __amdgpu_buffer_rsrc_t rs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.in_hi), 0, (int)plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.in_lo ? p.in_lo : p.in_hi), 0, (int)plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (int)(unsigned)std::min<size_t>((size_t)p.N * p.Ho * p.Wo * p.ldo * 4, 0xffffffffull), 0x00020000);

  // DMA piece: 16 pixels x 64 B of one patch row of one plane; lane = (pixel, 16-B chunk).  The chunk a lane FETCHES
  // is permuted with the pixel's (q >> 2) & 3 so that the fragment reads below are bank-conflict free.
  const int dpx = lane >> 2, dch = lane & 3;
  auto issue = [&](const Coord& c, int buf, bool live) {
    const int y0 = c.ty * CP_R, x0 = c.tx * CP_X;
#pragma unroll
    for (int jj = 0; jj < CP_NJ; ++jj) {
      const int i = wave + 4 * jj;                  // 0..23: (row, half-row, plane)
      const int plane = i & 1, seg = (i >> 1) & 1, rr = i >> 2;
      const int q = rr * CP_P + seg * 16 + dpx;     // patch pixel of this lane
      const int y = y0 + rr, x = x0 + seg * 16 + dpx;
      const bool ok = live && y < p.H && x < p.W;
      const unsigned voff = ok ? (unsigned)(((((size_t)c.n * p.H + y) * p.W + x) * 32 + ((dch ^ ((q >> 2) & 3)) << 3)) * 2) : 0xffffffffu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(plane ? rs_lo : rs_hi,
                                               (__attribute__((address_space(3))) void*)(&s_patch[buf][plane][(rr * CP_P + seg * 16) * 32]),
                                               16, voff, 0, 0, 0);
    }
  };

  const int frow = lane & 31, fh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;          // 2 x 2 waves: 64 pixels (two tile rows) x 32 channels each

  // the whole filter of this wave's 32 output channels, both halves of all nine taps, in registers
  cp_f16x8 bh[9][2], bl[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const size_t o = ((size_t)t * 64 + (wn * 32 + frow)) * 32 + (ks * 2 + fh) * 8;
      bh[t][ks] = *reinterpret_cast<const cp_f16x8*>(p.wt_hi + o);
      if (SPLIT3) bl[t][ks] = *reinterpret_cast<const cp_f16x8*>(p.wt_lo + o);
    }
  const float esc = p.scale[wn * 32 + frow], esh = p.shift[wn * 32 + frow];

  int buf = 0;
  issue(cur, 0, true);
  for (int t = t_begin; t < t_end; t += G, buf ^= 1) {
    const Coord nxt = decode(min(t + G, p.ntiles - 1));
    const int y0 = cur.ty * CP_R, x0 = cur.tx * CP_X;
    // patch(t) has landed (the DMA's LDS writes retire through vmcnt) and every wave is done with the other buffer
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue(nxt, buf ^ 1, t + G < t_end);
    __builtin_amdgcn_sched_barrier(0);
    cp_f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned pbase = cp_lds_addr(&s_patch[buf][0][0]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
      cp_f16x8 ah[2][2], al[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = (wm * 2 + i + ky) * CP_P + frow + kx;                 // patch pixel feeding output pixel frow
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const unsigned a = pbase + (unsigned)(q * 32 + (((ks * 2 + fh) ^ ((q >> 2) & 3)) << 3)) * 2u;
          ah[i][ks] = cp_ds_read_b128<0>(a);
          if (SPLIT3) al[i][ks] = cp_ds_read_b128<CP_PLANE_H * 2>(a);
        }
      }
      if (SPLIT3)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ah[0][0]), "+v"(ah[0][1]), "+v"(ah[1][0]), "+v"(ah[1][1]), "+v"(al[0][0]), "+v"(al[0][1]),
                       "+v"(al[1][0]), "+v"(al[1][1])::"memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[0][0]), "+v"(ah[0][1]), "+v"(ah[1][0]), "+v"(ah[1][1])::"memory");
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (SPLIT3) {
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i][ks], bh[tap][ks], acc[i], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i][ks], bl[tap][ks], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i][ks], bh[tap][ks], acc[i], 0, 0, 0);
      }
    }
    // epilogue: accumulator block i = tile row wm*2 + i, a lane holds channel wn*32 + frow of 16 pixels
    const int lim = min(CP_X, p.Wo - x0) - 4 * fh;
    const int co = wn * 32 + frow;
    const unsigned lane_off = (unsigned)((4 * fh * p.ldo + co) * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int y = y0 + wm * 2 + i;
      if (y >= p.Ho) continue;                      // wave-uniform
      const unsigned row_off = (unsigned)((((size_t)cur.n * p.Ho + y) * p.Wo + x0) * p.ldo * 4);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = (r & 3) + 8 * (r >> 2);
        float v = fmaf(acc[i][r], esc, esh);
        if (p.relu) v = fmaxf(v, 0.f);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, c < lim ? lane_off : 0xffffffffu,
                                              row_off + c * p.ldo * 4, 0);
      }
    }
    cur = nxt;
  }
}

bool conv3x3_patch_supported(int kh, int kw, int cin, int cout_pad, int stride, int dil, int pad_mode) {
  return kh == 3 && kw == 3 && cin == 32 && cout_pad == 64 && stride == 1 && dil == 1 && pad_mode == 0;
}

int launch_conv3x3_patch(const unsigned short* in_hi, const unsigned short* in_lo, const unsigned short* wt_hi_blocked,
                         const unsigned short* wt_lo_blocked, const float* scale, const float* shift, float* out, int N, int H,
                         int W, int ldo, int relu, hipStream_t s) {
  XDET_REQUIRE(in_hi && wt_hi_blocked && scale && shift && out && H >= 3 && W >= 3 && ldo >= 64, "conv3x3_patch: bad arguments");
  const int Ho = H - 2, Wo = W - 2;
  // planes and output are addressed with 32-bit byte offsets: image ranges below 2 GiB per launch
  const size_t per_image = std::max((size_t)H * W * 32 * 2, (size_t)Ho * Wo * ldo * 4);
  const int n_max = (int)std::max<size_t>(1, (((size_t)1 << 31) - 1) / per_image);
  for (int nb = 0; nb < N; nb += n_max) {
    const int n = std::min(n_max, N - nb);
    // the planes of an image range start at a multiple of 16 pixels only if nb * H * W is one; the [pix][32] layout of a
    // 32-channel plane is plainly contiguous, so any pixel offset is fine
    Conv3x3PatchParams p;
    p.in_hi = in_hi + (size_t)nb * H * W * 32;
    p.in_lo = in_lo ? in_lo + (size_t)nb * H * W * 32 : nullptr;
    p.wt_hi = wt_hi_blocked; p.wt_lo = wt_lo_blocked; p.scale = scale; p.shift = shift;
    p.out = out + (size_t)nb * Ho * Wo * ldo;
    p.N = n; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.ldo = ldo; p.relu = relu;
    p.TY = (int)cdiv(Ho, CP_R); p.TX = (int)cdiv(Wo, CP_X);
    const int64_t nt = (int64_t)n * p.TY * p.TX;
    p.ntiles = (int)nt;
    const dim3 g((unsigned)std::min<int64_t>(512, cdiv(nt, 8) * 8));
    if (wt_lo_blocked) hipLaunchKernelGGL((conv3x3_patch_kernel<true>), g, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<false>), g, dim3(256), 0, s, p);
    XDET_LAUNCH_CHECK();
  }
  return XDET_OK;
}

}  // namespace xdet
