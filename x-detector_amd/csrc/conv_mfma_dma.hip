// Implicit-GEMM convolution / dense layer, split-precision f16x3 on the 16-bit MFMA pipe, for
// activations that ALREADY live in HBM as two f16 planes (hi, lo) -- written that way by the
// producing kernel (depthwise conv, split pass) so that the split costs VALU work once per
// element instead of once per N-tile, and the im2col gather needs no registers at all:
//
//   * every operand tile goes HBM/L2 -> LDS by `global_load_lds_dwordx4` (16 B per lane, no VGPR
//     round trip); out-of-image taps and the M tail read a zero page instead of branching;
//   * LDS tiles are dense [row][32 halves] (the DMA destination is lane-linear), made
//     conflict-free for the MFMA operand reads by permuting the 16-B chunks of a row with
//     (row>>2)&3 on the SOURCE address and undoing it on the ds_read_b128 address;
//   * both operands are K-blocked in HBM -- activations [C/32][pixels][32], weights
//     [Kp/32][Cout_pad][32] -- so the 16 rows x 64 B one DMA instruction moves are one contiguous
//     1 KB run (tools/ubench/dma_rate.hip: 63 GB/s per CU for 1 KB runs vs 30 GB/s for 64 B segments
//     at a row stride when the stream misses L2);
//   * two LDS stages (64 KB for a 128x128 tile -> two workgroups per CU), one barrier per
//     32-deep K step: the DMA of step k+1 flies while step k is multiplied.
//
// Arithmetic is identical to conv_mfma_split.hip (a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, f32
// accumulate, per-channel power-of-two weight pre-scale folded into the epilogue).
#include "common.h"
#include <cstdlib>
#include <cstring>

namespace xdet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#define XDET_GLDS16(gptr, lptr)                                                                        \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),              \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <int BM, int BN, int WAVES_M, int WAVES_N, int NSPLIT, int NSTAGE>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_dma_f16_kernel(ConvParams p) {
  static_assert(NSTAGE == 2 || NSTAGE == 3, "2 or 3 LDS stages");
  constexpr int NW = WAVES_M * WAVES_N;          // waves per workgroup (4 or 8)
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_IT = BM / (16 * NW);           // DMA instructions per wave per plane per K step
  constexpr int B_IT = BN / (16 * NW);
  static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for the wave count");
  constexpr int ROWB = 32;                       // halves per LDS row (64 B)
  constexpr int STAGE = (2 * BM + 2 * BN) * ROWB;   // halves per stage

  extern __shared__ __attribute__((aligned(16))) u16 smem16[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (private L2s): give every
  // XCD whole M-tiles, all N-tiles of one M-tile back to back, so an activation tile is pulled over
  // the fabric once and then re-read from that XCD's L2 (the weights are small and shared by all).
  const int nby = p.Cout_pad / BN;
  const int slot = blockIdx.x >> 3;
  const int bx = (slot / nby) * 8 + (blockIdx.x & 7);
  if (bx * BM >= p.M) return;
  const int m0 = bx * BM;
  const int n0 = (slot % nby) * BN;

  // ---- per-lane DMA descriptors ----
  const int lr = lane >> 2;                      // row within a 16-row DMA slab
  const int pos = lane & 3;                      // 16-B slot within the 64-B LDS row
  int iy0[A_IT], ix0[A_IT], pbase[A_IT], achunk[A_IT];
#pragma unroll
  for (int q = 0; q < A_IT; ++q) {
    const int rt = (wave * A_IT + q) * 16 + lr;  // tile row
    achunk[q] = (pos ^ ((rt >> 2) & 3)) * 8;     // which 8-half chunk of the row this lane fetches
    const int m = m0 + rt;
    if (m < p.M) {
      const int hw = p.Ho * p.Wo;
      const int n = m / hw;
      const int rem = m - n * hw;
      const int oy = rem / p.Wo;
      const int ox = rem - oy * p.Wo;
      iy0[q] = oy * p.stride - p.pad_t;
      ix0[q] = ox * p.stride - p.pad_l;
      pbase[q] = n * p.H * p.W;
    } else {
      iy0[q] = -(1 << 20);
      ix0[q] = 0;
      pbase[q] = 0;
    }
  }
  size_t boff[B_IT];
#pragma unroll
  for (int q = 0; q < B_IT; ++q) {
    const int rt = (wave * B_IT + q) * 16 + lr;
    boff[q] = (size_t)(n0 + rt) * 32 + (pos ^ ((rt >> 2) & 3)) * 8;   // K-blocked weights [Kp/32][Cout_pad][32]
  }
  const int nk = p.Kp / 32;
  const int ntaps = p.KH * p.KW;

  auto issue = [&](int kt, int buf) {
    u16* Ah = smem16 + buf * STAGE;
    u16* Al = Ah + BM * ROWB;
    u16* Bh = Al + BM * ROWB;
    u16* Bl = Bh + BN * ROWB;
    // K order: channel chunk outer, filter tap inner.  The KH*KW shifted views of one 32-channel
    // slab are consumed back to back, so a multi-tap conv (3x3, 15x1, 1x15) pulls each activation
    // line over the fabric once and takes the other taps from L2; tap-outer order streamed the
    // whole [rows x Cin] slab per tap and evicted it before the next tap came round (15x the HBM
    // reads on the 2048-channel large-separable convs).
    const int cc = kt / ntaps;                   // block-uniform
    const int tap = kt - cc * ntaps;
    const int ky = tap / p.KW;
    const int dy = ky * p.dil, dx = (tap - ky * p.KW) * p.dil;
    const size_t k0 = (size_t)(tap * (p.Cin_p >> 5) + cc) * p.Cout_pad * 32;   // weight K-block
#pragma unroll
    for (int q = 0; q < A_IT; ++q) {
      const int iy = iy0[q] + dy, ix = ix0[q] + dx;
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const size_t off = ((size_t)cc * p.in_pix + (size_t)(pbase[q] + iy * p.W + ix)) * 32 + achunk[q];
      const u16* sh = ok ? p.in_hi + off : p.zeros;
      XDET_GLDS16(sh, Ah + (wave * A_IT + q) * 16 * ROWB);
      if (NSPLIT > 1) {
        const u16* sl = ok ? p.in_lo + off : p.zeros;
        XDET_GLDS16(sl, Al + (wave * A_IT + q) * 16 * ROWB);
      }
    }
#pragma unroll
    for (int q = 0; q < B_IT; ++q) {
      XDET_GLDS16(p.wt_hi + boff[q] + k0, Bh + (wave * B_IT + q) * 16 * ROWB);
      if (NSPLIT > 1) XDET_GLDS16(p.wt_lo + boff[q] + k0, Bl + (wave * B_IT + q) * 16 * ROWB);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31;
  const int fh = lane >> 5;
  // operand rows of this lane and their chunk permutation (tile-row bits 2..3)
  int aoff[TM], boffs[TN], asw[TM], bsw[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int rt = wm * WM + i * 32 + frow;
    aoff[i] = rt * ROWB;
    asw[i] = (rt >> 2) & 3;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int rt = wn * WN + j * 32 + frow;
    boffs[j] = rt * ROWB;
    bsw[j] = (rt >> 2) & 3;
  }

  // DMA instructions one wave issues per stage: the counted wait below leaves exactly the newest
  // stage in flight (LDS-DMA completions retire in order on vmcnt)
  constexpr int PER_STAGE = (A_IT + B_IT) * (NSPLIT > 1 ? 2 : 1);
  issue(0, 0);
  if (NSTAGE == 3 && nk > 1) issue(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt % NSTAGE;
    if (NSTAGE == 2) {
      __syncthreads();                           // DMA(kt) landed for every wave; the other stage is free
      if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
    } else {
      // stage kt must have landed, stage kt+1 may still be in flight; raw barrier so that nothing
      // drains the DMA queue; after it every wave has finished reading stage kt-1 == (kt+2)%3
      if (kt + 1 < nk) __builtin_amdgcn_s_waitcnt(0x0F70 | (PER_STAGE & 15) | ((PER_STAGE >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < nk) issue(kt + 2, (kt + 2) % NSTAGE);
    }
    const u16* Ah = smem16 + buf * STAGE;
    const u16* Al = Ah + BM * ROWB;
    const u16* Bh = Al + BM * ROWB;
    const u16* Bl = Bh + BN * ROWB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 2 + fh;                 // logical 8-half chunk of the 32-deep step
      f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int o = aoff[i] + ((c ^ asw[i]) << 3);
        ah[i] = *reinterpret_cast<const f16x8*>(Ah + o);
        if (NSPLIT > 1) al[i] = *reinterpret_cast<const f16x8*>(Al + o);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int o = boffs[j] + ((c ^ bsw[j]) << 3);
        bh[j] = *reinterpret_cast<const f16x8*>(Bh + o);
        if (NSPLIT > 1) bl[j] = *reinterpret_cast<const f16x8*>(Bl + o);
      }
      if (NSPLIT > 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
  }

  // Epilogue.  The MFMA accumulator layout gives each lane one column and 16 scattered rows, i.e.
  // 4-byte global stores (and residual loads).  Bounce each 32-row slab of the wave tile through
  // the (now idle) operand LDS so a lane owns 4 consecutive channels of a row: 16-B coalesced
  // residual loads and stores, 4x fewer memory instructions.
  constexpr int EP_LD = WN + 4;                  // floats per staged row (+4: keeps float4 rows 16-B aligned)
  constexpr int C4N = WN / 4;                    // float4 columns per row
  static_assert(NW * 32 * EP_LD * 4 <= NSTAGE * STAGE * 2, "epilogue staging must fit the operand LDS");
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
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ep[((r & 3) + 8 * (r >> 2) + 4 * fh) * EP_LD + j * 32 + frow] = acc[i][j][r];
#pragma unroll
    for (int q = 0; q < (32 * C4N) / 64; ++q) {
      const int row = (q * 64 + lane) / C4N;
      const float4 a = *reinterpret_cast<const float4*>(ep + row * EP_LD + c4 * 4);
      const int m = m0 + wm * WM + i * 32 + row;
      if (col_ok && m < p.M) {
        float4 v = make_float4(fmaf(a.x, sc4.x, sh4.x), fmaf(a.y, sc4.y, sh4.y), fmaf(a.z, sc4.z, sh4.z),
                               fmaf(a.w, sc4.w, sh4.w));
        if (p.res) {
          const float4 rr = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.ldr + co4);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (p.relu_out) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<float4*>(p.out + (size_t)m * p.ldo + co4) = v;
      }
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NSPLIT, int NSTAGE = 2>
static int launch_d(const ConvParams& p, hipStream_t s) {
  constexpr size_t lds = (size_t)NSTAGE * (2 * BM + 2 * BN) * 32 * sizeof(u16);
  auto kern = conv_dma_f16_kernel<BM, BN, WAVES_M, WAVES_N, NSPLIT, NSTAGE>;
  static bool attr_set = false;
  if (!attr_set) {
    XDET_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    attr_set = true;
  }
  dim3 grid((unsigned)(cdiv(cdiv(p.M, BM), 8) * 8 * (p.Cout_pad / BN)));
  hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES_M * WAVES_N), lds, s, p);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

int launch_conv_mfma_dma(const ConvParams& p, int n_tile, int nsplit, hipStream_t s) {
  XDET_REQUIRE(p.Kp % 32 == 0 && p.Cin_p % 32 == 0 && p.ldi >= p.Cin_p && p.ldi % 8 == 0,
               "conv(dma): channel counts must be padded to 32");
  XDET_REQUIRE(p.Cout_pad % n_tile == 0, "conv(dma): Cout_pad must be a multiple of the N tile");
  XDET_REQUIRE(p.in_hi && (nsplit == 1 || p.in_lo) && p.wt_hi && (nsplit == 1 || p.wt_lo) && p.zeros,
               "conv(dma): split planes missing");
  if (p.M <= 0) return XDET_OK;
  if (n_tile == 128 && nsplit == 3) {
    // Tile choice.  The kernel is bound by L2->LDS operand traffic, so the biggest tile wins as long
    // as the grid still covers the 256 CUs (measured on MI355X, tools/conv_bench.py --planes):
    // 256x256 once there are >= 2 workgroups per CU, or ~1 per CU with a long K loop to amortise its
    // prologue/epilogue; 256x128 from ~2/3 workgroup per CU; else 128x128 (two workgroups share a CU).
    static const char* tile_env = getenv("XDET_TILE");   // experiment override: 128x128 | 256x128 | 256x256
    const int64_t b256 = cdiv(p.M, 256) * (p.Cout_pad / 256), b128n = cdiv(p.M, 256) * (p.Cout_pad / 128);
    const int nk = p.Kp / 32;
    int tile = 0;
    static const char* t256 = getenv("XDET_T256");
    const int64_t thr256 = t256 ? atoi(t256) : 512;
    // ...or when the 256x256 grid fills whole rounds of the 256 CUs (within 6 %)
    const bool full_rounds = b256 >= 240 && (b256 % 256 == 0 || b256 % 256 >= 240);
    if (p.Cout_pad % 256 == 0 && (b256 >= thr256 || full_rounds || (b256 >= 200 && nk >= 40))) tile = 2;
    else if (b128n >= 170) tile = 1;
    if (tile_env) tile = !strcmp(tile_env, "256x256") ? (p.Cout_pad % 256 == 0 ? 2 : 1) : !strcmp(tile_env, "256x128") ? 1 : 0;
    static const char* w4_env = getenv("XDET_W4");
    if (tile == 2 && w4_env && w4_env[0] == '1') return launch_d<256, 256, 2, 2, 3>(p, s);
    if (tile == 2) return launch_d<256, 256, 2, 4, 3>(p, s);
    static const char* st_env = getenv("XDET_STAGES");
    const bool s3 = st_env && st_env[0] == '3';
    if (tile == 1) return s3 ? launch_d<256, 128, 4, 2, 3, 3>(p, s) : launch_d<256, 128, 4, 2, 3>(p, s);
    if (s3) return launch_d<128, 128, 2, 2, 3, 3>(p, s);
  }
  if (n_tile == 128)
    return nsplit == 1 ? launch_d<128, 128, 2, 2, 1>(p, s) : launch_d<128, 128, 2, 2, 3>(p, s);
  if (n_tile == 64) return nsplit == 1 ? launch_d<128, 64, 4, 1, 1>(p, s) : launch_d<128, 64, 4, 1, 3>(p, s);
  set_last_error("conv(dma): unsupported N tile");
  return XDET_ERR_UNSUPPORTED;
}

}  // namespace xdet
