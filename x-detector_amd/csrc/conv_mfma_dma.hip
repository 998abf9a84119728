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
//   * both operands are K-blocked in HBM -- activations [pixels/16][C/32][16][32], weights
//     [Kp/32][Cout_pad][32] -- so the 16 rows x 64 B one DMA instruction moves are one contiguous
//     1 KB run (tools/ubench/dma_rate.hip: 63 GB/s per CU for 1 KB runs vs 30 GB/s for 64 B segments
//     at a row stride when the stream misses L2);
//   * two LDS stages (64 KB for a 128x128 tile -> two workgroups per CU), one barrier per 32-deep K
//     step placed between its two 16-deep halves, fragments double-buffered in registers, the DMA
//     pieces of step k+2 interleaved with the MFMAs of step k (see the main loop).
//
// Arithmetic is identical to conv_mfma_split.hip (a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, f32
// accumulate, per-channel power-of-two weight pre-scale folded into the epilogue).
#include "common.h"
#include "conv_epilogue.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace xdet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int x8frag __attribute__((ext_vector_type(8)));     // 32 fp8 of one row: a lane's operand of the 32x32x64 block-scaled MFMA
typedef int x8half __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#define XDET_GLDS16(gptr, lptr)                                                                        \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),              \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// PW (pointwise: 1x1, stride 1, no padding -- 30 of the detector's 46 contractions, all of the dominant ones): the DMA
// sources are raw BUFFER loads whose per-lane offsets never change along K -- a 16-pixel x 32-channel block of the planes
// and a 16-row x 32-deep block of the K-blocked weights are both 1 KB runs, 1 KB further per K step -- so a piece is an
// M0 update plus one `buffer_load_dwordx4 ... lds` with the step's offset in an SGPR: no per-lane address arithmetic, no
// zero-page select (rows past M are sent out of bounds and read zeros).  The generic form spends ~55 VALU and ~70 SALU
// instructions per K step on addresses next to its 48 MFMAs, and every issue slot between two MFMAs costs matrix-pipe time
// (MI355X_MICROARCH.md).  Same bytes into the same LDS places: results are bit-identical (tests/test_gpu_layers.py).
// FOLD: the layer's reduction is DEFINED with a fixed split (p.ksplit ranges of whole channel chunks, result = the left
// fold of the per-range sums: conv_mfma_ksplit.hip, which runs the ranges on separate workgroups when the grid is small).
// At large batch this kernel walks all ranges in its one K pipeline and folds its accumulators at the range boundaries:
// the same expression tree, the same bits, without scratch traffic -- on the 256 x 128 tile, whose 64 accumulator registers
// per wave leave room for the running total.
// X8 (pointwise form only): the cross terms a_hi*w_lo + a_lo*w_hi come from fp8 copies of the operands (conv_params.h) through
// one v_mfma_scale_f32_32x32x64_f8f6f4 per accumulator block and 32-deep K step -- its 64-deep K is [hi8 | lo8] of the
// activations against [w_lo8 | w_hi8], a lane's scale byte the power-of-two factor of exactly its 32-deep half
// (tools/ubench/mfma_scale_semantics.hip) -- instead of four f16 MFMAs: 2 + 1 instead of 6 matrix instructions per block and
// step.  Per accumulator the order is hi*hi (first 16 channels), hi*hi (second 16), cross -- in every tile shape, so results do
// not depend on the kernel a batch size selects.
// GBUF (round 5; multi-tap layers whose planes and weights stay below 4 GiB -- the RPN 3x3 conv, the direct large-separable
// convs): the generic form's DMA sources as raw buffer loads too.  A tap's window position still has to be computed per
// lane and K step (the blocked plane layout is not linear in the pixel), but as ONE 32-bit offset -- out-of-image taps are
// an out-of-range offset (zero fill) instead of a select between a 64-bit address and the zero page, the channel block and
// the weights' K block ride in the scalar offset, and (cc, ky, kx) advance incrementally instead of by division: ~22 VALU
// + ~25 SALU per K step next to the 48 MFMAs where the pointer form spends ~55 + ~70.  Same bytes into the same LDS
// places: bit-identical.
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSPLIT, int NSTAGE, bool PW = false, bool FOLD = false, bool X8 = false,
          bool GBUF = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_dma_f16_kernel(ConvParams p_in) {
  static_assert(NSTAGE == 2, "two LDS stages");
  static_assert(!(GBUF && PW), "GBUF is the buffer form of the NON-pointwise layers");
  static_assert(!X8 || (PW && NSPLIT == 3 && !FOLD), "x8: pointwise f16x3 layers only");
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
  const int nby = p_in.Cout_pad / BN;
  const int slot = blockIdx.x >> 3;
  int bx, n0;
  ConvParams p = p_in;
  if (p_in.group_rows) {
    // Grouped GEMM (frequency bins): every group has its own weight matrix, so ALL tiles of a group go to
    // ONE XCD -- its L2 then holds that group's weights and activations once, instead of eight L2s each
    // pulling every group's weights over the fabric.  Workgroups are dealt round-robin to the XCDs:
    // blockIdx & 7 is the XCD, group = 8 * round + XCD, tiles of the group in slot order (N fastest).
    const int mt = p_in.group_rows / BM;           // M tiles per group
    const int tpg = mt * nby;
    const int g = (slot / tpg) * 8 + (blockIdx.x & 7);
    const int w = slot - (slot / tpg) * tpg;
    if ((int64_t)g * p_in.group_rows >= p_in.M) return;
    if (p_in.group_live_rows && (w / nby) * BM >= p_in.group_live_rows) return;     // a tile of padding rows only
    bx = g * mt + w / nby;
    n0 = (w % nby) * BN;
    p.wt_hi = p_in.wt_hi + (size_t)g * p_in.group_wt_stride;
    p.wt_lo = p_in.wt_lo ? p_in.wt_lo + (size_t)g * p_in.group_wt_stride : nullptr;
    p.scale = p_in.scale + (size_t)g * p_in.Cout_pad;
    p.shift = p_in.shift + (size_t)g * p_in.Cout_pad;
  } else {
    bx = (slot / nby) * 8 + (blockIdx.x & 7);
    if (bx * BM >= p_in.M) return;
    n0 = (slot % nby) * BN;
  }
  const int m0 = bx * BM;

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
  const unsigned c32n = (unsigned)(p.ldi >> 5);   // channel blocks per 16-pixel group of a split plane

  // PW: buffer resources over the whole planes / weight tensors (host: each below 4 GiB) and the K-invariant per-lane
  // byte offsets of this wave's pieces
  __amdgpu_buffer_rsrc_t r_ah, r_al, r_bh, r_bl;
  unsigned a_vo[A_IT], b_vo[B_IT];
  int g_cc = 0, g_ky = 0, g_kx = 0, g_tap = 0;   // GBUF: (channel block, tap) of the NEXT issue() -- calls come with kt = 0, 1, 2, ...
  if (GBUF) {
    const unsigned a_bytes = (unsigned)(((size_t)(((size_t)p.N * p.H * p.W + 15) >> 4) * c32n) << 10);
    const unsigned b_bytes = (unsigned)((size_t)nk * p.Cout_pad * 64);
    r_ah = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.in_hi), 0, (int)a_bytes, 0x00020000);
    r_al = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(NSPLIT > 1 ? p.in_lo : p.in_hi), 0, (int)a_bytes, 0x00020000);
    r_bh = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.wt_hi), 0, (int)b_bytes, 0x00020000);
    r_bl = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(NSPLIT > 1 ? p.wt_lo : p.wt_hi), 0, (int)b_bytes, 0x00020000);
#pragma unroll
    for (int q = 0; q < B_IT; ++q) b_vo[q] = (unsigned)(boff[q] * 2);
  }
  if (PW) {
    const unsigned a_bytes = (unsigned)(((size_t)((p.M + 15) >> 4) * c32n) << 10);
    const unsigned b_bytes = (unsigned)((size_t)nk * p.Cout_pad * 64);
    r_ah = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.in_hi), 0, (int)a_bytes, 0x00020000);
    r_al = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(NSPLIT > 1 ? p.in_lo : p.in_hi), 0, (int)a_bytes, 0x00020000);
    r_bh = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.wt_hi), 0, (int)b_bytes, 0x00020000);
    r_bl = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(NSPLIT > 1 ? p.wt_lo : p.wt_hi), 0, (int)b_bytes, 0x00020000);
#pragma unroll
    for (int q = 0; q < A_IT; ++q) {
      const int rt = (wave * A_IT + q) * 16 + lr;
      const int m = m0 + rt;                       // input pixel == output pixel
      a_vo[q] = m < p.M ? (((unsigned)(m >> 4) * c32n) << 10) + (unsigned)((((m & 15) << 5) + achunk[q]) * 2) : 0xffffffffu;
    }
#pragma unroll
    for (int q = 0; q < B_IT; ++q) b_vo[q] = (unsigned)(boff[q] * 2);
  }

  auto issue = [&](int kt, int buf) {
    u16* Ah = smem16 + buf * STAGE;
    u16* Al = Ah + BM * ROWB;
    u16* Bh = Al + BM * ROWB;
    u16* Bl = Bh + BN * ROWB;
    if (PW) {
      const unsigned a_so = (unsigned)kt << 10;                      // channel block kt of every 16-pixel group
      const unsigned b_so = (unsigned)kt * (unsigned)p.Cout_pad * 64u;   // K block kt of the weights
#pragma unroll
      for (int q = 0; q < A_IT; ++q) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_ah, (__attribute__((address_space(3))) void*)(Ah + (wave * A_IT + q) * 16 * ROWB),
                                                 16, (int)a_vo[q], (int)a_so, 0, 0);
        if (NSPLIT > 1)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r_al, (__attribute__((address_space(3))) void*)(Al + (wave * A_IT + q) * 16 * ROWB),
                                                   16, (int)a_vo[q], (int)a_so, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < B_IT; ++q) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_bh, (__attribute__((address_space(3))) void*)(Bh + (wave * B_IT + q) * 16 * ROWB),
                                                 16, (int)b_vo[q], (int)b_so, 0, 0);
        if (NSPLIT > 1)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r_bl, (__attribute__((address_space(3))) void*)(Bl + (wave * B_IT + q) * 16 * ROWB),
                                                   16, (int)b_vo[q], (int)b_so, 0, 0);
      }
    } else if (GBUF) {
      // (same K order as the pointer form below: channel block outer, tap inner)
      const int dy = g_ky * p.dil, dx = g_kx * p.dil;
      const unsigned a_so = (unsigned)g_cc << 10;
      const unsigned b_so = (unsigned)(g_tap * (p.Cin_p >> 5) + g_cc) * (unsigned)p.Cout_pad * 64u;
#pragma unroll
      for (int q = 0; q < A_IT; ++q) {
        const int iy = iy0[q] + dy, ix = ix0[q] + dx;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const unsigned pix = (unsigned)(pbase[q] + iy * p.W + ix);
        const unsigned vo = ok ? (((pix >> 4) * c32n) << 10) + (((pix & 15) << 5) + (unsigned)achunk[q]) * 2u : 0xffffffffu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_ah, (__attribute__((address_space(3))) void*)(Ah + (wave * A_IT + q) * 16 * ROWB),
                                                 16, (int)vo, (int)a_so, 0, 0);
        if (NSPLIT > 1)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r_al, (__attribute__((address_space(3))) void*)(Al + (wave * A_IT + q) * 16 * ROWB),
                                                   16, (int)vo, (int)a_so, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < B_IT; ++q) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_bh, (__attribute__((address_space(3))) void*)(Bh + (wave * B_IT + q) * 16 * ROWB),
                                                 16, (int)b_vo[q], (int)b_so, 0, 0);
        if (NSPLIT > 1)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r_bl, (__attribute__((address_space(3))) void*)(Bl + (wave * B_IT + q) * 16 * ROWB),
                                                   16, (int)b_vo[q], (int)b_so, 0, 0);
      }
      ++g_tap;
      if (++g_kx == p.KW) {
        g_kx = 0;
        if (++g_ky == p.KH) { g_ky = 0; g_tap = 0; ++g_cc; }
      }
    } else {
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
      const unsigned pix = (unsigned)(pbase[q] + iy * p.W + ix);
      const size_t off = ((size_t)((pix >> 4) * c32n + cc) << 9) + ((pix & 15) << 5) + achunk[q];
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

  // Main loop.  Fragments are double-buffered in registers and the single barrier of a K step sits
  // BETWEEN its two 16-deep halves, so the matrix pipe never waits for LDS across the barrier:
  //     ds_read  half 1 of stage kt        -> set B      (lands under the MFMAs below)
  //     MFMA     half 0 of stage kt        (set A, read before the previous barrier)
  //     barrier: every wave's DMA(kt+1) has landed AND every wave is done reading stage kt
  //     DMA      stage kt+2 -> the buffer stage kt just vacated (a whole step to land)
  //     ds_read  half 0 of stage kt+1      -> set A      (lands under the MFMAs below)
  //     MFMA     half 1 of stage kt        (set B)
  // FOLD state: steps per range = (channel chunks per range) x taps, as conv_mfma_ksplit.hip defines the ranges
  f32x16 tot[FOLD ? TM : 1][FOLD ? TN : 1];
  int fold_steps = 1 << 30, next_fold = 1 << 30;
  bool first_range = true;
  if constexpr (FOLD) {
    const int ncc = p.Cin_p >> 5, S = p.ksplit > 0 ? p.ksplit : 1;
    fold_steps = ((ncc + S - 1) / S) * ntaps;
    next_fold = fold_steps;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
  }
  // JN = live 32-column blocks of this wave's tile (TN, or fewer in the last N tile of a layer whose channel count is
  // padded up to the tile: 728 -> 768 leaves the 24th block of 28 layers all zero).  A wave whose last block is padding
  // skips its MFMAs and fragment reads -- same results (the block is never stored), 1/24 of those layers' matrix work.
  auto main_loop = [&](auto JN_) {
  constexpr int JN = decltype(JN_)::value;
  auto load_frags = [&](int buf, int ks, f16x8* ah, f16x8* al, f16x8* bh, f16x8* bl) {
    const u16* Ah = smem16 + buf * STAGE;
    const u16* Al = Ah + BM * ROWB;
    const u16* Bh = Al + BM * ROWB;
    const u16* Bl = Bh + BN * ROWB;
    const int c = ks * 2 + fh;                   // logical 8-half chunk of the 32-deep step
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int o = aoff[i] + ((c ^ asw[i]) << 3);
      ah[i] = *reinterpret_cast<const f16x8*>(Ah + o);
      if constexpr (NSPLIT > 1 && !X8) al[i] = *reinterpret_cast<const f16x8*>(Al + o);
    }
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int o = boffs[j] + ((c ^ bsw[j]) << 3);
      bh[j] = *reinterpret_cast<const f16x8*>(Bh + o);
      if constexpr (NSPLIT > 1 && !X8) bl[j] = *reinterpret_cast<const f16x8*>(Bl + o);
    }
  };
  // x8: this lane's 32 fp8 of a row's 64-byte record -- the hi8 half for lanes 0..31, the lo8 half for lanes 32..63 (the
  // instruction's K split) -- are its 16-byte chunks 2*fh and 2*fh + 1 (chunk-permuted like every row of the stage)
  auto load_x8 = [&](int buf, x8frag* a8, x8frag* b8) {
    const u16* Al = smem16 + buf * STAGE + BM * ROWB;
    const u16* Bl = Al + BM * ROWB + BN * ROWB;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const x8half q0 = *reinterpret_cast<const x8half*>(Al + aoff[i] + (((2 * fh) ^ asw[i]) << 3));
      const x8half q1 = *reinterpret_cast<const x8half*>(Al + aoff[i] + (((2 * fh + 1) ^ asw[i]) << 3));
      a8[i] = x8frag{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
    }
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const x8half q0 = *reinterpret_cast<const x8half*>(Bl + boffs[j] + (((2 * fh) ^ bsw[j]) << 3));
      const x8half q1 = *reinterpret_cast<const x8half*>(Bl + boffs[j] + (((2 * fh + 1) ^ bsw[j]) << 3));
      b8[j] = x8frag{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
    }
  };
  // E8M0 scale bytes of this lane's 32-deep half: activations hi8 * 2^e / lo8 * 2^(e - 11), weights w_lo8 * 2^-9 / w_hi8 * 2^2
  const int x8_sa = X8 ? 127 + p.x8_exp - (fh ? 11 : 0) : 0;
  const int x8_sb = fh ? 127 + 2 : 127 - 9;
  auto mma_x8 = [&](const x8frag* a8, const x8frag* b8) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < JN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], acc[i][j], 0, 0, 0, x8_sa, 0, x8_sb);
  };
  auto mma = [&](const f16x8* ah, const f16x8* al, const f16x8* bh, const f16x8* bl) {
    if constexpr (NSPLIT > 1 && !X8) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < JN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
  };

  f16x8 a0h[TM], a0l[X8 ? 1 : TM], b0h[TN], b0l[X8 ? 1 : TN];      // set A: half 0 of the current stage
  f16x8 a1h[TM], a1l[X8 ? 1 : TM], b1h[TN], b1l[X8 ? 1 : TN];      // set B: half 1
  x8frag a8[X8 ? TM : 1], b8[X8 ? TN : 1];                         // x8: the cross-term operands of the whole 32-deep step
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                               // stage 0 landed
  if (nk > 1) issue(1, 1);
  load_frags(0, 0, a0h, a0l, b0h, b0l);
  // one K step; STEADY = both the DMA two stages ahead and the next stage's fragments exist, so the
  // whole second half is one basic block and its DMA pieces / fragment reads can be interleaved with
  // the MFMAs (a piece costs ~60 cycles of issue among MFMAs, 100-185 in a burst of eight next to the
  // fragment reads -- MI355X_MICROARCH.md)
  auto step = [&](int kt, auto steady) {
    constexpr bool STEADY = decltype(steady)::value;
    const int buf = kt & 1;
    if constexpr (FOLD) {
      if (kt == next_fold) {                     // range boundary: every MFMA of the range before it has been issued
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            tot[i][j] = first_range ? acc[i][j] : tot[i][j] + acc[i][j];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
          }
        first_range = false;
        next_fold += fold_steps;
      }
    }
    load_frags(buf, 1, a1h, a1l, b1h, b1l);
    if constexpr (X8) load_x8(buf, a8, b8);
    __builtin_amdgcn_sched_barrier(0);
    mma(a0h, a0l, b0h, b0l);
    __builtin_amdgcn_sched_barrier(0);
    // explicit: that DMA(kt+1) has landed must not depend on the compiler's own tracking of LDS-DMA writes against later
    // LDS reads -- it put the vmcnt(0) here by itself in this kernel and left it out of a restructured copy of the loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                             // (+ lgkmcnt(0): set B landed) every wave is done with stage kt
    // fragment reads first: the compiler orders LDS reads against the DMA's LDS writes, so reads
    // placed after issue() could not move up between the DMA pieces
    if (STEADY || kt + 1 < nk) load_frags(buf ^ 1, 0, a0h, a0l, b0h, b0l);
    if (STEADY || kt + 2 < nk) issue(kt + 2, buf);
    if (!STEADY) __builtin_amdgcn_sched_barrier(0);
    mma(a1h, a1l, b1h, b1l);
    if constexpr (X8) mma_x8(a8, b8);
    if (STEADY) {
      constexpr int NPIECE = (A_IT + B_IT) * (NSPLIT > 1 ? 2 : 1);
      constexpr int NMMA = TM * JN * (X8 ? 2 : NSPLIT > 1 ? 3 : 1);
      constexpr int NRD = (TM + JN) * (NSPLIT > 1 && !X8 ? 2 : 1);
      __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
#pragma unroll
      for (int k = 0; k < NPIECE; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, NMMA / NPIECE > 0 ? NMMA / NPIECE : 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  int kt = 0;
  for (; kt + 2 < nk; ++kt) step(kt, std::true_type{});
  for (; kt < nk; ++kt) step(kt, std::false_type{});
  };
  // live column blocks: columns at or beyond ldo (= round_up(cout, 32)) are padding of the N tile
  const int jn_live = (p.ldo - (n0 + wn * WN) + 31) >> 5;      // wave-uniform
  static_assert(TN == 2, "the tiles of this kernel are two column blocks per wave");
  // (pointwise form only: the generic form has no registers to spare for a second copy of the loop's address state)
  if constexpr (PW) {
    if (jn_live == 1 && p.skip_dead != 0) main_loop(std::integral_constant<int, 1>{});
    else main_loop(std::integral_constant<int, TN>{});
  } else {
    main_loop(std::integral_constant<int, TN>{});
  }

  if constexpr (FOLD) {
    if (!first_range) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = tot[i][j] + acc[i][j];
    }
  }
  // (all rows below M -- every tile but a ragged last one: the form without per-row predicates, conv_epilogue.h)
  if (m0 + BM <= p.M) conv_epilogue_full<WM, WN, TM, TN, NW, NSTAGE * STAGE * 2>(p, acc, smem16, wave, lane, wm, wn, m0, n0);
  else conv_epilogue<WM, WN, TM, TN, NW, NSTAGE * STAGE * 2>(p, acc, smem16, wave, lane, wm, wn, m0, n0);
}

// ---------------------------------------------------------------------------------------
// Small-M variant (single images and small batches: a 900-row layer is 8 x 12 workgroups of 128 x 64).  With that few
// workgroups nothing hides a K step's latency but the step pipeline itself, and the two-stage loop above spends ~1.1 us
// per 32-deep step (DMA issue -> landed -> barrier -> fragments -> 12 MFMAs) for 0.19 us of matrix work.  Here the
// operand ring has SIX stages, a barrier covers a PAIR of K steps and two more pairs are in flight: the barrier waits with
// vmcnt(one pair's worth) for the pair it is about to read only, so a step costs its own issue + fragment reads + MFMAs
// and half a barrier.  The fragment reads are inline asm: the compiler would put a full vmcnt(0) in front of every LDS
// read while LDS-DMA writes are outstanding.  Every pair issues the same twelve DMA instructions (steps past the end
// fetch the zero page: zero operands add nothing) so that the vmcnt arithmetic is static.  Same K order, same product
// order: bit-identical to conv_dma_f16_kernel.  Batch-1 latency 1.90 -> 1.74 ms (four stages, one step per barrier)
// -> 1.70 ms.
// ---------------------------------------------------------------------------------------
template <int OFF>
__device__ __forceinline__ f16x8 cd_ds_read_b128(unsigned addr) {
  f16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}

// W8 (round 5): eight waves as 4 x 2 wave tiles of 32 x 32 -- two waves per SIMD, each with one accumulator block, three DMA pieces
// (16 rows of A hi and lo, one 16-row group of one weight plane) and half the fragment reads of a step: with one wave per SIMD
// nothing runs under a wave's DMA issue and fragment waits (the split-K ring kernel gained 4 % on the ResNet trunk from the same
// change).  Same K order and product order per accumulator: bit-identical.
// H64 (round 5): 64 x 64 tiles on four waves (2 x 2 wave tiles of 32 x 32) for grids that leave more than half of the CUs without a
// 128 x 64 tile (a single image's middle flow: 900 rows x 768 columns = 96 tiles -> 180): a workgroup's K loop runs at the rate its CU
// takes operand bytes in (~20 B/clk: 24.6 KB per 128 x 64 step), so spreading the same reduction over more CUs with less per step
// (16 KB) shortens it.  Same K order and product order per accumulator: bit-identical.
template <int NSPLIT, bool PW = false, bool W8 = false, bool H64 = false>
__global__ __launch_bounds__(W8 ? 512 : 256) void conv_dma_deep_kernel(ConvParams p_in) {
  constexpr int BM = H64 ? 64 : 128, BN = 64, NW = W8 ? 8 : 4;
  static_assert(!W8 || NSPLIT == 3, "eight waves: the f16x3 form");
  static_assert(!(W8 && H64), "64 x 64 tiles: four waves");
  constexpr bool WT = W8 || H64;                   // wave tiles of 32 x 32 in two columns (else 32 x 64 in one)
  constexpr int PAIR = 2;                          // K steps per barrier
  constexpr int NSTAGE = 3 * PAIR;                 // ring: the pair being read + two pairs in flight
  constexpr int TN = WT ? 1 : 2;                   // wave tile 32 x 64 (eight waves / 64 x 64 tiles: 32 x 32)
  constexpr int A_IT = BM / (16 * NW), B_IT = W8 ? 1 : BN / (16 * NW);   // (eight waves: ONE weight piece per wave, hi or lo)
  constexpr int ROWB = 32;
  constexpr int STAGE = (2 * BM + 2 * BN) * ROWB;  // halves per stage (24 KB)
  constexpr int PIECES = W8 ? 3 : (A_IT + B_IT) * (NSPLIT > 1 ? 2 : 1);
  extern __shared__ __attribute__((aligned(16))) u16 smem16[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nby = p_in.Cout_pad / BN;
  const int slot = blockIdx.x >> 3;
  int bx, n0;
  ConvParams p = p_in;
  if (p_in.group_rows) {                           // grouped GEMM: see conv_dma_f16_kernel
    const int mt = p_in.group_rows / BM;
    const int tpg = mt * nby;
    const int g = (slot / tpg) * 8 + (blockIdx.x & 7);
    const int w = slot - (slot / tpg) * tpg;
    if ((int64_t)g * p_in.group_rows >= p_in.M) return;
    if (p_in.group_live_rows && (w / nby) * BM >= p_in.group_live_rows) return;     // a tile of padding rows only
    bx = g * mt + w / nby;
    n0 = (w % nby) * BN;
    p.wt_hi = p_in.wt_hi + (size_t)g * p_in.group_wt_stride;
    p.wt_lo = p_in.wt_lo ? p_in.wt_lo + (size_t)g * p_in.group_wt_stride : nullptr;
    p.scale = p_in.scale + (size_t)g * p_in.Cout_pad;
    p.shift = p_in.shift + (size_t)g * p_in.Cout_pad;
  } else {
    bx = (slot / nby) * 8 + (blockIdx.x & 7);
    if (bx * BM >= p_in.M) return;
    n0 = (slot % nby) * BN;
  }
  const int m0 = bx * BM;

  const int lr = lane >> 2, pos = lane & 3;
  int iy0[A_IT], ix0[A_IT], pbase[A_IT], achunk[A_IT];
#pragma unroll
  for (int q = 0; q < A_IT; ++q) {
    const int rt = (wave * A_IT + q) * 16 + lr;
    achunk[q] = (pos ^ ((rt >> 2) & 3)) * 8;
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
    const int rt = (W8 ? (wave >> 1) : wave * B_IT + q) * 16 + lr;
    boff[q] = (size_t)(n0 + rt) * 32 + (pos ^ ((rt >> 2) & 3)) * 8;
  }
  const int nk = p.Kp / 32;
  const int ntaps = p.KH * p.KW;
  const unsigned c32n = (unsigned)(p.ldi >> 5);

  // PW (1x1, stride 1): raw buffer loads with K-invariant per-lane offsets, see conv_dma_f16_kernel
  unsigned a_vo[A_IT], b_vo[B_IT];
  unsigned a_bytes = 0, b_bytes = 0;
  if (PW) {
    a_bytes = (unsigned)(((size_t)((p.M + 15) >> 4) * c32n) << 10);
    b_bytes = (unsigned)((size_t)nk * p.Cout_pad * 64);
#pragma unroll
    for (int q = 0; q < A_IT; ++q) {
      const int rt = (wave * A_IT + q) * 16 + lr;
      const int m = m0 + rt;
      a_vo[q] = m < p.M ? (((unsigned)(m >> 4) * c32n) << 10) + (unsigned)((((m & 15) << 5) + achunk[q]) * 2) : 0xffffffffu;
    }
#pragma unroll
    for (int q = 0; q < B_IT; ++q) b_vo[q] = (unsigned)(boff[q] * 2);
  }

  auto issue = [&](int kt, int buf) {              // always PIECES instructions
    u16* Ah = smem16 + buf * STAGE;
    u16* Al = Ah + BM * ROWB;
    u16* Bh = Al + BM * ROWB;
    u16* Bl = Bh + BN * ROWB;
    const bool live = kt < nk;
    if (PW) {
      // a stage past the end reads a zero-length buffer: every lane out of bounds, zeros
      const __amdgpu_buffer_rsrc_t r_ah = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.in_hi), 0, live ? (int)a_bytes : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t r_al = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(NSPLIT > 1 ? p.in_lo : p.in_hi), 0, live ? (int)a_bytes : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t r_bh = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.wt_hi), 0, live ? (int)b_bytes : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t r_bl = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(NSPLIT > 1 ? p.wt_lo : p.wt_hi), 0, live ? (int)b_bytes : 0, 0x00020000);
      const unsigned a_so = (unsigned)kt << 10, b_so = (unsigned)kt * (unsigned)p.Cout_pad * 64u;
#pragma unroll
      for (int q = 0; q < A_IT; ++q) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_ah, (__attribute__((address_space(3))) void*)(Ah + (wave * A_IT + q) * 16 * ROWB), 16, (int)a_vo[q], (int)a_so, 0, 0);
        if (NSPLIT > 1)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r_al, (__attribute__((address_space(3))) void*)(Al + (wave * A_IT + q) * 16 * ROWB), 16, (int)a_vo[q], (int)a_so, 0, 0);
      }
      if constexpr (W8) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds((wave & 1) ? r_bl : r_bh, (__attribute__((address_space(3))) void*)(((wave & 1) ? Bl : Bh) + (wave >> 1) * 16 * ROWB), 16,
                                                 (int)b_vo[0], (int)b_so, 0, 0);
      } else {
#pragma unroll
      for (int q = 0; q < B_IT; ++q) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_bh, (__attribute__((address_space(3))) void*)(Bh + (wave * B_IT + q) * 16 * ROWB), 16, (int)b_vo[q], (int)b_so, 0, 0);
        if (NSPLIT > 1)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r_bl, (__attribute__((address_space(3))) void*)(Bl + (wave * B_IT + q) * 16 * ROWB), 16, (int)b_vo[q], (int)b_so, 0, 0);
      }
      }
    } else {
    const int cc = kt / ntaps;
    const int tap = kt - cc * ntaps;
    const int ky = tap / p.KW;
    const int dy = ky * p.dil, dx = (tap - ky * p.KW) * p.dil;
    const size_t k0 = (size_t)(tap * (p.Cin_p >> 5) + cc) * p.Cout_pad * 32;
#pragma unroll
    for (int q = 0; q < A_IT; ++q) {
      const int iy = iy0[q] + dy, ix = ix0[q] + dx;
      const bool ok = live && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const unsigned pix = (unsigned)(pbase[q] + iy * p.W + ix);
      const size_t off = ((size_t)((pix >> 4) * c32n + cc) << 9) + ((pix & 15) << 5) + achunk[q];
      XDET_GLDS16(ok ? p.in_hi + off : p.zeros, Ah + (wave * A_IT + q) * 16 * ROWB);
      if (NSPLIT > 1) XDET_GLDS16(ok ? p.in_lo + off : p.zeros, Al + (wave * A_IT + q) * 16 * ROWB);
    }
    if constexpr (W8) {
      XDET_GLDS16(live ? ((wave & 1) ? p.wt_lo : p.wt_hi) + boff[0] + k0 : p.zeros, ((wave & 1) ? Bl : Bh) + (wave >> 1) * 16 * ROWB);
    } else {
#pragma unroll
    for (int q = 0; q < B_IT; ++q) {
      XDET_GLDS16(live ? p.wt_hi + boff[q] + k0 : p.zeros, Bh + (wave * B_IT + q) * 16 * ROWB);
      if (NSPLIT > 1) XDET_GLDS16(live ? p.wt_lo + boff[q] + k0 : p.zeros, Bl + (wave * B_IT + q) * 16 * ROWB);
    }
    }
    }
  };

  f32x16 acc[1][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

  const int frow = lane & 31, fh = lane >> 5;
  // byte offsets of this lane's fragments inside a stage: [ks] for A (row wave*32 + frow), [ks][j] for B
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(smem16);
  unsigned a_off[2], b_off[2][TN];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int c = ks * 2 + fh;
    const int ra = (WT ? (wave >> 1) : wave) * 32 + frow;
    a_off[ks] = (unsigned)(ra * ROWB + ((c ^ ((ra >> 2) & 3)) << 3)) * 2u;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int rb = (WT ? (wave & 1) : j) * 32 + frow;
      b_off[ks][j] = (unsigned)(2 * BM * ROWB + rb * ROWB + ((c ^ ((rb >> 2) & 3)) << 3)) * 2u;
    }
  }

#pragma unroll
  for (int q = 0; q < 2 * PAIR; ++q) issue(q, q);
  int ring = 0;                                    // ring ring of the pair's first stage
  for (int kt = 0; kt < nk; kt += PAIR) {
    // this pair of stages has landed (the two younger pairs may still be in flight) and every wave is done with the
    // pair before it, whose rings the new DMA overwrites
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PAIR * PIECES) : "memory");
    {
      const int ns = ring + 2 * PAIR >= NSTAGE ? ring + 2 * PAIR - NSTAGE : ring + 2 * PAIR;
#pragma unroll
      for (int q = 0; q < PAIR; ++q) issue(kt + 2 * PAIR + q, ns + q);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < PAIR; ++q) {
      const unsigned sb = lds0 + (unsigned)((ring + q) * STAGE * 2);
      f16x8 ah[2], al[2], bh[2][TN], bl[2][TN];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        ah[ks] = cd_ds_read_b128<0>(sb + a_off[ks]);
        if (NSPLIT > 1) al[ks] = cd_ds_read_b128<BM * ROWB * 2>(sb + a_off[ks]);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[ks][j] = cd_ds_read_b128<0>(sb + b_off[ks][j]);
          if (NSPLIT > 1) bl[ks][j] = cd_ds_read_b128<BN * ROWB * 2>(sb + b_off[ks][j]);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (NSPLIT > 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[ks]), "+v"(al[ks])::"memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[ks])::"memory");
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          asm volatile("" : "+v"(bh[ks][j]));
          if (NSPLIT > 1) asm volatile("" : "+v"(bl[ks][j]));
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (NSPLIT > 1) {
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh[ks][j], acc[0][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl[ks][j], acc[0][j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh[ks][j], acc[0][j], 0, 0, 0);
      }
    }
    ring = ring + PAIR >= NSTAGE ? ring + PAIR - NSTAGE : ring + PAIR;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy tail steps write LDS too
  if (m0 + BM <= p.M) conv_epilogue_full<32, 32 * TN, 1, TN, NW, NSTAGE * STAGE * 2>(p, acc, smem16, wave, lane, WT ? (wave >> 1) : wave, WT ? (wave & 1) : 0, m0, n0);
  else conv_epilogue<32, 32 * TN, 1, TN, NW, NSTAGE * STAGE * 2>(p, acc, smem16, wave, lane, WT ? (wave >> 1) : wave, WT ? (wave & 1) : 0, m0, n0);
}

// pointwise layers whose planes and weights stay below 4 GiB (32-bit buffer offsets) take the buffer-load form
static bool pw_eligible(const ConvParams& p) {
  if (p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad_t != 0 || p.pad_l != 0 || p.H != p.Ho || p.W != p.Wo) return false;
  const size_t a_bytes = ((size_t)((p.M + 15) >> 4) * (size_t)(p.ldi >> 5)) << 10;
  const size_t b_bytes = (size_t)(p.Kp / 32) * p.Cout_pad * 64;
  return a_bytes < ((size_t)1 << 32) && b_bytes < ((size_t)1 << 32) && p.Cin_p == p.Kp;
}

template <int NSPLIT, bool PW = false, bool W8 = false, bool H64 = false>
static int launch_deep(const ConvParams& p, hipStream_t s) {
  if (!PW && pw_eligible(p)) return launch_deep<NSPLIT, true, W8, H64>(p, s);
  if constexpr (!W8 && !H64 && NSPLIT == 3) {
    // 128 x 64 tiles on at most half of the CUs: 64 x 64 tiles; else eight waves per workgroup
    if (!p.group_rows && cdiv(p.M, 128) * (p.Cout_pad / 64) <= 128) return launch_deep<NSPLIT, PW, false, true>(p, s);
    return launch_deep<NSPLIT, PW, true>(p, s);
  }
  constexpr int BM = H64 ? 64 : 128;
  constexpr size_t lds = (size_t)6 * (2 * BM + 2 * 64) * 32 * sizeof(u16);
  auto kern = conv_dma_deep_kernel<NSPLIT, PW, W8, H64>;
  static DeviceOnce once;
  XDET_TRY(ensure_dynamic_lds(once, reinterpret_cast<const void*>(kern), (int)lds));
  dim3 grid((unsigned)(cdiv(cdiv(p.M, BM), 8) * 8 * (p.Cout_pad / 64)));
  if (p.group_rows) {
    const int64_t groups = p.M / p.group_rows, tpg = (int64_t)(p.group_rows / BM) * (p.Cout_pad / 64);
    grid = dim3((unsigned)(cdiv(groups, 8) * 8 * tpg));
  }
  hipLaunchKernelGGL(kern, grid, dim3(W8 ? 512 : 256), lds, s, p);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// multi-tap layers (the 256 x 256 tile's: the RPN conv, the direct large-separable convs) below 4 GiB: the generic form with
// buffer loads (GBUF); beyond 4 GiB: the pointer form
static bool gbuf_eligible(const ConvParams& p) {
  if (p.group_rows != 0) return false;
  const size_t a_bytes = ((((size_t)p.N * p.H * p.W + 15) >> 4) * (size_t)(p.ldi >> 5)) << 10;
  const size_t b_bytes = (size_t)(p.Kp / 32) * p.Cout_pad * 64;
  return a_bytes < ((size_t)1 << 32) && b_bytes < ((size_t)1 << 32) && p.Kp == p.Cin_p * p.KH * p.KW;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NSPLIT, int NSTAGE = 2, bool PW = false, bool X8 = false, bool GBUF = false>
static int launch_d(const ConvParams& p, hipStream_t s) {
  if (!PW && pw_eligible(p)) return launch_d<BM, BN, WAVES_M, WAVES_N, NSPLIT, NSTAGE, true>(p, s);
  if constexpr (PW && !X8 && NSPLIT == 3) {
    if (p.x8) return launch_d<BM, BN, WAVES_M, WAVES_N, NSPLIT, NSTAGE, true, true>(p, s);
  }
  XDET_REQUIRE(X8 || !p.x8, "conv(dma): x8 planes need a pointwise f16x3 layer below 4 GiB");
  constexpr size_t lds = (size_t)NSTAGE * (2 * BM + 2 * BN) * 32 * sizeof(u16);
  if constexpr (!PW && !X8 && !GBUF && BM == 256 && BN == 256) {
    if (gbuf_eligible(p)) return launch_d<BM, BN, WAVES_M, WAVES_N, NSPLIT, NSTAGE, false, false, true>(p, s);
  }
  auto kern = conv_dma_f16_kernel<BM, BN, WAVES_M, WAVES_N, NSPLIT, NSTAGE, PW, false, X8, GBUF>;
  static DeviceOnce once;
  XDET_TRY(ensure_dynamic_lds(once, reinterpret_cast<const void*>(kern), (int)lds));
  dim3 grid((unsigned)(cdiv(cdiv(p.M, BM), 8) * 8 * (p.Cout_pad / BN)));
  if (p.group_rows) {
    const int64_t groups = p.M / p.group_rows, tpg = (int64_t)(p.group_rows / BM) * (p.Cout_pad / BN);
    grid = dim3((unsigned)(cdiv(groups, 8) * 8 * tpg));
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES_M * WAVES_N), lds, s, p);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// the large-batch form of a layer whose reduction carries a fixed split (see FOLD above); false = not applicable here
bool conv_dma_fold_applicable(const ConvParams& p, int n_tile, int nsplit) {
  return n_tile == 128 && nsplit == 3 && p.group_rows == 0 && p.ksplit > 1 && cdiv(p.M, 256) * (p.Cout_pad / 128) >= 170;
}
int launch_conv_mfma_dma_fold(const ConvParams& p, hipStream_t s) {
  constexpr size_t lds = (size_t)2 * (2 * 256 + 2 * 128) * 32 * sizeof(u16);
  auto kern = conv_dma_f16_kernel<256, 128, 4, 2, 3, 2, false, true>;
  static DeviceOnce once;
  XDET_TRY(ensure_dynamic_lds(once, reinterpret_cast<const void*>(kern), (int)lds));
  dim3 grid((unsigned)(cdiv(cdiv(p.M, 256), 8) * 8 * (p.Cout_pad / 128)));
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

int launch_conv_mfma_dma(const ConvParams& p_in, int n_tile, int nsplit, hipStream_t s) {
  ConvParams p = p_in;
  p.skip_dead = 1;
  XDET_REQUIRE(p.Kp % 32 == 0 && p.Cin_p % 32 == 0 && p.ldi >= p.Cin_p && p.ldi % 8 == 0,
               "conv(dma): channel counts must be padded to 32");
  XDET_REQUIRE(p.Cout_pad % n_tile == 0, "conv(dma): Cout_pad must be a multiple of the N tile");
  XDET_REQUIRE(p.in_hi && (nsplit == 1 || p.in_lo) && p.wt_hi && (nsplit == 1 || p.wt_lo) && p.zeros,
               "conv(dma): split planes missing");
  if (p.M <= 0) return XDET_OK;
  if (n_tile == 128 && nsplit == 3) {
    // Tile choice.  The biggest tile moves the fewest operand bytes and LDS-DMA pieces per MFMA, so it
    // wins as long as the grid still covers the 256 CUs (measured on MI355X, tools/conv_bench.py --planes):
    // 256x256 once there are >= 2 workgroups per CU, or ~1 per CU with a long K loop to amortise its
    // prologue/epilogue; 256x128 from ~2/3 workgroup per CU; else 128x128 (two workgroups share a CU).
    const int64_t b256 = cdiv(p.M, 256) * (p.Cout_pad / 256);
    const int nk = p.Kp / 32;
    int tile = 0;
    // (round 5: the 256 x 128 tile lost every same-box A/B against 128 x 128 tiles at two workgroups per CU -- ResNet-50 trunk at
    //  batch 8 1.715 -> 1.683 ms, detector at batch 8 3.045 -> 2.860 ms, default bench neutral: its one-round grids run prologue,
    //  K loop and a store-bound epilogue strictly in sequence on every CU; retired)
    // ...or when the 256x256 grid fills whole rounds of the 256 CUs (within 6 %)
    const bool full_rounds = b256 >= 240 && (b256 % 256 == 0 || b256 % 256 >= 240);
    if (p.Cout_pad % 256 == 0 && (b256 >= 512 || full_rounds || (b256 >= 200 && nk >= 40))) tile = 2;
    if (p.group_rows && p.group_rows % 256 != 0) tile = 0;      // a group must be whole M tiles
    if (tile == 2) return launch_d<256, 256, 2, 4, 3>(p, s);
  }
  if (n_tile == 128 && nsplit == 3 && !p.group_rows && !p.x8) {
    // about one round of 128 x 128 tiles with a long K loop (a single image's block4_sepconv2, the middle flow at batches 3-5):
    // the split-K kernel's four-stage ring with ONE range -- the same reduction, bit-identical -- runs a 32-deep step in ~0.65 us
    // where the two-stage kernel pays ~1.2 (the rule of Plan::maybe_ksplit for ResNet-50's stage 2, here per call: one range
    // changes no summation tree, so the choice may depend on the batch)
    const int64_t b128 = cdiv(p.M, 128) * (p.Cout_pad / 128);
    if (b128 > 128 && b128 <= 256 && p.Kp / 32 >= 16 && p.Kp == p.Cin_p * p.KH * p.KW && p.ldi % 32 == 0 &&
        conv_ksplit_supported(p.KH, p.KW, (int64_t)p.N * p.H * p.W, p.ldi, p.Cin_p, p.Cout_pad)) {
      ConvParams q = p;
      q.ksplit = 1;
      q.ks_partial = nullptr;
      return launch_conv_mfma_ksplit(q, 128, 3, 2, 0, s);
    }
  }
  if (n_tile == 128 && nsplit == 3 && p.group_rows && p.group_rows % 64 == 0 && !p.x8) {
    // A grouped GEMM with few live rows per group (the frequency bins of a single image: 30 rows of a 128-row tile, 44 bins x
    // 4 column tiles = 176 workgroups walking 128 K steps at ~1 us each on the two-stage kernel): 64 x 64 tiles on the
    // six-stage ring -- twice the workgroups, half the padding rows, ~0.5 us per step.  Same K order per element: same bits.
    const int live = p.group_live_rows ? p.group_live_rows : p.group_rows;
    const int64_t groups = p.M / p.group_rows, wgs = groups * cdiv(live, 64) * (p.Cout_pad / 64);
    // (measured and dropped: a two-pair ring, 64 KB, two workgroups per CU so that the 352 tiles are one round -- 101 -> 106 us)
    if (live <= 64 && wgs <= 768) return launch_deep<3, false, false, true>(p, s);
  }
  if (n_tile == 128) {
    // small batches: 128 x 128 tiles leave most of the 256 CUs idle (a 900-row layer is 8 x 4..6 workgroups); halve
    // the N tile to double the workgroup count -- per-element K order is unchanged, so results stay bit-identical
    const int64_t b128 = cdiv(p.M, 128) * (p.Cout_pad / 128);
    if (nsplit == 3 && b128 <= 128 && (!p.group_rows || p.group_rows % 128 == 0)) {
      // few workgroups: the deep operand ring (x8 planes: the two-stage kernel -- the deep ring has no x8 form)
      return p.x8 ? launch_d<128, 64, 4, 1, 3>(p, s) : launch_deep<3>(p, s);
    }
    return nsplit == 1 ? launch_d<128, 128, 2, 2, 1>(p, s) : launch_d<128, 128, 2, 2, 3>(p, s);
  }
  if (n_tile == 64) return nsplit == 1 ? launch_d<128, 64, 4, 1, 1>(p, s) : launch_d<128, 64, 4, 1, 3>(p, s);
  set_last_error("conv(dma): unsupported N tile");
  return XDET_ERR_UNSUPPORTED;
}

}  // namespace xdet
