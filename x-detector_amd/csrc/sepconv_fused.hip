// Fused separable block for the entry flow: (ReLU ->) depthwise 3x3 -> pointwise 1x1 -> BN (-> ReLU) in ONE
// kernel, "nothing in between" (net/xception_body.py:220-234).  The two-kernel form moves the depthwise result
// through HBM as split f16 planes (4 B written + 4 B read per element); at 237x237 / 119x119 with 64..256
// channels those layers are HBM-bound, so the round trip is a third of their time.  Here the depthwise output
// never leaves the CU:
//
//   patch of the f32 input (6 x 32 pixels x 32 channels)  --buffer_load ... lds (zero fill = SAME padding)-->  LDS
//   3x3 stencil in VALU (same FMA order as depthwise3x3_tile_kernel) -> hi/lo f16 -> LDS A tile [128 px][32 ch]
//   A x W (K-blocked f16 hi/lo weights straight from L2 into registers)  --3 x v_mfma_f32_32x32x16_f16-->  acc
//   after the last channel chunk: y = acc * scale + shift (folded BN), optional ReLU, f32 NHWC store
//   (straight from the accumulators: no LDS bounce, the next tile's patch is already in flight)
//
// A workgroup (4 waves) works on tiles of 4 rows x 30 pixels (a 128-row GEMM tile with 8 idle rows: 30 + 2 halo
// pixels are exactly four 1 KB DMA pieces) x 128 output channels; channel chunks of 32 are the K steps.  The
// patch of the next (tile, chunk) is in flight while the current one is filtered and multiplied; ~79 KB of LDS
// -> two workgroups per CU cover each other's barriers.  Arithmetic and accumulation order are exactly those
// of depthwise3x3_tile_kernel + split + conv_dma_f16_kernel, so the result is bit-identical to the two-kernel
// form (tests/test_gpu_layers.py).  Cout = 256 runs as two 128-wide passes over the same patch (second pass
// from L2); wider layers (728) stay on the two-kernel path.
#include "common.h"
#include <cstdlib>
#include <type_traits>

// Bottleneck probes for tools/ubench/sepconv_probe.hip ONLY (the product is compiled with level 0 = everything on):
//   1 = patch DMA + barriers only, 2 = + stencil / A tile, 3 = + MFMAs (no epilogue stores), 4 = everything but the DMA
#ifndef SF_PROBE_LEVEL
#define SF_PROBE_LEVEL 0
#endif
#ifndef SF_NSLOT
#define SF_NSLOT 3
#endif

namespace xdet {

constexpr bool SF_DO_DMA = SF_PROBE_LEVEL != 4;
constexpr bool SF_DO_STENCIL = SF_PROBE_LEVEL != 1;
constexpr bool SF_DO_MFMA = SF_PROBE_LEVEL == 0 || SF_PROBE_LEVEL >= 3;
constexpr bool SF_DO_STORE = SF_PROBE_LEVEL == 0 || SF_PROBE_LEVEL == 4 || SF_PROBE_LEVEL == 9;
#if SF_PROBE_LEVEL == 9
#define SF_STAMP(slot)                                                                         \
  do {                                                                                         \
    if (blockIdx.x == 8 && lane == 0 && s < 64) p.dbg[(wave * 64 + s) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define SF_STAMP(slot) do {} while (0)
#endif

typedef float sf_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 sf_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sf_f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef unsigned sf_u2 __attribute__((ext_vector_type(2)));

constexpr int SF_R = 4, SF_X = 30, SF_P = 32;       // tile rows, tile width, patch pitch (pixels)
constexpr int SF_ROWS = SF_R + 2;                   // patch rows
// The patch in LDS: 8-pixel pieces of 1 KB (one DMA instruction each), each followed by a 128 B pad.  The pad flips
// the bank half (128 B of a pixel = half of the 64 banks) of every second piece, so that the four lane groups of the
// stencil's ds_read_b128 (lanes of strips s and s+2 share a group and read pixels 8 apart) are conflict-free; with
// dense rows they were 2-way conflicts and the stencil reads -- 60 % of the kernel's LDS cycles -- took twice as long.
constexpr int SF_PIECE_F = 8 * 32 + 32;             // floats per piece incl. pad (1152 B)
constexpr int SF_ROW_F = (SF_P / 8) * SF_PIECE_F;   // floats per patch row (4608 B)
constexpr int SF_PATCH_F = SF_ROWS * SF_ROW_F;      // floats per patch buffer (27 KB)
constexpr int SF_KMAX = 256;                        // input channels (dw taps live in LDS)
constexpr int SF_COUT_MAX = 1024;                   // output channels (sepconv_fused_supported)
constexpr int SF_NJ = SF_ROWS * (SF_P / 8) / 4;     // DMA pieces per wave per chunk (6)

struct SepFusedParams {
  const float* in;       // NHWC f32, channel stride ld (= padded Cin, multiple of 32)
  const float* w9c;      // depthwise taps [9][ld]
  const u16* wt_hi;      // pointwise weights, K-blocked [ld/32][Cout_pad][32] f16 (pre-scaled hi / lo)
  const u16* wt_lo;
  const float* scale;    // [Cout_pad] folded BN (and weight pre-scale)
  const float* shift;
  float* out;            // NHWC f32, channel stride ldo
  int N, H, W, ld, ldo, Cout_pad, relu_in, relu_out;
  int TY, TX, NT, ntiles, tiles_per_block;
  // HPOOL: the epilogue writes the 3-column / stride-2 maximum of each output row (the horizontal half of the
  // max_pooling2d(3, 2, 'same') that follows the block, net/xception_body.py:281-286): out is [N][H][Wo][ldo]
  int Wo, pool_pad_l;
#if SF_PROBE_LEVEL == 9
  unsigned long long* dbg;   // timeline probe: [wave][step][8] s_memtime stamps of workgroup 8
#endif
};

// LDS accesses issued while the patch prefetch (an LDS-writing DMA) is in flight.  The compiler cannot tell
// that they never alias the DMA's target buffer and would drain vmcnt(0) in front of every one of them
// (stalling on the prefetch each chunk), so they are issued as inline asm with explicit lgkmcnt waits.
typedef __attribute__((address_space(3))) void* sf_lds_ptr;
__device__ __forceinline__ unsigned sf_lds_addr(const void* p) {
  return (unsigned)(size_t)(sf_lds_ptr)(p);
}
template <int OFF>
__device__ __forceinline__ void sf_ds_write_b64(unsigned addr, uint2 v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
typedef float sf_f32x4 __attribute__((ext_vector_type(4)));
typedef float sf_f32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ sf_f32x4 sf_ds_read_f4(unsigned addr) {
  sf_f32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int OFF>
__device__ __forceinline__ sf_f16x8 sf_ds_read_b128(unsigned addr) {
  sf_f16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}

// ReLU of a value that came out of an asm LDS read.  fmaxf() would first canonicalise it (a second v_max_f32 per
// element: the compiler cannot know it is not a signalling NaN) -- 72 extra VALU instructions per 32-channel chunk.
__device__ __forceinline__ float sf_relu(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}

// HPOOL tiles are 28 output columns apart and start pool_pad_l columns left of a multiple of 28: the 30 columns a tile
// computes then hold every window of its 14 pooled columns (window k = local columns 2k .. 2k+2).
constexpr int SF_XP = 28;

// hi (and lo) halves of four channels of A-tile row `r` of a strip (r = 0..3: 64 B apart; lo plane 8 KB behind hi)
template <bool SPLIT3>
__device__ __forceinline__ void sf_write_row(unsigned wa0, int r, sf_f16x4 hv, sf_f32x4 a) {
  sf_f16x4 lv = {(_Float16)(a.x - (float)hv[0]), (_Float16)(a.y - (float)hv[1]), (_Float16)(a.z - (float)hv[2]),
                 (_Float16)(a.w - (float)hv[3])};
  const uint2 h = *reinterpret_cast<uint2*>(&hv), l = *reinterpret_cast<uint2*>(&lv);
  switch (r) {   // r is a compile-time constant after unrolling: one case survives
    case 0: sf_ds_write_b64<0>(wa0, h); if (SPLIT3) sf_ds_write_b64<8192>(wa0, l); break;
    case 1: sf_ds_write_b64<64>(wa0, h); if (SPLIT3) sf_ds_write_b64<8192 + 64>(wa0, l); break;
    case 2: sf_ds_write_b64<128>(wa0, h); if (SPLIT3) sf_ds_write_b64<8192 + 128>(wa0, l); break;
    default: sf_ds_write_b64<192>(wa0, h); if (SPLIT3) sf_ds_write_b64<8192 + 192>(wa0, l); break;
  }
}

// hi = f16(a), lo = f16(a - float(hi)) of four values, as two packed pairs each.  v_fma_mixlo/hi_f16 take the f16 hi half
// straight as an fma operand and round the f32 difference (exact: it has at most 13 significant bits) to f16 into one
// half of the destination: one instruction per value where convert-back + subtract + convert were two -- the same bits.
__device__ __forceinline__ void sf_split4(sf_f32x4 a, uint2* h, uint2* l) {
  // ONE asm block: inline asm is opaque to the hazard recogniser, and gfx950 wants a wait state between a half-register
  // write (v_fma_mixlo / mixhi) and a VALU that reads the register -- the order below keeps another instruction between
  // the two halves of each lo word, the trailing s_nop covers whatever the compiler places behind the block
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

// v_permlane32_swap_b32 a, b: the upper 32 lanes of a and the lower 32 lanes of b change places, i.e. afterwards
// a = {a.lower, b.lower} and b = {a.upper, b.upper}.  (Inline asm: with the clang builtin, hipcc 7.2 dropped the second
// result of all but the first swap of an unrolled sequence and used the first result in its place.)
// (The s_nop: a VALU result read by the swap in the next issue slot is a hazard the compiler cannot see through the
//  asm -- round 5 found window maxima built from a stale register once the BN multiply sat directly in front.)
__device__ __forceinline__ void sf_permlane32_swap(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// WAVES_N = 2: the four waves as 2 x 2 over a 128 x 128 tile (64 x 64 each), a 256-channel layer as two passes.
// WAVES_N = 4: 1 x 4 over a 128 x 256 tile (a wave: all four tile rows x 64 channels) -- ONE pass: the patch DMA, the
// stencil and the A-tile round trip of a chunk are paid once for all 256 output channels (they, not the MFMAs, are what
// a chunk costs).  128 accumulator registers leave no room for the two-pass form's habits: the stencil runs in two
// halves of 2 pixels (28 fewer live registers, the taps read twice), the pointwise weights are loaded AFTER it (and the
// prefetch after them, to land under the chunk's 96 MFMAs), fragments one 16-deep half at a time.
template <bool SPLIT3, bool RELU_IN, bool HPOOL, int WAVES_N>
__global__ __launch_bounds__(256, 2) void sepconv_fused_kernel(SepFusedParams p) {
  constexpr int TM = WAVES_N;                     // 32-row accumulator blocks (= tile rows) per wave
  constexpr int BN = 64 * WAVES_N;                // output channels per pass
  __shared__ __attribute__((aligned(16))) float s_patch[2][SF_PATCH_F];
  __shared__ __attribute__((aligned(16))) u16 s_a[2 * 128 * 32];        // A tile: hi rows, then lo rows (16 KB)
  __shared__ __attribute__((aligned(16))) float s_w[9 * SF_KMAX];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Tile schedule.  Workgroups are dealt round-robin to the 8 XCDs (private 4 MB L2s).  XCD x owns the contiguous
  // tile band [x*per_xcd, (x+1)*per_xcd); its resident workgroups take the band's tiles in an interleaved order
  // (workgroup j: tiles j, j+G, j+2G, ...), so at any moment one XCD works on ~G CONSECUTIVE tiles: vertical
  // neighbours (shared halo rows) and the two 128-channel passes of a tile read the same input lines within
  // microseconds of each other and meet in that XCD's L2.  (A workgroup walking consecutive tiles by itself found
  // its halo rows evicted by the time it came back to them: 1.6x / 3.1x input bytes over the fabric.)
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, G = gridDim.x >> 3;
  const int per_xcd = (p.ntiles + 7) >> 3;
  const int t_begin = xcd * per_xcd + wg;
  const int t_end = min(p.ntiles, (xcd + 1) * per_xcd);
  if (t_begin >= t_end) return;
  const int KC = p.ld >> 5;                       // channel chunks = K steps

  // taps in LDS as [9][SF_KMAX]: a COMPILE-TIME row stride, so the nine tap reads of a chunk are one address register
  // plus immediate offsets (with the tensor's runtime channel stride the compiler kept nine per-lane address registers
  // and their nine bases alive across the chunk loop -- the registers the one-pass form was short of)
  for (int i = tid; i < 9 * p.ld; i += 256) {
    const int t = i / p.ld;
    s_w[t * SF_KMAX + (i - t * p.ld)] = p.w9c[i];
  }

  // ---- tile walk: N-pass fastest (same patch again, from L2), then ty, tx, image ----
  struct Coord { int nt, ty, tx, n; };
  auto decode = [&](int q) {
    Coord c;
    c.nt = q % p.NT; q /= p.NT;
    c.ty = q % p.TY; q /= p.TY;
    c.tx = q % p.TX;
    c.n = q / p.TX;
    return c;
  };
  Coord cur = decode(t_begin);

  // ---- DMA descriptors: piece i = wave + 4*jj moves 8 pixels x 128 B of patch row rr (raw buffer loads over the
  //      whole input tensor; a lane that must read padding is sent out of bounds and gets zeros) ----
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.in), 0, (int)((size_t)p.N * p.H * p.W * p.ld * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (int)(unsigned)std::min<size_t>((size_t)p.N * p.H * (HPOOL ? p.Wo : p.W) * p.ldo * 4, 0xffffffffull),
      0x00020000);
  // Offsets are rebuilt from wave-uniform (scalar) parts + one per-lane register each time: keeping six per-lane
  // descriptors resident cost 12 VGPRs the stencil needs.  `live == false` issues the same six instructions
  // with every lane out of bounds (zero fill): an unconditional instruction count keeps the compiler's vmcnt
  // bookkeeping exact, so waiting for the weights (issued earlier) never waits for the prefetch.
  const int px8 = lane >> 3;
  const int lane_off = (px8 * p.ld + (lane & 7) * 4) * 4;                            // bytes
  auto issue = [&](const Coord& c, int chunk, int buf, bool live) {
    const int y0 = c.ty * SF_R, x0 = HPOOL ? c.tx * SF_XP - p.pool_pad_l : c.tx * SF_X;
    const int tile_base = ((((c.n * p.H + y0) * p.W + x0) * p.ld) + chunk * 32) * 4;   // bytes, < 2^31 (host check)
#pragma unroll
    for (int jj = 0; jj < SF_NJ; ++jj) {
      const int i = wave + 4 * jj;
      const int rr = i >> 2, seg = i & 3;                                            // wave-uniform
      const bool rok = live && (unsigned)(y0 + rr - 1) < (unsigned)p.H;
      const bool ok = rok && (unsigned)(x0 + seg * 8 - 1 + px8) < (unsigned)p.W;
      const int soff = tile_base + ((rr - 1) * p.W + seg * 8 - 1) * p.ld * 4;
      const unsigned voff = ok ? (unsigned)(soff + lane_off) : 0xffffffffu;          // out of bounds -> zeros
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrc, (__attribute__((address_space(3))) void*)(&s_patch[buf][rr * SF_ROW_F + seg * SF_PIECE_F]), 16, voff, 0, 0, 0);
    }
  };

  // ---- roles ----
  const int c4 = lane & 7, strip = lane >> 3;     // depthwise: wave = tile row, 4 pixels x 4 channels per lane
  const int frow = lane & 31, fh = lane >> 5;     // MFMA fragments
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  unsigned a_rd[2];                               // LDS byte address of this lane's A fragment [ks] in its first row block
#pragma unroll                                    // (hi; lo = +8192; row block i = +2048 i, same chunk permutation)
  for (int ks = 0; ks < 2; ++ks) {
    const int rt = wm * TM * 32 + frow;
    a_rd[ks] = sf_lds_addr(s_a) + (unsigned)(rt * 32 + (((ks * 2 + fh) ^ ((rt >> 2) & 3)) << 3)) * 2u;
  }
  // depthwise: this lane's 4 output pixels of tile row `wave` are pxb .. pxb+3.  The tile is 30 wide: the last strip's
  // pixels 30, 31 are computed from whatever follows the patch row in LDS and land in A-tile rows that feed only masked
  // output rows (an MFMA row depends on its own A row only).  A strip's four A rows share one chunk permutation
  // ((row >> 2) & 3 with pxb a multiple of 4): one address register, the rows at immediate offsets.
  const int pxb = strip * 4;
  const unsigned wa0 = sf_lds_addr(s_a) + (unsigned)((wave * 32 + pxb) * 32 + (((c4 >> 1) ^ (strip & 3)) << 3) + (c4 & 1) * 4) * 2u;

  sf_f32x16 acc[TM][2];
  int buf = 0;
  issue(cur, 0, 0, true);
  for (int t = t_begin; t < t_end; t += G) {
    const Coord nxt = decode(min(t + G, p.ntiles - 1));
    const int y0 = cur.ty * SF_R, x0 = HPOOL ? cur.tx * SF_XP - p.pool_pad_l : cur.tx * SF_X, n0 = cur.nt * BN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // folded-BN scale/shift of this lane's two output channels: requested here, so that they are older than every
    // prefetch of the tile and using them in the epilogue never waits for a DMA (vmcnt retires in order)
    // (one-pass form: no four registers to spare across the chunks; loaded in front of the epilogue, behind a prefetch
    //  that has had the last chunk's 96 MFMAs to land)
    float esc[2], esh[2];
    auto load_bn = [&]() {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        esc[j] = p.scale[n0 + wn * 64 + j * 32 + frow];
        esh[j] = p.shift[n0 + wn * 64 + j * 32 + frow];
      }
    };
    if (TM == 2) load_bn();

    for (int chunk = 0; chunk < KC; ++chunk, buf ^= 1) {
      // patch(chunk) has landed (vmcnt(0): the DMA's LDS writes retire through vmcnt) and every wave is done with
      // the A tile and the other patch buffer.  Explicit: the compiler sees no LDS read of s_patch (they are asm)
      // and would not wait for the DMA at a plain __syncthreads().
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // -- this chunk's pointwise weights: L2 -> registers, issued BEFORE the DMA so that waiting for them
      //    (vmcnt retires in order) does not wait for the prefetch --
      sf_f16x8 bh[2][2], bl[2][2];
      auto load_b = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const size_t o = ((size_t)chunk * p.Cout_pad + (n0 + wn * 64 + j * 32 + frow)) * 32 + (ks * 2 + fh) * 8;
            bh[ks][j] = *reinterpret_cast<const sf_f16x8*>(p.wt_hi + o);
            if (SPLIT3) bl[ks][j] = *reinterpret_cast<const sf_f16x8*>(p.wt_lo + o);
          }
      };
      auto prefetch = [&]() {
        const bool more = chunk + 1 < KC;
        issue(more ? cur : nxt, more ? chunk + 1 : 0, buf ^ 1, more || t + G < t_end);
      };
      if (TM == 2) {
        load_b();
        __builtin_amdgcn_sched_barrier(0);
        prefetch();
      }
      __builtin_amdgcn_sched_barrier(0);
      // -- depthwise 3x3, one patch row (6 pixels x 4 channels) and its three taps at a time; per output the FMA
      //    order is ky, kx ascending = depthwise3x3_tile_kernel's --
      // pixels pxb .. pxb+3 sit in one piece; pxb+4, pxb+5 are in the next one (behind the pad) for the odd strips
      const unsigned t_addr = sf_lds_addr(&s_patch[buf][0]) +
                              (unsigned)((wave * SF_ROW_F + pxb * 32 + (pxb >> 3) * 32 + c4 * 4) * 4);
      const unsigned t_hi = t_addr + 512u + ((pxb & 7) == 4 ? 128u : 0u);
      const unsigned w_addr = sf_lds_addr(s_w) + (unsigned)((chunk * 32 + c4 * 4) * 4);
      // NP pixels at a time: 4 (two-pass form) or 2 + 2 (one-pass form: fewer live registers)
      constexpr int NP = TM == 2 ? 4 : 2;
#pragma unroll
      for (int half = 0; half < 4 / NP; ++half) {
        sf_f32x4 a[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) a[k] = (sf_f32x4){0.f, 0.f, 0.f, 0.f};
        auto tap_row = [&](auto KY) {
          constexpr int ky = decltype(KY)::value;
          sf_f32x4 col[NP + 2], ww[3];
          const unsigned ra = t_addr + ky * (SF_ROW_F * 4), rb = t_hi + ky * (SF_ROW_F * 4);
          if (NP == 4) {
            col[0] = sf_ds_read_f4<0>(ra);   col[1] = sf_ds_read_f4<128>(ra); col[2] = sf_ds_read_f4<256>(ra);
            col[3] = sf_ds_read_f4<384>(ra); col[NP] = sf_ds_read_f4<0>(rb);  col[NP + 1] = sf_ds_read_f4<128>(rb);
          } else if (half == 0) {
            col[0] = sf_ds_read_f4<0>(ra);   col[1] = sf_ds_read_f4<128>(ra); col[2] = sf_ds_read_f4<256>(ra);
            col[3] = sf_ds_read_f4<384>(ra);
          } else {
            col[0] = sf_ds_read_f4<256>(ra); col[1] = sf_ds_read_f4<384>(ra); col[2] = sf_ds_read_f4<0>(rb);
            col[3] = sf_ds_read_f4<128>(rb);
          }
          ww[0] = sf_ds_read_f4<(ky * 3 + 0) * SF_KMAX * 4>(w_addr);
          ww[1] = sf_ds_read_f4<(ky * 3 + 1) * SF_KMAX * 4>(w_addr);
          ww[2] = sf_ds_read_f4<(ky * 3 + 2) * SF_KMAX * 4>(w_addr);
          if (NP == 4)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(col[0]), "+v"(col[1]), "+v"(col[2]), "+v"(col[3]), "+v"(col[NP]), "+v"(col[NP + 1]),
                           "+v"(ww[0]), "+v"(ww[1]), "+v"(ww[2])::"memory");
          else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(col[0]), "+v"(col[1]), "+v"(col[2]), "+v"(col[3]), "+v"(ww[0]), "+v"(ww[1]), "+v"(ww[2])::"memory");
#pragma unroll
          for (int k = 0; k < NP + 2; ++k) {
            if (RELU_IN) {
              col[k].x = sf_relu(col[k].x); col[k].y = sf_relu(col[k].y);
              col[k].z = sf_relu(col[k].z); col[k].w = sf_relu(col[k].w);
            }
          }
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int k = 0; k < NP; ++k) {
              a[k].x = fmaf(col[k + kx].x, ww[kx].x, a[k].x); a[k].y = fmaf(col[k + kx].y, ww[kx].y, a[k].y);
              a[k].z = fmaf(col[k + kx].z, ww[kx].z, a[k].z); a[k].w = fmaf(col[k + kx].w, ww[kx].w, a[k].w);
            }
        };
        tap_row(std::integral_constant<int, 0>{});
        tap_row(std::integral_constant<int, 1>{});
        tap_row(std::integral_constant<int, 2>{});
        // -- split into f16 hi/lo, A tile rows wave*32 + pxb + k (chunk-permuted like the conv kernel's DMA layout) --
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const _Float16 h0 = (_Float16)a[k].x, h1 = (_Float16)a[k].y, h2 = (_Float16)a[k].z, h3 = (_Float16)a[k].w;
          sf_f16x4 hv = {h0, h1, h2, h3};
          sf_write_row<SPLIT3>(wa0, half * NP + k, hv, a[k]);
        }
      }
      // A tile visible to the workgroup: LDS writes only (lgkmcnt) -- the prefetch DMA stays in flight
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (TM != 2) {
        load_b();
        __builtin_amdgcn_sched_barrier(0);
        prefetch();
        __builtin_amdgcn_sched_barrier(0);
      }
      // -- pointwise: two 16-deep halves, products in the conv kernel's order (lo*hi, hi*lo, hi*hi) --
      auto read_half = [&](int ks, sf_f16x8* ah, sf_f16x8* al) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ah[i] = sf_ds_read_b128<0>(a_rd[ks] + i * 2048);
          if (SPLIT3) al[i] = sf_ds_read_b128<128 * 64>(a_rd[ks] + i * 2048);
        }
      };
      // one wait for the reads; the operands pass through it so that no MFMA can be scheduled above it
      auto wait_half = [&](sf_f16x8* ah, sf_f16x8* al) {
#pragma unroll
        for (int i = 0; i < TM; i += 2) {
          if (SPLIT3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[i]), "+v"(ah[i + 1]), "+v"(al[i]), "+v"(al[i + 1])::"memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[i]), "+v"(ah[i + 1])::"memory");
        }
      };
      auto mma_half = [&](int ks, const sf_f16x8* ah, const sf_f16x8* al) {
        if (SPLIT3) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[ks][j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[ks][j], acc[i][j], 0, 0, 0);
      };
      if (TM == 2) {
        sf_f16x8 ah[2][TM], al[2][TM];
        read_half(0, ah[0], al[0]);
        read_half(1, ah[1], al[1]);
        wait_half(ah[0], al[0]);
        wait_half(ah[1], al[1]);
        mma_half(0, ah[0], al[0]);
        mma_half(1, ah[1], al[1]);
      } else {
        sf_f16x8 ah[TM], al[TM];
        read_half(0, ah, al);
        wait_half(ah, al);
        mma_half(0, ah, al);
        __builtin_amdgcn_sched_barrier(0);
        read_half(1, ah, al);
        wait_half(ah, al);
        mma_half(1, ah, al);
      }
    }

    if (TM != 2) load_bn();
    if (HPOOL) {
      // Horizontal half of the pool, from the accumulators.  A lane holds columns c + 4 fh (c = 0..3, 8..11 in registers
      // 0..7, 16..19, 24..27 in 8..15) of one channel; v_permlane32_swap of register r with r + 8 leaves lanes 0..31
      // with columns 0..15 and lanes 32..63 with 16..31, window k = columns 2k..2k+2: the lower half takes k = 0..7
      // (its last window borrows column 16), the upper half k = 8..13.  Columns outside the image are -inf (SAME
      // padding never wins a maximum); max is exact, so the order of the nine comparisons does not matter.
      const int klim = min(SF_XP / 2, p.Wo - cur.tx * (SF_XP / 2)) - 8 * fh;   // this lane's windows are 8 fh + kk
      const bool edge = x0 < 0 || x0 + 32 > p.W;                               // wave-uniform
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int y = y0 + wm * TM + i;
        if (y >= p.H) continue;                                                // wave-uniform
        const unsigned row_off = (unsigned)((((size_t)cur.n * p.H + y) * p.Wo + cur.tx * (SF_XP / 2)) * p.ldo * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int co = n0 + wn * 64 + j * 32 + frow;
          const unsigned lane_off = co < p.ldo ? (unsigned)((8 * fh * p.ldo + co) * 4) : 0xffffffffu;
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            v[r] = fmaf(acc[i][j][r], esc[j], esh[j]);
            if (p.relu_out) v[r] = fmaxf(v[r], 0.f);
          }
          if (edge) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int col = x0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
              if ((unsigned)col >= (unsigned)p.W) v[r] = -INFINITY;
            }
          }
          float X[8], Y[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            X[r] = v[r];
            Y[r] = v[r + 8];
            sf_permlane32_swap(X[r], Y[r]);
          }
          // column q of this half (q = 0..15): X or Y [(q & 3) + 4 (q >> 3)] by (q & 4)
          auto colv = [&](int q) { return (q & 4) ? Y[(q & 3) + 4 * (q >> 3)] : X[(q & 3) + 4 * (q >> 3)]; };
          float c16a = X[0], col16 = X[0];
          sf_permlane32_swap(c16a, col16);                                    // lower half: the upper half's column 16
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            const float c2 = kk < 7 ? colv(2 * kk + 2) : col16;
            const float m = fmaxf(fmaxf(colv(2 * kk), colv(2 * kk + 1)), c2);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), orsrc, kk < klim ? lane_off : 0xffffffffu,
                                                  row_off + kk * p.ldo * 4, 0);
          }
        }
      }
      cur = nxt;
      continue;
    }
    const int lim = min(SF_X, p.W - x0) - 4 * fh;            // this lane's pixels are c + 4*fh, c = 0..3, 8..11, 16.., 24..
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int y = y0 + wm * TM + i;                        // tile row of this 32-row accumulator block
      if (y >= p.H) continue;                                // wave-uniform
      const unsigned row_off = (unsigned)((((size_t)cur.n * p.H + y) * p.W + x0) * p.ldo * 4);   // < 2^32 (host check)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int co = n0 + wn * 64 + j * 32 + frow;
        const unsigned lane_off = co < p.ldo ? (unsigned)((4 * fh * p.ldo + co) * 4) : 0xffffffffu;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          float v = fmaf(acc[i][j][r], esc[j], esh[j]);
          if (p.relu_out) v = fmaxf(v, 0.f);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, c < lim ? lane_off : 0xffffffffu, row_off + c * p.ldo * 4, 0);
        }
      }
    }
    cur = nxt;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Producer / consumer form (round 5).  The kernel above runs DMA wait -> stencil -> split -> A tile -> barrier -> MFMA
// back to back on the same four waves; two workgroups per CU were meant to cover each other's phases and measured
// 5,500 clocks per 32-channel chunk and CU against ~2,100 of VALU issue and 1,536 of MFMA issue (the 119^2 layers:
// 2.2-3.2 TB/s with the matrix pipe 24-35 % busy -- bound by neither).  Here the phases are different WAVES of one
// 512-thread workgroup, one of each on every SIMD:
//   waves 0-3 (producers): patch DMA three steps deep, 3x3 stencil, hi/lo split, A tile of step s into s_a[s & 1]
//   waves 4-7 (consumers): the MFMAs of step s-1 out of s_a[(s-1) & 1] against weights that were requested one
//                          half-step earlier, and the tile's epilogue (folded BN / ReLU / h-pool) after its last chunk
// with ONE s_barrier per step (a step = one 32-channel chunk of one tile): a SIMD's VALU (producer) and its matrix
// pipe (consumer) work on adjacent steps at the same time.  Step order, FMA order, product order and K order are those
// of the kernel above: bit-identical (tests/test_gpu_layers.py runs both forms against the two-kernel path).
template <bool SPLIT3, bool RELU_IN, bool HPOOL, int WAVES_N>
__global__ __launch_bounds__(512, 1) void sepconv_pc_kernel(SepFusedParams p) {
  constexpr int TM = WAVES_N;
  constexpr int BN = 64 * WAVES_N;
  constexpr int NSLOT = SF_NSLOT;
  __shared__ __attribute__((aligned(16))) float s_patch[NSLOT][SF_PATCH_F];
  __shared__ __attribute__((aligned(16))) u16 s_a[2][2 * 128 * 32];     // two A tiles (hi rows, then lo rows: 16 KB each)
  __shared__ __attribute__((aligned(16))) float s_w[9 * SF_KMAX];
  __shared__ float s_bn[2][SF_COUT_MAX];          // folded-BN scale / shift of every output channel (epilogue)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // tile schedule: the XCD bands of the kernel above, one workgroup per CU
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, G = gridDim.x >> 3;
  const int per_xcd = (p.ntiles + 7) >> 3;
  const int t_begin = xcd * per_xcd + wg;
  const int t_end = min(p.ntiles, (xcd + 1) * per_xcd);
  if (t_begin >= t_end) return;
  const int KC = p.ld >> 5;
  const int my_tiles = (t_end - t_begin + G - 1) / G;
  const int total = my_tiles * KC;                 // steps of this workgroup

  for (int i = tid; i < 9 * p.ld; i += 512) {
    const int t = i / p.ld;
    s_w[t * SF_KMAX + (i - t * p.ld)] = p.w9c[i];
  }
  for (int i = tid; i < p.Cout_pad; i += 512) {
    s_bn[0][i] = p.scale[i];
    s_bn[1][i] = p.shift[i];
  }

  struct Coord { int nt, ty, tx, n; };
  auto decode = [&](int q) {
    Coord c;
    c.nt = q % p.NT; q /= p.NT;
    c.ty = q % p.TY; q /= p.TY;
    c.tx = q % p.TX;
    c.n = q / p.TX;
    return c;
  };

  if (wave < 4) {
    // =================================================== producers ===================================================
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)((size_t)p.N * p.H * p.W * p.ld * 4), 0x00020000);
    const int px8 = lane >> 3;
    const int lane_off = (px8 * p.ld + (lane & 7) * 4) * 4;                            // bytes
    // The byte offsets of this wave's six pieces (chunk 0) are computed once per TILE (a lane that must read padding
    // gets an out-of-range offset: zero fill); a step's request is then six `s_mov m0` + `buffer_load ... lds` with
    // the chunk's 128 bytes in the scalar offset -- a single wave issues one instruction every ~4-5 clocks, and the
    // per-step address arithmetic (~140 SALU) used to be a third of the producer's step.
    unsigned pvoff[SF_NJ];
    auto tile_offsets = [&](const Coord& c, bool live) {
      const int y0 = c.ty * SF_R, x0 = HPOOL ? c.tx * SF_XP - p.pool_pad_l : c.tx * SF_X;
      const int tile_base = (((c.n * p.H + y0) * p.W + x0) * p.ld) * 4;                 // bytes, < 2^31 (host check)
#pragma unroll
      for (int jj = 0; jj < SF_NJ; ++jj) {
        const int i = wave + 4 * jj;
        const int rr = i >> 2, seg = i & 3;                                            // wave-uniform
        const bool rok = live && (unsigned)(y0 + rr - 1) < (unsigned)p.H;
        const bool ok = rok && (unsigned)(x0 + seg * 8 - 1 + px8) < (unsigned)p.W;
        const int soff = tile_base + ((rr - 1) * p.W + seg * 8 - 1) * p.ld * 4;
        pvoff[jj] = ok ? (unsigned)(soff + lane_off) : 0xffffffffu;
      }
    };
    auto issue = [&](int chunk, int slot) {
#pragma unroll
      for (int jj = 0; jj < SF_NJ; ++jj) {
        const int i = wave + 4 * jj;
        const int rr = i >> 2, seg = i & 3;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rsrc, (__attribute__((address_space(3))) void*)(&s_patch[slot][rr * SF_ROW_F + seg * SF_PIECE_F]), 16, pvoff[jj],
            chunk * 128, 0, 0);
      }
    };
    // cursor of the NEXT patch to request (NSLOT - 1 steps ahead of the stencil)
    int it = t_begin, ichunk = 0, islot = 0;
    tile_offsets(decode(it), SF_DO_DMA);
    auto issue_next = [&]() {
      issue(ichunk, islot);
      islot = islot == NSLOT - 1 ? 0 : islot + 1;
      if (++ichunk == KC) {
        ichunk = 0;
        it += G;
        tile_offsets(decode(min(it, p.ntiles - 1)), it < t_end && SF_DO_DMA);
      }
    };
#pragma unroll
    for (int k = 0; k < NSLOT - 1; ++k) issue_next();

    const int c4 = lane & 7, strip = lane >> 3;     // wave = tile row, 4 pixels x 4 channels per lane
    const int pxb = strip * 4;
    // A tile layout of this kernel: the 32 rows of a tile row sit at slot (r & 16) | ((r & 3) << 2) | ((r >> 2) & 3) -- the
    // four strips a ds_write_b64 half-wave holds (rows 4 apart) land in four CONSECUTIVE 64-byte slots, all 64 banks
    // once (rows in pixel order put them 256 B apart: 4-way conflicts, a quarter of the step's LDS cycles) -- and the
    // 16-byte chunks of a row are permuted by (r & 3), which keeps the fragment reads (16 lanes = 16 consecutive rows)
    // conflict-free as well.  Row k of a strip: + 256 k bytes, chunk ((c4 >> 1) ^ k).
    unsigned wa_k[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      wa_k[k] = sf_lds_addr(&s_a[0][0]) +
                (unsigned)((wave * 32 + (strip & 4) * 4 + k * 4 + (strip & 3)) * 32 + (((c4 >> 1) ^ k) << 3) + (c4 & 1) * 4) * 2u;
    const unsigned t_rel = (unsigned)((wave * SF_ROW_F + pxb * 32 + (pxb >> 3) * 32 + c4 * 4) * 4);
    const unsigned hi_rel = 512u + ((pxb & 7) == 4 ? 128u : 0u);
    const unsigned patch0 = sf_lds_addr(&s_patch[0][0]);
    const unsigned w_base = sf_lds_addr(s_w) + (unsigned)(c4 * 4 * 4);

    int rslot = 0, abuf = 0, pchunk = 0;
    for (int s = 0; s < total; ++s) {
      // patch(s) has landed (the six pieces of patch(s+1) may still be in flight), the A tile written in the last
      // step is complete, and after the barrier the consumers are done with s_a[abuf] (they read it in step s-1)
      SF_STAMP(0);                                   // arrival at the barrier (previous step's work done)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(SF_NJ * (NSLOT - 2)) : "memory");
      SF_STAMP(1);                                   // own DMA pieces of this step's patch have landed
      asm volatile("s_barrier" ::: "memory");
      SF_STAMP(2);                                   // released
      issue_next();
      SF_STAMP(4);                                   // next patch requested
      __builtin_amdgcn_sched_barrier(0);
      if (!SF_DO_STENCIL) continue;
      const unsigned t_addr = patch0 + (unsigned)rslot * (unsigned)(SF_PATCH_F * 4) + t_rel;
      const unsigned t_hi = t_addr + hi_rel;
      const unsigned w_addr = w_base + (unsigned)(pchunk * 32 * 4);
      const unsigned wa_off = (unsigned)abuf * 16384u;
      sf_f32x4 a[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = (sf_f32x4){0.f, 0.f, 0.f, 0.f};
      // the reads of a patch row (six pixels, three taps) are issued one row ahead of the row being filtered and
      // consumed behind counted lgkmcnt waits (LDS returns in order; the counter has four bits: <= 15 in flight)
      sf_f32x4 col[3][6], ww[3][3];
      auto read_row = [&](auto KY) {
        constexpr int ky = decltype(KY)::value;
        const unsigned ra = t_addr + ky * (SF_ROW_F * 4), rb = t_hi + ky * (SF_ROW_F * 4);
        col[ky][0] = sf_ds_read_f4<0>(ra);   col[ky][1] = sf_ds_read_f4<128>(ra); col[ky][2] = sf_ds_read_f4<256>(ra);
        col[ky][3] = sf_ds_read_f4<384>(ra); col[ky][4] = sf_ds_read_f4<0>(rb);   col[ky][5] = sf_ds_read_f4<128>(rb);
        ww[ky][0] = sf_ds_read_f4<(ky * 3 + 0) * SF_KMAX * 4>(w_addr);
        ww[ky][1] = sf_ds_read_f4<(ky * 3 + 1) * SF_KMAX * 4>(w_addr);
        ww[ky][2] = sf_ds_read_f4<(ky * 3 + 2) * SF_KMAX * 4>(w_addr);
      };
      auto fma_row = [&](auto KY, auto PENDING) {
        constexpr int ky = decltype(KY)::value;
        constexpr int pending = decltype(PENDING)::value;    // reads issued after this row's
        asm volatile("s_waitcnt lgkmcnt(%9)"
                     : "+v"(col[ky][0]), "+v"(col[ky][1]), "+v"(col[ky][2]), "+v"(col[ky][3]), "+v"(col[ky][4]), "+v"(col[ky][5]),
                       "+v"(ww[ky][0]), "+v"(ww[ky][1]), "+v"(ww[ky][2])
                     : "n"(pending)
                     : "memory");
        if (RELU_IN) {
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            col[ky][k].x = sf_relu(col[ky][k].x); col[ky][k].y = sf_relu(col[ky][k].y);
            col[ky][k].z = sf_relu(col[ky][k].z); col[ky][k].w = sf_relu(col[ky][k].w);
          }
        }
        // v_pk_fma_f32 on explicit channel pairs (left to the vectoriser, the ReLU-free instance came out as 144 scalar FMAs)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const sf_f32x2 c_lo = __builtin_shufflevector(col[ky][k + kx], col[ky][k + kx], 0, 1);
            const sf_f32x2 c_hi = __builtin_shufflevector(col[ky][k + kx], col[ky][k + kx], 2, 3);
            const sf_f32x2 w_lo = __builtin_shufflevector(ww[ky][kx], ww[ky][kx], 0, 1);
            const sf_f32x2 w_hi = __builtin_shufflevector(ww[ky][kx], ww[ky][kx], 2, 3);
            sf_f32x2 a_lo = __builtin_shufflevector(a[k], a[k], 0, 1), a_hi = __builtin_shufflevector(a[k], a[k], 2, 3);
            a_lo = __builtin_elementwise_fma(c_lo, w_lo, a_lo);
            a_hi = __builtin_elementwise_fma(c_hi, w_hi, a_hi);
            a[k] = __builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3);
          }
      };
      read_row(std::integral_constant<int, 0>{});
      read_row(std::integral_constant<int, 1>{});
      fma_row(std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{});
      SF_STAMP(5);
      read_row(std::integral_constant<int, 2>{});
      fma_row(std::integral_constant<int, 1>{}, std::integral_constant<int, 9>{});
      SF_STAMP(6);
      fma_row(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});
      SF_STAMP(7);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint2 h, l;
        sf_split4(a[k], &h, &l);
        sf_ds_write_b64<0>(wa_k[k] + wa_off, h);
        if (SPLIT3) sf_ds_write_b64<8192>(wa_k[k] + wa_off, l);
      }
      SF_STAMP(3);                                   // stencil + A-tile writes issued
      rslot = rslot == NSLOT - 1 ? 0 : rslot + 1;
      abuf ^= 1;
      pchunk = pchunk + 1 == KC ? 0 : pchunk + 1;
    }
    // publishes the last A tile; nothing of this wave is in flight when it ends
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    return;
  }

  // ===================================================== consumers =====================================================
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.out, 0, (int)(unsigned)std::min<size_t>((size_t)p.N * p.H * (HPOOL ? p.Wo : p.W) * p.ldo * 4, 0xffffffffull),
      0x00020000);
  const int cw = wave - 4;
  const int frow = lane & 31, fh = lane >> 5;
  const int wm = cw / WAVES_N, wn = cw % WAVES_N;
  unsigned a_rd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int slot = wm * TM * 32 + ((frow & 16) | ((frow & 3) << 2) | ((frow >> 2) & 3));
    a_rd[ks] = sf_lds_addr(&s_a[0][0]) + (unsigned)(slot * 32 + (((ks * 2 + fh) ^ (frow & 3)) << 3)) * 2u;
  }
  sf_f32x16 acc[TM][2];
  // Pointwise weights: L2 -> registers, one 16-deep half at a time, requested as soon as the half's registers are free
  // (right behind its MFMAs) for the NEXT step -- half a step + the barrier ahead of their use.  Loads and waits are
  // inline asm: left to the compiler, the first MFMA of a step sat behind `s_waitcnt vmcnt(0)` (loop-carried loads
  // next to the epilogue's conditional stores), i.e. behind the half requested a few instructions before the barrier,
  // and every step paid an L2 round trip.  vmcnt retires in order: with NB loads per half, `vmcnt(NB)` in front of a
  // half's MFMAs leaves exactly the other half's loads in flight (a tile's last step also waits for its stores).
  constexpr int NB = SPLIT3 ? 4 : 2;
  sf_f16x8 bh[2][2], bl[2][2];
  const unsigned bvoff = (unsigned)(((wn * 64 + frow) * 32 + fh * 8) * 2);                // bytes; j: + 2048, ks: + 32
#define SF_LOAD_B(dst, base, imm) \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(bvoff), "s"(base), "n"(imm) : "memory")
  auto load_b_half = [&](int ks, int chunk, int n0) {       // ks is a literal at every call site: one arm survives
    const size_t row0 = ((size_t)chunk * p.Cout_pad + n0) * 32;                          // wave-uniform
    const u16* bhp = p.wt_hi + row0;
    const u16* blp = p.wt_lo + row0;
    if (ks == 0) {
      SF_LOAD_B(bh[0][0], bhp, 0);
      SF_LOAD_B(bh[0][1], bhp, 2048);
      if (SPLIT3) { SF_LOAD_B(bl[0][0], blp, 0); SF_LOAD_B(bl[0][1], blp, 2048); }
    } else {
      SF_LOAD_B(bh[1][0], bhp, 32);
      SF_LOAD_B(bh[1][1], bhp, 32 + 2048);
      if (SPLIT3) { SF_LOAD_B(bl[1][0], blp, 32); SF_LOAD_B(bl[1][1], blp, 32 + 2048); }
    }
  };
#undef SF_LOAD_B
  auto wait_b_half = [&](int ks) {
    if (SPLIT3) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(bh[ks][0]), "+v"(bh[ks][1]), "+v"(bl[ks][0]), "+v"(bl[ks][1]) : "n"(NB) : "memory");
    else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bh[ks][0]), "+v"(bh[ks][1]) : "n"(NB) : "memory");
  };
  int ct = t_begin, cchunk = 0;
  Coord cur = decode(ct);
  load_b_half(0, 0, cur.nt * BN);
  load_b_half(1, 0, cur.nt * BN);
  float esc[2], esh[2];
  for (int s = 0; s <= total; ++s) {
    // the A tile of step s-1 is complete; LDS reads of the previous step have returned (they fed its MFMAs)
    SF_STAMP(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    SF_STAMP(1);
    if (s == 0 || !SF_DO_MFMA) continue;
    const int n0 = cur.nt * BN;
    if (cchunk == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    // the step after this one: its weights are requested as soon as a half's registers are free
    const bool last = cchunk + 1 == KC;
    const int nchunk = last ? 0 : cchunk + 1;
    Coord nxt = cur;
    if (last) nxt = decode(min(ct + G, p.ntiles - 1));     // (three divisions: once per tile, not per step)
    const int nn0 = nxt.nt * BN;
    const unsigned aoff = (unsigned)((s - 1) & 1) * 16384u;
    auto half = [&](const int ks) {
      sf_f16x8 ah[TM], al[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = sf_ds_read_b128<0>(a_rd[ks] + aoff + i * 2048);
        if (SPLIT3) al[i] = sf_ds_read_b128<128 * 64>(a_rd[ks] + aoff + i * 2048);
      }
#pragma unroll
      for (int i = 0; i < TM; i += 2) {
        if (SPLIT3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[i]), "+v"(ah[i + 1]), "+v"(al[i]), "+v"(al[i + 1])::"memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[i]), "+v"(ah[i + 1])::"memory");
      }
      wait_b_half(ks);
      // products in the conv kernel's order (lo*hi, hi*lo, hi*hi)
      if (SPLIT3) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[ks][j], acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[ks][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_b_half(ks, nchunk, nn0);
      __builtin_amdgcn_sched_barrier(0);
    };
    half(0);
    SF_STAMP(2);
    half(1);
    SF_STAMP(3);
    cchunk = nchunk;
    if (!last) continue;
    if (!SF_DO_STORE) {   // probe: keep the accumulators alive without the epilogue's stores
      float z = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) z += acc[i][j][r];
      if (z == 1234.5f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, z), orsrc, 0u, 0u, 0);
      ct += G;
      cur = nxt;
      continue;
    }

    // ---- epilogue of the tile (as in the kernel above; scale / shift from LDS: a global load here would sit in the
    //      same in-order vmcnt queue as the weight requests of the next step) ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      esc[j] = s_bn[0][n0 + wn * 64 + j * 32 + frow];
      esh[j] = s_bn[1][n0 + wn * 64 + j * 32 + frow];
    }
    // Instruction diet (the epilogue runs on the wave that owns the matrix pipe, with the producers parked at the
    // barrier): an interior tile's store predicate depends on the register index and the lane half only -- windows /
    // columns of the upper half that fall outside the tile are the same two in every interior tile -- so the per-store
    // select is hoisted into two offsets; edge tiles keep the per-element form.
    const int y0 = cur.ty * SF_R, x0 = HPOOL ? cur.tx * SF_XP - p.pool_pad_l : cur.tx * SF_X;
    unsigned lane_off_j[2], tail_off_j[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int co = n0 + wn * 64 + j * 32 + frow;
      lane_off_j[j] = co < p.ldo ? (unsigned)(((HPOOL ? 8 : 4) * fh * p.ldo + co) * 4) : 0xffffffffu;
      tail_off_j[j] = fh ? 0xffffffffu : lane_off_j[j];       // upper half: windows 6, 7 / columns 30, 31 do not exist
    }
    if (HPOOL) {
      const int nwin = min(SF_XP / 2, p.Wo - cur.tx * (SF_XP / 2));          // wave-uniform
      const int klim = nwin - 8 * fh;
      const bool edge = x0 < 0 || x0 + 32 > p.W;
      const bool interior = nwin == SF_XP / 2;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int y = y0 + wm * TM + i;
        if (y >= p.H) continue;
        const unsigned row_off = (unsigned)((((size_t)cur.n * p.H + y) * p.Wo + cur.tx * (SF_XP / 2)) * p.ldo * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fmaf(acc[i][j][r], esc[j], esh[j]);
          if (p.relu_out) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = sf_relu(v[r]);   // one v_max (fmaxf: + a canonicalising one)
          }
          if (edge) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int col = x0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
              if ((unsigned)col >= (unsigned)p.W) v[r] = -INFINITY;
            }
          }
          float X[8], Y[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            X[r] = v[r];
            Y[r] = v[r + 8];
            sf_permlane32_swap(X[r], Y[r]);
          }
          auto colv = [&](int q) { return (q & 4) ? Y[(q & 3) + 4 * (q >> 3)] : X[(q & 3) + 4 * (q >> 3)]; };
          float c16a = X[0], col16 = X[0];
          sf_permlane32_swap(c16a, col16);
          float m[8];
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            const float c2 = kk < 7 ? colv(2 * kk + 2) : col16;
            m[kk] = fmaxf(fmaxf(colv(2 * kk), colv(2 * kk + 1)), c2);
          }
          if (interior) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m[kk]), orsrc, kk < 6 ? lane_off_j[j] : tail_off_j[j],
                                                    row_off + kk * p.ldo * 4, 0);
          } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m[kk]), orsrc, kk < klim ? lane_off_j[j] : 0xffffffffu,
                                                    row_off + kk * p.ldo * 4, 0);
          }
        }
      }
    } else {
      const int ncol = min(SF_X, p.W - x0);                                  // wave-uniform
      const int lim = ncol - 4 * fh;
      const bool interior = ncol == SF_X;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int y = y0 + wm * TM + i;
        if (y >= p.H) continue;
        const unsigned row_off = (unsigned)((((size_t)cur.n * p.H + y) * p.W + x0) * p.ldo * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fmaf(acc[i][j][r], esc[j], esh[j]);
          if (p.relu_out) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = sf_relu(v[r]);   // one v_max (fmaxf: + a canonicalising one)
          }
          if (interior) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int c = (r & 3) + 8 * (r >> 2);
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), orsrc, c < 26 ? lane_off_j[j] : tail_off_j[j],
                                                    row_off + c * p.ldo * 4, 0);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int c = (r & 3) + 8 * (r >> 2);
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), orsrc, c < lim ? lane_off_j[j] : 0xffffffffu,
                                                    row_off + c * p.ldo * 4, 0);
            }
          }
        }
      }
    }
    ct += G;
    cur = nxt;
  }
}

#if SF_PROBE_LEVEL == 9
unsigned long long* g_probe_dbg = nullptr;
#endif

bool sepconv_fused_supported(int cin_ld, int cout_pad, int dil) {
  // 128 outputs: one 128-wide pass; multiples of 256 (block4_sepconv1: 728 -> 768): 256-wide passes over the same patch
  return dil == 1 && cin_ld % 32 == 0 && cin_ld <= SF_KMAX && (cout_pad == 128 || (cout_pad % 256 == 0 && cout_pad <= 1024));
}

// One launch per image range whose input stays below 2 GiB (32-bit buffer offsets of the LDS DMA).
// pool_pad_l >= 0: the horizontally pooled form (out = [N][H][(W+1)/2][ldo], see SepFusedParams)
int launch_sepconv_fused(const float* in, const float* w9c, const unsigned short* wt_hi_blocked,
                         const unsigned short* wt_lo_blocked, const float* scale, const float* shift, float* out, int N,
                         int H, int W, int ld, int ldo, int cout_pad, int relu_in, int relu_out, hipStream_t s,
                         int pool_pad_l) {
  XDET_REQUIRE(sepconv_fused_supported(ld, cout_pad, 1), "sepconv_fused: unsupported channel counts");
  XDET_REQUIRE(in && w9c && wt_hi_blocked && scale && shift && out, "sepconv_fused: NULL argument");
  const size_t per_image = (size_t)H * W * std::max(ld, ldo) * 4;   // input offsets are signed 32-bit, output unsigned
  XDET_REQUIRE(per_image < ((size_t)1 << 31), "sepconv_fused: one image exceeds 2 GiB");
  const int n_max = (int)std::max<size_t>(1, (((size_t)1 << 31) - 1) / per_image);
  for (int nb = 0; nb < N; nb += n_max) {
    const int n = std::min(n_max, N - nb);
    SepFusedParams p;
    p.in = in + (size_t)nb * H * W * ld;
    p.w9c = w9c; p.wt_hi = wt_hi_blocked; p.wt_lo = wt_lo_blocked; p.scale = scale; p.shift = shift;
    p.out = out + (size_t)nb * H * (pool_pad_l >= 0 ? (W + 1) / 2 : W) * ldo;
    p.N = n; p.H = H; p.W = W; p.ld = ld; p.ldo = ldo; p.Cout_pad = cout_pad; p.relu_in = relu_in; p.relu_out = relu_out;
    const bool hpool = pool_pad_l >= 0;
    // 256 output channels: one pass of the 1 x 4 wave layout (XDET_SEPCONV_TWO_PASS=1: two 128-wide passes, for A/B runs)
    static const bool two_pass = getenv("XDET_SEPCONV_TWO_PASS") != nullptr;
    const bool wide = cout_pad % 256 == 0 && !two_pass;
    p.Wo = (W + 1) / 2; p.pool_pad_l = hpool ? pool_pad_l : 0;
#if SF_PROBE_LEVEL == 9
    p.dbg = g_probe_dbg;
#endif
    p.TY = (int)cdiv(H, SF_R); p.TX = hpool ? (int)cdiv(p.Wo, SF_XP / 2) : (int)cdiv(W, SF_X); p.NT = wide ? cout_pad / 256 : cout_pad / 128;
    const int64_t nt = (int64_t)n * p.TY * p.TX * p.NT;
    p.ntiles = (int)nt;
    p.tiles_per_block = 0;
    // producer / consumer form: one 512-thread workgroup per CU; XDET_SEPCONV_SERIAL=1: the four-wave form above, two
    // workgroups per CU (LDS) -- for A/B runs; a multiple of 8 so that every XCD gets the same number
    static const bool serial = getenv("XDET_SEPCONV_SERIAL") != nullptr;
    const dim3 g((unsigned)std::min<int64_t>(serial ? 512 : 256, cdiv(nt, 8) * 8));
    auto go = [&](auto kern, int threads) { hipLaunchKernelGGL(kern, g, dim3(threads), 0, s, p); };
    auto pick = [&](auto split3, auto relu, auto pool) {
      constexpr bool S3 = decltype(split3)::value, RL = decltype(relu)::value, PL = decltype(pool)::value;
      if (serial) {
        if (wide) go(sepconv_fused_kernel<S3, RL, PL, 4>, 256);
        else go(sepconv_fused_kernel<S3, RL, PL, 2>, 256);
      } else {
        if (wide) go(sepconv_pc_kernel<S3, RL, PL, 4>, 512);
        else go(sepconv_pc_kernel<S3, RL, PL, 2>, 512);
      }
    };
    auto pick2 = [&](auto split3, auto relu) {
      if (hpool) pick(split3, relu, std::true_type{});
      else pick(split3, relu, std::false_type{});
    };
    auto pick1 = [&](auto split3) {
      if (relu_in) pick2(split3, std::true_type{});
      else pick2(split3, std::false_type{});
    };
    if (wt_lo_blocked) pick1(std::true_type{});
    else pick1(std::false_type{});
    XDET_LAUNCH_CHECK();
  }
  return XDET_OK;
}

}  // namespace xdet
