// Fused separable block for the entry flow: (ReLU ->) depthwise 3x3 -> pointwise 1x1 -> BN (-> ReLU) in ONE
// kernel, "nothing in between" (net/xception_body.py:220-234).  The two-kernel form moves the depthwise result
// through HBM as split f16 planes (4 B written + 4 B read per element); here it never leaves the CU:
//
//   patch of the f32 input (6 x 32 pixels x 32 channels)  --buffer_load ... lds (zero fill = SAME padding)-->  LDS
//   3x3 stencil in VALU (same FMA order as depthwise3x3_tile_kernel) -> hi/lo f16 -> LDS A tile [128 px][32 ch]
//   A x W (K-blocked f16 hi/lo weights straight from L2 into registers)  --3 x v_mfma_f32_32x32x16_f16-->  acc
//   after the last channel chunk: y = acc * scale + shift (folded BN), optional ReLU, f32 NHWC store
//   (straight from the accumulators: no LDS bounce)
//
// Tiles of 4 rows x 30 pixels (a 128-row GEMM tile with 8 idle rows: 30 + 2 halo pixels are exactly four 1 KB DMA
// pieces) x 128 or 256 output channels; channel chunks of 32 are the K steps.  Arithmetic and accumulation order are
// exactly those of depthwise3x3_tile_kernel + split + conv_dma_f16_kernel, so the result is bit-identical to the
// two-kernel form (tests/test_gpu_layers.py).  Wider layers run as 256-wide passes over the same patch (block4_sepconv1:
// three); the 30 x 30 layers (728 channels) stay on the two-kernel path (profiles/NOTES_r05.md: why).
// Rounds 2-4 ran the phases back to back on four waves, two workgroups per CU (git history; the same-box A/B against
// that library is in profiles/r05_ab_lines.txt); round 5: sepconv_pc_kernel below.
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

// ---------------------------------------------------------------------------------------------------------------
// Producer / consumer form (round 5).  Rounds 2-4 ran DMA wait -> stencil -> split -> A tile -> barrier -> MFMA back to
// back on the same four waves, two workgroups per CU (5,500 clocks per 32-channel chunk and CU; the 119^2 layers at
// 2.2-3.2 TB/s with the matrix pipe 24-35 % busy).  Here the phases are different WAVES of one 512-thread workgroup, one
// of each on every SIMD:
//   waves 0-3 (producers): patch DMA three steps deep, 3x3 stencil, hi/lo split, A tile of step s into s_a[s & 1]
//   waves 4-7 (consumers): the MFMAs of step s-1 out of s_a[(s-1) & 1] against weights that were requested one
//                          half-step earlier, and the tile's epilogue (folded BN / ReLU / h-pool) after its last chunk
// with ONE s_barrier per step (a step = one 32-channel chunk of one tile): a SIMD's VALU (producer) and its matrix
// pipe (consumer) work on adjacent steps at the same time -- as far as they can: measured, a co-resident wave's VALU
// instructions cost the MFMA stream ~7 clocks each, so a step costs the SUM of both streams and what made the kernel
// faster than its predecessor was the lower instruction count (profiles/NOTES_r05.md).  Step order, FMA order, product
// order and K order are the predecessor's: bit-identical to the two-kernel path (tests/test_gpu_layers.py).
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
  // Tile schedule.  Workgroups are dealt round-robin to the 8 XCDs (private 4 MB L2s).  XCD x owns the contiguous tile
  // band [x*per_xcd, (x+1)*per_xcd); its resident workgroups take the band's tiles in an interleaved order (workgroup j:
  // tiles j, j+G, j+2G, ...), so at any moment one XCD works on ~G CONSECUTIVE tiles: vertical neighbours (shared halo
  // rows) and the N passes of a tile read the same input lines within microseconds of each other and meet in that
  // XCD's L2.  One workgroup per CU.
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

    // ---- epilogue of the tile (scale / shift from LDS: a global load here would sit in the
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
    // 256 output channels: one pass of the 1 x 4 wave layout
    const bool wide = cout_pad % 256 == 0;
    p.Wo = (W + 1) / 2; p.pool_pad_l = hpool ? pool_pad_l : 0;
#if SF_PROBE_LEVEL == 9
    p.dbg = g_probe_dbg;
#endif
    p.TY = (int)cdiv(H, SF_R); p.TX = hpool ? (int)cdiv(p.Wo, SF_XP / 2) : (int)cdiv(W, SF_X); p.NT = wide ? cout_pad / 256 : cout_pad / 128;
    const int64_t nt = (int64_t)n * p.TY * p.TX * p.NT;
    p.ntiles = (int)nt;
    p.tiles_per_block = 0;
    // one 512-thread workgroup per CU; a multiple of 8 so that every XCD gets the same number
    const dim3 g((unsigned)std::min<int64_t>(256, cdiv(nt, 8) * 8));
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, g, dim3(512), 0, s, p); };
    auto pick = [&](auto split3, auto relu, auto pool) {
      constexpr bool S3 = decltype(split3)::value, RL = decltype(relu)::value, PL = decltype(pool)::value;
      if (wide) go(sepconv_pc_kernel<S3, RL, PL, 4>);
      else go(sepconv_pc_kernel<S3, RL, PL, 2>);
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
