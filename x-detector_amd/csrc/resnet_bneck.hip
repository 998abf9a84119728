// One ResNet v2 bottleneck block with an identity shortcut (net/resnet_v2.py:142-184) in ONE kernel:
//
//     out = x + conv1x1_c( relu(bn_c( conv3x3_b( relu(bn_b( conv1x1_a( relu(bn_a(x)) ))) ))) )
//
// As three launches (conv_mfma_dma.hip) a stage-1 block at BASELINE config 2 (8 x 120 x 120 pixels, 256 -> 64 -> 64 -> 256
// channels) moves relu(bn_a(x)) in (118 MB), the two 64-channel intermediates out and in again (4 x 29 MB), x in and the
// output out twice (f32 + the next block's pre-activation planes: 3 x 118 MB): 150 us for 16 GFLOP.  Here a workgroup
// keeps a tile's intermediates on the CU:
//
//   phase 1  conv1x1_a on the tile's (R + 2) x 32-pixel halo patch: A from the pre-activation planes by LDS DMA (two
//            stages), W_a streamed through a four-stage LDS ring; epilogue bn_b + ReLU, out-of-image pixels forced to 0
//            (they are conv3x3_b's SAME padding), split hi / lo -> `mid1` in LDS
//   phase 2  conv3x3_b: the nine taps are shifted fragment reads of mid1 (as conv3x3_patch.hip), W_b through the same ring
//            (K order channel chunk outer, tap inner: conv_dma_f16_kernel's); epilogue bn_c + ReLU, split -> `mid2` in LDS
//   phase 3  conv1x1_c: A from mid2, this wave's 32 output channels of W_c live in registers for the whole (persistent)
//            workgroup; epilogue + x (the identity shortcut) -> out (f32) and relu(bn_next(out)) as split planes for the
//            next block's conv1x1_a
//
// Tile = R rows x 30 pixels of one image (32-row GEMM blocks with two idle rows: the 30 + 2 halo pixels of a patch row
// are exactly two 16-pixel DMA pieces).  All MFMAs are issued with the operands swapped (weights as the first operand):
// the accumulators then hold one PIXEL per lane and four consecutive channels per register group, which is the layout
// mid1 / mid2 / the output rows want (8- and 16-byte stores) -- the products and their order per accumulator are those
// of conv_dma_f16_kernel (lo*hi, hi*lo, hi*hi per 16-deep half, K ascending), and the matrix unit's k-sum does not depend
// on which operand a factor comes from (profiles/NOTES_r05.md 7), so the block's output is bit-identical to the
// three-launch path (tests/test_gpu_resnet.py).
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace xdet {

typedef float bk_f32x16 __attribute__((ext_vector_type(16)));
typedef float bk_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 bk_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 bk_f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned bk_u2 __attribute__((ext_vector_type(2)));
typedef unsigned bk_u4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

struct BneckParams {
  const u16* xin_hi; const u16* xin_lo;    // relu(bn_a(x)) as planes [pix/16][Cin/32][16][32]
  const float* x;                          // the block input, NHWC f32, channel stride Cout (= Cin)
  const u16* wa_hi; const u16* wa_lo;      // K-blocked [Cin/32][CMID][32]
  const u16* wb_hi; const u16* wb_lo;      // K-blocked [9 * CMID/32][CMID][32], K block = tap * (CMID/32) + chunk
  const u16* wc_hi; const u16* wc_lo;      // K-blocked [CMID/32][Cout][32]
  const float* sc_a; const float* sh_a;    // folded bn_b (and conv1x1_a's weight pre-scale), [CMID]
  const float* sc_b; const float* sh_b;    // folded bn_c, [CMID]
  const float* sc_c; const float* sh_c;    // conv1x1_c's weight pre-scale / zero shift, [Cout]
  const float* pl_sc; const float* pl_sh;  // bn_next folded for the planes copy, [Cout] (NULL: no planes)
  float* out;                              // NHWC f32, channel stride Cout
  u16* out_hi; u16* out_lo;                // relu(out * pl_sc + pl_sh) as planes [pix/16][Cout/32][16][32]
  int N, H, W, Cin;
  int TY, TX, ntiles;
  int dbg;   // XDET_BNECK_DEBUG=1|2 (diagnosis only): channels 0..CMID-1 of `out` receive mid1 / mid2 (hi + lo) instead of the result
};

__device__ __forceinline__ unsigned bk_lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) void*)(p);
}
// LDS accesses next to in-flight LDS-DMA writes are inline asm: the compiler would put s_waitcnt vmcnt(0) in front of each
template <int OFF>
__device__ __forceinline__ bk_f16x8 bk_ds_read_h8(unsigned addr) {
  bk_f16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int OFF>
__device__ __forceinline__ bk_f32x4 bk_ds_read_f4(unsigned addr) {
  bk_f32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int OFF>
__device__ __forceinline__ void bk_ds_write_b64(unsigned addr, bk_u2 v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// ReLU that keeps NaN (conv_epilogue.h ep_relu)
__device__ __forceinline__ float bk_relu(float v) { return __builtin_elementwise_maximum(v, 0.f); }

// hi = f16(v), lo = f16(v - float(hi)) of four values (the conv epilogue's planes copy, conv_epilogue.h)
__device__ __forceinline__ void bk_split4(const float (&t)[4], bk_u2* h, bk_u2* l) {
  const _Float16 h0 = (_Float16)t[0], h1 = (_Float16)t[1], h2 = (_Float16)t[2], h3 = (_Float16)t[3];
  const bk_f16x4 hv = {h0, h1, h2, h3};
  const bk_f16x4 lv = {(_Float16)(t[0] - (float)h0), (_Float16)(t[1] - (float)h1), (_Float16)(t[2] - (float)h2),
                       (_Float16)(t[3] - (float)h3)};
  *h = __builtin_bit_cast(bk_u2, hv);
  *l = __builtin_bit_cast(bk_u2, lv);
}

template <int CMID, int R, int COUT>
struct BneckGeom {
  static constexpr int NB = CMID / 32;             // 32-column blocks of the two inner convs
  static constexpr int PR = R + 2;                 // patch rows
  static constexpr int M1 = PR * 32;               // GEMM rows of phase 1
  static constexpr int NBLK1 = PR * NB;            // accumulator blocks of phase 1 (8 waves: one or two each)
  static constexpr int NCC2 = CMID / 32, NK2 = 9 * NCC2;
  static constexpr int NB3 = COUT / 256;           // column blocks per wave in phase 3
  static constexpr int A_PLANE = M1 * 64, A_STAGE = 2 * A_PLANE;
  static constexpr int B_PLANE = CMID * 64, B_STAGE = 2 * B_PLANE;
  static constexpr int MP = CMID * 4 + 16;         // bytes per mid row: hi | lo | pad (pitch = 4 banks mod 64: 16 consecutive rows
                                                   // of a ds_read_b128 lane group hit 16 different 4-bank windows)
  static constexpr int OFF_A = 0;                  // two A stages (phase 1); mid2 (phases 2 -> 3) lies over them
  static constexpr int OFF_B = 2 * A_STAGE;        // four weight stages
  static constexpr int OFF_M1 = OFF_B + 4 * B_STAGE;
  static constexpr int OFF_T = OFF_M1 + (M1 + 4) * MP;   // (+4 rows: the idle columns 30, 31 read up to 3 pixels past the patch)
  static constexpr int T_FLOATS = 4 * CMID + 4 * COUT;
  static constexpr int LDS_BYTES = OFF_T + T_FLOATS * 4;
  static constexpr int APW = PR / 2;               // A pieces (16 rows x 64 B of one plane) per wave and K step
  static constexpr int BPW = CMID / 64;            // weight pieces per wave and K step
  static_assert(PR % 2 == 0 && CMID % 64 == 0 && COUT % 256 == 0, "piece counts");
  static_assert(R * 32 * MP <= 2 * A_STAGE, "mid2 lies over the A stages");
  static_assert(R * NB == 8, "phase 2: one accumulator block per wave");
  static_assert(NBLK1 <= 16, "phase 1: at most two accumulator blocks per wave");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int CMID, int R, int COUT>
__global__ __launch_bounds__(512) void resnet_bneck_kernel(BneckParams p) {
  using G = BneckGeom<CMID, R, COUT>;
  constexpr int NB = G::NB, PR = G::PR, M1 = G::M1, NBLK1 = G::NBLK1, NCC2 = G::NCC2, NK2 = G::NK2, NB3 = G::NB3;
  constexpr int A_PLANE = G::A_PLANE, A_STAGE = G::A_STAGE, B_PLANE = G::B_PLANE, B_STAGE = G::B_STAGE, MP = G::MP;
  constexpr int APW = G::APW, BPW = G::BPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fh = lane >> 5;
  const int lr = lane >> 2, pos = lane & 3;
  // persistent workgroups, XCD-banded tile order (sepconv_fused.hip): an XCD works on consecutive tiles, whose halo rows
  // and weights meet in its L2
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, GW = gridDim.x >> 3;
  const int per_xcd = (p.ntiles + 7) >> 3;
  const int t_begin = xcd * per_xcd + wg;
  const int t_end = min(p.ntiles, (xcd + 1) * per_xcd);
  if (t_begin >= t_end) return;

  const int NK1 = p.Cin >> 5;
  const unsigned c32i = (unsigned)(p.Cin >> 5);
  const unsigned lds0 = bk_lds_addr(smem);

  // ---- tables: folded BNs of the three epilogues ----
  {
    float* T = reinterpret_cast<float*>(smem + G::OFF_T);
    for (int i = tid; i < CMID; i += 512) {
      T[i] = p.sc_a[i]; T[CMID + i] = p.sh_a[i]; T[2 * CMID + i] = p.sc_b[i]; T[3 * CMID + i] = p.sh_b[i];
    }
    for (int i = tid; i < COUT; i += 512) {
      T[4 * CMID + i] = p.sc_c[i]; T[4 * CMID + COUT + i] = p.sh_c[i];
      T[4 * CMID + 2 * COUT + i] = p.pl_sc ? p.pl_sc[i] : 1.f;
      T[4 * CMID + 3 * COUT + i] = p.pl_sh ? p.pl_sh[i] : 0.f;
    }
  }
  const unsigned t_a = lds0 + G::OFF_T, t_b = t_a + 2 * CMID * 4, t_c = t_a + 4 * CMID * 4;

  // ---- buffer resources ----
  const size_t npix = (size_t)p.N * p.H * p.W;
  const unsigned xin_bytes = (unsigned)((((npix + 15) >> 4) * c32i) << 10);
  const __amdgpu_buffer_rsrc_t r_xin = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>((wave & 1) ? p.xin_lo : p.xin_hi), 0, (int)xin_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_wa = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>((wave & 1) ? p.wa_lo : p.wa_hi), 0, NK1 * CMID * 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_wb = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>((wave & 1) ? p.wb_lo : p.wb_hi), 0, NK2 * CMID * 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)(unsigned)(npix * COUT * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(unsigned)(npix * COUT * 4), 0x00020000);
  const unsigned pl_bytes = p.out_hi ? (unsigned)((((npix + 15) >> 4) * (COUT / 32)) << 10) : 0u;
  const __amdgpu_buffer_rsrc_t r_ohi = __builtin_amdgcn_make_buffer_rsrc(p.out_hi, 0, (int)pl_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_olo = __builtin_amdgcn_make_buffer_rsrc(p.out_lo, 0, (int)pl_bytes, 0x00020000);

  // ---- DMA pieces of this wave: plane = wave & 1 (hi / lo) ----
  // A (phase 1): 16-row groups g = (wave >> 1) + 4 * jj of the M1 patch rows; a lane fetches the 16-byte chunk
  // pos ^ ((row >> 2) & 3) of its row (the chunk permutation that makes the fragment reads conflict-free)
  // weights: 16-row groups gb = (wave >> 1) + 4 * jj of the CMID rows of a K block
  unsigned b_vo[BPW];
#pragma unroll
  for (int jj = 0; jj < BPW; ++jj) {
    const int row = ((wave >> 1) + 4 * jj) * 16 + lr;
    b_vo[jj] = (unsigned)(row * 64 + ((pos ^ ((row >> 2) & 3)) << 4));
  }
  unsigned a_vo[APW];
  struct Coord { int n, ty, tx; };
  auto decode = [&](int q) {
    Coord c;
    c.tx = q % p.TX; q /= p.TX;
    c.ty = q % p.TY;
    c.n = q / p.TY;
    return c;
  };
  auto tile_offsets = [&](const Coord& c) {
    const int y0 = c.ty * R, x0 = c.tx * 30;
#pragma unroll
    for (int jj = 0; jj < APW; ++jj) {
      const int rt = ((wave >> 1) + 4 * jj) * 16 + lr;
      const int y = y0 - 1 + (rt >> 5), x = x0 - 1 + (rt & 31);
      const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const unsigned pix = (unsigned)((c.n * p.H + y) * p.W + x);
      a_vo[jj] = ok ? (((pix >> 4) * c32i) << 10) + ((pix & 15) << 6) + (unsigned)((pos ^ ((rt >> 2) & 3)) << 4) : 0xffffffffu;
    }
  };
  auto issue_a = [&](int kt, int stage) {
#pragma unroll
    for (int jj = 0; jj < APW; ++jj)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          r_xin, (__attribute__((address_space(3))) void*)(smem + G::OFF_A + stage * A_STAGE + (wave & 1) * A_PLANE + ((wave >> 1) + 4 * jj) * 1024),
          16, (int)a_vo[jj], kt << 10, 0, 0);
  };
  // weight stream of a tile: steps 0 .. NK1-1 = K blocks of W_a, then NK2 steps of W_b in (chunk, tap) order
  auto issue_b = [&](int j) {
    const int slot = j & 3;
    unsigned char* dst = smem + G::OFF_B + slot * B_STAGE + (wave & 1) * B_PLANE;
    if (j < NK1) {
#pragma unroll
      for (int jj = 0; jj < BPW; ++jj)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_wa, (__attribute__((address_space(3))) void*)(dst + ((wave >> 1) + 4 * jj) * 1024), 16,
                                                 (int)b_vo[jj], j * (CMID * 64), 0, 0);
    } else {
      const int j2 = j - NK1;
      const int cc = j2 / 9, tap = j2 - cc * 9;
#pragma unroll
      for (int jj = 0; jj < BPW; ++jj)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_wb, (__attribute__((address_space(3))) void*)(dst + ((wave >> 1) + 4 * jj) * 1024), 16,
                                                 (int)b_vo[jj], (tap * NCC2 + cc) * (CMID * 64), 0, 0);
    }
  };

  // ---- conv1x1_c's weights of this wave's output channels: registers, for the lifetime of the workgroup ----
  bk_f16x8 wch[NB3][NCC2][2], wcl[NB3][NCC2][2];
#pragma unroll
  for (int j = 0; j < NB3; ++j)
#pragma unroll
    for (int kb = 0; kb < NCC2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const size_t o = ((size_t)kb * COUT + (wave * NB3 + j) * 32 + frow) * 32 + (ks * 2 + fh) * 8;
        wch[j][kb][ks] = *reinterpret_cast<const bk_f16x8*>(p.wc_hi + o);
        wcl[j][kb][ks] = *reinterpret_cast<const bk_f16x8*>(p.wc_lo + o);
      }

  // ---- fragment addresses ----
  // phase 1: blocks b = wave (and wave + 8 if it exists): (mi, nj) = (b / NB, b % NB)
  const int nj1 = wave % NB;
  const int mi1[2] = {wave / NB, (wave + 8) / NB};
  const bool two1 = wave + 8 < NBLK1;
  unsigned a1_off[2][2], b_off[2];             // [block][ks], [ks]: byte offsets inside an A stage / a weight stage (hi plane)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int c = ks * 2 + fh;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int rt = mi1[b] * 32 + frow;
      a1_off[b][ks] = (unsigned)(rt * 64 + ((c ^ ((rt >> 2) & 3)) << 4));
    }
  }
  // phases 1 and 2 read the same weight rows: phase 2's block of this wave is (mi2, nj2) with nj2 == nj1 when NB divides 8
  const int mi2 = wave / NB, nj2 = wave % NB;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int c = ks * 2 + fh;
    const int rb = nj1 * 32 + frow;
    b_off[ks] = (unsigned)(rb * 64 + ((c ^ ((rb >> 2) & 3)) << 4));
  }
  const unsigned m1_base = lds0 + G::OFF_M1, m2_base = lds0 + G::OFF_A;

  Coord cur = decode(t_begin);
  tile_offsets(cur);
  __syncthreads();                               // tables written
  issue_a(0, 0);
  issue_b(0);
  issue_b(1);

  for (int t = t_begin; t < t_end; t += GW) {
    const int y0 = cur.ty * R, x0 = cur.tx * 30;
    // =============================== phase 1: conv1x1_a on the halo patch ===============================
    bk_f32x16 acc1[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[b][r] = 0.f;
    for (int kt = 0; kt < NK1; ++kt) {
      // A(kt) and W(kt) have landed: younger than both is W(kt + 1) only (a tile's first step also waits for the
      // previous tile's epilogue stores: vmcnt retires in order); every wave is done with step kt - 1
      if (kt == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(BPW) : "memory");
      if (kt + 1 < NK1) issue_a(kt + 1, (kt + 1) & 1);
      issue_b(kt + 2);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned sa = lds0 + G::OFF_A + (kt & 1) * A_STAGE, sb = lds0 + G::OFF_B + (kt & 3) * B_STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bk_f16x8 bh = bk_ds_read_h8<0>(sb + b_off[ks]), bl = bk_ds_read_h8<B_PLANE>(sb + b_off[ks]);
        bk_f16x8 ah0 = bk_ds_read_h8<0>(sa + a1_off[0][ks]), al0 = bk_ds_read_h8<A_PLANE>(sa + a1_off[0][ks]);
        bk_f16x8 ah1 = ah0, al1 = al0;
        if (two1) {
          ah1 = bk_ds_read_h8<0>(sa + a1_off[1][ks]);
          al1 = bk_ds_read_h8<A_PLANE>(sa + a1_off[1][ks]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh), "+v"(bl), "+v"(ah0), "+v"(al0), "+v"(ah1), "+v"(al1)::"memory");
        acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al0, acc1[0], 0, 0, 0);
        if (two1) acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al1, acc1[1], 0, 0, 0);
        acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah0, acc1[0], 0, 0, 0);
        if (two1) acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah1, acc1[1], 0, 0, 0);
        acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah0, acc1[0], 0, 0, 0);
        if (two1) acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah1, acc1[1], 0, 0, 0);
      }
    }
    // ---- epilogue 1: bn_b + ReLU, zero outside the image, split -> mid1 ----
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (b == 1 && !two1) break;
      const int y = y0 - 1 + mi1[b], x = x0 - 1 + frow;
      const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const unsigned row = m1_base + (unsigned)((mi1[b] * 32 + frow) * MP);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = nj1 * 32 + 8 * q + 4 * fh;
        bk_f32x4 sc = bk_ds_read_f4<0>(t_a + c * 4), sh = bk_ds_read_f4<CMID * 4>(t_a + c * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc), "+v"(sh)::"memory");      // (tied: the FMAs must not move above the wait)
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[k] = bk_relu(fmaf(acc1[b][4 * q + k], sc[k], sh[k]));
          v[k] = ok ? v[k] : 0.f;
        }
        bk_u2 h, l;
        bk_split4(v, &h, &l);
        bk_ds_write_b64<0>(row + c * 2, h);
        bk_ds_write_b64<CMID * 2>(row + c * 2, l);
      }
    }
    // =============================== phase 2: conv3x3_b out of mid1 ===============================
    bk_f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll 1
    for (int cc = 0; cc < NCC2; ++cc) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int j = NK1 + cc * 9 + tap;
        // W(j) has landed (younger: W(j + 1), if there is one); mid1 is complete (first step) / every wave is done with W(j - 1)
        const bool last_cc = cc == NCC2 - 1;
        if (tap == 8 && last_cc) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(BPW) : "memory");
        if (!(last_cc && tap >= 7)) issue_b(j + 2);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned sb = lds0 + G::OFF_B + (j & 3) * B_STAGE;
        const int ky = tap / 3, kx = tap - ky * 3;
        const unsigned ar = m1_base + (unsigned)(((mi2 + ky) * 32 + frow + kx) * MP + cc * 64 + fh * 16);
        bk_f16x8 bh[2], bl[2], ah[2], al[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bh[ks] = bk_ds_read_h8<0>(sb + b_off[ks]);
          bl[ks] = bk_ds_read_h8<B_PLANE>(sb + b_off[ks]);
        }
        ah[0] = bk_ds_read_h8<0>(ar); al[0] = bk_ds_read_h8<CMID * 2>(ar);
        ah[1] = bk_ds_read_h8<32>(ar); al[1] = bk_ds_read_h8<CMID * 2 + 32>(ar);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[0]), "+v"(bl[0]), "+v"(bh[1]), "+v"(bl[1]), "+v"(ah[0]), "+v"(al[0]), "+v"(ah[1]), "+v"(al[1])::"memory");
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ks], al[ks], acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ks], ah[ks], acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ks], ah[ks], acc2, 0, 0, 0);
        }
      }
    }
    // ---- epilogue 2: bn_c + ReLU, split -> mid2 (over the A stages: nothing of phase 1 is in flight or being read) ----
    {
      const unsigned row = m2_base + (unsigned)((mi2 * 32 + frow) * MP);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = nj2 * 32 + 8 * q + 4 * fh;
        bk_f32x4 sc = bk_ds_read_f4<0>(t_b + c * 4), sh = bk_ds_read_f4<CMID * 4>(t_b + c * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc), "+v"(sh)::"memory");
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = bk_relu(fmaf(acc2[4 * q + k], sc[k], sh[k]));
        bk_u2 h, l;
        bk_split4(v, &h, &l);
        bk_ds_write_b64<0>(row + c * 2, h);
        bk_ds_write_b64<CMID * 2>(row + c * 2, l);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // mid2 complete
    // =============================== phase 3: conv1x1_c out of mid2 ===============================
    bk_f32x16 acc3[R][NB3];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < NB3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[i][j][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NCC2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bk_f16x8 ah[R], al[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const unsigned ar = m2_base + (unsigned)((i * 32 + frow) * MP + kb * 64 + (ks * 2 + fh) * 16);
          ah[i] = bk_ds_read_h8<0>(ar);
          al[i] = bk_ds_read_h8<CMID * 2>(ar);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[i]), "+v"(al[i])::"memory");
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
          for (int j = 0; j < NB3; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wch[j][kb][ks], al[i], acc3[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
          for (int j = 0; j < NB3; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcl[j][kb][ks], ah[i], acc3[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
          for (int j = 0; j < NB3; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wch[j][kb][ks], ah[i], acc3[i][j], 0, 0, 0);
      }
    // every wave is done reading mid2: the next tile's first stages may land on it while this tile's output is written
    asm volatile("s_barrier" ::: "memory");
    const Coord nxt = decode(min(t + GW, p.ntiles - 1));
    const Coord me = cur;
    auto prefetch_next = [&]() {
      if (t + GW < t_end) {
        cur = nxt;
        tile_offsets(cur);
        issue_a(0, 0);
        issue_b(0);
        issue_b(1);
      }
    };
    if (!p.dbg) prefetch_next();
    // ---- epilogue 3: + x, f32 out and the next block's pre-activation planes ----
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int y = y0 + i, x = x0 + frow;
      const bool ok = y < p.H && frow < 30 && x < p.W;
      const unsigned pix = (unsigned)((me.n * p.H + y) * p.W + x);
#pragma unroll
      for (int j = 0; j < NB3; ++j) {
        const int cb = (wave * NB3 + j) * 32;
        bk_f32x4 res[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned vo = ok ? (pix * COUT + (unsigned)(cb + 8 * q + 4 * fh)) * 4u : 0xffffffffu;
          res[q] = __builtin_bit_cast(bk_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)vo, 0, 0));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = cb + 8 * q + 4 * fh;
          bk_f32x4 sc = bk_ds_read_f4<0>(t_c + c * 4), sh = bk_ds_read_f4<COUT * 4>(t_c + c * 4);
          bk_f32x4 psc = bk_ds_read_f4<2 * COUT * 4>(t_c + c * 4), psh = bk_ds_read_f4<3 * COUT * 4>(t_c + c * 4);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc), "+v"(sh), "+v"(psc), "+v"(psh)::"memory");
          bk_f32x4 v;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[k] = fmaf(acc3[i][j][4 * q + k], sc[k], sh[k]);
            v[k] += res[q][k];
          }
          if (p.dbg) {
            bk_f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (cb < CMID) {
              const unsigned base = p.dbg == 1 ? m1_base + (unsigned)(((i + 1) * 32 + frow + 1) * MP) : m2_base + (unsigned)((i * 32 + frow) * MP);
              const bk_f16x4 hh = *reinterpret_cast<const bk_f16x4*>(smem + (base - lds0) + c * 2);
              const bk_f16x4 ll = *reinterpret_cast<const bk_f16x4*>(smem + (base - lds0) + CMID * 2 + c * 2);
#pragma unroll
              for (int k = 0; k < 4; ++k) z[k] = (float)hh[k] + (float)ll[k];
            }
            v = z;
          }
          const unsigned vo = ok ? (pix * COUT + (unsigned)c) * 4u : 0xffffffffu;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bk_u4, v), r_out, (int)vo, 0, 0);
          if (p.out_hi) {
            float tt[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) tt[k] = bk_relu(fmaf(v[k], psc[k], psh[k]));
            bk_u2 h, l;
            bk_split4(tt, &h, &l);
            const unsigned po = ok ? ((((pix >> 4) * (COUT / 32) + (unsigned)(c >> 5)) << 10) + ((pix & 15) << 6) + (unsigned)(c & 31) * 2u) : 0xffffffffu;
            __builtin_amdgcn_raw_buffer_store_b64(h, r_ohi, (int)po, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(l, r_olo, (int)po, 0, 0);
          }
        }
      }
    }
    if (p.dbg) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      prefetch_next();
    }
  }
}

bool resnet_bneck_supported(int cin, int cmid, int cout, int H, int W, int N) {
  const size_t npix = (size_t)N * H * W;
  return cin == cout && cmid == 64 && cout == 256 && npix * cout * 4 < ((size_t)1 << 32);
}

int launch_resnet_bneck(const BneckLaunch& a, int N, hipStream_t s) {
  XDET_REQUIRE(resnet_bneck_supported(a.cin, a.cmid, a.cout, a.H, a.W, N), "resnet_bneck: unsupported block geometry");
  XDET_REQUIRE(a.xin_hi && a.xin_lo && a.x && a.out && a.wa_hi && a.wa_lo && a.wb_hi && a.wb_lo && a.wc_hi && a.wc_lo,
               "resnet_bneck: NULL argument");
  if (N <= 0) return XDET_OK;
  constexpr int R = 4;
  BneckParams p;
  p.xin_hi = a.xin_hi; p.xin_lo = a.xin_lo; p.x = a.x;
  p.wa_hi = a.wa_hi; p.wa_lo = a.wa_lo; p.wb_hi = a.wb_hi; p.wb_lo = a.wb_lo; p.wc_hi = a.wc_hi; p.wc_lo = a.wc_lo;
  p.sc_a = a.sc_a; p.sh_a = a.sh_a; p.sc_b = a.sc_b; p.sh_b = a.sh_b; p.sc_c = a.sc_c; p.sh_c = a.sh_c;
  p.pl_sc = a.out_hi ? a.pl_sc : nullptr; p.pl_sh = a.out_hi ? a.pl_sh : nullptr;
  p.out = a.out; p.out_hi = a.out_hi; p.out_lo = a.out_lo;
  p.N = N; p.H = a.H; p.W = a.W; p.Cin = a.cin;
  p.TY = (int)cdiv(a.H, R); p.TX = (int)cdiv(a.W, 30);
  const int64_t nt = (int64_t)N * p.TY * p.TX;
  p.ntiles = (int)nt;
  p.dbg = getenv("XDET_BNECK_DEBUG") ? atoi(getenv("XDET_BNECK_DEBUG")) : 0;
  auto kern = resnet_bneck_kernel<64, R, 256>;
  constexpr int lds = BneckGeom<64, R, 256>::LDS_BYTES;
  static DeviceOnce once;
  XDET_TRY(ensure_dynamic_lds(once, reinterpret_cast<const void*>(kern), lds));
  const dim3 g((unsigned)std::min<int64_t>(256, cdiv(nt, 8) * 8));
  hipLaunchKernelGGL(kern, g, dim3(512), lds, s, p);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

}  // namespace xdet
