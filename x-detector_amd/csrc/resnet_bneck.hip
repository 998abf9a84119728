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
  const float* x;                          // the block input, NHWC f32, channel stride COUT (= Cin): conv1x1_a's operand is made
                                           // from it on the CU, and it is the identity shortcut
  const float* pre_sc; const float* pre_sh;   // bn_a folded: the pre-activation is relu(x * pre_sc + pre_sh), [COUT]
  const u16* wa_hi; const u16* wa_lo;      // K-blocked [COUT/32][CMID][32]
  const u16* wb_hi; const u16* wb_lo;      // K-blocked [9 * CMID/32][CMID][32], K block = tap * (CMID/32) + chunk
  const u16* wc_hi; const u16* wc_lo;      // K-blocked [CMID/32][COUT][32]
  const float* sc_a; const float* sh_a;    // folded bn_b (and conv1x1_a's weight pre-scale), [CMID]
  const float* sc_b; const float* sh_b;    // folded bn_c, [CMID]
  const float* sc_c; const float* sh_c;    // conv1x1_c's weight pre-scale / zero shift, [COUT]
  const float* pl_sc; const float* pl_sh;  // bn_next folded for the planes copy, [COUT] (NULL: no planes)
  float* out;                              // NHWC f32, channel stride COUT
  u16* out_hi; u16* out_lo;                // relu(out * pl_sc + pl_sh) as planes [pix/16][COUT/32][16][32] (NULL: not written)
  int N, H, W;
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

template <int V>
struct bk_int { static constexpr int value = V; };
template <int N, typename F, int I = 0>
__device__ __forceinline__ void bk_static_for(F&& f) {
  if constexpr (I < N) {
    f(bk_int<I>{});
    bk_static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

template <int CMID, int R, int COUT>
struct BneckGeom {
  static constexpr int NB = CMID / 32;             // 32-column blocks of the two inner convs
  static constexpr int PR = R + 2;                 // patch rows
  static constexpr int M1 = PR * 32;               // GEMM rows of phase 1
  static constexpr int NBLK1 = PR * NB;            // accumulator blocks of phase 1 (8 waves: one or two each)
  static constexpr int NK1 = COUT / 32;            // K steps of phase 1 (Cin = COUT: identity blocks)
  static constexpr int NCC2 = CMID / 32, NK2 = 9 * NCC2;
  static constexpr int TOTAL = NK1 + NK2;          // steps of a tile's weight stream
  static constexpr int NB3 = COUT / 256;           // column blocks per wave in phase 3
  static constexpr int A_PLANE = M1 * 64, A_STAGE = 2 * A_PLANE;
  static constexpr int B_PLANE = CMID * 64, B_STAGE = 2 * B_PLANE;
  static constexpr int MP = CMID * 4 + 16;         // bytes per mid row: hi | lo | pad (pitch = 4 banks mod 64: 16 consecutive rows
                                                   // of a ds_read_b128 lane group hit 16 different 4-bank windows)
  static constexpr int OFF_A = 0;                  // two A tiles (phase 1); mid2 (phases 2 -> 3) lies over them
  static constexpr int NRING = 6;                  // weight stages: phase 1 runs three steps ahead, phase 2 one triple of taps
  static constexpr int OFF_B = 2 * A_STAGE;
  static constexpr int OFF_M1 = OFF_B + NRING * B_STAGE;
  static constexpr int OFF_T = OFF_M1 + (M1 + 4) * MP;   // (+4 rows: the idle columns 30, 31 read up to 3 pixels past the patch)
  static constexpr int T_FLOATS = 4 * CMID + 6 * COUT;
  static constexpr int OFF_STAMP = OFF_T + T_FLOATS * 4;      // XDET_BNECK_DEBUG=9: s_memtime stamps of workgroup 0 (64 x 8 bytes)
  static constexpr int LDS_BYTES = OFF_STAMP + 512;
  static constexpr int XQ = M1 * 8 / 512;          // 4-channel items of a K step's A tile per lane
  static constexpr int BPW = CMID / 64;            // weight pieces per wave and K step
  static_assert(M1 * 8 % 512 == 0 && CMID % 64 == 0 && COUT % 256 == 0, "piece counts");
  static_assert(R * 32 * MP <= 2 * A_STAGE, "mid2 lies over the A tiles");
  static_assert(R * NB == 8, "phase 2: one accumulator block per wave");
  static_assert(NBLK1 <= 16, "phase 1: at most two accumulator blocks per wave");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  // The VMEM queue of a wave retires in order.  Step s of a tile (after its barrier) issues the x loads of step s + 3
  // (XQ instructions, phase 1 only) and then the weight piece(s) of stream step s + 3; the prologue issues groups -3, -2, -1.
  static constexpr int nx(int s) { return s + 3 < NK1 ? XQ : 0; }
  static constexpr int nb(int s) { return s + 3 < TOTAL ? BPW : 0; }
  // instructions younger than the weights of step s at the top of step s / than the x loads of step s + 1 inside step s
  static constexpr int younger_b(int s) { return nx(s - 2) + nb(s - 2) + nx(s - 1) + nb(s - 1); }
  static constexpr int younger_x(int s) { return nb(s - 2) + nx(s - 1) + nb(s - 1) + nx(s) + nb(s); }
};

template <int CMID, int R, int COUT>
__global__ __launch_bounds__(512) void resnet_bneck_kernel(BneckParams p) {
  using G = BneckGeom<CMID, R, COUT>;
  constexpr int NB = G::NB, NBLK1 = G::NBLK1, NK1 = G::NK1, NCC2 = G::NCC2, NB3 = G::NB3, XQ = G::XQ;
  constexpr int A_PLANE = G::A_PLANE, A_STAGE = G::A_STAGE, B_PLANE = G::B_PLANE, B_STAGE = G::B_STAGE, MP = G::MP;
  constexpr int BPW = G::BPW;
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
  const unsigned lds0 = bk_lds_addr(smem);

  // ---- tables: the folded BNs of the pre-activation and of the three epilogues ----
  {
    float* T = reinterpret_cast<float*>(smem + G::OFF_T);
    for (int i = tid; i < CMID; i += 512) {
      T[i] = p.sc_a[i]; T[CMID + i] = p.sh_a[i]; T[2 * CMID + i] = p.sc_b[i]; T[3 * CMID + i] = p.sh_b[i];
    }
    for (int i = tid; i < COUT; i += 512) {
      T[4 * CMID + i] = p.sc_c[i]; T[4 * CMID + COUT + i] = p.sh_c[i];
      T[4 * CMID + 2 * COUT + i] = p.pl_sc ? p.pl_sc[i] : 1.f;
      T[4 * CMID + 3 * COUT + i] = p.pl_sh ? p.pl_sh[i] : 0.f;
      T[4 * CMID + 4 * COUT + i] = p.pre_sc[i];
      T[4 * CMID + 5 * COUT + i] = p.pre_sh[i];
    }
  }
  const unsigned t_a = lds0 + G::OFF_T, t_b = t_a + 2 * CMID * 4, t_c = t_a + 4 * CMID * 4, t_p = t_c + 4 * COUT * 4;

  // ---- buffer resources ----
  const size_t npix = (size_t)p.N * p.H * p.W;
  const __amdgpu_buffer_rsrc_t r_wa = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>((wave & 1) ? p.wa_lo : p.wa_hi), 0, NK1 * CMID * 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_wb = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>((wave & 1) ? p.wb_lo : p.wb_hi), 0, G::NK2 * CMID * 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)(unsigned)(npix * COUT * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(unsigned)(npix * COUT * 4), 0x00020000);
  const unsigned pl_bytes = p.out_hi ? (unsigned)((((npix + 15) >> 4) * (COUT / 32)) << 10) : 0u;
  const __amdgpu_buffer_rsrc_t r_ohi = __builtin_amdgcn_make_buffer_rsrc(p.out_hi, 0, (int)pl_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_olo = __builtin_amdgcn_make_buffer_rsrc(p.out_lo, 0, (int)pl_bytes, 0x00020000);

  // ---- weight pieces of this wave (LDS DMA): plane = wave & 1 (hi / lo), 16-row groups (wave >> 1) + 4 jj of a K block's
  //      CMID rows; a lane fetches the 16-byte chunk pos ^ ((row >> 2) & 3) of its row (conflict-free fragment reads) ----
  unsigned b_vo[BPW];
#pragma unroll
  for (int jj = 0; jj < BPW; ++jj) {
    const int row = ((wave >> 1) + 4 * jj) * 16 + lr;
    b_vo[jj] = (unsigned)(row * 64 + ((pos ^ ((row >> 2) & 3)) << 4));
  }
  // weight stream of a tile: steps 0 .. NK1-1 = K blocks of W_a, then NK2 steps of W_b in (chunk, tap) order
  auto issue_b = [&](int j) {
    const int slot = j % G::NRING;
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

  // ---- phase 1's A operand is made on the CU: x (f32) -> bn_a -> ReLU -> hi / lo (the arithmetic of the planes copy a conv
  //      epilogue writes, conv_epilogue.h).  Item q of a lane: patch pixel rt = (tid >> 3) + 64 q, channels 4 g .. 4 g + 3 of the
  //      K step's 32 (g = tid & 7): a 16-byte load, 8 + 8 bytes into the A tile ----
  const int g4 = tid & 7;
  unsigned x_vo[XQ], aw_off[XQ];
#pragma unroll
  for (int q = 0; q < XQ; ++q) {
    const int rt = (tid >> 3) + 64 * q;
    aw_off[q] = (unsigned)(rt * 64 + (((g4 >> 1) ^ ((rt >> 2) & 3)) << 4) + (g4 & 1) * 8);
  }
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
    int tid_t = tid;
    asm volatile("" : "+v"(tid_t));                // (recomputed per tile, not kept across it: see frow_t below)
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int rt = (tid_t >> 3) + 64 * q;
      const int y = y0 - 1 + (rt >> 5), x = x0 - 1 + (rt & 31);
      const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const unsigned pix = (unsigned)((c.n * p.H + y) * p.W + x);
      x_vo[q] = ok ? (pix * COUT + (unsigned)(g4 * 4)) * 4u : 0xffffffffu;     // (outside: any finite value -- epilogue 1 zeroes those pixels)
    }
  };
  bk_f32x4 xr[3][XQ];                              // the loads of three K steps in flight
  auto load_x = [&](int kt, auto SET) {
    constexpr int set = decltype(SET)::value;
#pragma unroll
    for (int q = 0; q < XQ; ++q)
      xr[set][q] = __builtin_bit_cast(bk_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)x_vo[q], kt * 128, 0));
  };
  auto transform = [&](int kt, auto SET) {         // x of step kt (in registers) -> A tile kt & 1
    constexpr int set = decltype(SET)::value;
    bk_f32x4 sc = bk_ds_read_f4<0>(t_p + (kt * 32 + g4 * 4) * 4), sh = bk_ds_read_f4<COUT * 4>(t_p + (kt * 32 + g4 * 4) * 4);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc), "+v"(sh)::"memory");
    const unsigned base = lds0 + G::OFF_A + (kt & 1) * A_STAGE;
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      float t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = bk_relu(fmaf(xr[set][q][k], sc[k], sh[k]));
      bk_u2 h, l;
      bk_split4(t, &h, &l);
      bk_ds_write_b64<0>(base + aw_off[q], h);
      bk_ds_write_b64<A_PLANE>(base + aw_off[q], l);
    }
  };

  // ---- conv1x1_c's weights of this wave's output channels: straight from L2 into registers, requested at the start of a
  //      tile's phase 2 (a phase later they are there; held across the whole kernel they cost the other phases 32 registers
  //      and the compiler spilled -- every scratch reload waits with vmcnt(0), i.e. for the whole weight stream) ----
  bk_f16x8 wch[NB3][NCC2][2], wcl[NB3][NCC2][2];
  // (raw buffer loads: one 32-bit per-lane offset, the K block / half in the scalar offset -- as plain pointer loads the eight
  //  64-bit addresses were loop invariants the compiler kept across the whole tile loop, and spilled)
  const __amdgpu_buffer_rsrc_t r_wch = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.wc_hi), 0, NCC2 * COUT * 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_wcl = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.wc_lo), 0, NCC2 * COUT * 64, 0x00020000);
  auto load_wc = [&](int frow_l, int fh_l) {
    const int vo = frow_l * 64 + fh_l * 16;
#pragma unroll
    for (int j = 0; j < NB3; ++j)
#pragma unroll
      for (int kb = 0; kb < NCC2; ++kb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int so = (kb * COUT + (wave * NB3 + j) * 32) * 64 + ks * 32;
          wch[j][kb][ks] = __builtin_bit_cast(bk_f16x8, __builtin_amdgcn_raw_buffer_load_b128(r_wch, vo, so, 0));
          wcl[j][kb][ks] = __builtin_bit_cast(bk_f16x8, __builtin_amdgcn_raw_buffer_load_b128(r_wcl, vo, so, 0));
        }
  };

  // ---- fragment addresses ----
  // phase 1: blocks b = wave (and wave + 8 if it exists): (mi, nj) = (b / NB, b % NB)
  const int nj1 = wave % NB;
  const int mi1[2] = {wave / NB, (wave + 8) / NB};
  const bool two1 = wave + 8 < NBLK1;
  unsigned a1_off[2][2], b_off[2];             // [block][ks], [ks]: byte offsets inside an A tile / a weight stage (hi plane)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int c = ks * 2 + fh;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int rt = mi1[b] * 32 + frow;
      a1_off[b][ks] = (unsigned)(rt * 64 + ((c ^ ((rt >> 2) & 3)) << 4));
    }
  }
  // phases 1 and 2 read the same weight rows: phase 2's block of this wave is (mi2, nj2), nj2 == nj1
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
  int n_stamp = 0;
  auto stamp = [&]() {
    if (p.dbg == 9 && blockIdx.x == 0 && tid == 0 && n_stamp < 64)
      reinterpret_cast<unsigned long long*>(smem + G::OFF_STAMP)[n_stamp++] = __builtin_amdgcn_s_memtime();
  };
  auto prologue = [&]() {                          // groups -3, -2, -1 of the tile's VMEM queue
    load_x(0, bk_int<0>{}); issue_b(0);
    load_x(1, bk_int<1>{}); issue_b(1);
    load_x(2, bk_int<2>{}); issue_b(2);
  };
  prologue();

  for (int t = t_begin; t < t_end; t += GW) {
    const int y0 = cur.ty * R, x0 = cur.tx * 30;
    // (opaque per-tile copies: everything the epilogues derive from the lane id is then recomputed inside the tile loop --
    //  hoisted out of it, those ~40 addresses and predicates lived across all phases and were spilled)
    int frow_t = frow, fh_t = fh;
    asm volatile("" : "+v"(frow_t), "+v"(fh_t));
    unsigned o_vo[R];                              // byte offset of (output pixel, this wave's first channel + 4 fh) in x / out
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int y = y0 + i, x = x0 + frow_t;
      const bool ok = y < p.H && frow_t < 30 && x < p.W;
      const unsigned pix = (unsigned)((cur.n * p.H + y) * p.W + x);
      o_vo[i] = ok ? (pix * COUT + (unsigned)(wave * NB3 * 32 + 4 * fh_t)) * 4u : 0x80000000u;   // (beyond the tensor, with or without the scalar offset)
    }
    // The shortcut rows of this wave's 32 output channels are the SAME cache lines as the x loads of K step `wave` (channel
    // block = wave): requested in that step, right behind them, they come out of L2 -- requested a phase later they came
    // over the fabric a second time (~10 B/clk per CU: 13,000 clocks per tile).  They sit in the in-order VMEM queue at the
    // end of step `wave`'s group: the counted waits of the following three steps see NRES more entries.
    constexpr int NRES = R * 4;
    static_assert(NB3 == 1, "one channel block per wave (the shortcut loads ride on one K step)");
    bk_f32x4 res[R][NB3][4];
    // =============================== phase 1: conv1x1_a on the halo patch ===============================
    // (the prologue's loads are older than the previous tile's epilogue stores: everything has landed after this wait, and
    //  the counted waits of the steps below are upper bounds)
    stamp();                                       // 0: tile start
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();                                       // 1: everything of the prologue (and the last epilogue) has landed
    transform(0, bk_int<0>{});
    bk_f32x16 acc1[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[b][r] = 0.f;
    bk_static_for<NK1>([&](auto KT) {
      constexpr int kt = decltype(KT)::value;
      // A tile kt & 1 is complete (own writes retired, then the barrier), W(kt) has landed, every wave is done with step kt - 1
      if ((unsigned)(kt - wave - 1) < 3u) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(G::younger_b(kt) + NRES) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(G::younger_b(kt)) : "memory");
      if constexpr (kt + 3 < NK1) load_x(kt + 3, bk_int<kt % 3>{});
      issue_b(kt + 3);
      if (kt == wave) {
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            res[i][0][q] = __builtin_bit_cast(bk_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)o_vo[i], 8 * q * 4, 0));
      }
      const unsigned sa = lds0 + G::OFF_A + (kt & 1) * A_STAGE, sb = lds0 + G::OFF_B + (kt % G::NRING) * B_STAGE;
      bk_f16x8 bh[2], bl[2], ah0[2], al0[2], ah1[2], al1[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bh[ks] = bk_ds_read_h8<0>(sb + b_off[ks]);
        bl[ks] = bk_ds_read_h8<B_PLANE>(sb + b_off[ks]);
        ah0[ks] = bk_ds_read_h8<0>(sa + a1_off[0][ks]);
        al0[ks] = bk_ds_read_h8<A_PLANE>(sa + a1_off[0][ks]);
        ah1[ks] = ah0[ks]; al1[ks] = al0[ks];
        if (two1) {
          ah1[ks] = bk_ds_read_h8<0>(sa + a1_off[1][ks]);
          al1[ks] = bk_ds_read_h8<A_PLANE>(sa + a1_off[1][ks]);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[ks]), "+v"(bl[ks]), "+v"(ah0[ks]), "+v"(al0[ks]), "+v"(ah1[ks]), "+v"(al1[ks])::"memory");
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ks], al0[ks], acc1[0], 0, 0, 0);
        if (two1) acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ks], al1[ks], acc1[1], 0, 0, 0);
        acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ks], ah0[ks], acc1[0], 0, 0, 0);
        if (two1) acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ks], ah1[ks], acc1[1], 0, 0, 0);
        acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ks], ah0[ks], acc1[0], 0, 0, 0);
        if (two1) acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ks], ah1[ks], acc1[1], 0, 0, 0);
      }
      // the next step's A tile, under this step's MFMAs
      if constexpr (kt + 1 < NK1) {
        if ((unsigned)(kt - wave) < 3u) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::younger_x(kt) + NRES) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::younger_x(kt)) : "memory");
        transform(kt + 1, bk_int<(kt + 1) % 3>{});
      }
    });
    stamp();                                       // 2: phase 1 done
    // ---- epilogue 1: bn_b + ReLU, zero outside the image, split -> mid1 ----
    {
      bk_f32x4 sc[4], sh[4];                       // this lane's 16 channels of the block column (the same for both blocks)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = nj1 * 32 + 8 * q + 4 * fh_t;
        sc[q] = bk_ds_read_f4<0>(t_a + c * 4);
        sh[q] = bk_ds_read_f4<CMID * 4>(t_a + c * 4);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc[q]), "+v"(sh[q])::"memory");   // (tied: the FMAs stay below)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (b == 1 && !two1) break;
        const int y = y0 - 1 + mi1[b], x = x0 - 1 + frow_t;
        const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const unsigned row = m1_base + (unsigned)((mi1[b] * 32 + frow_t) * MP + (nj1 * 32 + 4 * fh_t) * 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[k] = bk_relu(fmaf(acc1[b][4 * q + k], sc[q][k], sh[q][k]));
            v[k] = ok ? v[k] : 0.f;
          }
          bk_u2 h, l;
          bk_split4(v, &h, &l);
          if (q == 0) { bk_ds_write_b64<0>(row, h); bk_ds_write_b64<CMID * 2>(row, l); }
          if (q == 1) { bk_ds_write_b64<16>(row, h); bk_ds_write_b64<CMID * 2 + 16>(row, l); }
          if (q == 2) { bk_ds_write_b64<32>(row, h); bk_ds_write_b64<CMID * 2 + 32>(row, l); }
          if (q == 3) { bk_ds_write_b64<48>(row, h); bk_ds_write_b64<CMID * 2 + 48>(row, l); }
        }
      }
    }
    stamp();                                       // 3: epilogue 1 issued
    // ---- conv1x1_c's weights: requested now, used a phase later.  They enter the in-order VMEM queue behind W(NK1 + 2) and in
    //      front of W(NK1 + 3): the first three steps of phase 2 count them as younger ----
    load_wc(frow_t, fh_t);
    // =============================== phase 2: conv3x3_b out of mid1 ===============================
    bk_f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    // One barrier per TRIPLE of taps (a row of the 3 x 3 window of one 32-channel chunk): its three weight stages were requested
    // a triple earlier; the next triple's are requested right behind the barrier into the stages the previous one used.
    constexpr int NLATE = NB3 * NCC2 * 4;          // the loads requested in front of this phase (conv1x1_c's weights)
    static_assert(3 * BPW + NLATE + NRES < 64, "vmcnt is a 6-bit counter");
    static_assert(G::NK2 % 3 == 0 && G::NRING == 6, "two triples of weight stages");
#pragma unroll 1
    for (int T = 0; T < G::NK2 / 3; ++T) {
      const int cc = T / 3, ky = T - cc * 3;
      const int j0 = NK1 + 3 * T;
      // first triple: younger than its stages are conv1x1_c's weights and, in the wave whose K step was phase 1's last, the
      // shortcut rows; later triples: everything older has to be there anyway
      if (T == 0) {
        if (wave == NK1 - 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NLATE + NRES) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NLATE) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      if (T + 1 < G::NK2 / 3) { issue_b(j0 + 3); issue_b(j0 + 4); issue_b(j0 + 5); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const unsigned sb = lds0 + G::OFF_B + (unsigned)((j0 + kx) % G::NRING) * B_STAGE;
        const unsigned ar = m1_base + (unsigned)(((mi2 + ky) * 32 + frow_t + kx) * MP + cc * 64 + fh_t * 16);
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
    stamp();                                       // 4: phase 2 done
    // ---- epilogue 2: bn_c + ReLU, split -> mid2 (over the A tiles: nobody reads them any more) ----
    {
      bk_f32x4 sc[4], sh[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = nj2 * 32 + 8 * q + 4 * fh_t;
        sc[q] = bk_ds_read_f4<0>(t_b + c * 4);
        sh[q] = bk_ds_read_f4<CMID * 4>(t_b + c * 4);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc[q]), "+v"(sh[q])::"memory");
      const unsigned row = m2_base + (unsigned)((mi2 * 32 + frow_t) * MP + (nj2 * 32 + 4 * fh_t) * 2);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = bk_relu(fmaf(acc2[4 * q + k], sc[q][k], sh[q][k]));
        bk_u2 h, l;
        bk_split4(v, &h, &l);
        if (q == 0) { bk_ds_write_b64<0>(row, h); bk_ds_write_b64<CMID * 2>(row, l); }
        if (q == 1) { bk_ds_write_b64<16>(row, h); bk_ds_write_b64<CMID * 2 + 16>(row, l); }
        if (q == 2) { bk_ds_write_b64<32>(row, h); bk_ds_write_b64<CMID * 2 + 32>(row, l); }
        if (q == 3) { bk_ds_write_b64<48>(row, h); bk_ds_write_b64<CMID * 2 + 48>(row, l); }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // mid2 complete
    stamp();                                       // 5: mid2 complete
    // =============================== phase 3: conv1x1_c out of mid2, in two halves of the tile's rows ===============================
    // (a half's accumulators, the weights and ALL shortcut rows are live together: 32 + 32 + 64 registers; with the whole
    //  tile's 64 accumulators next to them the compiler spilled)
    constexpr int RH = R / 2;
    static_assert(R % 2 == 0, "two halves");
    unsigned p_vo[R];                              // planes: byte offset of (pixel, this wave's first channel block, 4 fh)
    if (p.out_hi) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int y = y0 + i, x = x0 + frow_t;
        const bool ok = y < p.H && frow_t < 30 && x < p.W;
        const unsigned pix = (unsigned)((cur.n * p.H + y) * p.W + x);
        p_vo[i] = ok ? ((((pix >> 4) * (COUT / 32) + (unsigned)(wave * NB3)) << 10) + ((pix & 15) << 6) + (unsigned)(8 * fh_t)) : 0x80000000u;
      }
    }
    auto half = [&](auto HALF) {
      constexpr int i0 = decltype(HALF)::value * RH;
      bk_f32x16 acc3[RH][NB3];
#pragma unroll
      for (int i = 0; i < RH; ++i)
#pragma unroll
        for (int j = 0; j < NB3; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc3[i][j][r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < NCC2; ++kb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bk_f16x8 ah[RH], al[RH];
#pragma unroll
          for (int i = 0; i < RH; ++i) {
            const unsigned ar = m2_base + (unsigned)(((i0 + i) * 32 + frow_t) * MP + kb * 64 + (ks * 2 + fh_t) * 16);
            ah[i] = bk_ds_read_h8<0>(ar);
            al[i] = bk_ds_read_h8<CMID * 2>(ar);
          }
#pragma unroll
          for (int i = 0; i < RH; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[i]), "+v"(al[i])::"memory");
#pragma unroll
          for (int i = 0; i < RH; ++i)
#pragma unroll
            for (int j = 0; j < NB3; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wch[j][kb][ks], al[i], acc3[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < RH; ++i)
#pragma unroll
            for (int j = 0; j < NB3; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcl[j][kb][ks], ah[i], acc3[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < RH; ++i)
#pragma unroll
            for (int j = 0; j < NB3; ++j) acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wch[j][kb][ks], ah[i], acc3[i][j], 0, 0, 0);
        }
      if constexpr (decltype(HALF)::value == 1) {
        // every wave is done reading mid2 (the next tile's first A tile is written over it after the epilogue) and the weight
        // ring: the next tile's first loads travel while this half's output is written
        asm volatile("s_barrier" ::: "memory");
        stamp();                                   // 6: phase 3 done
        if (t + GW < t_end) {
          cur = decode(t + GW);
          tile_offsets(cur);
          prologue();
        }
      }
      // ---- epilogue 3: + x, f32 out (and the next block's pre-activation planes, if anyone reads them) ----
#pragma unroll
      for (int j = 0; j < NB3; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = (wave * NB3 + j) * 32 + 8 * q + 4 * fh_t;
          bk_f32x4 sc = bk_ds_read_f4<0>(t_c + c * 4), sh = bk_ds_read_f4<COUT * 4>(t_c + c * 4);
          bk_f32x4 psc = bk_ds_read_f4<2 * COUT * 4>(t_c + c * 4), psh = bk_ds_read_f4<3 * COUT * 4>(t_c + c * 4);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc), "+v"(sh), "+v"(psc), "+v"(psh)::"memory");
#pragma unroll
          for (int i = 0; i < RH; ++i) {
            bk_f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              v[k] = fmaf(acc3[i][j][4 * q + k], sc[k], sh[k]);
              v[k] += res[i0 + i][j][q][k];
            }
            if (p.dbg == 1 || p.dbg == 2) {
              bk_f32x4 z = {0.f, 0.f, 0.f, 0.f};
              if ((wave * NB3 + j) * 32 < CMID) {
                const unsigned base = p.dbg == 1 ? m1_base + (unsigned)(((i0 + i + 1) * 32 + frow_t + 1) * MP) : m2_base + (unsigned)(((i0 + i) * 32 + frow_t) * MP);
                const bk_f16x4 hh = *reinterpret_cast<const bk_f16x4*>(smem + (base - lds0) + c * 2);
                const bk_f16x4 ll = *reinterpret_cast<const bk_f16x4*>(smem + (base - lds0) + CMID * 2 + c * 2);
#pragma unroll
                for (int k = 0; k < 4; ++k) z[k] = (float)hh[k] + (float)ll[k];
              }
              v = z;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bk_u4, v), r_out, (int)o_vo[i0 + i], (j * 32 + 8 * q) * 4, 0);
            if (p.out_hi) {
              float tt[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) tt[k] = bk_relu(fmaf(v[k], psc[k], psh[k]));
              bk_u2 h, l;
              bk_split4(tt, &h, &l);
              __builtin_amdgcn_raw_buffer_store_b64(h, r_ohi, (int)p_vo[i0 + i], j * 1024 + q * 16, 0);
              __builtin_amdgcn_raw_buffer_store_b64(l, r_olo, (int)p_vo[i0 + i], j * 1024 + q * 16, 0);
            }
          }
        }
    };
    half(bk_int<0>{});
    half(bk_int<1>{});
    stamp();                                       // 7: epilogue 3 issued
    if (p.dbg) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (mid2 was read above)
  }
  if (p.dbg == 9 && blockIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (wave == 0) {                               // (the lane id recomputed: `tid` kept alive to here cost a spilled register)
      int l;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
      reinterpret_cast<unsigned long long*>(p.out)[l] = reinterpret_cast<const unsigned long long*>(smem + G::OFF_STAMP)[l];
    }
  }
}

bool resnet_bneck_supported(int cin, int cmid, int cout, int H, int W, int N) {
  const size_t npix = (size_t)N * H * W;
  return cin == cout && cmid == 64 && cout == 256 && npix * cout * 4 < ((size_t)1 << 31);
}

int launch_resnet_bneck(const BneckLaunch& a, int N, hipStream_t s) {
  XDET_REQUIRE(resnet_bneck_supported(a.cin, a.cmid, a.cout, a.H, a.W, N), "resnet_bneck: unsupported block geometry");
  XDET_REQUIRE(a.pre_sc && a.pre_sh && a.x && a.out && a.wa_hi && a.wa_lo && a.wb_hi && a.wb_lo && a.wc_hi && a.wc_lo,
               "resnet_bneck: NULL argument");
  if (N <= 0) return XDET_OK;
  constexpr int R = 4;
  BneckParams p;
  p.x = a.x; p.pre_sc = a.pre_sc; p.pre_sh = a.pre_sh;
  p.wa_hi = a.wa_hi; p.wa_lo = a.wa_lo; p.wb_hi = a.wb_hi; p.wb_lo = a.wb_lo; p.wc_hi = a.wc_hi; p.wc_lo = a.wc_lo;
  p.sc_a = a.sc_a; p.sh_a = a.sh_a; p.sc_b = a.sc_b; p.sh_b = a.sh_b; p.sc_c = a.sc_c; p.sh_c = a.sh_c;
  p.pl_sc = a.out_hi ? a.pl_sc : nullptr; p.pl_sh = a.out_hi ? a.pl_sh : nullptr;
  p.out = a.out; p.out_hi = a.out_hi; p.out_lo = a.out_lo;
  p.N = N; p.H = a.H; p.W = a.W;
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
