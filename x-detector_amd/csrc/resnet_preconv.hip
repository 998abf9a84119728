// The opening 1x1 conv of a ResNet v2 bottleneck block (net/resnet_v2.py:142-166: batch_norm_relu -> conv 1x1 -> batch_norm_relu)
// with the PRE-ACTIVATION MADE ON THE CU:
//
//     y = relu(bn_b( conv1x1( relu(bn_a(x)) ) ))  as split planes,  x = the raw f32 block input
//
// The three-launch form reads relu(bn_a(x)) as split planes that the previous block's closing conv wrote next to its f32
// output -- a second copy of the widest tensor of the stage (118 MB per batch of 8 at 120 x 120 x 256) written and read once
// each for nothing but a BN + ReLU + f16 split.  Here the operand is made where it is consumed, exactly as in phase 1 of
// resnet_bneck.hip: 16-byte loads of x three K steps ahead in registers -> bn_a -> ReLU -> hi / lo (the arithmetic of the
// planes copy a conv epilogue writes, so the bits agree) -> one of two LDS A tiles under the previous step's MFMAs; the
// weights come through a four-stage LDS-DMA ring three steps ahead; one barrier per 32-deep K step.  The producer of x then
// does not write the planes copy at all (ResNetTrunk: planes_optional_next / BneckGroup::next_fused).
//
// Tile = 128 (or 64) consecutive pixels x all CMID output channels (8 waves: 4 row blocks x 2 column halves, or 2 x 4),
// persistent workgroups.
// MFMAs with the operands swapped (weights first): a lane holds one pixel and four consecutive channels per register
// group, the epilogue (bn_b + ReLU, split) stores 8 bytes per plane straight into the [pix/16][C/32][16][32] layout.
// K order, product order and epilogue arithmetic are conv_dma_f16_kernel's: bit-identical (tests/test_gpu_resnet_bneck.py).
#include "common.h"
#include <algorithm>

namespace xdet {

typedef float pc_f32x16 __attribute__((ext_vector_type(16)));
typedef float pc_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pc_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pc_f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned pc_u2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

struct PreconvParams {
  const float* x;                           // [M][CIN] f32
  const float* pre_sc; const float* pre_sh; // bn_a folded, [CIN]
  const u16* w_hi; const u16* w_lo;         // K-blocked [CIN/32][CMID][32]
  const float* sc; const float* sh;         // bn_b folded (and the weight pre-scale), [CMID]
  u16* out_hi; u16* out_lo;                 // planes [M/16][CMID/32][16][32]
  int M, ntiles;
};

template <int OFF>
__device__ __forceinline__ pc_f16x8 pc_ds_read_h8(unsigned addr) {
  pc_f16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int OFF>
__device__ __forceinline__ pc_f32x4 pc_ds_read_f4(unsigned addr) {
  pc_f32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int OFF>
__device__ __forceinline__ void pc_ds_write_b64(unsigned addr, pc_u2 v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ float pc_relu(float v) { return __builtin_elementwise_maximum(v, 0.f); }   // keeps NaN (conv_epilogue.h)
__device__ __forceinline__ void pc_split4(const float (&t)[4], pc_u2* h, pc_u2* l) {
  const _Float16 h0 = (_Float16)t[0], h1 = (_Float16)t[1], h2 = (_Float16)t[2], h3 = (_Float16)t[3];
  const pc_f16x4 hv = {h0, h1, h2, h3};
  const pc_f16x4 lv = {(_Float16)(t[0] - (float)h0), (_Float16)(t[1] - (float)h1), (_Float16)(t[2] - (float)h2),
                       (_Float16)(t[3] - (float)h3)};
  *h = __builtin_bit_cast(pc_u2, hv);
  *l = __builtin_bit_cast(pc_u2, lv);
}
template <int V>
struct pc_int { static constexpr int value = V; };
template <int N, typename F, int I = 0>
__device__ __forceinline__ void pc_static_for(F&& f) {
  if constexpr (I < N) {
    f(pc_int<I>{});
    pc_static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

template <int CMID, int CIN, int BM_>
struct PreconvGeom {
  static constexpr int BM = BM_;                   // pixels per tile (128, or 64 where 128 leaves too few tiles)
  static constexpr int MB = BM / 32, WPM = 8 / MB;  // row blocks; waves per row block
  static constexpr int NK = CIN / 32;
  static constexpr int NBW = (CMID / 32) / WPM;    // 32-column blocks per wave
  static constexpr int A_PLANE = BM * 64, A_STAGE = 2 * A_PLANE;
  static constexpr int B_PLANE = CMID * 64, B_STAGE = 2 * B_PLANE;
  static constexpr int OFF_A = 0, OFF_B = 2 * A_STAGE, OFF_T = OFF_B + 4 * B_STAGE;
  static constexpr int LDS_BYTES = OFF_T + (2 * CIN + 2 * CMID) * 4;
  static constexpr int XQ = BM * 8 / 512;          // 4-channel items of a K step's A tile per lane (2)
  static constexpr int BPW = CMID / 64;            // weight pieces per wave and K step
  static_assert(CMID % 64 == 0 && CIN % 32 == 0 && (BM == 64 || BM == 128) && (CMID / 32) % WPM == 0 && LDS_BYTES <= 160 * 1024, "geometry");
  // in-order VMEM queue of a wave: step s (after its barrier) issues the x loads of step s + 3, then the weight pieces of
  // step s + 3; the prologue issues groups -3, -2, -1 (resnet_bneck.hip, BneckGeom)
  static constexpr int nx(int s) { return s + 3 < NK ? XQ : 0; }
  static constexpr int nb(int s) { return s + 3 < NK ? BPW : 0; }
  static constexpr int younger_b(int s) { return nx(s - 2) + nb(s - 2) + nx(s - 1) + nb(s - 1); }
  static constexpr int younger_x(int s) { return nb(s - 2) + nx(s - 1) + nb(s - 1) + nx(s) + nb(s); }
};

template <int CMID, int CIN, int BM_>
__global__ __launch_bounds__(512) void resnet_preconv_kernel(PreconvParams p) {
  using G = PreconvGeom<CMID, CIN, BM_>;
  constexpr int NK = G::NK, NBW = G::NBW, XQ = G::XQ, BPW = G::BPW, BM = G::BM;
  constexpr int A_PLANE = G::A_PLANE, A_STAGE = G::A_STAGE, B_PLANE = G::B_PLANE, B_STAGE = G::B_STAGE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fh = lane >> 5;
  const int lr = lane >> 2, pos = lane & 3;
  const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, GW = gridDim.x >> 3;
  const int per_xcd = (p.ntiles + 7) >> 3;
  const int t_begin = xcd * per_xcd + wg;
  const int t_end = min(p.ntiles, (xcd + 1) * per_xcd);
  if (t_begin >= t_end) return;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(smem);

  {
    float* T = reinterpret_cast<float*>(smem + G::OFF_T);
    for (int i = tid; i < CIN; i += 512) { T[i] = p.pre_sc[i]; T[CIN + i] = p.pre_sh[i]; }
    for (int i = tid; i < CMID; i += 512) { T[2 * CIN + i] = p.sc[i]; T[2 * CIN + CMID + i] = p.sh[i]; }
  }
  const unsigned t_p = lds0 + G::OFF_T, t_e = t_p + 2 * CIN * 4;

  const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>((wave & 1) ? p.w_lo : p.w_hi), 0, NK * CMID * 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)(unsigned)((size_t)p.M * CIN * 4), 0x00020000);
  const unsigned pl_bytes = (unsigned)((((size_t)p.M + 15) >> 4) * (CMID / 32) << 10);
  const __amdgpu_buffer_rsrc_t r_ohi = __builtin_amdgcn_make_buffer_rsrc(p.out_hi, 0, (int)pl_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_olo = __builtin_amdgcn_make_buffer_rsrc(p.out_lo, 0, (int)pl_bytes, 0x00020000);

  // weight pieces of this wave: plane = wave & 1, 16-row groups (wave >> 1) + 4 jj of a K block's CMID rows
  unsigned b_vo[BPW];
#pragma unroll
  for (int jj = 0; jj < BPW; ++jj) {
    const int row = ((wave >> 1) + 4 * jj) * 16 + lr;
    b_vo[jj] = (unsigned)(row * 64 + ((pos ^ ((row >> 2) & 3)) << 4));
  }
  auto issue_b = [&](int j) {
    unsigned char* dst = smem + G::OFF_B + (j & 3) * B_STAGE + (wave & 1) * B_PLANE;
#pragma unroll
    for (int jj = 0; jj < BPW; ++jj)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)(dst + ((wave >> 1) + 4 * jj) * 1024), 16,
                                               (int)b_vo[jj], j * (CMID * 64), 0, 0);
  };
  // A operand: item q of a lane = tile row (tid >> 3) + 64 q, channels 4 g .. 4 g + 3 of the K step's 32 (g = tid & 7)
  const int g4 = tid & 7;
  unsigned x_vo[XQ], aw_off[XQ];
#pragma unroll
  for (int q = 0; q < XQ; ++q) {
    const int rt = (tid >> 3) + 64 * q;
    aw_off[q] = (unsigned)(rt * 64 + (((g4 >> 1) ^ ((rt >> 2) & 3)) << 4) + (g4 & 1) * 8);
  }
  auto tile_offsets = [&](int m0) {
    int tid_t = tid;
    asm volatile("" : "+v"(tid_t));
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const int m = m0 + (tid_t >> 3) + 64 * q;
      x_vo[q] = m < p.M ? ((unsigned)m * CIN + (unsigned)(g4 * 4)) * 4u : 0xffffffffu;
    }
  };
  pc_f32x4 xr[3][XQ];
  auto load_x = [&](int kt, auto SET) {
    constexpr int set = decltype(SET)::value;
#pragma unroll
    for (int q = 0; q < XQ; ++q)
      xr[set][q] = __builtin_bit_cast(pc_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_x, (int)x_vo[q], kt * 128, 0));
  };
  auto transform = [&](int kt, auto SET) {
    constexpr int set = decltype(SET)::value;
    pc_f32x4 sc = pc_ds_read_f4<0>(t_p + (kt * 32 + g4 * 4) * 4), sh = pc_ds_read_f4<CIN * 4>(t_p + (kt * 32 + g4 * 4) * 4);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc), "+v"(sh)::"memory");
    const unsigned base = lds0 + G::OFF_A + (kt & 1) * A_STAGE;
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      float t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = pc_relu(fmaf(xr[set][q][k], sc[k], sh[k]));
      pc_u2 h, l;
      pc_split4(t, &h, &l);
      pc_ds_write_b64<0>(base + aw_off[q], h);
      pc_ds_write_b64<A_PLANE>(base + aw_off[q], l);
    }
  };
  // fragments: this wave's row block mi = wave / WPM, column blocks (wave % WPM) * NBW + j
  const int mi = wave / G::WPM, nh = wave % G::WPM;
  unsigned a_off[2], b_off[2][NBW];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int c = ks * 2 + fh;
    const int rt = mi * 32 + frow;
    a_off[ks] = (unsigned)(rt * 64 + ((c ^ ((rt >> 2) & 3)) << 4));
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
      const int rb = (nh * NBW + j) * 32 + frow;
      b_off[ks][j] = (unsigned)(rb * 64 + ((c ^ ((rb >> 2) & 3)) << 4));
    }
  }

  tile_offsets(t_begin * BM);
  __syncthreads();
  auto prologue = [&]() {
    load_x(0, pc_int<0>{}); issue_b(0);
    load_x(1, pc_int<1>{}); issue_b(1);
    load_x(2, pc_int<2>{}); issue_b(2);
  };
  prologue();

  for (int t = t_begin; t < t_end; t += GW) {
    const int m0 = t * BM;
    int frow_t = frow, fh_t = fh;
    asm volatile("" : "+v"(frow_t), "+v"(fh_t));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the prologue (older than the last tile's stores) has landed
    transform(0, pc_int<0>{});
    pc_f32x16 acc[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    pc_static_for<NK>([&](auto KT) {
      constexpr int kt = decltype(KT)::value;
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(G::younger_b(kt)) : "memory");
      if constexpr (kt + 3 < NK) {
        load_x(kt + 3, pc_int<kt % 3>{});
        issue_b(kt + 3);
      }
      const unsigned sa = lds0 + G::OFF_A + (kt & 1) * A_STAGE, sb = lds0 + G::OFF_B + (kt & 3) * B_STAGE;
      pc_f16x8 ah[2], al[2], bh[2][NBW], bl[2][NBW];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        ah[ks] = pc_ds_read_h8<0>(sa + a_off[ks]);
        al[ks] = pc_ds_read_h8<A_PLANE>(sa + a_off[ks]);
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
          bh[ks][j] = pc_ds_read_h8<0>(sb + b_off[ks][j]);
          bl[ks][j] = pc_ds_read_h8<B_PLANE>(sb + b_off[ks][j]);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[ks]), "+v"(al[ks])::"memory");
#pragma unroll
        for (int j = 0; j < NBW; ++j) asm volatile("" : "+v"(bh[ks][j]), "+v"(bl[ks][j]));
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ks][j], al[ks], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ks][j], ah[ks], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ks][j], ah[ks], acc[j], 0, 0, 0);
      }
      if constexpr (kt + 1 < NK) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::younger_x(kt)) : "memory");
        transform(kt + 1, pc_int<(kt + 1) % 3>{});
      }
    });
    // every wave is done with the ring and the A tiles: the next tile's first loads travel under this tile's epilogue
    asm volatile("s_barrier" ::: "memory");
    if (t + GW < t_end) {
      tile_offsets((t + GW) * BM);
      prologue();
    }
    // ---- epilogue: bn_b + ReLU, split, 8 bytes per plane and (pixel, 4 channels) ----
    {
      const int m = m0 + mi * 32 + frow_t;
      const unsigned pbase = m < p.M ? ((((unsigned)m >> 4) * (CMID / 32)) << 10) + (((unsigned)m & 15) << 6) + (unsigned)(8 * fh_t) : 0x80000000u;
#pragma unroll
      for (int j = 0; j < NBW; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = (nh * NBW + j) * 32 + 8 * q + 4 * fh_t;
          pc_f32x4 sc = pc_ds_read_f4<0>(t_e + c * 4), sh = pc_ds_read_f4<CMID * 4>(t_e + c * 4);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sc), "+v"(sh)::"memory");
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = pc_relu(fmaf(acc[j][4 * q + k], sc[k], sh[k]));
          pc_u2 h, l;
          pc_split4(v, &h, &l);
          __builtin_amdgcn_raw_buffer_store_b64(h, r_ohi, (int)pbase, ((nh * NBW + j) << 10) + q * 16, 0);
          __builtin_amdgcn_raw_buffer_store_b64(l, r_olo, (int)pbase, ((nh * NBW + j) << 10) + q * 16, 0);
        }
    }
  }
}

// (cin, cmid): stage 2's blocks (256 / 512 -> 128; 128-pixel tiles: 900 / 225 at batch 8).  Stage 3's (512 / 1024 -> 256 as
// 64-pixel tiles: 450 / 113) was built and measured: the 7,200-row layers do not fill the chip either way and every tile
// streams the whole 1 MB filter -- trunk 1.867 -> 1.878 ms, not kept (the template still takes BM = 64).
bool resnet_preconv_supported(int cin, int cmid, int64_t M) {
  return cmid == 128 && (cin == 256 || cin == 512) && (size_t)M * cin * 4 < ((size_t)1 << 31);
}

int launch_resnet_preconv(const float* x, const float* pre_sc, const float* pre_sh, const unsigned short* w_hi,
                          const unsigned short* w_lo, const float* sc, const float* sh, unsigned short* out_hi,
                          unsigned short* out_lo, int64_t M, int cin, int cmid, hipStream_t s) {
  XDET_REQUIRE(resnet_preconv_supported(cin, cmid, M), "resnet_preconv: unsupported channel counts / tensor size");
  XDET_REQUIRE(x && pre_sc && pre_sh && w_hi && w_lo && sc && sh && out_hi && out_lo, "resnet_preconv: NULL argument");
  if (M <= 0) return XDET_OK;
  PreconvParams p;
  p.x = x; p.pre_sc = pre_sc; p.pre_sh = pre_sh; p.w_hi = w_hi; p.w_lo = w_lo; p.sc = sc; p.sh = sh;
  p.out_hi = out_hi; p.out_lo = out_lo;
  p.M = (int)M;
  auto go = [&](auto kern, int lds, int bm) {
    static DeviceOnce once;
    XDET_TRY(ensure_dynamic_lds(once, reinterpret_cast<const void*>(kern), lds));
    p.ntiles = (int)cdiv(M, bm);
    const dim3 g((unsigned)std::min<int64_t>(256, cdiv(p.ntiles, 8) * 8));
    hipLaunchKernelGGL(kern, g, dim3(512), lds, s, p);
    XDET_LAUNCH_CHECK();
    return (int)XDET_OK;
  };
  if (cin == 256) return go(resnet_preconv_kernel<128, 256, 128>, PreconvGeom<128, 256, 128>::LDS_BYTES, 128);
  return go(resnet_preconv_kernel<128, 512, 128>, PreconvGeom<128, 512, 128>::LDS_BYTES, 128);
}

}  // namespace xdet
