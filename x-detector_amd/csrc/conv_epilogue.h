// Shared epilogue of the 16-bit MFMA conv kernels (conv_mfma_dma.hip, conv_mfma_split.hip).
#pragma once
#include "common.h"
#include <type_traits>

namespace xdet {

typedef float ep_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 ep_f16x4 __attribute__((ext_vector_type(4)));

// ReLU that keeps NaN: fmaxf(NaN, 0) = 0 would launder an overflowed split-precision operand (hi = inf -> NaN out
// of the MFMAs) into a clean zero.  IEEE-754-2019 maximum() propagates NaN and is ONE instruction on gfx950
// (v_maximum3_f32 v, v, 0, 0); the compare + select form cost three issue slots per element (VCC hazard in between).
__device__ __forceinline__ float ep_relu(float v) { return __builtin_elementwise_maximum(v, 0.f); }

// The lane id, recomputed (two VALU instructions) at the top of an epilogue: taken from the kernel's `lane` it -- and
// whatever the compiler derives from it ahead of time -- would have to stay alive across the K loop, which in the
// 256 x 256 kernels has no register to spare (the compiler spilled one there).
__device__ __forceinline__ int ep_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// The MFMA accumulator layout gives each lane one
// column and 16 scattered rows, i.e. 4-byte global stores (and residual loads).  Bounce each 32-row slab
// of the wave tile through the (now idle) operand LDS so a lane owns 4 consecutive channels of a row:
// 16-B coalesced residual loads and stores, 4x fewer memory instructions.
template <int WM, int WN, int TM, int TN, int NW, int LDS_BYTES>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, ep_f32x16 (&acc)[TM][TN], unsigned short* smem16, int wave,
                                              int /*lane*/, int wm, int wn, int m0, int n0) {
  const int lane = ep_lane();
  const int frow = lane & 31;
  const int fh = lane >> 5;
  constexpr int EP_LD = WN + 4;                  // floats per staged row (+4: keeps float4 rows 16-B aligned)
  constexpr int C4N = WN / 4;                    // float4 columns per row
  static_assert(NW * 32 * EP_LD * 4 <= LDS_BYTES, "epilogue staging must fit the operand LDS");
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
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.out_hi && p.pl_scale && col_ok) {
    psc = *reinterpret_cast<const float4*>(p.pl_scale + co4);
    psh = *reinterpret_cast<const float4*>(p.pl_shift + co4);
  }
  constexpr int NQ = (32 * C4N) / 64;            // float4 rows a lane handles per 32-row slab
  // residual rows of slab i are requested before slab i-1 is stored, so their HBM latency hides
  // behind the LDS bounce instead of being paid once per slab (one workgroup per CU: nothing else
  // would cover it)
  float4 rr[NQ];
  auto load_res = [&](int i) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int m = m0 + wm * WM + i * 32 + (q * 64 + lane) / C4N;
      rr[q] = (col_ok && m < p.M) ? *reinterpret_cast<const float4*>(p.res + (size_t)m * p.ldr + co4)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (p.res) load_res(0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ep[((r & 3) + 8 * (r >> 2) + 4 * fh) * EP_LD + j * 32 + frow] = acc[i][j][r];
    float4 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(ep + ((q * 64 + lane) / C4N) * EP_LD + c4 * 4);
      v[q] = make_float4(fmaf(a.x, sc4.x, sh4.x), fmaf(a.y, sc4.y, sh4.y), fmaf(a.z, sc4.z, sh4.z),
                         fmaf(a.w, sc4.w, sh4.w));
      if (p.res) { v[q].x += rr[q].x; v[q].y += rr[q].y; v[q].z += rr[q].z; v[q].w += rr[q].w; }
      if (p.relu_out) {
        v[q].x = ep_relu(v[q].x); v[q].y = ep_relu(v[q].y); v[q].z = ep_relu(v[q].z); v[q].w = ep_relu(v[q].w);
      }
    }
    if (p.res && i + 1 < TM) load_res(i + 1);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int m = m0 + wm * WM + i * 32 + (q * 64 + lane) / C4N;
      if (col_ok && m < p.M) {
        if (p.out) *reinterpret_cast<float4*>(p.out + (size_t)m * p.ldo + co4) = v[q];
        if (p.out_hi) {   // second copy as split planes for a consumer on the LDS-DMA path
          float4 t = v[q];
          if (p.pl_scale) {
            t.x = fmaf(t.x, psc.x, psh.x); t.y = fmaf(t.y, psc.y, psh.y);
            t.z = fmaf(t.z, psc.z, psh.z); t.w = fmaf(t.w, psc.w, psh.w);
          }
          if (p.planes_relu) { t.x = ep_relu(t.x); t.y = ep_relu(t.y); t.z = ep_relu(t.z); t.w = ep_relu(t.w); }
          const _Float16 h0 = (_Float16)t.x, h1 = (_Float16)t.y, h2 = (_Float16)t.z, h3 = (_Float16)t.w;
          ep_f16x4 hv = {h0, h1, h2, h3};
          ep_f16x4 lv = {(_Float16)(t.x - (float)h0), (_Float16)(t.y - (float)h1), (_Float16)(t.z - (float)h2),
                      (_Float16)(t.w - (float)h3)};
          const size_t o = ((((size_t)m >> 4) * (size_t)(p.pl_c32 ? p.pl_c32 : p.ldo >> 5) + (size_t)(co4 >> 5)) << 9) + ((m & 15) << 5) + (co4 & 31);
          *reinterpret_cast<uint2*>(p.out_hi + o) = *reinterpret_cast<uint2*>(&hv);
          *reinterpret_cast<uint2*>(p.out_lo + o) = *reinterpret_cast<uint2*>(&lv);
        }
      }
    }
  }
}


// The same epilogue for a tile whose rows are ALL below M (every tile but the last ragged one), with the address work
// taken out of the instruction stream: the conv epilogue is issue-bound, not HBM-bound (s_memrealtime stamps inside the
// 256 x 256 kernel, profiles/NOTES_r04.md: ~7 us of a 60 us tile with both waves of a SIMD in it and the matrix pipe
// idle; the stores are acknowledged 0.3 us after the last one is issued), and ~2/3 of its instructions were 64-bit address
// arithmetic, per-row exec masks and branches.  Here every global access is a raw BUFFER access over this wave's rows of
// the tensor: the per-lane offset (row-in-pass, column) is computed once, the (slab, pass) part of the address is a
// scalar (the SGPR offset of the loads and 64-bit stores, one v_add for the 128-bit stores), and a lane whose columns are padding of the N tile carries an out-of-range offset -- the
// hardware drops its stores and returns zeros for its loads -- so there is no predicate and no branch.  Same values
// through the same arithmetic as conv_epilogue: bit-identical.
typedef float ep_f4 __attribute__((ext_vector_type(4)));
typedef unsigned ep_u2 __attribute__((ext_vector_type(2)));
typedef unsigned ep_u4 __attribute__((ext_vector_type(4)));

template <int WM, int WN, int TM, int TN, int NW, int LDS_BYTES>
__device__ __forceinline__ void conv_epilogue_full(const ConvParams& p, ep_f32x16 (&acc)[TM][TN], unsigned short* smem16, int wave,
                                                   int /*lane*/, int wm, int wn, int m0, int n0) {
  const int lane = ep_lane();
  const int frow = lane & 31;
  const int fh = lane >> 5;
  constexpr int EP_LD = WN + 4;
  constexpr int C4N = WN / 4;
  constexpr int RPQ = 64 / C4N;                  // rows one float4 pass of the wave covers
  constexpr int NQ = 32 / RPQ;                   // passes per 32-row slab
  static_assert(NW * 32 * EP_LD * 4 <= LDS_BYTES, "epilogue staging must fit the operand LDS");
  static_assert(RPQ <= 16 && 16 % RPQ == 0 && WM % 32 == 0, "a pass stays inside one 16-pixel group of the planes");
  __syncthreads();                               // every wave is done reading its operands
  float* ep = reinterpret_cast<float*>(smem16) + wave * (32 * EP_LD);
  const int c4 = lane % C4N;
  const int r0 = lane / C4N;
  const int co4 = n0 + wn * WN + c4 * 4;
  const bool col_ok = co4 < p.ldo;
  const int row0 = m0 + wm * WM;                 // first row of this wave's tile (wave-uniform)
  constexpr unsigned OOR = 0x80000000u;          // beyond every buffer below, with or without the SGPR offset added
  float4 sc4 = make_float4(0.f, 0.f, 0.f, 0.f), sh4 = sc4;
  if (col_ok) {
    sc4 = *reinterpret_cast<const float4*>(p.scale + co4);
    sh4 = *reinterpret_cast<const float4*>(p.shift + co4);
  }
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.out_hi && p.pl_scale && col_ok) {
    psc = *reinterpret_cast<const float4*>(p.pl_scale + co4);
    psh = *reinterpret_cast<const float4*>(p.pl_shift + co4);
  }
  // this wave's WM rows of out / res (row stride ldo / ldr floats) and of the planes ([pix/16][ldo/32][16][32] halves)
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.out ? p.out + (size_t)row0 * p.ldo : nullptr), 0, p.out ? WM * p.ldo * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.res ? p.res + (size_t)row0 * p.ldr : nullptr), 0, p.res ? WM * p.ldr * 4 : 0, 0x00020000);
  const int c32 = p.pl_c32 ? p.pl_c32 : p.ldo >> 5;
  const int pl_bytes = p.out_hi ? (WM / 16) * c32 * 1024 : 0;
  const __amdgpu_buffer_rsrc_t r_hi = __builtin_amdgcn_make_buffer_rsrc(
      p.out_hi ? p.out_hi + (((size_t)(row0 >> 4) * c32) << 9) : nullptr, 0, pl_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_lo = __builtin_amdgcn_make_buffer_rsrc(
      p.out_hi ? p.out_lo + (((size_t)(row0 >> 4) * c32) << 9) : nullptr, 0, pl_bytes, 0x00020000);
  const unsigned v_out = col_ok ? (unsigned)(r0 * p.ldo + co4) * 4u : OOR;
  const unsigned v_res = col_ok ? (unsigned)(r0 * p.ldr + co4) * 4u : OOR;
  const unsigned v_pl = col_ok ? (unsigned)((co4 >> 5) * 1024 + (co4 & 31) * 2 + r0 * 64) : OOR;
  // The layer's options are wave-uniform run-time flags; tested inside the passes the compiler turns them into selects
  // (both sides computed for every element: 516 v_cndmask + 256 v_maximum3 in a 256 x 256 tile's epilogue whatever the
  // layer asked for).  One copy of the passes per (residual, ReLU, planes) combination instead, chosen once.
  auto run = [&](auto RES_, auto RELU_, auto PLANES_) {
    constexpr bool RES = decltype(RES_)::value, RELU = decltype(RELU_)::value, PLANES = decltype(PLANES_)::value;
    ep_f4 rr[NQ];
    auto load_res = [&](int i) {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        rr[q] = __builtin_bit_cast(ep_f4, __builtin_amdgcn_raw_buffer_load_b128(r_res, (int)v_res, (i * 32 + q * RPQ) * p.ldr * 4, 0));
    };
    if (RES) load_res(0);
    const bool pl_affine = PLANES && p.pl_scale != nullptr, pl_relu = PLANES && p.planes_relu != 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ep[((r & 3) + 8 * (r >> 2) + 4 * fh) * EP_LD + j * 32 + frow] = acc[i][j][r];
      // passes in groups of HQ: with all NQ results and NQ residual rows in flight next to the 128 accumulator registers
      // the planes + residual variants of the 256 x 256 kernels spilled
      constexpr int HQ = NQ > 4 ? 4 : NQ;
#pragma unroll
      for (int g = 0; g < NQ; g += HQ) {
        float4 v[HQ];
#pragma unroll
        for (int h = 0; h < HQ; ++h) {
          const int q = g + h;
          const float4 a = *reinterpret_cast<const float4*>(ep + (q * RPQ + r0) * EP_LD + c4 * 4);
          v[h] = make_float4(fmaf(a.x, sc4.x, sh4.x), fmaf(a.y, sc4.y, sh4.y), fmaf(a.z, sc4.z, sh4.z),
                             fmaf(a.w, sc4.w, sh4.w));
          if (RES) { v[h].x += rr[q][0]; v[h].y += rr[q][1]; v[h].z += rr[q][2]; v[h].w += rr[q][3]; }
          if (RELU) {
            v[h].x = ep_relu(v[h].x); v[h].y = ep_relu(v[h].y); v[h].z = ep_relu(v[h].z); v[h].w = ep_relu(v[h].w);
          }
        }
        if (RES && g + HQ >= NQ && i + 1 < TM) load_res(i + 1);
#pragma unroll
        for (int h = 0; h < HQ; ++h) {
          const int rq = i * 32 + (g + h) * RPQ;   // first row of the pass within the wave's tile (compile-time)
          if (p.out) {
            const ep_f4 o = {v[h].x, v[h].y, v[h].z, v[h].w};
            // (the pass's row offset goes into the VGPR offset, not the SGPR one: with a REGISTER soffset the compiler
            // does not keep the two wait states between a 128-bit store and a VALU write of its data registers -- LLVM
            // models that hazard for immediate soffsets only -- and gfx950 has it either way: the next pass's v_pk_fma
            // landed in .z/.w of the row being stored, tools/diag_epilogue.py)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ep_u4, o), r_out, (int)(v_out + (unsigned)(rq * p.ldo * 4)), 0, 0);
          }
          if (PLANES) {   // second copy as split planes for a consumer on the LDS-DMA path
            float4 t = v[h];
            if (pl_affine) {
              asm volatile("" ::: "memory");     // (a real branch, not four selects per element)
              t.x = fmaf(t.x, psc.x, psh.x); t.y = fmaf(t.y, psc.y, psh.y);
              t.z = fmaf(t.z, psc.z, psh.z); t.w = fmaf(t.w, psc.w, psh.w);
            }
            if (pl_relu) {
              asm volatile("" ::: "memory");
              t.x = ep_relu(t.x); t.y = ep_relu(t.y); t.z = ep_relu(t.z); t.w = ep_relu(t.w);
            }
            const _Float16 h0 = (_Float16)t.x, h1 = (_Float16)t.y, h2 = (_Float16)t.z, h3 = (_Float16)t.w;
            ep_f16x4 hv = {h0, h1, h2, h3};
            ep_f16x4 lv = {(_Float16)(t.x - (float)h0), (_Float16)(t.y - (float)h1), (_Float16)(t.z - (float)h2),
                           (_Float16)(t.w - (float)h3)};
            const int so = (rq >> 4) * c32 * 1024 + (rq & 15) * 64;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ep_u2, hv), r_hi, (int)v_pl, so, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ep_u2, lv), r_lo, (int)v_pl, so, 0);
          }
        }
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;
  const int combo = (p.res ? 4 : 0) | (p.relu_out ? 2 : 0) | (p.out_hi ? 1 : 0);
  switch (combo) {
    case 0: run(F{}, F{}, F{}); break;
    case 1: run(F{}, F{}, T{}); break;
    case 2: run(F{}, T{}, F{}); break;
    case 3: run(F{}, T{}, T{}); break;
    case 4: run(T{}, F{}, F{}); break;
    case 5: run(T{}, F{}, T{}); break;
    case 6: run(T{}, T{}, F{}); break;
    default: run(T{}, T{}, T{}); break;
  }
}

}  // namespace xdet
