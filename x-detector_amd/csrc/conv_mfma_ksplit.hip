// Implicit-GEMM convolution with a FIXED split of the reduction, for the layers whose grids cannot fill the chip:
// few output rows, long K loops (ResNet-50 stages 3-4 at batch 8: 7,200 / 1,800 rows against 72- / 144-step K loops,
// net/resnet_v2.py:142-184; the RPN 3x3 conv and the 2048 -> 25 head GEMM of a single image,
// net/xception_body.py:381-400,540-558).  Same operands, same LDS-DMA staging and the same per-element product order as
// conv_mfma_dma.hip; what is new is how the K steps are dealt out:
//
//   * the K steps of a layer are cut into `ksplit` ranges of equal length -- ksplit is a constant OF THE LAYER (chosen at
//     plan time from its geometry), never of the batch -- and the result is DEFINED as the left fold
//         out = epilogue(((p_0 + p_1) + p_2) + ... + p_{S-1}),      p_r = sum over range r, accumulated from zero,
//     every `+` one f32 addition per element;
//   * PARALLEL mode (small grids): one workgroup per (tile, range) parks its raw accumulators in a scratch slab; a second,
//     light launch (one workgroup per tile) folds the S slabs in range order and runs the epilogue -- no floating-point
//     atomics, no order that depends on who finishes when;
//   * SEQUENTIAL mode (large grids): one workgroup per tile walks all ranges in one uninterrupted K pipeline and folds
//     its accumulators at the range boundaries (`tot = tot + acc; acc = 0`): no scratch traffic at all.
//   Both modes evaluate the same expression tree, so an image's result is bit-identical whatever the batch size selects
//   (tests/test_gpu_layers.py::test_ksplit_*), and with ksplit = 1 the kernel reproduces conv_dma_f16_kernel bit for bit.
//
// Tile 128 x 128 (2 x 2 waves of 64 x 64) or 128 x 64 (4 x 1 waves of 32 x 64) with a FOUR-stage operand ring: the stage
// being read plus two in flight plus the one being refilled; one barrier per 32-deep step, which waits with
// vmcnt(two stages' worth) for the oldest stage only.  LDS reads are inline asm (the compiler would drain vmcnt(0) in
// front of every LDS read while LDS-DMA writes are outstanding); every step issues the same number of DMA instructions
// (past the end from the zero page) so that the vmcnt arithmetic is static.
#include "common.h"
#include "conv_epilogue.h"
#include <cstdlib>
#include <cstring>

namespace xdet {

typedef float ks_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 ks_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef float ks_f4 __attribute__((ext_vector_type(4)));

#define XDET_KS_GLDS16(gptr, lptr)                                                                     \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),              \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <int OFF>
__device__ __forceinline__ ks_f16x8 ks_ds_read_b128(unsigned addr) {
  ks_f16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}

template <int V>
struct ks_int { static constexpr int value = V; };
template <int N, typename F, int I = 0>
__device__ __forceinline__ void ks_static_for(F&& f) {
  if constexpr (I < N) {
    f(ks_int<I>{});
    ks_static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

// PARALLEL: gridDim covers (tile, range) pairs; else one workgroup per tile walks every range.
// NT = KH * KW (1 or 9), a compile-time constant: the tap loop is unrolled, so the per-lane source offset of every
// (tap, piece) -- the only part of a DMA address that is not linear in the channel chunk, because of the planes'
// [pixel / 16][chunk][16][32] blocking and of the image border -- is computed ONCE per tile and kept in a register
// (0xffffffff = out of the image: a raw buffer load then returns zeros, no zero page, no select).  A K step issues its
// pieces as `M0 update + buffer_load_dwordx4 ... lds` with the chunk's offset in an SGPR and spends no VALU instruction
// on addresses (the generic LDS-DMA kernels recompute ~12 per piece, and every issue slot between MFMAs costs
// matrix-pipe time).  Ranges are whole channel chunks: range r = chunks [r * cs, (r + 1) * cs), all taps of each.
template <int BN, int WAVES_M, int WAVES_N, int NSPLIT, bool PARALLEL, int NT>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_dma_ksplit_kernel(ConvParams p) {
  // NW = 8 (round 5; the 128 x 128 tile): two waves per SIMD, each with half the
  // accumulator blocks, half the DMA pieces and half the fragment reads of a step -- with one wave per SIMD nothing runs under
  // a wave's DMA issue and fragment waits, and a step took ~1,800 clocks for 768 of MFMAs.  Same K order, same product
  // order per accumulator: bit-identical to the four-wave form.
  constexpr int BM = 128, NW = WAVES_M * WAVES_N;
  static_assert(NW == 4 || NW == 8, "four or eight waves");
  constexpr bool HALFB = NW == 8 && BN == 64;               // eight waves on the 64-wide tile: ONE weight piece per wave (hi or lo)
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int NSTAGE = 4;
  constexpr bool HALFB_ = WAVES_M * WAVES_N == 8 && BN == 64;
  constexpr int A_IT = BM / (16 * NW), B_IT = HALFB_ ? 1 : BN / (16 * NW);
  constexpr int ROWB = 32;
  constexpr int STAGE = (2 * BM + 2 * BN) * ROWB;          // halves per stage
  constexpr int PIECES = HALFB_ ? 3 : (A_IT + B_IT) * (NSPLIT > 1 ? 2 : 1);
  static_assert(!HALFB_ || NSPLIT == 3, "eight waves on the 64-wide tile: the f16x3 form");
  extern __shared__ __attribute__((aligned(16))) u16 smem16[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int S = p.ksplit;
  const int nby = p.Cout_pad / BN;
  // XCD-aware order as in conv_dma_f16_kernel (all N tiles of an M tile on one XCD); in parallel mode the S ranges of a
  // tile are S consecutive slots of the same XCD
  const int slot = blockIdx.x >> 3;
  const int tslot = PARALLEL ? slot / S : slot;
  const int range = PARALLEL ? slot - tslot * S : 0;
  const int bx = (tslot / nby) * 8 + (blockIdx.x & 7);
  if (bx * BM >= p.M) return;
  const int n0 = (tslot % nby) * BN;
  const int m0 = bx * BM;
  const int tile = bx * nby + tslot % nby;                 // dense tile index (scratch slabs)

  const int ncc = p.Cin_p >> 5;                            // channel chunks; K steps = ncc * NT
  const unsigned c32n = (unsigned)(p.ldi >> 5);
  const int cs = (ncc + S - 1) / S;                        // chunks per range (the last range may be shorter)
  const int cc_lo = PARALLEL ? min(range * cs, ncc) : 0;
  const int cc_hi = PARALLEL ? min(cc_lo + cs, ncc) : ncc;

  // ---- per-lane source offsets (bytes), constant along the channel chunks ----
  const int lr = lane >> 2, pos = lane & 3;
  unsigned a_vo[NT][A_IT], b_vo[B_IT];
#pragma unroll
  for (int q = 0; q < A_IT; ++q) {
    const int rt = (wave * A_IT + q) * 16 + lr;
    const unsigned achunk = (unsigned)((pos ^ ((rt >> 2) & 3)) * 16);
    const int m = m0 + rt;
    const bool row_ok = m < p.M;
    const int hw = p.Ho * p.Wo;
    const int n = row_ok ? m / hw : 0;
    const int rem = m - n * hw;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l, pbase = n * p.H * p.W;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      constexpr int KW = NT == 9 ? 3 : 1;
      const int iy = iy0 + (t / KW) * p.dil, ix = ix0 + (t % KW) * p.dil;
      const bool ok = row_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const unsigned pix = (unsigned)(pbase + iy * p.W + ix);
      a_vo[t][q] = ok ? (((pix >> 4) * c32n) << 10) + ((pix & 15) << 6) + achunk : 0xffffffffu;
    }
  }
#pragma unroll
  for (int q = 0; q < B_IT; ++q) {
    const int rt = (HALFB ? (wave >> 1) : wave * B_IT + q) * 16 + lr;
    b_vo[q] = (unsigned)(((n0 + rt) * 32 + (pos ^ ((rt >> 2) & 3)) * 8) * 2);
  }
  const unsigned a_bytes = (unsigned)((((size_t)p.N * p.H * p.W + 15) >> 4) * c32n << 10);
  const unsigned b_bytes = (unsigned)((size_t)ncc * NT * p.Cout_pad * 64);
  const unsigned b_cc = (unsigned)p.Cout_pad * 64u;        // bytes per K block of the weights

  // a stage past the end of this workgroup's span still issues its PIECES instructions (the vmcnt arithmetic is
  // static) -- against a zero-length buffer: every lane out of bounds, zeros into a ring slot nobody reads
  auto issue = [&](int cc, auto TAP, int buf) {
    constexpr int tap = decltype(TAP)::value;
    const bool live = cc < cc_hi;
    const __amdgpu_buffer_rsrc_t r_ah = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.in_hi), 0, live ? (int)a_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_al = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(NSPLIT > 1 ? p.in_lo : p.in_hi), 0, live ? (int)a_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_bh = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.wt_hi), 0, live ? (int)b_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_bl = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(NSPLIT > 1 ? p.wt_lo : p.wt_hi), 0, live ? (int)b_bytes : 0, 0x00020000);
    u16* Ah = smem16 + buf * STAGE;
    u16* Al = Ah + BM * ROWB;
    u16* Bh = Al + BM * ROWB;
    u16* Bl = Bh + BN * ROWB;
    const unsigned a_so = (unsigned)cc << 10;
    const unsigned b_so = (unsigned)(tap * ncc + cc) * b_cc;     // K block tap * ncc + cc (k = tap * Cin_p + ci)
#pragma unroll
    for (int q = 0; q < A_IT; ++q) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_ah, (__attribute__((address_space(3))) void*)(Ah + (wave * A_IT + q) * 16 * ROWB), 16,
                                               (int)a_vo[tap][q], (int)a_so, 0, 0);
      if (NSPLIT > 1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_al, (__attribute__((address_space(3))) void*)(Al + (wave * A_IT + q) * 16 * ROWB), 16,
                                                 (int)a_vo[tap][q], (int)a_so, 0, 0);
    }
    if constexpr (HALFB) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds((wave & 1) ? r_bl : r_bh, (__attribute__((address_space(3))) void*)(((wave & 1) ? Bl : Bh) + (wave >> 1) * 16 * ROWB), 16,
                                               (int)b_vo[0], (int)b_so, 0, 0);
    } else {
#pragma unroll
    for (int q = 0; q < B_IT; ++q) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_bh, (__attribute__((address_space(3))) void*)(Bh + (wave * B_IT + q) * 16 * ROWB), 16,
                                               (int)b_vo[q], (int)b_so, 0, 0);
      if (NSPLIT > 1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_bl, (__attribute__((address_space(3))) void*)(Bl + (wave * B_IT + q) * 16 * ROWB), 16,
                                                 (int)b_vo[q], (int)b_so, 0, 0);
    }
    }
  };

  ks_f32x16 acc[TM][TN], tot[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

  const int frow = lane & 31, fh = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(smem16);
  // byte offsets of this lane's fragments inside a stage, per 16-deep half
  unsigned a_off[2][TM], b_off[2][TN];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = h * 2 + fh;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ra = wm * WM + i * 32 + frow;
      a_off[h][i] = (unsigned)(ra * ROWB + ((c ^ ((ra >> 2) & 3)) << 3)) * 2u;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int rb = wn * WN + j * 32 + frow;
      b_off[h][j] = (unsigned)(2 * BM * ROWB + rb * ROWB + ((c ^ ((rb >> 2) & 3)) << 3)) * 2u;
    }
  }

  // prologue: the first NSTAGE - 1 stages of the span
  ks_static_for<NSTAGE - 1>([&](auto J) {
    constexpr int j = decltype(J)::value;
    issue(cc_lo + j / NT, ks_int<j % NT>{}, j);
  });
  int ring = 0;
  int next_fold = PARALLEL ? (1 << 30) : cs;               // sequential mode: fold at every range boundary
  bool first = true;
  for (int cc = cc_lo; cc < cc_hi; ++cc) {
    if (!PARALLEL && cc == next_fold) {
      // range boundary: tot = (first ? acc : tot + acc), one f32 add per element, then a fresh accumulator
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          tot[i][j] = first ? acc[i][j] : tot[i][j] + acc[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
      first = false;
      next_fold += cs;
    }
    ks_static_for<NT>([&](auto TAP) {
      constexpr int tap = decltype(TAP)::value;
      // this stage has landed (two younger stages may still be in flight) and every wave is done reading the stage
      // before it, whose ring slot the new DMA overwrites
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NSTAGE - 2) * PIECES) : "memory");
      issue(cc + (tap + NSTAGE - 1) / NT, ks_int<(tap + NSTAGE - 1) % NT>{}, (ring + NSTAGE - 1) & (NSTAGE - 1));
      __builtin_amdgcn_sched_barrier(0);
      const unsigned sb = lds0 + (unsigned)(ring * STAGE * 2);
      ks_f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ah[h][i] = ks_ds_read_b128<0>(sb + a_off[h][i]);
          if (NSPLIT > 1) al[h][i] = ks_ds_read_b128<BM * ROWB * 2>(sb + a_off[h][i]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[h][j] = ks_ds_read_b128<0>(sb + b_off[h][j]);
          if (NSPLIT > 1) bl[h][j] = ks_ds_read_b128<BN * ROWB * 2>(sb + b_off[h][j]);
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // the LDS returns a wave's reads in order: half 0 is complete once at most half 1's reads are outstanding (a
        // scalar load the compiler may have in flight only makes the wait stricter)
        constexpr int PER_HALF = (TM + TN) * (NSPLIT > 1 ? 2 : 1);
        if (h == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PER_HALF) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // tie the fragments of this half to the wait (the MFMAs below must not be scheduled above it)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          asm volatile("" : "+v"(ah[h][i]));
          if (NSPLIT > 1) asm volatile("" : "+v"(al[h][i]));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          asm volatile("" : "+v"(bh[h][j]));
          if (NSPLIT > 1) asm volatile("" : "+v"(bl[h][j]));
        }
        if (NSPLIT > 1) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[h][i], bh[h][j], acc[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[h][i], bl[h][j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[h][i], bh[h][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                 // half 0's MFMAs stay in front of the wait for half 1
      }
      ring = (ring + 1) & (NSTAGE - 1);
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the dummy tail stages write LDS too

  if (!PARALLEL) {
    if (S > 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = first ? acc[i][j] : tot[i][j] + acc[i][j];
    }
  } else if (S > 1 && !p.ks_ticket) {
    // park the raw accumulators: slab [tile][range][v][tid] of float4 (every store instruction one contiguous 4 / 8 KB)
    constexpr int V = TM * TN * 4;
    ks_f4* slab = reinterpret_cast<ks_f4*>(p.ks_partial) + ((size_t)tile * S + range) * V * (64 * NW);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          slab[((i * TN + j) * 4 + q) * (64 * NW) + tid] =
              ks_f4{acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
    return;                                                // the fold + epilogue: conv_ksplit_fold_kernel, next launch
  }
  if (PARALLEL && S > 1 && p.ks_ticket) {
    // ---- the fold INSIDE this launch, by the last workgroup of the tile to arrive (round 6).  Round 4 tried this with
    // device-scope release / acquire fences around the ticket and dropped it: the fences write back / invalidate a whole XCD's
    // L2 (30-40 us per launch).  Here nothing is fenced (cdna_hip_programming.md Guideline 16, form R1): the slabs are written
    // with write-through (`sc1`) 16-byte stores and read with `sc1` loads, every wave drains its stores, ONE lane takes the
    // ticket with a relaxed agent-scope add.  Whoever draws ksplit - 1 knows that every other range's stores were drained
    // before its add: it folds the S slabs -- its own too, from memory, so that the order is the range order whoever comes
    // last -- and runs the epilogue.  No polling: nobody waits for anybody.
    constexpr int V = TM * TN * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.ks_partial + (size_t)tile * S * V * (NW * 256), 0,
                                                                        S * V * (NW * 1024), 0x00020000);
    const int vo = tid * 16;
    typedef unsigned ks_u4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const ks_f4 x = {acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ks_u4, x), rs, vo, (range * V + (i * TN + j) * 4 + q) * (NW * 1024), 16);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* s_last = reinterpret_cast<int*>(smem16);          // (the operand ring is dead)
    if (tid == 0) {
      const int got = __hip_atomic_fetch_add(p.ks_ticket + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *s_last = got == S - 1;
      if (got == S - 1) __hip_atomic_store(p.ks_ticket + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!*s_last) return;
    __syncthreads();                                       // (s_last is read before the epilogue reuses the LDS)
    ks_f4 t[V];
#pragma unroll 1
    for (int r0 = 0; r0 < S; r0 += 2) {
      ks_f4 u0[V], u1[V];
      const bool two = r0 + 1 < S;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        u0[v] = __builtin_bit_cast(ks_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (r0 * V + v) * (NW * 1024), 16));
        u1[v] = __builtin_bit_cast(ks_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (two ? (r0 + 1) * V + v : r0 * V + v) * (NW * 1024), 16));
      }
#pragma unroll
      for (int v = 0; v < V; ++v) {
        t[v] = r0 == 0 ? u0[v] : t[v] + u0[v];             // ((p0 + p1) + p2) + ...
        if (two) t[v] = t[v] + u1[v];
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const ks_f4 x = t[(i * TN + j) * 4 + q];
          acc[i][j][q * 4] = x.x; acc[i][j][q * 4 + 1] = x.y; acc[i][j][q * 4 + 2] = x.z; acc[i][j][q * 4 + 3] = x.w;
        }
  }
  if (m0 + BM <= p.M) conv_epilogue_full<WM, WN, TM, TN, NW, NSTAGE * STAGE * 2>(p, acc, smem16, wave, lane, wm, wn, m0, n0);
  else conv_epilogue<WM, WN, TM, TN, NW, NSTAGE * STAGE * 2>(p, acc, smem16, wave, lane, wm, wn, m0, n0);
}

// Second launch of the parallel mode: ONE WAVE per (tile, wave sub-tile) folds the S slabs in range order -- the thread
// that owned an element in the accumulator layout owns it here, so no transpose is needed -- and runs the shared epilogue
// on its 64 x 64 (32 x 64) sub-tile.  The S loads of an element are issued together (S is a template parameter) and only
// the additions are ordered: a fold with a run-time loop over S was a chain of dependent L2 round trips (42 us for the RPN
// conv of one image, more than its GEMM).  Since round 6 this launch is the FALLBACK (no ticket array, or more tiles than it
// has tickets): the GEMM launch folds by itself, see the `ks_ticket` branch of conv_dma_ksplit_kernel.  (Round 4 had tried
// that with device-scope release / acquire fences around the ticket -- they write back and invalidate a whole XCD's L2,
// 30-40 us per launch with 228 workgroups doing it -- and dropped it; write-through stores need no fence.)
template <int BN, int WAVES_M, int WAVES_N, int S>
__global__ __launch_bounds__(64) void conv_ksplit_fold_kernel(ConvParams p) {
  constexpr int BM = 128;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int V = TM * TN * 4;
  constexpr int LDS_BYTES = 32 * (WN + 4) * 4;
  __shared__ __attribute__((aligned(16))) u16 smem16[LDS_BYTES / 2];
  constexpr int NW = WAVES_M * WAVES_N;                    // waves of the GEMM workgroup (4 or 8)
  const int lane = threadIdx.x;
  const int wave = blockIdx.x % NW;                        // the wave of the GEMM kernel whose sub-tile this is
  const int wg = blockIdx.x / NW;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int nby = p.Cout_pad / BN;
  const int slot = wg >> 3;
  const int bx = (slot / nby) * 8 + (wg & 7);
  if (bx * BM >= p.M) return;
  const int n0 = (slot % nby) * BN, m0 = bx * BM;
  const int tile = bx * nby + slot % nby;
  const int tid = wave * 64 + lane;                        // thread index inside the GEMM workgroup
  // slab loads as raw buffer loads: the thread's offset in ONE register, the (range, element-group) offset in an SGPR --
  // with flat 64-bit addresses the compiler kept all 16 x S of them live and spilled
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.ks_partial + (size_t)tile * S * V * (NW * 256), 0, S * V * (NW * 1024), 0x00020000);
  const int vo = tid * 16;
  // two ranges per pass of a real loop (2 x V loads in flight, then their additions in range order): unrolled over all S
  // ranges the compiler put every load up front and spilled; one range per pass is a chain of S memory round trips
  ks_f4 t[V];
#pragma unroll 1
  for (int r0 = 0; r0 < S; r0 += 2) {
    ks_f4 u0[V], u1[V];
    const bool two = r0 + 1 < S;
#pragma unroll
    for (int v = 0; v < V; ++v) {
      u0[v] = __builtin_bit_cast(ks_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (r0 * V + v) * (NW * 1024), 0));
      u1[v] = __builtin_bit_cast(ks_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (two ? (r0 + 1) * V + v : r0 * V + v) * (NW * 1024), 0));
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
      t[v] = r0 == 0 ? u0[v] : t[v] + u0[v];               // ((p0 + p1) + p2) + ...
      if (two) t[v] = t[v] + u1[v];
    }
  }
  ks_f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const ks_f4 x = t[(i * TN + j) * 4 + q];
        acc[i][j][q * 4] = x.x; acc[i][j][q * 4 + 1] = x.y; acc[i][j][q * 4 + 2] = x.z; acc[i][j][q * 4 + 3] = x.w;
      }
  if (m0 + BM <= p.M) conv_epilogue_full<WM, WN, TM, TN, 1, LDS_BYTES>(p, acc, smem16, 0, lane, wm, wn, m0, n0);
  else conv_epilogue<WM, WN, TM, TN, 1, LDS_BYTES>(p, acc, smem16, 0, lane, wm, wn, m0, n0);
}

template <int BN, int WAVES_M, int WAVES_N>
static int launch_fold(const ConvParams& p, int64_t tiles, hipStream_t s) {
  const dim3 g((unsigned)(tiles * WAVES_M * WAVES_N));
  switch (p.ksplit) {
    case 2: hipLaunchKernelGGL((conv_ksplit_fold_kernel<BN, WAVES_M, WAVES_N, 2>), g, dim3(64), 0, s, p); break;
    case 3: hipLaunchKernelGGL((conv_ksplit_fold_kernel<BN, WAVES_M, WAVES_N, 3>), g, dim3(64), 0, s, p); break;
    case 4: hipLaunchKernelGGL((conv_ksplit_fold_kernel<BN, WAVES_M, WAVES_N, 4>), g, dim3(64), 0, s, p); break;
    case 8: hipLaunchKernelGGL((conv_ksplit_fold_kernel<BN, WAVES_M, WAVES_N, 8>), g, dim3(64), 0, s, p); break;
    default:
      set_last_error("conv(ksplit): the parallel mode folds 2, 3, 4 or 8 ranges");
      return XDET_ERR_UNSUPPORTED;
  }
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

template <int BN, int WAVES_M, int WAVES_N, int NSPLIT, bool PARALLEL, int NT>
static int launch_ks(const ConvParams& p, hipStream_t s) {
  if constexpr (WAVES_M * WAVES_N == 4 && NSPLIT == 3) {
    // the f16x3 form always runs eight waves per workgroup (two per SIMD: profiles/NOTES_r05.md 10)
    return launch_ks<BN, BN == 128 ? 2 : 4, BN == 128 ? 4 : 2, NSPLIT, PARALLEL, NT>(p, s);
  }
  constexpr size_t lds = (size_t)4 * (2 * 128 + 2 * BN) * 32 * sizeof(u16);
  auto kern = conv_dma_ksplit_kernel<BN, WAVES_M, WAVES_N, NSPLIT, PARALLEL, NT>;
  static DeviceOnce once;
  XDET_TRY(ensure_dynamic_lds(once, reinterpret_cast<const void*>(kern), (int)lds));
  const int64_t tiles = cdiv(cdiv(p.M, 128), 8) * 8 * (p.Cout_pad / BN);
  dim3 grid((unsigned)(tiles * (PARALLEL ? p.ksplit : 1)));
  hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES_M * WAVES_N), lds, s, p);
  XDET_LAUNCH_CHECK();
  if (PARALLEL && !p.ks_ticket) return launch_fold<BN, WAVES_M, WAVES_N>(p, tiles, s);
  return XDET_OK;
}

int64_t conv_ksplit_tiles(int64_t M, int cout_pad, int n_tile) { return cdiv(M, 128) * (cout_pad / (n_tile == 128 ? 128 : 64)); }

// what the split-K kernel can run: 1x1 or 3x3 taps (compile-time unrolled), planes and weights below 4 GiB each
bool conv_ksplit_supported(int kh, int kw, int64_t n_pix_in, int ld_in, int cin_p, int cout_pad) {
  if (!((kh == 1 && kw == 1) || (kh == 3 && kw == 3))) return false;
  const size_t a_bytes = (size_t)((n_pix_in + 15) >> 4) * (size_t)(ld_in >> 5) << 10;
  const size_t b_bytes = (size_t)(cin_p >> 5) * kh * kw * cout_pad * 64;
  return a_bytes < ((size_t)1 << 32) && b_bytes < ((size_t)1 << 32);
}

template <int BN, int WAVES_M, int WAVES_N, int NSPLIT>
static int launch_ks_modes(const ConvParams& p, bool par, hipStream_t s) {
  const bool nine = p.KH == 3;
  if (par) return nine ? launch_ks<BN, WAVES_M, WAVES_N, NSPLIT, true, 9>(p, s) : launch_ks<BN, WAVES_M, WAVES_N, NSPLIT, true, 1>(p, s);
  return nine ? launch_ks<BN, WAVES_M, WAVES_N, NSPLIT, false, 9>(p, s) : launch_ks<BN, WAVES_M, WAVES_N, NSPLIT, false, 1>(p, s);
}

// mode: 0 = by grid size, 1 = parallel (needs scratch for every tile), 2 = sequential
int launch_conv_mfma_ksplit(const ConvParams& p_in, int n_tile, int nsplit, int mode, int64_t scratch_tiles, hipStream_t s) {
  ConvParams p = p_in;
  if (cdiv(cdiv(p.M, 128), 8) * 8 * (p.Cout_pad / (n_tile == 128 ? 128 : 64)) > 4096) p.ks_ticket = nullptr;   // (the ticket array's size: fold by a second launch)
  XDET_REQUIRE(p.Kp % 32 == 0 && p.Cin_p % 32 == 0 && p.ldi >= p.Cin_p && p.ldi % 32 == 0 && p.Kp == p.Cin_p * p.KH * p.KW,
               "conv(ksplit): channel counts must be padded to 32");
  XDET_REQUIRE(n_tile == 64 || n_tile == 128, "conv(ksplit): N tile must be 64 or 128");
  XDET_REQUIRE(p.Cout_pad % n_tile == 0, "conv(ksplit): Cout_pad must be a multiple of the N tile");
  XDET_REQUIRE(p.in_hi && (nsplit == 1 || p.in_lo) && p.wt_hi && (nsplit == 1 || p.wt_lo), "conv(ksplit): split planes missing");
  XDET_REQUIRE(p.ksplit >= 1 && p.ksplit <= 16 && p.group_rows == 0, "conv(ksplit): 1 <= ksplit <= 16, no grouped GEMMs");
  XDET_REQUIRE(conv_ksplit_supported(p.KH, p.KW, (int64_t)p.N * p.H * p.W, p.ldi, p.Cin_p, p.Cout_pad),
               "conv(ksplit): 1x1 or 3x3 filters, planes and weights below 4 GiB");
  if (p.M <= 0) return XDET_OK;
  const int64_t tiles = conv_ksplit_tiles(p.M, p.Cout_pad, n_tile);
  bool par = p.ksplit > 1 && tiles * p.ksplit <= 448;     // up to ~1.75 rounds of the 256 CUs (one workgroup per CU)
  if (mode == 1) par = p.ksplit > 1;
  if (p.ksplit != 2 && p.ksplit != 3 && p.ksplit != 4 && p.ksplit != 8) par = false;   // what the fold launch is instantiated for
  if (mode == 2) par = false;
  // chosen by grid size: both modes give the same bits, so a scratch that is too small just means the sequential one;
  // only the forced parallel mode (tests) insists
  if (mode == 1) XDET_REQUIRE(!par || (p.ks_partial && tiles <= scratch_tiles), "conv(ksplit): scratch too small for the parallel mode");
  else par = par && p.ks_partial && tiles <= scratch_tiles;
  // large grids: the same fold inside the 256 x 128 LDS-DMA kernel (needs the zero page: the generic addressing)
  if (!par && mode == 0 && p.zeros && conv_dma_fold_applicable(p, n_tile, nsplit)) return launch_conv_mfma_dma_fold(p, s);
  if (n_tile == 128) return nsplit == 3 ? launch_ks_modes<128, 2, 2, 3>(p, par, s) : launch_ks_modes<128, 2, 2, 1>(p, par, s);
  return nsplit == 3 ? launch_ks_modes<64, 4, 1, 3>(p, par, s) : launch_ks_modes<64, 4, 1, 1>(p, par, s);
}

}  // namespace xdet
