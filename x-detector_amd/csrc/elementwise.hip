// HBM-bound window / element-wise kernels of the backbone (NHWC, 16 B per lane):
//   * NCHW image -> NHWC4 (network entry, light_head_rfcn_eval.py:85 feeds channels_first)
//   * depthwise 3x3 (dilation 1|2, SAME) with optional leading ReLU: the depthwise half of
//     tf.layers.separable_conv2d (net/xception_body.py:220-234,268,354,366)
//   * max-pool 3x3/2 SAME + residual add (net/xception_body.py:281-286,302-307,321-326)
#include "common.h"
#include <cstdlib>

namespace xdet {

__global__ void nchw_to_nhwc4_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C, int H,
                                     int W, int ldo) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / ((int64_t)H * W);
    const int64_t px = i - n * H * W;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C && c < 4; ++c) v[c] = in[(n * C + c) * H * W + px];
    *reinterpret_cast<float4*>(out + i * ldo) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

int launch_nchw_to_nhwc4(const float* in, float* out, int N, int C, int H, int W, int ldo, hipStream_t s) {
  XDET_REQUIRE(C <= 4 && ldo == 4, "nchw_to_nhwc4: at most 4 channels");
  const int64_t total = (int64_t)N * H * W;
  const int blocks = (int)std::min<int64_t>(cdiv(total, 256), 4096);
  hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(blocks), dim3(256), 0, s, in, out, N, C, H, W, ldo);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// Depthwise 3x3, stride 1, SAME, dilation DIL.  One thread = 4 channels x a strip of SX consecutive
// output pixels of one row: the 3 x (SX + 2*DIL) input columns are loaded once and reused by the SX
// outputs (3-4.5 loads per output instead of 9); lanes run along channels (16 B each, coalesced).
// grid.x = N*H rows, grid.y covers ceil(W/SX) * ld/4 items; all index math is 32-bit.
// SPLIT: write the result as f16 hi/lo planes (its only consumer is a pointwise conv on the
// split-precision path) instead of f32.
constexpr int DW_SX = 4;

typedef _Float16 f16x4e __attribute__((ext_vector_type(4)));

// offset (in halves) of channel c of pixel `pix` in a split plane.  Layout: [pix/16][ld/32][16][32]:
// 16 pixels x 32 channels form one contiguous 1 KB block (what one LDS-DMA instruction of the conv
// kernel moves), and all channel blocks of a 16-pixel group are adjacent, so a producer that walks
// pixels x all channels (depthwise, split) still writes one compact region like plain NHWC.
__device__ __forceinline__ int64_t blocked_off(int64_t pix, int c, int c32n) {
  return (((pix >> 4) * c32n + (c >> 5)) << 9) + ((pix & 15) << 5) + (c & 31);
}


// four f32 -> four fp8 (e4m3, OCP) in one dword: x * mul, clamped to the format's +-448 first (the conversion does not
// saturate by itself), round to nearest even
__device__ __forceinline__ unsigned pack_e4m3x4(float a, float b, float c, float d, float mul) {
  a = __builtin_amdgcn_fmed3f(a * mul, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b * mul, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c * mul, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d * mul, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (unsigned)r;
}

// SPLIT: 0 = f32 out, 1 = f16 hi / lo planes, 2 = the x8 form of the planes (conv_params.h): `hi` as before, and in the place
// of the 32 f16 `lo` values of a (pixel, 32-channel block) 32 bytes fp8(hi * x8_hi) + 32 bytes fp8(lo * x8_lo)
template <int SPLIT>
__device__ __forceinline__ void dw_store_strip(const float4* acc, float* __restrict__ out, unsigned short* __restrict__ hi,
                                               unsigned short* __restrict__ lo, int64_t rowpix, int x0, int W, int ld,
                                               int c, float x8_hi = 0.f, float x8_lo = 0.f) {
  if (SPLIT == 2) {
    uint2 h[4];
    unsigned h8[4], l8[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = acc[k];
      const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y, h2 = (_Float16)v.z, h3 = (_Float16)v.w;
      f16x4e hv = {h0, h1, h2, h3};
      h[k] = *reinterpret_cast<uint2*>(&hv);
      h8[k] = pack_e4m3x4((float)h0, (float)h1, (float)h2, (float)h3, x8_hi);
      l8[k] = pack_e4m3x4(v.x - (float)h0, v.y - (float)h1, v.z - (float)h2, v.w - (float)h3, x8_lo);
    }
    // lane pairs swap halves as below: the even lane ends up with 8 channels of pixels 0 and 2, the odd lane of 1 and 3
    const bool odd = threadIdx.x & 1;
    const int cb = c & ~7, c32n = ld >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint2 sh = odd ? h[2 * t] : h[2 * t + 1];
      const unsigned s8h = odd ? h8[2 * t] : h8[2 * t + 1], s8l = odd ? l8[2 * t] : l8[2 * t + 1];
      uint2 rh;
      rh.x = __shfl_xor(sh.x, 1); rh.y = __shfl_xor(sh.y, 1);
      const unsigned r8h = __shfl_xor(s8h, 1), r8l = __shfl_xor(s8l, 1);
      const uint2 mh = odd ? h[2 * t + 1] : h[2 * t];
      const unsigned m8h = odd ? h8[2 * t + 1] : h8[2 * t], m8l = odd ? l8[2 * t + 1] : l8[2 * t];
      const uint4 oh = odd ? make_uint4(rh.x, rh.y, mh.x, mh.y) : make_uint4(mh.x, mh.y, rh.x, rh.y);
      const uint2 o8h = odd ? make_uint2(r8h, m8h) : make_uint2(m8h, r8h);
      const uint2 o8l = odd ? make_uint2(r8l, m8l) : make_uint2(m8l, r8l);
      const int x = x0 + 2 * t + (odd ? 1 : 0);
      if (x < W) {
        *reinterpret_cast<uint4*>(hi + blocked_off(rowpix + x, cb, c32n)) = oh;
        unsigned char* rec = reinterpret_cast<unsigned char*>(lo + blocked_off(rowpix + x, cb & ~31, c32n));   // the block's 64-byte record
        *reinterpret_cast<uint2*>(rec + (cb & 31)) = o8h;
        *reinterpret_cast<uint2*>(rec + 32 + (cb & 31)) = o8l;
      }
    }
  } else if (SPLIT) {
    // lane pairs swap halves: the even lane ends up with all 8 channels of pixels 0 and 2, the odd lane
    // with those of pixels 1 and 3 -> 16-B stores, 128-B runs per 8 lanes (instead of 8-B stores in 64-B runs)
    uint2 h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = acc[k];
      const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y, h2 = (_Float16)v.z, h3 = (_Float16)v.w;
      f16x4e hv = {h0, h1, h2, h3};
      f16x4e lv = {(_Float16)(v.x - (float)h0), (_Float16)(v.y - (float)h1), (_Float16)(v.z - (float)h2),
                   (_Float16)(v.w - (float)h3)};
      h[k] = *reinterpret_cast<uint2*>(&hv);
      l[k] = *reinterpret_cast<uint2*>(&lv);
    }
    const bool odd = threadIdx.x & 1;
    const int cb = c & ~7, c32n = ld >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint2 sh = odd ? h[2 * t] : h[2 * t + 1], sl = odd ? l[2 * t] : l[2 * t + 1];
      uint2 rh, rl;
      rh.x = __shfl_xor(sh.x, 1); rh.y = __shfl_xor(sh.y, 1);
      rl.x = __shfl_xor(sl.x, 1); rl.y = __shfl_xor(sl.y, 1);
      const uint2 mh = odd ? h[2 * t + 1] : h[2 * t], ml = odd ? l[2 * t + 1] : l[2 * t];
      const uint4 oh = odd ? make_uint4(rh.x, rh.y, mh.x, mh.y) : make_uint4(mh.x, mh.y, rh.x, rh.y);
      const uint4 ol = odd ? make_uint4(rl.x, rl.y, ml.x, ml.y) : make_uint4(ml.x, ml.y, rl.x, rl.y);
      const int x = x0 + 2 * t + (odd ? 1 : 0);
      if (x < W) {
        const int64_t o = blocked_off(rowpix + x, cb, c32n);
        *reinterpret_cast<uint4*>(hi + o) = oh;
        *reinterpret_cast<uint4*>(lo + o) = ol;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (x0 + k < W) *reinterpret_cast<float4*>(out + (size_t)(rowpix + x0 + k) * ld + c) = acc[k];
  }
}

template <int DIL, bool SPLIT>
__global__ __launch_bounds__(256) void depthwise3x3_kernel(const float* __restrict__ in, const float* __restrict__ w9c,
                                                           float* __restrict__ out, unsigned short* __restrict__ hi,
                                                           unsigned short* __restrict__ lo, int H, int W, int ld,
                                                           int relu_in, int nrows, int yblocks) {
  static_assert(DW_SX == 4, "strip of 4");
  constexpr int NC = DW_SX + 2 * DIL;
  const int c4n = ld >> 2;
  const int nstrip = (W + DW_SX - 1) / DW_SX;
  // 1-D grid, XCD-aware: workgroup b runs on XCD b & 7; inside an XCD consecutive workgroups walk the
  // `yblocks` strip/channel slices of one row first and the rows of that XCD's band second, so the
  // workgroups in flight together cover few rows x all slices and a row's neighbours (which re-read it)
  // are scheduled right after it -- with slices outermost the set in flight spanned the whole band
  // (~6 MB per XCD against a 4 MB L2) and rows were fetched 2.9x.
  const int slot = blockIdx.x >> 3;
  const int rl = slot / yblocks;
  const int item = (slot - rl * yblocks) * 256 + threadIdx.x;
  if (item >= nstrip * c4n) return;
  const int strip = item / c4n;
  const int c = (item - strip * c4n) * 4;
  const int x0 = strip * DW_SX;
  // One workgroup per output row (x 256 strip/channel items), 64 VGPRs = full occupancy: this stencil
  // is latency-bound, and every variant that traded occupancy for reuse in registers (sliding window
  // down the rows: 148-180 VGPRs) or walked several rows per workgroup measured 10-60 % slower.
  const int per_xcd = (nrows + 7) >> 3;
  const int row = (blockIdx.x & 7) * per_xcd + rl;                // n*H + y
  if (rl >= per_xcd || row >= nrows) return;
  const int y = row % H;
  const float* base = in + (size_t)(row - y) * W * ld + c;       // image origin + channel offset
  {
    float4 acc[DW_SX];
#pragma unroll
    for (int k = 0; k < DW_SX; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y + (ky - 1) * DIL;
      if ((unsigned)iy >= (unsigned)H) continue;
      const float* rp = base + (size_t)iy * W * ld;
      float4 col[NC];
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const int ix = x0 - DIL + k;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)ix < (unsigned)W) {
          v = *reinterpret_cast<const float4*>(rp + (size_t)ix * ld);
          if (relu_in) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          }
        }
        col[k] = v;
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float4 w = *reinterpret_cast<const float4*>(w9c + (ky * 3 + kx) * ld + c);
#pragma unroll
        for (int k = 0; k < DW_SX; ++k) {
          const float4 v = col[k + kx * DIL];
          acc[k].x = fmaf(v.x, w.x, acc[k].x); acc[k].y = fmaf(v.y, w.y, acc[k].y);
          acc[k].z = fmaf(v.z, w.z, acc[k].z); acc[k].w = fmaf(v.w, w.w, acc[k].w);
        }
      }
    }
    dw_store_strip<SPLIT ? 1 : 0>(acc, out, hi, lo, (int64_t)row * W, x0, W, ld, c);
  }
}

// ---------------------------------------------------------------------------------------
// Depthwise 3x3 (dilation 1 or 2) as a software-pipelined tile kernel.  The per-row kernel above is
// latency-bound (three dependent load phases per thread, every input pixel requested 4.5 times through
// the TA/L1; PMC: waves 82 % in s_waitcnt at 54 % of the streaming rate).  Here a workgroup walks a
// range of tiles (DT_R output rows x DT_X pixels x 32 channels); the (DT_R+2*DIL) x (DT_X+2*DIL) input
// patch of the NEXT tile is pulled into LDS by buffer_load_dwordx4 ... lds (no VGPRs; a lane that must read
// padding is sent out of the buffer's bounds and gets zeros) while the current tile is computed out of LDS, so each input pixel crosses the TA once
// per tile and the HBM latency hides behind compute instead of behind occupancy.  FMA order per output
// is the per-row kernel's (ky, kx ascending): results are bit-identical.
// ---------------------------------------------------------------------------------------
constexpr int DT_R = 4, DT_X = 32, DT_P = 40;

template <int DIL, int SPLIT>
__global__ __launch_bounds__(256) void depthwise3x3_tile_kernel(const float* __restrict__ in,
                                                                const float* __restrict__ w9c, float* __restrict__ out,
                                                                unsigned short* __restrict__ hi,
                                                                unsigned short* __restrict__ lo, int N, int H, int W,
                                                                int ld, int relu_in, int ntiles, int TY, int TX,
                                                                int tiles_per_block, int n_first, float x8_hi, float x8_lo) {
  // `in` points at image n_first of the tensor and N images are covered (one launch per range of images below 2 GiB:
  // the DMA's buffer offsets are 32-bit); `out` / `hi` / `lo` are the whole tensor's, indexed with n_first + n
  static_assert(DT_X + 2 * DIL <= DT_P, "patch row does not fit the LDS pitch");
  constexpr int DT_ROWS = DT_R + 2 * DIL, DT_TILE_F = DT_ROWS * DT_P * 32;   // floats per LDS tile buffer
  constexpr int NC = 4 + 2 * DIL;                                          // input pixels a strip of 4 needs
  extern __shared__ __attribute__((aligned(16))) float dw_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // = output row of the tile
  const int c4 = lane & 7, strip = lane >> 3;
  const int t_begin = blockIdx.x * tiles_per_block;
  const int t_end = min(ntiles, t_begin + tiles_per_block);
  if (t_begin >= t_end) return;

  // Tile order: ty fastest (vertically adjacent tiles are consecutive: shared halo rows come from L2),
  // then tx, image, channel chunk.  Coordinates are advanced incrementally (no divisions in the loop).
  struct Coord { int ty, tx, n, chunk; };
  auto advance = [&](Coord& c) {
    if (++c.ty == TY) { c.ty = 0; if (++c.tx == TX) { c.tx = 0; if (++c.n == N) { c.n = 0; ++c.chunk; } } }
  };
  Coord cur;
  {
    int q = t_begin;
    cur.ty = q % TY; q /= TY;
    cur.tx = q % TX; q /= TX;
    cur.n = q % N;
    cur.chunk = q / N;
  }
  // DMA pieces of this wave: instruction i = wave + 4*jj moves 8 pixels x 128 B of patch row rr_j,
  // segment seg_j.  Everything that does not depend on the tile is computed once: per lane the float
  // offset of its 16 B relative to the tile's first pixel and its x relative to the tile's x0.  The
  // loads are raw buffer loads over the whole input tensor: a lane that must read padding gets the
  // offset 0xffffffff, which the buffer bounds check turns into zeros (no zero page, no branches).
  constexpr int NSEG = DT_P / 8, NPIECE = DT_ROWS * NSEG, NJ = (NPIECE + 3) / 4;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((size_t)N * H * W * ld * 4), 0x00020000);
  int lane_rel[NJ], x_rel[NJ], p_row[NJ], p_x[NJ], p_lds[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int i = wave + 4 * jj;
    const int rr = i / NSEG, seg = i - rr * NSEG;
    p_row[jj] = i < NPIECE ? rr - DIL : -(1 << 20);            // wave-uniform: row relative to the tile, x of lane 0,
    p_x[jj] = seg * 8 - DIL;                                    // LDS float offset of the piece
    p_lds[jj] = (rr * DT_P + seg * 8) * 32;
    x_rel[jj] = seg * 8 + (lane >> 3) - DIL;
    lane_rel[jj] = (((rr - DIL) * W + x_rel[jj]) * ld + (lane & 7) * 4) * 4;    // bytes
  }
  auto issue = [&](const Coord& c, int buf) {
    const int y0 = c.ty * DT_R, x0 = c.tx * DT_X;
    const int tile_base = ((((c.n * H + y0) * W + x0) * ld) + c.chunk * 32) * 4;   // bytes, < 2^31 (checked on the host)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      if (p_row[jj] < -DIL || x0 + p_x[jj] >= W + DIL) continue;  // no such piece / a segment right of the halo columns
      const bool rok = (unsigned)(y0 + p_row[jj]) < (unsigned)H;  // wave-uniform
      const bool ok = rok && (unsigned)(x0 + x_rel[jj]) < (unsigned)W;
      const unsigned voff = ok ? (unsigned)(tile_base + lane_rel[jj]) : 0xffffffffu;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsrc, (__attribute__((address_space(3))) void*)(dw_lds + buf * DT_TILE_F + p_lds[jj]), 16, voff, 0, 0, 0);
    }
  };

  float4 w[9];
  int cur_chunk = -1;
  issue(cur, 0);
  for (int t = t_begin, it = 0; t < t_end; ++t, ++it) {
    const int buf = it & 1;
    const int c = cur.chunk * 32 + c4 * 4;
    if (cur.chunk != cur_chunk) {            // (re)load the taps ahead of the barrier, whose vmcnt(0) retires them:
#pragma unroll                               // issued after the DMA they would drag its completion into the FMAs
      for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const float4*>(w9c + k * ld + c);
      cur_chunk = cur.chunk;
    }
    __syncthreads();                         // tile t has landed; everyone is done with the other buffer
    Coord nxt = cur;
    advance(nxt);
    const int y = cur.ty * DT_R + wave, x0 = cur.tx * DT_X, n = cur.n;
    cur = nxt;
    // The whole 3 x 6 neighbourhood goes to registers BEFORE the next tile's DMA is issued: the compiler
    // orders every LDS read after an outstanding LDS-DMA write with a full vmcnt(0) wait, which would
    // serialise the prefetch with this tile's reads.
    const float* T = dw_lds + buf * DT_TILE_F + (strip * 4) * 32 + c4 * 4;
    const float lo_clip = relu_in ? 0.f : -INFINITY;
    float4 col[3][NC];
    if (y < H) {                              // wave-uniform
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          float4 v = *reinterpret_cast<const float4*>(T + ((wave + ky * DIL) * DT_P + k) * 32);
          v.x = fmaxf(v.x, lo_clip); v.y = fmaxf(v.y, lo_clip); v.z = fmaxf(v.z, lo_clip); v.w = fmaxf(v.w, lo_clip);
          col[ky][k] = v;
        }
    }
    if (t + 1 < t_end) issue(nxt, buf ^ 1);
    if (y >= H) continue;
    float4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float4 ww = w[ky * 3 + kx];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 v = col[ky][k + kx * DIL];
          acc[k].x = fmaf(v.x, ww.x, acc[k].x); acc[k].y = fmaf(v.y, ww.y, acc[k].y);
          acc[k].z = fmaf(v.z, ww.z, acc[k].z); acc[k].w = fmaf(v.w, ww.w, acc[k].w);
        }
      }
    dw_store_strip<SPLIT>(acc, out, hi, lo, ((int64_t)(n_first + n) * H + y) * W, x0 + strip * 4, W, ld, c, x8_hi, x8_lo);
  }
}

static int launch_dw(const float* in, const float* w9c, float* out, unsigned short* hi, unsigned short* lo, int N,
                     int H, int W, int C, int ld, int dil, int relu_in, hipStream_t s, int x8 = 0, int x8_exp = 0) {
  const float x8_hi = std::ldexp(1.f, -x8_exp), x8_lo = std::ldexp(1.f, 11 - x8_exp);
  XDET_REQUIRE(ld % 4 == 0 && ld >= C, "depthwise: channel stride must be a multiple of 4");
  XDET_REQUIRE(dil == 1 || dil == 2, "depthwise: dilation must be 1 or 2");
  if ((int64_t)N * H == 0) return XDET_OK;
  const int items = ((W + DW_SX - 1) / DW_SX) * (ld / 4);
  const int nrows = N * H;
  const int yblocks = (int)cdiv(items, 256);
  const dim3 grid((unsigned)(cdiv(nrows, 8) * 8 * yblocks));
  const bool split = hi != nullptr;
  // the LDS-pipelined tile kernel; its 32-bit buffer offsets cover 2 GiB: one launch per range of whole images below that
  // (the 397 x 397 x 128 tensor of the 800 x 800 input at batch 96 is 7.7 GB; round 3 sent it to the per-row kernel:
  // 11.7 % of that workload's GPU time).  A single image beyond 2 GiB still takes the per-row kernel.
  const int64_t per_image = (int64_t)H * W * ld * 4;
  if (ld % 32 == 0 && per_image < ((int64_t)1 << 31)) {
    const int n_max = (int)std::max<int64_t>(1, (((int64_t)1 << 31) - 1) / per_image);
    const int TY = (int)cdiv(H, DT_R), TX = (int)cdiv(W, DT_X);
    const size_t lds = (size_t)2 * (DT_R + 2 * dil) * DT_P * 32 * sizeof(float);
    if (dil == 2) {
      static DeviceOnce once_t, once_f, once_x;
      XDET_TRY(ensure_dynamic_lds(once_t, reinterpret_cast<const void*>(depthwise3x3_tile_kernel<2, 1>), (int)lds));
      XDET_TRY(ensure_dynamic_lds(once_f, reinterpret_cast<const void*>(depthwise3x3_tile_kernel<2, 0>), (int)lds));
      XDET_TRY(ensure_dynamic_lds(once_x, reinterpret_cast<const void*>(depthwise3x3_tile_kernel<2, 2>), (int)lds));
    }
    for (int nb = 0; nb < N; nb += n_max) {
      const int n = std::min(n_max, N - nb);
      const float* in_r = in + (size_t)nb * H * W * ld;
      const int64_t nt = (int64_t)(ld / 32) * n * TY * TX;
      const int blocks = (int)std::min<int64_t>(nt, lds > 80 * 1024 ? 256 : 512);   // workgroups per CU that fit in LDS
      const int tpb = (int)cdiv(nt, blocks);
      const dim3 g((unsigned)cdiv(nt, tpb));
#define XDET_DW_TILE(D, S) hipLaunchKernelGGL((depthwise3x3_tile_kernel<D, S>), g, dim3(256), lds, s, in_r, w9c, out, hi, lo, n, H, W, ld, relu_in, (int)nt, TY, TX, tpb, nb, x8_hi, x8_lo)
      if (dil == 1) {
        if (split && x8) XDET_DW_TILE(1, 2);
        else if (split) XDET_DW_TILE(1, 1);
        else XDET_DW_TILE(1, 0);
      } else {
        if (split && x8) XDET_DW_TILE(2, 2);
        else if (split) XDET_DW_TILE(2, 1);
        else XDET_DW_TILE(2, 0);
      }
#undef XDET_DW_TILE
      XDET_LAUNCH_CHECK();
    }
    return XDET_OK;
  }
  XDET_REQUIRE(!x8, "depthwise: x8 planes need the tile kernel (channel stride a multiple of 32, images below 2 GiB)");
  if (dil == 1 && !split) hipLaunchKernelGGL((depthwise3x3_kernel<1, false>), grid, dim3(256), 0, s, in, w9c, out, hi, lo, H, W, ld, relu_in, nrows, yblocks);
  else if (dil == 1) hipLaunchKernelGGL((depthwise3x3_kernel<1, true>), grid, dim3(256), 0, s, in, w9c, out, hi, lo, H, W, ld, relu_in, nrows, yblocks);
  else if (!split) hipLaunchKernelGGL((depthwise3x3_kernel<2, false>), grid, dim3(256), 0, s, in, w9c, out, hi, lo, H, W, ld, relu_in, nrows, yblocks);
  else hipLaunchKernelGGL((depthwise3x3_kernel<2, true>), grid, dim3(256), 0, s, in, w9c, out, hi, lo, H, W, ld, relu_in, nrows, yblocks);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

int launch_depthwise3x3(const float* in, const float* w9c, float* out, int N, int H, int W, int C, int ld, int dil,
                        int relu_in, hipStream_t s) {
  return launch_dw(in, w9c, out, nullptr, nullptr, N, H, W, C, ld, dil, relu_in, s);
}

int launch_depthwise3x3_split(const float* in, const float* w9c, unsigned short* hi, unsigned short* lo, int N,
                              int H, int W, int C, int ld, int dil, int relu_in, hipStream_t s, int x8, int x8_exp) {
  XDET_REQUIRE(hi && lo, "depthwise(split): NULL planes");
  XDET_REQUIRE(ld % 32 == 0, "depthwise(split): channel stride must be a multiple of 32");
  return launch_dw(in, w9c, nullptr, hi, lo, N, H, W, C, ld, dil, relu_in, s, x8, x8_exp);
}

// ---- split-precision planes: x = hi + lo, both f16 (the A operand format of conv_mfma_dma.hip) ----
// in: NHWC f32 [n_pix][ld]; hi/lo: split planes [n_pix/16][ld/32][16][32] (blocked_off above).
// One thread = 8 channels of one pixel (two float4 in, one 16-B store per plane); consecutive lanes
// walk the OUTPUT order, so a wave writes one 1 KB block (16 pixels x 32 channels) and reads whole
// 128-B lines (one pixel's 32-channel chunk per 4 lanes); consecutive waves take the next channel
// block of the same 16 pixels.
typedef _Float16 f16x8e __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const float4 a, const float4 b, f16x8e* h, f16x8e* l) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 t = (_Float16)v[i];
    (*h)[i] = t;
    (*l)[i] = (_Float16)(v[i] - (float)t);
  }
}

// mul: a power of two (the tensor's activation pre-scale 2^-e, 1 by default): the planes hold x * mul
template <bool X8>
__global__ __launch_bounds__(256) void split_f32_kernel(const float* __restrict__ in, unsigned short* __restrict__ hi,
                                                        unsigned short* __restrict__ lo, int64_t n_pix, int ld,
                                                        int relu, float mul, float x8_hi, float x8_lo) {
  const int c32n = ld >> 5;
  const int64_t n8 = ((n_pix + 15) >> 4) * c32n * 64;   // 16-B output chunks per plane (64 per 1 KB block)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t blk = i >> 6;                  // (pix/16) * c32n + cc
    const int64_t grp = blk / c32n;
    const int cc = (int)(blk - grp * c32n);
    const int64_t pix = grp * 16 + ((i >> 2) & 15);
    if (pix >= n_pix) continue;
    const float* src = in + pix * ld + cc * 32 + (i & 3) * 8;
    float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    if (relu) {
      a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
      b.x = fmaxf(b.x, 0.f); b.y = fmaxf(b.y, 0.f); b.z = fmaxf(b.z, 0.f); b.w = fmaxf(b.w, 0.f);
    }
    a.x *= mul; a.y *= mul; a.z *= mul; a.w *= mul;
    b.x *= mul; b.y *= mul; b.z *= mul; b.w *= mul;
    f16x8e h, l;
    split8(a, b, &h, &l);
    *reinterpret_cast<f16x8e*>(hi + i * 8) = h;
    if (X8) {   // 8 channels of the (pixel, block) record: 8 bytes of hi8 at +c, 8 bytes of lo8 at +32 + c
      const float hf[8] = {(float)h[0], (float)h[1], (float)h[2], (float)h[3], (float)h[4], (float)h[5], (float)h[6], (float)h[7]};
      const float vf[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      unsigned char* rec = reinterpret_cast<unsigned char*>(lo + (i & ~(int64_t)3) * 8);
      const int cb = (int)(i & 3) * 8;
      *reinterpret_cast<uint2*>(rec + cb) = make_uint2(pack_e4m3x4(hf[0], hf[1], hf[2], hf[3], x8_hi), pack_e4m3x4(hf[4], hf[5], hf[6], hf[7], x8_hi));
      *reinterpret_cast<uint2*>(rec + 32 + cb) =
          make_uint2(pack_e4m3x4(vf[0] - hf[0], vf[1] - hf[1], vf[2] - hf[2], vf[3] - hf[3], x8_lo),
                     pack_e4m3x4(vf[4] - hf[4], vf[5] - hf[5], vf[6] - hf[6], vf[7] - hf[7], x8_lo));
    } else {
      *reinterpret_cast<f16x8e*>(lo + i * 8) = l;
    }
  }
}

int launch_split_f32(const float* in, unsigned short* hi, unsigned short* lo, int64_t n_pix, int ld, int relu,
                     hipStream_t s, float mul, int x8, int x8_exp) {
  XDET_REQUIRE(ld > 0 && ld % 32 == 0, "split: channel stride must be a multiple of 32");
  if (n_pix == 0) return XDET_OK;
  const int64_t n = cdiv(n_pix, 16) * 16 * ld;
  const int blocks = (int)std::min<int64_t>(cdiv(n / 8, 256), 256 * 32);
  if (x8) hipLaunchKernelGGL(split_f32_kernel<true>, dim3(blocks), dim3(256), 0, s, in, hi, lo, n_pix, ld, relu, mul, std::ldexp(1.f, -x8_exp), std::ldexp(1.f, 11 - x8_exp));
  else hipLaunchKernelGGL(split_f32_kernel<false>, dim3(blocks), dim3(256), 0, s, in, hi, lo, n_pix, ld, relu, mul, 0.f, 0.f);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// Every second pixel of every second row -> split planes of the [N, ceil(H/2), ceil(W/2), ld] tensor: the A operand
// of a 1x1 / stride-2 SAME convolution (the residual projections conv2d_1..3, net/xception_body.py:261-266): a 1x1
// kernel needs no padding, output (oy, ox) reads input (2 oy, 2 ox).  The subsampled tensor is a quarter of the
// input, and the projection then runs on the LDS-DMA GEMM instead of gathering strided rows through registers.
__global__ __launch_bounds__(256) void split_f32_subsample2_kernel(const float* __restrict__ in,
                                                                   unsigned short* __restrict__ hi,
                                                                   unsigned short* __restrict__ lo, int N, int H, int W,
                                                                   int Ho, int Wo, int ld, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, float mul, int c32_dst) {
  // c32_dst: channel blocks per 16-pixel group of the DESTINATION (= ld / 32, or more when hi / lo point at this tensor's first
  // block inside a wider concatenated operand)
  const int c32n = ld >> 5;
  const int64_t n_pix = (int64_t)N * Ho * Wo;
  const int64_t n8 = ((n_pix + 15) >> 4) * c32n * 64;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t blk = i >> 6;
    const int64_t grp = blk / c32n;
    const int cc = (int)(blk - grp * c32n);
    const int64_t pix = grp * 16 + ((i >> 2) & 15);
    if (pix >= n_pix) continue;
    const int n = (int)(pix / ((int64_t)Ho * Wo));
    const int rem = (int)(pix - (int64_t)n * Ho * Wo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    const float* src = in + (((size_t)n * H + 2 * oy) * W + 2 * ox) * ld + cc * 32 + (i & 3) * 8;
    float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    if (scale) {   // relu(x * scale + shift): a pre-activation BN in front of the projection (net/resnet_v2.py:142-156)
      const float* sc = scale + cc * 32 + (i & 3) * 8;
      const float* sh = shift + cc * 32 + (i & 3) * 8;
      const float4 s0 = *reinterpret_cast<const float4*>(sc), s1 = *reinterpret_cast<const float4*>(sc + 4);
      const float4 h0 = *reinterpret_cast<const float4*>(sh), h1 = *reinterpret_cast<const float4*>(sh + 4);
      a.x = fmaxf(fmaf(a.x, s0.x, h0.x), 0.f); a.y = fmaxf(fmaf(a.y, s0.y, h0.y), 0.f);
      a.z = fmaxf(fmaf(a.z, s0.z, h0.z), 0.f); a.w = fmaxf(fmaf(a.w, s0.w, h0.w), 0.f);
      b.x = fmaxf(fmaf(b.x, s1.x, h1.x), 0.f); b.y = fmaxf(fmaf(b.y, s1.y, h1.y), 0.f);
      b.z = fmaxf(fmaf(b.z, s1.z, h1.z), 0.f); b.w = fmaxf(fmaf(b.w, s1.w, h1.w), 0.f);
    }
    a.x *= mul; a.y *= mul; a.z *= mul; a.w *= mul;
    b.x *= mul; b.y *= mul; b.z *= mul; b.w *= mul;
    f16x8e h, l;
    split8(a, b, &h, &l);
    const int64_t o = ((grp * c32_dst + cc) << 6) + (i & 63);
    *reinterpret_cast<f16x8e*>(hi + o * 8) = h;
    *reinterpret_cast<f16x8e*>(lo + o * 8) = l;
  }
}

int launch_split_f32_subsample2(const float* in, unsigned short* hi, unsigned short* lo, int N, int H, int W, int ld,
                                hipStream_t s, const float* scale, const float* shift, float mul, int c32_dst) {
  XDET_REQUIRE(ld > 0 && ld % 32 == 0, "split: channel stride must be a multiple of 32");
  if (c32_dst == 0) c32_dst = ld >> 5;
  XDET_REQUIRE(c32_dst >= (ld >> 5), "split: the destination has fewer channel blocks than the source");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t n_pix = (int64_t)N * Ho * Wo;
  if (n_pix == 0) return XDET_OK;
  const int64_t n = cdiv(n_pix, 16) * 16 * ld;
  const int blocks = (int)std::min<int64_t>(cdiv(n / 8, 256), 256 * 32);
  hipLaunchKernelGGL(split_f32_subsample2_kernel, dim3(blocks), dim3(256), 0, s, in, hi, lo, N, H, W, Ho, Wo, ld, scale,
                     shift, mul, c32_dst);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// tf.layers.max_pooling2d(3, 2, 'same') (+ residual).  One thread = 4 channels of one output column, walking
// MP_ROWS output rows downwards: consecutive windows share an input row (2 oy + 1 - pad), whose 3-column maximum is
// carried in registers, so every output costs 6 loads instead of 9 and the shared rows are not fetched again at all
// (one thread per output pixel re-read them through L2, and with more rows in flight than the 4 MB L2 of an XCD holds,
// from HBM: 1.4x the input by FETCH_SIZE).  The 1.5x horizontal overlap stays inside a wave's neighbouring lanes
// (TA / L1).  max is exact, so the result does not depend on the order.
constexpr int MP_ROWS = 8;

__device__ __forceinline__ float4 mp_max4(float4 a, float4 b) {
  // (plain v_max_f32: fmaxf() on a loaded value costs a second, canonicalising v_max_f32)
  float4 r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r.x) : "v"(a.x), "v"(b.x));
  asm("v_max_f32 %0, %1, %2" : "=v"(r.y) : "v"(a.y), "v"(b.y));
  asm("v_max_f32 %0, %1, %2" : "=v"(r.z) : "v"(a.z), "v"(b.z));
  asm("v_max_f32 %0, %1, %2" : "=v"(r.w) : "v"(a.w), "v"(b.w));
  return r;
}

// grid.x = N * ceil(Ho / MP_ROWS) row bands (XCD-aware), grid.y covers Wo * ld/4 items
__global__ __launch_bounds__(256) void maxpool3x3s2_add_kernel(const float* __restrict__ in,
                                                               const float* __restrict__ res,
                                                               float* __restrict__ out, int H, int W, int ld, int Ho,
                                                               int Wo, int pad_t, int pad_l, int nbands, int bands_per_image) {
  const int c4n = ld >> 2;
  const int item = blockIdx.y * 256 + threadIdx.x;
  if (item >= Wo * c4n) return;
  const int ox = item / c4n;
  const int c = (item - ox * c4n) * 4;
  const int per_xcd = gridDim.x >> 3;          // XCD-aware bands (see depthwise3x3_kernel)
  const int band = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (band >= nbands) return;
  const int n = band / bands_per_image;
  const int oy0 = (band - n * bands_per_image) * MP_ROWS;
  const float* base = in + (size_t)n * H * W * ld + c;
  const float4 ninf = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  const int ix0 = ox * 2 - pad_l;
  // 3-column maximum of one input row (-inf outside the image: SAME padding never wins a max)
  auto hmax = [&](int iy) {
    float4 m = ninf;
    if ((unsigned)iy >= (unsigned)H) return m;
    const float* row = base + (size_t)iy * W * ld;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ix0 + kx;
      if ((unsigned)ix < (unsigned)W) m = mp_max4(m, *reinterpret_cast<const float4*>(row + (size_t)ix * ld));
    }
    return m;
  };
  float4 carry = hmax(oy0 * 2 - pad_t);        // first row of the first window
  const int oy1 = min(Ho, oy0 + MP_ROWS);
  for (int oy = oy0; oy < oy1; ++oy) {
    const int iy = oy * 2 - pad_t;
    const float4 a = hmax(iy + 1), b = hmax(iy + 2);
    float4 m = mp_max4(mp_max4(carry, a), b);
    carry = b;                                 // row iy + 2 is the first row of the next window
    const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * ld + c;
    if (res) {
      const float4 r = *reinterpret_cast<const float4*>(res + o);
      m.x += r.x; m.y += r.y; m.z += r.z; m.w += r.w;
    }
    *reinterpret_cast<float4*>(out + o) = m;
  }
}

int launch_maxpool3x3s2_add(const float* in, const float* res, float* out, int N, int H, int W, int C, int ld,
                            int Ho, int Wo, int pad_t, int pad_l, hipStream_t s) {
  XDET_REQUIRE(ld % 4 == 0 && ld >= C, "maxpool: channel stride must be a multiple of 4");
  if ((int64_t)N * Ho == 0) return XDET_OK;
  const int bpi = (int)cdiv(Ho, MP_ROWS), nbands = N * bpi;
  const dim3 grid((unsigned)(cdiv(nbands, 8) * 8), (unsigned)cdiv((int64_t)Wo * (ld / 4), 256));
  hipLaunchKernelGGL(maxpool3x3s2_add_kernel, grid, dim3(256), 0, s, in, res, out, H, W, ld, Ho, Wo, pad_t, pad_l,
                     nbands, bpi);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// Channel blocks of one planes tensor [pix/16][c32_src][16][32] into a wider one [pix/16][c32_dst][16][32] (the pointers address
// the first destination block inside a 16-pixel group), scaled by the power of two r = ratio of the two tensors' activation
// pre-scales (exact for both halves short of f16 overflow / underflow; 1 unless a calibration moved one of them).
__global__ __launch_bounds__(256) void planes_copy_blocks_kernel(const unsigned short* __restrict__ shi, const unsigned short* __restrict__ slo,
                                                                 unsigned short* __restrict__ dhi, unsigned short* __restrict__ dlo,
                                                                 int64_t groups, int c32_src, int c32_dst, float r) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const int64_t n = groups * c32_src * 64;             // 16-byte pieces per plane
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t blk = i >> 6;
    const int64_t grp = blk / c32_src;
    const int cc = (int)(blk - grp * c32_src);
    const int64_t o = ((grp * c32_dst + cc) << 6) + (i & 63);
    h8 h = *reinterpret_cast<const h8*>(shi + i * 8), l = *reinterpret_cast<const h8*>(slo + i * 8);
    if (r != 1.f) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { h[k] = (_Float16)((float)h[k] * r); l[k] = (_Float16)((float)l[k] * r); }
    }
    *reinterpret_cast<h8*>(dhi + o * 8) = h;
    *reinterpret_cast<h8*>(dlo + o * 8) = l;
  }
}

int launch_planes_copy_blocks(const unsigned short* shi, const unsigned short* slo, unsigned short* dhi, unsigned short* dlo,
                              int64_t n_pix, int ld_src, int c32_dst, float r, hipStream_t s) {
  XDET_REQUIRE(shi && slo && dhi && dlo && ld_src > 0 && ld_src % 32 == 0 && c32_dst >= ld_src / 32, "planes copy: bad arguments");
  if (n_pix == 0) return XDET_OK;
  const int64_t groups = cdiv(n_pix, 16), n = groups * (ld_src >> 5) * 64;
  const int blocks = (int)std::min<int64_t>(cdiv(n, 256), 256 * 32);
  hipLaunchKernelGGL(planes_copy_blocks_kernel, dim3(blocks), dim3(256), 0, s, shi, slo, dhi, dlo, groups, ld_src >> 5, c32_dst, r);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// ResNet v2 stem tail (net/resnet_v2.py:311-330 initial_max_pool, then the first block's batch_norm_relu :142-156): the pooled
// tensor has ONE reader, the first block's pre-activation, and that one is read as split planes only (its shortcut is a
// projection of the pre-activation).  Same window walk as maxpool3x3s2_add_kernel; the result goes through bn + ReLU and
// out as planes [pix/16][ld/32][16][32] (the arithmetic of bn_relu_kernel in net.hip: fma, NaN-keeping ReLU, * mul, hi / lo)
// -- the pooled f32 tensor (29.5 MB per batch of 8) is neither written nor read back.
__global__ __launch_bounds__(256) void maxpool3x3s2_bn_planes_kernel(const float* __restrict__ in, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                                                     int H, int W, int ld, int Ho, int Wo, int pad_t, int pad_l,
                                                                     int nbands, int bands_per_image, float mul,
                                                                     unsigned short* __restrict__ hi2, unsigned short* __restrict__ lo2,
                                                                     int c32_2, float mul2) {
  // hi2 / lo2 (optional): a second copy, scaled by mul2, into channel blocks of a wider concatenated operand
  // [pix/16][c32_2][16][32] (the pointers address this tensor's first block in it)
  const int c4n = ld >> 2;
  const int item = blockIdx.y * 256 + threadIdx.x;
  if (item >= Wo * c4n) return;
  const int ox = item / c4n;
  const int c = (item - ox * c4n) * 4;
  const int per_xcd = gridDim.x >> 3;
  const int band = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (band >= nbands) return;
  const int n = band / bands_per_image;
  const int oy0 = (band - n * bands_per_image) * MP_ROWS;
  const float* base = in + (size_t)n * H * W * ld + c;
  const float4 ninf = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  const int ix0 = ox * 2 - pad_l;
  auto hmax = [&](int iy) {
    float4 m = ninf;
    if ((unsigned)iy >= (unsigned)H) return m;
    const float* row = base + (size_t)iy * W * ld;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ix0 + kx;
      if ((unsigned)ix < (unsigned)W) m = mp_max4(m, *reinterpret_cast<const float4*>(row + (size_t)ix * ld));
    }
    return m;
  };
  const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
  float4 carry = hmax(oy0 * 2 - pad_t);
  const int oy1 = min(Ho, oy0 + MP_ROWS);
  for (int oy = oy0; oy < oy1; ++oy) {
    const int iy = oy * 2 - pad_t;
    const float4 a = hmax(iy + 1), b = hmax(iy + 2);
    const float4 m = mp_max4(mp_max4(carry, a), b);
    carry = b;
    float v[4] = {fmaf(m.x, sc.x, sh.x), fmaf(m.y, sc.y, sh.y), fmaf(m.z, sc.z, sh.z), fmaf(m.w, sc.w, sh.w)};
    const float m4[4] = {v[0], v[1], v[2], v[3]};
    _Float16 h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = !(v[k] <= 0.f) ? v[k] : 0.f;
      v[k] *= mul;
      h[k] = (_Float16)v[k];
      l[k] = (_Float16)(v[k] - (float)h[k]);
    }
    const int64_t pix = ((int64_t)n * Ho + oy) * Wo + ox;
    const size_t po = ((size_t)((pix >> 4) * (ld >> 5) + (c >> 5)) << 9) + ((size_t)(pix & 15) << 5) + (size_t)(c & 31);
    *reinterpret_cast<uint2*>(hi + po) = *reinterpret_cast<const uint2*>(h);
    *reinterpret_cast<uint2*>(lo + po) = *reinterpret_cast<const uint2*>(l);
    if (hi2) {
      if (mul2 != mul) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float u = (!(m4[k] <= 0.f) ? m4[k] : 0.f) * mul2;
          h[k] = (_Float16)u;
          l[k] = (_Float16)(u - (float)h[k]);
        }
      }
      const size_t p2 = ((size_t)((pix >> 4) * c32_2 + (c >> 5)) << 9) + ((size_t)(pix & 15) << 5) + (size_t)(c & 31);
      *reinterpret_cast<uint2*>(hi2 + p2) = *reinterpret_cast<const uint2*>(h);
      *reinterpret_cast<uint2*>(lo2 + p2) = *reinterpret_cast<const uint2*>(l);
    }
  }
}

int launch_maxpool3x3s2_bn_planes(const float* in, const float* scale, const float* shift, unsigned short* hi, unsigned short* lo,
                                  int N, int H, int W, int C, int ld, int Ho, int Wo, int pad_t, int pad_l, float mul,
                                  hipStream_t s, unsigned short* hi2, unsigned short* lo2, int c32_2, float mul2) {
  XDET_REQUIRE(ld % 32 == 0 && ld >= C && in && scale && shift && hi && lo, "maxpool + bn planes: channel stride must be a multiple of 32");
  XDET_REQUIRE(!hi2 || (lo2 && c32_2 >= ld / 32), "maxpool + bn planes: bad second destination");
  if ((int64_t)N * Ho == 0) return XDET_OK;
  const int bpi = (int)cdiv(Ho, MP_ROWS), nbands = N * bpi;
  const dim3 grid((unsigned)(cdiv(nbands, 8) * 8), (unsigned)cdiv((int64_t)Wo * (ld / 4), 256));
  hipLaunchKernelGGL(maxpool3x3s2_bn_planes_kernel, grid, dim3(256), 0, s, in, scale, shift, hi, lo, H, W, ld, Ho, Wo, pad_t, pad_l,
                     nbands, bpi, mul, hi2, lo2, c32_2, mul2);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// The vertical half of max_pooling2d(3, 2, 'same') (+ residual) over a tensor whose rows were already pooled
// horizontally by the producing kernel (sepconv_fused.hip, HPOOL): in [N][H][Wo][ld] -> out [N][Ho][Wo][ld].
// Same walk as maxpool3x3s2_add_kernel: one thread = 4 channels of a column, the shared row carried in a register.
// Round 5: the pooled sum x of an entry-flow block has two readers -- the next block's 1x1 / stride-2 projection (raw x, every
// second pixel of every second row) and its first separable conv (relu(x), net/xception_body.py:261-277).  With SUB the pass
// also writes what they read: the split planes of the raw subsampled tensor (what split_f32_subsample2_kernel made in a pass
// of its own: hi = f16(x mul), lo = f16(x mul - hi), blocked [pix/16][ld/32][16][32]) and the f32 tensor as relu(x), which
// takes the 72 v_max per chunk out of the fused separable kernel's window reads.  Same values on both paths.
template <bool SUB>
__global__ __launch_bounds__(256) void maxpool_v3s2_add_kernel(const float* __restrict__ in, const float* __restrict__ res,
                                                               float* __restrict__ out, int H, int Wo, int ld, int Ho,
                                                               int pad_t, int nbands, int bands_per_image,
                                                               unsigned short* __restrict__ sub_hi,
                                                               unsigned short* __restrict__ sub_lo, float sub_mul, int Hs, int Ws) {
  const int c4n = ld >> 2;
  const int item = blockIdx.y * 256 + threadIdx.x;
  if (item >= Wo * c4n) return;
  const int per_xcd = gridDim.x >> 3;
  const int band = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (band >= nbands) return;
  const int n = band / bands_per_image;
  const int oy0 = (band - n * bands_per_image) * MP_ROWS;
  const float* base = in + (size_t)n * H * Wo * ld + (size_t)item * 4;      // item = ox * c4n + c4
  // (by value: `cond ? *ptr : ninf` is an lvalue conditional -- it took ninf's address and put it in scratch memory)
  //  and the compiler turns `if (ok) v = *ptr` back into a load through a selected pointer: load a valid row always,
  //  select on the VALUES)
  auto rowv = [&](int iy) {
    const bool ok = (unsigned)iy < (unsigned)H;
    float4 v = *reinterpret_cast<const float4*>(base + (size_t)(ok ? iy : 0) * Wo * ld);
    v.x = ok ? v.x : -INFINITY; v.y = ok ? v.y : -INFINITY; v.z = ok ? v.z : -INFINITY; v.w = ok ? v.w : -INFINITY;
    return v;
  };
  float4 carry = rowv(oy0 * 2 - pad_t);
  const int oy1 = min(Ho, oy0 + MP_ROWS);
  for (int oy = oy0; oy < oy1; ++oy) {
    const int iy = oy * 2 - pad_t;
    const float4 a = rowv(iy + 1), b = rowv(iy + 2);
    float4 m = mp_max4(mp_max4(carry, a), b);
    carry = b;
    const size_t o = ((size_t)n * Ho + oy) * Wo * ld + (size_t)item * 4;
    if (res) {
      const float4 r = *reinterpret_cast<const float4*>(res + o);
      m.x += r.x; m.y += r.y; m.z += r.z; m.w += r.w;
    }
    if (SUB) {
      const int ox = item / c4n, c = (item - ox * c4n) * 4;
      if (!((oy | ox) & 1)) {
        const int64_t pix = ((int64_t)n * Hs + (oy >> 1)) * Ws + (ox >> 1);
        const size_t po = ((size_t)((pix >> 4) * (ld >> 5) + (c >> 5)) << 9) + ((size_t)(pix & 15) << 5) + (size_t)(c & 31);
        const float v[4] = {m.x * sub_mul, m.y * sub_mul, m.z * sub_mul, m.w * sub_mul};
        _Float16 h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          h[k] = (_Float16)v[k];
          l[k] = (_Float16)(v[k] - (float)h[k]);
        }
        *reinterpret_cast<uint2*>(sub_hi + po) = *reinterpret_cast<const uint2*>(h);
        *reinterpret_cast<uint2*>(sub_lo + po) = *reinterpret_cast<const uint2*>(l);
      }
      // (the NaN-keeping ReLU of the other kernels: fmaxf would turn a non-finite block output into 0 at the pixels that are
      //  not subsampled and hide it from the range check)
      m.x = !(m.x <= 0.f) ? m.x : 0.f; m.y = !(m.y <= 0.f) ? m.y : 0.f; m.z = !(m.z <= 0.f) ? m.z : 0.f; m.w = !(m.w <= 0.f) ? m.w : 0.f;
    }
    *reinterpret_cast<float4*>(out + o) = m;
  }
}

// sub_hi != NULL: also write the raw subsampled planes (x * sub_mul; ld a multiple of 32) and store relu(x) -- see the kernel
int launch_maxpool_v3s2_add(const float* in_hpooled, const float* res, float* out, int N, int H, int Wo, int C, int ld,
                            int Ho, int pad_t, hipStream_t s, unsigned short* sub_hi, unsigned short* sub_lo, float sub_mul) {
  XDET_REQUIRE(ld % 4 == 0 && ld >= C, "maxpool: channel stride must be a multiple of 4");
  XDET_REQUIRE(!sub_hi || (sub_lo && ld % 32 == 0), "maxpool: the subsampled planes need both planes and a channel stride that is a multiple of 32");
  if ((int64_t)N * Ho == 0) return XDET_OK;
  const int bpi = (int)cdiv(Ho, MP_ROWS), nbands = N * bpi;
  const dim3 grid((unsigned)(cdiv(nbands, 8) * 8), (unsigned)cdiv((int64_t)Wo * (ld / 4), 256));
  const int Hs = (Ho + 1) / 2, Ws = (Wo + 1) / 2;
  if (sub_hi)
    hipLaunchKernelGGL(maxpool_v3s2_add_kernel<true>, grid, dim3(256), 0, s, in_hpooled, res, out, H, Wo, ld, Ho, pad_t, nbands,
                       bpi, sub_hi, sub_lo, sub_mul, Hs, Ws);
  else
    hipLaunchKernelGGL(maxpool_v3s2_add_kernel<false>, grid, dim3(256), 0, s, in_hpooled, res, out, H, Wo, ld, Ho, pad_t, nbands,
                       bpi, nullptr, nullptr, 1.f, Hs, Ws);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// Diagnostic pass of the "check_range" net option: |x| <= limit (65504, the largest f16) for every element of an f32
// activation tensor, else bad[image] = 1.  The split-precision convs turn a larger activation into hi = inf, and the
// NaN that follows does not survive the next ReLU (max(NaN, 0) = 0): without this pass an overflow ends as a silently
// empty detection list.  BN-normalised checkpoints stay orders of magnitude inside the range; the pass is for
// validating a new checkpoint once (it reads every activation again: ~+30 % time).
// group4 > 0: the tensor is a stack of groups of group4 float4s (the frequency bins of the spectral large-separable
// convs, rows [bin][n*F + o]); the image of an element is its index WITHIN the group / per_image4, rows past the
// batch are padding
__global__ __launch_bounds__(256) void range_check_kernel(const float4* __restrict__ x, int64_t n4, int64_t per_image4,
                                                          float limit, int* __restrict__ bad, int64_t group4, int N, int relu) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = x[i];
    // relu: the consumer reads max(x, 0) -- a large NEGATIVE value is harmless, only +values count against the limit
    const float m = relu ? fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w))
                         : fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    const float am = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    const bool nan = v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w || !(am <= 3.4028235e38f);   // NaN or +-inf
    if (nan || !(m <= limit)) {
      const int64_t img = (group4 ? i % group4 : i) / per_image4;
      if (img < N) bad[img] = 1;
    }
  }
}

// ... and the same for a tensor that only exists as split planes [pix/16][ld/32][16][32] (stem, depthwise, conv
// epilogues that feed the LDS-DMA kernel): hi = inf / NaN is how an element beyond the f16 range looks there.
__global__ __launch_bounds__(256) void range_check_planes_kernel(const uint4* __restrict__ hi, int64_t n8, int c32n,
                                                                 int64_t pix_per_image, int64_t n_pix,
                                                                 int* __restrict__ bad, int64_t group_pix, int N) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 v = hi[i];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    bool b = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) b |= (w[k] & 0x7c00u) == 0x7c00u || (w[k] & 0x7c000000u) == 0x7c000000u;
    if (b) {
      const int64_t pix = (i >> 6) / c32n * 16 + ((i >> 2) & 15);      // 64 chunks of 8 halves per 1 KB block
      const int64_t img = (group_pix ? pix % group_pix : pix) / pix_per_image;
      if (pix < n_pix && img < N) bad[img] = 1;
    }
  }
}

int launch_range_check_planes(const unsigned short* hi, int N, int64_t pix_per_image, int ld, int* bad_per_image,
                              hipStream_t s, int groups, int64_t group_pix) {
  const int64_t n_pix = groups > 1 ? (int64_t)groups * group_pix : (int64_t)N * pix_per_image;
  const int64_t n8 = cdiv(n_pix, 16) * 16 * ld / 8;
  if (n8 == 0) return XDET_OK;
  const int blocks = (int)std::min<int64_t>(cdiv(n8, 256), 256 * 16);
  hipLaunchKernelGGL(range_check_planes_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(hi), n8, ld >> 5,
                     pix_per_image, n_pix, bad_per_image, groups > 1 ? group_pix : (int64_t)0, N);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

int launch_range_check(const float* x, int N, size_t per_image, float limit, int* bad_per_image, hipStream_t s,
                       int groups, size_t group_elems, int relu) {
  XDET_REQUIRE(per_image % 4 == 0 && group_elems % 4 == 0, "range_check: tensor size must be a multiple of 4");
  const int64_t n4 = groups > 1 ? (int64_t)groups * (int64_t)(group_elems / 4) : (int64_t)N * (int64_t)(per_image / 4);
  if (n4 == 0) return XDET_OK;
  const int blocks = (int)std::min<int64_t>(cdiv(n4, 256), 256 * 16);
  hipLaunchKernelGGL(range_check_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(x), n4,
                     (int64_t)(per_image / 4), limit, bad_per_image, groups > 1 ? (int64_t)(group_elems / 4) : (int64_t)0, N, relu);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// ---- range calibration (activation pre-scale of the split-precision planes, net.hip calibrate()) ----
// largest |hi| of a split plane as f16 bits (inf / NaN sort above every finite value: bits >= 0x7c00)
__global__ __launch_bounds__(256) void absmax_planes_kernel(const uint4* __restrict__ hi, int64_t n8, unsigned* __restrict__ out) {
  unsigned m = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 v = hi[i];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) m = max(m, max(w[k] & 0x7fffu, (w[k] >> 16) & 0x7fffu));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
// largest |x| (relu: largest max(x, 0)) of an f32 tensor as f32 bits (NaN / inf: bits >= 0x7f800000)
__global__ __launch_bounds__(256) void absmax_f32_kernel(const float4* __restrict__ x, int64_t n4, int relu, unsigned* __restrict__ out) {
  unsigned m = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = x[i];
    const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned b = __float_as_uint(f[k]);
      if (relu && (b >> 31) && (b & 0x7fffffffu) <= 0x7f800000u) continue;    // negative (not NaN): ReLU maps it to 0
      m = max(m, b & 0x7fffffffu);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
int launch_absmax_planes(const unsigned short* hi, int64_t n_halves, unsigned* out, hipStream_t s) {
  const int64_t n8 = n_halves / 8;
  if (n8 <= 0) return XDET_OK;
  const int blocks = (int)std::min<int64_t>(cdiv(n8, 256), 256 * 8);
  hipLaunchKernelGGL(absmax_planes_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(hi), n8, out);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}
int launch_absmax_f32(const float* x, int64_t n, int relu, unsigned* out, hipStream_t s) {
  const int64_t n4 = n / 4;
  if (n4 <= 0) return XDET_OK;
  const int blocks = (int)std::min<int64_t>(cdiv(n4, 256), 256 * 8);
  hipLaunchKernelGGL(absmax_f32_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(x), n4, relu, out);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

__global__ void relu_copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = in[i];
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    out[i] = v;
  }
}

int launch_relu_copy(const float* in, float* out, int64_t n, hipStream_t s) {
  XDET_REQUIRE(n % 4 == 0, "relu_copy: length must be a multiple of 4");
  const int blocks = (int)std::min<int64_t>(cdiv(n / 4, 256), 256 * 16);
  hipLaunchKernelGGL(relu_copy_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(in),
                     reinterpret_cast<float4*>(out), n / 4);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}


// ---------------------------------------------------------------------------------------
// Stem conv of the Xception entry (block1_conv1, net/xception_body.py:243-250): 3x3 / stride 2 / VALID, 3 -> 32
// channels, inference BN + ReLU, straight from the NCHW network input to the split f16 planes its only consumer
// (block1_conv2 on the LDS-DMA path) reads.  K = 27: no matrix-core shape fits, and the generic small-cin MFMA
// kernel spent its time gathering 16-byte NHWC4 pixels (16 TFLOP/s, 0.38 ms per 64 images) behind a separate
// NCHW -> NHWC4 pass.  Here one thread owns one output pixel and all 32 channels: 27 coalesced loads (lanes =
// consecutive output columns, stride-2 input columns), 864 f32 FMAs against weights that are wave-uniform and come
// through the scalar cache, exact f32 accumulation in (ky, kx, ci) order, then hi/lo f16: each lane writes its
// pixel's 64 contiguous bytes per plane.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_conv3x3s2_kernel(const float* __restrict__ in_nchw,
                                                             const float* __restrict__ w /*[27][32], k = (ky*3+kx)*3+ci*/,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             unsigned short* __restrict__ hi, unsigned short* __restrict__ lo,
                                                             int N, int S, int Ho, int Wo) {
  const int64_t total = (int64_t)N * Ho * Wo;
  const int64_t pix = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (pix >= total) return;
  const int n = (int)(pix / ((int64_t)Ho * Wo));
  const int rem = (int)(pix - (int64_t)n * Ho * Wo);
  const int oy = rem / Wo, ox = rem - oy * Wo;
  const float* base = in_nchw + (size_t)n * 3 * S * S + (size_t)(oy * 2) * S + ox * 2;
  float acc[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) acc[c] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float x = base[(size_t)ci * S * S + ky * S + kx];
        const float* wk = w + ((ky * 3 + kx) * 3 + ci) * 32;
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[c] = fmaf(x, wk[c], acc[c]);
      }
  const size_t o = ((size_t)(pix >> 4) << 9) + ((pix & 15) << 5);          // [pix/16][1][16][32]
#pragma unroll
  for (int c8 = 0; c8 < 4; ++c8) {
    _Float16 h[8], l[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = c8 * 8 + k;
      const float v = fmaxf(fmaf(acc[c], scale[c], shift[c]), 0.f);
      h[k] = (_Float16)v;
      l[k] = (_Float16)(v - (float)h[k]);
    }
    *reinterpret_cast<uint4*>(hi + o + c8 * 8) = *reinterpret_cast<uint4*>(h);
    *reinterpret_cast<uint4*>(lo + o + c8 * 8) = *reinterpret_cast<uint4*>(l);
  }
}

int launch_stem_conv3x3s2(const float* in_nchw, const float* w27x32, const float* scale, const float* shift,
                          unsigned short* hi, unsigned short* lo, int N, int S, hipStream_t s) {
  const int Ho = (S - 3) / 2 + 1;
  const int64_t total = (int64_t)N * Ho * Ho;
  if (total == 0) return XDET_OK;
  hipLaunchKernelGGL(stem_conv3x3s2_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, in_nchw, w27x32, scale, shift,
                     hi, lo, N, S, Ho, Ho);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

}  // namespace xdet
