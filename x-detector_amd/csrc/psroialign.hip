// PsRoiAlign forward for gfx950 -- replaces the reference's TF custom op
// (cpp/PSROIPooling/ps_roi_align_op.cc:81-201 CPU functor, ps_roi_align_op.cu:36-132 CUDA kernel).
//
// THIS FILE IS COMPILED WITH -ffp-contract=off: the reference arithmetic is a fixed sequence
// of separately rounded f32 operations plus an f64 bilinear blend (the `1.` literals in
// ps_roi_align_op.cc:171-174), and bin assignment / argmax must be index-exact.
//
// Work decomposition (not the reference's one-thread-per-element grid-stride loop): one
// 64-lane wavefront per (image, roi); the lanes walk the ROI's C = gh*gw*bank output
// elements, whose flat index equals the input channel (pos*bank + ch), so with the NHWC
// feature map used by the fused pipeline a lane group reads `bank` consecutive channels of
// the same pixel (40 contiguous bytes for bank = 10) and the ROI geometry (identical for
// all elements of an ROI) is computed once per wave from scalar loads.  layout 0 = NCHW (the op's public contract), 1 = NHWC with channel stride ldc.
#include "common.h"
#include <cfloat>
#include <cstdlib>
#include <cstring>

namespace xdet {

__device__ __forceinline__ float feat_at(const float* __restrict__ f, int layout, int ldc, int H, int W, int64_t n,
                                         int c, int y, int x) {
  if (layout == 0) return f[((n * ldc + c) * H + y) * W + x];
  return f[((n * H + y) * W + x) * ldc + c];
}

// Corner de-duplication for small bins (two-channel form).  A bin's n_h x n_w samples are less than a pixel apart
// (step = bin / (int(bin) + 1) < 1), so their 4 n_h n_w bilinear corners fall on at most (n_h+1) x (n_w+1) distinct
// pixels starting at the first sample's (iy, ix): 9 instead of 16 gathers for 2 x 2 samples, 16 instead of 36 for 3 x 3
// -- and the gathers' address path is what bounds the kernel.  The lane loads that pixel grid once (8 bytes each) into its
// own column of a small LDS array ([entry][lane]: conflict-free, no cross-lane traffic, no barrier) and takes every
// sample's four corners from there at a per-lane index (registers cannot be indexed per lane; select chains measured
// slower in round 2).  The clamped neighbours min(iy + 1, H - 1) / min(ix + 1, W - 1) are the grid's own clamped rows /
// columns.  Values, order and typing of the blend are the kernel's: bit-exact.  Returns false (nothing accumulated) if a
// sample's pixel does not lie within `its index` of the first one -- cannot happen for step < 1 in exact arithmetic; the
// caller then takes the direct path.
template <int NH, int NW, bool USE_MAX>
__device__ __forceinline__ bool psroi_grid_bin(const float* __restrict__ fimg, float2* __restrict__ grid, int H, int W, int sy,
                                               int sx, int coff, float x0, float y0, float step_w, float step_h, double half_w,
                                               double half_h, float (&acc)[2], int (&arg)[2]) {
  constexpr bool use_max = USE_MAX;
  int bx[NW], by[NH];
  float fx[NW], fy[NH];
  int ix0 = 0, iy0 = 0;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const float x = (float)((double)(x0 + step_w * (float)j) + half_w);
    const int ix = (int)x;
    fx[j] = x - (float)ix;
    if (j == 0) ix0 = ix;
    bx[j] = ix - ix0;
    ok = ok && (unsigned)bx[j] <= (unsigned)j;
  }
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const float y = (float)((double)(y0 + step_h * (float)i) + half_h);
    const int iy = (int)y;
    fy[i] = y - (float)iy;
    if (i == 0) iy0 = iy;
    by[i] = iy - iy0;
    ok = ok && (unsigned)by[i] <= (unsigned)i;
  }
  if (!ok) return false;
#pragma unroll
  for (int a = 0; a <= NH; ++a) {
    const int ro = min(iy0 + a, H - 1) * sy + coff;
#pragma unroll
    for (int b = 0; b <= NW; ++b)
      grid[(a * (NW + 1) + b) * 64] = *reinterpret_cast<const float2*>(fimg + ro + min(ix0 + b, W - 1) * sx);
    if (NH * NW > 4) __builtin_amdgcn_sched_barrier(0);      // one grid row in flight at a time (registers)
  }
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const double wy0 = 1. - fy[i], wy1 = fy[i];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const float2* g = grid + (by[i] * (NW + 1) + bx[j]) * 64;
      const float2 f00 = g[0], f01 = g[64], f10 = g[(NW + 1) * 64], f11 = g[(NW + 2) * 64];
      const double wx0 = 1. - fx[j], wx1 = fx[j];
      const double w00 = wx0 * wy0, w10 = wx0 * wy1, w01 = wx1 * wy0;
      const float fxfy = fx[j] * fy[i];
      const float a00[2] = {f00.x, f00.y}, a10[2] = {f10.x, f10.y}, a01[2] = {f01.x, f01.y}, a11[2] = {f11.x, f11.y};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const double v = w00 * a00[u] + w10 * a10[u] + w01 * a01[u] + fxfy * a11[u];
        const float t = (float)v;
        if (use_max) {
          if (acc[u] < t) { acc[u] = t; arg[u] = NW * i + j; }
        } else {
          acc[u] += t;
        }
      }
      if (NH * NW > 6) __builtin_amdgcn_sched_barrier(0);     // larger grids: one sample at a time (registers)
    }
    if (NH * NW > 4) __builtin_amdgcn_sched_barrier(0);
  }
  return true;
}

// VEC = channels per lane: 1 (any layout), or 2 for the NHWC form with an even bank -- a lane then owns two neighbouring
// channels of one bin and takes each bilinear corner as ONE 8-byte load.  The kernel is bound by the address path of its
// gathers (four per sample and element, every wave instruction touching ~7 different lines), not by the f64 blend: halving
// the load instructions is what pays (a sample-table variant with 2.3x fewer vector instructions was SLOWER).
template <int VEC, bool USE_MAX>
__global__ __launch_bounds__(256, 4) void psroialign_fwd_kernel(const float* __restrict__ feat,
                                                             const float* __restrict__ rois,
                                                             float* __restrict__ pooled, int32_t* __restrict__ index,
                                                             int N, int C, int H, int W, int R, int gw, int gh,
                                                             int layout, int ldc, int out_ld, int corners, int dedup, int split) {
  constexpr int use_max = USE_MAX ? 1 : 0;
  // pixel grids of psroi_grid_bin: [wave][entry <= 16][lane] (two-channel form only)
  __shared__ __attribute__((aligned(8))) float2 s_grid[VEC == 2 ? 4 * 16 * 64 : 1];
  const int bank = C / (gw * gh);
  const int lane = threadIdx.x & 63;
  // XCD-aware ROI order.  Workgroups are dealt round-robin to the 8 XCDs (private L2s); with ROIs numbered
  // straight through, every XCD ended up sampling every image's feature map (7x the map bytes over the fabric).
  // Here image n belongs to XCD n & 7: that XCD's workgroups walk the image's ROI blocks back to back, so a map
  // (1.8 MB at 30x30x490) crosses the fabric once and every further sample of it is an L2 hit.
  // Fewer than 8 images: an image's ROI blocks are dealt to P = 8 / N XCDs (the map is read P times over the fabric, and
  // all 256 CUs work -- with one XCD per image a single image's 1000 ROIs ran on 32 CUs: 90 us on the critical path).
  // split (few ROIs in the whole call): ONE ROI per workgroup, its elements dealt to the four waves -- a single image's 300
  // ROIs are then 1,200 waves of one pass each instead of 300 waves of four dependent passes (latency, not throughput)
  const int bpi = split ? R : (R + 3) >> 2;            // workgroups (4 ROIs each; split: one) per image
  const int e_first = (split ? (int)(threadIdx.x >> 6) * 64 : 0) + lane, e_step = split ? 256 : 64;
  const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  int64_t n;
  int rblk;
  if (N >= 8) {
    n = (int64_t)(slot / bpi) * 8 + xcd;
    rblk = slot % bpi;
  } else {
    const int P = 8 / N;
    n = xcd % N;
    rblk = slot * P + xcd / N;
    if (xcd >= N * P) return;
  }
  const int r_roi = split ? rblk : rblk * 4 + (threadIdx.x >> 6);
  if (n >= N || rblk >= bpi || r_roi >= R) return;
  const int64_t nr = n * R + r_roi;
  const float* roi = rois + nr * 4;
  float r0 = roi[0], r1 = roi[1], r2 = roi[2], r3 = roi[3];
  if (corners) {   // _point2center, net/xception_body.py:215-218
    const float hh = r2 - r0, ww = r3 - r1;
    r0 = r0 + hh / 2.f;
    r1 = r1 + ww / 2.f;
    r2 = hh;
    r3 = ww;
  }
  float* prow = pooled + nr * out_ld;
  int32_t* irow = index ? index + nr * out_ld : nullptr;

  if (r2 < FLT_MIN || r3 < FLT_MIN) {   // degenerate ROI: zero (reference leaves the index unwritten)
    for (int e = e_first; e < C; e += e_step) {
      prow[e] = 0.f;
      if (irow) irow[e] = 0;
    }
    return;
  }
  const float yc = r0 * (float)H;
  const float xc = r1 * (float)W;
  const float rh = fmaxf(r2 * (float)H, 1.f);
  const float rw = fmaxf(r3 * (float)W, 1.f);
  const float ymin = fmaxf(yc - rh / 2.f, 0.f);
  const float xmin = fmaxf(xc - rw / 2.f, 0.f);
  const float ymax = fminf(yc + rh / 2.f, (float)H - FLT_MIN);
  const float xmax = fminf(xc + rw / 2.f, (float)W - FLT_MIN);
  const float bin_w = (xmax - xmin) / (float)gw;
  const float bin_h = (ymax - ymin) / (float)gh;
  const int n_w = (int)bin_w + 1;
  const int n_h = (int)bin_h + 1;
  const float step_w = bin_w / (float)n_w;
  const float step_h = bin_h / (float)n_h;
  const double half_w = (double)step_w / 2.;
  const double half_h = (double)step_h / 2.;

  // Direct path (one gather per bilinear corner).  Everything that does not depend on the sample row is hoisted: 32-bit
  // element offsets from a per-image base pointer, and the x geometry of a bin's <= JMAX sample columns (integer columns,
  // fractions and the double-precision weight 1.-fx) once per element instead of once per sample.  (Round 2 read the
  // kernel as VALU-bound; round 3 showed the gathers' address path is the bound: fewer instructions did not help, fewer
  // and wider gathers did -- VEC = 2 and psroi_grid_bin above.)  Every
  // expression keeps the reference's typing and order: (1.-fx)*(1.-fy)*f00 + (1.-fx)*fy*f10 + fx*(1.-fy)*f01 in
  // double, fx*fy*f11 in float, summed left to right.
  constexpr int JMAX = (VEC == 2 && !USE_MAX) ? 6 : 8;   // ('mean' with two channels: 8 hoisted columns spill 2 registers at four waves)
  const int sy = layout == 0 ? W : W * ldc, sx = layout == 0 ? 1 : ldc, sc = layout == 0 ? H * W : 1;
  const float* __restrict__ fimg = feat + (size_t)n * H * W * ldc;       // NCHW: ldc == C
  typedef float vecf __attribute__((ext_vector_type(VEC)));
  auto ldv = [&](int off) {
    vecf r;
    if (VEC == 1) r[0] = fimg[off];
    else r = *reinterpret_cast<const vecf*>(fimg + off);                   // 8-byte aligned: even offset (launch check)
    return r;
  };
  // one element (VEC channels of one bin) by direct corner loads: the general path
  auto element_direct = [&](int e, float (&acc)[VEC], int (&arg)[VEC]) {
    const int pos = e / bank;
    const int row = pos / gw;
    const int col = pos - row * gw;
    const int c_in = e;                  // pos*bank + ch
    const float x0 = xmin + bin_w * (float)col;
    const float y0 = ymin + bin_h * (float)row;
    auto blend = [&](double w00, double w10, double w01, float fxfy, vecf f00, vecf f10, vecf f01, vecf f11, int sidx) {
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        const double v = w00 * f00[u] + w10 * f10[u] + w01 * f01[u] + fxfy * f11[u];
        const float t = (float)v;
        if (use_max) {
          if (acc[u] < t) { acc[u] = t; arg[u] = sidx; }
        } else {
          acc[u] += t;
        }
      }
    };
    if (VEC == 1 && layout == 0 && n_w <= JMAX) {
      // NCHW, the op's public layout: the two corners of a sample row are NEIGHBOURS in memory (x, x + 1), so one 8-byte
      // load (dword-aligned is enough for global memory) fetches both -- half the gathers, which are what bounds this
      // kernel; the last column (x + 1 clamped to W - 1, i.e. the same pixel twice) takes a 4-byte load.  Values, typing
      // and order of the blend are untouched.
      const int coff = c_in * sc;
      int xo0[JMAX];
      bool pair[JMAX];
      float fxs[JMAX];
      double wx0[JMAX];
#pragma unroll
      for (int j = 0; j < JMAX; ++j) {
        if (j < n_w) {
          const float x = (float)((double)(x0 + step_w * (float)j) + half_w);
          const int ix = (int)x;
          const float fx = x - (float)ix;
          xo0[j] = ix + coff;
          pair[j] = ix + 1 <= W - 1;
          fxs[j] = fx;
          wx0[j] = 1. - fx;
        }
      }
      typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
      for (int i = 0; i < n_h; ++i) {
        const float y = (float)((double)(y0 + step_h * (float)i) + half_h);
        const int iy = (int)y;
        const float fy = y - (float)iy;
        const int yo0 = iy * sy, yo1 = min(iy + 1, H - 1) * sy;
        const double wy0 = 1. - fy, wy1 = fy;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
          if (j < n_w) {
            vecf f00, f10, f01, f11;
            if (pair[j]) {
              const f2u a = *reinterpret_cast<const f2u*>(fimg + yo0 + xo0[j]), b = *reinterpret_cast<const f2u*>(fimg + yo1 + xo0[j]);
              f00[0] = a[0]; f01[0] = a[1]; f10[0] = b[0]; f11[0] = b[1];
            } else {
              f00[0] = f01[0] = fimg[yo0 + xo0[j]];
              f10[0] = f11[0] = fimg[yo1 + xo0[j]];
            }
            blend(wx0[j] * wy0, wx0[j] * wy1, (double)fxs[j] * wy0, fxs[j] * fy, f00, f10, f01, f11, n_w * i + j);
          }
        }
      }
    } else if (n_w <= JMAX) {
      const int coff = c_in * sc;
      int xo0[JMAX], xo1[JMAX];
      float fxs[JMAX];
      double wx0[JMAX];                  // (fx as a double is one conversion at the use: 16 registers less)
#pragma unroll
      for (int j = 0; j < JMAX; ++j) {
        if (j < n_w) {                   // wave-uniform
          const float x = (float)((double)(x0 + step_w * (float)j) + half_w);
          const int ix = (int)x;
          const float fx = x - (float)ix;
          xo0[j] = ix * sx + coff;
          xo1[j] = min(ix + 1, W - 1) * sx + coff;
          fxs[j] = fx;
          wx0[j] = 1. - fx;
        }
      }
      for (int i = 0; i < n_h; ++i) {
        const float y = (float)((double)(y0 + step_h * (float)i) + half_h);
        const int iy = (int)y;
        const float fy = y - (float)iy;
        const int yo0 = iy * sy, yo1 = min(iy + 1, H - 1) * sy;
        const double wy0 = 1. - fy, wy1 = fy;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
          if (j < n_w) {
            const vecf f00 = ldv(yo0 + xo0[j]), f10 = ldv(yo1 + xo0[j]), f01 = ldv(yo0 + xo1[j]), f11 = ldv(yo1 + xo1[j]);
            // (1.-fx)*(1.-fy)*f00 + (1.-fx)*fy*f10 + fx*(1.-fy)*f01 in double, fx*fy*f11 in float, summed left to right
            blend(wx0[j] * wy0, wx0[j] * wy1, (double)fxs[j] * wy0, fxs[j] * fy, f00, f10, f01, f11, n_w * i + j);
          }
        }
      }
    } else {
      for (int i = 0; i < n_h; ++i) {
        const float y = (float)((double)(y0 + step_h * (float)i) + half_h);
        const int iy = (int)y;
        const float fy = y - (float)iy;
        const int iy1 = min(iy + 1, H - 1);
        for (int j = 0; j < n_w; ++j) {
          const float x = (float)((double)(x0 + step_w * (float)j) + half_w);
          const int ix = (int)x;
          const float fx = x - (float)ix;
          const int ix1 = min(ix + 1, W - 1);
          const int coff = c_in * sc;
          const vecf f00 = ldv(iy * sy + ix * sx + coff), f10 = ldv(iy1 * sy + ix * sx + coff);
          const vecf f01 = ldv(iy * sy + ix1 * sx + coff), f11 = ldv(iy1 * sy + ix1 * sx + coff);
          blend((1. - fx) * (1. - fy), (1. - fx) * fy, fx * (1. - fy), fx * fy, f00, f10, f01, f11, n_w * i + j);
        }
      }
    }
  };
  auto finish = [&](int e, const float (&acc)[VEC], const int (&arg)[VEC]) {
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      float a = acc[u];
      if (!use_max) a /= (float)(n_h * n_w);
      prow[e + u] = a;
      if (irow) irow[e + u] = use_max ? arg[u] : 0;
    }
  };

  // two separate element loops (not one loop with two bodies): the register allocation is then the larger of the two
  // paths, not their union
  // wave-uniform: bins whose pixel grid (n_h + 1) x (n_w + 1) fits the 16 LDS entries of a lane, corners de-duplicated
  // ('mean' takes the two-channel direct path only: with the grid on top its running sums and counts need 16 registers
  //  more than the four-waves-per-SIMD budget of 128 -- compiled: 16 spilled -- and at three waves the grid gives its gain
  //  back, as it did for 'max' at 134 registers in round 3)
  if (VEC == 2 && USE_MAX && dedup && n_h * n_w > 1 && (n_h + 1) * (n_w + 1) <= 16 && n_h <= 5 && n_w <= 5) {
    float2* grid = s_grid + (threadIdx.x >> 6) * (16 * 64) + lane;
    bool declined = false;
    for (int ev = e_first; ev * VEC < C; ev += e_step) {
      const int e = ev * VEC;
      const int pos = e / bank;
      const int row = pos / gw;
      const int col = pos - row * gw;
      const float x0 = xmin + bin_w * (float)col;
      const float y0 = ymin + bin_h * (float)row;
      float a2[2] = {use_max ? -FLT_MAX : 0.f, use_max ? -FLT_MAX : 0.f};
      int g2[2] = {0, 0};
      bool done = false;
#define XDET_PSROI_GRID(NH, NW)                                                                                             \
  case NH * 8 + NW:                                                                                                         \
    done = psroi_grid_bin<NH, NW, USE_MAX>(fimg, grid, H, W, sy, sx, e * sc, x0, y0, step_w, step_h, half_w, half_h, a2, g2); \
    break;
      switch (n_h * 8 + n_w) {
        XDET_PSROI_GRID(1, 2) XDET_PSROI_GRID(1, 3) XDET_PSROI_GRID(2, 1) XDET_PSROI_GRID(2, 2) XDET_PSROI_GRID(2, 3)
        XDET_PSROI_GRID(3, 1) XDET_PSROI_GRID(3, 2) XDET_PSROI_GRID(3, 3)
        XDET_PSROI_GRID(1, 4) XDET_PSROI_GRID(2, 4) XDET_PSROI_GRID(4, 1) XDET_PSROI_GRID(4, 2)
        XDET_PSROI_GRID(1, 5) XDET_PSROI_GRID(5, 1)
        default: break;
      }
#undef XDET_PSROI_GRID
      // psroi_grid_bin declines a bin only if float rounding put a sample further from the first one than its index
      // (never in exact arithmetic).  Then the whole ROI is redone by the direct loop below -- results are written per
      // element and do not depend on the path, so the elements already stored are simply stored again.
      if (__builtin_amdgcn_ballot_w64(!done) != 0) {
        declined = true;
        break;
      }
      float acc[VEC];
      int arg[VEC];
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        acc[u] = a2[u & 1];
        arg[u] = g2[u & 1];
      }
      finish(e, acc, arg);
    }
    if (!declined) return;
  }
  for (int ev = e_first; ev * VEC < C; ev += e_step) {
    const int e = ev * VEC;              // first of this lane's VEC channels (all in one bin: VEC divides bank)
    float acc[VEC];
    int arg[VEC];
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      acc[u] = use_max ? -FLT_MAX : 0.f;
      arg[u] = 0;
    }
    element_direct(e, acc, arg);
    finish(e, acc, arg);
  }
}

// [C][HW] -> [HW][C] per image through 32 x 32 LDS tiles (both sides coalesced): the front end of the NCHW form below
__global__ __launch_bounds__(256) void psroi_nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int HW) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* src = in + (size_t)n * C * HW;
  float* dst = out + (size_t)n * HW * C;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, px = p0 + tx;
    tile[ty + 8 * k][tx] = (c < C && px < HW) ? src[(size_t)c * HW + px] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = p0 + ty + 8 * k, c = c0 + tx;
    if (px < HW && c < C) dst[(size_t)px * C + c] = tile[tx][ty + 8 * k];
  }
}

int launch_psroialign(const float* feat, const float* rois, float* pooled, int32_t* index, int N, int C, int H,
                      int W, int R, int gw, int gh, int use_max, int layout, int ldc, int out_ld,
                      int rois_are_corners, hipStream_t s) {
  // same checks as PSROIAlignOp (ps_roi_align_op.cc:209-226)
  XDET_REQUIRE(gw > 0 && gh > 0, "Need Attr grid_dim_width/grid_dim_height > 0");
  XDET_REQUIRE(N >= 0 && C > 0 && H > 0 && W > 0 && R >= 0, "inputs must be in 'NCHW' format.");
  XDET_REQUIRE(C % (gw * gh) == 0, "channels must be divisible by grid_dim_width * grid_dim_height");
  XDET_REQUIRE(layout == 0 || layout == 1, "feat_layout must be 0 (NCHW) or 1 (NHWC)");
  XDET_REQUIRE(ldc >= C && out_ld >= C, "channel strides must be >= C");
  if ((int64_t)N * R == 0) return XDET_OK;
  // NCHW (the reference op's layout, light_head_rfcn_eval.py:85): neighbouring channels are H * W floats apart, so every
  // lane of a gather touches its own cache line (553 us for 64 x 300 ROIs of the mixed set against 120 us for the NHWC
  // form).  With enough ROIs it pays to transpose the map once into a stream-ordered scratch allocation and run the NHWC
  // two-channel kernel on it: same values, same arithmetic.  (An odd bank, a stream that is being
  // captured -- an allocation there would become a graph node, and a failed runtime call would invalidate the capture --
  // or a failed allocation keep the direct form.)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }
  if (layout == 0 && cap == hipStreamCaptureStatusNone && (C / (gw * gh)) % 2 == 0 && C % 2 == 0 &&
      (int64_t)R * C >= (int64_t)4 * H * W) {
    float* scratch = nullptr;
    const size_t bytes = (size_t)N * H * W * C * sizeof(float);
    if (hipMallocAsync(reinterpret_cast<void**>(&scratch), bytes, s) == hipSuccess && scratch) {
      hipLaunchKernelGGL(psroi_nchw_to_nhwc_kernel, dim3((unsigned)cdiv(H * W, 32), (unsigned)cdiv(C, 32), (unsigned)N), dim3(256), 0, s,
                         feat, scratch, C, H * W);
      const hipError_t te = hipGetLastError();           // the transpose's own launch error, before anything else is issued
      int rc = te == hipSuccess ? XDET_OK : hip_fail(te, "psroi_nchw_to_nhwc_kernel launch", __FILE__, __LINE__);
      if (rc == XDET_OK)
        rc = launch_psroialign(scratch, rois, pooled, index, N, C, H, W, R, gw, gh, use_max, 1, C, out_ld, rois_are_corners, s);
      (void)hipFreeAsync(scratch, s);
      return rc;
    }
    (void)hipGetLastError();                             // no scratch: the direct form below
  }
  // image n on XCD n & 7; fewer than 8 images: each image on 8 / N XCDs (see the kernel)
  const int split = (int64_t)N * R <= 2048 ? 1 : 0;      // one ROI per workgroup (see the kernel)
  const int64_t bpi = split ? R : cdiv(R, 4);
  const int64_t blocks = N >= 8 ? cdiv(N, 8) * 8 * bpi : 8 * cdiv(bpi, 8 / N);
  // two channels per lane (8-byte corner loads) where the layout allows it: NHWC, even bank and channel stride,
  // 8-byte aligned map
  constexpr int dedup = 1;                               // the two-channel form always reads through its LDS corner grid
  const int bank = C / (gw * gh);
  const bool two = layout == 1 && bank % 2 == 0 && ldc % 2 == 0 && reinterpret_cast<uintptr_t>(feat) % 8 == 0;
  const int cs = layout == 0 ? C : ldc;
#define XDET_PSROI_LAUNCH(V, M)                                                                                        \
  hipLaunchKernelGGL((psroialign_fwd_kernel<V, M>), dim3((unsigned)blocks), dim3(256), 0, s, feat, rois, pooled, index, N, C, \
                     H, W, R, gw, gh, layout, cs, out_ld, rois_are_corners, (V) == 2 ? dedup : 0, split)
  if (two && use_max) XDET_PSROI_LAUNCH(2, true);       // the net's form ('max', NHWC)
  else if (two) XDET_PSROI_LAUNCH(2, false);            // 'mean', NHWC: two channels per lane (8-byte corner loads), direct path
  else if (use_max) XDET_PSROI_LAUNCH(1, true);         // NCHW (the op's public contract): neighbouring channels are H*W
  else XDET_PSROI_LAUNCH(1, false);                     // floats apart -- one channel per lane, direct gathers
#undef XDET_PSROI_LAUNCH
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

}  // namespace xdet

// ---------------------------------------------------------------------------------------
// F2: PsRoiAlignGrad -- the reference's training backward
// (cpp/PSROIPooling/ps_roi_align_grad_op.cu:36-171): zero-fill, then scatter grad * bilinear weight
// with 4 float atomics per sample ('max': the argmax sample only; 'mean': every sample, grad/(n_h n_w)).
// Same wavefront-per-ROI decomposition and literal geometry as the forward kernel above.
// ---------------------------------------------------------------------------------------
namespace xdet {

__device__ __forceinline__ float* gout_at(float* __restrict__ g, int layout, int ldc, int H, int W, int64_t n, int c,
                                          int y, int x) {
  if (layout == 0) return g + ((n * ldc + c) * H + y) * W + x;
  return g + ((n * H + y) * W + x) * ldc + c;
}

__global__ __launch_bounds__(256) void psroialign_grad_kernel(const float* __restrict__ rois,
                                                              const float* __restrict__ grad_pooled,
                                                              const int32_t* __restrict__ pooled_index,
                                                              float* __restrict__ grad_out, int N, int C, int H, int W,
                                                              int R, int gw, int gh, int use_max, int layout, int ldc) {
  const int bank = C / (gw * gh);
  const int lane = threadIdx.x & 63;
  const int64_t nr = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (nr >= (int64_t)N * R) return;
  const int64_t n = nr / R;
  const float* roi = rois + nr * 4;
  const float r0 = roi[0], r1 = roi[1], r2 = roi[2], r3 = roi[3];
  if (r2 < FLT_MIN || r3 < FLT_MIN) return;
  const float yc = r0 * (float)H, xc = r1 * (float)W;
  const float rh = fmaxf(r2 * (float)H, 1.f), rw = fmaxf(r3 * (float)W, 1.f);
  const float ymin = fmaxf(yc - rh / 2.f, 0.f), xmin = fmaxf(xc - rw / 2.f, 0.f);
  const float ymax = fminf(yc + rh / 2.f, (float)H - FLT_MIN), xmax = fminf(xc + rw / 2.f, (float)W - FLT_MIN);
  const float bin_w = (xmax - xmin) / (float)gw, bin_h = (ymax - ymin) / (float)gh;
  const int n_w = (int)bin_w + 1, n_h = (int)bin_h + 1;
  const float step_w = bin_w / (float)n_w, step_h = bin_h / (float)n_h;
  const double half_w = (double)step_w / 2., half_h = (double)step_h / 2.;
  for (int e = lane; e < C; e += 64) {
    const int pos = e / bank, row = pos / gw, col = pos - row * gw;
    const float x0 = xmin + bin_w * (float)col, y0 = ymin + bin_h * (float)row;
    const int64_t w = nr * C + e;
    const int pi = use_max ? pooled_index[w] : 0;
    const int i_lo = use_max ? pi / n_w : 0, i_hi = use_max ? i_lo + 1 : n_h;
    const int j_lo = use_max ? pi % n_w : 0, j_hi = use_max ? j_lo + 1 : n_w;
    const float g = use_max ? grad_pooled[w] : grad_pooled[w] / (float)(n_w * n_h);
    for (int i = i_lo; i < i_hi; ++i) {
      const float y = (float)((double)(y0 + step_h * (float)i) + half_h);
      const int iy = (int)y;
      const float fy = y - (float)iy;
      const int iy1 = min(iy + 1, H - 1);
      for (int j = j_lo; j < j_hi; ++j) {
        const float x = (float)((double)(x0 + step_w * (float)j) + half_w);
        const int ix = (int)x;
        const float fx = x - (float)ix;
        const int ix1 = min(ix + 1, W - 1);
        atomicAdd(gout_at(grad_out, layout, ldc, H, W, n, e, iy, ix), (float)((1. - fx) * (1. - fy) * g));
        atomicAdd(gout_at(grad_out, layout, ldc, H, W, n, e, iy1, ix), (float)((1. - fx) * fy * g));
        atomicAdd(gout_at(grad_out, layout, ldc, H, W, n, e, iy, ix1), (float)(fx * (1. - fy) * g));
        atomicAdd(gout_at(grad_out, layout, ldc, H, W, n, e, iy1, ix1), (float)(fx * fy * g));
      }
    }
  }
}

int launch_psroialign_grad(const float* rois, const float* grad_pooled, const int32_t* pooled_index, float* grad_out,
                           int N, int C, int H, int W, int R, int gw, int gh, int use_max, int layout, int ldc,
                           hipStream_t s) {
  XDET_REQUIRE(gw > 0 && gh > 0, "Need Attr grid_dim_width/grid_dim_height > 0");
  XDET_REQUIRE(N >= 0 && C > 0 && H > 0 && W > 0 && R >= 0, "inputs must be in 'NCHW' format.");
  XDET_REQUIRE(C % (gw * gh) == 0, "channels must be divisible by grid_dim_width * grid_dim_height");
  XDET_REQUIRE(layout == 0 || layout == 1, "feat_layout must be 0 (NCHW) or 1 (NHWC)");
  XDET_REQUIRE(rois && grad_pooled && grad_out && (!use_max || pooled_index), "psroialign_grad: NULL argument");
  const int cs = layout == 0 ? C : ldc;
  XDET_REQUIRE(cs >= C, "channel stride must be >= C");
  XDET_HIP(hipMemsetAsync(grad_out, 0, (size_t)N * cs * H * W * sizeof(float), s));
  const int64_t n_waves = (int64_t)N * R;
  if (n_waves == 0) return XDET_OK;
  hipLaunchKernelGGL(psroialign_grad_kernel, dim3((unsigned)cdiv(n_waves, 4)), dim3(256), 0, s, rois, grad_pooled,
                     pooled_index, grad_out, N, C, H, W, R, gw, gh, use_max, layout, cs);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

}  // namespace xdet
