// RPN tail on the GPU (the reference pins this stage to /cpu:0, net/xception_body.py:425,
// where a single NonMaxSuppressionV2 takes ~85 ms of its ~270 ms step, README.md:6-8).
//
//   rpn_decode      A4+A6  pairwise softmax -> objectness; anchor decode
//                          (light_head_rfcn_eval.py:389-397, anchor_manipulator.py:641-669)
//   get_proposals   A7     clip -> size/centre filter -> top_k(pre_n) -> NMS(thr, post_n)
//                          -> drop pads -> upsample (net/xception_body.py:402-448, :41-213)
//
// No host round trip and no data-dependent launch shape: every buffer is fixed-size and
// the counts live in device memory, so the whole stage is hipGraph-capturable.
//   1. prepare : clipped box + 64-bit sort key (score bits << 32 | ~index), 0 = filtered
//   2. rank    : rank_i = #{j : key_j > key_i}  (all keys distinct -> a permutation; equals
//                tf.nn.top_k order: descending score, ties -> lower index)
//   3. scatter : sorted_boxes[rank] = box  for rank < pre_n
//   4. mask    : bit (i,j) = IoU(i,j) > thr for j > i, 64x64 tiles
//   5. scan    : one wavefront walks the candidates in score order, 64 at a time
//   6. gather  : rois[j] = kept[j mod n_keep]  (tile + identity "shuffle" of :196-213)
#include "common.h"

namespace xdet {

typedef unsigned long long u64;

size_t proposal_workspace_bytes(int N, int n_anchor, int pre_n, int post_n) {
  const size_t w64 = (size_t)cdiv(pre_n, 64);
  size_t b = 0;
  auto add = [&](size_t x) { b += (x + 255) / 256 * 256; };
  add((size_t)N * n_anchor * 8);
  add((size_t)N * n_anchor * 16);
  add((size_t)N * n_anchor * 4);
  add((size_t)N * 16);
  add((size_t)N * pre_n * 16);
  add((size_t)N * pre_n * 4);
  add((size_t)N * pre_n * w64 * 8);
  add((size_t)N * post_n * 4);
  return b;
}

void proposal_workspace_carve(void* base, int N, int n_anchor, int pre_n, int post_n, ProposalWorkspace* ws) {
  const size_t w64 = (size_t)cdiv(pre_n, 64);
  char* p = static_cast<char*>(base);
  auto take = [&](size_t x) { char* r = p; p += (x + 255) / 256 * 256; return r; };
  ws->keys = reinterpret_cast<u64*>(take((size_t)N * n_anchor * 8));
  ws->cboxes = reinterpret_cast<float*>(take((size_t)N * n_anchor * 16));
  ws->ranks = reinterpret_cast<int*>(take((size_t)N * n_anchor * 4));
  ws->counts = reinterpret_cast<int*>(take((size_t)N * 16));
  ws->sboxes = reinterpret_cast<float*>(take((size_t)N * pre_n * 16));
  ws->sscores = reinterpret_cast<float*>(take((size_t)N * pre_n * 4));
  ws->mask = reinterpret_cast<u64*>(take((size_t)N * pre_n * w64 * 8));
  ws->kept = reinterpret_cast<int*>(take((size_t)N * post_n * 4));
}

// ---------------------------------------------------------------------------------------
// A4 + A6
// ---------------------------------------------------------------------------------------
__global__ void rpn_decode_kernel(const float* __restrict__ rpn_out, int ld, int cls_off, int box_off, int N, int HW,
                                  int Ww, int A, const float* __restrict__ anc_yx, const float* __restrict__ anc_hw,
                                  float* __restrict__ objectness, float* __restrict__ boxes) {
  const int64_t total = (int64_t)N * HW * A;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int a = (int)(i % A);
    const int64_t px = i / A;              // n*HW + pos
    const int pos = (int)(px % HW);
    const float* row = rpn_out + px * ld;
    // softmax over (bg, fg), take fg: tf.nn.softmax(...)[:, -1]
    const float l0 = row[cls_off + 2 * a], l1 = row[cls_off + 2 * a + 1];
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    objectness[i] = e1 / (e0 + e1);
    const float4 d = *reinterpret_cast<const float4*>(row + box_off + 4 * a);   // (cy, cx, h, w) deltas
    const float yref = anc_yx[2 * pos], xref = anc_yx[2 * pos + 1];
    const float href = anc_hw[2 * a], wref = anc_hw[2 * a + 1];
    const float ph = expf(d.z) * href;
    const float pw = expf(d.w) * wref;
    const float pcy = d.x * href + yref;
    const float pcx = d.y * wref + xref;
    *reinterpret_cast<float4*>(boxes + i * 4) =
        make_float4(pcy - ph / 2.f, pcx - pw / 2.f, pcy + ph / 2.f, pcx + pw / 2.f);
  }
}

int launch_rpn_decode(const float* rpn_out, int ld, int cls_off, int box_off, int N, int Hh, int Ww, int A,
                      const float* anchors_yx, const float* anchors_hw, float* objectness, float* boxes,
                      hipStream_t s) {
  XDET_REQUIRE(box_off % 4 == 0 && ld % 4 == 0, "rpn_decode: box channels must be 16-byte aligned");
  const int64_t total = (int64_t)N * Hh * Ww * A;
  if (total == 0) return XDET_OK;
  hipLaunchKernelGGL(rpn_decode_kernel, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 4096)), dim3(256), 0, s,
                     rpn_out, ld, cls_off, box_off, N, Hh * Ww, Ww, A, anchors_yx, anchors_hw, objectness, boxes);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// ---------------------------------------------------------------------------------------
// A7
// ---------------------------------------------------------------------------------------
__global__ void prop_prepare_kernel(const float* __restrict__ score, const float* __restrict__ boxes, int n_anchor,
                                    float min_size, u64* __restrict__ keys, float* __restrict__ cboxes,
                                    int* __restrict__ ranks, int* __restrict__ counts) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = false;
  if (i < n_anchor) {
    const int64_t g = (int64_t)n * n_anchor + i;
    const float4 b = *reinterpret_cast<const float4*>(boxes + g * 4);
    // _bboxes_clip to [0,0,1,1]  (:173-194)
    float ymin = fmaxf(b.x, 0.f), xmin = fmaxf(b.y, 0.f);
    const float ymax = fminf(b.z, 1.f), xmax = fminf(b.w, 1.f);
    ymin = fminf(ymin, ymax);
    xmin = fminf(xmin, xmax);
    *reinterpret_cast<float4*>(cboxes + g * 4) = make_float4(ymin, xmin, ymax, xmax);
    // _filter_and_sort_boxes (:133-158)
    const float ws = xmax - xmin, hs = ymax - ymin;
    const float xc = xmin + ws / 2.f, yc = ymin + hs / 2.f;
    const float sc = score[g];
    // a score <= 0 survives top_k in the reference but is dropped as padding by
    // _upsample_rois (:199-200); such entries sort last, so excluding them here is equivalent.
    valid = ws > min_size && hs > min_size && xc > 0.f && yc > 0.f && xc < 1.f && yc < 1.f && sc > 0.f;
    keys[g] = valid ? (((u64)__float_as_uint(sc) << 32) | (u64)(0xFFFFFFFFu - (unsigned)i)) : 0ull;
    ranks[g] = 0;
  }
  const u64 ball = __ballot(valid);
  if ((threadIdx.x & 63) == 0 && ball) atomicAdd(&counts[n * 4 + 0], __popcll(ball));
}

constexpr int RANK_TILE = 1024;
constexpr int RANK_IPT = 4;      // keys owned per thread (amortises the LDS broadcast reads)

// grid (ceil(n/1024), J splits, N): each thread owns 4 keys and counts larger keys in its j slice
__global__ __launch_bounds__(256) void prop_rank_kernel(const u64* __restrict__ keys, int n_anchor, int j_per_split,
                                                        int* __restrict__ ranks) {
  __shared__ u64 tile[RANK_TILE];
  const int n = blockIdx.z;
  const u64* k = keys + (int64_t)n * n_anchor;
  u64 mine[RANK_IPT];
  int cnt[RANK_IPT];
  bool any = false;
#pragma unroll
  for (int q = 0; q < RANK_IPT; ++q) {
    const int i = blockIdx.x * (256 * RANK_IPT) + q * 256 + threadIdx.x;
    mine[q] = i < n_anchor ? k[i] : 0ull;
    cnt[q] = 0;
    any |= mine[q] != 0ull;
  }
  const int j0 = blockIdx.y * j_per_split;
  const int j1 = min(j0 + j_per_split, n_anchor);
  for (int jb = j0; jb < j1; jb += RANK_TILE) {
    __syncthreads();
    for (int t = threadIdx.x; t < RANK_TILE; t += 256) tile[t] = (jb + t < j1) ? k[jb + t] : 0ull;
    __syncthreads();
    if (any) {
#pragma unroll 4
      for (int t = 0; t < RANK_TILE; ++t) {
        const u64 o = tile[t];
#pragma unroll
        for (int q = 0; q < RANK_IPT; ++q) cnt[q] += o > mine[q];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < RANK_IPT; ++q) {
    const int i = blockIdx.x * (256 * RANK_IPT) + q * 256 + threadIdx.x;
    if (mine[q] != 0ull && cnt[q]) atomicAdd(&ranks[(int64_t)n * n_anchor + i], cnt[q]);
  }
}

__global__ void prop_scatter_kernel(const u64* __restrict__ keys, const int* __restrict__ ranks,
                                    const float* __restrict__ cboxes, int n_anchor, int pre_n,
                                    float* __restrict__ sboxes, float* __restrict__ sscores, int* __restrict__ counts) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) counts[n * 4 + 1] = min(counts[n * 4 + 0], pre_n);   // n_cand
  if (i >= n_anchor) return;
  const int64_t g = (int64_t)n * n_anchor + i;
  const u64 key = keys[g];
  if (key == 0ull) return;
  const int r = ranks[g];
  if (r < pre_n) {
    *reinterpret_cast<float4*>(sboxes + ((int64_t)n * pre_n + r) * 4) = *reinterpret_cast<const float4*>(cboxes + g * 4);
    sscores[(int64_t)n * pre_n + r] = __uint_as_float((unsigned)(key >> 32));
  }
}

// IoU as tf.image.non_max_suppression computes it (NonMaxSuppressionV2): corners min/max
// normalised, zero when either area is <= 0.
__device__ __forceinline__ bool iou_gt(const float4 a, const float4 b, float thr) {
  const float ay0 = fminf(a.x, a.z), ay1 = fmaxf(a.x, a.z), ax0 = fminf(a.y, a.w), ax1 = fmaxf(a.y, a.w);
  const float by0 = fminf(b.x, b.z), by1 = fmaxf(b.x, b.z), bx0 = fminf(b.y, b.w), bx1 = fmaxf(b.y, b.w);
  const float aa = (ay1 - ay0) * (ax1 - ax0);
  const float ab = (by1 - by0) * (bx1 - bx0);
  if (aa <= 0.f || ab <= 0.f) return false;
  const float ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.f);
  const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.f);
  const float inter = ih * iw;
  return inter / ((aa + ab) - inter) > thr;
}

// grid (W64 col blocks, W64 row blocks, N), 64 threads: thread t = row rb*64+t against 64 columns
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ sboxes, const int* __restrict__ counts,
                                                      int pre_n, int w64, float thr, u64* __restrict__ mask) {
  const int n = blockIdx.z;
  const int cb = blockIdx.x, rb = blockIdx.y;
  const int n_cand = counts[n * 4 + 1];
  const int row = rb * 64 + threadIdx.x;
  if (rb * 64 >= n_cand) return;                 // rows never visited by the scan
  u64 bits = 0ull;
  if (cb >= rb && cb * 64 < n_cand) {
    __shared__ float4 cbox[64];
    const int col = cb * 64 + threadIdx.x;
    cbox[threadIdx.x] = col < n_cand ? *reinterpret_cast<const float4*>(sboxes + ((int64_t)n * pre_n + col) * 4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (row < n_cand) {
      const float4 me = *reinterpret_cast<const float4*>(sboxes + ((int64_t)n * pre_n + row) * 4);
      const int jstart = (cb == rb) ? threadIdx.x + 1 : 0;
      for (int j = jstart; j < 64; ++j) {
        if (cb * 64 + j < n_cand && iou_gt(me, cbox[j], thr)) bits |= 1ull << j;
      }
    }
  }
  if (row < pre_n) mask[((int64_t)n * pre_n + row) * w64 + cb] = bits;
}

__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
  const unsigned lo = __shfl((unsigned)v, src), hi = __shfl((unsigned)(v >> 32), src);
  return ((u64)hi << 32) | lo;
}

// one wavefront per image; MAXW = max mask words per lane (w64 <= 64*MAXW)
template <int MAXW>
__global__ __launch_bounds__(64) void nms_scan_kernel(const u64* __restrict__ mask, int* __restrict__ counts,
                                                      int pre_n, int w64, int post_n, int* __restrict__ kept) {
  const int n = blockIdx.x;
  const int lane = threadIdx.x;
  const int n_cand = counts[n * 4 + 1];
  const u64* M = mask + (int64_t)n * pre_n * w64;
  int* K = kept + (int64_t)n * post_n;
  u64 removed[MAXW];
#pragma unroll
  for (int q = 0; q < MAXW; ++q) removed[q] = 0ull;
  int n_keep = 0;
  const int n_chunk = (n_cand + 63) / 64;
  for (int c = 0; c < n_chunk && n_keep < post_n; ++c) {
    const int i = c * 64 + lane;
    const u64 diag = i < n_cand ? M[(int64_t)i * w64 + c] : 0ull;
    u64 cur = 0ull;
#pragma unroll
    for (int q = 0; q < MAXW; ++q)
      if ((c >> 6) == q) cur = shfl_u64(removed[q], c & 63);
    const int lim = min(64, n_cand - c * 64);
    u64 keepmask = 0ull;
    int kcount = n_keep;
    for (int b = 0; b < lim && kcount < post_n; ++b) {     // wave-uniform serial resolve of the 64x64 diagonal tile
      const u64 d = shfl_u64(diag, b);
      if (!((cur >> b) & 1ull)) {
        keepmask |= 1ull << b;
        cur |= d;
        ++kcount;
      }
    }
    if ((keepmask >> lane) & 1ull) K[n_keep + __popcll(keepmask & ((1ull << lane) - 1ull))] = i;
    n_keep = kcount;
    if (n_keep >= post_n) break;
    // fold the kept rows into the removed set (independent loads, no serial dependency)
    u64 km = keepmask;
    while (km) {
      const int b = __ffsll((long long)km) - 1;
      km &= km - 1ull;
      const u64* rowp = M + (int64_t)(c * 64 + b) * w64;
#pragma unroll
      for (int q = 0; q < MAXW; ++q)
        if (q * 64 + lane < w64) removed[q] |= rowp[q * 64 + lane];
    }
  }
  if (lane == 0) counts[n * 4 + 2] = n_keep;
}

__global__ void prop_gather_kernel(const float* __restrict__ sboxes, const int* __restrict__ kept,
                                   const int* __restrict__ counts, int pre_n, int post_n, float* __restrict__ rois) {
  const int n = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= post_n) return;
  const int n_keep = counts[n * 4 + 2];
  float4 b = make_float4(.2f, .2f, .8f, .8f);    // empty-set fallback (:202)
  if (n_keep > 0) {
    const int src = kept[(int64_t)n * post_n + (j % n_keep)];
    b = *reinterpret_cast<const float4*>(sboxes + ((int64_t)n * pre_n + src) * 4);
  }
  *reinterpret_cast<float4*>(rois + ((int64_t)n * post_n + j) * 4) = b;
}

int launch_get_proposals(const float* objectness, const float* boxes, int N, int n_anchor, int pre_n, int post_n,
                         float nms_thr, float min_size, const ProposalWorkspace& ws, float* rois, hipStream_t s) {
  XDET_REQUIRE(N > 0 && n_anchor > 0 && pre_n > 0 && post_n > 0, "get_proposals: sizes must be positive");
  const int w64 = (int)cdiv(pre_n, 64);
  XDET_REQUIRE(w64 <= 64 * 4, "get_proposals: rpn_pre_nms_top_n too large (max 16384)");
  XDET_HIP(hipMemsetAsync(ws.counts, 0, (size_t)N * 16, s));
  XDET_HIP(hipMemsetAsync(ws.sboxes, 0, (size_t)N * pre_n * 16, s));
  XDET_HIP(hipMemsetAsync(ws.sscores, 0, (size_t)N * pre_n * 4, s));
  const unsigned gb = (unsigned)cdiv(n_anchor, 256);
  hipLaunchKernelGGL(prop_prepare_kernel, dim3(gb, N), dim3(256), 0, s, objectness, boxes, n_anchor, min_size,
                     ws.keys, ws.cboxes, ws.ranks, ws.counts);
  XDET_LAUNCH_CHECK();
  const unsigned gr = (unsigned)cdiv(n_anchor, 256 * RANK_IPT);
  const int splits = std::max(1, std::min<int>((int)cdiv(n_anchor, RANK_TILE), 2048 / (int)(gr * N) + 1));
  const int j_per_split = (int)cdiv(cdiv(n_anchor, splits), RANK_TILE) * RANK_TILE;
  hipLaunchKernelGGL(prop_rank_kernel, dim3(gr, (unsigned)cdiv(n_anchor, j_per_split), N), dim3(256), 0, s, ws.keys,
                     n_anchor, j_per_split, ws.ranks);
  XDET_LAUNCH_CHECK();
  hipLaunchKernelGGL(prop_scatter_kernel, dim3(gb, N), dim3(256), 0, s, ws.keys, ws.ranks, ws.cboxes, n_anchor, pre_n,
                     ws.sboxes, ws.sscores, ws.counts);
  XDET_LAUNCH_CHECK();
  hipLaunchKernelGGL(nms_mask_kernel, dim3(w64, w64, N), dim3(64), 0, s, ws.sboxes, ws.counts, pre_n, w64, nms_thr,
                     ws.mask);
  XDET_LAUNCH_CHECK();
  if (w64 <= 64) {
    hipLaunchKernelGGL(nms_scan_kernel<1>, dim3(N), dim3(64), 0, s, ws.mask, ws.counts, pre_n, w64, post_n, ws.kept);
  } else if (w64 <= 128) {
    hipLaunchKernelGGL(nms_scan_kernel<2>, dim3(N), dim3(64), 0, s, ws.mask, ws.counts, pre_n, w64, post_n, ws.kept);
  } else {
    hipLaunchKernelGGL(nms_scan_kernel<4>, dim3(N), dim3(64), 0, s, ws.mask, ws.counts, pre_n, w64, post_n, ws.kept);
  }
  XDET_LAUNCH_CHECK();
  hipLaunchKernelGGL(prop_gather_kernel, dim3((unsigned)cdiv(post_n, 256), N), dim3(256), 0, s, ws.sboxes, ws.kept,
                     ws.counts, pre_n, post_n, rois);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

}  // namespace xdet
