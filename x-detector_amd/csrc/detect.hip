// Detection post-processing on the GPU (the reference pins it to /device:CPU:0,
// light_head_rfcn_eval.py:273).
//
//   ext_decode_rois  A11  preprocessing/anchor_manipulator.py:671-683
//   bboxes_eval      A12  light_head_rfcn_eval.py:263-287 -> utility/eval_helper.py:
//                         tf_bboxes_select (:556-625) -> bboxes_clip (:365-404) -> filter_boxes
//                         (:278-317) -> bboxes_resize (:423-447) -> bboxes_sort (:333-361)
//                         -> bboxes_nms_batch (:449-506)
//
// One workgroup per (image, class): softmax column, mask, clip, filter, rank-sort (top 2*topk),
// bitmask NMS and the ordered scan all stay in LDS; output is the reference's zero-padded
// per-class (scores[topk], boxes[topk,4]).
#include "common.h"
#include <cfloat>

namespace xdet {

typedef unsigned long long u64;

__global__ void ext_decode_rois_kernel(const float* __restrict__ rois, const float* __restrict__ reg, int ld_reg,
                                       int64_t n, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 r = *reinterpret_cast<const float4*>(rois + i * 4);
    const float* p = reg + i * ld_reg;
    const float href = r.z - r.x, wref = r.w - r.y;
    const float yref = r.x + href / 2.f, xref = r.y + wref / 2.f;
    const float ph = expf(p[2]) * href;
    const float pw = expf(p[3]) * wref;
    const float pcy = p[0] * href + yref;
    const float pcx = p[1] * wref + xref;
    *reinterpret_cast<float4*>(out + i * 4) =
        make_float4(pcy - ph / 2.f, pcx - pw / 2.f, pcy + ph / 2.f, pcx + pw / 2.f);
  }
}

int launch_ext_decode_rois(const float* rois, const float* reg, int ld_reg, int64_t n, float* out, hipStream_t s) {
  if (n == 0) return XDET_OK;
  hipLaunchKernelGGL(ext_decode_rois_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 2048)), dim3(256), 0, s,
                     rois, reg, ld_reg, n, out);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

// IoU(a, b) > thr as tf.image.non_max_suppression decides it (corners min/max-normalised, zero when either area is <= 0,
// strict >), without a division on the fast path (as proposals.hip's iou_gt_fast): the correctly rounded
// quotient can only disagree with the product test inside a 1e-5 relative band around the threshold, and only there
// is the division evaluated.  The per-class NMS mask is ~45,000 IoUs per (image, class) workgroup.
__device__ __forceinline__ bool iou_gt_fast_d(const float4 a, const float4 b, float thr) {
  const float ay0 = fminf(a.x, a.z), ay1 = fmaxf(a.x, a.z), ax0 = fminf(a.y, a.w), ax1 = fmaxf(a.y, a.w);
  const float by0 = fminf(b.x, b.z), by1 = fmaxf(b.x, b.z), bx0 = fminf(b.y, b.w), bx1 = fmaxf(b.y, b.w);
  const float ih = fminf(ay1, by1) - fmaxf(ay0, by0);
  const float iw = fminf(ax1, bx1) - fmaxf(ax0, bx0);
  if (ih <= 0.f || iw <= 0.f) return false;          // no overlap: IoU = 0 <= thr (thr >= 0)
  const float aa = (ay1 - ay0) * (ax1 - ax0);
  const float ab = (by1 - by0) * (bx1 - bx0);
  if (aa <= 0.f || ab <= 0.f) return false;
  const float inter = ih * iw;
  const float uni = (aa + ab) - inter;
  const float t = thr * uni;
  if (inter > t * 1.00001f) return true;
  if (inter < t * 0.99999f) return false;
  return inter / uni > thr;
}

// lane `src` (wave-uniform) of a 64-bit value as a scalar: v_readlane, not a ds_bpermute round trip (proposals.hip)
__device__ __forceinline__ u64 readlane_u64d(u64 v, int src) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
  return ((u64)hi << 32) | lo;
}

// the same decision from pre-normalised corners (y0, x0, y1, x1 with y0 <= y1, x0 <= x1) and pre-computed areas
__device__ __forceinline__ bool iou_norm_gt_d(const float4 a, const float aa, const float4 b, const float ab, float thr) {
  const float ih = fminf(a.z, b.z) - fmaxf(a.x, b.x);
  const float iw = fminf(a.w, b.w) - fmaxf(a.y, b.y);
  if (ih <= 0.f || iw <= 0.f) return false;
  if (aa <= 0.f || ab <= 0.f) return false;
  const float inter = ih * iw;
  const float uni = (aa + ab) - inter;
  const float t = thr * uni;
  if (inter > t * 1.00001f) return true;
  if (inter < t * 0.99999f) return false;
  return inter / uni > thr;
}

constexpr int EV_MAXR = 1024;   // ROIs per image supported by one workgroup
constexpr int EV_MAXS = 512;    // 2*nms_topk upper bound (sorted candidates)
constexpr int EV_W = EV_MAXS / 64;
constexpr int EV_T = 1024;      // threads per workgroup

// grid (num_classes-1, N), EV_T threads.  One workgroup per (image, class) is all the parallelism a single image offers
// (20 workgroups on 256 CUs), so the workgroup is as wide as it can be: the rank sort and the IoU mask are spread over
// 16 waves instead of 4 (single-image latency; at large batches the total work is what counts and is unchanged).
__global__ __launch_bounds__(EV_T) void bboxes_eval_kernel(const float* __restrict__ cls, int ld_cls,
                                                          const float* __restrict__ boxes, int R, int num_classes,
                                                          const int* __restrict__ image_shapes,
                                                          const float* __restrict__ bbox_img, int net_h, int net_w,
                                                          float select_thr, float nms_thr, int nms_topk,
                                                          float* __restrict__ det_scores,
                                                          float* __restrict__ det_boxes,
                                                          const int* __restrict__ bad_per_image) {
  __shared__ u64 keys[EV_MAXR];
  __shared__ float4 bx[EV_MAXR];
  __shared__ float4 sbox[EV_MAXS];
  __shared__ float sscore[EV_MAXS];
  __shared__ u64 mask[EV_MAXS * EV_W];
  __shared__ u64 s_removed[EV_W];
  // (aliases, so that two workgroups still share a CU: the sorted boxes with min/max-normalised corners -- what iou_gt_fast_d
  //  makes of its arguments -- and their areas live in `bx`, dead once the sorted copies exist; the rank counters in `mask`,
  //  which is not written before they are dead)
  float4* const snorm = bx;
  float* const sarea = reinterpret_cast<float*>(bx + EV_MAXS);
  int* const s_rank = reinterpret_cast<int*>(mask);
  static_assert(EV_MAXS * 16 + EV_MAXS * 4 <= EV_MAXR * 16 && EV_MAXR * 4 <= EV_MAXS * EV_W * 8, "aliases fit");
  __shared__ int s_nvalid;
  __shared__ int s_keptidx[EV_MAXS];
  __shared__ int s_nkeep;
  __shared__ int s_bad;

  const int c = blockIdx.x + 1;
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
  if (tid == 0) { s_nvalid = 0; s_nkeep = 0; s_bad = 0; }
  __syncthreads();

  const float4 ref = *reinterpret_cast<const float4*>(bbox_img + n * 4);
  // eval_helper.filter_boxes :296  min_size = max(1e-4, ratio*sqrt(H*W / (net_h*net_w)))
  const float area = (float)((long long)image_shapes[n * 2] * (long long)image_shapes[n * 2 + 1]);
  const float min_size = fmaxf(0.0001f, 0.03f * sqrtf(area / (float)(net_h * net_w)));
  const float sx = ref.z - ref.x, sy = ref.w - ref.y;   // bboxes_resize scale (h, w)

  int local_valid = 0, local_bad = bad_per_image ? bad_per_image[n] : 0;   // (proposal stage: non-finite RPN outputs)
  for (int r = tid; r < R; r += EV_T) {
    const float* lg = cls + ((int64_t)n * R + r) * ld_cls;
    float m = lg[0];
    float lsum = lg[0];
    for (int k = 1; k < num_classes; ++k) { m = fmaxf(m, lg[k]); lsum += lg[k]; }
    // A non-finite logit (an activation that left the f16 range of the split-precision convs upstream, or a broken
    // checkpoint) would otherwise vanish here: its score compares false against the threshold and the image simply
    // has no detections.  It is made loud instead: the (image, class) slot is marked NaN below, the host raises.
    local_bad |= !(fabsf(lsum) <= FLT_MAX);
    float sum = 0.f, ec = 0.f;
    for (int k = 0; k < num_classes; ++k) {
      const float e = expf(lg[k] - m);
      sum += e;
      if (k == c) ec = e;
    }
    float s = ec / sum;
    const float fmask = s > select_thr ? 1.f : 0.f;          // tf_bboxes_select_layer :581-585
    s = s * fmask;
    float4 b = *reinterpret_cast<const float4*>(boxes + ((int64_t)n * R + r) * 4);
    b.x *= fmask; b.y *= fmask; b.z *= fmask; b.w *= fmask;
    // bboxes_clip(bbox_img, .)
    float ymin = fmaxf(b.x, ref.x), xmin = fmaxf(b.y, ref.y);
    const float ymax = fminf(b.z, ref.z), xmax = fminf(b.w, ref.w);
    ymin = fminf(ymin, ymax);
    xmin = fminf(xmin, xmax);
    // filter_boxes
    const float ws = xmax - xmin, hs = ymax - ymin;
    const float xc = xmin + ws / 2.f, yc = ymin + hs / 2.f;
    // zero-score survivors only ever act as zero padding downstream -> drop them here
    const bool valid = ws > min_size && hs > min_size && xc > 0.f && yc > 0.f && xc < 1.f && yc < 1.f && s > 0.f;
    // bboxes_resize
    bx[r] = make_float4((ymin - ref.x) / sx, (xmin - ref.y) / sy, (ymax - ref.x) / sx, (xmax - ref.y) / sy);
    keys[r] = valid ? (((u64)__float_as_uint(s) << 32) | (u64)(0xFFFFFFFFu - (unsigned)r)) : 0ull;
    local_valid += valid;
  }
  if (local_valid) atomicAdd(&s_nvalid, local_valid);
  if (local_bad) atomicOr(&s_bad, 1);
  __syncthreads();

  const int max_sorted = min(2 * nms_topk, EV_MAXS);
  const int n_sorted = min(s_nvalid, max_sorted);            // bboxes_sort: top_k(min(n, 2*topk))
  // rank sort (descending score, ties -> lower ROI index).  The R x R comparisons are spread over the whole workgroup: P threads
  // per ROI count a slice of the keys each (a single image has 300 ROIs for 1024 threads)
  for (int r = tid; r < R; r += EV_T) s_rank[r] = 0;
  __syncthreads();
  {
    const int P = max(1, EV_T / R);
    const int slice = (R + P - 1) / P;
    for (int t = tid; t < R * P; t += EV_T) {
      const int part = t / R, r = t - part * R;
      const u64 mine = keys[r];
      if (mine == 0ull) continue;
      const int j0 = part * slice, j1 = min(R, j0 + slice);
      int cnt = 0;
#pragma unroll 8
      for (int j = j0; j < j1; ++j) cnt += keys[j] > mine;
      if (cnt) atomicAdd(&s_rank[r], cnt);
    }
  }
  __syncthreads();
  {
    // R <= EV_T: one ROI per thread.  Read everything first: the sorted copies overwrite `bx`
    const bool have = tid < R && keys[tid] != 0ull;
    const u64 mine = have ? keys[tid] : 0ull;
    const int rank = have ? s_rank[tid] : max_sorted;
    const float4 b = have ? bx[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (rank < max_sorted) {
      sbox[rank] = b;
      sscore[rank] = __uint_as_float((unsigned)(mine >> 32));
      const float y0 = fminf(b.x, b.z), y1 = fmaxf(b.x, b.z), x0 = fminf(b.y, b.w), x1 = fmaxf(b.y, b.w);
      snorm[rank] = make_float4(y0, x0, y1, x1);
      sarea[rank] = (y1 - y0) * (x1 - x0);
    }
    if (tid < EV_W) s_removed[tid] = 0ull;
  }
  __syncthreads();

  // NMS bitmask: bit (i,j) for j > i.  One wave per (row i, 64-column block): a lane keeps ITS column's box in registers across
  // the rows, the row's box is a broadcast read, the word is a ballot -- no per-lane loop over 64 columns with divergent exits,
  // and the blocks left of the diagonal (no j > i in them) are written as zeros without an IoU.  iou_norm_gt_d is iou_gt_fast_d
  // on the pre-normalised corners and areas: the same comparisons on the same values.
  const int w64 = (n_sorted + 63) / 64;
  {
    const int wv = tid >> 6, ln = tid & 63;
    for (int wq = 0; wq < w64; ++wq) {
      const int col = wq * 64 + ln;
      const bool col_ok = col < n_sorted;
      const float4 cb = col_ok ? snorm[col] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float ca = col_ok ? sarea[col] : 0.f;
      for (int i = wv; i < n_sorted; i += EV_T / 64) {
        u64 bits = 0ull;
        if (i < wq * 64 + 63) {                       // (wave-uniform) some column of this block is right of the diagonal
          const float4 me = snorm[i];
          const float ma = sarea[i];
          bits = __ballot(col_ok && col > i && iou_norm_gt_d(me, ma, cb, ca, nms_thr));
        }
        if (ln == 0) mask[i * EV_W + wq] = bits;
      }
    }
  }
  __syncthreads();

  // ordered scan by wave 0 (lane q < EV_W holds removed word q).  Inside a 64-box chunk only the boxes that SURVIVE are visited
  // (lowest bit still available, then its row's bits leave the set): the same sequence as a walk over all 64 bits, in as many
  // steps as boxes are kept.  The rows of the kept boxes then go into the later chunks' removed words in parallel (each kept
  // lane ORs its own row in), not one LDS round trip per kept box.
  if (tid < 64) {
    const int lane = tid;
    u64 removed = 0ull;
    int n_keep = 0;
    for (int cch = 0; cch < w64 && n_keep < nms_topk; ++cch) {
      const int i = cch * 64 + lane;
      const u64 diag = i < n_sorted ? mask[i * EV_W + cch] : 0ull;
      const u64 cur = readlane_u64d(removed, cch);             // scalar chain: cur, avail, keepmask are wave-uniform
      const int lim = __builtin_amdgcn_readfirstlane(min(64, n_sorted - cch * 64));
      u64 avail = ~cur & (lim == 64 ? ~0ull : ((1ull << lim) - 1ull));
      u64 keepmask = 0ull;
      int kc = n_keep;
      while (avail != 0ull && kc < nms_topk) {
        const int b = __ffsll((long long)avail) - 1;
        keepmask |= 1ull << b;
        ++kc;
        avail &= ~(readlane_u64d(diag, b) | (1ull << b));
      }
      const bool kept = (keepmask >> lane) & 1ull;
      if (kept) s_keptidx[n_keep + __popcll(keepmask & ((1ull << lane) - 1ull))] = i;
      n_keep = kc;
      if (cch + 1 < w64 && n_keep < nms_topk) {
        if (kept)
          for (int q = cch + 1; q < w64; ++q) {
            const u64 row = mask[i * EV_W + q];
            if (row) atomicOr(&s_removed[q], row);
          }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        removed = lane < EV_W ? *reinterpret_cast<volatile u64*>(&s_removed[lane]) : 0ull;
      }
    }
    if (lane == 0) s_nkeep = n_keep;
  }
  __syncthreads();

  const int n_keep = s_nkeep;
  float* os = det_scores + ((int64_t)n * (num_classes - 1) + (c - 1)) * nms_topk;
  float* ob = det_boxes + ((int64_t)n * (num_classes - 1) + (c - 1)) * nms_topk * 4;
  for (int k = tid; k < nms_topk; k += EV_T) {
    float s = 0.f;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < n_keep) {
      const int src = s_keptidx[k];
      s = sscore[src];
      b = sbox[src];
    }
    if (k == 0 && s_bad) s = NAN;
    os[k] = s;
    *reinterpret_cast<float4*>(ob + k * 4) = b;
  }
}

int launch_bboxes_eval(const float* cls, int ld_cls, const float* boxes, int N, int R, int num_classes,
                       const int* image_shapes, const float* bbox_img, int net_h, int net_w, float select_thr,
                       float nms_thr, int nms_topk, float* det_scores, float* det_boxes, hipStream_t s,
                       const int* bad_per_image) {
  XDET_REQUIRE(R > 0 && R <= EV_MAXR, "bboxes_eval: 1 <= rois per image <= 1024");
  XDET_REQUIRE(nms_topk > 0 && 2 * nms_topk <= EV_MAXS, "bboxes_eval: nms_topk must be in 1..256");
  XDET_REQUIRE(num_classes >= 2 && num_classes <= ld_cls, "bboxes_eval: bad num_classes");
  if (N == 0) return XDET_OK;
  hipLaunchKernelGGL(bboxes_eval_kernel, dim3(num_classes - 1, N), dim3(EV_T), 0, s, cls, ld_cls, boxes, R,
                     num_classes, image_shapes, bbox_img, net_h, net_w, select_thr, nms_thr, nms_topk, det_scores,
                     det_boxes, bad_per_image);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

}  // namespace xdet
