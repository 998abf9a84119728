// Detection post-processing on the GPU (the reference pins it to /device:CPU:0,
// light_head_rfcn_eval.py:273).
//
//   ext_decode_rois  A11  preprocessing/anchor_manipulator.py:671-683
//   bboxes_eval      A12  light_head_rfcn_eval.py:263-287 -> utility/eval_helper.py:
//                         tf_bboxes_select (:556-625) -> bboxes_clip (:365-404) -> filter_boxes
//                         (:278-317) -> bboxes_resize (:423-447) -> bboxes_sort (:333-361)
//                         -> bboxes_nms_batch (:449-506)
//
// One workgroup per (image, class): class score, mask, clip, filter, order of the valid keys (top 2*topk), the
// per-class NMS (column bits + fixed-point resolve, nms_pairs.h) all stay in LDS; output is the reference's zero-padded
// per-class (scores[topk], boxes[topk,4]).
#include "common.h"
#include "nms_pairs.h"
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

// softmax probability of class c of one ROI's logits, and whether any logit is non-finite (tf.nn.softmax, eval.py:265)
__device__ __forceinline__ void class_probs_begin(const float* __restrict__ lg, int num_classes, float* m_out, float* sum_out,
                                                  bool* bad) {
  float m = lg[0];
  float lsum = lg[0];
  for (int k = 1; k < num_classes; ++k) { m = fmaxf(m, lg[k]); lsum += lg[k]; }
  *bad = !(fabsf(lsum) <= FLT_MAX);
  float sum = 0.f;
  for (int k = 0; k < num_classes; ++k) sum += expf(lg[k] - m);
  *m_out = m;
  *sum_out = sum;
}

// A11 + the class probabilities A12 starts from, once per ROI (the whole forward): bboxes_eval runs one workgroup per
// (image, class), and each of the 20 used to redo the 21-way softmax of every ROI from row-strided logits -- 430 of the
// kernel's 580 us per 128 images at R = 1000.  probs is class-major [N][num_classes][R]: the class workgroup reads its
// column coalesced.  A non-finite logit marks the image in `bad` (bboxes_eval makes it a NaN the host raises on).
__global__ void head_decode_probs_kernel(const float* __restrict__ rois, const float* __restrict__ cls_reg, int ld, int num_classes,
                                         int R, int64_t n, float* __restrict__ out, float* __restrict__ probs,
                                         int* __restrict__ bad) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 r = *reinterpret_cast<const float4*>(rois + i * 4);
    const float* lg = cls_reg + i * ld;
    const float* p = lg + num_classes;
    const float href = r.z - r.x, wref = r.w - r.y;
    const float yref = r.x + href / 2.f, xref = r.y + wref / 2.f;
    const float ph = expf(p[2]) * href;
    const float pw = expf(p[3]) * wref;
    const float pcy = p[0] * href + yref;
    const float pcx = p[1] * wref + xref;
    *reinterpret_cast<float4*>(out + i * 4) =
        make_float4(pcy - ph / 2.f, pcx - pw / 2.f, pcy + ph / 2.f, pcx + pw / 2.f);
    const int64_t img = i / R;
    const int ri = (int)(i - img * R);
    if (num_classes <= 32) {
      // the exponentials once, in registers (class_probs_begin's values in class_probs_begin's order: the same bits)
      float e[32];
      float m = lg[0], lsum = lg[0];
#pragma unroll
      for (int k = 1; k < 32; ++k)
        if (k < num_classes) { m = fmaxf(m, lg[k]); lsum += lg[k]; }
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k)
        if (k < num_classes) { e[k] = expf(lg[k] - m); sum += e[k]; }
#pragma unroll
      for (int k = 0; k < 32; ++k)
        if (k < num_classes) probs[(img * num_classes + k) * R + ri] = e[k] / sum;
      if (!(fabsf(lsum) <= FLT_MAX)) bad[img] = 1;
    } else {
      float m, sum;
      bool isbad;
      class_probs_begin(lg, num_classes, &m, &sum, &isbad);
      for (int k = 0; k < num_classes; ++k) probs[(img * num_classes + k) * R + ri] = expf(lg[k] - m) / sum;
      if (isbad) bad[img] = 1;
    }
  }
}

// The same pass with 32 lanes per ROI (num_classes + 4 <= 32: the light head's 21 + 4): a lane owns one column of the ROI's
// row -- ONE coalesced 128-byte read instead of 25 row-strided ones -- the maximum is a butterfly over the 32 lanes, and the
// sums run over the lanes' values IN CLASS ORDER (every lane adds the same 21 shuffled values: class_probs_begin's order,
// the same bits).  A single image's 300 ROIs are 38 workgroups instead of 2 threads-per-ROI workgroups: 12 -> ~5 us on the
// critical path of a single-image forward.
__global__ __launch_bounds__(256) void head_decode_probs_lanes_kernel(const float* __restrict__ rois, const float* __restrict__ cls_reg,
                                                                      int ld, int num_classes, int R, int64_t n,
                                                                      float* __restrict__ out, float* __restrict__ probs,
                                                                      int* __restrict__ bad) {
  const int lane = threadIdx.x & 63, k = lane & 31, gbase = lane & 32;
  for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 5; i < n; i += ((int64_t)gridDim.x * blockDim.x) >> 5) {
    const float x = k < num_classes + 4 ? cls_reg[i * ld + k] : 0.f;
    float m = k < num_classes ? x : -INFINITY;
#pragma unroll
    for (int d = 16; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));       // (all 32 lanes of the group end with the maximum)
    const float e = expf(x - m);
    float sum = 0.f, lsum = 0.f;
    for (int c = 0; c < num_classes; ++c) {
      sum += __shfl(e, gbase + c);
      lsum = c == 0 ? __shfl(x, gbase) : lsum + __shfl(x, gbase + c);
    }
    const int64_t img = i / R;
    const int ri = (int)(i - img * R);
    if (k < num_classes) probs[(img * num_classes + k) * R + ri] = e / sum;
    const float p0 = __shfl(x, gbase + num_classes), p1 = __shfl(x, gbase + num_classes + 1);
    const float p2 = __shfl(x, gbase + num_classes + 2), p3 = __shfl(x, gbase + num_classes + 3);
    if (k == 0) {
      if (!(fabsf(lsum) <= FLT_MAX)) bad[img] = 1;
      const float4 r = *reinterpret_cast<const float4*>(rois + i * 4);
      const float href = r.z - r.x, wref = r.w - r.y;
      const float yref = r.x + href / 2.f, xref = r.y + wref / 2.f;
      const float ph = expf(p2) * href;
      const float pw = expf(p3) * wref;
      const float pcy = p0 * href + yref;
      const float pcx = p1 * wref + xref;
      *reinterpret_cast<float4*>(out + i * 4) = make_float4(pcy - ph / 2.f, pcx - pw / 2.f, pcy + ph / 2.f, pcx + pw / 2.f);
    }
  }
}

int launch_head_decode_probs(const float* rois, const float* cls_reg, int ld, int num_classes, int R, int64_t n, float* out,
                             float* probs, int* bad, hipStream_t s) {
  if (n == 0) return XDET_OK;
  if (num_classes + 4 <= 32 && ld >= num_classes + 4)
    hipLaunchKernelGGL(head_decode_probs_lanes_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n * 32, 256), 16384)), dim3(256), 0, s,
                       rois, cls_reg, ld, num_classes, R, n, out, probs, bad);
  else
    hipLaunchKernelGGL(head_decode_probs_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 2048)), dim3(256), 0, s, rois,
                       cls_reg, ld, num_classes, R, n, out, probs, bad);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

int launch_ext_decode_rois(const float* rois, const float* reg, int ld_reg, int64_t n, float* out, hipStream_t s) {
  if (n == 0) return XDET_OK;
  hipLaunchKernelGGL(ext_decode_rois_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 2048)), dim3(256), 0, s,
                     rois, reg, ld_reg, n, out);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

constexpr int EV_MAXR = 1024;   // ROIs per image supported by one workgroup
constexpr int EV_MAXS = 512;    // 2*nms_topk upper bound (sorted candidates)
constexpr int EV_W = EV_MAXS / 64;
constexpr int EV_T = 1024;      // threads per workgroup

// grid (num_classes-1, N), EV_T threads.  One workgroup per (image, class) is all the parallelism a single image offers
// (20 workgroups on 256 CUs), so the workgroup is as wide as it can be: the rank sort and the IoU mask are spread over
// 16 waves instead of 4 (single-image latency; at large batches the total work is what counts and is unchanged).
// PRE: `cls` holds the class-major probabilities [N][num_classes][R] instead of the logits rows
template <bool PRE>
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
  __shared__ u64 s_keepw[2 * EV_W];
  // (aliases, so that two workgroups still share a CU: the sorted boxes with min/max-normalised corners -- what
  //  nms_pair_bits takes -- and their areas live in `bx`, dead once the sorted copies exist; the rank counters in `mask`,
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

  int local_bad = bad_per_image ? bad_per_image[n] : 0;     // (proposal stage: non-finite RPN outputs)
  for (int r0 = 0; r0 < R; r0 += EV_T) {
    const int r = r0 + tid;
    bool valid = false;
    float s = 0.f;
    if (r < R) {
      if (PRE) {
        s = cls[((int64_t)n * num_classes + c) * R + r];        // class-major probabilities (head_decode_probs_kernel)
      } else {
        // A non-finite logit (an activation that left the f16 range of the split-precision convs upstream, or a broken
        // checkpoint) would otherwise vanish here: its score compares false against the threshold and the image simply
        // has no detections.  It is made loud instead: the (image, class) slot is marked NaN below, the host raises.
        const float* lg = cls + ((int64_t)n * R + r) * ld_cls;
        float m, sum;
        bool isbad;
        class_probs_begin(lg, num_classes, &m, &sum, &isbad);
        local_bad |= isbad;
        s = expf(lg[c] - m) / sum;
      }
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
      valid = ws > min_size && hs > min_size && xc > 0.f && yc > 0.f && xc < 1.f && yc < 1.f && s > 0.f;
      // bboxes_resize
      bx[r] = make_float4((ymin - ref.x) / sx, (xmin - ref.y) / sy, (ymax - ref.x) / sx, (xmax - ref.y) / sy);
    }
    // Only the ROIs that pass (score above the class threshold: a few per cent of them) go on: their keys are packed at
    // the front of `keys` (in any order -- the rank below is by key, and the key carries the ROI index), so the rank sort
    // is V x V comparisons instead of R x R: at the reference's R = 1000 (light_head_rfcn_eval.py:111) the R x R form
    // was 11x the R = 300 work in each of the 20 x N workgroups.
    const u64 vb = __ballot(valid);
    int base = 0;
    if ((tid & 63) == 0 && vb) base = atomicAdd(&s_nvalid, __popcll(vb));
    base = __builtin_amdgcn_readfirstlane(base);
    if (valid)
      keys[base + __popcll(vb & ((1ull << (tid & 63)) - 1ull))] =
          ((u64)__float_as_uint(s) << 32) | (u64)(0xFFFFFFFFu - (unsigned)r);
  }
  if (local_bad) atomicOr(&s_bad, 1);
  __syncthreads();

  const int V = s_nvalid;
  const int max_sorted = min(2 * nms_topk, EV_MAXS);
  const int n_sorted = min(V, max_sorted);                   // bboxes_sort: top_k(min(n, 2*topk))
  // Order of the valid keys (descending score, ties -> lower ROI index: the keys are distinct).  Few keys: rank counting, the
  // V x V comparisons spread over the whole workgroup (P threads per key count a slice of the keys each).  Many (the
  // reference's R = 1000 puts up to ~1000 ROIs above a class threshold: 1 M comparisons, the longest phase of the class that
  // decides a single image's latency): a bitonic sort of the packed keys in place, 55 compare-exchange stages for 1024 keys,
  // after which a key's rank is its position.  Same order either way.
  const bool sorted_in_place = V > 512;      // (same-box A/B: at R = 300, V <= 298, the sort is slower than counting: 1.008 -> 1.020 ms)
  if (sorted_in_place) {
    constexpr int P = EV_MAXR;                           // 512 < V <= 1024
    for (int i = V + tid; i < P; i += EV_T) keys[i] = 0ull;
    __syncthreads();
    for (int kk = 2; kk <= P; kk <<= 1)
      for (int j = kk >> 1; j > 0; j >>= 1) {
        if (tid < (P >> 1)) {
          const int i = ((tid & ~(j - 1)) << 1) | (tid & (j - 1));
          const int o = i + j;
          const u64 a = keys[i], b = keys[o];
          const bool desc = (i & kk) == 0;
          if (desc ? a < b : a > b) { keys[i] = b; keys[o] = a; }
        }
        __syncthreads();
      }
  } else {
    for (int r = tid; r < V; r += EV_T) s_rank[r] = 0;
    __syncthreads();
    if (V > 0) {
      const int P = max(1, EV_T / V);
      const int slice = (V + P - 1) / P;
      for (int t = tid; t < V * P; t += EV_T) {
        const int part = t / V, r = t - part * V;
        const u64 mine = keys[r];
        const int j0 = part * slice, j1 = min(V, j0 + slice);
        int cnt = 0;
#pragma unroll 8
        for (int j = j0; j < j1; ++j) cnt += keys[j] > mine;
        if (cnt) atomicAdd(&s_rank[r], cnt);
      }
    }
    __syncthreads();
  }
  {
    // V <= R <= EV_T: one key per thread.  Read everything first: the sorted copies overwrite `bx`
    const bool have = tid < V;
    const u64 mine = have ? keys[tid] : 0ull;
    const int rank = have ? (sorted_in_place ? tid : s_rank[tid]) : max_sorted;
    const float4 b = have ? bx[0xFFFFFFFFu - (unsigned)mine] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (rank < max_sorted) {
      sbox[rank] = b;
      sscore[rank] = __uint_as_float((unsigned)(mine >> 32));
      const float y0 = fminf(b.x, b.z), y1 = fmaxf(b.x, b.z), x0 = fminf(b.y, b.w), x1 = fmaxf(b.y, b.w);
      snorm[rank] = make_float4(y0, x0, y1, x1);
      const float ar = (y1 - y0) * (x1 - x0);
      sarea[rank] = ar > 0.f ? ar : __builtin_inff();
    }
  }
  __syncthreads();

  // Per-class NMS as the proposal stage does it (proposals.hip nms_panel_kernel; here the whole class is one panel of up to
  // EV_MAXS candidates): (i) the strict upper triangle of "IoU(i, j) > thr" as COLUMN bits -- a lane holds candidate j and
  // walks the 64 boxes of row block I broadcast from LDS (nms_pairs.h: ~16 VALU operations per pair), the 64 x 64 blocks dealt
  // round-robin to the 16 waves; (ii) the keep set as the fixed point of keep_j = ok_j & !(col_j & keep) -- unique, the greedy
  // one, a few rounds of ballots -- instead of a serial walk over the kept boxes; (iii) the first nms_topk of it.
  // (Rounds 3-5: row words by ballot per (row, column block) and an ordered scan by one wave: 31 + 14 us of the 60 us this
  //  kernel took for the class that decides a single image's latency at R = 1000.)
  const int w64 = (n_sorted + 63) / 64;
  const float thr_hi = nms_thr * 1.00001f, thr_lo = nms_thr * 0.99999f;
  u64* const colL = mask;                                   // [w64 (w64 + 1) / 2][64]
  u64* const Kb = s_keepw;                                // keep words of the current / next round: [2][EV_W]
  {
    const int wv = tid >> 6, ln = tid & 63;
    const int n_unit = w64 * (w64 + 1) / 2;
    for (int v = wv; v < n_unit; v += EV_T / 64) {
      int J = 0;
      while ((J + 1) * (J + 2) / 2 <= v) ++J;
      const int I = v - J * (J + 1) / 2;
      const int col = J * 64 + ln;
      const bool col_ok = col < n_sorted;
      const float4 me = col_ok ? snorm[col] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float ma = col_ok ? sarea[col] : __builtin_inff();
      u64 bits = nms_pair_bits(snorm + I * 64, sarea + I * 64, min(64, n_sorted - I * 64), me, ma, nms_thr, thr_hi, thr_lo);
      if (I == J) bits &= (1ull << ln) - 1ull;              // only earlier candidates suppress
      colL[v * 64 + ln] = bits;
    }
  }
  __syncthreads();
  {
    const int ln = tid & 63, Jm = tid >> 6;                 // thread = candidate (tid < EV_MAXS), wave = its 64-block
    const bool ok = tid < n_sorted;
    u64 col[EV_W];
#pragma unroll
    for (int I = 0; I < EV_W; ++I) col[I] = (ok && I <= Jm) ? colL[(Jm * (Jm + 1) / 2 + I) * 64 + ln] : 0ull;
    {
      const u64 k0 = __ballot(ok);
      if (ln == 0 && Jm < EV_W) Kb[Jm] = k0;
    }
    __syncthreads();
    int cur = 0;
    for (int round = 0; round <= EV_MAXS; ++round) {
      u64 acc = 0ull;
#pragma unroll
      for (int I = 0; I < EV_W; ++I) acc |= col[I] & Kb[cur * EV_W + I];
      const bool nk = ok && acc == 0ull;
      const u64 kw = __ballot(nk);
      const u64 old = Jm < EV_W ? Kb[cur * EV_W + Jm] : 0ull;
      if (ln == 0 && Jm < EV_W) Kb[(cur ^ 1) * EV_W + Jm] = kw;
      cur ^= 1;
      if (!__syncthreads_or(Jm < EV_W && kw != old)) break;
    }
    int before = 0, total = 0;
#pragma unroll
    for (int I = 0; I < EV_W; ++I) {
      const int cnt = __popcll(Kb[cur * EV_W + I]);
      before += I < Jm ? cnt : 0;
      total += cnt;
    }
    if (Jm < EV_W && ((Kb[cur * EV_W + Jm] >> ln) & 1ull)) {
      const int slot = before + __popcll(Kb[cur * EV_W + Jm] & ((1ull << ln) - 1ull));
      if (slot < nms_topk) s_keptidx[slot] = tid;
    }
    if (tid == 0) s_nkeep = min(total, nms_topk);
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

static int launch_bboxes_eval_any(bool pre, const float* cls, int ld_cls, const float* boxes, int N, int R, int num_classes,
                                  const int* image_shapes, const float* bbox_img, int net_h, int net_w, float select_thr,
                                  float nms_thr, int nms_topk, float* det_scores, float* det_boxes, hipStream_t s,
                                  const int* bad_per_image) {
  XDET_REQUIRE(R > 0 && R <= EV_MAXR, "bboxes_eval: 1 <= rois per image <= 1024");
  XDET_REQUIRE(nms_topk > 0 && 2 * nms_topk <= EV_MAXS, "bboxes_eval: nms_topk must be in 1..256");
  XDET_REQUIRE(num_classes >= 2 && (pre || num_classes <= ld_cls), "bboxes_eval: bad num_classes");
  XDET_REQUIRE(nms_thr >= 0.f, "bboxes_eval: the NMS threshold must be >= 0");
  if (N == 0) return XDET_OK;
  if (pre)
    hipLaunchKernelGGL(bboxes_eval_kernel<true>, dim3(num_classes - 1, N), dim3(EV_T), 0, s, cls, ld_cls, boxes, R, num_classes,
                       image_shapes, bbox_img, net_h, net_w, select_thr, nms_thr, nms_topk, det_scores, det_boxes, bad_per_image);
  else
    hipLaunchKernelGGL(bboxes_eval_kernel<false>, dim3(num_classes - 1, N), dim3(EV_T), 0, s, cls, ld_cls, boxes, R, num_classes,
                       image_shapes, bbox_img, net_h, net_w, select_thr, nms_thr, nms_topk, det_scores, det_boxes, bad_per_image);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

int launch_bboxes_eval(const float* cls, int ld_cls, const float* boxes, int N, int R, int num_classes,
                       const int* image_shapes, const float* bbox_img, int net_h, int net_w, float select_thr,
                       float nms_thr, int nms_topk, float* det_scores, float* det_boxes, hipStream_t s,
                       const int* bad_per_image) {
  return launch_bboxes_eval_any(false, cls, ld_cls, boxes, N, R, num_classes, image_shapes, bbox_img, net_h, net_w, select_thr,
                                nms_thr, nms_topk, det_scores, det_boxes, s, bad_per_image);
}

int launch_bboxes_eval_probs(const float* probs, const float* boxes, int N, int R, int num_classes, const int* image_shapes,
                             const float* bbox_img, int net_h, int net_w, float select_thr, float nms_thr, int nms_topk,
                             float* det_scores, float* det_boxes, hipStream_t s, const int* bad_per_image) {
  return launch_bboxes_eval_any(true, probs, 0, boxes, N, R, num_classes, image_shapes, bbox_img, net_h, net_w, select_thr,
                                nms_thr, nms_topk, det_scores, det_boxes, s, bad_per_image);
}

}  // namespace xdet
