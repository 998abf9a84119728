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
//   1. prepare : clipped box + 64-bit sort key (score bits << 32 | ~index), 0 = filtered;
//                histogram of the keys' top 16 bits
//   2. select  : threshold bin T = the lowest bin that is still needed to cover pre_n keys
//   3. compact : keys with bin >= T (>= pre_n of them, typically barely more) -> candidate list
//   4. sort    : the candidates' keys, descending, in LDS (one workgroup per image; all keys distinct ->
//                tf.nn.top_k order: descending score, ties -> lower index) -> sorted_boxes[0..pre_n)
//      (lists longer than 8192 keys: rank_i = #{j : key_j > key_i} by counting, then scatter)
//   6. nms     : one workgroup per image walks the candidates in score order 64 at a time; a chunk is
//                tested against the boxes kept so far (LDS) and against itself, resolved by one
//                wavefront; stops at post_n
//   7. gather  : rois[j] = kept[j mod n_keep]  (tile + identity "shuffle" of :196-213)
#include "common.h"
#include <cfloat>
#include <cstdlib>

namespace xdet {

typedef unsigned long long u64;

constexpr int HIST_BINS = 16384;     // key >> 48 = score float bits >> 16 (scores are in (0, 1])

size_t proposal_workspace_bytes(int N, int n_anchor, int pre_n, int post_n) {
  size_t b = 0;
  auto add = [&](size_t x) { b += (x + 255) / 256 * 256; };
  add((size_t)N * n_anchor * 8);
  add((size_t)N * n_anchor * 16);
  add((size_t)N * n_anchor * 4);
  add((size_t)N * 16);
  add((size_t)N * pre_n * 16);
  add((size_t)N * pre_n * 4);
  add((size_t)N * n_anchor * 8);
  add(((size_t)N * HIST_BINS + (size_t)round_up(N, 4)) * 4);   // hist + bad (zeroed together)
  add((size_t)N * 4);
  add((size_t)N * post_n * 4);
  return b;
}

void proposal_workspace_carve(void* base, int N, int n_anchor, int pre_n, int post_n, ProposalWorkspace* ws) {
  char* p = static_cast<char*>(base);
  auto take = [&](size_t x) { char* r = p; p += (x + 255) / 256 * 256; return r; };
  ws->keys = reinterpret_cast<u64*>(take((size_t)N * n_anchor * 8));
  ws->cboxes = reinterpret_cast<float*>(take((size_t)N * n_anchor * 16));
  ws->ranks = reinterpret_cast<int*>(take((size_t)N * n_anchor * 4));
  ws->counts = reinterpret_cast<int*>(take((size_t)N * 16));
  ws->sboxes = reinterpret_cast<float*>(take((size_t)N * pre_n * 16));
  ws->sscores = reinterpret_cast<float*>(take((size_t)N * pre_n * 4));
  ws->cand = reinterpret_cast<u64*>(take((size_t)N * n_anchor * 8));
  ws->hist = reinterpret_cast<int*>(take(((size_t)N * HIST_BINS + (size_t)round_up(N, 4)) * 4));
  ws->bad = ws->hist + (size_t)N * HIST_BINS;
  ws->tbin = reinterpret_cast<int*>(take((size_t)N * 4));
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
                                    int* __restrict__ hist, int* __restrict__ bad) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_anchor) return;
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
  // A non-finite score or box compares false below and would just drop out: the image would quietly lose proposals
  // (all of them if the backbone overflowed).  Remembered per image; bboxes_eval turns it into a NaN the host raises on.
  if (!(fabsf(sc) <= FLT_MAX) || !(fabsf(b.x + b.y + b.z + b.w) <= FLT_MAX)) bad[n] = 1;
  // a score <= 0 survives top_k in the reference but is dropped as padding by
  // _upsample_rois (:199-200); such entries sort last, so excluding them here is equivalent.
  const bool valid = ws > min_size && hs > min_size && xc > 0.f && yc > 0.f && xc < 1.f && yc < 1.f && sc > 0.f;
  const u64 key = valid ? (((u64)__float_as_uint(sc) << 32) | (u64)(0xFFFFFFFFu - (unsigned)i)) : 0ull;
  keys[g] = key;
  if (valid) atomicAdd(&hist[n * HIST_BINS + min((int)(key >> 48), HIST_BINS - 1)], 1);
}

// one 1024-thread workgroup per image: lowest histogram bin still needed to cover pre_n keys
// clears the key histogram and the (zero-padded) sorted box / score arrays of one forward
__global__ void prop_zero_kernel(uint4* __restrict__ a, int64_t na, uint4* __restrict__ b, int64_t nb,
                                 unsigned* __restrict__ c, int64_t nc) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (int64_t i = t0; i < na; i += stride) a[i] = z;
  for (int64_t i = t0; i < nb; i += stride) b[i] = z;
  for (int64_t i = t0; i < nc; i += stride) c[i] = 0u;
}

__global__ __launch_bounds__(1024) void prop_select_kernel(const int* __restrict__ hist, int pre_n,
                                                           int* __restrict__ tbin, int* __restrict__ counts) {
  __shared__ int part[1024];
  const int n = blockIdx.x, t = threadIdx.x;
  const int* h = hist + n * HIST_BINS;
  constexpr int PER = HIST_BINS / 1024;
  int s = 0;
  for (int b = 0; b < PER; ++b) s += h[t * PER + b];
  part[t] = s;
  __syncthreads();
  int above = 0;                                   // keys in bins owned by threads > t
  for (int u = t + 1; u < 1024; ++u) above += part[u];
  if (t == 0) {
    const int n_valid = above + s;
    counts[n * 4 + 0] = n_valid;
    counts[n * 4 + 1] = min(n_valid, pre_n);       // n_cand
    counts[n * 4 + 2] = 0;
    counts[n * 4 + 3] = 0;                         // candidate-list length, filled by compact
    if (n_valid < pre_n) tbin[n] = 0;
  }
  if (above < pre_n && above + s >= pre_n) {       // the crossing lies in this thread's bins
    int acc = above;
    for (int b = PER - 1; b >= 0; --b) {
      acc += h[t * PER + b];
      if (acc >= pre_n) { tbin[n] = t * PER + b; break; }
    }
  }
}

__global__ void prop_compact_kernel(const u64* __restrict__ keys, int n_anchor, const int* __restrict__ tbin,
                                    u64* __restrict__ cand, int* __restrict__ ranks, int* __restrict__ counts) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const u64 key = i < n_anchor ? keys[(int64_t)n * n_anchor + i] : 0ull;
  const bool take = key != 0ull && min((int)(key >> 48), HIST_BINS - 1) >= tbin[n];
  const u64 ball = __ballot(take);
  if (!ball) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == (__ffsll((long long)ball) - 1)) base = atomicAdd(&counts[n * 4 + 3], __popcll(ball));
  base = __shfl(base, __ffsll((long long)ball) - 1);
  if (take) {
    const int slot = base + __popcll(ball & ((1ull << lane) - 1ull));
    cand[(int64_t)n * n_anchor + slot] = key;
    ranks[(int64_t)n * n_anchor + slot] = 0;
  }
}

constexpr int RANK_TILE = 1024;
constexpr int RANK_IPT = 4;      // keys owned per thread (amortises the LDS broadcast reads)
constexpr int RANK_SPLITS = 8;

// grid (ceil(n_anchor/1024), RANK_SPLITS, N): each thread owns 4 candidate keys and counts larger
// keys in its slice of the candidate list; workgroups beyond the list length exit at once
// Steps 4+5 for the usual case (candidate list <= SORT_MAX keys): one workgroup per image sorts the
// 64-bit keys in LDS (bitonic, descending; the keys are distinct, so the order is exactly the rank the
// counting kernel below computes) and writes the first pre_n boxes / scores.  ~90 compare-exchange
// passes over <= 8192 keys on ONE CU per image instead of an O(L^2) count spread over the chip: ~20x
// less CU time next to the large-separable convs it runs beside.  Longer lists (many keys sharing the
// threshold bin) fall through to prop_rank / prop_scatter, which exit early otherwise.
constexpr int SORT_MAX = 8192;
__global__ __launch_bounds__(1024) void prop_sort_kernel(const u64* __restrict__ cand,
                                                         const float* __restrict__ cboxes, int n_anchor, int pre_n,
                                                         const int* __restrict__ counts, float* __restrict__ sboxes,
                                                         float* __restrict__ sscores, int sort_max) {
  __shared__ u64 keys[SORT_MAX];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int L = counts[n * 4 + 3];
  if (L > sort_max || L <= 0) return;
  int P = 64;
  while (P < L) P <<= 1;
  const u64* k = cand + (int64_t)n * n_anchor;
  for (int i = tid; i < max(P, 512); i += 1024) keys[i] = i < L ? k[i] : 0ull;   // (a wave's block is 512 keys)
  __syncthreads();
  // Bitonic network, wave-synchronous where it can be.  Wave w owns the 512 keys [512 w, 512 w + 512): every
  // compare-exchange with a stride below 512 stays inside one wave's block, and the LDS operations of one wave execute
  // in order -- those stages need no workgroup barrier at all, only the compiler kept from reordering across them.  Of the
  // 91 stages of an 8192-key sort, 81 are of that kind; the barriers drop from 91 to 15: 139 -> 67 us per image (on the
  // critical path of a single-image forward).  Measured and dropped: three strides per LDS pass with eight keys per thread
  // in registers (35 passes instead of 91) -- 92 us: a thread's eight keys are 64 B apart for the small strides, a 16-way
  // bank conflict on every ds_read_b64 / ds_write_b64 of a third of the passes.
  const int lane = tid & 63, base = (tid >> 6) * 512;
  auto wave_stages = [&](int kk, int j_first) {          // strides j_first, j_first / 2, ..., 1 of merge size kk
    if (base < P) {
      for (int j = j_first; j > 0; j >>= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int pr = lane + 64 * q;                  // pair index inside the block
          const int i = base + (((pr & ~(j - 1)) << 1) | (pr & (j - 1)));
          const int o = i + j;
          const u64 a = keys[i], b = keys[o];
          const bool desc = (i & kk) == 0;
          if (desc ? a < b : a > b) { keys[i] = b; keys[o] = a; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  };
  for (int kk = 2; kk <= min(P, 512); kk <<= 1) wave_stages(kk, kk >> 1);
  __syncthreads();
  for (int kk = 1024; kk <= P; kk <<= 1) {
    for (int j = kk >> 1; j >= 512; j >>= 1) {
      for (int i = tid; i < P; i += 1024) {
        const int o = i ^ j;
        if (o > i) {
          const u64 a = keys[i], b = keys[o];
          const bool desc = (i & kk) == 0;
          if (desc ? a < b : a > b) { keys[i] = b; keys[o] = a; }
        }
      }
      __syncthreads();
    }
    wave_stages(kk, 256);
    __syncthreads();
  }
  const int m = min(L, pre_n);
  for (int r = tid; r < m; r += 1024) {
    const u64 key = keys[r];
    const unsigned idx = 0xFFFFFFFFu - (unsigned)key;
    *reinterpret_cast<float4*>(sboxes + ((int64_t)n * pre_n + r) * 4) =
        *reinterpret_cast<const float4*>(cboxes + ((int64_t)n * n_anchor + idx) * 4);
    sscores[(int64_t)n * pre_n + r] = __uint_as_float((unsigned)(key >> 32));
  }
}

__global__ __launch_bounds__(256) void prop_rank_kernel(const u64* __restrict__ cand, int n_anchor,
                                                        const int* __restrict__ counts, int* __restrict__ ranks, int sort_max) {
  __shared__ u64 tile[RANK_TILE];
  const int n = blockIdx.z;
  const int c = counts[n * 4 + 3];
  if (c <= sort_max) return;                       // handled by prop_sort_kernel
  const int i0 = blockIdx.x * (256 * RANK_IPT);
  const int jps = ((c + RANK_SPLITS - 1) / RANK_SPLITS + RANK_TILE - 1) / RANK_TILE * RANK_TILE;
  const int j0 = blockIdx.y * jps;
  if (i0 >= c || j0 >= c) return;
  const int j1 = min(j0 + jps, c);
  const u64* k = cand + (int64_t)n * n_anchor;
  u64 mine[RANK_IPT];
  int cnt[RANK_IPT];
#pragma unroll
  for (int q = 0; q < RANK_IPT; ++q) {
    const int i = i0 + q * 256 + threadIdx.x;
    mine[q] = i < c ? k[i] : ~0ull;
    cnt[q] = 0;
  }
  for (int jb = j0; jb < j1; jb += RANK_TILE) {
    __syncthreads();
    for (int t = threadIdx.x; t < RANK_TILE; t += 256) tile[t] = (jb + t < j1) ? k[jb + t] : 0ull;
    __syncthreads();
#pragma unroll 4
    for (int t = 0; t < RANK_TILE; ++t) {
      const u64 o = tile[t];
#pragma unroll
      for (int q = 0; q < RANK_IPT; ++q) cnt[q] += o > mine[q];
    }
  }
#pragma unroll
  for (int q = 0; q < RANK_IPT; ++q) {
    const int i = i0 + q * 256 + threadIdx.x;
    if (i < c && cnt[q]) atomicAdd(&ranks[(int64_t)n * n_anchor + i], cnt[q]);
  }
}

__global__ void prop_scatter_kernel(const u64* __restrict__ cand, const int* __restrict__ ranks,
                                    const float* __restrict__ cboxes, int n_anchor, int pre_n,
                                    const int* __restrict__ counts, float* __restrict__ sboxes,
                                    float* __restrict__ sscores, int sort_max) {
  const int n = blockIdx.y;
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (counts[n * 4 + 3] <= sort_max || slot >= counts[n * 4 + 3]) return;
  const int64_t g = (int64_t)n * n_anchor + slot;
  const int r = ranks[g];
  if (r >= pre_n) return;
  const u64 key = cand[g];
  const unsigned idx = 0xFFFFFFFFu - (unsigned)key;
  *reinterpret_cast<float4*>(sboxes + ((int64_t)n * pre_n + r) * 4) =
      *reinterpret_cast<const float4*>(cboxes + ((int64_t)n * n_anchor + idx) * 4);
  sscores[(int64_t)n * pre_n + r] = __uint_as_float((unsigned)(key >> 32));
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

// value of lane `src` (wave-uniform) as a scalar: v_readlane_b32, a few cycles, where __shfl() is a ds_bpermute round
// trip through the LDS crossbar (~50 cycles) -- the serial resolve below is a 64-step dependent chain of these, and it,
// not the IoU tests, was what the kernel's time went into (124 us per image on the single-image critical path)
__device__ __forceinline__ u64 readlane_u64(u64 v, int src) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 readfirstlane_u64(u64 v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((u64)hi << 32) | lo;
}

// IoU(a,b) > thr with the reference's rounding, but without a division on the fast path: the
// correctly rounded quotient can only disagree with a product test inside a 1e-5 relative band
// around the threshold; only there is the division actually evaluated.
__device__ __forceinline__ bool iou_gt_fast(const float4 a, const float4 b, float thr) {
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

// ---------------------------------------------------------------------------------------
// Greedy NMS against the KEPT list (one workgroup of NMS_WAVES waves per image).  tf.image.non_max_suppression
// only ever compares a candidate with boxes that were kept before it, and stops at max_output_size: with
// R = 300 kept boxes that is <= 64 x 300 IoUs per 64-candidate chunk and a few dozen chunks.  (The first
// version built the full upper-triangular IoU bit matrix chip-wide -- 5000^2 / 2 IoUs per image, 200 MB of
// mask for 64 images -- and scanned it: ~25x the CU time.)  Per chunk of 64 candidates (score order):
//   1. every wave tests the chunk against a strided 1/NMS_WAVES of the kept boxes (kept box broadcast from
//      LDS, candidate b in lane b), one ballot per wave -> bits of candidates already suppressed;
//   2. the 64 x 64 intra-chunk matrix, NMS_WAVES column-strided parts OR-ed through LDS;
//   3. wave 0 resolves the chunk serially (wave-uniform shuffles) and appends the survivors to the
//      kept list.
// Comparisons are iou_gt_fast(earlier, later, thr), strict >, visited in score order.
// ---------------------------------------------------------------------------------------
constexpr int NMS_WAVES = 16;   // the kept-list test of a chunk (64 candidates x n_keep boxes) is split over this many waves
__global__ __launch_bounds__(64 * NMS_WAVES) void nms_greedy_kernel(const float* __restrict__ sboxes, int* __restrict__ counts,
                                                         int pre_n, int post_n, float thr, int* __restrict__ kept) {
  extern __shared__ __attribute__((aligned(16))) unsigned char nmsg_smem[];
  float4* kbox = reinterpret_cast<float4*>(nmsg_smem);          // [post_n] boxes kept so far
  __shared__ float4 cb[64];
  __shared__ u64 part_sup[NMS_WAVES];
  __shared__ u64 part_diag[NMS_WAVES][64];
  __shared__ int s_keep;
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_cand = counts[n * 4 + 1];
  const float4* B = reinterpret_cast<const float4*>(sboxes) + (int64_t)n * pre_n;
  int* K = kept + (int64_t)n * post_n;
  if (tid == 0) s_keep = 0;
  __syncthreads();
  const int n_chunk = (n_cand + 63) / 64;
  for (int c = 0; c < n_chunk; ++c) {
    const int n_keep = s_keep;                               // uniform: only changes between barriers
    if (n_keep >= post_n) break;
    const int rows = min(64, n_cand - c * 64);
    if (tid < 64) cb[tid] = tid < rows ? B[c * 64 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const float4 me = cb[lane];
    // 1. against the kept list
    bool hit = false;
    if (lane < rows)
      for (int k = wave; k < n_keep; k += NMS_WAVES)
        if (iou_gt_fast(kbox[k], me, thr)) { hit = true; break; }
    const u64 hb = __ballot(hit);
    if (lane == 0) part_sup[wave] = hb;
    // 2. intra-chunk: does candidate `lane` suppress a later candidate j of the chunk?
    u64 bits = 0ull;
    if (lane < rows)
      for (int j = lane + 1 + wave; j < rows; j += NMS_WAVES)
        if (iou_gt_fast(me, cb[j], thr)) bits |= 1ull << j;
    part_diag[wave][lane] = bits;
    __syncthreads();
    // 3. serial resolve of the chunk
    if (wave == 0) {
      u64 cur_v = 0ull, diag = 0ull;
#pragma unroll
      for (int w = 0; w < NMS_WAVES; ++w) { cur_v |= part_sup[w]; diag |= part_diag[w][lane]; }
      // the chain runs on the scalar unit: `cur`, `keepmask` and the row of candidate b are wave-uniform
      u64 cur = readfirstlane_u64(cur_v);
      u64 keepmask = 0ull;
      int kc = n_keep;
      const int rows_u = __builtin_amdgcn_readfirstlane(rows);
      for (int b = 0; b < rows_u && kc < post_n; ++b) {
        const u64 d = readlane_u64(diag, b);
        if (!((cur >> b) & 1ull)) { keepmask |= 1ull << b; cur |= d; ++kc; }
      }
      if ((keepmask >> lane) & 1ull) {
        const int slot = n_keep + __popcll(keepmask & ((1ull << lane) - 1ull));
        K[slot] = c * 64 + lane;
        kbox[slot] = me;
      }
      if (lane == 0) s_keep = kc;
    }
    __syncthreads();
  }
  if (tid == 0) counts[n * 4 + 2] = s_keep;
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
  // one launch instead of three hipMemsetAsync (which the runtime splits into ~12 fill kernels per forward)
  hipLaunchKernelGGL(prop_zero_kernel, dim3(512), dim3(256), 0, s, reinterpret_cast<uint4*>(ws.hist),
                     ((int64_t)N * HIST_BINS + round_up(N, 4)) * 4 / 16, reinterpret_cast<uint4*>(ws.sboxes), (int64_t)N * pre_n,
                     reinterpret_cast<unsigned*>(ws.sscores), (int64_t)N * pre_n);
  XDET_LAUNCH_CHECK();
  const unsigned gb = (unsigned)cdiv(n_anchor, 256);
  hipLaunchKernelGGL(prop_prepare_kernel, dim3(gb, N), dim3(256), 0, s, objectness, boxes, n_anchor, min_size,
                     ws.keys, ws.cboxes, ws.hist, ws.bad);
  XDET_LAUNCH_CHECK();
  hipLaunchKernelGGL(prop_select_kernel, dim3(N), dim3(1024), 0, s, ws.hist, pre_n, ws.tbin, ws.counts);
  XDET_LAUNCH_CHECK();
  hipLaunchKernelGGL(prop_compact_kernel, dim3(gb, N), dim3(256), 0, s, ws.keys, n_anchor, ws.tbin, ws.cand, ws.ranks,
                     ws.counts);
  XDET_LAUNCH_CHECK();
  // The order of the (distinct) keys, two ways with the same result: a bitonic sort in LDS by ONE workgroup per image
  // (the cheapest in CU time: what a batch wants, its images sort side by side) or rank counting spread over the chip
  // (160 workgroups per image: what one or two images want -- the sort is ~70 us on one of 256 CUs, on the critical path
  // of a single-image forward).  Lists beyond the sort's LDS capacity always take the counting path.
  static const int small_n = getenv("XDET_PROP_RANK_N") ? atoi(getenv("XDET_PROP_RANK_N")) : 0;     // A/B knob: measured at one image, 1.302 ms against 1.296 with the sort -- the sort is not on the critical path -- so off
  const int sort_max = N <= small_n ? 0 : SORT_MAX;
  if (sort_max > 0) {
    hipLaunchKernelGGL(prop_sort_kernel, dim3(N), dim3(1024), 0, s, ws.cand, ws.cboxes, n_anchor, pre_n, ws.counts,
                       ws.sboxes, ws.sscores, sort_max);
    XDET_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(prop_rank_kernel, dim3((unsigned)cdiv(n_anchor, 256 * RANK_IPT), RANK_SPLITS, N), dim3(256), 0, s,
                     ws.cand, n_anchor, ws.counts, ws.ranks, sort_max);
  XDET_LAUNCH_CHECK();
  hipLaunchKernelGGL(prop_scatter_kernel, dim3(gb, N), dim3(256), 0, s, ws.cand, ws.ranks, ws.cboxes, n_anchor, pre_n,
                     ws.counts, ws.sboxes, ws.sscores, sort_max);
  XDET_LAUNCH_CHECK();
  {
    XDET_REQUIRE((size_t)post_n * 16 <= 96 * 1024, "get_proposals: rpn_post_nms_top_n too large (max 6144)");
    static DeviceOnce once;
    XDET_TRY(ensure_dynamic_lds(once, reinterpret_cast<const void*>(nms_greedy_kernel), 96 * 1024));
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(N), dim3(64 * NMS_WAVES), (size_t)post_n * 16, s, ws.sboxes, ws.counts, pre_n, post_n,
                       nms_thr, ws.kept);
    XDET_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(prop_gather_kernel, dim3((unsigned)cdiv(post_n, 256), N), dim3(256), 0, s, ws.sboxes, ws.kept,
                     ws.counts, pre_n, post_n, rois);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

}  // namespace xdet
