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
#include "nms_pairs.h"
#include <cfloat>
#include <cstdlib>

namespace xdet {

typedef unsigned long long u64;

constexpr int HIST_BINS = 16384;     // key >> 48 = score float bits >> 16 (scores are in (0, 1])

// proposal NMS (nms_panel_kernel below): panels of 64 * NBLK candidates, clusters of G workgroups per image
constexpr int NMS_THREADS = 1024, NMS_WAVES = NMS_THREADS / 64;
constexpr int NMS_SUP_PANEL = 256;            // the granularity ws.nms_sup is sized by (the smallest panel)
constexpr int NMS_MAX_CLUSTER = 16;
constexpr int NMS_LDS_MAX = 159 * 1024;      // dynamic LDS a workgroup may ask for (160 KB per CU less the static part)
constexpr int NMS_CLUSTER_NBLK = 8;           // a cluster's panel: 512 candidates
constexpr int NMS_CLUSTER_IMAGES = 64;        // only batches up to this many images ever run clusters (ws.nms_col is sized by it)
constexpr unsigned NMS_SPIN_LIMIT = 1u << 20; // ~1 s of polling: a cluster whose members are not co-resident gives up, loudly
template <int NBLK> struct NmsPanel {
  static constexpr int S = 64 * NBLK;                  // candidates per panel
  static constexpr int TRI = NBLK * (NBLK + 1) / 2;    // 64 x 64 blocks (I <= J) of the panel's triangle
};
// int words zeroed by prop_zero_kernel at the start of every forward: hist[N][HIST_BINS], bad[round_up(N,4)],
// nms_ctl[N][4] (cluster barrier counter, give-up flag), nms_sup[N][panels][8] as 64-bit words
static inline size_t prop_zeroed_words(int N, int pre_n) {
  return (size_t)N * HIST_BINS + (size_t)round_up(N, 4) + (size_t)N * 4 + (size_t)N * cdiv(pre_n, NMS_SUP_PANEL) * 16;
}

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
  add(prop_zeroed_words(N, pre_n) * 4);                        // hist + bad + the NMS cluster words (zeroed together)
  add((size_t)N * 4);
  add((size_t)N * post_n * 4);
  add((size_t)std::min(N, NMS_CLUSTER_IMAGES) * 3 * NmsPanel<NMS_CLUSTER_NBLK>::TRI * 64 * 8);
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
  ws->hist = reinterpret_cast<int*>(take(prop_zeroed_words(N, pre_n) * 4));
  ws->bad = ws->hist + (size_t)N * HIST_BINS;
  ws->nms_ctl = reinterpret_cast<unsigned*>(ws->bad + round_up(N, 4));
  ws->nms_sup = reinterpret_cast<u64*>(ws->nms_ctl + (size_t)N * 4);
  ws->tbin = reinterpret_cast<int*>(take((size_t)N * 4));
  ws->kept = reinterpret_cast<int*>(take((size_t)N * post_n * 4));
  ws->nms_col = reinterpret_cast<u64*>(take((size_t)std::min(N, NMS_CLUSTER_IMAGES) * 3 * NmsPanel<NMS_CLUSTER_NBLK>::TRI * 64 * 8));
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
struct ZeroRanges {
  static constexpr int MAX = 6;
  uint4* p[MAX];       // 16-byte aligned
  int64_t n[MAX];      // in uint4 units
};
__global__ void prop_zero_kernel(ZeroRanges z) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int r = 0; r < ZeroRanges::MAX; ++r)
    for (int64_t i = t0; i < z.n[r]; i += stride) z.p[r][i] = zero;
}

__global__ __launch_bounds__(1024) void prop_select_kernel(const int* __restrict__ hist, int pre_n,
                                                           int* __restrict__ tbin, int* __restrict__ counts) {
  __shared__ int wsum[16];
  const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int* h = hist + n * HIST_BINS;
  constexpr int PER = HIST_BINS / 1024;
  int s = 0;
  for (int b = 0; b < PER; ++b) s += h[t * PER + b];
  // keys in bins owned by threads > t: a suffix sum over the 1024 threads (inside the wave by shuffles, then over the 16 waves)
  int suf = s;                                     // inclusive suffix sum inside the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_down(suf, d);
    if (lane + d < 64) suf += o;
  }
  if (lane == 0) wsum[wave] = suf;
  __syncthreads();
  int above = suf - s;
  for (int w = wave + 1; w < 16; ++w) above += wsum[w];
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

// ---------------------------------------------------------------------------------------
// Greedy NMS (tf.image.non_max_suppression: visit in score order, keep a candidate unless a box kept before it has
// IoU > thr with it, stop at max_output_size), in PANELS of 64 * NBLK candidates:
//   A. every candidate of the panel against every box kept so far                  -> one "suppressed" bit per candidate
//   B. the panel against itself: column j of the strict upper triangle as bits     -> col_j = { i < j : IoU(i, j) > thr }
//   C. resolve: keep_j = ok_j & !sup_j & !(col_j & keep).  keep_j only depends on keep_i for i < j, so the equation has ONE
//      solution -- the greedy one -- and iterating it from any start fixes at least one more leading candidate per round:
//      a handful of rounds of (NBLK AND/ORs + one ballot) instead of a 64 * NBLK-step serial chain.
// A and B are one inner loop (a wave holds 64 candidates, one per lane, and walks 64 reference boxes broadcast from LDS,
// ~16 VALU operations per pair); their units of 64 x 64 pairs are dealt round-robin to every wave of the image's CLUSTER
// of G workgroups.  With G > 1 (few images: the proposal stage of a single image is on the critical path of the forward,
// and the reference's own operating point is rpn_post_nms_top_n = 1000 at batch 1, light_head_rfcn_eval.py:109-111,212)
// the workgroups exchange the panel's bits through global memory behind ONE cluster barrier per panel and then each
// resolves the panel redundantly, appending the same boxes to its own LDS copy of the kept list: no second barrier.
// The previous form -- one workgroup per image, 64 candidates per step -- spent 1,029 us per image at R = 1000 on the
// VALU of one CU (r05d_R1000_kernel_stats.csv).
// Pair decisions are those of NonMaxSuppressionV2: corners min/max normalised, IoU = 0 when either area <= 0, strict >.
// ---------------------------------------------------------------------------------------
__host__ __device__ inline size_t nms_lds_bytes(int post_n, int nblk) {
  const size_t p4 = (size_t)(post_n + 3) / 4 * 4;
  return p4 * 20 + (size_t)64 * nblk * 20 + (size_t)nblk * (nblk + 1) / 2 * 64 * 8 + 64 + 2 * NMS_WAVES * 8 + 16;
}

// Barrier of the G workgroups of one cluster.  Everything the cluster exchanges is written with agent-scope (write-through,
// `sc1`) stores / atomics and read back with agent-scope loads, so the barrier needs no fences (cdna_hip_programming.md
// Guideline 16, form R1: every storing wave drains its stores, ONE lane arrives on a monotonic counter and polls it relaxed):
// a release fence would write back every dirty line of the XCD's L2 -- the large-separable convs of the main stream are
// filling it while this runs -- and an acquire fence would drop the CU's L1 under a co-resident kernel.
// Returns false when the poll gave up (the other workgroups of the cluster never arrived).
__device__ __forceinline__ bool nms_cluster_barrier(unsigned* ctl, unsigned target, int* s_fail) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > NMS_SPIN_LIMIT || __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_fail = 1;
        break;
      }
    }
  }
  __syncthreads();
  return *s_fail == 0;
}

template <bool CLUSTER, int NBLK>
__global__ __launch_bounds__(NMS_THREADS) void nms_panel_kernel(const float* __restrict__ sboxes, int* __restrict__ counts,
                                                                int pre_n, int post_n, float thr, int* __restrict__ kept,
                                                                int G, u64* nms_sup, int n_sup, u64* nms_col,
                                                                unsigned* nms_ctl, int* __restrict__ bad) {
  constexpr int S = NmsPanel<NBLK>::S, TRI = NmsPanel<NBLK>::TRI;
  extern __shared__ __attribute__((aligned(16))) unsigned char nms_smem[];
  const int p4 = (post_n + 3) / 4 * 4;
  float4* kbox = reinterpret_cast<float4*>(nms_smem);                       // [p4]  boxes kept so far, normalised
  float4* cb = kbox + p4;                                                   // [S]   the panel's candidates, normalised
  u64* colL = reinterpret_cast<u64*>(cb + S);                               // [TRI][64]  (a single workgroup's exchange)
  u64* supL = colL + TRI * 64;                                              // [8]
  u64* Kb = supL + 8;                                                       // [2][NMS_WAVES] keep words of the resolve
  float* karea = reinterpret_cast<float*>(Kb + 2 * NMS_WAVES);              // [p4]
  float* ca = karea + p4;                                                   // [S]
  int& s_fail = *reinterpret_cast<int*>(ca + S);                           // (no static LDS: it would shift the dynamic base off 16 B)
  const int n = CLUSTER ? blockIdx.x / G : blockIdx.x, g = CLUSTER ? blockIdx.x % G : 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = (CLUSTER ? G : 1) * NMS_WAVES, t = g * NMS_WAVES + wave;   // the cluster's waves
  const int n_cand = __builtin_amdgcn_readfirstlane(counts[n * 4 + 1]);
  const float4* B = reinterpret_cast<const float4*>(sboxes) + (int64_t)n * pre_n;
  int* K = kept + (int64_t)n * post_n;
  u64* supG = nms_sup + (int64_t)n * n_sup * 8;
  u64* colG = CLUSTER ? nms_col + (int64_t)n * 3 * TRI * 64 : nullptr;
  unsigned* ctl = nms_ctl + n * 4;
  const float thr_hi = thr * 1.00001f, thr_lo = thr * 0.99999f;
  if (tid == 0) s_fail = 0;
  int n_keep = 0;                                            // uniform over the workgroup AND over the cluster
  for (int s = 0; s * S < n_cand && n_keep < post_n; ++s) {
    const int base = s * S, rows = min(S, n_cand - base), nb = (rows + 63) >> 6;
    if (tid < S) {
      const float4 b = tid < rows ? B[base + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 nbx = make_float4(fminf(b.x, b.z), fminf(b.y, b.w), fmaxf(b.x, b.z), fmaxf(b.y, b.w));
      cb[tid] = nbx;
      const float area = (nbx.z - nbx.x) * (nbx.w - nbx.y);
      ca[tid] = area > 0.f ? area : __builtin_inff();    // (a padding lane's zero box: never suppresses, never kept)
    }
    if (tid < 8) supL[tid] = 0ull;
    __syncthreads();
    // ---- A + B: units of (64 candidates) x (<= 64 reference boxes), round-robin over the cluster's waves
    const int n_slice = (n_keep + 63) >> 6, nA = nb * n_slice, nB = nb * (nb + 1) / 2;
    u64* colW = CLUSTER ? colG + (s % 3) * (TRI * 64) : colL;
    for (int u = t; u < nA + nB; u += T) {
      if (u < nA) {
        const int J = u % nb, ks = (u / nb) << 6;
        const u64 bits = nms_pair_bits(kbox + ks, karea + ks, min(64, n_keep - ks), cb[J * 64 + lane], ca[J * 64 + lane],
                                       thr, thr_hi, thr_lo);
        const u64 hit = __ballot(bits != 0ull);
        if (lane == 0 && hit) atomicOr(&supL[J], hit);
      } else {
        const int v = u - nA;
        int J = 0;
        while ((J + 1) * (J + 2) / 2 <= v) ++J;
        const int I = v - J * (J + 1) / 2;
        u64 bits = nms_pair_bits(cb + I * 64, ca + I * 64, 64, cb[J * 64 + lane], ca[J * 64 + lane], thr, thr_hi, thr_lo);
        if (I == J) bits &= (1ull << lane) - 1ull;           // strict upper triangle: only earlier candidates suppress
        if (CLUSTER) __hip_atomic_store(&colW[v * 64 + lane], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else colW[v * 64 + lane] = bits;
      }
    }
    __syncthreads();
    if (CLUSTER) {
      if (tid < nb && supL[tid]) __hip_atomic_fetch_or(&supG[s * 8 + tid], supL[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!nms_cluster_barrier(ctl, (unsigned)G * (unsigned)(s + 1), &s_fail)) {
        if (tid == 0) bad[n] = 1;                            // bboxes_eval turns it into a NaN the host raises on
        return;
      }
    }
    // ---- C: resolve the panel (threads 0 .. S-1 own one candidate each; every wave takes part in the barriers)
    const int Jm = tid >> 6;
    u64 col[NBLK];
    bool ok = false;
    if (tid < S) {
      const u64 sw = CLUSTER ? __hip_atomic_load(&supG[s * 8 + Jm], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : supL[Jm];
      ok = tid < rows && !((sw >> lane) & 1ull);
#pragma unroll
      for (int I = 0; I < NBLK; ++I) {
        u64* cw = &colW[(Jm * (Jm + 1) / 2 + I) * 64 + lane];
        col[I] = I <= Jm && Jm < nb ? (CLUSTER ? __hip_atomic_load(cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *cw) : 0ull;
      }
    }
    {
      const u64 k0 = __ballot(ok);
      if (lane == 0) Kb[wave] = k0;
    }
    __syncthreads();
    int cur = 0;
    for (int round = 0; round < S + 1; ++round) {
      bool nk = false;
      u64 old = 0ull;
      if (tid < S) {
        u64 acc = 0ull;
#pragma unroll
        for (int I = 0; I < NBLK; ++I) acc |= col[I] & Kb[cur * NMS_WAVES + I];
        nk = ok && acc == 0ull;
        old = Kb[cur * NMS_WAVES + Jm];
      }
      const u64 kw = __ballot(nk);
      if (lane == 0) Kb[(cur ^ 1) * NMS_WAVES + wave] = kw;
      cur ^= 1;
      if (!__syncthreads_or(tid < S && kw != old)) break;
    }
    // ---- append the survivors (in candidate order) to the kept list, up to post_n
    int before = 0, total = 0;
#pragma unroll
    for (int I = 0; I < NBLK; ++I) {
      const int c = __popcll(Kb[cur * NMS_WAVES + I]);
      before += I < Jm ? c : 0;
      total += c;
    }
    if (tid < S && ((Kb[cur * NMS_WAVES + Jm] >> lane) & 1ull)) {
      const int slot = n_keep + before + __popcll(Kb[cur * NMS_WAVES + Jm] & ((1ull << lane) - 1ull));
      if (slot < post_n) {
        kbox[slot] = cb[tid];
        karea[slot] = ca[tid];
        if (g == 0) K[slot] = base + tid;
      }
    }
    n_keep = __builtin_amdgcn_readfirstlane(min(post_n, n_keep + total));
    __syncthreads();
  }
  if (g == 0 && tid == 0) counts[n * 4 + 2] = n_keep;
}

// workgroups per image: a cluster only where the images alone cannot fill the chip; its members must all be resident
// (they wait for each other), so images x G stays within half of the 256 CUs
int nms_cluster_size(int N) {
  if (N > NMS_CLUSTER_IMAGES) return 1;
  int G = NMS_MAX_CLUSTER;
  while (G > 1 && N * G > 128) G >>= 1;
  return G;
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
  XDET_REQUIRE(nms_thr >= 0.f, "get_proposals: the NMS threshold must be >= 0");
  {
    // (the workspace may be carved for more images than this call runs: every per-image array is cleared for ITS first N images)
    const int n_sup = (int)cdiv(pre_n, NMS_SUP_PANEL);
    ZeroRanges z;
    z.p[0] = reinterpret_cast<uint4*>(ws.hist);    z.n[0] = (int64_t)N * HIST_BINS / 4;
    z.p[1] = reinterpret_cast<uint4*>(ws.bad);     z.n[1] = round_up(N, 4) / 4;
    z.p[2] = reinterpret_cast<uint4*>(ws.nms_ctl); z.n[2] = N;
    z.p[3] = reinterpret_cast<uint4*>(ws.nms_sup); z.n[3] = (int64_t)N * n_sup * 4;
    z.p[4] = reinterpret_cast<uint4*>(ws.sboxes);  z.n[4] = (int64_t)N * pre_n;
    z.p[5] = reinterpret_cast<uint4*>(ws.sscores); z.n[5] = cdiv((int64_t)N * pre_n, 4);   // (allocations are 256-byte multiples)
    hipLaunchKernelGGL(prop_zero_kernel, dim3(512), dim3(256), 0, s, z);
  }
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
  // The order of the (distinct) keys: a bitonic sort in LDS by ONE workgroup per image (the cheapest in CU time; the images of
  // a batch sort side by side).  Lists beyond the sort's LDS capacity (many keys sharing the threshold bin) take the
  // rank-counting pair below, which exits at once otherwise.  (Rank counting for every list of a single image, 160 workgroups
  // instead of one, was measured three times -- 1.302 against 1.296 ms, 1.158 against 1.152, 1.030 against 1.016 ms per image:
  // no gain.)
  const int sort_max = SORT_MAX;
  hipLaunchKernelGGL(prop_sort_kernel, dim3(N), dim3(1024), 0, s, ws.cand, ws.cboxes, n_anchor, pre_n, ws.counts, ws.sboxes,
                     ws.sscores, sort_max);
  XDET_LAUNCH_CHECK();
  hipLaunchKernelGGL(prop_rank_kernel, dim3((unsigned)cdiv(n_anchor, 256 * RANK_IPT), RANK_SPLITS, N), dim3(256), 0, s,
                     ws.cand, n_anchor, ws.counts, ws.ranks, sort_max);
  XDET_LAUNCH_CHECK();
  hipLaunchKernelGGL(prop_scatter_kernel, dim3(gb, N), dim3(256), 0, s, ws.cand, ws.ranks, ws.cboxes, n_anchor, pre_n,
                     ws.counts, ws.sboxes, ws.sscores, sort_max);
  XDET_LAUNCH_CHECK();
  {
    const int G = nms_cluster_size(N);
    u64* sup = ws.nms_sup;
    const int n_sup = (int)cdiv(pre_n, NMS_SUP_PANEL);
    if (G > 1) {
      constexpr int NB = NMS_CLUSTER_NBLK;
      const size_t lds = nms_lds_bytes(post_n, NB);
      XDET_REQUIRE(lds <= NMS_LDS_MAX, "get_proposals: rpn_post_nms_top_n too large (max 6144)");
      static DeviceOnce once;
      XDET_TRY(ensure_dynamic_lds(once, reinterpret_cast<const void*>(nms_panel_kernel<true, NB>), NMS_LDS_MAX));
      hipLaunchKernelGGL((nms_panel_kernel<true, NB>), dim3(N * G), dim3(NMS_THREADS), lds, s, ws.sboxes, ws.counts, pre_n,
                         post_n, nms_thr, ws.kept, G, sup, n_sup, ws.nms_col, ws.nms_ctl, ws.bad);
    } else {
      constexpr int NB = 4;
      const size_t lds = nms_lds_bytes(post_n, NB);
      XDET_REQUIRE(lds <= NMS_LDS_MAX, "get_proposals: rpn_post_nms_top_n too large (max 6144)");
      static DeviceOnce once;
      XDET_TRY(ensure_dynamic_lds(once, reinterpret_cast<const void*>(nms_panel_kernel<false, NB>), NMS_LDS_MAX));
      hipLaunchKernelGGL((nms_panel_kernel<false, NB>), dim3(N), dim3(NMS_THREADS), lds, s, ws.sboxes, ws.counts, pre_n,
                         post_n, nms_thr, ws.kept, 1, sup, n_sup, ws.nms_col, ws.nms_ctl, ws.bad);
    }
    XDET_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(prop_gather_kernel, dim3((unsigned)cdiv(post_n, 256), N), dim3(256), 0, s, ws.sboxes, ws.kept,
                     ws.counts, pre_n, post_n, rois);
  XDET_LAUNCH_CHECK();
  return XDET_OK;
}

}  // namespace xdet
