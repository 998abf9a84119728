// TEST INFRASTRUCTURE / CPU BASELINE -- not part of the product (only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this).  A C++17 + OpenMP restatement of the reference's eval forward
// (light_head_rfcn_eval.py:364-433 with net/xception_body.py:236-560, anchor_manipulator.py:641-757,
// eval_helper.py:249-625) in fp32 on the host cores -- the "C++ -O3 + OpenMP over all host cores" CPU baseline
// SURVEY.md 8d(1) specifies, since the reference's TF1 CPU runtime cannot run here.  It mirrors
// oracle/lighthead_oracle.py function by function (that NumPy file is the parity oracle; this one is validated
// against it in tests/test_cpu_baseline.py) and calls oracle/psroialign_ref.c for PsRoiAlign.
//
// Layout NHWC f32.  Every dense layer is an im2col-free GEMM: for each filter tap the micro-kernel takes eight
// row pointers (input pixel vectors, or a zero vector outside the image) and accumulates an 8 x 32 output block
// in registers against weights packed [Cout/32][tap][Cin][32]; bias / folded inference BN / residual / ReLU in
// the epilogue.  AVX-512 or AVX2 chosen at load time (target_clones); OpenMP over (row block, column block).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <new>
#include <cstdlib>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include <omp.h>

extern "C" int oracle_psroialign_fwd(const float* inputs, const float* rois, float* pooled, int32_t* index, int N, int C,
                                     int H, int W, int R, int grid_w, int grid_h, int use_max, int layout, int ldc);

namespace {

// Tensor storage comes from a per-THREAD cache of blocks: a forward allocates the same sequence of sizes for every image, so
// after a thread's first image every request is served from what the image before it released -- no mmap / munmap / first-
// touch page faults per tensor.  (With the C library's allocator each of the ~100 tensors of an image was a fresh mapping;
// 128 image-parallel threads of ONE process then queue on the kernel's address-space lock: 18 images/s on a 2 x 64-core host
// where 8 threads of the same code reach 0.74 images/s each.)  Blocks are never returned to the system; a thread keeps the
// ~0.35 GB one image needs.
struct BlockCache {
  std::multimap<size_t, void*> free_blocks;
  ~BlockCache() { for (auto& kv : free_blocks) ::free(kv.second); }
};
static thread_local BlockCache tl_blocks;
template <class T>
struct PoolAlloc {
  using value_type = T;
  PoolAlloc() = default;
  template <class U> PoolAlloc(const PoolAlloc<U>&) {}
  static size_t rounded(size_t n) { return (n * sizeof(T) + 4095) & ~(size_t)4095; }
  T* allocate(size_t n) {
    const size_t bytes = rounded(n);
    auto it = tl_blocks.free_blocks.lower_bound(bytes);
    if (it != tl_blocks.free_blocks.end() && it->first <= bytes + bytes / 4) {
      void* p = it->second;
      tl_blocks.free_blocks.erase(it);
      return static_cast<T*>(p);
    }
    void* p = aligned_alloc(4096, bytes);
    if (!p) throw std::bad_alloc();
    return static_cast<T*>(p);
  }
  void deallocate(T* p, size_t n) { tl_blocks.free_blocks.insert({rounded(n), p}); }   // (a block handed out for a smaller
                                                                                     //  request comes back under that size)
  template <class U> bool operator==(const PoolAlloc<U>&) const { return true; }
  template <class U> bool operator!=(const PoolAlloc<U>&) const { return false; }
};
typedef std::vector<float, PoolAlloc<float>> fvec;

struct T4 {           // NHWC tensor
  int n = 0, h = 0, w = 0, c = 0;
  fvec v;
  void alloc(int n_, int h_, int w_, int c_) { n = n_; h = h_; w = w_; c = c_; v.assign((size_t)n * h * w * c, 0.f); }
  float* px(int in, int y, int x) { return v.data() + (((size_t)in * h + y) * w + x) * c; }
  const float* px(int in, int y, int x) const { return v.data() + (((size_t)in * h + y) * w + x) * c; }
};

struct HostW { std::vector<float> v; std::vector<int64_t> dims; };

void same_pad(int n, int k, int s, int d, int* before, int* out) {
  const int ke = (k - 1) * d + 1;
  *out = (n + s - 1) / s;
  const int total = std::max((*out - 1) * s + ke - n, 0);
  *before = total / 2;
}

// ---- the GEMM micro-kernel: C[8][32] += sum_k a[r][k] * b[k][0..31] ------------------------------------------
constexpr int MR = 8, NR = 32;
__attribute__((target_clones("avx512f", "avx2", "default")))
void micro(const float* const* a, const float* b, int K, float* acc /*[MR][NR]*/) {
  typedef float v16 __attribute__((vector_size(64)));     // one zmm under AVX-512 (16 accumulators + 2 + 1 of 32 registers)
  v16 c[MR][2];
  for (int r = 0; r < MR; ++r)
    for (int j = 0; j < 2; ++j) memcpy(&c[r][j], acc + r * NR + j * 16, 64);
  for (int k = 0; k < K; ++k) {
    const float* bk = b + (size_t)k * NR;
    v16 b0, b1;
    memcpy(&b0, bk, 64); memcpy(&b1, bk + 16, 64);
    for (int r = 0; r < MR; ++r) {
      const float av = a[r][k];
      const v16 ab = {av, av, av, av, av, av, av, av, av, av, av, av, av, av, av, av};
      c[r][0] += ab * b0; c[r][1] += ab * b1;
    }
  }
  for (int r = 0; r < MR; ++r)
    for (int j = 0; j < 2; ++j) memcpy(acc + r * NR + j * 16, &c[r][j], 64);
}

struct Conv {
  int kh = 1, kw = 1, cin = 0, cout = 0, stride = 1, dil = 1, pad_mode = 1;   // 0 VALID, 1 SAME
  int relu = 0;
  std::vector<float> wp;            // packed [cout/32 (padded)][tap][cin][32]
  std::vector<float> scale, shift;  // folded BN / bias
  void pack(const float* hwio, const float* sc, const float* sh) {
    const int nb = (cout + NR - 1) / NR, taps = kh * kw;
    wp.assign((size_t)nb * taps * cin * NR, 0.f);
    for (int t = 0; t < taps; ++t)
      for (int ci = 0; ci < cin; ++ci)
        for (int co = 0; co < cout; ++co)
          wp[(((size_t)(co / NR) * taps + t) * cin + ci) * NR + (co % NR)] = hwio[((size_t)t * cin + ci) * cout + co];
    scale.assign(cout, 1.f);
    shift.assign(cout, 0.f);
    if (sc) std::copy(sc, sc + cout, scale.begin());
    if (sh) std::copy(sh, sh + cout, shift.begin());
  }
  // y = conv(relu_in ? relu(x) : x) * scale + shift (+ res) (relu)
  void run(const T4& x, T4* y, const T4* res, bool relu_in) const {
    int ho, wo, pt = 0, pl = 0;
    if (pad_mode == 1) { same_pad(x.h, kh, stride, dil, &pt, &ho); same_pad(x.w, kw, stride, dil, &pl, &wo); }
    else { ho = (x.h - ((kh - 1) * dil + 1)) / stride + 1; wo = (x.w - ((kw - 1) * dil + 1)) / stride + 1; }
    y->alloc(x.n, ho, wo, cout);
    const T4* xin = &x;
    T4 xr;
    if (relu_in) {
      xr = x;
#pragma omp parallel for
      for (int64_t i = 0; i < (int64_t)xr.v.size(); ++i) xr.v[i] = std::max(xr.v[i], 0.f);
      xin = &xr;
    }
    const int64_t M = (int64_t)x.n * ho * wo;
    const int nb = (cout + NR - 1) / NR, taps = kh * kw;
    const int64_t mb = (M + MR - 1) / MR;
    static const std::vector<float> zeros(8192, 0.f);
    // column block outermost: consecutive work items of a thread share one packed weight panel (K x 32 floats, L2
    // resident) and stream 8-row slabs of the activations past it
#pragma omp parallel for collapse(2) schedule(static)
    for (int bn = 0; bn < nb; ++bn)
      for (int64_t bm = 0; bm < mb; ++bm) {
        alignas(64) float acc[MR * NR];
        memset(acc, 0, sizeof(acc));
        int in_[MR], oy[MR], ox[MR];
        for (int r = 0; r < MR; ++r) {
          const int64_t m = std::min(bm * MR + r, M - 1);
          in_[r] = (int)(m / ((int64_t)ho * wo));
          const int rem = (int)(m - (int64_t)in_[r] * ho * wo);
          oy[r] = rem / wo; ox[r] = rem - oy[r] * wo;
        }
        for (int t = 0; t < taps; ++t) {
          const int ky = t / kw, kx = t - ky * kw;
          const float* a[MR];
          for (int r = 0; r < MR; ++r) {
            const int iy = oy[r] * stride - pt + ky * dil, ix = ox[r] * stride - pl + kx * dil;
            a[r] = ((unsigned)iy < (unsigned)x.h && (unsigned)ix < (unsigned)x.w) ? xin->px(in_[r], iy, ix) : zeros.data();
          }
          micro(a, wp.data() + ((size_t)bn * taps + t) * cin * NR, cin, acc);
        }
        for (int r = 0; r < MR; ++r) {
          const int64_t m = bm * MR + r;
          if (m >= M) break;
          float* o = y->v.data() + m * cout;
          const float* rr = res ? res->v.data() + m * cout : nullptr;
          for (int j = 0; j < NR; ++j) {
            const int co = bn * NR + j;
            if (co >= cout) break;
            float v = acc[r * NR + j] * scale[co] + shift[co];
            if (rr) v += rr[co];
            if (relu) v = std::max(v, 0.f);
            o[co] = v;
          }
        }
      }
  }
};

void depthwise(const T4& x, const float* k33c /*[3][3][C]*/, int dil, bool relu_in, T4* y) {
  y->alloc(x.n, x.h, x.w, x.c);
  const int C = x.c;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < x.n; ++n)
    for (int yy = 0; yy < x.h; ++yy)
      for (int xx = 0; xx < x.w; ++xx) {
        float* o = y->px(n, yy, xx);
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = yy + (ky - 1) * dil;
          if ((unsigned)iy >= (unsigned)x.h) continue;
          for (int kx = 0; kx < 3; ++kx) {
            const int ix = xx + (kx - 1) * dil;
            if ((unsigned)ix >= (unsigned)x.w) continue;
            const float* in = x.px(n, iy, ix);
            const float* w = k33c + (size_t)(ky * 3 + kx) * C;
            if (relu_in) for (int c = 0; c < C; ++c) o[c] += std::max(in[c], 0.f) * w[c];
            else for (int c = 0; c < C; ++c) o[c] += in[c] * w[c];
          }
        }
      }
}

void maxpool_add(const T4& x, const T4& res, T4* y) {
  int ho, wo, pt, pl;
  same_pad(x.h, 3, 2, 1, &pt, &ho);
  same_pad(x.w, 3, 2, 1, &pl, &wo);
  y->alloc(x.n, ho, wo, x.c);
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < x.n; ++n)
    for (int oy = 0; oy < ho; ++oy)
      for (int ox = 0; ox < wo; ++ox) {
        float* o = y->px(n, oy, ox);
        for (int c = 0; c < x.c; ++c) o[c] = -FLT_MAX;
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy * 2 - pt + ky, ix = ox * 2 - pl + kx;
            if ((unsigned)iy >= (unsigned)x.h || (unsigned)ix >= (unsigned)x.w) continue;
            const float* in = x.px(n, iy, ix);
            for (int c = 0; c < x.c; ++c) o[c] = std::max(o[c], in[c]);
          }
        const float* r = res.px(n, oy, ox);
        for (int c = 0; c < x.c; ++c) o[c] += r[c];
      }
}

// ---- boxes ------------------------------------------------------------------------------------------------------
inline float iou_tf(const float* a, const float* b) {          // tf.image.non_max_suppression's IoU
  const float ya0 = std::min(a[0], a[2]), xa0 = std::min(a[1], a[3]), ya1 = std::max(a[0], a[2]), xa1 = std::max(a[1], a[3]);
  const float yb0 = std::min(b[0], b[2]), xb0 = std::min(b[1], b[3]), yb1 = std::max(b[0], b[2]), xb1 = std::max(b[1], b[3]);
  const float aa = (ya1 - ya0) * (xa1 - xa0), ab = (yb1 - yb0) * (xb1 - xb0);
  if (aa <= 0.f || ab <= 0.f) return 0.f;
  const float ih = std::max(std::min(ya1, yb1) - std::max(ya0, yb0), 0.f), iw = std::max(std::min(xa1, xb1) - std::max(xa0, xb0), 0.f);
  const float inter = ih * iw;
  return inter / (aa + ab - inter);
}

// greedy NMS over boxes already sorted by descending score; returns kept indices (at most max_out)
std::vector<int> nms_sorted(const std::vector<float>& boxes, int n, int max_out, float thr) {
  std::vector<int> keep;
  for (int i = 0; i < n && (int)keep.size() < max_out; ++i) {
    bool ok = true;
    for (int j : keep)
      if (iou_tf(&boxes[(size_t)i * 4], &boxes[(size_t)j * 4]) > thr) { ok = false; break; }
    if (ok) keep.push_back(i);
  }
  return keep;
}

// descending score, ties -> lower index first (tf.nn.top_k)
std::vector<int> topk_desc(const std::vector<float>& s, int k) {
  std::vector<int> idx(s.size());
  std::iota(idx.begin(), idx.end(), 0);
  k = std::min<int>(k, (int)idx.size());
  std::partial_sort(idx.begin(), idx.begin() + k, idx.end(), [&](int a, int b) { return s[a] > s[b] || (s[a] == s[b] && a < b); });
  idx.resize(k);
  return idx;
}

struct Net {
  int S = 480, R = 300, pre_n = 5000, num_classes = 21, A = 22, grid = 7, bank = 10, nms_topk = 200;
  float rpn_nms = 0.7f, rpn_min = 16.f / 480.f, select_thr = 0.01f, nms_thr = 0.3f;
  std::map<std::string, HostW> w;
  bool built = false;
  std::map<std::string, Conv> convs;
  std::map<std::string, std::vector<float>> dwk;
  std::string err;

  const HostW& get(const std::string& n) {
    auto it = w.find(n);
    if (it == w.end()) throw std::runtime_error("missing weight " + n);
    return it->second;
  }
  void fold_bn(const std::string& bn, int C, float eps, const float* bias, std::vector<float>* sc, std::vector<float>* sh) {
    const auto &g = get(bn + "/gamma").v, &b = get(bn + "/beta").v, &m = get(bn + "/moving_mean").v, &v = get(bn + "/moving_variance").v;
    sc->resize(C); sh->resize(C);
    for (int c = 0; c < C; ++c) {
      const float s = g[c] / std::sqrt(v[c] + eps);
      (*sc)[c] = s;
      (*sh)[c] = b[c] - m[c] * s + (bias ? bias[c] * s : 0.f);
    }
  }
  Conv& mk(const std::string& name, int kh, int kw, int cin, int cout, int stride, int dil, int pad, int relu,
           const float* hwio, const float* sc, const float* sh) {
    Conv& c = convs[name];
    c.kh = kh; c.kw = kw; c.cin = cin; c.cout = cout; c.stride = stride; c.dil = dil; c.pad_mode = pad; c.relu = relu;
    c.pack(hwio, sc, sh);
    return c;
  }
  void conv_bn(const std::string& name, const std::string& bn, int k, int cin, int cout, int stride, int pad, int relu) {
    std::vector<float> sc, sh;
    fold_bn(bn, cout, 1e-4f, nullptr, &sc, &sh);
    mk(name, k, k, cin, cout, stride, 1, pad, relu, get(name + "/kernel").v.data(), sc.data(), sh.data());
  }
  void sep(const std::string& name, int cin, int cout, int relu) {
    std::vector<float> sc, sh;
    fold_bn(name + "_bn", cout, 1e-4f, nullptr, &sc, &sh);
    mk(name, 1, 1, cin, cout, 1, 1, 1, relu, get(name + "/pointwise_kernel").v.data(), sc.data(), sh.data());
    dwk[name] = get(name + "/depthwise_kernel").v;       // [3][3][cin][1]
  }
  void build() {
    conv_bn("block1_conv1", "block1_conv1_bn", 3, 3, 32, 2, 0, 1);
    conv_bn("block1_conv2", "block1_conv2_bn", 3, 32, 64, 1, 0, 1);
    const char* resn[3] = {"conv2d_1", "conv2d_2", "conv2d_3"};
    const char* bnn[3] = {"batch_normalization_1", "batch_normalization_2", "batch_normalization_3"};
    const int ch[4] = {64, 128, 256, 728};
    for (int b = 0; b < 3; ++b) {
      conv_bn(resn[b], bnn[b], 1, ch[b], ch[b + 1], 2, 1, 0);
      sep("block" + std::to_string(b + 2) + "_sepconv1", ch[b], ch[b + 1], 0);
      sep("block" + std::to_string(b + 2) + "_sepconv2", ch[b + 1], ch[b + 1], 0);
    }
    for (int blk = 5; blk <= 12; ++blk)
      for (int i = 1; i <= 3; ++i) sep("block" + std::to_string(blk) + "_sepconv" + std::to_string(i), 728, 728, 0);
    conv_bn("conv2d_4", "batch_normalization_4", 1, 728, 1024, 1, 1, 0);
    sep("block13_sepconv1", 728, 728, 0);
    sep("block13_sepconv2", 728, 1024, 0);
    sep("block14_sepconv1", 1024, 1536, 1);
    sep("block14_sepconv2", 1536, 2048, 1);
    mk("rpn_head/conv2d", 3, 3, 728, 512, 1, 1, 1, 1, get("rpn_head/conv2d/kernel").v.data(), nullptr, get("rpn_head/conv2d/bias").v.data());
    mk("rpn_head/conv2d_1", 1, 1, 512, 2 * A, 1, 1, 1, 0, get("rpn_head/conv2d_1/kernel").v.data(), nullptr, get("rpn_head/conv2d_1/bias").v.data());
    mk("rpn_head/conv2d_2", 1, 1, 512, 4 * A, 1, 1, 1, 0, get("rpn_head/conv2d_2/kernel").v.data(), nullptr, get("rpn_head/conv2d_2/bias").v.data());
    const int C = bank * grid * grid;
    for (int br = 0; br < 2; ++br) {
      const std::string p = "large_sep_feature/Branch_" + std::to_string(br);
      mk(p + "/conv2d", 15, 1, 2048, 256, 1, 1, 1, 0, get(p + "/conv2d/kernel").v.data(), nullptr, get(p + "/conv2d/bias").v.data());
      mk(p + "/conv2d_1", 1, 15, 256, C, 1, 1, 1, 0, get(p + "/conv2d_1/kernel").v.data(), nullptr, get(p + "/conv2d_1/bias").v.data());
    }
    mk("final_head/subnet_fc", 1, 1, C, 2048, 1, 1, 0, 1, get("final_head/subnet_fc/kernel").v.data(), nullptr, get("final_head/subnet_fc/bias").v.data());
    mk("final_head/fc_cls", 1, 1, 2048, num_classes, 1, 1, 0, 0, get("final_head/fc_cls/kernel").v.data(), nullptr, get("final_head/fc_cls/bias").v.data());
    mk("final_head/fc_loc", 1, 1, 2048, 4, 1, 1, 0, 0, get("final_head/fc_loc/kernel").v.data(), nullptr, get("final_head/fc_loc/bias").v.data());
    built = true;
  }
  void sep_run(const std::string& name, const T4& x, bool pre_relu, int dil, const T4* res, T4* y) {
    T4 d;
    depthwise(x, dwk[name].data(), dil, pre_relu, &d);
    convs[name].run(d, y, res, false);
  }

  // Batches with at least one image per two threads run IMAGE-parallel: every thread takes whole images through the
  // whole graph (the `omp parallel for`s inside are then nested regions and run on the calling thread), so nothing is
  // synchronised per layer and nothing serial (tensor allocation, the ReLU copies) is left between parallel regions --
  // the layer-parallel form below was fastest at 16 threads and slower beyond on a 2 x 64-core host.
  // Smaller batches keep the layer-parallel form.
  void forward(const float* images_nchw, int N, float* det_scores, float* det_boxes) {
    const int nt = omp_get_max_threads();
    if (N >= 2 && 2 * N >= nt && !omp_in_parallel()) {
      std::string first_error;
#pragma omp parallel for schedule(dynamic, 1)
      for (int n = 0; n < N; ++n) {
        try {
          forward_batch(images_nchw + (size_t)n * 3 * S * S, 1, det_scores + (size_t)n * (num_classes - 1) * nms_topk,
                        det_boxes + (size_t)n * (num_classes - 1) * nms_topk * 4);
        } catch (const std::exception& e) {
#pragma omp critical
          if (first_error.empty()) first_error = e.what();
        }
      }
      if (!first_error.empty()) throw std::runtime_error(first_error);
      return;
    }
    forward_batch(images_nchw, N, det_scores, det_boxes);
  }

  void forward_batch(const float* images_nchw, int N, float* det_scores, float* det_boxes) {
    T4 x;
    x.alloc(N, S, S, 3);
    for (int n = 0; n < N; ++n)
      for (int c = 0; c < 3; ++c)
        for (int i = 0; i < S * S; ++i) x.v[((size_t)n * S * S + i) * 3 + c] = images_nchw[((size_t)n * 3 + c) * S * S + i];
    T4 a, b, r, t, p;
    convs["block1_conv1"].run(x, &a, nullptr, false);
    convs["block1_conv2"].run(a, &x, nullptr, false);
    const char* resn[3] = {"conv2d_1", "conv2d_2", "conv2d_3"};
    for (int blk = 0; blk < 3; ++blk) {
      convs[resn[blk]].run(x, &r, nullptr, false);
      const std::string nm = "block" + std::to_string(blk + 2);
      sep_run(nm + "_sepconv1", x, blk != 0, 1, nullptr, &a);
      sep_run(nm + "_sepconv2", a, true, 1, nullptr, &b);
      maxpool_add(b, r, &p);
      x = p;
    }
    for (int blk = 5; blk <= 12; ++blk) {
      const std::string nm = "block" + std::to_string(blk);
      sep_run(nm + "_sepconv1", x, true, 1, nullptr, &a);
      sep_run(nm + "_sepconv2", a, true, 1, nullptr, &b);
      sep_run(nm + "_sepconv3", b, true, 1, &x, &t);
      x = t;
    }
    T4 mid = x;                                            // mid_outputs = relu(x): applied on load below
    convs["conv2d_4"].run(x, &r, nullptr, false);
    sep_run("block13_sepconv1", x, true, 1, nullptr, &a);
    sep_run("block13_sepconv2", a, true, 1, &r, &b);
    sep_run("block14_sepconv1", b, false, 2, nullptr, &a);
    T4 out;
    sep_run("block14_sepconv2", a, false, 2, nullptr, &out);
    // RPN
    T4 hid, cls, box;
    convs["rpn_head/conv2d"].run(mid, &hid, nullptr, true);
    convs["rpn_head/conv2d_1"].run(hid, &cls, nullptr, false);
    convs["rpn_head/conv2d_2"].run(hid, &box, nullptr, false);
    // large separable
    T4 feat, f0, f1, u0, u1;
    convs["large_sep_feature/Branch_0/conv2d"].run(out, &f0, nullptr, false);
    convs["large_sep_feature/Branch_0/conv2d_1"].run(f0, &u0, nullptr, false);
    convs["large_sep_feature/Branch_1/conv2d"].run(out, &f1, nullptr, false);
    convs["large_sep_feature/Branch_1/conv2d_1"].run(f1, &u1, nullptr, false);
    {
      const int C = u0.c;
      std::vector<float> sc, sh;
      fold_bn("large_sep_feature/batch_normalization", C, 1e-5f, nullptr, &sc, &sh);
      feat.alloc(u0.n, u0.h, u0.w, C);
#pragma omp parallel for
      for (int64_t i = 0; i < (int64_t)feat.v.size(); ++i) {
        const int c = (int)(i % C);
        feat.v[i] = std::max((u0.v[i] + u1.v[i]) * sc[c] + sh[c], 0.f);
      }
    }
    const int F = cls.h, na = F * F * A, C = bank * grid * grid;
    // anchors (anchor_manipulator.py:698-757)
    std::vector<float> ah(A), aw(A);
    {
      int k = 0;
      ah[0] = aw[0] = 0.1f; ++k;
      const double scales[7] = {0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8}, ratios[3] = {1., 2., .5};
      for (double s : scales) for (double ra : ratios) { ah[k] = (float)(s / std::sqrt(ra)); aw[k] = (float)(s * std::sqrt(ra)); ++k; }
    }
    std::vector<float> proposals((size_t)N * R * 4);
#pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < N; ++n) {
      std::vector<float> obj(na), bx((size_t)na * 4);
      for (int y = 0; y < F; ++y)
        for (int xx = 0; xx < F; ++xx) {
          const float cy0 = ((float)y + 0.5f) * 16.f / (float)S, cx0 = ((float)xx + 0.5f) * 16.f / (float)S;
          const float* c2 = cls.px(n, y, xx);
          const float* b4 = box.px(n, y, xx);
          for (int k = 0; k < A; ++k) {
            const int i = (y * F + xx) * A + k;
            const float l0 = c2[2 * k], l1 = c2[2 * k + 1], mx = std::max(l0, l1);
            const float e0 = std::exp(l0 - mx), e1 = std::exp(l1 - mx);
            obj[i] = e1 / (e0 + e1);
            const float hh = std::exp(b4[4 * k + 2]) * ah[k], ww = std::exp(b4[4 * k + 3]) * aw[k];
            const float cy = b4[4 * k] * ah[k] + cy0, cx = b4[4 * k + 1] * aw[k] + cx0;
            float* o = &bx[(size_t)i * 4];
            o[0] = cy - hh / 2.f; o[1] = cx - ww / 2.f; o[2] = cy + hh / 2.f; o[3] = cx + ww / 2.f;
          }
        }
      // get_proposals (net/xception_body.py:402-448)
      std::vector<float> vs; std::vector<float> vb;
      for (int i = 0; i < na; ++i) {
        float* o = &bx[(size_t)i * 4];
        float y0 = std::min(std::max(o[0], 0.f), 1.f), x0 = std::min(std::max(o[1], 0.f), 1.f);
        float y1 = std::min(std::max(o[2], 0.f), 1.f), x1 = std::min(std::max(o[3], 0.f), 1.f);
        y0 = std::min(y0, y1); x0 = std::min(x0, x1);
        const float hh = y1 - y0, ww = x1 - x0, cy = y0 + hh / 2.f, cx = x0 + ww / 2.f;
        if (ww > rpn_min && hh > rpn_min && cx > 0.f && cx < 1.f && cy > 0.f && cy < 1.f) {
          vs.push_back(obj[i]);
          vb.insert(vb.end(), {y0, x0, y1, x1});
        }
      }
      const std::vector<int> order = topk_desc(vs, pre_n);
      std::vector<float> sb((size_t)pre_n * 4, 0.f), ss(pre_n, 0.f);
      for (size_t i = 0; i < order.size(); ++i) { ss[i] = vs[order[i]]; memcpy(&sb[i * 4], &vb[(size_t)order[i] * 4], 16); }
      const std::vector<int> keep = nms_sorted(sb, pre_n, R, rpn_nms);
      std::vector<float> kb;
      for (int k : keep) if (ss[k] > 0.f) kb.insert(kb.end(), &sb[(size_t)k * 4], &sb[(size_t)k * 4] + 4);
      int nk = (int)kb.size() / 4;
      if (nk == 0) { kb = {0.2f, 0.2f, 0.8f, 0.8f}; nk = 1; }
      float* pr = &proposals[(size_t)n * R * 4];
      for (int i = 0; i < R; ++i) memcpy(pr + (size_t)i * 4, &kb[(size_t)(i % nk) * 4], 16);   // tile; shuffle tail = identity
    }
    // head: PsRoiAlign (oracle/psroialign_ref.c, NHWC map, (cy,cx,h,w) ROIs), ROI chunks across the cores
    std::vector<float> rois((size_t)N * R * 4), pooled((size_t)N * R * C);
    std::vector<int32_t> pidx((size_t)N * R * C);
    for (size_t i = 0; i < (size_t)N * R; ++i) {
      const float* b4 = &proposals[i * 4];
      const float hh = b4[2] - b4[0], ww = b4[3] - b4[1];
      rois[i * 4] = b4[0] + hh / 2.f; rois[i * 4 + 1] = b4[1] + ww / 2.f; rois[i * 4 + 2] = hh; rois[i * 4 + 3] = ww;
    }
    const int RC = 8, nchunk = (R + RC - 1) / RC;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int n = 0; n < N; ++n)
      for (int ch = 0; ch < nchunk; ++ch) {
        const int r0 = ch * RC, rn = std::min(RC, R - r0);
        const size_t o = (size_t)n * R + r0;
        oracle_psroialign_fwd(feat.v.data() + (size_t)n * F * F * C, &rois[o * 4], &pooled[o * C], &pidx[o * C], 1, C, F, F,
                              rn, grid, grid, 1, /*layout NHWC*/ 1, C);
      }
    T4 pl, fc, lc, lr;
    pl.alloc(N, R, 1, C);
    pl.v.assign(pooled.begin(), pooled.end());
    convs["final_head/subnet_fc"].run(pl, &fc, nullptr, false);
    convs["final_head/fc_cls"].run(fc, &lc, nullptr, false);
    convs["final_head/fc_loc"].run(fc, &lr, nullptr, false);
    // ext_decode_rois + bboxes_eval
    const int nc = num_classes;
    const float min_size = std::max(0.0001f, 0.03f * std::sqrt((float)(S * S) / (float)(S * S)));
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int n = 0; n < N; ++n)
      for (int c = 1; c < nc; ++c) {
        std::vector<float> s, b;
        for (int r = 0; r < R; ++r) {
          const float* lg = &lc.v[((size_t)n * R + r) * nc];
          float mx = lg[0];
          for (int k = 1; k < nc; ++k) mx = std::max(mx, lg[k]);
          float den = 0.f;
          for (int k = 0; k < nc; ++k) den += std::exp(lg[k] - mx);
          const float pcls = std::exp(lg[c] - mx) / den;
          if (!(pcls > select_thr)) continue;                   // masked-to-zero rows never survive the size filter
          const float* pb = &proposals[((size_t)n * R + r) * 4];
          const float* d = &lr.v[((size_t)n * R + r) * 4];
          const float hr = pb[2] - pb[0], wr = pb[3] - pb[1], yr = pb[0] + hr / 2.f, xr = pb[1] + wr / 2.f;
          const float hh = std::exp(d[2]) * hr, ww = std::exp(d[3]) * wr, cy = d[0] * hr + yr, cx = d[1] * wr + xr;
          float y0 = cy - hh / 2.f, x0 = cx - ww / 2.f, y1 = cy + hh / 2.f, x1 = cx + ww / 2.f;
          y0 = std::min(std::max(y0, 0.f), 1.f); x0 = std::min(std::max(x0, 0.f), 1.f);
          y1 = std::min(std::max(y1, 0.f), 1.f); x1 = std::min(std::max(x1, 0.f), 1.f);
          y0 = std::min(y0, y1); x0 = std::min(x0, x1);
          const float bh = y1 - y0, bw = x1 - x0, ccy = y0 + bh / 2.f, ccx = x0 + bw / 2.f;
          if (!(bw > min_size && bh > min_size && ccx > 0.f && ccx < 1.f && ccy > 0.f && ccy < 1.f)) continue;
          s.push_back(pcls);
          b.insert(b.end(), {y0, x0, y1, x1});
        }
        const std::vector<int> order = topk_desc(s, 2 * nms_topk);
        std::vector<float> sb2(order.size() * 4), ss2(order.size());
        for (size_t i = 0; i < order.size(); ++i) { ss2[i] = s[order[i]]; memcpy(&sb2[i * 4], &b[(size_t)order[i] * 4], 16); }
        const std::vector<int> keep = nms_sorted(sb2, (int)order.size(), nms_topk, nms_thr);
        float* os = det_scores + ((size_t)n * (nc - 1) + (c - 1)) * nms_topk;
        float* ob = det_boxes + ((size_t)n * (nc - 1) + (c - 1)) * nms_topk * 4;
        std::fill(os, os + nms_topk, 0.f);
        std::fill(ob, ob + (size_t)nms_topk * 4, 0.f);
        for (size_t i = 0; i < keep.size(); ++i) { os[i] = ss2[keep[i]]; memcpy(ob + i * 4, &sb2[(size_t)keep[i] * 4], 16); }
      }
  }
};

}  // namespace

extern "C" {

void* lhcpu_create(int image_size, int rpn_post_nms_top_n) {
  Net* n = new Net();
  n->S = image_size;
  n->R = rpn_post_nms_top_n;
  return n;
}
int lhcpu_set_weight(void* net, const char* name, const float* data, int ndim, const int64_t* dims) {
  Net* n = static_cast<Net*>(net);
  HostW t;
  size_t cnt = 1;
  for (int i = 0; i < ndim; ++i) { t.dims.push_back(dims[i]); cnt *= (size_t)dims[i]; }
  t.v.assign(data, data + cnt);
  n->w[name] = std::move(t);
  return 0;
}
int lhcpu_build(void* net) {
  try { static_cast<Net*>(net)->build(); } catch (const std::exception& e) { static_cast<Net*>(net)->err = e.what(); return -1; }
  return 0;
}
const char* lhcpu_error(void* net) { return static_cast<Net*>(net)->err.c_str(); }
int lhcpu_forward(void* net, const float* images_nchw, int N, float* det_scores, float* det_boxes) {
  Net* n = static_cast<Net*>(net);
  if (!n->built) return -1;
  try { n->forward(images_nchw, N, det_scores, det_boxes); } catch (const std::exception& e) { n->err = e.what(); return -2; }
  return 0;
}
int lhcpu_threads(void) { return omp_get_max_threads(); }
void lhcpu_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
void lhcpu_destroy(void* net) { delete static_cast<Net*>(net); }

}  // extern "C"
