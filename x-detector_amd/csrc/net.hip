// Host side of libxdet_hip.so: layer objects, the Light-Head R-CNN eval graph
// (lighr_head_model_fn, light_head_rfcn_eval.py:364-433) and the ResNet-50 v2 trunk
// (net/resnet_v2.py:311-345) as static launch plans over the kernels in this directory,
// plus the C-ABI of include/xdet.h.
#include "common.h"
#include "../../include/xdet.h"

#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace xdet {

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  char buf[512];
  snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
  g_last_error = buf;
  return XDET_ERR_HIP;
}

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Layers and nets live on the device that was current when they were created; every entry point that
// launches on their behalf makes that device current for the call (and restores the caller's), so two
// detectors on two GPUs can share one process / one host thread per device.
struct DeviceGuard {
  int prev = -1, want = -1;
  explicit DeviceGuard(int dev) : want(dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != want) (void)hipSetDevice(want);
  }
  ~DeviceGuard() {
    if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
  }
};

struct HostTensor {
  std::vector<float> v;
  std::vector<int64_t> dims;
};
typedef std::map<std::string, HostTensor> WeightMap;

static int upload(const std::vector<float>& h, float** d) {
  XDET_HIP(hipMalloc(reinterpret_cast<void**>(d), std::max<size_t>(h.size(), 1) * sizeof(float)));
  XDET_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return XDET_OK;
}

static int g_default_precision = PREC_F32;

static inline unsigned short f32_to_f16_rne(float f) {
  unsigned x;
  memcpy(&x, &f, 4);
  const unsigned sign = (x >> 16) & 0x8000u;
  x &= 0x7FFFFFFFu;
  if (x >= 0x47800000u) return (unsigned short)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));   // inf / nan
  if (x < 0x38800000u) {                       // below 2^-14: subnormal half = round(|f| * 2^24)
    float af;
    memcpy(&af, &x, 4);
    return (unsigned short)(sign | (unsigned)lrintf(af * 16777216.0f));
  }
  const unsigned mant = x & 0x7FFFFFu, exp = (x >> 23) - 127 + 15;
  unsigned h = (exp << 10) | (mant >> 13);
  const unsigned rem = mant & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;      // round to nearest even (may carry to inf)
  return (unsigned short)(sign | h);
}
// OCP fp8 e4m3 (bias 7, largest 448 = 0x7e, no infinities), round to nearest even, saturating: the format of the x8 cross-term
// operands (conv_params.h; the device side is v_cvt_pk_fp8_f32 in the producing kernels)
static inline unsigned char f32_to_e4m3(float f) {
  const unsigned char sgn = std::signbit(f) ? 0x80 : 0;
  const float a = std::fabs(f);
  if (!(a == a)) return 0x7f;
  if (a >= 448.f) return sgn | 0x7e;
  if (a < 0.015625f) {                                   // below the smallest normal 2^-6: subnormals are multiples of 2^-9
    const int q = (int)std::nearbyint(std::ldexp((double)a, 9));
    return sgn | (unsigned char)q;                       // q = 8 is the encoding of 2^-6 itself
  }
  int e;
  (void)std::frexp(a, &e);                               // a = m * 2^e, m in [0.5, 1)
  e -= 1;
  int q = (int)std::nearbyint(std::ldexp((double)a, 3 - e));   // 8 .. 16
  if (q == 16) { q = 8; ++e; }
  return sgn | (unsigned char)(((e + 7) << 3) | (q - 8));
}
static inline float f16_to_f32(unsigned short h) {
  const unsigned sign = (unsigned)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
  float v;
  if (e == 0) v = ldexpf((float)m, -24);
  else if (e == 31) v = m ? NAN : INFINITY;
  else v = ldexpf((float)(m | 0x400u), (int)e - 25);
  unsigned u;
  memcpy(&u, &v, 4);
  u |= sign;
  memcpy(&v, &u, 4);
  return v;
}

static void same_pad(int n, int k, int s, int d, int* before, int* out) {
  const int k_eff = (k - 1) * d + 1;
  *out = (n + s - 1) / s;
  const int total = std::max((*out - 1) * s + k_eff - n, 0);
  *before = total / 2;   // the extra pixel goes to the bottom / right (TF SAME rule)
}

// ---------------------------------------------------------------------------------------
// layer objects
// ---------------------------------------------------------------------------------------
struct LayerBase {
  virtual ~LayerBase() {}
  int kind = 0;   // 1 conv, 2 depthwise
  int device = 0; // the HIP device the weights live on (current device at creation)
};

struct ConvLayer : LayerBase {
  int kh, kw, cin, cout, stride, dil, pad_mode, pad_t, pad_l, relu_out;
  int groups = 1;     // > 1: grouped GEMM (1x1 only): weight matrix g serves rows [g*group_rows, (g+1)*group_rows)
  int cin_p, kp, cout_pad, n_tile;
  bool small_cin;
  int precision = PREC_F32;
  float *d_wt = nullptr, *d_scale = nullptr, *d_shift = nullptr;
  unsigned short *d_wt_hi = nullptr, *d_wt_lo = nullptr;
  // the same f16 weights K-blocked as [Kp/32][Cout_pad][32] for the LDS-DMA kernel: one K-step of a
  // tile's B operand is then one contiguous run
  unsigned short *d_wt_hi_b = nullptr, *d_wt_lo_b = nullptr;
  unsigned short* d_wt_x8_b = nullptr;   // x8 form of the K-blocked lo plane (conv_params.h): 32 B fp8(w_lo * 2^9) | 32 B fp8(w_hi * 2^-2) per (K block, row)
  unsigned short* d_zeros = nullptr;   // 256 B of zeros on the layer's device: the source of out-of-image taps
  // Activation pre-scale (split-precision range, Plan::calibrate): the A-operand planes hold x * 2^-in_exp and the
  // epilogue scale carries 2^in_exp (exact: powers of two); a planes copy of the output is written as out * 2^-out_exp
  // through the epilogue's pl_scale / pl_shift.  Both 0 unless a calibration found a tensor near the f16 range.
  std::vector<float> h_scale;          // host copy of d_scale at in_exp = 0
  int in_exp = 0, out_exp = 0;
  // Fixed split of the reduction (conv_mfma_ksplit.hip): a constant of the LAYER, set at plan time; once enabled the
  // layer runs on that kernel at every batch size (its two modes are bit-identical), so results never depend on the batch.
  int ksplit = 0;                      // 0 = the layer is not on the split-K kernel
  int ks_mode = 0;                     // 0 = by grid size, 1 = parallel ranges, 2 = one workgroup per tile (tests)
  int pl_c32 = 0;                      // planes destination wider than the output (conv_params.h pl_c32); 0: its own planes tensor
  bool ks_narrow = false;              // 128 x 64 tiles although the layer is a multiple of 128 wide (one range, twice the workgroups)
  int64_t ks_tiles = 0;                // tiles the scratch below was sized for
  float* d_ks_partial = nullptr;
  int* d_ks_ticket = nullptr;          // 4096 zeroed ints: arrival tickets of the in-kernel fold (conv_params.h ks_ticket)
  bool ks_owns_scratch = false;        // false: the slab belongs to the plan (one per stream, shared by its layers)
  size_t ks_scratch_bytes(int S, int64_t tiles) const {
    return S > 1 ? (size_t)std::max<int64_t>(tiles, 1) * S * 128 * (cout_pad % 128 == 0 ? 128 : 64) * sizeof(float) : 0;
  }
  // shared == nullptr: the layer allocates its own slab (stand-alone layers behind xdet_conv_set_ksplit); inside a plan the
  // ops of one stream run one after the other, so all its split-K layers borrow ONE slab sized for the largest of them
  int enable_ksplit(int S, int64_t max_parallel_tiles, float* shared = nullptr) {
    XDET_REQUIRE(S >= 1 && S <= 16 && dma_capable() && groups == 1, "ksplit: 1..16 ranges, a split-precision non-grouped layer");
    if (d_ks_partial && ks_owns_scratch) { (void)hipFree(d_ks_partial); (void)hipFree(d_ks_ticket); }
    d_ks_partial = nullptr;
    d_ks_ticket = nullptr;
    ks_owns_scratch = false;
    ks_tiles = std::max<int64_t>(max_parallel_tiles, 1);
    if (S > 1) {
      if (shared) d_ks_partial = shared;
      else {
        XDET_HIP(hipMalloc(reinterpret_cast<void**>(&d_ks_partial), ks_scratch_bytes(S, ks_tiles)));
        XDET_HIP(hipMalloc(reinterpret_cast<void**>(&d_ks_ticket), 4096 * sizeof(int)));
        XDET_HIP(hipMemset(d_ks_ticket, 0, 4096 * sizeof(int)));
        ks_owns_scratch = true;
      }
    }
    ksplit = S;
    return XDET_OK;
  }
  float *d_pl_scale = nullptr, *d_pl_shift = nullptr;
  int set_in_exp(int e) {
    std::vector<float> v(h_scale);
    for (float& x : v) x = ldexpf(x, e);
    XDET_HIP(hipMemcpy(d_scale, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    in_exp = e;
    return XDET_OK;
  }
  // planes copy of the output: relu?(out * pl_scale + pl_shift).  Without a folded BN pl = (2^-e, 0); with one (the
  // pre-activation of the next ResNet block, net/resnet_v2.py:142-156) pl = (bn_scale 2^-e, bn_shift 2^-e):
  // relu(x s + h) 2^-e = relu(x (s 2^-e) + h 2^-e), exactly.
  std::vector<float> h_pl_bn_scale, h_pl_bn_shift;   // the folded BN at e = 0 (empty: none)
  bool planes_bn() const { return !h_pl_bn_scale.empty(); }
  int set_planes_bn(const std::vector<float>& sc, const std::vector<float>& sh) {
    h_pl_bn_scale = sc;
    h_pl_bn_shift = sh;
    return set_out_exp(0);
  }
  int set_out_exp(int e) {
    const size_t n = (size_t)std::max(cout_pad, ld_out());
    if (!d_pl_scale) {
      XDET_HIP(hipMalloc(reinterpret_cast<void**>(&d_pl_scale), n * sizeof(float)));
      XDET_HIP(hipMalloc(reinterpret_cast<void**>(&d_pl_shift), n * sizeof(float)));
    }
    std::vector<float> a(n, ldexpf(1.f, -e)), b(n, 0.f);
    if (planes_bn())
      for (size_t i = 0; i < n; ++i) {
        a[i] = i < h_pl_bn_scale.size() ? ldexpf(h_pl_bn_scale[i], -e) : 0.f;
        b[i] = i < h_pl_bn_shift.size() ? ldexpf(h_pl_bn_shift[i], -e) : 0.f;
      }
    XDET_HIP(hipMemcpy(d_pl_scale, a.data(), n * sizeof(float), hipMemcpyHostToDevice));
    XDET_HIP(hipMemcpy(d_pl_shift, b.data(), n * sizeof(float), hipMemcpyHostToDevice));
    out_exp = e;
    return XDET_OK;
  }

  ~ConvLayer() override {
    if (d_ks_partial && ks_owns_scratch) { (void)hipFree(d_ks_partial); (void)hipFree(d_ks_ticket); }
    if (d_pl_scale) (void)hipFree(d_pl_scale);
    if (d_pl_shift) (void)hipFree(d_pl_shift);
    if (d_zeros) (void)hipFree(d_zeros);
    if (d_wt_hi_b) (void)hipFree(d_wt_hi_b);
    if (d_wt_lo_b) (void)hipFree(d_wt_lo_b);
    if (d_wt_x8_b) (void)hipFree(d_wt_x8_b);
    if (d_wt_hi) (void)hipFree(d_wt_hi);
    if (d_wt_lo) (void)hipFree(d_wt_lo);
    if (d_wt) (void)hipFree(d_wt);
    if (d_scale) (void)hipFree(d_scale);
    if (d_shift) (void)hipFree(d_shift);
  }

  // groups_ > 1: w_hwio is [groups][cin][cout] (1x1), scale/shift are [groups][cout]
  int init(int kh_, int kw_, int cin_, int cout_, int stride_, int dil_, int pad_mode_, int pad_t_, int pad_l_,
           const float* w_hwio, const float* scale, const float* shift, int relu_out_, int groups_ = 1) {
    XDET_REQUIRE(kh_ > 0 && kw_ > 0 && cin_ > 0 && cout_ > 0 && stride_ > 0 && dil_ > 0, "conv: bad geometry");
    XDET_REQUIRE(groups_ >= 1 && (groups_ == 1 || (kh_ == 1 && kw_ == 1 && stride_ == 1 && cin_ % 32 == 0)),
                 "conv: grouped GEMMs are 1x1, stride 1, cin % 32 == 0");
    groups = groups_;
    XDET_REQUIRE(pad_mode_ >= 0 && pad_mode_ <= 2, "conv: pad_mode must be 0|1|2");
    XDET_REQUIRE(w_hwio != nullptr, "conv: kernel is NULL");
    kind = 1;
    kh = kh_; kw = kw_; cin = cin_; cout = cout_; stride = stride_; dil = dil_;
    pad_mode = pad_mode_; pad_t = pad_t_; pad_l = pad_l_; relu_out = relu_out_;
    small_cin = cin <= 4;
    cin_p = small_cin ? 4 : round_up(cin, 32);
    kp = round_up(kh * kw * cin_p, 32);
    n_tile = round_up(cout, 64) < round_up(cout, 128) ? 64 : 128;
    cout_pad = round_up(cout, n_tile);
    const size_t G = (size_t)groups, mat = (size_t)cout_pad * kp;
    std::vector<float> wt(G * mat, 0.f), sc(G * cout_pad, 0.f), sh(G * cout_pad, 0.f);
    for (size_t g = 0; g < G; ++g) {
      for (int t = 0; t < kh * kw; ++t)
        for (int ci = 0; ci < cin; ++ci) {
          const float* src = w_hwio + ((g * kh * kw + t) * cin + ci) * cout;
          float* dst = &wt[g * mat + (size_t)t * cin_p + ci];
          for (int co = 0; co < cout; ++co) dst[(size_t)co * kp] = src[co];
        }
      for (int co = 0; co < cout; ++co) {
        sc[g * cout_pad + co] = scale ? scale[g * cout + co] : 1.f;
        sh[g * cout_pad + co] = shift ? shift[g * cout + co] : 0.f;
      }
    }
    precision = g_default_precision;
    XDET_REQUIRE(groups == 1 || precision != PREC_F32, "conv: grouped GEMMs need a split-precision mode");
    XDET_HIP(hipGetDevice(&device));
    if (precision != PREC_F32 && !small_cin) {
      XDET_HIP(hipMalloc(reinterpret_cast<void**>(&d_zeros), 256));
      XDET_HIP(hipMemset(d_zeros, 0, 256));
    }
    if (precision == PREC_F32) {
      XDET_TRY(upload(wt, &d_wt));
    } else {
      // per-output-channel power-of-two pre-scale so that max|w| lands in [512, 1024): w_hi cannot
      // overflow f16 and w_lo (~2^-11 |w|) stays a normal f16; undone exactly in the epilogue scale
      std::vector<unsigned short> hi(wt.size()), lo(precision == PREC_F16X3 ? wt.size() : 0);
      for (size_t gc = 0; gc < G * cout_pad; ++gc) {
        float* row = &wt[gc * kp];
        float mx = 0.f;
        for (int k = 0; k < kp; ++k) mx = std::max(mx, std::fabs(row[k]));
        int e = 0;
        if (mx > 0.f) (void)frexpf(mx, &e);
        const int sh_k = mx > 0.f ? 10 - e : 0;
        for (int k = 0; k < kp; ++k) {
          const float wv = ldexpf(row[k], sh_k);
          const unsigned short h = f32_to_f16_rne(wv);
          hi[gc * kp + k] = h;
          if (precision == PREC_F16X3) lo[gc * kp + k] = f32_to_f16_rne(wv - f16_to_f32(h));
        }
        sc[gc] = ldexpf(sc[gc], -sh_k);
      }
      std::vector<float>().swap(wt);
      if (groups == 1) {      // [Cout_pad][Kp] copies: the register-staged kernel (strided / small-cin convs)
        XDET_HIP(hipMalloc(reinterpret_cast<void**>(&d_wt_hi), hi.size() * 2));
        XDET_HIP(hipMemcpy(d_wt_hi, hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
        if (precision == PREC_F16X3) {
          XDET_HIP(hipMalloc(reinterpret_cast<void**>(&d_wt_lo), lo.size() * 2));
          XDET_HIP(hipMemcpy(d_wt_lo, lo.data(), lo.size() * 2, hipMemcpyHostToDevice));
        }
      }
      if (!small_cin) {       // K-blocked [g][Kp/32][Cout_pad][32] copies: the LDS-DMA kernel
        std::vector<unsigned short> blk(hi.size());
        auto kblock = [&](const std::vector<unsigned short>& src) {
          for (size_t g = 0; g < G; ++g)
            for (int co = 0; co < cout_pad; ++co) {
              const unsigned short* r = &src[(g * cout_pad + co) * kp];
              for (int k = 0; k < kp; ++k) blk[g * mat + ((size_t)(k >> 5) * cout_pad + co) * 32 + (k & 31)] = r[k];
            }
        };
        kblock(hi);
        XDET_HIP(hipMalloc(reinterpret_cast<void**>(&d_wt_hi_b), blk.size() * 2));
        XDET_HIP(hipMemcpy(d_wt_hi_b, blk.data(), blk.size() * 2, hipMemcpyHostToDevice));
        if (precision == PREC_F16X3) {
          kblock(lo);
          XDET_HIP(hipMalloc(reinterpret_cast<void**>(&d_wt_lo_b), blk.size() * 2));
          XDET_HIP(hipMemcpy(d_wt_lo_b, blk.data(), blk.size() * 2, hipMemcpyHostToDevice));
          if (kh == 1 && kw == 1 && stride == 1 && groups == 1) {   // a pointwise layer may be fed x8 planes
            unsigned char* b8 = reinterpret_cast<unsigned char*>(blk.data());
            for (int co = 0; co < cout_pad; ++co)
              for (int k = 0; k < kp; ++k) {
                unsigned char* rec = b8 + ((size_t)(k >> 5) * cout_pad + co) * 64;
                rec[k & 31] = f32_to_e4m3(std::ldexp(f16_to_f32(lo[(size_t)co * kp + k]), 9));
                rec[32 + (k & 31)] = f32_to_e4m3(std::ldexp(f16_to_f32(hi[(size_t)co * kp + k]), -2));
              }
            XDET_HIP(hipMalloc(reinterpret_cast<void**>(&d_wt_x8_b), blk.size() * 2));
            XDET_HIP(hipMemcpy(d_wt_x8_b, blk.data(), blk.size() * 2, hipMemcpyHostToDevice));
          }
        }
      }
    }
    XDET_TRY(upload(sc, &d_scale));
    XDET_TRY(upload(sh, &d_shift));
    h_scale = sc;
    return XDET_OK;
  }

  void out_shape(int H, int W, int* Ho, int* Wo, int* pt, int* pl) const {
    if (pad_mode == 1) {
      same_pad(H, kh, stride, dil, pt, Ho);
      same_pad(W, kw, stride, dil, pl, Wo);
    } else {
      *pt = pad_mode == 2 ? pad_t : 0;
      *pl = pad_mode == 2 ? pad_l : 0;
      // explicit padding is symmetric-or-trailing as in resnet_v2.fixed_padding: total = k-1
      const int tot_h = pad_mode == 2 ? (kh - 1) * dil : 0, tot_w = pad_mode == 2 ? (kw - 1) * dil : 0;
      *Ho = (H + tot_h - ((kh - 1) * dil + 1)) / stride + 1;
      *Wo = (W + tot_w - ((kw - 1) * dil + 1)) / stride + 1;
    }
  }
  int ld_in() const { return small_cin ? 4 : round_up(cin, 32); }
  int ld_out() const { return round_up(cout, 32); }
  double flops(int H, int W) const {
    int Ho, Wo, a, b;
    out_shape(H, W, &Ho, &Wo, &a, &b);
    return 2.0 * Ho * Wo * (double)cin * cout * kh * kw;
  }

  int forward(const float* in, int N, int H, int W, int ldi, float* out, int ldo, const float* res, int relu_in,
              hipStream_t s, const unsigned short* in_hi = nullptr, const unsigned short* in_lo = nullptr,
              const unsigned short* zeros = nullptr, unsigned short* out_hi = nullptr,
              unsigned short* out_lo = nullptr, int planes_relu = 0, const float* pl_scale = nullptr,
              const float* pl_shift = nullptr, int group_rows = 0, int x8 = 0, int x8_exp = 0, int group_live_rows = 0) const {
    XDET_REQUIRE(ldi == ld_in(), "conv: ld_in must be round_up(cin,32) (4 for cin<=4)");
    XDET_REQUIRE(ldo == ld_out(), "conv: ld_out must be round_up(cout,32)");
    ConvParams p;
    p.in = in; p.wt = d_wt; p.wt_hi = d_wt_hi; p.wt_lo = d_wt_lo; p.out = out; p.scale = d_scale; p.shift = d_shift; p.res = res;
    p.N = N; p.H = H; p.W = W; p.ldi = ldi; p.ldo = ldo; p.ldr = ldo;
    out_shape(H, W, &p.Ho, &p.Wo, &p.pad_t, &p.pad_l);
    XDET_REQUIRE(p.Ho > 0 && p.Wo > 0, "conv: empty output");
    p.Cin_p = cin_p; p.Kp = kp; p.Cout_pad = cout_pad;
    p.KH = kh; p.KW = kw; p.stride = stride; p.dil = dil;
    p.M = N * p.Ho * p.Wo;
    p.relu_in = relu_in; p.relu_out = relu_out;
    p.in_hi = in_hi; p.in_lo = in_lo; p.zeros = zeros;
    p.out_hi = precision == PREC_F32 ? nullptr : out_hi; p.out_lo = out_lo; p.planes_relu = planes_relu;
    p.pl_scale = pl_scale; p.pl_shift = pl_shift; p.pl_c32 = pl_c32;
    p.group_rows = 0; p.group_wt_stride = 0;
    if (groups > 1) {
      XDET_REQUIRE(in_hi && group_rows > 0 && group_rows % 128 == 0 && (int64_t)group_rows * groups == p.M,
                   "conv(grouped): M must be groups * group_rows, group_rows a multiple of 128, input as planes");
      p.group_rows = group_rows;
      p.group_wt_stride = (long long)cout_pad * kp;
      p.group_live_rows = group_live_rows;
    }
    if (precision == PREC_F32) return launch_conv_mfma_f32(p, small_cin, n_tile, s);
    if (in_hi) {   // A operand already split into f16 planes by its producer: LDS-DMA kernel
      XDET_REQUIRE(!small_cin && relu_in == 0, "conv(dma): needs >= 32 input channels and no ReLU-on-load");
      p.wt_hi = d_wt_hi_b; p.wt_lo = d_wt_lo_b;   // K-blocked copies
      if (x8) {                                   // `in_lo` holds [hi8 | lo8] records: the cross terms run on the fp8 MFMA
        XDET_REQUIRE(d_wt_x8_b && precision == PREC_F16X3 && ksplit < 1, "conv: x8 planes need a pointwise f16x3 layer without a split-K");
        p.wt_lo = d_wt_x8_b; p.x8 = 1; p.x8_exp = x8_exp;
      }
      if (ksplit >= 1 && groups == 1) {
        // a split-K layer's summation tree is part of its definition: a batch whose planes leave the kernel's 4 GiB
        // addressing is an error, not a silent change of kernel family (results must not depend on the batch).  ONE range
        // is the plain kernels' reduction (bit-identical): such a layer just runs on them.
        if (!conv_ksplit_supported(kh, kw, (int64_t)N * H * W, ldi, cin_p, cout_pad)) {
          if (ksplit == 1 && !ks_narrow) return launch_conv_mfma_dma(p, n_tile, precision == PREC_F16X3 ? 3 : 1, s);
          XDET_REQUIRE(ksplit == 1, "conv(ksplit): this batch's planes or weights exceed the split-K kernel's 4 GiB addressing (or the "
                                    "filter is not 1x1 / 3x3); a split reduction cannot change kernel family: run smaller batches");
          return launch_conv_mfma_dma(p, 64, precision == PREC_F16X3 ? 3 : 1, s);
        }
        p.ksplit = ksplit; p.ks_partial = d_ks_partial; p.ks_ticket = d_ks_ticket;
        return launch_conv_mfma_ksplit(p, cout_pad % 128 == 0 && !ks_narrow ? 128 : 64, precision == PREC_F16X3 ? 3 : 1, ks_mode, ks_tiles, s);
      }
      return launch_conv_mfma_dma(p, n_tile, precision == PREC_F16X3 ? 3 : 1, s);
    }
    return launch_conv_mfma_split(p, small_cin, n_tile, precision == PREC_F16X3 ? 3 : 1, s);
  }
  // can this layer consume pre-split f16 planes (conv_mfma_dma.hip)?
  bool dma_capable() const { return precision != PREC_F32 && !small_cin; }
};

struct DepthwiseLayer : LayerBase {
  int C, dil, ld;
  float* d_w = nullptr;
  std::vector<float> h_w;        // host copy of the taps: a planes-producing depthwise carries its activation pre-scale in them
  int out_exp = 0;
  int set_out_exp(int e) {       // every partial sum of the FMA chain scales exactly with a power of two
    std::vector<float> v(h_w);
    for (float& x : v) x = ldexpf(x, -e);
    XDET_HIP(hipMemcpy(d_w, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    out_exp = e;
    return XDET_OK;
  }
  ~DepthwiseLayer() override { if (d_w) (void)hipFree(d_w); }
  int init(int C_, int dil_, const float* w33c1) {
    XDET_REQUIRE(C_ > 0 && dil_ > 0 && w33c1, "depthwise: bad arguments");
    kind = 2;
    XDET_HIP(hipGetDevice(&device));
    C = C_; dil = dil_; ld = round_up(C, 32);
    std::vector<float> w((size_t)9 * ld, 0.f);
    for (int t = 0; t < 9; ++t)
      for (int c = 0; c < C; ++c) w[(size_t)t * ld + c] = w33c1[(size_t)t * C + c];
    h_w = w;
    return upload(w, &d_w);
  }
  int forward(const float* in, int N, int H, int W, int ld_, float* out, int relu_in, hipStream_t s) const {
    XDET_REQUIRE(ld_ == ld, "depthwise: ld must be round_up(C,32)");
    return launch_depthwise3x3(in, d_w, out, N, H, W, C, ld, dil, relu_in, s);
  }
};

// ---------------------------------------------------------------------------------------
// static launch plans
// ---------------------------------------------------------------------------------------
struct Buf {
  float* p = nullptr;
  unsigned short *hi = nullptr, *lo = nullptr;   // optional split-precision f16 planes of the same tensor
  bool planes_relu = false;                      // the planes hold relu(tensor)
  bool no_f32 = false;                           // only the planes are ever written (p stays unused)
  int pidx = -1;                                 // entry of Plan::pscales: the planes hold x * 2^-exp
  int H = 0, W = 0, C = 0, ld = 0;
  size_t per_image() const { return (size_t)H * W * ld; }
};

struct Op {
  std::string name;
  int stage;
  double flops;      // algorithmic dense 2*MAC per image (0 for non-contraction ops, < 0: auxiliary pass of one)
  std::function<int(int, hipStream_t)> run;
  double mfma_flops = -1.0;   // FLOPs the op actually executes on the matrix cores per image, all products counted
                              // (-1: the default, products-per-term x flops; the spectral GEMMs and VALU convs set it)
};

struct ProfRec { int op; hipEvent_t a, b; };

struct Plan {
  int plan_kind = -1; // 0 = LightHeadNet, 1 = ResNetTrunk: the C-ABI takes void* handles and checks what it was given
  int device = 0;     // HIP device the plan's weights and workspace live on (current device at creation)
  int max_batch = 1;
  bool profiling = false;
  std::vector<ProfRec> prof;
  std::vector<void*> allocs;
  std::vector<std::pair<const float*, size_t>> f32_bufs;   // every activation tensor of new_buf(): (pointer, floats per image)
  struct PlanesRec { const unsigned short* hi; int64_t pix_per_image; int ld; };
  std::vector<PlanesRec> planes_bufs;                      // ... and of new_planes()
  int net_precision = PREC_F32;                            // precision mode the plan was built in
  std::vector<std::unique_ptr<LayerBase>> layers;
  std::vector<Op> ops;
  WeightMap w;

  virtual ~Plan() {
    for (void* p : allocs) (void)hipFree(p);
  }
  size_t allocated_bytes = 0;
  int alloc_bytes(size_t bytes, void** out, bool zero = true) {
    XDET_HIP(hipMalloc(out, std::max<size_t>(bytes, 256)));
    allocated_bytes += std::max<size_t>(bytes, 256);
    if (zero) XDET_HIP(hipMemset(*out, 0, std::max<size_t>(bytes, 256)));
    allocs.push_back(*out);
    return XDET_OK;
  }
  // ---- workspace ----
  // One block per intermediate tensor, EXCEPT that a builder may hand a tensor's block back once its last consumer has been
  // planned (release_f32 / release_planes); a later tensor that fits takes it over (best fit, at most 2x its size): the
  // middle flow's 24 x (planes pair + f32 block output) live in a handful of blocks, the entry flow's blocks serve the
  // exit flow and the head.
  // Ops of one stream run in plan order, so a recycled block is never live twice; the side stream (RPN branch) has its own
  // pool.  Named buffers of the graph-builder API (mid_x, out, feat, ...) are never released.  Option "workspace" = "reuse"
  // (default) | "ssa"; check_range needs every tensor intact after the forward and turns reuse off.
  bool reuse_workspace = true;
  // "workspace" = "poison" (tests): as "reuse", and in front of the first op planned after a recycled f32 block was handed
  // out, everything of that block BEYOND the new tensor (its loader slack and the rest of the larger predecessor) is filled
  // with NaN bits on every forward -- if any consumer used a byte from behind its tensor, the detections would differ from
  // the one-block-per-tensor net's (tests/test_gpu_e2e.py)
  bool poison_recycled = false;
  struct PoisonRec { size_t op; float* at; size_t words; };
  std::vector<PoisonRec> poison;
  std::map<void*, size_t> ws_bytes;                      // blocks handed out by take() that are live
  std::multimap<size_t, void*> ws_free[2][2];            // released blocks by size, per stream pool and kind (f32 / planes)
  std::map<void*, int> ws_kind;
  int ws_pool = 0;                                       // 0 = main stream, 1 = side stream
  size_t ws_recycled_bytes = 0;
  // f32 tensors: best fit, at most twice the tensor's size (consumers address a tensor by its own shape -- buffer
  // descriptors are sized to the tensor, not to the block -- so what lies behind it is never interpreted; padded channels
  // are written by every producer).  Planes: a released planes block of EXACTLY the same size only -- the last 16-pixel
  // group of a planes tensor has pixel slots nobody writes, which the calibration's measurement and the GEMM's discarded
  // tail rows do read: they must hold what a tensor of the same shape left there (finite f16), not another tensor's bytes.
  int take(size_t bytes, void** out, int kind = 0, size_t slack_bytes = 0) {
    bytes = std::max<size_t>(bytes, 256);
    if (reuse_workspace) {
      auto& fl = ws_free[ws_pool][kind];
      auto it = kind == 1 ? fl.find(bytes) : fl.lower_bound(bytes);
      if (it != fl.end() && it->first <= 2 * bytes) {
        *out = it->second;
        const size_t block = it->first;
        fl.erase(it);
        ws_bytes[*out] = block;                          // (released again under its own size)
        ws_recycled_bytes += bytes;
        if (poison_recycled && kind == 0 && bytes > slack_bytes)
          poison.push_back({ops.size(), reinterpret_cast<float*>(static_cast<char*>(*out) + (bytes - slack_bytes)),
                            (block - (bytes - slack_bytes)) / 4});
        return XDET_OK;
      }
    }
    XDET_TRY(alloc_bytes(bytes, out));
    ws_bytes[*out] = bytes;
    ws_kind[*out] = kind;
    return XDET_OK;
  }
  void give(void* p) {
    if (!reuse_workspace || !p) return;
    auto it = ws_bytes.find(p);
    if (it == ws_bytes.end()) return;                    // not a take() block, or released already
    ws_free[ws_pool][ws_kind[p]].insert({it->second, p});
    ws_bytes.erase(it);
  }
  void release_f32(const Buf& b) { give(b.p); }
  void release_planes(const Buf& b) { give(b.hi); give(b.lo); }
  int new_buf(int H, int W, int C, Buf* b) {
    b->H = H; b->W = W; b->C = C;
    b->no_f32 = false;                             // (a Buf that was copied from a planes-only tensor must not keep that flag)
    b->ld = C <= 4 ? 4 : round_up(C, 32);
    // +128 floats of slack: the conv loader may read a full 32-channel slice of the last pixel
    XDET_TRY(take(((size_t)max_batch * b->per_image() + 128) * sizeof(float), reinterpret_cast<void**>(&b->p), 0, 128 * sizeof(float)));
    f32_bufs.emplace_back(b->p, b->per_image());
    return XDET_OK;
  }
  const HostTensor* find(const std::string& name) const {
    auto it = w.find(name);
    return it == w.end() ? nullptr : &it->second;
  }
  int need(const std::string& name, const HostTensor** t, std::initializer_list<int64_t> dims) const {
    *t = find(name);
    if (!*t) {
      set_last_error("missing weight: " + name);
      return XDET_ERR_STATE;
    }
    if ((*t)->dims != std::vector<int64_t>(dims)) {
      set_last_error("weight has wrong shape: " + name);
      return XDET_ERR_INVALID_ARG;
    }
    return XDET_OK;
  }
  // inference BN folded to y = x*scale + shift; an optional conv bias is folded in too
  int fold_bn(const std::string& bn, int C, float eps, const float* bias, std::vector<float>* scale,
              std::vector<float>* shift) const {
    const HostTensor *g, *b, *m, *v;
    XDET_TRY(need(bn + "/gamma", &g, {C}));
    XDET_TRY(need(bn + "/beta", &b, {C}));
    XDET_TRY(need(bn + "/moving_mean", &m, {C}));
    XDET_TRY(need(bn + "/moving_variance", &v, {C}));
    scale->resize(C);
    shift->resize(C);
    for (int c = 0; c < C; ++c) {
      const float sc = g->v[c] / std::sqrt(v->v[c] + eps);
      (*scale)[c] = sc;
      (*shift)[c] = b->v[c] - m->v[c] * sc + (bias ? bias[c] * sc : 0.f);
    }
    return XDET_OK;
  }

  // ---- activation pre-scale of the split-precision operands ----
  // An f16 hi part overflows beyond 65504 (the reference computes in f32 everywhere and has BN-less edges:
  // net/xception_body.py:381-400,450-475).  Every tensor that exists as split planes -- in HBM, or inside a fused
  // separable block -- has a power-of-two exponent e: the planes hold x * 2^-e and the consuming contraction folds 2^e
  // back into its epilogue scale, both exact.  e = 0 (no change at all) unless calibrate() measures a tensor above
  // kRangeTarget on a calibration batch.
  struct PlaneScale {
    std::string name;
    int exp = 0;
    const unsigned short* hi = nullptr;               // measured on the planes themselves ...
    std::function<int64_t(int)> halves;               // ... over this many halves for a batch of N
    const float* src = nullptr;                       // or (the operand a fused block keeps on the CU) bounded from its f32
    size_t src_per_image = 0;                         // input: max|x| (after the input ReLU) * bound
    int src_relu = 0;
    float bound = 1.f;
    std::vector<std::function<int(int)>> apply;       // re-derive the device parameters that carry 2^-e / 2^e
    std::function<int(hipStream_t)> clear;            // optional: zero the padding a measurement would otherwise scan
    // x8 form (option "cross" = "fp8"; conv_params.h): planes written by a depthwise tile kernel for ONE pointwise consumer on
    // the LDS-DMA kernels.  Both ends are fixed at build time (x8_ok); the form is switched on by the calibration pass, which
    // measures the tensor and chooses x8_exp so that its largest |hi| * 2^-x8_exp lands in (128, 256].
    bool x8_cand = false, x8_ok = false, x8_on = false;
    int x8_exp = 0;
    float last_max = 0.f;                             // largest magnitude of the operand in the last calibration pass
    int op_index = -1;                                // measured right behind this op of the plan (-1: after the whole forward)
  };
  std::function<int(int, hipStream_t)> after_op;      // calibration hook: called by run_stage behind every op
  bool cross8 = false;                                // option "cross" = "f16" | "fp8"
  bool plane_x8(int pidx, int* e) const {
    const bool on = pidx >= 0 && pscales[pidx].x8_ok && pscales[pidx].x8_on;
    *e = on ? pscales[pidx].x8_exp : 0;
    return on;
  }
  std::vector<PlaneScale> pscales;
  std::vector<const float*> f32_split_inputs;          // f32 tensors a register-split conv reads (no pre-scale exists for them)
  // check_range: what an f32 tensor may hold.  Only two kinds of f32 tensors are ever turned into f16 without a planes
  // tensor of their own (which range_check_planes sees directly): the input of a register-split conv (|x| <= 65504) and
  // the input of a fused separable block, whose depthwise result is split on the CU (relu?(x) * sum|w| * 2^-e <= 65504).
  // Everything else -- including tensors whose PLANES carry a pre-scale, and the spectral products y1 / y2, which are
  // never split -- is legitimately beyond 65504 after a calibration and is only checked for NaN / inf.
  void f32_limit(const float* p, float* limit, int* relu) const {
    *limit = 3.4028235e38f;
    *relu = 0;
    bool any = false, all_relu = true;
    for (const float* q : f32_split_inputs)
      if (q == p) { *limit = std::min(*limit, 65504.f); any = true; all_relu = false; }
    for (const PlaneScale& ps : pscales)
      if (ps.src == p && ps.bound > 0.f) {
        *limit = std::min(*limit, ldexpf(65504.f / ps.bound, ps.exp));
        any = true;
        all_relu = all_relu && ps.src_relu != 0;
      }
    *relu = any && all_relu;
  }
  static constexpr float kRangeTarget = 4096.f;       // 16x below the f16 maximum: headroom for images unlike the calibration batch
  float pmul(int pidx) const { return pidx >= 0 ? ldexpf(1.f, -pscales[pidx].exp) : 1.f; }
  int set_plane_exp(int pidx, int e) {
    pscales[pidx].exp = e;
    for (auto& f : pscales[pidx].apply) XDET_TRY(f(e));
    return XDET_OK;
  }

  // Choose the activation pre-scale exponents from a calibration batch.  Pass by pass: run the forward (`run`),
  // take the largest magnitude of every split-precision operand (as stored, i.e. already scaled), and raise the exponent
  // of the tensors above kRangeTarget.  A tensor that overflowed (inf / NaN in its hi plane) invalidates everything
  // computed from it, so a pass stops adjusting at the first such tensor (+8 binades) and the next pass re-measures.
  // Plans whose activations stay below the target keep every exponent at 0: nothing changes, bit for bit.
  int calibrate_planes(int N, hipStream_t s, int* n_scaled, const std::function<int(hipStream_t)>& run) {
    if (n_scaled) *n_scaled = 0;
    if (pscales.empty()) return XDET_OK;
    unsigned* d_max = nullptr;
    XDET_HIP(hipMalloc(reinterpret_cast<void**>(&d_max), pscales.size() * sizeof(unsigned)));
    std::vector<unsigned> h_max(pscales.size());
    int rc = XDET_OK;
    for (const PlaneScale& p : pscales)
      if (p.clear && rc == XDET_OK) rc = p.clear(s);
    const int max_passes = (int)pscales.size() + 4;
    const bool verbose = getenv("XDET_CALIBRATE_VERBOSE") != nullptr;
    int pass = 0;
    for (; pass < max_passes && rc == XDET_OK; ++pass) {
      if ((rc = hipMemsetAsync(d_max, 0, pscales.size() * sizeof(unsigned), s) == hipSuccess ? XDET_OK : XDET_ERR_HIP) != XDET_OK) break;
      // tensors with a producing op in the plan are measured right behind it, on the stream it ran on (their workspace
      // block may belong to another tensor by the end of the forward); the rest after the whole forward
      auto measure = [&](size_t i, hipStream_t st) {
        const PlaneScale& p = pscales[i];
        return p.hi ? launch_absmax_planes(p.hi, p.halves(N), d_max + i, st)
                    : launch_absmax_f32(p.src, (int64_t)N * (int64_t)p.src_per_image, p.src_relu, d_max + i, st);
      };
      after_op = [&](int op, hipStream_t st) {
        for (size_t i = 0; i < pscales.size(); ++i)
          if (pscales[i].op_index == op) XDET_TRY(measure(i, st));
        return (int)XDET_OK;
      };
      rc = run(s);
      after_op = nullptr;
      if (rc != XDET_OK) break;
      for (size_t i = 0; i < pscales.size() && rc == XDET_OK; ++i)
        if (pscales[i].op_index < 0) rc = measure(i, s);
      if (rc != XDET_OK) break;
      if (hipMemcpyAsync(h_max.data(), d_max, h_max.size() * sizeof(unsigned), hipMemcpyDeviceToHost, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess) { rc = XDET_ERR_HIP; break; }
      bool changed = false;
      for (size_t i = 0; i < pscales.size() && rc == XDET_OK; ++i) {
        const PlaneScale& p = pscales[i];
        float m;                                     // magnitude of the operand as the MFMAs would see it now
        bool broken;
        if (p.hi) {
          broken = h_max[i] >= 0x7c00u;
          m = broken ? 0.f : f16_to_f32((unsigned short)h_max[i]);
        } else {
          broken = h_max[i] >= 0x7f800000u;
          float x;
          memcpy(&x, &h_max[i], 4);
          m = broken ? 0.f : ldexpf(x * p.bound, -p.exp);
          if (!broken && !(m <= 3.0e38f)) broken = true;
        }
        pscales[i].last_max = broken ? 0.f : m;
        if (verbose && (broken || m > kRangeTarget))
          fprintf(stderr, "xdet calibrate: pass %d  %-70s exp %d  max %s%g\n", pass, p.name.c_str(), p.exp,
                  broken ? "inf/NaN " : "", (double)m);
        if (broken) {
          rc = set_plane_exp((int)i, p.exp + 8);
          changed = true;
          break;                                     // downstream tensors were computed from garbage: re-measure
        }
        if (m > kRangeTarget) {
          int e = 0;
          (void)frexpf(m / kRangeTarget, &e);        // m / target in [2^(e-1), 2^e)
          rc = set_plane_exp((int)i, p.exp + e);
          changed = true;
        }
      }
      if (!changed) break;
    }
    (void)hipFree(d_max);
    XDET_TRY(rc);
    if (pass >= max_passes) {
      set_last_error("calibrate: the activation ranges did not settle (non-finite inputs or weights?)");
      return XDET_ERR_STATE;
    }
    if (n_scaled) {
      int n = 0;
      for (const PlaneScale& p : pscales) n += p.exp != 0;
      *n_scaled = n;
    }
    for (PlaneScale& p : pscales)                    // the settled pass measured every tensor as its planes hold it
      if (p.x8_ok) {
        int e = 0;
        if (p.last_max > 0.f) {
          (void)frexpf(p.last_max / 256.f, &e);      // last_max / 256 in [2^(e-1), 2^e)
          if (ldexpf(256.f, e - 1) == p.last_max) --e;   // exactly a power of two: (128, 256] includes its upper end
        }
        p.x8_exp = e;
        p.x8_on = true;
      }
    return XDET_OK;
  }

  // ---- split-precision planes ----
  unsigned short* zeros = nullptr;
  int get_zeros() {
    if (!zeros) XDET_TRY(alloc_bytes(256, reinterpret_cast<void**>(&zeros)));
    return XDET_OK;
  }
  int new_planes(Buf* b) {
    // [pixels/16][ld/32][16][32]: the pixel count is rounded up to a whole 16-pixel group
    const size_t bytes = ((size_t)cdiv((int64_t)max_batch * b->H * b->W, 16) * 16 * b->ld + 256) * sizeof(unsigned short);
    XDET_TRY(take(bytes, reinterpret_cast<void**>(&b->hi), 1));
    XDET_TRY(take(bytes, reinterpret_cast<void**>(&b->lo), 1));
    planes_bufs.push_back({b->hi, (int64_t)b->H * b->W, b->ld});
    PlaneScale ps;
    ps.hi = b->hi;
    ps.op_index = (int)ops.size();   // every caller pushes the producing op next: a calibration measures the planes right
                                     // behind it (their block may be recycled by a later tensor)
    const int64_t pix = (int64_t)b->H * b->W;
    const int ldp = b->ld;
    ps.halves = [pix, ldp](int N) { return cdiv((int64_t)N * pix, 16) * 16 * ldp; };
    // the last 16-pixel group of a batch smaller than max_batch also holds pixels of the next image, which only an
    // earlier, larger batch ever wrote: zero the planes before a calibration measures them
    unsigned short* hi_p = b->hi;
    ps.clear = [hi_p, bytes](hipStream_t st) { XDET_HIP(hipMemsetAsync(hi_p, 0, bytes, st)); return (int)XDET_OK; };
    b->pidx = (int)pscales.size();
    pscales.push_back(ps);
    return get_zeros();
  }
  // element-wise f32 -> (hi, lo) f16 planes (optionally through a ReLU) for a tensor whose producer
  // only wrote f32; returns a Buf that aliases `in` and carries the planes
  int add_split(const std::string& name, int stage, const Buf& in, int relu, Buf* out) {
    *out = in;
    out->hi = out->lo = nullptr;
    XDET_TRY(new_planes(out));
    const Buf i = in, o = *out;
    pscales[o.pidx].name = name;
    ops.push_back({name, stage, 0.0, [=](int N, hipStream_t s) {
                     return launch_split_f32(i.p, o.hi, o.lo, (int64_t)N * i.H * i.W, i.ld, relu, s, pmul(o.pidx));
                   }});
    return XDET_OK;
  }

  // ---- op builders ----
  // Set by a builder right before add_conv / conv_bn when it knows the output feeds a conv on the
  // LDS-DMA path: 1 = the conv also writes its output as split planes (the consumer then needs no
  // split pass), 2 = the planes hold relu(output) (for a consumer that applies ReLU on load),
  // 3 = planes ONLY: every consumer is on the LDS-DMA path, the f32 tensor is never written.
  // With emit_bn_scale/shift set (device arrays, ld floats) the planes copy is relu(out*scale+shift):
  // a following BN+ReLU pre-activation folded into this conv's epilogue.
  int emit_planes_next = 0;
  // With emit_planes_next = 3: the planes go into channel blocks [0, ld_out/32) of THIS wider planes tensor (same pixels) instead
  // of a tensor of their own -- the left part of a concatenated operand [this conv's output | what another op wrote]
  const Buf* emit_planes_into = nullptr;
  std::vector<float> emit_bn_scale, emit_bn_shift;   // host, ld floats each (empty: no folded BN)
  // Set by a builder right before add_conv: the planes copy of this conv's output is only read by the three-launch form of
  // the next block -- a forward in which planes_dropped() says that block runs fused (resnet_bneck.hip makes its own
  // pre-activation from the f32 tensor) does not write it
  bool planes_optional_next = false;
  std::vector<char> planes_drop_ok;      // per registered conv: its one planes reader turned out to be a kernel that does not read them
  int last_optional_idx = -1;            // index the last add_conv registered (-1: it registered nothing)
  virtual bool planes_dropped() const { return false; }
  // Split-K policy (conv_mfma_ksplit.hip).  ksplit_design_batch > 0: a conv whose grid at THAT batch size (a constant
  // of the plan: 8 for the ResNet trunk = BASELINE config 2, 1 for the detector's single-image latency path) leaves
  // most of the 256 CUs idle gets its K steps cut into S ranges, S = what fills the chip at the design batch, at
  // least 8 steps per range.  S depends on the layer and the plan constant only -- never on the batch of a call.
  int ksplit_design_batch = 0;
  bool ksplit_next = false;            // builders of plans that split only some layers set this before add_conv
  bool ksplit_all = false;
  static int pow2_floor(int64_t v) { int r = 1; while ((int64_t)r * 2 <= v) r *= 2; return r; }
  int maybe_ksplit(ConvLayer* L, int Hi, int Wi, int Ho, int Wo) {
    const bool want = ksplit_design_batch > 0 && (ksplit_all || ksplit_next);
    ksplit_next = false;
    if (!want || !L->dma_capable() || L->groups != 1 || L->cout_pad % 64 != 0) return XDET_OK;
    if (!conv_ksplit_supported(L->kh, L->kw, (int64_t)max_batch * Hi * Wi, L->ld_in(), L->cin_p, L->cout_pad)) return XDET_OK;
    const int bn = L->cout_pad % 128 == 0 ? 128 : 64, nk = L->kp / 32;
    const int64_t tiles = conv_ksplit_tiles((int64_t)ksplit_design_batch * Ho * Wo, L->cout_pad, bn);
    constexpr int max_s = 8, max_tiles = 128;     // at most 8 ranges; only layers of at most half a round of tiles are split
    // A layer of about one round of 128 x 128 tiles with a long K loop (ResNet-50 stage 2 at batch 8: the 3x3 convs and the
    // opening 1x1 convs, 225 tiles, 36 / 16 steps) is not short of workgroups but of pipeline: the two-stage kernels pay
    // ~1.2 us per 32-deep step there; the split-K kernel's four-stage ring (taps unrolled at compile time) runs the same
    // reduction with ONE range -- bit-identical to the plain kernels -- at ~0.65 us per step (trunk 1.99 -> 1.92 ms, same-box
    // A/B)
    constexpr int one_tiles = 256;
    if (bn == 128 && tiles > max_tiles && tiles <= one_tiles && nk >= 16) {
      XDET_TRY(L->enable_ksplit(1, 1));
      return XDET_OK;
    }
    if (tiles > max_tiles || nk < 16) return XDET_OK;
    const int S = std::min(std::min(max_s, pow2_floor(256 / tiles)), pow2_floor(nk / 8));
    if (S < 2) return XDET_OK;
    // A pointwise layer that two ranges of 128-wide tiles only just spread over the chip (ResNet-50 stage 3's opening 1x1 at
    // batch 8: 114 tiles, 32 steps -> 228 workgroups + a fold launch) runs as ONE range of 128 x 64 tiles instead: the same
    // 228 workgroups, no scratch traffic, no second launch
    if (S == 2 && bn == 128 && L->kh == 1 && tiles * 2 <= 256 && nk <= 32) {
      XDET_TRY(L->enable_ksplit(1, 1));
      L->ks_narrow = true;
      return XDET_OK;
    }
    // scratch: borrowed from the plan -- bound in finish_ksplit() once every layer's need is known
    XDET_TRY(L->enable_ksplit(1, 448 / S));
    L->ksplit = S;
    ks_layers.push_back({L, ks_on_aux_stream});
    return XDET_OK;
  }
  struct KsLayer { ConvLayer* L; bool aux; };
  std::vector<KsLayer> ks_layers;
  bool ks_on_aux_stream = false;       // builders set this around layers that run on the side stream (the RPN branch)
  float* ks_slab[2] = {nullptr, nullptr};
  int* ks_ticket[2] = {nullptr, nullptr};
  // one slab per stream (main / side), sized for the largest split-K layer on it (a slab per layer was
  // 0.5-0.8 GB per ResNet trunk)
  int finish_ksplit() {
    size_t need[2] = {0, 0};
    for (const KsLayer& k : ks_layers) need[k.aux] = std::max(need[k.aux], k.L->ks_scratch_bytes(k.L->ksplit, k.L->ks_tiles));
    for (int a = 0; a < 2; ++a)
      if (need[a] && !ks_slab[a]) {
        XDET_TRY(alloc_bytes(need[a], reinterpret_cast<void**>(&ks_slab[a]), false));
        XDET_TRY(alloc_bytes(4096 * sizeof(int), reinterpret_cast<void**>(&ks_ticket[a])));     // (zeroed; every fold puts its ticket back)
      }
    for (const KsLayer& k : ks_layers) { k.L->d_ks_partial = ks_slab[k.aux]; k.L->d_ks_ticket = ks_ticket[k.aux]; }
    return XDET_OK;
  }
  bool fuse_sepconv = true;      // option "sepconv" = "fused" | "split"
  bool subsample_projections = true;
  bool pool_writes_projection_input = true;   // option "pool_sub" = "on" | "off": blocks 2-3's pool pass also writes the next
                                              // projection's subsampled planes and stores relu(sum) (off: A/B runs, tests)
  bool patch_conv3x3 = true;     // option "conv3x3" = "patch" | "gemm"
  bool fuse_hpool = true;        // option "pool" = "split" | "whole": horizontal half of an entry-flow pool in the producer
  int pool_fuse_min_pixels = 100 * 100;   // ... for the 237 x 237 block (+0.8 %) and the 119 x 119 one (time-neutral, -0.9 GB)
  int add_conv(const std::string& name, int stage, const Buf& in_, ConvLayer* L, const Buf* res, int relu_in, Buf* out) {
    Buf in = in_;
    bool own_split = false;
    const int emit = L->precision == PREC_F32 ? 0 : emit_planes_next;
    const bool folded_bn = emit && !emit_bn_scale.empty();
    if (folded_bn) XDET_TRY(L->set_planes_bn(emit_bn_scale, emit_bn_shift));
    emit_planes_next = 0;
    const Buf* planes_into = emit_planes_into;
    emit_planes_into = nullptr;
    emit_bn_scale.clear();
    emit_bn_shift.clear();
    // split path: big stride-1 contractions take their A operand as f16 planes through the LDS DMA;
    // if the producer did not emit planes, one cheap element-wise pass makes them (ReLU folded in)
    // (a strided conv takes planes too if its producer wrote them: the DMA gathers any pixel per row; it is not worth
    //  a separate split pass over a tensor of which it reads a quarter)
    if (L->dma_capable() && (L->stride == 1 || (in.hi && !(in.planes_relu && !relu_in)))) {
      if (in.hi && in.planes_relu && !relu_in) in.hi = in.lo = nullptr;   // relu(x) planes are no use here
      XDET_REQUIRE(!in.no_f32 || (in.hi && !relu_in), "plan: this tensor exists as planes only");
      if (in.hi && relu_in && in.planes_relu) {
        relu_in = 0;                               // the producer already wrote relu(x) planes
      } else if (!in.hi || relu_in) {
        Buf sp;
        XDET_TRY(add_split(name + "/split_in", stage, in_, relu_in, &sp));
        in = sp;
        relu_in = 0;
        own_split = true;                          // these planes have this conv as their only reader
      }
    } else {
      XDET_REQUIRE(!in.no_f32, "plan: this tensor exists as planes only");
      in.hi = in.lo = nullptr;
      // this conv splits its f32 input in registers (conv_mfma_split.hip): no planes, no pre-scale -> the f32 tensor
      // itself must stay inside the f16 range (check_range)
      if (L->precision != PREC_F32) f32_split_inputs.push_back(in.p);
    }
    int Ho, Wo, a, b;
    L->out_shape(in.H, in.W, &Ho, &Wo, &a, &b);
    XDET_TRY(new_buf(Ho, Wo, L->cout, out));
    XDET_REQUIRE(in.ld == L->ld_in() && out->ld == L->ld_out(), "plan: conv channel strides do not match");
    if (in.hi) XDET_TRY(maybe_ksplit(L, in.H, in.W, Ho, Wo)); else ksplit_next = false;
    XDET_REQUIRE(!planes_into || emit == 3, "plan: a concatenated planes destination needs a planes-only output");
    if (emit) {
      if (planes_into) {
        XDET_REQUIRE(planes_into->hi && planes_into->pidx >= 0 && planes_into->H == Ho && planes_into->W == Wo &&
                         planes_into->ld % 32 == 0 && planes_into->ld > out->ld && !folded_bn,
                     "plan: the concatenated planes destination does not match this conv's output");
        out->hi = planes_into->hi;
        out->lo = planes_into->lo;
        out->pidx = planes_into->pidx;
        L->pl_c32 = planes_into->ld >> 5;
        pscales[out->pidx].op_index = (int)ops.size();     // complete (and measured) behind this op, its last producer
      } else {
        XDET_TRY(new_planes(out));
      }
      out->planes_relu = emit == 2;
      out->no_f32 = emit == 3;
      if (!planes_into) pscales[out->pidx].name = name + " (planes of the output)";
      pscales[out->pidx].apply.push_back([L](int e) { return L->set_out_exp(e); });
    }
    if (in.hi && in.pidx >= 0) pscales[in.pidx].apply.push_back([L](int e) { return L->set_in_exp(e); });
    if (in.hi && in.pidx >= 0 && pscales[in.pidx].x8_cand) {
      // the consumer end of an x8 plane: a pointwise f16x3 layer on the buffer-load form of the LDS-DMA kernels, no split-K
      const size_t a_bytes = ((size_t)cdiv((int64_t)max_batch * in.H * in.W, 16) * (size_t)(in.ld >> 5)) << 10;
      const size_t b_bytes = (size_t)(L->kp / 32) * L->cout_pad * 64;
      pscales[in.pidx].x8_ok = L->d_wt_x8_b && L->precision == PREC_F16X3 && L->ksplit < 1 && L->groups == 1 && relu_in == 0 &&
                               L->cin_p == L->kp && a_bytes < ((size_t)1 << 32) && b_bytes < ((size_t)1 << 32);
      pscales[in.pidx].x8_cand = false;              // (one consumer: a second reader of these planes would have to agree)
    } else if (in.hi && in.pidx >= 0 && pscales[in.pidx].x8_ok) {
      pscales[in.pidx].x8_ok = false;                // a second consumer: keep the f16 form
    }
    const int x8_pidx = in.hi ? in.pidx : -1;
    const bool planes_optional = planes_optional_next && emit != 0 && emit != 3;
    planes_optional_next = false;
    last_optional_idx = -1;
    if (planes_optional) {
      last_optional_idx = (int)planes_drop_ok.size();
      planes_drop_ok.push_back(0);
    }
    const int drop_idx = last_optional_idx;
    const Buf i = in, o = *out;
    const float* rp = res ? res->p : nullptr;
    const unsigned short* z = zeros;
    if (res) XDET_REQUIRE(res->H == Ho && res->W == Wo && res->ld == o.ld, "plan: residual shape mismatch");
    ops.push_back({name, stage, L->flops(in.H, in.W), [=](int N, hipStream_t s) {
                     // the planes copy: relu(out * bn_scale + bn_shift) * 2^-out_exp for a folded BN, else (relu?)(out) * 2^-out_exp
                     const bool aff = folded_bn || L->out_exp != 0;
                     int x8_exp = 0;
                     const int x8 = plane_x8(x8_pidx, &x8_exp) ? 1 : 0;
                     const bool drop = planes_optional && planes_drop_ok[drop_idx] && planes_dropped();
                     return L->forward(i.p, N, i.H, i.W, i.ld, o.no_f32 ? nullptr : o.p, o.ld, rp, relu_in, s, i.hi, i.lo,
                                       z, drop ? nullptr : o.hi, drop ? nullptr : o.lo, (o.planes_relu || folded_bn) ? 1 : 0, aff ? L->d_pl_scale : nullptr,
                                       aff ? L->d_pl_shift : nullptr, 0, x8, x8_exp);
                   }});
    if (own_split) release_planes(in);
    return XDET_OK;
  }
  // `planes_only`: the single consumer is a pointwise conv on the split path -> write f16 planes, no f32
  int add_dw(const std::string& name, int stage, const Buf& in, DepthwiseLayer* L, int relu_in, Buf* out,
             bool planes_only = false) {
    XDET_REQUIRE(!in.no_f32, "plan: this tensor exists as planes only");
    if (planes_only) {
      *out = Buf();
      out->H = in.H; out->W = in.W; out->C = in.C; out->ld = in.ld;
      XDET_TRY(new_planes(out));
      pscales[out->pidx].name = name;
      pscales[out->pidx].apply.push_back([L](int e) { return L->set_out_exp(e); });
      // (the depthwise TILE kernel writes the x8 form: images below 2 GiB)
      pscales[out->pidx].x8_cand = cross8 && (int64_t)in.H * in.W * in.ld * 4 < ((int64_t)1 << 31);
      const Buf i = in, o = *out;
      ops.push_back({name, stage, 0.0, [=](int N, hipStream_t s) {
                       int x8_exp = 0;
                       const int x8 = plane_x8(o.pidx, &x8_exp) ? 1 : 0;
                       return launch_depthwise3x3_split(i.p, L->d_w, o.hi, o.lo, N, i.H, i.W, i.C, i.ld, L->dil,
                                                        relu_in, s, x8, x8_exp);
                     }});
      return XDET_OK;
    }
    XDET_TRY(new_buf(in.H, in.W, in.C, out));
    const Buf i = in, o = *out;
    ops.push_back({name, stage, 0.0, [=](int N, hipStream_t s) {
                     return L->forward(i.p, N, i.H, i.W, i.ld, o.p, relu_in, s);
                   }});
    return XDET_OK;
  }
  int add_pool(const std::string& name, int stage, const Buf& in, const Buf* res, Buf* out) {
    XDET_REQUIRE(!in.no_f32 && !(res && res->no_f32), "plan: this tensor exists as planes only");
    int Ho, Wo, pt, pl;
    same_pad(in.H, 3, 2, 1, &pt, &Ho);
    same_pad(in.W, 3, 2, 1, &pl, &Wo);
    XDET_TRY(new_buf(Ho, Wo, in.C, out));
    if (res) XDET_REQUIRE(res->H == Ho && res->W == Wo && res->ld == out->ld, "plan: pool residual shape mismatch");
    const Buf i = in, o = *out;
    const float* rp = res ? res->p : nullptr;
    ops.push_back({name, stage, 0.0, [=](int N, hipStream_t s) {
                     return launch_maxpool3x3s2_add(i.p, rp, o.p, N, i.H, i.W, i.C, i.ld, Ho, Wo, pt, pl, s);
                   }});
    return XDET_OK;
  }
  ConvLayer* keep(ConvLayer* L) { layers.emplace_back(L); return L; }
  DepthwiseLayer* keep(DepthwiseLayer* L) { layers.emplace_back(L); return L; }

  // tf.layers.conv2d(use_bias=False) + BN (+ReLU)
  // pre_sc / pre_sh (1x1 stride-2 projections only): the input is a raw tensor and the conv reads
  // relu(in * pre_sc + pre_sh) -- the pre-activation BN of a ResNet v2 block, applied in the subsample pass
  int conv_bn(const std::string& name, const std::string& bn, float eps, int stage, const Buf& in, int k, int cout,
              int stride, int pad_mode, int relu_out, const Buf* res, int relu_in, Buf* out, int pad_expl = 0,
              const float* pre_sc = nullptr, const float* pre_sh = nullptr, const Buf* presub = nullptr) {
    const HostTensor* kt;
    XDET_TRY(need(name + "/kernel", &kt, {k, k, in.C, cout}));
    std::vector<float> sc, sh;
    const float *scp = nullptr, *shp = nullptr;
    if (!bn.empty()) {
      XDET_TRY(fold_bn(bn, cout, eps, nullptr, &sc, &sh));
      scp = sc.data(); shp = sh.data();
    }
    ConvLayer* L = keep(new ConvLayer());
    XDET_REQUIRE(!pre_sc || (k == 1 && stride == 2 && g_default_precision != PREC_F32 && in.C >= 32 && !in.no_f32 &&
                             in.ld % 32 == 0),
                 "plan: a pre-activation can only be folded into a 1x1 stride-2 projection on the split path");
    if (k == 1 && stride == 2 && (pad_mode == 1 || (pad_mode == 2 && pad_expl == 0)) &&
        g_default_precision != PREC_F32 && in.C >= 32 && in.ld % 32 == 0 && !relu_in && !in.no_f32 &&
        (subsample_projections || pre_sc)) {
      // 1x1 / stride 2 / SAME or unpadded (the residual projections): output (oy, ox) = input (2 oy, 2 ox) x W.  Subsample + split
      // in one small pass (a quarter of the input), then a plain stride-1 GEMM on the LDS-DMA kernel -- the strided
      // gather through registers ran at 60-120 TFLOP/s.  Same products in the same order: bit-identical.
      XDET_TRY(L->init(1, 1, in.C, cout, 1, 1, 1, 0, 0, kt->v.data(), scp, shp, relu_out));
      if (presub && presub->hi) {
        // the producer of `in` (the pool pass in front of this block) already wrote the raw subsampled planes
        XDET_REQUIRE(!pre_sc && presub->H == (in.H + 1) / 2 && presub->W == (in.W + 1) / 2 && presub->ld == in.ld,
                     "plan: pre-made subsampled planes do not match the projection's input");
        XDET_TRY(add_conv(name, stage, *presub, L, res, 0, out));
        release_planes(*presub);
        return XDET_OK;
      }
      Buf sub;
      sub.H = (in.H + 1) / 2; sub.W = (in.W + 1) / 2; sub.C = in.C; sub.ld = in.ld; sub.no_f32 = true;
      XDET_TRY(new_planes(&sub));
      pscales[sub.pidx].name = name + "/subsample_split";
      const Buf i = in, o = sub;
      ops.push_back({name + "/subsample_split", stage, 0.0, [=](int N, hipStream_t s) {
                       return launch_split_f32_subsample2(i.p, o.hi, o.lo, N, i.H, i.W, i.ld, s, pre_sc, pre_sh, pmul(o.pidx));
                     }});
      XDET_TRY(add_conv(name, stage, sub, L, res, 0, out));
      release_planes(sub);                         // read by this projection only
      return XDET_OK;
    }
    XDET_REQUIRE(!(presub && presub->hi), "plan: pre-made subsampled planes, but this conv is not a subsampled 1x1 / stride-2 projection");
    XDET_TRY(L->init(k, k, in.C, cout, stride, 1, pad_mode, pad_expl, pad_expl, kt->v.data(), scp, shp, relu_out));
    // 3x3 VALID over 32 channels (block1_conv2): the input tile is staged in LDS once and the nine taps are shifted
    // fragment reads of it (conv3x3_patch.hip) instead of nine DMA'd K steps; same products, same order
    if (patch_conv3x3 && L->dma_capable() && in.hi && !in.planes_relu && !relu_in && !res && emit_planes_next == 0 &&
        in.ld == 32 && conv3x3_patch_supported(k, k, in.C, L->cout_pad, stride, 1, pad_mode) && L->cout_pad == L->ld_out()) {
      XDET_TRY(new_buf(in.H - 2, in.W - 2, cout, out));
      if (in.pidx >= 0) pscales[in.pidx].apply.push_back([L](int e) { return L->set_in_exp(e); });
      const Buf i = in, o = *out;
      ops.push_back({name + " [LDS-staged tile]", stage, L->flops(in.H, in.W), [=](int N, hipStream_t s) {
                       return launch_conv3x3_patch(i.hi, i.lo, L->d_wt_hi_b, L->d_wt_lo_b, L->d_scale, L->d_shift, o.p, N, i.H,
                                                   i.W, o.ld, L->relu_out, s);
                     }});
      return XDET_OK;
    }
    return add_conv(name, stage, in, L, res, relu_in, out);
  }
  // (ReLU ->) separable_conv2d -> BN (-> +residual) (-> ReLU)   net/xception_body.py:220-234
  // pool_res != NULL: the block is followed by max_pooling2d(3, 2, 'same') + tf.add(pool_res) (entry flow,
  // net/xception_body.py:281-286); *out is then the pooled sum.  Where the fused kernel runs and the map is large enough
  // to be HBM-bound (pool_fuse_min_hw), its epilogue does the horizontal half of the pool and a light pass the vertical
  // half + the add: the full-resolution block output is written and read at half size.
  // next_sub != NULL (with pool_res, when the pool runs as the split vertical pass): the pass also writes the raw pooled sum's
  // subsampled planes into *next_sub (what the NEXT block's 1x1 / stride-2 projection reads) and stores relu(sum) as *out
  // (what its first separable conv reads): maxpool_v3s2_add_kernel<SUB>.  next_sub->hi stays NULL where that form is not taken.
  int sep_bn(const std::string& name, float eps, int stage, const Buf& in, int cout, int pre_relu, int dilation,
             int relu_out, const Buf* res, Buf* out, const Buf* pool_res = nullptr, Buf* next_sub = nullptr) {
    const HostTensor *dk, *pk;
    XDET_TRY(need(name + "/depthwise_kernel", &dk, {3, 3, in.C, 1}));
    XDET_TRY(need(name + "/pointwise_kernel", &pk, {1, 1, in.C, cout}));
    DepthwiseLayer* D = keep(new DepthwiseLayer());
    XDET_TRY(D->init(in.C, dilation, dk->v.data()));
    std::vector<float> sc, sh;
    XDET_TRY(fold_bn(name + "_bn", cout, eps, nullptr, &sc, &sh));
    ConvLayer* L = keep(new ConvLayer());
    XDET_TRY(L->init(1, 1, in.C, cout, 1, 1, 1, 0, 0, pk->v.data(), sc.data(), sh.data(), relu_out));
    // Entry-flow shapes (<= 256 input channels, 128 / 256 outputs, dilation 1, no residual, f32 consumers only):
    // ONE fused kernel, the depthwise result never leaves the CU (sepconv_fused.hip); bit-identical to the
    // two-kernel form below, which the wide 30x30 layers keep (their GEMM needs the big MFMA tiles).
    if (fuse_sepconv && L->dma_capable() && !res && emit_planes_next == 0 && !in.no_f32 &&
        sepconv_fused_supported(in.ld, L->cout_pad, dilation) && L->ld_out() <= L->cout_pad) {
      {
        // the depthwise result is split on the CU and never reaches HBM: its range is bounded from the block input,
        // |dw[c]| <= max|relu?(x)| * sum_t |w[t, c]|, and the pre-scale rides in the taps / the pointwise epilogue scale
        PlaneScale ps;
        ps.name = name + " (depthwise result inside the fused block)";
        ps.src = in.p;
        ps.src_per_image = in.per_image();
        ps.src_relu = pre_relu;
        float bound = 0.f;
        for (int c = 0; c < in.C; ++c) {
          float a = 0.f;
          for (int t = 0; t < 9; ++t) a += std::fabs(dk->v[(size_t)t * in.C + c]);
          bound = std::max(bound, a);
        }
        ps.bound = bound;
        ps.apply.push_back([D](int e) { return D->set_out_exp(e); });
        ps.apply.push_back([L](int e) { return L->set_in_exp(e); });
        ps.op_index = (int)ops.size();               // the fused op is pushed next: its input is intact right behind it
        pscales.push_back(ps);
      }
      if (pool_res && fuse_hpool && in.H * in.W >= pool_fuse_min_pixels) {
        int Ho, Wo, pt, pl;
        same_pad(in.H, 3, 2, 1, &pt, &Ho);
        same_pad(in.W, 3, 2, 1, &pl, &Wo);
        Buf hp;
        XDET_TRY(new_buf(in.H, Wo, cout, &hp));
        XDET_TRY(new_buf(Ho, Wo, cout, out));
        XDET_REQUIRE(pool_res->H == Ho && pool_res->W == Wo && pool_res->ld == out->ld && !pool_res->no_f32,
                     "plan: pool residual shape mismatch");
        const Buf i = in, h = hp, o = *out;
        const float* rp = pool_res->p;
        ops.push_back({name + "/fused_dw+pw+hpool", stage, L->flops(in.H, in.W), [=](int N, hipStream_t s) {
                         return launch_sepconv_fused(i.p, D->d_w, L->d_wt_hi_b, L->d_wt_lo_b, L->d_scale, L->d_shift, h.p, N,
                                                     i.H, i.W, i.ld, h.ld, L->cout_pad, pre_relu, L->relu_out, s, pl);
                       }});
        Buf sub;
        if (next_sub && out->ld % 32 == 0) {
          sub.H = (Ho + 1) / 2; sub.W = (Wo + 1) / 2; sub.C = out->C; sub.ld = out->ld; sub.no_f32 = true;
          XDET_TRY(new_planes(&sub));                // (measured right behind the pass below: op_index)
          pscales[sub.pidx].name = name + "/vpool_add (subsampled planes for the next projection)";
          *next_sub = sub;
        }
        const Buf sb = sub;
        ops.push_back({name + (sb.hi ? "/vpool_add+subsample_split+relu" : "/vpool_add"), stage, 0.0, [=](int N, hipStream_t s) {
                         return launch_maxpool_v3s2_add(h.p, rp, o.p, N, h.H, h.W, h.C, h.ld, Ho, pt, s, sb.hi, sb.lo,
                                                        sb.hi ? pmul(sb.pidx) : 1.f);
                       }});
        release_f32(hp);
        return XDET_OK;
      }
      Buf full;
      Buf* dst = pool_res ? &full : out;
      XDET_TRY(new_buf(in.H, in.W, cout, dst));
      const Buf i = in, o = *dst;
      ops.push_back({name + "/fused_dw+pw", stage, L->flops(in.H, in.W), [=](int N, hipStream_t s) {
                       return launch_sepconv_fused(i.p, D->d_w, L->d_wt_hi_b, L->d_wt_lo_b, L->d_scale, L->d_shift, o.p, N,
                                                   i.H, i.W, i.ld, o.ld, L->cout_pad, pre_relu, L->relu_out, s);
                     }});
      if (!pool_res) return XDET_OK;
      XDET_TRY(add_pool(name + "/pool_add", stage, full, pool_res, out));
      release_f32(full);
      return XDET_OK;
    }
    Buf t;
    XDET_TRY(add_dw(name + "/depthwise", stage, in, D, pre_relu, &t, /*planes_only=*/L->dma_capable()));
    if (!pool_res) {
      XDET_TRY(add_conv(name + "/pointwise", stage, t, L, res, 0, out));
      release_planes(t);                           // the depthwise result: read by its pointwise conv only
      release_f32(t);
      return XDET_OK;
    }
    Buf full;
    XDET_TRY(add_conv(name + "/pointwise", stage, t, L, res, 0, &full));
    release_planes(t);
    release_f32(t);
    XDET_TRY(add_pool(name + "/pool_add", stage, full, pool_res, out));
    release_f32(full);
    return XDET_OK;
  }
  int run_stage(int stage, int N, hipStream_t s) {
    for (size_t i = 0; i < ops.size(); ++i) {
      const Op& op = ops[i];
      if (op.stage != stage) continue;
      for (const PoisonRec& pr : poison)                 // (option "workspace" = "poison": tests only)
        if (pr.op == i) XDET_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(pr.at), 0x7fc00000, pr.words, s));
      if (profiling) {
        ProfRec r;
        r.op = (int)i;
        XDET_HIP(hipEventCreate(&r.a));
        XDET_HIP(hipEventCreate(&r.b));
        XDET_HIP(hipEventRecord(r.a, s));
        XDET_TRY(op.run(N, s));
        XDET_HIP(hipEventRecord(r.b, s));
        prof.push_back(r);
      } else {
        XDET_TRY(op.run(N, s));
      }
      if (after_op) XDET_TRY(after_op((int)i, s));
      static const bool trace = getenv("XDET_TRACE_OPS") != nullptr;   // diagnosis: name the op a device fault belongs to
      if (trace) {
        // (a synchronise on a capturing stream would invalidate the capture: a graph forward is only named, not waited for)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }
        fprintf(stderr, "xdet op %zu: %s ...", i, op.name.c_str());
        fflush(stderr);
        if (cap == hipStreamCaptureStatusNone) {
          const hipError_t e = hipStreamSynchronize(s);
          fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
        } else {
          fprintf(stderr, " (captured)\n");
        }
      }
    }
    return XDET_OK;
  }
  // drain the recorded (start, stop) event pairs into per-op totals; call after a stream sync
  int profile_read(int max_ops, int* n_ops, double* ms, int* launches, double* flops) {
    *n_ops = (int)std::min<size_t>(ops.size(), (size_t)max_ops);
    for (int i = 0; i < *n_ops; ++i) { ms[i] = 0; launches[i] = 0; flops[i] = ops[i].flops; }
    for (ProfRec& r : prof) {
      float t = 0.f;
      XDET_HIP(hipEventSynchronize(r.b));
      XDET_HIP(hipEventElapsedTime(&t, r.a, r.b));
      if (r.op < *n_ops) { ms[r.op] += t; launches[r.op] += 1; }
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    prof.clear();
    return XDET_OK;
  }
};

// ST_EXIT: the exit flow (conv2d_4, blocks 13-14, net/xception_body.py:340-376) -- part of the backbone, its own stage so that
// the RPN branch, which only needs mid_outputs (:339), can fork in front of it
enum { ST_BODY = 0, ST_RPN = 1, ST_LSEP = 2, ST_HEAD = 3, ST_EXIT = 4 };

static inline double nsplit_of(const ConvLayer* L) { return L->precision == PREC_F16X3 ? 3.0 : 1.0; }

typedef std::array<uintptr_t, 6> GraphKey;

struct LightHeadNet : Plan {
  xdet_lighthead_config cfg;
  bool built = false;
  Buf in4, mid_x, out, rpn_out, feat, pooled, fc, cls_reg;
  float *objectness = nullptr, *rpn_boxes = nullptr, *proposals = nullptr, *head_boxes = nullptr;
  float* class_probs = nullptr;   // [B][num_classes][R] softmax of the head's logits, class-major (head_decode_probs_kernel)
  float *anc_yx = nullptr, *anc_hw = nullptr;
  float* mid_relu = nullptr;   // materialised ReLU(x) ("mid_outputs", xception_body.py:339) for API users
  void* prop_ws_mem = nullptr;
  ProposalWorkspace prop_ws;
  int* def_shapes = nullptr;
  float* def_bbox = nullptr;
  int fmap = 0, n_anchor = 0;
  int large_sep_mode = 0;               // 0 = auto, 1 = direct (15,1)/(1,15) convs, 2 = spectral (DFT-domain GEMMs)
  bool large_sep_spectral = false;      // decided at build
  bool rpn_side_stream = true;          // option "rpn_stream" = "side" | "main"
  bool check_range = false;             // option "check_range" = "off" | "on": validate every activation against the f16 range
  bool latency_ksplit = true;           // option "ksplit" = "on" | "off" | "all": fixed split-K for the narrow head GEMM (on),
  bool rpn_ksplit = false;              // ... and the RPN conv as well (all)
  std::vector<std::function<int(int, hipStream_t)>> extra_range_checks;   // tensors that are not plain [N][pixels][ld] (DFT bins)
  bool stem_direct = false;             // block1_conv1 as the dedicated NCHW -> planes kernel
  const float* cur_images = nullptr;
  hipStream_t aux = nullptr;            // side stream of the RPN/proposal branch
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  static constexpr size_t kMaxGraphs = 8;
  std::map<GraphKey, hipGraphExec_t> graphs;   // keyed by (N, images, image_shapes, bbox_img, det_scores, det_boxes)
  std::vector<GraphKey> graph_order;           // capture order, for eviction

  ~LightHeadNet() {
    for (auto& g : graphs) (void)hipGraphExecDestroy(g.second);
    // the side stream and its fork / join events (created on the first forward): a net that leaves them behind leaks a
    // hardware queue per instance -- a long test session (~150 nets) ran the runtime out of them and died inside a capture
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (aux) (void)hipStreamDestroy(aux);
  }

  int build_body() {
    const float eps = 1e-4f;   // net/xception_body.py:20
    const int S = cfg.image_size;
    in4.H = S; in4.W = S; in4.C = 3;
    XDET_TRY(new_buf(S, S, 3, &in4));
    Buf x, r, t;
    if (g_default_precision != PREC_F32) {
      // stem conv straight from the NCHW input to the planes block1_conv2 reads (elementwise.hip): K = 27 fits no
      // matrix-core shape, a VALU kernel with scalar-cache weights is HBM-bound instead of gather-bound
      const HostTensor* kt;
      XDET_TRY(need("block1_conv1/kernel", &kt, {3, 3, 3, 32}));
      std::vector<float> sc, sh;
      XDET_TRY(fold_bn("block1_conv1_bn", 32, eps, nullptr, &sc, &sh));
      float *d_w, *d_sc, *d_sh;
      XDET_TRY(alloc_bytes(kt->v.size() * 4, reinterpret_cast<void**>(&d_w), false));
      XDET_TRY(alloc_bytes(32 * 4, reinterpret_cast<void**>(&d_sc), false));
      XDET_TRY(alloc_bytes(32 * 4, reinterpret_cast<void**>(&d_sh), false));
      XDET_HIP(hipMemcpy(d_w, kt->v.data(), kt->v.size() * 4, hipMemcpyHostToDevice));
      XDET_HIP(hipMemcpy(d_sc, sc.data(), 32 * 4, hipMemcpyHostToDevice));
      XDET_HIP(hipMemcpy(d_sh, sh.data(), 32 * 4, hipMemcpyHostToDevice));
      x = Buf();
      x.H = x.W = (S - 3) / 2 + 1; x.C = 32; x.ld = 32; x.no_f32 = true;
      XDET_TRY(new_planes(&x));
      pscales[x.pidx].name = "block1_conv1 [stem]";
      pscales[x.pidx].apply.push_back([=](int e) {     // relu(x*sc + sh) * 2^-e = relu(x*(sc 2^-e) + sh 2^-e), exactly
        std::vector<float> a(sc), b(sh);
        for (float& v : a) v = ldexpf(v, -e);
        for (float& v : b) v = ldexpf(v, -e);
        XDET_HIP(hipMemcpy(d_sc, a.data(), 32 * 4, hipMemcpyHostToDevice));
        XDET_HIP(hipMemcpy(d_sh, b.data(), 32 * 4, hipMemcpyHostToDevice));
        return (int)XDET_OK;
      });
      const Buf o = x;
      stem_direct = true;
      ops.push_back({"block1_conv1 [stem, NCHW in]", ST_BODY, 2.0 * o.H * o.W * 27.0 * 32.0, [=](int N, hipStream_t st) {
                       return launch_stem_conv3x3s2(cur_images, d_w, d_sc, d_sh, o.hi, o.lo, N, S, st);
                     }});
      ops.back().mfma_flops = 0.0;      // VALU kernel
    } else {
      emit_planes_next = 3;   // feeds block1_conv2 (LDS-DMA path) only
      XDET_TRY(conv_bn("block1_conv1", "block1_conv1_bn", eps, ST_BODY, in4, 3, 32, 2, 0, 1, nullptr, 0, &x));
    }
    XDET_TRY(conv_bn("block1_conv2", "block1_conv2_bn", eps, ST_BODY, x, 3, 64, 1, 0, 1, nullptr, 0, &t));
    x = t;
    struct Blk { const char* res; const char* bn; const char* s1; const char* s2; int c; int first_relu; };
    const Blk blks[3] = {{"conv2d_1", "batch_normalization_1", "block2_sepconv1", "block2_sepconv2", 128, 0},
                         {"conv2d_2", "batch_normalization_2", "block3_sepconv1", "block3_sepconv2", 256, 1},
                         {"conv2d_3", "batch_normalization_3", "block4_sepconv1", "block4_sepconv2", 728, 1}};
    Buf presub;                                    // raw subsampled planes of x, written by the pool pass that produced x
    bool x_is_relu = false;                        // ... which then stored x as relu(x)
    for (int bi = 0; bi < 3; ++bi) {
      const Blk& b = blks[bi];
      XDET_TRY(conv_bn(b.res, b.bn, eps, ST_BODY, x, 1, b.c, 2, 1, 0, nullptr, 0, &r, 0, nullptr, nullptr,
                       presub.hi ? &presub : nullptr));
      Buf a, p;
      // relu -> sepconv2 (net/xception_body.py:271-277) is the ONLY consumer of sepconv1's BN output: the ReLU is taken
      // in sepconv1's epilogue (one v_max per element on its way out) instead of on sepconv2's 3 x 3 window reads (72 of
      // the stencil's 192 VALU instructions per chunk when the block runs as the fused kernel) -- the same values
      XDET_TRY(sep_bn(b.s1, eps, ST_BODY, x, b.c, x_is_relu ? 0 : b.first_relu, 1, /*relu_out=*/1, nullptr, &a));
      // The pooled sum of blocks 2 and 3 is read by exactly two consumers, the next block's projection (raw, every second
      // pixel) and its sepconv1 (through a ReLU): where the pool runs as the split vertical pass it writes both forms.
      // (Block 4's sum is the middle flow's residual stream: it stays raw.)
      Buf nsub;
      const bool two_readers = bi < 2 && pool_writes_projection_input && subsample_projections &&
                               g_default_precision != PREC_F32 && b.c % 32 == 0;
      XDET_TRY(sep_bn(b.s2, eps, ST_BODY, a, b.c, /*pre_relu=*/0, 1, 0, nullptr, &p, &r, two_readers ? &nsub : nullptr));
      presub = nsub;
      x_is_relu = nsub.hi != nullptr;
      release_f32(x);                              // the block input: read by the projection and sepconv1
      release_f32(a);
      release_f32(r);
      x = p;
    }
    for (int blk = 5; blk <= 12; ++blk) {
      const Buf res = x;
      Buf a, b2, c3;
      const std::string pre = "block" + std::to_string(blk);
      XDET_TRY(sep_bn(pre + "_sepconv1", eps, ST_BODY, x, 728, 1, 1, 0, nullptr, &a));
      XDET_TRY(sep_bn(pre + "_sepconv2", eps, ST_BODY, a, 728, 1, 1, 0, nullptr, &b2));
      if (blk == 12) emit_planes_next = 2;   // mid_outputs = ReLU(this) feeds the RPN 3x3 conv
      XDET_TRY(sep_bn(pre + "_sepconv3", eps, ST_BODY, b2, 728, 1, 1, 0, &res, &c3));
      release_f32(a);                              // a block's three tensors die with it: the middle flow lives in four
      release_f32(b2);                             // f32 blocks and one planes pair instead of 24 + 24
      release_f32(res);
      x = c3;
    }
    mid_x = x;   // mid_outputs = ReLU(mid_x); consumers apply the ReLU on load
    XDET_TRY(conv_bn("conv2d_4", "batch_normalization_4", eps, ST_EXIT, x, 1, 1024, 1, 1, 0, nullptr, 0, &r));
    Buf a, b2, c3, d4;
    XDET_TRY(sep_bn("block13_sepconv1", eps, ST_EXIT, x, 728, 1, 1, 0, nullptr, &a));
    XDET_TRY(sep_bn("block13_sepconv2", eps, ST_EXIT, a, 1024, 1, 1, 0, &r, &b2));
    XDET_TRY(sep_bn("block14_sepconv1", eps, ST_EXIT, b2, 1536, 0, 2, 1, nullptr, &c3));   // :354-364
    // The large-separable convs run either as direct implicit GEMMs over split planes or in the DFT domain
    // (spectral.hip: ~5x fewer MFMA FLOPs; one GEMM per frequency bin with M = N*fmap rows).  The spectral form
    // wins at every batch size -- at one image the direct (15,1) conv is a 900-row GEMM with K = 30,720 on 32
    // workgroups (0.9 ms), the 22 bins are 176 workgroups with K = 4,096 -- so `auto` takes it whenever the
    // feature-map size has a transform instantiated.  Decided per NET, never per call: an image's result
    // does not depend on the batch it arrives in.
    large_sep_spectral = g_default_precision != PREC_F32 && spectral_supported(c3.H) && c3.H == c3.W && large_sep_mode != 1;
    if (large_sep_mode == 2) XDET_REQUIRE(large_sep_spectral, "large_sep=spectral needs a split-precision mode and a 16/30/50 feature map");
    emit_planes_next = large_sep_spectral ? 0 : 1;   // the direct (15,1) conv takes planes; the DFT pass reads f32
    XDET_TRY(sep_bn("block14_sepconv2", eps, ST_EXIT, c3, 2048, 0, 2, 1, nullptr, &d4));   // :366-376
    release_f32(a);
    release_f32(r);
    release_f32(b2);
    release_f32(c3);
    out = d4;
    fmap = out.H;
    return XDET_OK;
  }

  int build_rpn() {
    struct PoolGuard { int& p; int old; ~PoolGuard() { p = old; } } pool_guard{ws_pool, ws_pool};
    ws_pool = rpn_side_stream ? 1 : 0;      // the branch runs beside the exit flow: no workspace block shared with the main stream
    const int A = cfg.num_anchors;
    const HostTensor *k0, *b0, *k1, *b1, *k2, *b2;
    XDET_TRY(need("rpn_head/conv2d/kernel", &k0, {3, 3, 728, 512}));
    XDET_TRY(need("rpn_head/conv2d/bias", &b0, {512}));
    XDET_TRY(need("rpn_head/conv2d_1/kernel", &k1, {1, 1, 512, 2 * A}));
    XDET_TRY(need("rpn_head/conv2d_1/bias", &b1, {2 * A}));
    XDET_TRY(need("rpn_head/conv2d_2/kernel", &k2, {1, 1, 512, 4 * A}));
    XDET_TRY(need("rpn_head/conv2d_2/bias", &b2, {4 * A}));
    ConvLayer* L0 = keep(new ConvLayer());
    XDET_TRY(L0->init(3, 3, 728, 512, 1, 1, 1, 0, 0, k0->v.data(), nullptr, b0->v.data(), 1));
    Buf hid;
    emit_planes_next = 3;   // only the fused 1x1 heads read it
    // (a single image is 8 x 4 tiles against 207 K steps.  Fixed split-K -- option "ksplit" = "all" -- takes it from 116 to
    //  37 us, but costs the 256 x 256 tile at bench-size batches: 1.92 -> 2.36 ms per 128 images, -0.9 % end to end.
    //  With the fork in front of the exit flow the conv is off the critical path of a single image anyway.)
    ksplit_next = latency_ksplit && rpn_ksplit;
    ks_on_aux_stream = rpn_side_stream;     // the RPN branch runs beside the exit flow: its own scratch slab
    XDET_TRY(add_conv("rpn_head/conv2d", ST_RPN, mid_x, L0, nullptr, /*relu_in=*/1, &hid));
    ks_on_aux_stream = false;
    // cls (2A) and box (4A) 1x1 heads share their input: one GEMM over the concatenated filters
    const int co = 6 * A;
    std::vector<float> kc((size_t)512 * co), bc(co);
    for (int ci = 0; ci < 512; ++ci) {
      for (int j = 0; j < 2 * A; ++j) kc[(size_t)ci * co + j] = k1->v[(size_t)ci * 2 * A + j];
      for (int j = 0; j < 4 * A; ++j) kc[(size_t)ci * co + 2 * A + j] = k2->v[(size_t)ci * 4 * A + j];
    }
    for (int j = 0; j < 2 * A; ++j) bc[j] = b1->v[j];
    for (int j = 0; j < 4 * A; ++j) bc[2 * A + j] = b2->v[j];
    ConvLayer* L1 = keep(new ConvLayer());
    XDET_TRY(L1->init(1, 1, 512, co, 1, 1, 1, 0, 0, kc.data(), nullptr, bc.data(), 0));
    XDET_TRY(add_conv("rpn_head/conv2d_1+2", ST_RPN, hid, L1, nullptr, 0, &rpn_out));
    return XDET_OK;
  }

  int build_large_sep() {
    const int mid = 256, co = cfg.bank * cfg.grid * cfg.grid;
    const HostTensor *a0, *a0b, *a1, *a1b, *c0, *c0b, *c1, *c1b;
    XDET_TRY(need("large_sep_feature/Branch_0/conv2d/kernel", &a0, {15, 1, 2048, mid}));
    XDET_TRY(need("large_sep_feature/Branch_0/conv2d/bias", &a0b, {mid}));
    XDET_TRY(need("large_sep_feature/Branch_1/conv2d/kernel", &a1, {15, 1, 2048, mid}));
    XDET_TRY(need("large_sep_feature/Branch_1/conv2d/bias", &a1b, {mid}));
    XDET_TRY(need("large_sep_feature/Branch_0/conv2d_1/kernel", &c0, {1, 15, mid, co}));
    XDET_TRY(need("large_sep_feature/Branch_0/conv2d_1/bias", &c0b, {co}));
    XDET_TRY(need("large_sep_feature/Branch_1/conv2d_1/kernel", &c1, {1, 15, mid, co}));
    XDET_TRY(need("large_sep_feature/Branch_1/conv2d_1/bias", &c1b, {co}));
    // both (15,1) convs read the same input: one conv with the filters concatenated (2*mid outputs)
    std::vector<float> ka((size_t)15 * 2048 * 2 * mid), ba(2 * mid);
    for (size_t tc = 0; tc < (size_t)15 * 2048; ++tc) {
      memcpy(&ka[tc * 2 * mid], &a0->v[tc * mid], mid * sizeof(float));
      memcpy(&ka[tc * 2 * mid + mid], &a1->v[tc * mid], mid * sizeof(float));
    }
    for (int j = 0; j < mid; ++j) { ba[j] = a0b->v[j]; ba[mid + j] = a1b->v[j]; }
    ConvLayer* LA = keep(new ConvLayer());
    XDET_TRY(LA->init(15, 1, 2048, 2 * mid, 1, 1, 1, 0, 0, ka.data(), nullptr, ba.data(), 0));
    Buf t;
    emit_planes_next = 3;   // only the (1,15) conv reads it
    XDET_TRY(add_conv("large_sep_feature/Branch_0+1/conv2d", ST_LSEP, out, LA, nullptr, 0, &t));
    // branch_0b + branch_1b = one (1,15) conv over the 2*mid stacked channels; then BN(1e-5)+ReLU
    std::vector<float> kb((size_t)15 * 2 * mid * co);
    for (int tap = 0; tap < 15; ++tap)
      for (int ci = 0; ci < mid; ++ci) {
        memcpy(&kb[((size_t)tap * 2 * mid + ci) * co], &c0->v[((size_t)tap * mid + ci) * co], co * sizeof(float));
        memcpy(&kb[((size_t)tap * 2 * mid + mid + ci) * co], &c1->v[((size_t)tap * mid + ci) * co], co * sizeof(float));
      }
    std::vector<float> bsum(co), sc, sh;
    for (int j = 0; j < co; ++j) bsum[j] = c0b->v[j] + c1b->v[j];
    XDET_TRY(fold_bn("large_sep_feature/batch_normalization", co, 1e-5f, bsum.data(), &sc, &sh));
    ConvLayer* LB = keep(new ConvLayer());
    XDET_TRY(LB->init(1, 15, 2 * mid, co, 1, 1, 1, 0, 0, kb.data(), sc.data(), sh.data(), 1));
    XDET_TRY(add_conv("large_sep_feature/Branch_0+1/conv2d_1", ST_LSEP, t, LB, nullptr, 0, &feat));
    return XDET_OK;
  }

  // net/xception_body.py:450-475 in the DFT domain of the convolved axis (spectral.hip): per frequency bin one
  // real GEMM [N*F, 2*Cin] x [2*Cin, 2*Cout] on the split-precision MFMA kernel (grouped launch), a forward
  // DFT pass in front and an inverse pass (+bias / +BN+ReLU) behind each of the two convolutions
  int build_large_sep_spectral() {
    const int mid = 256, co = cfg.bank * cfg.grid * cfg.grid, F = out.H, NB = spectral_points(F) / 2;
    const int cin = out.C, cin_ld = out.ld, mid2 = 2 * mid, co_ld = round_up(co, 32);
    const HostTensor *a0, *a0b, *a1, *a1b, *c0, *c0b, *c1, *c1b;
    XDET_TRY(need("large_sep_feature/Branch_0/conv2d/kernel", &a0, {15, 1, cin, mid}));
    XDET_TRY(need("large_sep_feature/Branch_0/conv2d/bias", &a0b, {mid}));
    XDET_TRY(need("large_sep_feature/Branch_1/conv2d/kernel", &a1, {15, 1, cin, mid}));
    XDET_TRY(need("large_sep_feature/Branch_1/conv2d/bias", &a1b, {mid}));
    XDET_TRY(need("large_sep_feature/Branch_0/conv2d_1/kernel", &c0, {1, 15, mid, co}));
    XDET_TRY(need("large_sep_feature/Branch_0/conv2d_1/bias", &c0b, {co}));
    XDET_TRY(need("large_sep_feature/Branch_1/conv2d_1/kernel", &c1, {1, 15, mid, co}));
    XDET_TRY(need("large_sep_feature/Branch_1/conv2d_1/bias", &c1b, {co}));
    // the same branch fusion as the direct form: (15,1) with 2*mid outputs, (1,15) over the stacked channels
    std::vector<float> ka((size_t)15 * cin * mid2), kb((size_t)15 * mid2 * co), ba(mid2), ones(std::max(mid2, co_ld), 1.f);
    for (size_t tc = 0; tc < (size_t)15 * cin; ++tc) {
      memcpy(&ka[tc * mid2], &a0->v[tc * mid], mid * sizeof(float));
      memcpy(&ka[tc * mid2 + mid], &a1->v[tc * mid], mid * sizeof(float));
    }
    for (int j = 0; j < mid; ++j) { ba[j] = a0b->v[j]; ba[mid + j] = a1b->v[j]; }
    for (int tap = 0; tap < 15; ++tap)
      for (int ci = 0; ci < mid; ++ci) {
        memcpy(&kb[((size_t)tap * mid2 + ci) * co], &c0->v[((size_t)tap * mid + ci) * co], co * sizeof(float));
        memcpy(&kb[((size_t)tap * mid2 + mid + ci) * co], &c1->v[((size_t)tap * mid + ci) * co], co * sizeof(float));
      }
    std::vector<float> bsum(co), sc, sh;
    for (int j = 0; j < co; ++j) bsum[j] = c0b->v[j] + c1b->v[j];
    XDET_TRY(fold_bn("large_sep_feature/batch_normalization", co, 1e-5f, bsum.data(), &sc, &sh));
    sc.resize(co_ld, 0.f);
    sh.resize(co_ld, 0.f);
    ConvLayer* LA = keep(new ConvLayer());
    ConvLayer* LB = keep(new ConvLayer());
    {
      std::vector<float> wa;
      spectral_weights(ka.data(), 15, cin, mid2, cin_ld, mid2, F, &wa);
      XDET_TRY(LA->init(1, 1, 2 * cin_ld, 2 * mid2, 1, 1, 0, 0, 0, wa.data(), nullptr, nullptr, 0, NB));
    }
    {
      std::vector<float> wb;
      spectral_weights(kb.data(), 15, mid2, co, mid2, co_ld, F, &wb);
      XDET_TRY(LB->init(1, 1, 2 * mid2, 2 * co_ld, 1, 1, 0, 0, 0, wb.data(), nullptr, nullptr, 0, NB));
    }
    std::vector<float> tf, ti;
    spectral_tables(F, &tf, &ti);
    float *d_tf, *d_tf_b, *d_ti, *d_ones, *d_ba, *d_sc, *d_sh;
    auto up = [&](const std::vector<float>& h, float** d) {
      XDET_TRY(alloc_bytes(h.size() * 4, reinterpret_cast<void**>(d), false));
      XDET_HIP(hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
      return (int)XDET_OK;
    };
    XDET_TRY(up(tf, &d_tf)); XDET_TRY(up(tf, &d_tf_b)); XDET_TRY(up(ti, &d_ti)); XDET_TRY(up(ones, &d_ones)); XDET_TRY(up(ba, &d_ba));
    XDET_TRY(up(sc, &d_sc)); XDET_TRY(up(sh, &d_sh));
    // workspace, sized for max_batch: rows of every bin are padded to a whole number of 256-row GEMM tiles
    const size_t mp_max = (size_t)round_up(max_batch * F, 256), rows = (size_t)NB * mp_max;
    unsigned short *xa_hi, *xa_lo, *xb_hi, *xb_lo;
    float *y1, *tmid, *y2;
    XDET_TRY(alloc_bytes(rows * 2 * cin_ld * 2 + 512, reinterpret_cast<void**>(&xa_hi)));
    XDET_TRY(alloc_bytes(rows * 2 * cin_ld * 2 + 512, reinterpret_cast<void**>(&xa_lo)));
    XDET_TRY(alloc_bytes(rows * 2 * mid2 * 2 + 512, reinterpret_cast<void**>(&xb_hi)));
    XDET_TRY(alloc_bytes(rows * 2 * mid2 * 2 + 512, reinterpret_cast<void**>(&xb_lo)));
    XDET_TRY(alloc_bytes(rows * 2 * mid2 * 4 + 512, reinterpret_cast<void**>(&y1)));
    XDET_TRY(alloc_bytes((size_t)max_batch * F * F * mid2 * 4 + 512, reinterpret_cast<void**>(&tmid)));
    XDET_TRY(alloc_bytes(rows * 2 * co_ld * 4 + 512, reinterpret_cast<void**>(&y2)));
    XDET_TRY(new_buf(F, F, co, &feat));
    const Buf o = out, ft = feat;
    // check_range covers the DFT-domain tensors too.  A bin sums up to F samples (the DC bin all of them), so the f16
    // range of the planes is reached at activations of ~65504 / F, and tmid is an un-normalised 15-tap conv output:
    // this is where a checkpoint with large activations would overflow first.
    f32_bufs.emplace_back(tmid, (size_t)F * F * mid2);
    {
      // (prop_ws.bad is carved later, in build(): the lambda reads it through `this` at call time)
      auto mpad_rc = [F](int N) { return N * F <= 128 ? 128 : round_up(N * F, 256); };
      extra_range_checks.push_back([=](int N, hipStream_t s) {
        const int mp = mpad_rc(N);
        XDET_TRY(launch_range_check_planes(xa_hi, N, F, 2 * cin_ld, prop_ws.bad, s, NB, mp));
        XDET_TRY(launch_range_check_planes(xb_hi, N, F, 2 * mid2, prop_ws.bad, s, NB, mp));
        // the per-bin products are f32 and never split: NaN / inf only
        XDET_TRY(launch_range_check(y1, N, (size_t)F * 2 * mid2, 3.4028235e38f, prop_ws.bad, s, NB, (size_t)mp * 2 * mid2));
        return launch_range_check(y2, N, (size_t)F * 2 * co_ld, 3.4028235e38f, prop_ws.bad, s, NB, (size_t)mp * 2 * co_ld);
      });
    }
    const std::string pre = "large_sep_feature/Branch_0+1/";
    {
      // activation pre-scale of the DFT-domain operands: the forward transform is linear, so 2^-e rides in its table
      // (one copy per transform) and 2^e in the per-bin GEMM's epilogue scale.  These are the tensors with the least
      // headroom: a bin sums up to F samples, and the second one transforms an un-normalised 15-tap conv output.
      auto mpad_ps = [F](int N) { return N * F <= 128 ? 128 : round_up(N * F, 256); };
      const std::vector<float> tf_host = tf;
      PlaneScale pa, pb;
      pa.name = pre + "conv2d/dft_y (DFT-domain planes)";
      pa.hi = xa_hi;
      pa.halves = [=](int N) { return (int64_t)NB * mpad_ps(N) * 2 * cin_ld; };
      pa.apply.push_back([=](int e) {
        std::vector<float> t(tf_host);
        for (float& v : t) v = ldexpf(v, -e);
        XDET_HIP(hipMemcpy(d_tf, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        return LA->set_in_exp(e);
      });
      pb.name = pre + "conv2d_1/dft_x (DFT-domain planes)";
      pb.hi = xb_hi;
      pb.halves = [=](int N) { return (int64_t)NB * mpad_ps(N) * 2 * mid2; };
      pb.apply.push_back([=](int e) {
        std::vector<float> t(tf_host);
        for (float& v : t) v = ldexpf(v, -e);
        XDET_HIP(hipMemcpy(d_tf_b, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        return LB->set_in_exp(e);
      });
      // rows [N*F, m_pad) of a bin are padding that only an earlier, larger batch ever wrote: a measurement over
      // whole bins must not see that batch's (possibly overflowed) values
      const size_t bytes_a = rows * 2 * cin_ld * 2, bytes_b = rows * 2 * mid2 * 2;
      pa.clear = [=](hipStream_t st) { XDET_HIP(hipMemsetAsync(xa_hi, 0, bytes_a, st)); return (int)XDET_OK; };
      pb.clear = [=](hipStream_t st) { XDET_HIP(hipMemsetAsync(xb_hi, 0, bytes_b, st)); return (int)XDET_OK; };
      pscales.push_back(pa);
      pscales.push_back(pb);
    }
    const double fl_a = 2.0 * F * F * (double)cin * mid2 * 15, fl_b = 2.0 * F * F * (double)mid2 * co * 15;
    // rows per bin: whole 256-row GEMM tiles; a single image or two (N*F <= 128) get the 128-row tile instead
    auto mpad = [F](int N) { return N * F <= 128 ? 128 : round_up(N * F, 256); };
    // flops < 0 marks an auxiliary pass of a contraction: its time counts with the conv kernels, it has no FLOPs of its own
    ops.push_back({pre + "conv2d/dft_y", ST_LSEP, -1.0, [=](int N, hipStream_t s) {
                     return launch_dft_fwd(o.p, F, o.ld, 0, N, mpad(N), d_tf, xa_hi, xa_lo, s);
                   }});
    ops.push_back({pre + "conv2d [spectral]", ST_LSEP, fl_a, [=](int N, hipStream_t s) {
                     return LA->forward(nullptr, 1, 1, NB * mpad(N), 2 * cin_ld, y1, 2 * mid2, nullptr, 0, s, xa_hi, xa_lo,
                                        LA->d_zeros, nullptr, nullptr, 0, nullptr, nullptr, mpad(N), 0, 0, N * F);
                   }});
    ops.back().mfma_flops = 2.0 * nsplit_of(LA) * NB * (double)F * (2.0 * cin_ld) * (2.0 * mid2);
    ops.push_back({pre + "conv2d/idft_y+bias", ST_LSEP, -1.0, [=](int N, hipStream_t s) {
                     return launch_dft_inv(y1, F, 2 * mid2, mid2, N, mpad(N), d_ti, d_ones, d_ba, 0, tmid, mid2, 0, s);
                   }});
    ops.push_back({pre + "conv2d_1/dft_x", ST_LSEP, -1.0, [=](int N, hipStream_t s) {
                     return launch_dft_fwd(tmid, F, mid2, 1, N, mpad(N), d_tf_b, xb_hi, xb_lo, s);
                   }});
    ops.push_back({pre + "conv2d_1 [spectral]", ST_LSEP, fl_b, [=](int N, hipStream_t s) {
                     return LB->forward(nullptr, 1, 1, NB * mpad(N), 2 * mid2, y2, 2 * co_ld, nullptr, 0, s, xb_hi, xb_lo,
                                        LB->d_zeros, nullptr, nullptr, 0, nullptr, nullptr, mpad(N), 0, 0, N * F);
                   }});
    ops.back().mfma_flops = 2.0 * nsplit_of(LB) * NB * (double)F * (2.0 * mid2) * (2.0 * co_ld);
    ops.push_back({pre + "conv2d_1/idft_x+bn+relu", ST_LSEP, -1.0, [=](int N, hipStream_t s) {
                     return launch_dft_inv(y2, F, 2 * co_ld, co_ld, N, mpad(N), d_ti, d_sc, d_sh, 1, ft.p, ft.ld, 1, s);
                   }});
    return XDET_OK;
  }

  int build_head() {
    const int R = cfg.rpn_post_nms_top_n, C = cfg.bank * cfg.grid * cfg.grid, nc = cfg.num_classes;
    const HostTensor *k0, *b0, *k1, *b1, *k2, *b2;
    XDET_TRY(need("final_head/subnet_fc/kernel", &k0, {C, 2048}));
    XDET_TRY(need("final_head/subnet_fc/bias", &b0, {2048}));
    XDET_TRY(need("final_head/fc_cls/kernel", &k1, {2048, nc}));
    XDET_TRY(need("final_head/fc_cls/bias", &b1, {nc}));
    XDET_TRY(need("final_head/fc_loc/kernel", &k2, {2048, 4}));
    XDET_TRY(need("final_head/fc_loc/bias", &b2, {4}));
    // ROI rows are the GEMM M dimension: treat [N*R, C] as an NHWC tensor with H = R, W = 1
    XDET_TRY(new_buf(R, 1, C, &pooled));
    ConvLayer* L0 = keep(new ConvLayer());
    XDET_TRY(L0->init(1, 1, C, 2048, 1, 1, 0, 0, 0, k0->v.data(), nullptr, b0->v.data(), 1));
    emit_planes_next = 1;
    XDET_TRY(add_conv("final_head/subnet_fc", ST_HEAD, pooled, L0, nullptr, 0, &fc));
    const int co = nc + 4;
    std::vector<float> kc((size_t)2048 * co), bc(co);
    for (int ci = 0; ci < 2048; ++ci) {
      for (int j = 0; j < nc; ++j) kc[(size_t)ci * co + j] = k1->v[(size_t)ci * nc + j];
      for (int j = 0; j < 4; ++j) kc[(size_t)ci * co + nc + j] = k2->v[(size_t)ci * 4 + j];
    }
    for (int j = 0; j < nc; ++j) bc[j] = b1->v[j];
    for (int j = 0; j < 4; ++j) bc[nc + j] = b2->v[j];
    ConvLayer* L1 = keep(new ConvLayer());
    XDET_TRY(L1->init(1, 1, 2048, co, 1, 1, 0, 0, 0, kc.data(), nullptr, bc.data(), 0));
    ksplit_next = latency_ksplit;     // 300 rows x 25 outputs: 3 tiles against 64 K steps
    XDET_TRY(add_conv("final_head/fc_cls+fc_loc", ST_HEAD, fc, L1, nullptr, 0, &cls_reg));
    return XDET_OK;
  }

  int build() {
    XDET_REQUIRE(!built, "net already built");
    XDET_REQUIRE(cfg.max_batch > 0 && cfg.image_size >= 64, "bad max_batch / image_size");
    XDET_REQUIRE(cfg.num_anchors == 22, "anchor table is the reference's 22-anchor set (1 extra + 7 scales x 3 ratios)");
    max_batch = cfg.max_batch;
    net_precision = g_default_precision;
    ksplit_design_batch = 1;            // the reference evaluates single images (light_head_rfcn_eval.py:212); only layers a
    ksplit_all = false;                 // builder marks (ksplit_next) are split
    XDET_TRY(build_body());
    XDET_TRY(build_rpn());
    XDET_TRY(large_sep_spectral ? build_large_sep_spectral() : build_large_sep());
    XDET_TRY(build_head());
    XDET_TRY(finish_ksplit());
    const int B = max_batch, A = cfg.num_anchors, R = cfg.rpn_post_nms_top_n;
    n_anchor = fmap * fmap * A;
    // A5: AnchorCreator.get_layer_anchors (anchor_manipulator.py:698-757), layer_step 16, offset .5
    std::vector<float> yx((size_t)fmap * fmap * 2), hw((size_t)A * 2);
    for (int y = 0; y < fmap; ++y)
      for (int x = 0; x < fmap; ++x) {
        yx[((size_t)y * fmap + x) * 2 + 0] = ((float)y + 0.5f) * 16.f / (float)cfg.image_size;
        yx[((size_t)y * fmap + x) * 2 + 1] = ((float)x + 0.5f) * 16.f / (float)cfg.image_size;
      }
    {
      int a = 0;
      hw[0] = 0.1f; hw[1] = 0.1f; ++a;
      const double scales[7] = {0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8}, ratios[3] = {1., 2., .5};
      for (double sc : scales)
        for (double ra : ratios) {
          hw[a * 2 + 0] = (float)(sc / std::sqrt(ra));
          hw[a * 2 + 1] = (float)(sc * std::sqrt(ra));
          ++a;
        }
    }
    XDET_TRY(alloc_bytes(yx.size() * 4, reinterpret_cast<void**>(&anc_yx)));
    XDET_TRY(alloc_bytes(hw.size() * 4, reinterpret_cast<void**>(&anc_hw)));
    XDET_HIP(hipMemcpy(anc_yx, yx.data(), yx.size() * 4, hipMemcpyHostToDevice));
    XDET_HIP(hipMemcpy(anc_hw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    XDET_TRY(alloc_bytes((size_t)B * n_anchor * 4, reinterpret_cast<void**>(&objectness)));
    XDET_TRY(alloc_bytes((size_t)B * n_anchor * 16, reinterpret_cast<void**>(&rpn_boxes)));
    XDET_TRY(alloc_bytes((size_t)B * R * 16, reinterpret_cast<void**>(&proposals)));
    XDET_TRY(alloc_bytes((size_t)B * R * 16, reinterpret_cast<void**>(&head_boxes)));
    XDET_TRY(alloc_bytes((size_t)B * cfg.num_classes * R * 4, reinterpret_cast<void**>(&class_probs)));
    XDET_TRY(alloc_bytes((size_t)B * mid_x.per_image() * 4, reinterpret_cast<void**>(&mid_relu)));
    XDET_TRY(alloc_bytes(proposal_workspace_bytes(B, n_anchor, cfg.rpn_pre_nms_top_n, R), &prop_ws_mem));
    proposal_workspace_carve(prop_ws_mem, B, n_anchor, cfg.rpn_pre_nms_top_n, R, &prop_ws);
    std::vector<int> shp((size_t)B * 2, cfg.image_size);
    std::vector<float> bb((size_t)B * 4);
    for (int i = 0; i < B; ++i) { bb[i * 4] = 0.f; bb[i * 4 + 1] = 0.f; bb[i * 4 + 2] = 1.f; bb[i * 4 + 3] = 1.f; }
    XDET_TRY(alloc_bytes(shp.size() * 4, reinterpret_cast<void**>(&def_shapes)));
    XDET_TRY(alloc_bytes(bb.size() * 4, reinterpret_cast<void**>(&def_bbox)));
    XDET_HIP(hipMemcpy(def_shapes, shp.data(), shp.size() * 4, hipMemcpyHostToDevice));
    XDET_HIP(hipMemcpy(def_bbox, bb.data(), bb.size() * 4, hipMemcpyHostToDevice));
    w.clear();   // host copies are no longer needed
    built = true;
    return XDET_OK;
  }

  int check(int N) const {
    if (!built) {
      set_last_error("net not built");
      return XDET_ERR_STATE;
    }
    XDET_REQUIRE(N > 0 && N <= max_batch, "batch must be in 1..max_batch");
    return XDET_OK;
  }
  int xception_body(const float* images, int N, hipStream_t s) {
    XDET_TRY(check(N));
    XDET_REQUIRE(images != nullptr, "images is NULL");
    cur_images = images;         // the stem op reads the NCHW input directly (graphs are keyed on this pointer)
    if (!stem_direct) XDET_TRY(launch_nchw_to_nhwc4(images, in4.p, N, 3, cfg.image_size, cfg.image_size, 4, s));
    XDET_TRY(run_stage(ST_BODY, N, s));
    return run_stage(ST_EXIT, N, s);
  }
  int entry_and_middle_flow(const float* images, int N, hipStream_t s) {      // up to mid_outputs
    XDET_TRY(check(N));
    XDET_REQUIRE(images != nullptr, "images is NULL");
    cur_images = images;
    if (!stem_direct) XDET_TRY(launch_nchw_to_nhwc4(images, in4.p, N, 3, cfg.image_size, cfg.image_size, 4, s));
    return run_stage(ST_BODY, N, s);
  }
  int rpn_decode(int N, hipStream_t s) {
    XDET_TRY(check(N));
    return launch_rpn_decode(rpn_out.p, rpn_out.ld, 0, 2 * cfg.num_anchors, N, fmap, fmap, cfg.num_anchors, anc_yx,
                             anc_hw, objectness, rpn_boxes, s);
  }
  int get_proposals(int N, hipStream_t s) {
    XDET_TRY(check(N));
    return launch_get_proposals(objectness, rpn_boxes, N, n_anchor, cfg.rpn_pre_nms_top_n, cfg.rpn_post_nms_top_n,
                                cfg.rpn_nms_thres, cfg.rpn_min_size, prop_ws, proposals, s);
  }
  int get_head(int N, hipStream_t s) {
    XDET_TRY(check(N));
    const int C = cfg.bank * cfg.grid * cfg.grid;
    XDET_TRY(launch_psroialign(feat.p, proposals, pooled.p, nullptr, N, C, feat.H, feat.W, cfg.rpn_post_nms_top_n,
                               cfg.grid, cfg.grid, 1, 1, feat.ld, pooled.ld, /*corners=*/1, s));
    return run_stage(ST_HEAD, N, s);
  }
  int head_decode(int N, hipStream_t s) {
    XDET_TRY(check(N));
    return launch_ext_decode_rois(proposals, cls_reg.p + cfg.num_classes, cls_reg.ld,
                                  (int64_t)N * cfg.rpn_post_nms_top_n, head_boxes, s);
  }
  // the whole forward: A11 and the softmax A12 starts from in one pass over the ROIs, then A12 from the probabilities
  int head_decode_probs(int N, hipStream_t s) {
    XDET_TRY(check(N));
    return launch_head_decode_probs(proposals, cls_reg.p, cls_reg.ld, cfg.num_classes, cfg.rpn_post_nms_top_n,
                                    (int64_t)N * cfg.rpn_post_nms_top_n, head_boxes, class_probs, prop_ws.bad, s);
  }
  int bboxes_eval_probs(int N, const int* shapes, const float* bbox, float* ds, float* db, hipStream_t s) {
    XDET_TRY(check(N));
    return launch_bboxes_eval_probs(class_probs, head_boxes, N, cfg.rpn_post_nms_top_n, cfg.num_classes,
                                    shapes ? shapes : def_shapes, bbox ? bbox : def_bbox, cfg.image_size, cfg.image_size,
                                    cfg.select_threshold, cfg.nms_threshold, cfg.nms_topk, ds, db, s, prop_ws.bad);
  }
  // the stage entry points (xdet_net_head_decode / xdet_net_bboxes_eval): each complete on its own, from "cls_reg" / "head_boxes"
  int bboxes_eval(int N, const int* shapes, const float* bbox, float* ds, float* db, hipStream_t s) {
    XDET_TRY(check(N));
    return launch_bboxes_eval(cls_reg.p, cls_reg.ld, head_boxes, N, cfg.rpn_post_nms_top_n, cfg.num_classes,
                              shapes ? shapes : def_shapes, bbox ? bbox : def_bbox, cfg.image_size, cfg.image_size,
                              cfg.select_threshold, cfg.nms_threshold, cfg.nms_topk, ds, db, s, prop_ws.bad);
  }
  void drop_graphs() {
    for (auto& g : graphs) (void)hipGraphExecDestroy(g.second);
    graphs.clear();
    graph_order.clear();
  }
  int calibrate(const float* images, int N, hipStream_t s, int* n_scaled) {
    XDET_TRY(check(N));
    XDET_REQUIRE(images != nullptr, "calibrate: images is NULL");
    if (n_scaled) *n_scaled = 0;
    if (net_precision == PREC_F32 || pscales.empty()) return XDET_OK;
    drop_graphs();                                   // graphs bake kernel arguments (the split passes' multipliers)
    const size_t slots = (size_t)N * (cfg.num_classes - 1) * cfg.nms_topk;
    float *ds = nullptr, *db = nullptr;
    XDET_HIP(hipMalloc(reinterpret_cast<void**>(&ds), slots * 4));
    XDET_HIP(hipMalloc(reinterpret_cast<void**>(&db), slots * 16));
    const int rc = calibrate_planes(N, s, n_scaled, [&](hipStream_t st) { return forward_eager(images, N, nullptr, nullptr, ds, db, st); });
    (void)hipFree(ds);
    (void)hipFree(db);
    return rc;
  }
  int forward_eager(const float* images, int N, const int* shapes, const float* bbox, float* ds, float* db,
                    hipStream_t s) {
    XDET_TRY(entry_and_middle_flow(images, N, s));
    // fork: the RPN branch (3x3 conv, 1x1 heads, decode, top-k, NMS -- a long conv and then small latency-bound
    // launches) needs mid_outputs only (net/xception_body.py:339,381-400): it runs on a side stream under the exit flow
    // AND the large-separable convs (round 4; it used to fork behind the exit flow, where a single image's RPN conv --
    // 207 K steps on 64 workgroups -- was the longer branch and sat on the critical path).
    // While per-op profiling is on, the branch stays on the main stream: an event pair around a launch that
    // shares the chip with the other branch's kernels would time the sharing, not the kernel.
    if (profiling || !rpn_side_stream) {
      XDET_TRY(run_stage(ST_EXIT, N, s));
      XDET_TRY(run_stage(ST_RPN, N, s));
      XDET_TRY(rpn_decode(N, s));
      XDET_TRY(get_proposals(N, s));
      XDET_TRY(run_stage(ST_LSEP, N, s));
    } else {
      if (!aux) {
        XDET_HIP(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
        XDET_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        XDET_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
      }
      XDET_HIP(hipEventRecord(ev_fork, s));
      XDET_HIP(hipStreamWaitEvent(aux, ev_fork, 0));
      // The longer branch (exit flow + large-separable convs: the critical path of a single image) is issued FIRST.  In a
      // captured graph the branch whose nodes were created first continues on the queue of the fork node; the other one
      // starts behind a cross-queue signal -- and when the RPN branch was issued first, the replayed exit flow started only
      // when the RPN 3x3 conv had FINISHED (76 us at one image; rocprofv3 kernel trace of round 6, 29 of 29 steps).
      XDET_TRY(run_stage(ST_EXIT, N, s));
      XDET_TRY(run_stage(ST_LSEP, N, s));
      XDET_TRY(run_stage(ST_RPN, N, aux));
      XDET_TRY(rpn_decode(N, aux));
      XDET_TRY(get_proposals(N, aux));
      XDET_HIP(hipEventRecord(ev_join, aux));
      XDET_HIP(hipStreamWaitEvent(s, ev_join, 0));   // join before the head consumes the proposals
    }
    XDET_TRY(get_head(N, s));
    XDET_TRY(head_decode_probs(N, s));
    if (check_range && net_precision != PREC_F32) {
      for (const auto& b : f32_bufs) {
        float limit;
        int relu;
        f32_limit(b.first, &limit, &relu);
        XDET_TRY(launch_range_check(b.first, N, b.second, limit, prop_ws.bad, s, 1, 0, relu));
      }
      for (const auto& b : planes_bufs) XDET_TRY(launch_range_check_planes(b.hi, N, b.pix_per_image, b.ld, prop_ws.bad, s));
      for (const auto& f : extra_range_checks) XDET_TRY(f(N, s));
    }
    return bboxes_eval_probs(N, shapes, bbox, ds, db, s);
  }
};

// ---------------------------------------------------------------------------------------
// A13: ResNet-50 v2 trunk
// ---------------------------------------------------------------------------------------
struct ResNetTrunk : Plan {
  int image_size = 480;
  bool built = false;
  Buf in4, outb;
  double flops = 0;
  bool ksplit_enabled = true;                      // xdet_resnet_set_option "ksplit" = off: the round-3 launch plan (A/B measurements)
  bool stem7_enabled = true;                       // option "stem7" = off: the stem conv on the generic small-cin kernel (A/B runs, tests)
  bool stem7_direct = false;                       // decided at build: the stem op reads the NCHW images itself
  const float* cur_images = nullptr;               // (graphs are keyed on this pointer)
  bool stem_pool_bn = true;                        // option "stem_pool" = off: pool and pre-activation as two passes (A/B runs, tests)
  bool bneck_enabled = true;                       // identity blocks the fused kernel supports run as one launch
  // ONE decision per forward, made at the top of xdet_resnet_forward and read by every op: the fused kernels run (and the
  // planes they keep on the CU are not written).  Part of the graph key.  A trunk instance serves one stream at a time.
  bool bneck_fused_now = false;
  bool projcat_enabled = true;                     // option "projcat" = off: projection shortcuts as their own GEMM + a residual add (A/B runs, tests)
  unsigned short *stem_cat_hi = nullptr, *stem_cat_lo = nullptr;   // second destination of the stem's pool + pre-activation pass
  float stem_cat_mul = 1.f;
  int stem_cat_c32 = 0, stem_cat_pidx = -1;        // (stage 1's projection block reads the block input inside a concatenated operand)
  bool preconv_enabled = true;                     // option "preconv" = off: stage 2's opening 1x1 convs read planes (A/B runs, tests)
  struct PreconvLayers { ConvLayer *Lprev, *La; };  // a conv1x1 that makes its pre-activation from the raw input (resnet_preconv.hip)
  std::vector<PreconvLayers> preconv_layers;
  struct BneckGroup {
    ConvLayer *La, *Lb, *Lc, *Lprev;               // the block's three convs; the closing conv of the block before it
    BneckLaunch a;
    size_t op_first;
    bool next_fused = false;
  };
  std::vector<BneckGroup> bneck_groups;
  // one decision per forward (the answer cannot change inside one): no calibration pass is measuring, and no planes tensor a
  // fused kernel keeps on the CU carries an activation pre-scale
  bool planes_dropped() const override { return (!bneck_groups.empty() || !preconv_layers.empty()) && bneck_fused_now; }
  bool bneck_all_ok() const {
    if (after_op) return false;
    for (const BneckGroup& g : bneck_groups)
      if (g.Lprev->out_exp != 0 || g.La->in_exp != 0 || g.La->out_exp != 0 || g.Lb->in_exp != 0 || g.Lb->out_exp != 0 || g.Lc->in_exp != 0)
        return false;
    for (const PreconvLayers& g : preconv_layers)
      if (g.Lprev->out_exp != 0 || g.La->in_exp != 0 || g.La->out_exp != 0) return false;
    return true;
  }
  typedef std::array<uintptr_t, 4> Key;            // (N, images, out, fused forms or not): everything a captured graph bakes in
  std::map<Key, hipGraphExec_t> graphs;
  std::vector<Key> graph_order;
  ~ResNetTrunk() {
    for (auto& g : graphs) (void)hipGraphExecDestroy(g.second);
  }

  // Pre-activation bottlenecks (net/resnet_v2.py:142-184).  The BN+ReLU that follows the two inner
  // convs is folded into their epilogues; the one that opens a block cannot be (the raw block input
  // is also the identity shortcut), so it is its own element-wise pass.
  int add_bn_relu(const std::string& bn, const Buf& in, Buf* out);
  int build();
};

// also writes the result as split planes [pix/16][ld/32][16][32] when hi != NULL (its consumers are 1x1
// convs on the LDS-DMA path)
__global__ void bn_relu_kernel(const float* __restrict__ in, const float* __restrict__ scale,
                               const float* __restrict__ shift, float* __restrict__ out,
                               unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, int64_t npix, int ld,
                               float mul) {
  // mul = 2^-e, the planes' activation pre-scale (1 by default): hi + lo = relu(bn(x)) * mul; the f32 copy is unscaled
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const int c4n = ld >> 2;
  const int64_t total = npix * c4n;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    float4 v = reinterpret_cast<const float4*>(in)[i];
    const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
    // (a ReLU that keeps NaN, as the conv epilogue's: an overflowed split operand upstream must reach the caller, not turn into 0)
    v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
    v.x = !(v.x <= 0.f) ? v.x : 0.f; v.y = !(v.y <= 0.f) ? v.y : 0.f; v.z = !(v.z <= 0.f) ? v.z : 0.f; v.w = !(v.w <= 0.f) ? v.w : 0.f;
    reinterpret_cast<float4*>(out)[i] = v;
    if (hi) {
      const int64_t pix = i / c4n;
      v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
      const _Float16 h0 = (_Float16)v.x, h1 = (_Float16)v.y, h2 = (_Float16)v.z, h3 = (_Float16)v.w;
      h4 hv = {h0, h1, h2, h3};
      h4 lv = {(_Float16)(v.x - (float)h0), (_Float16)(v.y - (float)h1), (_Float16)(v.z - (float)h2),
               (_Float16)(v.w - (float)h3)};
      const int64_t o = (((pix >> 4) * (ld >> 5) + (c >> 5)) << 9) + ((pix & 15) << 5) + (c & 31);
      *reinterpret_cast<uint2*>(hi + o) = *reinterpret_cast<uint2*>(&hv);
      *reinterpret_cast<uint2*>(lo + o) = *reinterpret_cast<uint2*>(&lv);
    }
  }
}

int ResNetTrunk::add_bn_relu(const std::string& bn, const Buf& in, Buf* out) {
  std::vector<float> sc, sh;
  XDET_TRY(fold_bn(bn, in.C, 1e-5f, nullptr, &sc, &sh));
  sc.resize(in.ld, 0.f);
  sh.resize(in.ld, 0.f);
  float *dsc, *dsh;
  XDET_TRY(alloc_bytes(sc.size() * 4, reinterpret_cast<void**>(&dsc)));
  XDET_TRY(alloc_bytes(sh.size() * 4, reinterpret_cast<void**>(&dsh)));
  XDET_HIP(hipMemcpy(dsc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
  XDET_HIP(hipMemcpy(dsh, sh.data(), sh.size() * 4, hipMemcpyHostToDevice));
  XDET_TRY(new_buf(in.H, in.W, in.C, out));
  if (g_default_precision != PREC_F32 && in.ld % 32 == 0) {
    XDET_TRY(new_planes(out));
    pscales[out->pidx].name = bn + " (pre-activation planes)";
  }
  const Buf i = in, o = *out;
  ops.push_back({bn, 0, 0.0, [=](int N, hipStream_t s) {
                   const int64_t npix = (int64_t)N * i.H * i.W;
                   const int blocks = (int)std::min<int64_t>(cdiv(npix * (i.ld / 4), 256), 256 * 32);
                   hipLaunchKernelGGL(bn_relu_kernel, dim3(blocks), dim3(256), 0, s, i.p, dsc, dsh, o.p, o.hi, o.lo, npix,
                                      i.ld, pmul(o.pidx));
                   XDET_LAUNCH_CHECK();
                   return (int)XDET_OK;
                 }});
  return XDET_OK;
}

int ResNetTrunk::build() {
  XDET_REQUIRE(!built, "net already built");
  // stages 3-4 at BASELINE config 2's batch 8 are 57 / 15 M tiles against 72- / 144-step K loops: split-K (option "ksplit")
  if (g_default_precision != PREC_F32 && ksplit_enabled) { ksplit_design_batch = 8; ksplit_all = true; }
  net_precision = g_default_precision;
  int ci = 0, bi = 0;
  auto cname = [&]() { std::string n = ci == 0 ? "conv2d" : "conv2d_" + std::to_string(ci); ++ci; return n; };
  auto bname = [&]() { std::string n = bi == 0 ? "batch_normalization" : "batch_normalization_" + std::to_string(bi); ++bi; return n; };
  XDET_TRY(new_buf(image_size, image_size, 3, &in4));
  Buf x, t;
  // conv2d_fixed_padding(7, stride 2): explicit pad 3/3 then VALID (:89-100)
  XDET_TRY(conv_bn(cname(), "", 0.f, 0, in4, 7, 64, 2, 2, 0, nullptr, 0, &x, 3));
  // ... on its own kernel, straight from the NCHW input (resnet_stem.hip): the generic small-cin kernel gathers one 16-byte
  // load per (pixel, tap) from an NHWC4 copy of the image; same products, same order
  if (stem7_enabled && g_default_precision == PREC_F16X3 && ops.size() == 1 &&
      resnet_stem7x7_supported(7, 7, 3, 64, 2, 2, 3, image_size)) {
    ConvLayer* L0 = static_cast<ConvLayer*>(layers.back().get());
    const Buf o = x;
    const int S = image_size;
    ops[0].run = [=](int N, hipStream_t s) {
      return launch_resnet_stem7x7(cur_images, L0->d_wt_hi, L0->d_wt_lo, L0->d_scale, L0->d_shift, o.p, N, S, s);
    };
    ops[0].name += " [LDS-staged patch, from NCHW]";
    stem7_direct = true;
  }
  // initial_max_pool (:311-330).  On the split path its one reader is the first block's pre-activation, which is read as
  // planes only: pool + that block's bn + ReLU + split in one pass (no pooled f32 tensor, one launch less)
  bool stem_pre_fused = false;
  Buf stem_pre;
  if (g_default_precision != PREC_F32 && x.ld % 32 == 0 && stem_pool_bn) {
    int Ho, Wo, pt, pl;
    same_pad(x.H, 3, 2, 1, &pt, &Ho);
    same_pad(x.W, 3, 2, 1, &pl, &Wo);
    const std::string bn0 = "batch_normalization";
    std::vector<float> sc, sh;
    XDET_TRY(fold_bn(bn0, x.C, 1e-5f, nullptr, &sc, &sh));
    sc.resize(x.ld, 0.f);
    sh.resize(x.ld, 0.f);
    float *dsc, *dsh;
    XDET_TRY(alloc_bytes(sc.size() * 4, reinterpret_cast<void**>(&dsc)));
    XDET_TRY(alloc_bytes(sh.size() * 4, reinterpret_cast<void**>(&dsh)));
    XDET_HIP(hipMemcpy(dsc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
    XDET_HIP(hipMemcpy(dsh, sh.data(), sh.size() * 4, hipMemcpyHostToDevice));
    stem_pre = Buf();
    stem_pre.H = Ho; stem_pre.W = Wo; stem_pre.C = x.C; stem_pre.ld = x.ld; stem_pre.no_f32 = true;
    XDET_TRY(new_planes(&stem_pre));
    pscales[stem_pre.pidx].name = bn0 + " (pre-activation planes)";
    const Buf i = x, o = stem_pre;
    ops.push_back({"initial_max_pool + " + bn0, 0, 0.0, [=](int N, hipStream_t s) {
                     return launch_maxpool3x3s2_bn_planes(i.p, dsc, dsh, o.hi, o.lo, N, i.H, i.W, i.C, i.ld, Ho, Wo, pt, pl,
                                                          pmul(o.pidx), s, stem_cat_hi, stem_cat_lo, stem_cat_c32, pmul(stem_cat_pidx) * stem_cat_mul);
                   }});
    stem_pre_fused = true;
    // block 0 takes its shortcut from a projection of the pre-activation: the pooled tensor has no f32 reader.  `x` keeps the
    // shape only (no pointer, no planes, no flags: the block loop copies it into its output descriptors)
    x = Buf();
    x.H = Ho; x.W = Wo; x.C = stem_pre.C; x.ld = stem_pre.ld;
  } else {
    XDET_TRY(add_pool("initial_max_pool", 0, x, nullptr, &t));
    x = t;
  }
  const int filters[4] = {64, 128, 256, 512}, blocks[4] = {3, 4, 6, 3}, strides[4] = {1, 2, 2, 2};
  ConvLayer* prev_Lc = nullptr;  // the closing conv of the block before (its planes affine is the next block's pre-activation BN)
  int prev_group = -1;
  int prev_drop_idx = -1;        // planes_drop_ok entry of the previous block's closing conv (-1: none / that block runs fused)
  bool have_prev_pl = false;
  Buf fused_pre;                 // planes-only pre-activation of the NEXT block, written by this block's last conv
  bool have_fused = false;
  const float *fused_sc = nullptr, *fused_sh = nullptr;   // its BN (a strided projection applies it to its own subsample)
  for (int st = 0; st < 4; ++st)
    for (int b = 0; b < blocks[st]; ++b) {
      const int f = filters[st], s = b == 0 ? strides[st] : 1;
      Buf pre, shortcut = x, y1, y2, y3;
      if (have_fused) {
        pre = fused_pre;
        (void)bname();                              // its BN was folded into the previous block's epilogue
      } else if (st == 0 && b == 0 && stem_pre_fused) {
        pre = stem_pre;
        (void)bname();                              // ... into the pool pass
      } else {
        XDET_TRY(add_bn_relu(bname(), x, &pre));
      }
      have_fused = false;
      // A projection shortcut folded into the block's closing GEMM (net/resnet_v2.py:160-184: shortcut = projection(pre), output =
      // conv3(...) + shortcut, neither followed by a BN): [y2 | pre'] x [w_c ; w_proj] is ONE contraction over cmid + cin
      // channels -- no projection launch, no 4f-channel shortcut tensor written and read back (118 / 59 / 29 / 15 MB each way at
      // batch 8).  The operand is one planes tensor: the 3x3 conv's epilogue writes its channel blocks [0, cmid/32), the pass that
      // makes pre' (the stem's pool + pre-activation pass / the stride-2 subsample of relu(bn(x))) the blocks behind them.  One f32
      // accumulation over both parts instead of two roundings and an add: not bit-identical to the two-GEMM form, same tolerance
      // against the oracle (tests/test_gpu_resnet.py).
      const bool cat_s1 = s == 1 && st == 0 && pre.hi && pre.pidx >= 0 && !pre.planes_relu;
      const bool cat_s2 = s == 2 && pre.no_f32 && x.p && fused_sc && x.ld % 32 == 0;
      const bool cat_on = projcat_enabled && b == 0 && g_default_precision == PREC_F16X3 && (cat_s1 || cat_s2) && f % 32 == 0 &&
                          pre.ld % 32 == 0 && pre.ld == pre.C;
      const std::string cproj = b == 0 ? cname() : std::string();
      Buf cat;
      int cat_d = 0;
      if (cat_on) {
        cat.H = (pre.H + s - 1) / s; cat.W = (pre.W + s - 1) / s; cat.C = f + pre.C; cat.ld = f + pre.ld; cat.no_f32 = true;
        XDET_TRY(new_planes(&cat));
        pscales[cat.pidx].name = cproj + " + closing conv: [3x3 output | block input] operand";
        const size_t off = (size_t)(f >> 5) << 9;          // halves: the block input's first channel block inside a 16-pixel group
        // The two parts share one weight pre-scale per output channel and one activation pre-scale: balance them.  The block-input
        // part is stored as pre * 2^-cat_d and w_proj enters as w_proj * 2^cat_d (exact), cat_d = the binade distance of the two
        // matrices' largest magnitudes -- a projection whose weights are 2^17 below the closing conv's (and whose input is 2^17
        // above: test_trunk_pre_activations_beyond_the_f16_range) would otherwise lose its lo plane to f16 underflow.
        {
          const HostTensor *kc0, *kp0;
          XDET_TRY(need("conv2d_" + std::to_string(ci + 2) + "/kernel", &kc0, {1, 1, f, 4 * f}));   // (c3: three names after cproj)
          XDET_TRY(need(cproj + "/kernel", &kp0, {1, 1, pre.C, 4 * f}));
          float mc = 0.f, mp = 0.f;
          for (float v : kc0->v) mc = std::max(mc, std::fabs(v));
          for (float v : kp0->v) mp = std::max(mp, std::fabs(v));
          int ec = 0, ep = 0;
          if (mc > 0.f && mp > 0.f && std::isfinite(mc) && std::isfinite(mp)) { (void)frexpf(mc, &ec); (void)frexpf(mp, &ep); }
          cat_d = ec - ep;
        }
        const float part_mul = ldexpf(1.f, -cat_d);
        if (cat_s1 && stem_pre_fused) {               // the stem's pool + pre-activation pass writes the block input twice
          stem_cat_hi = cat.hi + off; stem_cat_lo = cat.lo + off; stem_cat_c32 = cat.ld >> 5; stem_cat_pidx = cat.pidx;
          stem_cat_mul = part_mul;
        } else if (cat_s1) {                          // (two-pass stem, option "stem_pool" = off: a copy of the pre-activation planes)
          const Buf i = pre, o = cat;
          ops.push_back({cproj + "/copy of the block input [into the closing conv's operand]", 0, 0.0, [=](int N, hipStream_t s_) {
                           return launch_planes_copy_blocks(i.hi, i.lo, o.hi + off, o.lo + off, (int64_t)N * i.H * i.W, i.ld, o.ld >> 5,
                                                            pmul(o.pidx) * part_mul / pmul(i.pidx), s_);
                         }});
        } else {
          const Buf i = x, o = cat;
          const float *psc = fused_sc, *psh = fused_sh;
          ops.push_back({cproj + "/subsample_split [into the closing conv's operand]", 0, 0.0, [=](int N, hipStream_t s_) {
                           return launch_split_f32_subsample2(i.p, o.hi + off, o.lo + off, N, i.H, i.W, i.ld, s_, psc, psh, pmul(o.pidx) * part_mul,
                                                              o.ld >> 5);
                         }});
        }
        shortcut = Buf();
      } else if (b == 0) {
        if (pre.no_f32 && s > 1) {
          // the pre-activation exists as full-resolution planes only (conv1 reads those); the stride-2 projection
          // takes the raw block input and applies the BN + ReLU to the quarter of the pixels it reads
          XDET_TRY(conv_bn(cproj, "", 0.f, 0, x, 1, 4 * f, s, 2, 0, nullptr, 0, &shortcut, 0, fused_sc, fused_sh));
        } else {
          XDET_TRY(conv_bn(cproj, "", 0.f, 0, pre, 1, 4 * f, s, s > 1 ? 2 : 1, 0, nullptr, 0, &shortcut, 0));
        }
      }
      const std::string c1 = cname(), b1 = bname(), c2 = cname(), b2 = bname(), c3 = cname();
      // conv1x1 -> (BN+ReLU fused into its epilogue) -> conv3x3/s -> (BN+ReLU fused) -> conv1x1 + shortcut
      const size_t op_first = ops.size();
      const int my_prev_group = prev_group;       // the bneck group of the block before, if it runs fused
      prev_group = -1;
      emit_planes_next = 3;                       // the 3x3 (stride 1 or 2) takes its input as planes (only)
      XDET_TRY(conv_bn(c1, b1, 1e-5f, 0, pre, 1, f, 1, 1, 1, nullptr, 0, &y1));
      ConvLayer* La = static_cast<ConvLayer*>(layers.back().get());
      // The opening 1x1 of a stage-2 block on resnet_preconv.hip: it makes relu(bn(x)) from the raw block input itself, so
      // the producer of x does not write the planes copy (decided per forward with the fused blocks: bneck_all_ok()).
      const bool preconv = preconv_enabled && g_default_precision == PREC_F16X3 && have_prev_pl && pre.hi && pre.no_f32 && x.p &&
                           ops.size() == op_first + 1 && y1.hi && La->ksplit <= 1 &&     // (a layer with a split reduction keeps its summation tree)
                           resnet_preconv_supported(pre.C, f, (int64_t)max_batch * pre.H * pre.W) &&
                           !(bneck_enabled && b > 0 && resnet_bneck_supported(pre.C, f, 4 * f, pre.H, pre.W, max_batch));
      if (preconv) {
        preconv_layers.push_back({prev_Lc, La});
        if (my_prev_group >= 0) bneck_groups[my_prev_group].next_fused = true;
        if (prev_drop_idx >= 0) planes_drop_ok[prev_drop_idx] = 1;
        ConvLayer* Lp = prev_Lc;
        const Buf xi = x, yo = y1;
        const int cin = pre.C, cm = f;
        const auto run_a = ops[op_first].run;
        ops[op_first].run = [=](int N, hipStream_t st) {
          if (!bneck_fused_now) return run_a(N, st);
          return launch_resnet_preconv(xi.p, Lp->d_pl_scale, Lp->d_pl_shift, La->d_wt_hi_b, La->d_wt_lo_b, La->d_scale, La->d_shift,
                                       yo.hi, yo.lo, (int64_t)N * xi.H * xi.W, cin, cm, st);
        };
        ops[op_first].name += " [pre-activation on the CU]";
      }
      emit_planes_next = 3;                       // the closing 1x1 always does
      if (cat_on) emit_planes_into = &cat;
      XDET_TRY(conv_bn(c2, b2, 1e-5f, 0, y1, 3, f, s, s > 1 ? 2 : 1, 1, nullptr, 0, &y2, 1));
      XDET_REQUIRE(!cat_on || (y2.hi == cat.hi && y2.H == cat.H && y2.W == cat.W), "plan: the 3x3 conv did not take the concatenated operand");
      ConvLayer* Lb = static_cast<ConvLayer*>(layers.back().get());
      // The next block opens with BN+ReLU of this block's output.  Fold it in: the closing conv writes its f32
      // output (the identity shortcut) AND relu(bn_next(output)) as planes, and the separate element-wise pass
      // disappears.  (A stage opener's strided projection cannot read those full-resolution planes: see above.)
      const bool last = st == 3 && b == blocks[st] - 1;
      float *nsc = nullptr, *nsh = nullptr;
      if (!last && g_default_precision != PREC_F32) {
        const std::string nbn = bi == 0 ? "batch_normalization" : "batch_normalization_" + std::to_string(bi);
        std::vector<float> sc, sh;
        XDET_TRY(fold_bn(nbn, 4 * f, 1e-5f, nullptr, &sc, &sh));
        const int ldn = round_up(4 * f, 32);
        sc.resize(ldn, 0.f);
        sh.resize(ldn, 0.f);
        XDET_TRY(alloc_bytes(sc.size() * 4, reinterpret_cast<void**>(&nsc)));
        XDET_TRY(alloc_bytes(sh.size() * 4, reinterpret_cast<void**>(&nsh)));
        XDET_HIP(hipMemcpy(nsc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
        XDET_HIP(hipMemcpy(nsh, sh.data(), sh.size() * 4, hipMemcpyHostToDevice));
        emit_planes_next = 1;
        emit_bn_scale = sc;                         // (host copies: the conv keeps its own device arrays, which carry the
        emit_bn_shift = sh;                         //  planes' pre-scale; nsc / nsh below stay as they are for the projection)
      }
      // (a stage's opening block in front of an identity block that can run fused: its planes copy is optional)
      // (the planes copy of this block's output has one reader, the next block's opening conv: if that block turns out to run on
      //  a kernel that makes its own pre-activation, it marks this conv's planes as droppable)
      planes_optional_next = g_default_precision == PREC_F16X3 && nsc != nullptr;
      if (cat_on) {
        const HostTensor *kc, *kp;
        XDET_TRY(need(c3 + "/kernel", &kc, {1, 1, f, 4 * f}));
        XDET_TRY(need(cproj + "/kernel", &kp, {1, 1, pre.C, 4 * f}));
        std::vector<float> kcat(kc->v);                                   // HWIO, 1 x 1: rows = input channels
        for (float v : kp->v) kcat.push_back(ldexpf(v, cat_d));             // (the block-input part is stored * 2^-cat_d)
        ConvLayer* L = keep(new ConvLayer());
        XDET_TRY(L->init(1, 1, f + pre.C, 4 * f, 1, 1, 1, 0, 0, kcat.data(), nullptr, nullptr, 0));
        XDET_TRY(add_conv(c3 + " + " + cproj + " [shortcut projection folded into the reduction]", 0, cat, L, nullptr, 0, &y3));
      } else {
        XDET_TRY(conv_bn(c3, "", 0.f, 0, y2, 1, 4 * f, 1, 1, 0, &shortcut, 0, &y3));
      }
      ConvLayer* Lc = static_cast<ConvLayer*>(layers.back().get());
      const int my_drop_idx = last_optional_idx;  // this block's closing conv, if its planes copy is optional
      // An identity block as ONE kernel (resnet_bneck.hip), reading the raw block input (it applies the pre-activation BN +
      // ReLU itself, with the arithmetic of the planes copy the previous block's closing conv writes) and writing the
      // pre-activation planes of the next block only if that one runs as three launches.  The three ops stay in the plan:
      // a calibration pass measures the inner planes tensors behind them, and a trunk in which any tensor of a fusable
      // block carries an activation pre-scale runs every block as three launches (all or nothing per forward: a fused
      // block does not write the planes an unfused successor would read).
      if (bneck_enabled && b > 0 && g_default_precision == PREC_F16X3 && nsc && ops.size() == op_first + 3 && pre.hi && y3.hi &&
          have_prev_pl && La->ksplit < 1 && Lb->ksplit < 1 && Lc->ksplit < 1 &&
          resnet_bneck_supported(pre.C, f, 4 * f, pre.H, pre.W, max_batch)) {
        BneckGroup gr;
        gr.La = La; gr.Lb = Lb; gr.Lc = Lc; gr.Lprev = prev_Lc;
        gr.a.x = shortcut.p;
        gr.a.wa_hi = La->d_wt_hi_b; gr.a.wa_lo = La->d_wt_lo_b; gr.a.wb_hi = Lb->d_wt_hi_b; gr.a.wb_lo = Lb->d_wt_lo_b;
        gr.a.wc_hi = Lc->d_wt_hi_b; gr.a.wc_lo = Lc->d_wt_lo_b;
        gr.a.sc_a = La->d_scale; gr.a.sh_a = La->d_shift; gr.a.sc_b = Lb->d_scale; gr.a.sh_b = Lb->d_shift;
        gr.a.sc_c = Lc->d_scale; gr.a.sh_c = Lc->d_shift;
        gr.a.pre_sc = gr.a.pre_sh = gr.a.pl_sc = gr.a.pl_sh = nullptr;     // (the planes affines: read at launch time)
        gr.a.out = y3.p; gr.a.out_hi = y3.hi; gr.a.out_lo = y3.lo;
        gr.a.H = pre.H; gr.a.W = pre.W; gr.a.cin = pre.C; gr.a.cmid = f; gr.a.cout = 4 * f;
        gr.op_first = op_first;
        if (!bneck_groups.empty() && bneck_groups.back().op_first + 3 == op_first) bneck_groups.back().next_fused = true;
        if (prev_drop_idx >= 0) planes_drop_ok[prev_drop_idx] = 1;
        bneck_groups.push_back(gr);
        const size_t gi = bneck_groups.size() - 1;
        prev_group = (int)gi;
        const auto run_a = ops[op_first].run, run_b = ops[op_first + 1].run, run_c = ops[op_first + 2].run;
        ops[op_first].run = [=](int N, hipStream_t st) {
          if (!bneck_fused_now) return run_a(N, st);
          const BneckGroup& G = bneck_groups[gi];
          BneckLaunch l = G.a;
          l.pre_sc = G.Lprev->d_pl_scale; l.pre_sh = G.Lprev->d_pl_shift;
          l.pl_sc = G.Lc->d_pl_scale; l.pl_sh = G.Lc->d_pl_shift;
          if (G.next_fused) l.out_hi = l.out_lo = nullptr;          // the next block makes its own pre-activation
          return launch_resnet_bneck(l, N, st);
        };
        ops[op_first + 1].run = [=](int N, hipStream_t st) { return bneck_fused_now ? (int)XDET_OK : run_b(N, st); };
        ops[op_first + 2].run = [=](int N, hipStream_t st) { return bneck_fused_now ? (int)XDET_OK : run_c(N, st); };
        ops[op_first].name += " [+2: one kernel]";
      }
      prev_Lc = Lc;
      have_prev_pl = nsc != nullptr && g_default_precision != PREC_F32;
      prev_drop_idx = prev_group >= 0 ? -1 : my_drop_idx;      // (a fused block never runs its closing conv's op)
      if (nsc) {
        fused_pre = y3;                             // same shape; lives as planes only
        fused_pre.p = nullptr;
        fused_pre.no_f32 = true;
        fused_pre.planes_relu = false;
        y3.hi = y3.lo = nullptr;                    // the f32 tensor itself has no planes
        have_fused = true;
        fused_sc = nsc;
        fused_sh = nsh;
      }
      x = y3;
    }
  XDET_TRY(add_bn_relu(bname(), x, &outb));
  for (const Op& op : ops) flops += std::max(op.flops, 0.0);
  w.clear();
  XDET_TRY(finish_ksplit());
  built = true;
  return XDET_OK;
}

}  // namespace xdet

// =========================================================================================
// C-ABI
// =========================================================================================
using namespace xdet;

extern "C" {

const char* xdet_last_error(void) { return g_last_error.c_str(); }
int xdet_version(void) { return 1; }
int xdet_device_count(int* n) { XDET_HIP(hipGetDeviceCount(n)); return XDET_OK; }
int xdet_set_device(int dev) { XDET_HIP(hipSetDevice(dev)); return XDET_OK; }
int xdet_device_pci_bus_id(int dev, char* buf, int buflen) {
  XDET_REQUIRE(buf && buflen >= 16, "device_pci_bus_id: need a buffer of >= 16 bytes");
  XDET_HIP(hipDeviceGetPCIBusId(buf, buflen, dev));
  return XDET_OK;
}
int xdet_probe_ipc(void) {
  void* p = nullptr;
  XDET_HIP(hipMalloc(&p, 1 << 16));
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, p);
  (void)hipFree(p);
  XDET_HIP(e);
  return XDET_OK;
}
int xdet_set_default_precision(int mode) {
  XDET_REQUIRE(mode == PREC_F32 || mode == PREC_F16X3 || mode == PREC_F16, "precision must be 0 (f32), 1 (f16x3) or 2 (f16)");
  g_default_precision = mode;
  return XDET_OK;
}
int xdet_get_default_precision(void) { return g_default_precision; }

int xdet_malloc(void** dptr, size_t bytes) { XDET_REQUIRE(dptr, "dptr is NULL"); XDET_HIP(hipMalloc(dptr, std::max<size_t>(bytes, 16))); return XDET_OK; }
int xdet_free(void* dptr) { if (dptr) XDET_HIP(hipFree(dptr)); return XDET_OK; }
int xdet_memset(void* dptr, int value, size_t bytes, void* stream) { XDET_HIP(hipMemsetAsync(dptr, value, bytes, S(stream))); return XDET_OK; }
int xdet_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) { XDET_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(stream))); return XDET_OK; }
int xdet_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) { XDET_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(stream))); XDET_HIP(hipStreamSynchronize(S(stream))); return XDET_OK; }
int xdet_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) { XDET_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S(stream))); return XDET_OK; }
int xdet_stream_create(void** stream) { hipStream_t s; XDET_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); *stream = s; return XDET_OK; }
int xdet_stream_destroy(void* stream) { XDET_HIP(hipStreamDestroy(S(stream))); return XDET_OK; }
int xdet_stream_sync(void* stream) { XDET_HIP(hipStreamSynchronize(S(stream))); return XDET_OK; }
int xdet_event_create(void** ev) { hipEvent_t e; XDET_HIP(hipEventCreate(&e)); *ev = e; return XDET_OK; }
int xdet_event_destroy(void* ev) { XDET_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev))); return XDET_OK; }
int xdet_event_record(void* ev, void* stream) { XDET_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), S(stream))); return XDET_OK; }
int xdet_event_elapsed_ms(void* a, void* b, float* ms) {
  XDET_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(b)));
  XDET_HIP(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(a), reinterpret_cast<hipEvent_t>(b)));
  return XDET_OK;
}

int xdet_psroialign_fwd(const float* feat, const float* rois, float* pooled, int32_t* index, int N, int C, int H,
                        int W, int R, int grid_w, int grid_h, int use_max, int feat_layout, int ldc, int out_ld,
                        int rois_are_corners, void* stream) {
  XDET_REQUIRE(feat && rois && pooled, "inputs/rois/pooled_features must not be NULL");
  return launch_psroialign(feat, rois, pooled, index, N, C, H, W, R, grid_w, grid_h, use_max, feat_layout,
                           feat_layout == 0 ? C : ldc, out_ld, rois_are_corners, S(stream));
}

int xdet_psroialign_grad(const float* rois, const float* grad_pooled, const int32_t* pooled_index, float* grad_feat,
                         int N, int C, int H, int W, int R, int grid_w, int grid_h, int use_max, int feat_layout,
                         int ldc, void* stream) {
  return launch_psroialign_grad(rois, grad_pooled, pooled_index, grad_feat, N, C, H, W, R, grid_w, grid_h, use_max,
                                feat_layout, feat_layout == 0 ? C : ldc, S(stream));
}

int xdet_conv_create(void** layer, int kh, int kw, int cin, int cout, int stride, int dilation, int pad_mode,
                     int pad_t, int pad_l, const float* k, const float* scale, const float* shift, int relu_out) {
  XDET_REQUIRE(layer, "layer is NULL");
  std::unique_ptr<ConvLayer> L(new ConvLayer());
  XDET_TRY(L->init(kh, kw, cin, cout, stride, dilation, pad_mode, pad_t, pad_l, k, scale, shift, relu_out));
  *layer = static_cast<LayerBase*>(L.release());
  return XDET_OK;
}
int xdet_conv_forward(void* layer, const float* in, int N, int H, int W, int ld_in, float* out, int ld_out,
                      const float* residual, int relu_in, void* stream) {
  LayerBase* b = static_cast<LayerBase*>(layer);
  XDET_REQUIRE(b && b->kind == 1, "not a conv layer");
  DeviceGuard guard(b->device);
  return static_cast<ConvLayer*>(b)->forward(in, N, H, W, ld_in, out, ld_out, residual, relu_in, S(stream));
}
int xdet_split_f32(const float* in, uint16_t* hi, uint16_t* lo, int64_t n_pix, int ld, int relu, void* stream) {
  XDET_REQUIRE(in && hi && lo, "split: NULL argument");
  return launch_split_f32(in, hi, lo, n_pix, ld, relu, S(stream));
}
int xdet_conv_forward_planes(void* layer, const uint16_t* in_hi, const uint16_t* in_lo, int N, int H, int W,
                             int ld_in, float* out, int ld_out, const float* residual, void* stream) {
  LayerBase* b = static_cast<LayerBase*>(layer);
  XDET_REQUIRE(b && b->kind == 1, "not a conv layer");
  ConvLayer* L = static_cast<ConvLayer*>(b);
  XDET_REQUIRE(L->dma_capable(), "layer was not created in a split-precision mode (or has < 32 input channels)");
  XDET_REQUIRE(in_hi && (in_lo || L->precision == PREC_F16), "conv(planes): NULL planes");
  DeviceGuard guard(L->device);
  return L->forward(nullptr, N, H, W, ld_in, out, ld_out, residual, 0, S(stream), in_hi, in_lo, L->d_zeros);
}
int xdet_split_f32_x8(const float* in, uint16_t* hi, uint16_t* lo8, int64_t n_pix, int ld, int relu, int x8_exp, void* stream) {
  XDET_REQUIRE(in && hi && lo8, "split: NULL argument");
  XDET_REQUIRE(x8_exp > -100 && x8_exp < 100, "split(x8): exponent out of range");
  return launch_split_f32(in, hi, lo8, n_pix, ld, relu, S(stream), 1.f, 1, x8_exp);
}
int xdet_conv_forward_planes_x8(void* layer, const uint16_t* in_hi, const uint16_t* in_lo8, int N, int H, int W, int ld_in,
                                float* out, int ld_out, const float* residual, int x8_exp, void* stream) {
  LayerBase* b = static_cast<LayerBase*>(layer);
  XDET_REQUIRE(b && b->kind == 1, "not a conv layer");
  ConvLayer* L = static_cast<ConvLayer*>(b);
  XDET_REQUIRE(L->dma_capable() && L->d_wt_x8_b, "conv(x8): a pointwise (1x1, stride 1) layer created in the f16x3 mode is needed");
  XDET_REQUIRE(in_hi && in_lo8, "conv(planes): NULL planes");
  XDET_REQUIRE(x8_exp > -100 && x8_exp < 100, "conv(x8): exponent out of range");
  DeviceGuard guard(L->device);
  return L->forward(nullptr, N, H, W, ld_in, out, ld_out, residual, 0, S(stream), in_hi, in_lo8, L->d_zeros, nullptr, nullptr, 0,
                    nullptr, nullptr, 0, 1, x8_exp);
}
int xdet_conv_set_ksplit(void* layer, int ksplit, int mode, int max_parallel_tiles) {
  LayerBase* b = static_cast<LayerBase*>(layer);
  XDET_REQUIRE(b && b->kind == 1, "not a conv layer");
  ConvLayer* L = static_cast<ConvLayer*>(b);
  XDET_REQUIRE(mode >= 0 && mode <= 2 && max_parallel_tiles >= 0, "conv_set_ksplit: mode 0|1|2, max_parallel_tiles >= 0");
  DeviceGuard guard(L->device);
  if (ksplit == 0) { L->ksplit = 0; return XDET_OK; }
  XDET_REQUIRE((L->kh == 1 && L->kw == 1) || (L->kh == 3 && L->kw == 3), "conv_set_ksplit: the split-K kernel runs 1x1 and 3x3 filters");
  XDET_TRY(L->enable_ksplit(ksplit, max_parallel_tiles));
  L->ks_mode = mode;
  return XDET_OK;
}
int xdet_conv_out_shape(void* layer, int H, int W, int* Ho, int* Wo) {
  LayerBase* b = static_cast<LayerBase*>(layer);
  XDET_REQUIRE(b && b->kind == 1, "not a conv layer");
  int a, c;
  static_cast<ConvLayer*>(b)->out_shape(H, W, Ho, Wo, &a, &c);
  return XDET_OK;
}
int xdet_layer_destroy(void* layer) {
  if (!layer) return XDET_OK;
  DeviceGuard guard(static_cast<LayerBase*>(layer)->device);
  delete static_cast<LayerBase*>(layer);
  return XDET_OK;
}
int xdet_depthwise_create(void** layer, int C, int dilation, const float* k) {
  XDET_REQUIRE(layer, "layer is NULL");
  std::unique_ptr<DepthwiseLayer> L(new DepthwiseLayer());
  XDET_TRY(L->init(C, dilation, k));
  *layer = static_cast<LayerBase*>(L.release());
  return XDET_OK;
}
int xdet_depthwise_forward(void* layer, const float* in, int N, int H, int W, int ld, float* out, int relu_in,
                           void* stream) {
  LayerBase* b = static_cast<LayerBase*>(layer);
  XDET_REQUIRE(b && b->kind == 2, "not a depthwise layer");
  DeviceGuard guard(b->device);
  return static_cast<DepthwiseLayer*>(b)->forward(in, N, H, W, ld, out, relu_in, S(stream));
}
int xdet_sepconv_fused_forward(void* dw_layer, void* pw_layer, const float* in, int N, int H, int W, int ld_in, float* out,
                               int ld_out, int relu_in, void* stream) {
  LayerBase* a = static_cast<LayerBase*>(dw_layer);
  LayerBase* b = static_cast<LayerBase*>(pw_layer);
  XDET_REQUIRE(a && a->kind == 2 && b && b->kind == 1, "sepconv_fused: need a depthwise and a conv layer");
  DepthwiseLayer* D = static_cast<DepthwiseLayer*>(a);
  ConvLayer* L = static_cast<ConvLayer*>(b);
  XDET_REQUIRE(L->dma_capable() && L->kh == 1 && L->kw == 1 && L->stride == 1 && L->groups == 1,
               "sepconv_fused: the pointwise layer must be a 1x1 stride-1 conv created in a split-precision mode");
  XDET_REQUIRE(D->ld == ld_in && L->ld_in() == ld_in && L->ld_out() == ld_out && ld_out <= L->cout_pad &&
                   sepconv_fused_supported(ld_in, L->cout_pad, D->dil),
               "sepconv_fused: needs <= 256 input channels (multiple of 32), <= 128 or 129..1024 outputs, dilation 1");
  DeviceGuard guard(L->device);
  return launch_sepconv_fused(in, D->d_w, L->d_wt_hi_b, L->d_wt_lo_b, L->d_scale, L->d_shift, out, N, H, W, ld_in, ld_out,
                              L->cout_pad, relu_in, L->relu_out, S(stream));
}
int xdet_conv3x3_patch_forward(void* layer, const uint16_t* in_hi, const uint16_t* in_lo, int N, int H, int W, float* out,
                               int ld_out, void* stream) {
  LayerBase* b = static_cast<LayerBase*>(layer);
  XDET_REQUIRE(b && b->kind == 1, "not a conv layer");
  ConvLayer* L = static_cast<ConvLayer*>(b);
  XDET_REQUIRE(L->dma_capable() && L->groups == 1 && L->ld_in() == 32 && L->ld_out() == ld_out && L->cout_pad == ld_out &&
                   conv3x3_patch_supported(L->kh, L->kw, L->cin, L->cout_pad, L->stride, L->dil, L->pad_mode),
               "conv3x3_patch: needs a 3x3 / stride 1 / VALID conv over 32 channels with <= 64 outputs, split-precision mode");
  XDET_REQUIRE(in_hi && (in_lo || L->precision == PREC_F16), "conv3x3_patch: NULL planes");
  DeviceGuard guard(L->device);
  return launch_conv3x3_patch(in_hi, L->precision == PREC_F16 ? nullptr : in_lo, L->d_wt_hi_b, L->d_wt_lo_b, L->d_scale,
                              L->d_shift, out, N, H, W, ld_out, L->relu_out, S(stream));
}
int xdet_resnet_bneck_forward(void* conv_a, void* conv_b, void* conv_c, const float* pre_scale, const float* pre_shift,
                              const float* x, int N, int H, int W, float* out, const float* next_scale,
                              const float* next_shift, uint16_t* out_hi, uint16_t* out_lo, void* stream) {
  LayerBase* b[3] = {static_cast<LayerBase*>(conv_a), static_cast<LayerBase*>(conv_b), static_cast<LayerBase*>(conv_c)};
  XDET_REQUIRE(b[0] && b[1] && b[2] && b[0]->kind == 1 && b[1]->kind == 1 && b[2]->kind == 1, "resnet_bneck: three conv layers");
  ConvLayer *A = static_cast<ConvLayer*>(b[0]), *B = static_cast<ConvLayer*>(b[1]), *C = static_cast<ConvLayer*>(b[2]);
  XDET_REQUIRE(A->precision == PREC_F16X3 && B->precision == PREC_F16X3 && C->precision == PREC_F16X3 && A->dma_capable() &&
                   B->dma_capable() && C->dma_capable() && A->groups == 1 && B->groups == 1 && C->groups == 1,
               "resnet_bneck: the three layers must be created in mode 1 (f16x3)");
  XDET_REQUIRE(A->kh == 1 && A->kw == 1 && A->stride == 1 && A->relu_out == 1 && B->kh == 3 && B->kw == 3 && B->stride == 1 &&
                   B->dil == 1 && B->pad_mode == 1 && B->relu_out == 1 && C->kh == 1 && C->kw == 1 && C->stride == 1 &&
                   C->relu_out == 0 && A->cout == B->cin && B->cout == C->cin && B->cin == B->cout && C->cout == A->cin &&
                   A->cout_pad == A->cout && B->cout_pad == B->cout && C->cout_pad == C->cout,
               "resnet_bneck: need 1x1 (ReLU) -> 3x3 SAME stride 1 (ReLU) -> 1x1 with Cin -> Cmid -> Cmid -> Cin channels");
  XDET_REQUIRE(resnet_bneck_supported(A->cin, A->cout, C->cout, H, W, N), "resnet_bneck: unsupported channel counts / tensor size");
  XDET_REQUIRE(pre_scale && pre_shift && x && out && (!out_hi || (out_lo && next_scale && next_shift)), "resnet_bneck: NULL argument");
  BneckLaunch a;
  a.x = x; a.pre_sc = pre_scale; a.pre_sh = pre_shift;
  a.wa_hi = A->d_wt_hi_b; a.wa_lo = A->d_wt_lo_b; a.wb_hi = B->d_wt_hi_b; a.wb_lo = B->d_wt_lo_b; a.wc_hi = C->d_wt_hi_b; a.wc_lo = C->d_wt_lo_b;
  a.sc_a = A->d_scale; a.sh_a = A->d_shift; a.sc_b = B->d_scale; a.sh_b = B->d_shift; a.sc_c = C->d_scale; a.sh_c = C->d_shift;
  a.pl_sc = next_scale; a.pl_sh = next_shift;
  a.out = out; a.out_hi = out_hi; a.out_lo = out_lo;
  a.H = H; a.W = W; a.cin = A->cin; a.cmid = A->cout; a.cout = C->cout;
  DeviceGuard guard(A->device);
  return launch_resnet_bneck(a, N, S(stream));
}
int xdet_sepconv_fused_hpool_forward(void* dw_layer, void* pw_layer, const float* in, int N, int H, int W, int ld_in,
                                     float* out_hpooled, int ld_out, int relu_in, void* stream) {
  LayerBase* a = static_cast<LayerBase*>(dw_layer);
  LayerBase* b = static_cast<LayerBase*>(pw_layer);
  XDET_REQUIRE(a && a->kind == 2 && b && b->kind == 1, "sepconv_fused: need a depthwise and a conv layer");
  DepthwiseLayer* D = static_cast<DepthwiseLayer*>(a);
  ConvLayer* L = static_cast<ConvLayer*>(b);
  XDET_REQUIRE(L->dma_capable() && L->kh == 1 && L->kw == 1 && L->stride == 1 && L->groups == 1,
               "sepconv_fused: the pointwise layer must be a 1x1 stride-1 conv created in a split-precision mode");
  XDET_REQUIRE(D->ld == ld_in && L->ld_in() == ld_in && L->ld_out() == ld_out && ld_out <= L->cout_pad &&
                   sepconv_fused_supported(ld_in, L->cout_pad, D->dil),
               "sepconv_fused: needs <= 256 input channels (multiple of 32), <= 128 or 129..1024 outputs, dilation 1");
  int Wo, pl;
  same_pad(W, 3, 2, 1, &pl, &Wo);
  DeviceGuard guard(L->device);
  return launch_sepconv_fused(in, D->d_w, L->d_wt_hi_b, L->d_wt_lo_b, L->d_scale, L->d_shift, out_hpooled, N, H, W, ld_in,
                              ld_out, L->cout_pad, relu_in, L->relu_out, S(stream), pl);
}
int xdet_maxpool_v3s2_add(const float* in_hpooled, const float* residual, float* out, int N, int H, int Wo, int C, int ld,
                          void* stream) {
  int Ho, pt;
  same_pad(H, 3, 2, 1, &pt, &Ho);
  return launch_maxpool_v3s2_add(in_hpooled, residual, out, N, H, Wo, C, ld, Ho, pt, S(stream));
}
int xdet_maxpool3x3s2_add(const float* in, const float* residual, float* out, int N, int H, int W, int C, int ld,
                          void* stream) {
  int Ho, Wo, pt, pl;
  same_pad(H, 3, 2, 1, &pt, &Ho);
  same_pad(W, 3, 2, 1, &pl, &Wo);
  return launch_maxpool3x3s2_add(in, residual, out, N, H, W, C, ld, Ho, Wo, pt, pl, S(stream));
}
int xdet_preprocess_eval(const uint8_t* image_hwc, int H, int W, float* out_chw, int out_size, void* stream) {
  return launch_preprocess_eval(image_hwc, H, W, out_chw, out_size, S(stream));
}
int xdet_nchw_to_nhwc4(const float* in, float* out, int N, int C, int H, int W, void* stream) {
  return launch_nchw_to_nhwc4(in, out, N, C, H, W, 4, S(stream));
}

int xdet_rpn_decode(const float* rpn_out, int ld, int cls_off, int box_off, int N, int Hh, int Ww, int A,
                    const float* anchors_yx, const float* anchors_hw, float* objectness, float* boxes, void* stream) {
  return launch_rpn_decode(rpn_out, ld, cls_off, box_off, N, Hh, Ww, A, anchors_yx, anchors_hw, objectness, boxes,
                           S(stream));
}
size_t xdet_proposals_workspace_bytes(int N, int n_anchor, int pre_n, int post_n) {
  return proposal_workspace_bytes(N, n_anchor, pre_n, post_n);
}
int xdet_get_proposals(const float* objectness, const float* boxes, int N, int n_anchor, int pre_n, int post_n,
                       float nms_thr, float min_size, void* workspace, float* rois, int* counts_out, void* stream) {
  XDET_REQUIRE(objectness && boxes && workspace && rois, "get_proposals: NULL argument");
  ProposalWorkspace ws;
  proposal_workspace_carve(workspace, N, n_anchor, pre_n, post_n, &ws);
  XDET_TRY(launch_get_proposals(objectness, boxes, N, n_anchor, pre_n, post_n, nms_thr, min_size, ws, rois, S(stream)));
  if (counts_out) XDET_HIP(hipMemcpyAsync(counts_out, ws.counts, (size_t)N * 16, hipMemcpyDeviceToDevice, S(stream)));
  return XDET_OK;
}
int xdet_ext_decode_rois(const float* rois, const float* reg, int ld_reg, int64_t n, float* out, void* stream) {
  return launch_ext_decode_rois(rois, reg, ld_reg, n, out, S(stream));
}
int xdet_bboxes_eval(const float* cls, int ld_cls, const float* boxes, int N, int R, int num_classes,
                     const int* image_shapes, const float* bbox_img, int net_h, int net_w, float select_thr,
                     float nms_thr, int nms_topk, float* det_scores, float* det_boxes, void* stream) {
  XDET_REQUIRE(cls && boxes && image_shapes && bbox_img && det_scores && det_boxes, "bboxes_eval: NULL argument");
  return launch_bboxes_eval(cls, ld_cls, boxes, N, R, num_classes, image_shapes, bbox_img, net_h, net_w, select_thr,
                            nms_thr, nms_topk, det_scores, det_boxes, S(stream));
}

// A handle is a void*: both plan types start with their Plan base, whose kind tag says what the pointer really is
// (handing a resnet handle to a light-head entry point used to be undefined behaviour).
static Plan* plan_of(void* net) { return static_cast<Plan*>(net); }
#define XDET_NET_KIND(net, kind, what)                                                                     \
  XDET_REQUIRE((net) != nullptr && plan_of(net)->plan_kind == (kind), what ": not a handle of this net type")

// ---- light-head net ----
static int set_weight(Plan* p, const char* name, const float* data, int ndim, const int64_t* dims) {
  XDET_REQUIRE(p && name && data && ndim >= 1 && ndim <= 4 && dims, "set_weight: bad arguments");
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) { t.dims.push_back(dims[i]); n *= (size_t)dims[i]; }
  t.v.assign(data, data + n);
  p->w[name] = std::move(t);
  return XDET_OK;
}

int xdet_net_create(void** net, const xdet_lighthead_config* cfg) {
  XDET_REQUIRE(net && cfg, "net/cfg is NULL");
  LightHeadNet* n = new LightHeadNet();
  n->plan_kind = 0;
  n->cfg = *cfg;
  XDET_HIP(hipGetDevice(&n->device));
  *net = n;
  return XDET_OK;
}
int xdet_net_set_weight(void* net, const char* name, const float* data, int ndim, const int64_t* dims) {
  XDET_NET_KIND(net, 0, "net_set_weight");
  return set_weight(static_cast<LightHeadNet*>(net), name, data, ndim, dims);
}
int xdet_net_set_option(void* net, const char* key, const char* value) {
  XDET_NET_KIND(net, 0, "net_set_option");
  LightHeadNet* n = static_cast<LightHeadNet*>(net);
  XDET_REQUIRE(n && key && value, "set_option: NULL argument");
  XDET_REQUIRE(!n->built, "set_option: the net is already built");
  const std::string k(key), v(value);
  if (k == "large_sep") {
    XDET_REQUIRE(v == "auto" || v == "direct" || v == "spectral", "large_sep must be auto | direct | spectral");
    n->large_sep_mode = v == "auto" ? 0 : v == "direct" ? 1 : 2;
    return XDET_OK;
  }
  if (k == "rpn_stream") {
    XDET_REQUIRE(v == "side" || v == "main", "rpn_stream must be side | main");
    n->rpn_side_stream = v == "side";
    return XDET_OK;
  }
  if (k == "pool_sub") {
    XDET_REQUIRE(v == "on" || v == "off", "pool_sub: on | off");
    n->pool_writes_projection_input = v == "on";
    return XDET_OK;
  }
  if (k == "workspace") {
    XDET_REQUIRE(v == "reuse" || v == "ssa" || v == "poison", "workspace: reuse | ssa | poison");
    XDET_REQUIRE(!(v != "ssa" && n->check_range), "workspace=reuse: check_range validates every tensor after the forward and needs workspace=ssa");
    n->reuse_workspace = v != "ssa";
    n->poison_recycled = v == "poison";
    return XDET_OK;
  }
  if (k == "sepconv") {
    XDET_REQUIRE(v == "fused" || v == "split", "sepconv must be fused | split");
    n->fuse_sepconv = v == "fused";
    return XDET_OK;
  }
  if (k == "ksplit") {
    XDET_REQUIRE(v == "on" || v == "off" || v == "all", "ksplit must be on | off | all");
    n->latency_ksplit = v != "off";
    n->rpn_ksplit = v == "all";
    return XDET_OK;
  }
  if (k == "cross") {
    XDET_REQUIRE(v == "f16" || v == "fp8", "cross must be f16 | fp8");
    n->cross8 = v == "fp8";
    return XDET_OK;
  }
  if (k == "check_range") {
    XDET_REQUIRE(v == "on" || v == "off", "check_range must be on | off");
    n->check_range = v == "on";
    if (n->check_range) n->reuse_workspace = false;   // the validation pass reads every tensor after the forward
    return XDET_OK;
  }
  if (k == "pool") {
    XDET_REQUIRE(v == "split" || v == "whole" || v == "split_all", "pool must be split | whole | split_all");
    n->fuse_hpool = v != "whole";
    if (v == "split_all") n->pool_fuse_min_pixels = 0;
    return XDET_OK;
  }
  if (k == "conv3x3") {
    XDET_REQUIRE(v == "patch" || v == "gemm", "conv3x3 must be patch | gemm");
    n->patch_conv3x3 = v == "patch";
    return XDET_OK;
  }
  set_last_error("unknown option: " + k);
  return XDET_ERR_INVALID_ARG;
}
int xdet_net_build(void* net) {
  XDET_NET_KIND(net, 0, "net_build");
  XDET_REQUIRE(net, "net is NULL");
  DeviceGuard guard(static_cast<LightHeadNet*>(net)->device);
  return static_cast<LightHeadNet*>(net)->build();
}
int xdet_net_destroy(void* net) {
  if (!net) return XDET_OK;
  XDET_NET_KIND(net, 0, "net_destroy");
  DeviceGuard guard(static_cast<LightHeadNet*>(net)->device);
  delete static_cast<LightHeadNet*>(net);
  return XDET_OK;
}

int xdet_net_buffer(void* net, const char* name, void** dptr, int64_t dims[4], int* ld) {
  XDET_NET_KIND(net, 0, "net_buffer");
  LightHeadNet* n = static_cast<LightHeadNet*>(net);
  XDET_REQUIRE(n && n->built && name && dptr && dims && ld, "net_buffer: bad arguments");
  const std::string s(name);
  const int B = n->max_batch, R = n->cfg.rpn_post_nms_top_n;
  auto from_buf = [&](const Buf& b) { *dptr = b.p; dims[0] = B; dims[1] = b.H; dims[2] = b.W; dims[3] = b.C; *ld = b.ld; };
  if (s == "mid_x") from_buf(n->mid_x);
  else if (s == "mid") { from_buf(n->mid_x); *dptr = n->mid_relu; }
  else if (s == "out") from_buf(n->out);
  else if (s == "rpn_out") from_buf(n->rpn_out);
  else if (s == "feat") from_buf(n->feat);
  else if (s == "pooled") from_buf(n->pooled);
  else if (s == "fc") from_buf(n->fc);
  else if (s == "cls_reg") from_buf(n->cls_reg);
  else if (s == "objectness") { *dptr = n->objectness; dims[0] = B; dims[1] = n->n_anchor; dims[2] = 1; dims[3] = 1; *ld = 1; }
  else if (s == "rpn_boxes") { *dptr = n->rpn_boxes; dims[0] = B; dims[1] = n->n_anchor; dims[2] = 1; dims[3] = 4; *ld = 4; }
  else if (s == "proposals") { *dptr = n->proposals; dims[0] = B; dims[1] = R; dims[2] = 1; dims[3] = 4; *ld = 4; }
  else if (s == "head_boxes") { *dptr = n->head_boxes; dims[0] = B; dims[1] = R; dims[2] = 1; dims[3] = 4; *ld = 4; }
  else if (s == "prop_counts") { *dptr = n->prop_ws.counts; dims[0] = B; dims[1] = 4; dims[2] = 1; dims[3] = 1; *ld = 1; }
  else if (s == "sorted_boxes") { *dptr = n->prop_ws.sboxes; dims[0] = B; dims[1] = n->cfg.rpn_pre_nms_top_n; dims[2] = 1; dims[3] = 4; *ld = 4; }
  else if (s == "sorted_scores") { *dptr = n->prop_ws.sscores; dims[0] = B; dims[1] = n->cfg.rpn_pre_nms_top_n; dims[2] = 1; dims[3] = 1; *ld = 1; }
  else {
    set_last_error("unknown buffer: " + s);
    return XDET_ERR_INVALID_ARG;
  }
  return XDET_OK;
}

int xdet_net_xception_body(void* net, const float* images, int N, void* stream) {
  XDET_NET_KIND(net, 0, "net_xception_body");
  LightHeadNet* n = static_cast<LightHeadNet*>(net);
  XDET_REQUIRE(n, "net is NULL");
  DeviceGuard guard(n->device);
  XDET_TRY(n->xception_body(images, N, S(stream)));
  // materialise mid_outputs = ReLU(x) for API users (the fused forward applies it on load instead)
  return launch_relu_copy(n->mid_x.p, n->mid_relu, (int64_t)N * n->mid_x.per_image(), S(stream));
}
int xdet_net_get_rpn(void* net, int N, void* stream) {
  XDET_NET_KIND(net, 0, "net_get_rpn");
  LightHeadNet* n = static_cast<LightHeadNet*>(net);
  XDET_REQUIRE(n, "net is NULL");
  DeviceGuard guard(n->device);
  XDET_TRY(n->check(N));
  return n->run_stage(ST_RPN, N, S(stream));
}
int xdet_net_large_sep(void* net, int N, void* stream) {
  XDET_NET_KIND(net, 0, "net_large_sep");
  LightHeadNet* n = static_cast<LightHeadNet*>(net);
  XDET_REQUIRE(n, "net is NULL");
  DeviceGuard guard(n->device);
  XDET_TRY(n->check(N));
  return n->run_stage(ST_LSEP, N, S(stream));
}
int xdet_net_rpn_decode(void* net, int N, void* stream) {
  XDET_NET_KIND(net, 0, "net_rpn_decode");
  XDET_REQUIRE(net, "net is NULL");
  DeviceGuard guard(static_cast<LightHeadNet*>(net)->device);
  return static_cast<LightHeadNet*>(net)->rpn_decode(N, S(stream));
}
int xdet_net_get_proposals(void* net, int N, void* stream) {
  XDET_NET_KIND(net, 0, "net_get_proposals");
  XDET_REQUIRE(net, "net is NULL");
  DeviceGuard guard(static_cast<LightHeadNet*>(net)->device);
  return static_cast<LightHeadNet*>(net)->get_proposals(N, S(stream));
}
int xdet_net_get_head(void* net, int N, void* stream) {
  XDET_NET_KIND(net, 0, "net_get_head");
  XDET_REQUIRE(net, "net is NULL");
  DeviceGuard guard(static_cast<LightHeadNet*>(net)->device);
  return static_cast<LightHeadNet*>(net)->get_head(N, S(stream));
}
int xdet_net_head_decode(void* net, int N, void* stream) {
  XDET_NET_KIND(net, 0, "net_head_decode");
  XDET_REQUIRE(net, "net is NULL");
  DeviceGuard guard(static_cast<LightHeadNet*>(net)->device);
  return static_cast<LightHeadNet*>(net)->head_decode(N, S(stream));
}
int xdet_net_bboxes_eval(void* net, int N, const int* image_shapes, const float* bbox_img, float* det_scores,
                         float* det_boxes, void* stream) {
  XDET_NET_KIND(net, 0, "net_bboxes_eval");
  XDET_REQUIRE(net && det_scores && det_boxes, "bboxes_eval: NULL argument");
  DeviceGuard guard(static_cast<LightHeadNet*>(net)->device);
  return static_cast<LightHeadNet*>(net)->bboxes_eval(N, image_shapes, bbox_img, det_scores, det_boxes, S(stream));
}

int xdet_net_forward(void* net, const float* images, int N, const int* image_shapes, const float* bbox_img,
                     float* det_scores, float* det_boxes, int use_graph, void* stream) {
  XDET_NET_KIND(net, 0, "net_forward");
  LightHeadNet* n = static_cast<LightHeadNet*>(net);
  XDET_REQUIRE(n && images && det_scores && det_boxes, "forward: NULL argument");
  XDET_TRY(n->check(N));
  DeviceGuard guard(n->device);
  hipStream_t s = S(stream);
  if (!use_graph) return n->forward_eager(images, N, image_shapes, bbox_img, det_scores, det_boxes, s);
  XDET_REQUIRE(s != nullptr, "graph replay needs an explicit (non-default) stream");
  // A captured graph bakes in every pointer it was recorded with, so the cache key is the whole
  // argument tuple: a call with another input, shape, bbox or output buffer captures its own graph
  // (double-buffered outputs keep one graph each; the cache is bounded, oldest-first eviction).
  const GraphKey key = {{(uintptr_t)N, (uintptr_t)images, (uintptr_t)image_shapes, (uintptr_t)bbox_img,
                         (uintptr_t)det_scores, (uintptr_t)det_boxes}};
  auto it = n->graphs.find(key);
  if (it == n->graphs.end()) {
    if (n->graphs.size() >= LightHeadNet::kMaxGraphs) {
      const GraphKey old = n->graph_order.front();
      n->graph_order.erase(n->graph_order.begin());
      (void)hipGraphExecDestroy(n->graphs[old]);
      n->graphs.erase(old);
    }
    hipGraph_t g = nullptr;
    XDET_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = n->forward_eager(images, N, image_shapes, bbox_img, det_scores, det_boxes, s);
    const hipError_t e = hipStreamEndCapture(s, &g);
    if (rc != XDET_OK || e != hipSuccess) {
      if (g) (void)hipGraphDestroy(g);
      if (rc != XDET_OK) return rc;
      XDET_HIP(e);
    }
    hipGraphExec_t ge = nullptr;
    const hipError_t ei = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    XDET_HIP(ei);
    n->graphs[key] = ge;
    n->graph_order.push_back(key);
    it = n->graphs.find(key);
  }
  XDET_HIP(hipGraphLaunch(it->second, s));
  return XDET_OK;
}

int xdet_net_calibrate(void* net, const float* images, int N, int* n_scaled, void* stream) {
  XDET_NET_KIND(net, 0, "net_calibrate");
  LightHeadNet* n = static_cast<LightHeadNet*>(net);
  DeviceGuard guard(n->device);
  return n->calibrate(images, N, S(stream), n_scaled);
}
// (either net type: the list lives in the common Plan base)
int xdet_net_plane_scales(void* net, int max_n, int* n_out, int* exps) {
  XDET_REQUIRE(net && n_out && (plan_of(net)->plan_kind == 0 || plan_of(net)->plan_kind == 1), "plane_scales: bad arguments");
  Plan* n = plan_of(net);
  *n_out = (int)n->pscales.size();
  for (int i = 0; exps && i < max_n && i < *n_out; ++i) exps[i] = n->pscales[i].exp;
  return XDET_OK;
}
int xdet_net_x8_planes(void* net, int* n_on) {
  XDET_REQUIRE(net && n_on && (plan_of(net)->plan_kind == 0 || plan_of(net)->plan_kind == 1), "x8_planes: bad arguments");
  int n = 0;
  for (const auto& p : plan_of(net)->pscales) n += p.x8_ok && p.x8_on;
  *n_on = n;
  return XDET_OK;
}
int xdet_net_plane_scale_name(void* net, int idx, char* buf, int buflen) {
  XDET_REQUIRE(net && (plan_of(net)->plan_kind == 0 || plan_of(net)->plan_kind == 1), "plane_scale_name: bad arguments");
  Plan* n = plan_of(net);
  XDET_REQUIRE(buf && buflen > 0 && idx >= 0 && idx < (int)n->pscales.size(), "plane_scale_name: bad arguments");
  snprintf(buf, buflen, "%s", n->pscales[idx].name.c_str());
  return XDET_OK;
}

int xdet_net_graph_count(void* net, int* count) {
  XDET_NET_KIND(net, 0, "net_graph_count");
  LightHeadNet* n = static_cast<LightHeadNet*>(net);
  XDET_REQUIRE(n && count, "graph_count: NULL argument");
  *count = (int)n->graphs.size();
  return XDET_OK;
}

int xdet_net_memory(void* net, size_t* allocated_bytes, size_t* recycled_bytes) {
  XDET_REQUIRE(net && allocated_bytes && recycled_bytes && (plan_of(net)->plan_kind == 0 || plan_of(net)->plan_kind == 1), "net_memory: bad arguments");
  *allocated_bytes = plan_of(net)->allocated_bytes;
  *recycled_bytes = plan_of(net)->ws_recycled_bytes;
  return XDET_OK;
}
int xdet_net_flops_per_image(void* net, double* backbone, double* rpn, double* large_sep, double* head) {
  XDET_NET_KIND(net, 0, "net_flops_per_image");
  LightHeadNet* n = static_cast<LightHeadNet*>(net);
  XDET_REQUIRE(n && n->built, "net not built");
  double f[5] = {0, 0, 0, 0, 0};
  for (const Op& op : n->ops) f[op.stage] += std::max(op.flops, 0.0);
  if (backbone) *backbone = f[ST_BODY] + f[ST_EXIT];
  if (rpn) *rpn = f[ST_RPN];
  if (large_sep) *large_sep = f[ST_LSEP];
  if (head) *head = f[ST_HEAD];
  return XDET_OK;
}

// mode 0 = light-head net, 1 = resnet trunk
static Plan* as_plan(void* net, int kind) {
  return kind == 0 ? static_cast<Plan*>(static_cast<LightHeadNet*>(net)) : static_cast<Plan*>(static_cast<ResNetTrunk*>(net));
}
int xdet_profile_enable(void* net, int kind, int enable) {
  XDET_REQUIRE(net, "net is NULL");
  as_plan(net, kind)->profiling = enable != 0;
  return XDET_OK;
}
int xdet_profile_read(void* net, int kind, int max_ops, int* n_ops, double* ms, int* launches, double* flops) {
  XDET_REQUIRE(net && n_ops && ms && launches && flops, "profile_read: NULL argument");
  return as_plan(net, kind)->profile_read(max_ops, n_ops, ms, launches, flops);
}
int xdet_profile_mfma_flops(void* net, int kind, int max_ops, int* n_ops, double* issued) {
  XDET_REQUIRE(net && n_ops && issued, "profile_mfma_flops: NULL argument");
  Plan* p = as_plan(net, kind);
  *n_ops = (int)std::min<size_t>(p->ops.size(), (size_t)max_ops);
  const double per_term = g_default_precision == PREC_F16X3 ? 3.0 : 1.0;
  for (int i = 0; i < *n_ops; ++i) {
    const Op& op = p->ops[i];
    issued[i] = op.mfma_flops >= 0.0 ? op.mfma_flops : per_term * std::max(op.flops, 0.0);
  }
  return XDET_OK;
}
int xdet_profile_op_name(void* net, int kind, int op, char* buf, int buflen) {
  XDET_REQUIRE(net && buf && buflen > 0, "profile_op_name: bad arguments");
  Plan* p = as_plan(net, kind);
  XDET_REQUIRE(op >= 0 && op < (int)p->ops.size(), "profile_op_name: op out of range");
  snprintf(buf, buflen, "%s", p->ops[op].name.c_str());
  return XDET_OK;
}

// ---- resnet trunk ----
int xdet_resnet_create(void** net, int image_size, int max_batch) {
  XDET_REQUIRE(net && image_size >= 64 && max_batch > 0, "resnet_create: bad arguments");
  ResNetTrunk* r = new ResNetTrunk();
  r->plan_kind = 1;
  r->image_size = image_size;
  r->max_batch = max_batch;
  XDET_HIP(hipGetDevice(&r->device));
  *net = r;
  return XDET_OK;
}
int xdet_resnet_set_weight(void* net, const char* name, const float* data, int ndim, const int64_t* dims) {
  XDET_NET_KIND(net, 1, "resnet_set_weight");
  return set_weight(static_cast<ResNetTrunk*>(net), name, data, ndim, dims);
}
int xdet_resnet_set_option(void* net, const char* key, const char* value) {
  XDET_NET_KIND(net, 1, "resnet_set_option");
  ResNetTrunk* r = static_cast<ResNetTrunk*>(net);
  XDET_REQUIRE(r && key && value, "resnet_set_option: NULL argument");
  XDET_REQUIRE(!r->built, "resnet_set_option: the trunk is already built");
  const std::string k(key), v(value);
  XDET_REQUIRE(v == "on" || v == "off", "resnet_set_option: the value must be on | off");
  const bool on = v == "on";
  if (k == "ksplit") r->ksplit_enabled = on;
  else if (k == "stem7") r->stem7_enabled = on;
  else if (k == "stem_pool") r->stem_pool_bn = on;
  else if (k == "bneck") r->bneck_enabled = on;
  else if (k == "projcat") r->projcat_enabled = on;
  else if (k == "preconv") r->preconv_enabled = on;
  else {
    set_last_error("unknown option: " + k);
    return XDET_ERR_INVALID_ARG;
  }
  return XDET_OK;
}
int xdet_resnet_build(void* net) {
  XDET_NET_KIND(net, 1, "resnet_build");
  XDET_REQUIRE(net, "net is NULL");
  DeviceGuard guard(static_cast<ResNetTrunk*>(net)->device);
  return static_cast<ResNetTrunk*>(net)->build();
}
int xdet_resnet_forward(void* net, const float* images, int N, float* out_nhwc, void* stream) {
  XDET_NET_KIND(net, 1, "resnet_forward");
  ResNetTrunk* r = static_cast<ResNetTrunk*>(net);
  XDET_REQUIRE(r && r->built && images, "resnet_forward: bad arguments");
  XDET_REQUIRE(N > 0 && N <= r->max_batch, "batch must be in 1..max_batch");
  DeviceGuard guard(r->device);
  hipStream_t s = S(stream);
  r->cur_images = images;
  r->bneck_fused_now = r->bneck_all_ok();
  if (!r->stem7_direct) XDET_TRY(launch_nchw_to_nhwc4(images, r->in4.p, N, 3, r->image_size, r->image_size, 4, s));
  XDET_TRY(r->run_stage(0, N, s));
  if (out_nhwc)
    XDET_HIP(hipMemcpyAsync(out_nhwc, r->outb.p, (size_t)N * r->outb.per_image() * 4, hipMemcpyDeviceToDevice, s));
  return XDET_OK;
}
int xdet_resnet_forward_graph(void* net, const float* images, int N, float* out_nhwc, void* stream) {
  XDET_NET_KIND(net, 1, "resnet_forward_graph");
  ResNetTrunk* r = static_cast<ResNetTrunk*>(net);
  XDET_REQUIRE(r && r->built && images, "resnet_forward: bad arguments");
  XDET_REQUIRE(N > 0 && N <= r->max_batch, "batch must be in 1..max_batch");
  if (r->profiling) return xdet_resnet_forward(net, images, N, out_nhwc, stream);   // event pairs cannot be replayed
  hipStream_t s = S(stream);
  XDET_REQUIRE(s != nullptr, "graph replay needs an explicit (non-default) stream");
  DeviceGuard guard(r->device);
  const ResNetTrunk::Key key = {{(uintptr_t)N, (uintptr_t)images, (uintptr_t)out_nhwc, (uintptr_t)r->bneck_all_ok()}};
  auto it = r->graphs.find(key);
  if (it == r->graphs.end()) {
    if (r->graphs.size() >= 8) {
      const ResNetTrunk::Key old = r->graph_order.front();
      r->graph_order.erase(r->graph_order.begin());
      (void)hipGraphExecDestroy(r->graphs[old]);
      r->graphs.erase(old);
    }
    hipGraph_t g = nullptr;
    XDET_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = xdet_resnet_forward(net, images, N, out_nhwc, stream);
    const hipError_t e = hipStreamEndCapture(s, &g);
    if (rc != XDET_OK || e != hipSuccess) {
      if (g) (void)hipGraphDestroy(g);
      if (rc != XDET_OK) return rc;
      XDET_HIP(e);
    }
    hipGraphExec_t ge = nullptr;
    const hipError_t ei = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    XDET_HIP(ei);
    r->graphs[key] = ge;
    r->graph_order.push_back(key);
    it = r->graphs.find(key);
  }
  XDET_HIP(hipGraphLaunch(it->second, s));
  return XDET_OK;
}
// activation pre-scale of the trunk's split-precision operands (as xdet_net_calibrate): pre-activation planes
// (bn_relu pass / the previous block's epilogue, whose folded BN carries 2^-e), the inner convs' planes, the strided
// projections' subsample pass
int xdet_resnet_calibrate(void* net, const float* images, int N, int* n_scaled, void* stream) {
  XDET_NET_KIND(net, 1, "resnet_calibrate");
  ResNetTrunk* r = static_cast<ResNetTrunk*>(net);
  XDET_REQUIRE(r->built && images && N > 0 && N <= r->max_batch, "resnet_calibrate: bad arguments");
  DeviceGuard guard(r->device);
  if (n_scaled) *n_scaled = 0;
  if (r->net_precision == PREC_F32) return XDET_OK;
  for (auto& g : r->graphs) (void)hipGraphExecDestroy(g.second);      // graphs bake kernel arguments
  r->graphs.clear();
  r->graph_order.clear();
  return r->calibrate_planes(N, S(stream), n_scaled, [&](hipStream_t st) { return xdet_resnet_forward(net, images, N, nullptr, st); });
}
int xdet_resnet_out_shape(void* net, int* Ho, int* Wo, int* C) {
  XDET_NET_KIND(net, 1, "resnet_out_shape");
  ResNetTrunk* r = static_cast<ResNetTrunk*>(net);
  XDET_REQUIRE(r && r->built, "resnet not built");
  *Ho = r->outb.H; *Wo = r->outb.W; *C = r->outb.C;
  return XDET_OK;
}
int xdet_resnet_flops_per_image(void* net, double* flops) {
  XDET_NET_KIND(net, 1, "resnet_flops_per_image");
  ResNetTrunk* r = static_cast<ResNetTrunk*>(net);
  XDET_REQUIRE(r && r->built && flops, "resnet not built");
  *flops = r->flops;
  return XDET_OK;
}
int xdet_resnet_destroy(void* net) {
  if (!net) return XDET_OK;
  XDET_NET_KIND(net, 1, "resnet_destroy");
  DeviceGuard guard(static_cast<ResNetTrunk*>(net)->device);
  delete static_cast<ResNetTrunk*>(net);
  return XDET_OK;
}

}  // extern "C"
