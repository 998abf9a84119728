/*
 * oracle/psroialign_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's PsRoiAlign forward
 * (cpp/PSROIPooling/ps_roi_align_op.cc:94-192, same expression order as the CUDA
 * kernel ps_roi_align_op.cu:36-132).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the shipped path never does.
 *
 * Pinning: checked against the known-answer vectors SURVEY.md 8c records for the
 * reference's own test inputs (cpp/PSROIPooling/test_op.py:52-81) in
 * tests/test_oracle_psroialign.py.  The reference op itself needs the TensorFlow
 * headers, which this image lacks, so it is unbuildable here (no oracle/_ref).
 *
 * Exactness: the reference is built by g++ -O2 for baseline x86-64 (no FMA), so every
 * a*b+c is two roundings; the bilinear blend is evaluated in double because of the
 * `1.` literals (ps_roi_align_op.cc:171-174).  Build this file with -ffp-contract=off.
 */
#include <float.h>
#include <stdint.h>

/* layout: 0 = NCHW (the op's contract), 1 = NHWC (the fused pipeline's feature map) */
static inline float feat_at(const float *inputs, int layout, int C, int H, int W,
                            int n, int c, int y, int x) {
  if (layout == 0) return inputs[(((int64_t)n * C + c) * H + y) * W + x];
  return inputs[(((int64_t)n * H + y) * W + x) * C + c];
}

/* ldc: channel stride of an NHWC map whose channel dimension is padded (>= C). */
int oracle_psroialign_fwd(const float *inputs, const float *rois, float *pooled, int32_t *index,
                          int N, int C, int H, int W, int R, int grid_w, int grid_h,
                          int use_max, int layout, int ldc) {
  if (grid_w <= 0 || grid_h <= 0) return -1;
  const int grid_size = grid_w * grid_h;
  const int bank = C / grid_size;
  if (bank * grid_size != C) return -2;
  if (layout == 1 && ldc < C) return -3;
  const int Cs = layout == 1 ? ldc : C;
  const int64_t total = (int64_t)N * R * C;
  for (int64_t w = 0; w < total; ++w) {
    const int pos = (int)((w % C) / bank);
    const int row = pos / grid_w;
    const int col = pos % grid_w;
    const int ch = (int)(w % bank);
    const int64_t pool_index = w / C;
    const int n = (int)(pool_index / R);
    const int r = (int)(pool_index % R);
    const float *roi = rois + ((int64_t)n * R + r) * 4;
    const int c_in = pos * bank + ch;
    if (roi[2] < FLT_MIN || roi[3] < FLT_MIN) {
      pooled[w] = 0.f;
      index[w] = 0; /* reference leaves it unwritten (ps_roi_align_op.cc:123-126) */
      continue;
    }
    float yc = (float)(roi[0] * H);
    float xc = (float)(roi[1] * W);
    float rh = roi[2] * H; if (rh < 1.f) rh = 1.f;
    float rw = roi[3] * W; if (rw < 1.f) rw = 1.f;
    float ymin = yc - (float)(rh / 2.); if (ymin < 0.f) ymin = 0.f;
    float xmin = xc - (float)(rw / 2.); if (xmin < 0.f) xmin = 0.f;
    float hmax = (float)H - FLT_MIN, wmax = (float)W - FLT_MIN;
    float ymax = yc + (float)(rh / 2.); if (ymax > hmax) ymax = hmax;
    float xmax = xc + (float)(rw / 2.); if (xmax > wmax) xmax = wmax;
    float roi_h = ymax - ymin, roi_w = xmax - xmin;
    float bin_w = roi_w / grid_w;
    float bin_h = roi_h / grid_h;
    int n_w = (int)bin_w + 1;
    int n_h = (int)bin_h + 1;
    float step_w = bin_w / n_w;
    float step_h = bin_h / n_h;
    float x0 = xmin + bin_w * col;
    float y0 = ymin + bin_h * row;
    int arg = 0;
    float acc = use_max ? -FLT_MAX : 0.f;
    for (int i = 0; i < n_h; ++i) {
      for (int j = 0; j < n_w; ++j) {
        float x = (float)((double)(x0 + step_w * j) + (double)step_w / 2.);
        float y = (float)((double)(y0 + step_h * i) + (double)step_h / 2.);
        int ix = (int)x, iy = (int)y;
        float fx = x - ix, fy = y - iy;
        int iy1 = iy + 1 < H - 1 ? iy + 1 : H - 1;
        int ix1 = ix + 1 < W - 1 ? ix + 1 : W - 1;
        double v = (1. - fx) * (1. - fy) * feat_at(inputs, layout, Cs, H, W, n, c_in, iy, ix) +
                   (1. - fx) * fy * feat_at(inputs, layout, Cs, H, W, n, c_in, iy1, ix) +
                   fx * (1. - fy) * feat_at(inputs, layout, Cs, H, W, n, c_in, iy, ix1) +
                   fx * fy * feat_at(inputs, layout, Cs, H, W, n, c_in, iy1, ix1);
        float t = (float)v;
        if (use_max) {
          if (acc < t) { acc = t; arg = n_w * i + j; }
        } else {
          acc += t;
        }
      }
    }
    if (!use_max) acc /= (float)(n_h * n_w);
    pooled[w] = acc;
    index[w] = use_max ? arg : 0;
  }
  return 0;
}

/*
 * PsRoiAlignGrad: restatement of the reference's backward
 * (cpp/PSROIPooling/ps_roi_align_grad_op.cu:36-139 scatter formulation; the CPU op
 * ps_roi_align_grad_op.cc:186-322 computes the same sums gather-style).  grad_out is zero-filled,
 * then every pooled element scatters grad * bilinear weight to its 4 neighbours: for 'max' only
 * the argmax sample (pooled_index), for 'mean' every sample with grad / (n_h*n_w).  Weights are
 * evaluated in double and cast to float as in the reference.  Sequential accumulation order here;
 * the GPU uses atomics, so parity is to rounding, not bit-exact.  parity unpinned (no reference
 * vectors exist); pinned mathematically by tests/test_oracle_psroialign.py (adjoint identity).
 */
int oracle_psroialign_grad(const float *rois, const float *grad_pooled, const int32_t *pooled_index,
                           float *grad_out, int N, int C, int H, int W, int R, int grid_w, int grid_h,
                           int use_max, int layout, int ldc) {
  if (grid_w <= 0 || grid_h <= 0) return -1;
  const int grid_size = grid_w * grid_h;
  const int bank = C / grid_size;
  if (bank * grid_size != C) return -2;
  const int Cs = layout == 1 ? ldc : C;
  const int64_t total_out = (int64_t)N * Cs * H * W;
  for (int64_t i = 0; i < total_out; ++i) grad_out[i] = 0.f;
  const int64_t total = (int64_t)N * R * C;
  for (int64_t w = 0; w < total; ++w) {
    const int pos = (int)((w % C) / bank);
    const int row = pos / grid_w, col = pos % grid_w;
    const int64_t pool_index = w / C;
    const int n = (int)(pool_index / R), r = (int)(pool_index % R);
    const float *roi = rois + ((int64_t)n * R + r) * 4;
    const int c_in = (int)(w % C);
    if (roi[2] < FLT_MIN || roi[3] < FLT_MIN) continue;
    float yc = (float)(roi[0] * H), xc = (float)(roi[1] * W);
    float rh = roi[2] * H; if (rh < 1.f) rh = 1.f;
    float rw = roi[3] * W; if (rw < 1.f) rw = 1.f;
    float ymin = yc - (float)(rh / 2.); if (ymin < 0.f) ymin = 0.f;
    float xmin = xc - (float)(rw / 2.); if (xmin < 0.f) xmin = 0.f;
    float hmax = (float)H - FLT_MIN, wmax = (float)W - FLT_MIN;
    float ymax = yc + (float)(rh / 2.); if (ymax > hmax) ymax = hmax;
    float xmax = xc + (float)(rw / 2.); if (xmax > wmax) xmax = wmax;
    float bin_w = (xmax - xmin) / grid_w, bin_h = (ymax - ymin) / grid_h;
    int n_w = (int)bin_w + 1, n_h = (int)bin_h + 1;
    float step_w = bin_w / n_w, step_h = bin_h / n_h;
    float x0 = xmin + bin_w * col, y0 = ymin + bin_h * row;
    const int i_lo = use_max ? pooled_index[w] / n_w : 0, i_hi = use_max ? i_lo + 1 : n_h;
    const int j_lo = use_max ? pooled_index[w] % n_w : 0, j_hi = use_max ? j_lo + 1 : n_w;
    const float g = use_max ? grad_pooled[w] : grad_pooled[w] / (float)(n_w * n_h);
    for (int i = i_lo; i < i_hi; ++i)
      for (int j = j_lo; j < j_hi; ++j) {
        float x = (float)((double)(x0 + step_w * j) + (double)step_w / 2.);
        float y = (float)((double)(y0 + step_h * i) + (double)step_h / 2.);
        int ix = (int)x, iy = (int)y;
        float fx = x - ix, fy = y - iy;
        int iy1 = iy + 1 < H - 1 ? iy + 1 : H - 1, ix1 = ix + 1 < W - 1 ? ix + 1 : W - 1;
#define GOUT(yy, xx) grad_out[layout == 0 ? ((((int64_t)n * C + c_in) * H + (yy)) * W + (xx)) \
                                          : ((((int64_t)n * H + (yy)) * W + (xx)) * Cs + c_in)]
        GOUT(iy, ix) += (float)((1. - fx) * (1. - fy) * g);
        GOUT(iy1, ix) += (float)((1. - fx) * fy * g);
        GOUT(iy, ix1) += (float)(fx * (1. - fy) * g);
        GOUT(iy1, ix1) += (float)(fx * fy * g);
#undef GOUT
      }
  }
  return 0;
}
