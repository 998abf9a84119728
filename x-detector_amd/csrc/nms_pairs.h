// Pair arithmetic of the two greedy NMS stages (proposals.hip: tf.image.non_max_suppression over the RPN candidates,
// net/xception_body.py:57-67; detect.hip: per class, utility/eval_helper.py:449-506): one lane holds a candidate, a wave
// walks up to 64 reference boxes broadcast from LDS and collects "IoU(ref[i], me) > thr" as bit i.
// Compiled into translation units built with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace xdet {

typedef unsigned long long nms_u64;

// bit i of the result: IoU(ref[i], me) > thr.  Boxes are normalised (y0 <= y1, x0 <= x1) with their areas beside them; an
// area <= 0 is stored as +inf, which makes every IoU with that box compare false below exactly as the reference's
// "either area <= 0 -> 0" does, at no cost per pair.
// Two product tests decide every pair outside a 1e-5 relative band around the threshold (~16 VALU operations per pair, no
// division); a lane that met a pair inside the band (or a NaN from thr = 0) redoes its 64 pairs with the reference's own
// expression: separately rounded product, difference and quotient.
static __device__ __noinline__ nms_u64 nms_pair_bits_exact(const float4* ref, const float* ref_area, int cnt, const float4 me,
                                                const float ma, const float thr) {
  nms_u64 bits = 0ull;
  for (int i = 0; i < cnt; ++i) {
    const float4 r = ref[i];
    const float ih = fmaxf(fminf(r.z, me.z) - fmaxf(r.x, me.x), 0.f);
    const float iw = fmaxf(fminf(r.w, me.w) - fmaxf(r.y, me.y), 0.f);
    const float inter = __fmul_rn(ih, iw);
    if (__fdiv_rn(inter, __fsub_rn(__fadd_rn(ref_area[i], ma), inter)) > thr) bits |= 1ull << i;
  }
  return bits;
}
// v_min_f32 / v_max_f32 as they are: fminf / fmaxf put a canonicalising v_max_f32 x, x, x in front of every operand (IEEE
// minNum of a signalling NaN) -- eight more VALU operations per pair for inputs that are finite by construction
static __device__ __forceinline__ float vmin(float a, float b) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
static __device__ __forceinline__ float vmax(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
static __device__ __forceinline__ nms_u64 nms_pair_bits(const float4* __restrict__ ref, const float* __restrict__ ref_area, int cnt,
                                             const float4 me, const float ma, const float thr, const float thr_hi,
                                             const float thr_lo) {
  unsigned lo = 0u, hi = 0u;
  bool band = false;
  auto one = [&](int i) -> bool {
    const float4 r = ref[i];
    const float ih = vmax(vmin(r.z, me.z) - vmax(r.x, me.x), 0.f);
    const float iw = vmin(r.w, me.w) - vmax(r.y, me.y);
    const float inter = ih * iw;                      // <= 0 unless the boxes overlap
    const float uni = (ref_area[i] + ma) - inter;
    const bool h = fmaf(-thr_hi, uni, inter) > 0.f;
    band |= !h && !(fmaf(-thr_lo, uni, inter) < 0.f);
    return h;
  };
  const int c0 = min(cnt, 32);
#pragma unroll 4
  for (int i = 0; i < c0; ++i) lo |= one(i) ? (1u << i) : 0u;
#pragma unroll 4
  for (int i = 32; i < cnt; ++i) hi |= one(i) ? (1u << (i - 32)) : 0u;
  nms_u64 bits = ((nms_u64)hi << 32) | lo;
  if (band) bits = nms_pair_bits_exact(ref, ref_area, cnt, me, ma, thr);
  return bits;
}

}  // namespace xdet
