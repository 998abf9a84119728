// What exactly does v_mfma_scale_f32_32x32x64_f8f6f4 compute?  One wave, fp8 (e4m3, OCP) operands, checked against the host:
//   * A operand: lane l holds row (l & 31), K elements 32*(l >> 5) .. +31, byte k in its 8 VGPRs (little endian);
//   * B operand: lane l holds column (l & 31), the same K split;
//   * scale operands: E8M0 byte (value 2^(byte - 127)) per lane = per (row, 32-deep K block); opsel picks the byte of the VGPR;
//   * C/D: the 32x32 f32 layout of the other 32x32 MFMAs (col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)).
// This is the instruction the fp8 cross terms of profiles/NOTES_r04.md would use (a_hi8 | a_lo8 along K, one scale per half).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_scale_semantics.hip -o /tmp/mss && /tmp/mss
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

static float e4m3_to_float(unsigned char b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 0) v = ldexpf((float)m, -9);                    // subnormal: m * 2^-3 * 2^-6
  else if (e == 15 && m == 7) v = NAN;
  else v = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -v : v;
}

__global__ void k(const unsigned char* A, const unsigned char* B, const int* sa, const int* sb, float* D, int opsel) {
  const int l = threadIdx.x;
  v8i a, b;
  for (int r = 0; r < 8; ++r) {
    a[r] = reinterpret_cast<const int*>(A + ((l & 31) * 64 + (l >> 5) * 32))[r];
    b[r] = reinterpret_cast<const int*>(B + ((l & 31) * 64 + (l >> 5) * 32))[r];
  }
  v16f c = {};
  if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[l], 0, sb[l]);
  else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, sa[l], 1, sb[l]);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

int main() {
  std::vector<unsigned char> A(32 * 64), B(32 * 64);
  srand(3);
  for (auto& x : A) { x = rand() & 0xff; if ((x & 0x7f) == 0x7f) x ^= 1; }     // no NaN
  for (auto& x : B) { x = rand() & 0xff; if ((x & 0x7f) == 0x7f) x ^= 1; }
  std::vector<int> sa(64), sb(64);
  for (int l = 0; l < 64; ++l) {
    const int ea = 127 + (l >> 5 ? -11 : 0) + (l & 3), eb = 127 + (l >> 5 ? 2 : -9) - (l & 1);   // different per row and per K block
    sa[l] = ea | ((ea + 5) << 8);                          // byte 0 for opsel 0, byte 1 for opsel 1
    sb[l] = eb | ((eb - 3) << 8);
  }
  unsigned char *dA, *dB;
  int *dsa, *dsb;
  float* dD;
  (void)hipMalloc(&dA, A.size()); (void)hipMalloc(&dB, B.size()); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256); (void)hipMalloc(&dD, 4096);
  (void)hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  (void)hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
  for (int opsel = 0; opsel < 2; ++opsel) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD, opsel);
    std::vector<float> D(1024);
    (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int kb = 0; kb < 2; ++kb) {
          double part = 0;
          for (int kk = 0; kk < 32; ++kk) part += (double)e4m3_to_float(A[i * 64 + kb * 32 + kk]) * e4m3_to_float(B[j * 64 + kb * 32 + kk]);
          const int ba = (sa[kb * 32 + i] >> (8 * opsel)) & 0xff, bb = (sb[kb * 32 + j] >> (8 * opsel)) & 0xff;
          ref += ldexp(part, (ba - 127) + (bb - 127));
        }
        worst = fmax(worst, fabs(ref - D[i * 32 + j]));
        scale = fmax(scale, fabs(ref));
      }
    // (not bit-exact: the 64-term sums are not formed exactly inside the instruction -- ~1e-4 of the largest result, which is
    // nothing for a cross term that is 2^-11 of its GEMM; a wrong lane / scale model would be off by O(1))
    printf("opsel %d: max |device - model| %.3g of max |model| %.3g (%.1e relative) -> %s\n", opsel, worst, scale, worst / scale,
           worst <= 1e-3 * scale ? "the model holds" : "MISMATCH");
  }
  return 0;
}
