// Micro-benchmark: sustained v_mfma_f32_32x32x16_f16 rate of the whole chip (power/clock included),
// register operands only, and the same loop fed by ds_read_b128 from LDS at the conv kernel's ratio
// (24 reads per 48 MFMAs per wave).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_regs(float* sink, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
  for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(threadIdx.x * 0.001f + r); b[r] = (_Float16)(r * 0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 1234.5f) sink[0] = s;
}

// TM x TN register tile fed from LDS each step, like one ks of the conv kernel (hi/lo planes, 3 products)
template <int TM, int TN>
__global__ __launch_bounds__(512) void mfma_lds(float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (unsigned short)(0x3c00 + (i & 7));
  __syncthreads();
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = (lane & 31) * 32 + (lane >> 5) * 8;
  for (int it = 0; it < iters; ++it) {
    f16x8 ah[TM], al[TM], bh[TN], bl[TN];
    const int o = base + ((it + wave) & 7) * 1024;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[i] = *reinterpret_cast<const f16x8*>(lds + ((o + i * 1024) & 32767));
      al[i] = *reinterpret_cast<const f16x8*>(lds + ((o + i * 1024 + 16) & 32767));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[j] = *reinterpret_cast<const f16x8*>(lds + ((o + 8192 + j * 1024) & 32767));
      bl[j] = *reinterpret_cast<const f16x8*>(lds + ((o + 8192 + j * 1024 + 16) & 32767));
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 1234.5f) sink[0] = s;
}

template <typename F>
static void run(const char* name, F launch, int iters, double flop_per_iter_per_block, int blocks) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch(100);
  hipDeviceSynchronize();
  hipEventRecord(a);
  launch(iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-44s %8.3f ms  %8.1f TFLOP/s\n", name, ms, flop_per_iter_per_block * iters * blocks / ms / 1e9);
}

int main() {
  float* sink; hipMalloc(&sink, 64);
  const double mf = 2.0 * 32 * 32 * 16;
  for (int rep = 0; rep < 2; ++rep) {
    run("regs 8 waves/CU, 4 acc", [&](int it) { hipLaunchKernelGGL((mfma_regs<4>), dim3(256), dim3(512), 0, 0, sink, it); }, 20000, mf * 4 * 8, 256);
    run("regs 8 waves/CU, 8 acc", [&](int it) { hipLaunchKernelGGL((mfma_regs<8>), dim3(256), dim3(512), 0, 0, sink, it); }, 20000, mf * 8 * 8, 256);
    run("regs 4 waves/CU, 8 acc", [&](int it) { hipLaunchKernelGGL((mfma_regs<8>), dim3(256), dim3(256), 0, 0, sink, it); }, 20000, mf * 8 * 4, 256);
    run("regs 8 waves/CU, 8 acc, long (thermal)", [&](int it) { hipLaunchKernelGGL((mfma_regs<8>), dim3(256), dim3(512), 0, 0, sink, it); }, 400000, mf * 8 * 8, 256);
    run("lds-fed 8 waves/CU 4x2 (conv 256x256 shape)", [&](int it) { hipLaunchKernelGGL((mfma_lds<4, 2>), dim3(256), dim3(512), 65536, 0, sink, it); }, 20000, mf * 24 * 8, 256);
    run("lds-fed 4 waves/CU 4x4", [&](int it) { hipLaunchKernelGGL((mfma_lds<4, 4>), dim3(256), dim3(256), 65536, 0, sink, it); }, 20000, mf * 48 * 4, 256);
    run("lds-fed 8 waves/CU 2x2 (128x128 x2 blocks)", [&](int it) { hipLaunchKernelGGL((mfma_lds<2, 2>), dim3(256), dim3(512), 65536, 0, sink, it); }, 20000, mf * 12 * 8, 256);
  }
  return 0;
}
