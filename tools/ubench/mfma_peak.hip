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

#define GLDS16(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g), (__attribute__((address_space(3))) void*)(l), 16, 0, 0)
// The conv kernel's K step in miniature (256x256 tile, 8 waves as 2x4, f16x3, two LDS stages):
//   [12 ds_read (half 1), 24 MFMA (half 0), vmcnt(0)+barrier, NLOAD x 1 KB operand ingest, 12 ds_read, 24 MFMA]
// LDSREAD: fragments come from LDS with the conv kernel's conflict-free swizzle (else: registers only)
// PATH 0: ingest by LDS-DMA (global_load_lds_dwordx4); PATH 1: global_load_dwordx4 -> VGPR -> ds_write_b128
template <int NLOAD, bool MFMA_ON, bool LDSREAD, int PATH, int SPREAD = 0>
__global__ __launch_bounds__(512) void conv_step(const char* __restrict__ src, size_t span, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  for (int i = threadIdx.x; i < 65536; i += blockDim.x) lds[i] = (unsigned short)(0x3c00 + (i & 7));
  __syncthreads();
  constexpr int TM = 4, TN = 2;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3, frow = lane & 31, fh = lane >> 5;
  size_t goff = ((size_t)blockIdx.x * 8 + wave) * 8192 % span;
  f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
  for (int s2 = 0; s2 < 2; ++s2) {
    for (int i = 0; i < TM; ++i) for (int r = 0; r < 8; ++r) { ah[s2][i][r] = (_Float16)(lane + r + i); al[s2][i][r] = (_Float16)(r * 0.25f); }
    for (int j = 0; j < TN; ++j) for (int r = 0; r < 8; ++r) { bh[s2][j][r] = (_Float16)(lane - r + j); bl[s2][j][r] = (_Float16)(r * 0.125f); }
  }
  auto load = [&](int set, int stage, int ks) {
    if (!LDSREAD) return;
    const unsigned short* S = lds + stage * 32768;
    const int c = ks * 2 + fh;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rt = wm * 128 + i * 32 + frow;
      const int o = rt * 32 + ((c ^ ((rt >> 2) & 3)) << 3);
      ah[set][i] = *reinterpret_cast<const f16x8*>(S + o);
      al[set][i] = *reinterpret_cast<const f16x8*>(S + 8192 + o);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int rt = wn * 64 + j * 32 + frow;
      const int o = rt * 32 + ((c ^ ((rt >> 2) & 3)) << 3);
      bh[set][j] = *reinterpret_cast<const f16x8*>(S + 16384 + o);
      bl[set][j] = *reinterpret_cast<const f16x8*>(S + 24576 + o);
    }
  };
  auto mma = [&](int set) {
    if (!MFMA_ON) {
#pragma unroll
      for (int i = 0; i < TM; ++i) { asm volatile("" ::"v"(ah[set][i])); asm volatile("" ::"v"(al[set][i])); }
#pragma unroll
      for (int j = 0; j < TN; ++j) { asm volatile("" ::"v"(bh[set][j])); asm volatile("" ::"v"(bl[set][j])); }
      return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[set][i], bh[set][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[set][i], bl[set][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[set][i], bh[set][j], acc[i][j], 0, 0, 0);
      }
  };
  float4 stg[NLOAD > 0 ? NLOAD : 1];
  const long long c0 = clock64(), w0 = wall_clock64();
  load(0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    const int stage = it & 1;
    load(1, stage, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    __builtin_amdgcn_sched_barrier(0);
    if (PATH == 1 && it > 0) {   // last step's register-staged tile goes to LDS now
#pragma unroll
      for (int k = 0; k < NLOAD; ++k)
        *reinterpret_cast<float4*>(lds + (stage ^ 1) * 32768 + (wave * NLOAD + k) * 512 + lane * 8) = stg[k];
    }
    __syncthreads();
    unsigned short* dst = lds + stage * 32768 + wave * NLOAD * 512;   // the stage just vacated
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
      const char* g = src + ((goff + (size_t)k * 1024) & (span - 1)) + lane * 16;
      if (PATH == 0) GLDS16(g, dst + k * 512);
      else stg[k] = *reinterpret_cast<const float4*>(g);
    }
    goff = (goff + 65536 * 4) & (span - 1);
    load(0, stage ^ 1, 0);
    if (SPREAD == 0) __builtin_amdgcn_sched_barrier(0);
    mma(1);
    if (SPREAD == 1) {   // one ingest instruction, then three MFMAs, ... ; the fragment reads ride along
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 1234.5f) sink[0] = s + lds[40000];
  if (blockIdx.x == 7 && threadIdx.x == 0) {
    reinterpret_cast<long long*>(sink)[2] = clock64() - c0;
    reinterpret_cast<long long*>(sink)[3] = wall_clock64() - w0;
  }
}

static float* g_sink = nullptr;
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
  long long cw[4] = {0, 0, 0, 0};
  if (g_sink) { hipMemcpy(cw, g_sink, 32, hipMemcpyDeviceToHost); hipMemset(g_sink, 0, 32); }
  printf("%-58s %8.3f ms  %8.1f TFLOP/s  shader clk %6.0f MHz\n", name, ms, flop_per_iter_per_block * iters * blocks / ms / 1e9,
         cw[3] ? (double)cw[2] / cw[3] * 100.0 : 0.0);
}

int main() {
  float* sink; hipMalloc(&sink, 64); hipMemset(sink, 0, 64); g_sink = sink;
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
  char* src; const size_t span = 512u << 20;
  hipMalloc(&src, span + (1 << 20)); hipMemset(src, 0, span + (1 << 20));
  const double step = mf * 48 * 8;
  for (size_t sp : {(size_t)2 << 20, span}) {
    printf("--- ingest source span %zu MB; TFLOP/s = issued MFMAs (48 per wave per step); NLOAD=8 is 64 KB per step\n", sp >> 20);
#define CS(N, M, L, P, ...) run("step nload=" #N " mfma=" #M " ldsread=" #L " path=" #P " spread=" #__VA_ARGS__, [&](int it) { hipLaunchKernelGGL((conv_step<N, M, L, P, ##__VA_ARGS__>), dim3(256), dim3(512), 131072, 0, src, sp, sink, it); }, 5000, step, 256)
    CS(0, true, false, 0); CS(0, true, true, 0);
    CS(8, true, false, 0); CS(8, true, true, 0); CS(4, true, true, 0);
    CS(8, false, false, 0); CS(8, false, true, 0);
    CS(8, true, false, 0, 1); CS(8, true, true, 0, 1); CS(4, true, true, 1); CS(4, false, true, 1); CS(4, true, false, 1); CS(2, true, true, 1); CS(2, true, true, 0);
  }
  return 0;
}
