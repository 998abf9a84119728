// Micro-benchmark: per-CU ingest rate of global_load_lds_dwordx4 from an L2-resident buffer for
// different row-segment shapes (how many contiguous bytes each row contributes per instruction),
// and of plain global_load_dwordx4 to registers.  hipcc --offload-arch=gfx950 -O3 dma_rate.hip -o dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned short u16;
#define GLDS16(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g), (__attribute__((address_space(3))) void*)(l), 16, 0, 0)

// SEG = contiguous bytes per row per instruction (64, 128, 256, 1024); row stride = 1472 B (736 halves)
template <int SEG, int INFLIGHT, int SWZ = 0, int BAR = 0>
__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ src, size_t span, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int LPR = SEG / 16;              // lanes per row
  constexpr int RPI = 64 / LPR;              // rows per instruction
  const size_t row_stride = 1472;
  size_t base = ((size_t)blockIdx.x * 8 + wave) * RPI * row_stride * 7 % span;
  const int rowi = lane / LPR;
  const int posi = SWZ ? ((lane % LPR) ^ ((rowi >> 2) & (LPR - 1))) : (lane % LPR);
  const size_t lane_off = (size_t)rowi * row_stride + posi * 16;
  char* l = lds + wave * (INFLIGHT * 1024);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) {
      size_t off = (base + (size_t)k * SEG) % (span - 65536);
      off &= ~(size_t)127;
      GLDS16(src + off + lane_off, l + k * 1024);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (BAR) __builtin_amdgcn_s_barrier();
    base += INFLIGHT * RPI * row_stride;
  }
  if (threadIdx.x == 0 && iters < 0) sink[0] = lds[0];
}

template <int SEG>
__global__ __launch_bounds__(512) void reg_kernel(const char* __restrict__ src, size_t span, int iters, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = SEG / 16, RPI = 64 / LPR;
  const size_t row_stride = 1472;
  size_t base = ((size_t)blockIdx.x * 8 + wave) * RPI * row_stride * 7 % span;
  const size_t lane_off = (size_t)(lane / LPR) * row_stride + (lane % LPR) * 16;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      size_t off = (base + (size_t)k * SEG) % (span - 65536);
      off &= ~(size_t)127;
      const float4 v = *reinterpret_cast<const float4*>(src + off + lane_off);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    base += 6 * RPI * row_stride;
  }
  if (acc.x == 1234.5f) sink[0] = acc.y + acc.z + acc.w;
}

template <typename F>
static void run(const char* name, F launch, int iters, double bytes_per_block_iter, int blocks) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch(10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  launch(iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double tot = bytes_per_block_iter * iters * blocks;
  printf("%-34s %8.3f ms  %7.2f TB/s  %6.1f GB/s per CU\n", name, ms, tot / ms / 1e9, tot / ms / 1e6 / 256);
}

int main() {
  const size_t span = 96u << 20;             // 96 MB: MALL-resident, mostly L2-missing; also try 3 MB
  char* src; float* sink;
  hipMalloc(&src, span + (1 << 20)); hipMemset(src, 1, span + (1 << 20)); hipMalloc(&sink, 64);
  for (size_t sp : {(size_t)2 << 20, (size_t)24 << 20, span}) {
    printf("--- working set %zu MB, 256 blocks x 8 waves\n", sp >> 20);
    const int blocks = 256, iters = 2000;
#define DMA(SEG, INF) run("dma seg=" #SEG " inflight=" #INF, [&](int it) { hipLaunchKernelGGL((dma_kernel<SEG, INF>), dim3(blocks), dim3(512), 8 * INF * 1024, 0, src, sp, it, sink); }, iters, 8.0 * INF * 1024, blocks)
    DMA(64, 6); DMA(128, 6); DMA(256, 6); DMA(1024, 6); DMA(64, 12); DMA(128, 12);
#define DMAX(SEG, INF, SW, BR) run("dma seg=" #SEG " inflight=" #INF " swz=" #SW " bar=" #BR, [&](int it) { hipLaunchKernelGGL((dma_kernel<SEG, INF, SW, BR>), dim3(blocks), dim3(512), 8 * INF * 1024, 0, src, sp, it, sink); }, iters, 8.0 * INF * 1024, blocks)
    DMAX(64, 6, 1, 0); DMAX(64, 6, 0, 1); DMAX(64, 6, 1, 1);
#define REG(SEG) run("regs seg=" #SEG, [&](int it) { hipLaunchKernelGGL((reg_kernel<SEG>), dim3(blocks), dim3(512), 0, 0, src, sp, it, sink); }, iters, 8.0 * 6 * 1024, blocks)
    REG(64); REG(128); REG(1024);
  }
  return 0;
}
