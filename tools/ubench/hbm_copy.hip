// Micro-benchmark: what a plain streaming kernel gets out of HBM on this box (read+write bytes / time), to put the
// 4.4-4.6 TB/s of the HBM-bound kernels (depthwise, pool, fused block at 237^2) and hipMemcpyDtoD (4.5) in context.
//   hipcc --offload-arch=gfx950 -O3 hbm_copy.hip -o hbm_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float vf4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const vf4* __restrict__ a, vf4* __restrict__ b, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n; i += stride) {
    vf4 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k)
      if (i + k * 256 < n) v[k] = NT ? __builtin_nontemporal_load(a + i + k * 256) : a[i + k * 256];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k)
      if (i + k * 256 < n) {
        if (NT) __builtin_nontemporal_store(v[k], b + i + k * 256);
        else b[i + k * 256] = v[k];
      }
  }
}
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ a, float* sink, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += stride) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k * 256 < n) { float4 v = a[i + k * 256]; s += v.x + v.y + v.z + v.w; }
  }
  if (s == 12345.678f) sink[0] = s;
}

template <typename F>
static float time_ms(F f, int it) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < it; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / it;
}

int main() {
  const size_t bytes = (size_t)1840 << 20, n = bytes / 16;
  float4 *a, *b;
  float* sink;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
  hipMemset(a, 1, bytes);
  for (int g : {256, 512, 1024, 2048, 4096, 16384, 65536}) {
    float t1 = time_ms([&] { copy_kernel<1, false><<<g, 256>>>((const vf4*)a, (vf4*)b, n); }, 10);
    float t4 = time_ms([&] { copy_kernel<4, false><<<g, 256>>>((const vf4*)a, (vf4*)b, n); }, 10);
    float t4n = time_ms([&] { copy_kernel<4, true><<<g, 256>>>((const vf4*)a, (vf4*)b, n); }, 10);
    float t8n = time_ms([&] { copy_kernel<8, true><<<g, 256>>>((const vf4*)a, (vf4*)b, n); }, 10);
    float tr = time_ms([&] { read_kernel<<<g, 256>>>(a, sink, n); }, 10);
    printf("grid %6d: copy x1 %.2f  x4 %.2f  x4 nt %.2f  x8 nt %.2f TB/s (read+write)   read-only %.2f TB/s\n", g,
           2 * bytes / t1 / 1e9, 2 * bytes / t4 / 1e9, 2 * bytes / t4n / 1e9, 2 * bytes / t8n / 1e9, bytes / tr / 1e9);
  }
  return 0;
}
