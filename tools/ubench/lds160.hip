// Is the whole 160 KB of a CU's LDS addressable by ONE workgroup with plain DS instructions?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds160.hip -o /tmp/lds160 && /tmp/lds160
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k(unsigned* out, int words) {
  extern __shared__ unsigned s[];
  for (int i = threadIdx.x; i < words; i += 512) s[i] = 0x9e3779b9u * (unsigned)i + 1u;
  __syncthreads();
  unsigned bad = 0;
  for (int i = threadIdx.x; i < words; i += 512) bad += s[i] != 0x9e3779b9u * (unsigned)i + 1u;
  // first word of every 4 KB page, to see aliasing
  if (threadIdx.x < 40) out[1 + threadIdx.x] = s[threadIdx.x * 1024];
  atomicAdd(out, bad);
}
int main() {
  int dev = 0, v = 0;
  hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
  printf("MaxSharedMemoryPerBlock %d\n", v);
  unsigned* d;
  hipMalloc(&d, 64 * 4);
  for (int kb : {64, 128, 144, 160}) {
    hipMemset(d, 0, 64 * 4);
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(512), kb * 1024, 0, d, kb * 256);
    hipError_t e2 = hipDeviceSynchronize();
    std::vector<unsigned> h(64);
    hipMemcpy(h.data(), d, 64 * 4, hipMemcpyDeviceToHost);
    printf("%3d KB: attr %d run %d mismatching words %u  page heads ok %d\n", kb, (int)e, (int)e2, h[0],
           (int)(h[1 + kb / 4 - 1] == 0x9e3779b9u * (unsigned)((kb / 4 - 1) * 1024) + 1u));
  }
  return 0;
}
