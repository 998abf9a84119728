// Bottleneck probe for the fused separable block (GPU box): the product kernel source compiled with parts switched
// off (SF_PROBE_LEVEL, see sepconv_fused.hip) or another patch-ring depth (SF_NSLOT), timed on the entry-flow shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSF_PROBE_LEVEL=1 -DSF_NSLOT=3 tools/ubench/sepconv_probe.hip -o probe1
//   ./probe1 [batch]
#include "../../x-detector_amd/csrc/sepconv_fused.hip"
#include <vector>
namespace xdet {
void set_last_error(const std::string& s) { fprintf(stderr, "error: %s\n", s.c_str()); }
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  fprintf(stderr, "%s:%d %s: %s\n", file, line, what, hipGetErrorString(e));
  return XDET_ERR_HIP;
}
}  // namespace xdet

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 64;
  struct Shape { const char* name; int hw, cin, cout, hpool, relu_in; };
  const Shape shapes[] = {{"block2_sepconv1", 237, 64, 128, 0, 0}, {"block2_sepconv2+hpool", 237, 128, 128, 1, 0},
                          {"block3_sepconv1", 119, 128, 256, 0, 1}, {"block3_sepconv2+hpool", 119, 256, 256, 1, 0},
                          {"block4_sepconv1", 60, 256, 768, 0, 1}};
  printf("probe level %d, patch ring %d, batch %d\n", SF_PROBE_LEVEL, SF_NSLOT, N);
  for (const Shape& sh : shapes) {
    const size_t n_in = (size_t)N * sh.hw * sh.hw * sh.cin, n_out = (size_t)N * sh.hw * sh.hw * sh.cout;
    float *in, *out, *w9, *sc, *shf;
    unsigned short *wh, *wl;
    hipMalloc(&in, n_in * 4); hipMalloc(&out, n_out * 4); hipMalloc(&w9, 9 * sh.cin * 4);
    hipMalloc(&sc, sh.cout * 4); hipMalloc(&shf, sh.cout * 4);
    hipMalloc(&wh, (size_t)sh.cin * sh.cout * 2); hipMalloc(&wl, (size_t)sh.cin * sh.cout * 2);
    hipMemset(in, 0, n_in * 4); hipMemset(w9, 0, 9 * sh.cin * 4); hipMemset(sc, 0, sh.cout * 4); hipMemset(shf, 0, sh.cout * 4);
    hipMemset(wh, 0, (size_t)sh.cin * sh.cout * 2); hipMemset(wl, 0, (size_t)sh.cin * sh.cout * 2);
#if SF_PROBE_LEVEL == 9
    static unsigned long long* g_dbg = nullptr;
    if (!g_dbg) hipMalloc(&g_dbg, 8 * 64 * 8 * 8);
    hipMemset(g_dbg, 0, 8 * 64 * 8 * 8);
    xdet::g_probe_dbg = g_dbg;
#endif
    auto go = [&]() {
      return xdet::launch_sepconv_fused(in, w9, wh, wl, sc, shf, out, N, sh.hw, sh.hw, sh.cin, sh.cout, sh.cout, sh.relu_in, 0, 0,
                                        sh.hpool ? 0 : -1);
    };
    if (go() != 0) return 1;
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 10;
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) go();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= iters;
    const double gb = (n_in + (sh.hpool ? n_out / 2 : n_out)) * 4 / 1e9;
    printf("  %-24s %8.3f ms   %6.2f TB/s (in + out once)\n", sh.name, ms, gb / ms);
#if SF_PROBE_LEVEL == 9
    {
      // one more launch, then the stamps of workgroup 8 (waves 0 = producer, 4 = consumer), in clocks relative to step 8
      hipMemset(g_dbg, 0, 8 * 64 * 8 * 8);
      go();
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(8 * 64 * 8);
      hipMemcpy(h.data(), g_dbg, h.size() * 8, hipMemcpyDeviceToHost);
      const unsigned long long t0 = h[(0 * 64 + 8) * 8 + 0];
      printf("    step | producer wave 0: arrive landed released dma-issued row0 row1 row2 writes-issued | consumer wave 4: arrive released half0 half1(+epilogue)\n");
      for (int st = 8; st < 34; ++st) {
        printf("    %4d |", st);
        const int order[8] = {0, 1, 2, 4, 5, 6, 7, 3};
        for (int k = 0; k < 8; ++k) printf(" %7lld", (long long)(h[(0 * 64 + st) * 8 + order[k]] - t0));
        printf("  |");
        for (int k = 0; k < 4; ++k) printf(" %7lld", (long long)(h[(4 * 64 + st) * 8 + k] - t0));
        printf("\n");
      }
    }
#endif
    hipFree(in); hipFree(out); hipFree(w9); hipFree(sc); hipFree(shf); hipFree(wh); hipFree(wl);
  }
  return 0;
}
