// micro-benchmark harness over libmilan_hip's internal launch_gemm (scratch)
#include "../../neuron-descriptions_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace milan;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
static float* dalloc(size_t floats, int fill) {
  float* p; CK(hipMalloc((void**)&p, floats * 4));
  CK(hipMemset(p, fill, floats * 4));
  return p;
}
int main(int argc, char** argv) {
  float* zero = dalloc(64, 0);
  struct Cfg { long M; int N, K; int res; int hint; };
  std::vector<Cfg> cfgs;
  for (int i = 1; i + 4 < argc + 0 || i + 4 <= argc - 1 + 1; i += 5) {
    if (i + 4 >= argc) break;
    cfgs.push_back({atol(argv[i]), atoi(argv[i+1]), atoi(argv[i+2]), atoi(argv[i+3]), atoi(argv[i+4])});
  }
  for (auto& c : cfgs) {
    // 0x3c00 pattern = f16 1.0 everywhere (hi and lo): values irrelevant for timing
    float* A = dalloc((size_t)c.M * c.K, 0x3c);
    float* W = dalloc((size_t)c.N * c.K, 0x3c);
    float* R = c.res ? dalloc((size_t)c.M * c.N, 0x3c) : nullptr;
    float* C = dalloc((size_t)c.M * c.N, 0);
    GemmArgs g = linear_args(A, c.K, W, nullptr, C, c.N, (int)c.M, c.N, c.K,
                             c.res ? EPI_BIAS_RES_RELU : EPI_BIAS_RELU, zero, R, c.N);
    g.a_split = 1; g.out_split = c.hint < 100; g.aux_split = c.res && c.hint < 100; if (c.hint >= 100) c.hint -= 100; g.acc_scale = 1.f; g.tile_hint = c.hint;
    // conv-like geometry so rows are "pixels"
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) if (launch_gemm(g, 0)) { printf("launch failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(a, 0));
    for (int r = 0; r < reps; ++r) launch_gemm(g, 0);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    {  // experiments build: per-phase cycles (wave 0 of every workgroup, averaged per tile)
      long long* pr; CK(hipMalloc((void**)&pr, 1024)); CK(hipMemset(pr, 0, 1024));
      g.prof = pr; launch_gemm(g, 0); CK(hipDeviceSynchronize()); g.prof = nullptr;
      long long hp[128]; CK(hipMemcpy(hp, pr, 1024, hipMemcpyDeviceToHost));
      const double tiles = (double)((c.M + 255) / 256) * ((c.N + 255) / 256);
      const double kts = tiles * (c.K / 16);
      if (hp[5]) printf("  cycles per tile (wave 0): prologue issue %.0f, fill wait %.0f, drain + scale %.0f, epilogue %.0f; per k-tile: compute + DMA issue %.0f, DMA wait %.0f, barrier %.0f\n",
                        hp[0] / tiles, hp[1] / tiles, hp[3] / tiles, hp[4] / tiles, hp[5] / kts, hp[6] / kts, (hp[7] + hp[2]) / kts);
      if (hp[5]) for (int wv = 0; wv < 8; ++wv)
        printf("    wave %d per k-tile: compute + issue %.0f, DMA wait %.0f, barrier %.0f; epilogue %.0f\n", wv, hp[8 * wv + 5] / kts,
               hp[8 * wv + 6] / kts, (hp[8 * wv + 7] + hp[8 * wv + 2]) / kts, hp[8 * wv + 4] / tiles);
      hipFree(pr);
    }
    const double bytes = 4.0 * ((double)c.M * c.K + (double)c.M * c.N * (c.res ? 2 : 1));
    const double fl = 2.0 * c.M * c.N * c.K;
    printf("M=%ld N=%d K=%d res=%d hint=%d: %.3f ms  %.1f TF-eq  %.2f TB/s alg  (%.1f us per tile-round of 256 WG)\n",
           c.M, c.N, c.K, c.res, c.hint, ms, fl / ms / 1e9, bytes / ms / 1e9,
           ms * 1e3 / ((double)((c.M + 255) / 256) * ((c.N + 255) / 256) / 256.0));
    hipFree(A); hipFree(W); if (R) hipFree(R); hipFree(C);
  }
  return 0;
}
