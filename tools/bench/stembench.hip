// timing of launch_stem_fused on synthetic buffers, with ablations (scratch)
#include "../../neuron-descriptions_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
using namespace milan;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void fill(float* p, long n, unsigned seed) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = ((x >> 8) * (1.f / 16777216.f) - 0.5f) * 0.1f;
  }
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 3840;
  const int H = 224, W = 224, G = (W + 2) / 2, h1 = 112, w1 = 112, hp = 56, wp = 56;
  float *in, *ws, *raw, *y, *sc, *sh, *zero; int* bbox;
  CK(hipMalloc(&in, (size_t)n * H * G * 32)); CK(hipMalloc(&ws, 64 * 224 * 4));
  CK(hipMalloc(&raw, (size_t)n * h1 * w1 * 256)); CK(hipMalloc(&y, (size_t)n * hp * wp * 256));
  CK(hipMalloc(&sc, 256)); CK(hipMalloc(&sh, 256)); CK(hipMalloc(&zero, 256)); CK(hipMemset(zero, 0, 256));
  CK(hipMalloc(&bbox, (size_t)n * 16));
  // split-format content does not matter for timing; use small f16-ish bit patterns
  CK(hipMemset(in, 0x2c, (size_t)n * H * G * 32)); CK(hipMemset(ws, 0x2c, 64 * 224 * 4));
  hipLaunchKernelGGL(fill, dim3(1), dim3(64), 0, 0, sc, 64L, 1u); hipLaunchKernelGGL(fill, dim3(1), dim3(64), 0, 0, sh, 64L, 2u);
  std::vector<int> bb(n * 4);
  for (int i = 0; i < n; ++i) { bb[4*i] = 20 + i % 30; bb[4*i+1] = bb[4*i] + 40; bb[4*i+2] = 10 + i % 50; bb[4*i+3] = bb[4*i+2] + 45; }
  CK(hipMemcpy(bbox, bb.data(), n * 16, hipMemcpyHostToDevice));
  StemArgs a{}; a.in = in; a.ws = ws; a.acc_scale = 0.5f; a.scale = sc; a.shift = sh; a.raw = raw; a.y = y; a.bbox = bbox; a.zero = zero;
  a.n = n; a.H = H; a.G = G; a.h1 = h1; a.w1 = w1; a.hp = hp; a.wp = wp;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int dbg[] = {0, 1, 2, 3, 4, 8, 10, 11, 15, 7};
  for (int d : dbg) {
    a.debug = d;
    for (int r = 0; r < 2; ++r) if (launch_stem_fused(a, 0)) { printf("fail %s\n", milan_last_error()); return 1; }
    hipEventRecord(e0, 0); for (int r = 0; r < 5; ++r) launch_stem_fused(a, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("debug %2d (%s%s%s%s): %.3f ms\n", d, d & 1 ? "noMFMA " : "", d & 2 ? "noStore " : "", d & 4 ? "noDMA " : "", d & 8 ? "noStage " : "", ms / 5);
  }
  a.debug = 0; a.bbox = nullptr;
  hipEventRecord(e0, 0); for (int r = 0; r < 5; ++r) launch_stem_fused(a, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); printf("all raw rows written: %.3f ms\n", ms / 5);
  return 0;
}
