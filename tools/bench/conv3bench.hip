// timing of launch_conv3_p64 against launch_gemm on layer1's 3x3 shape, with ablations
// (experiments build: make -C neuron-descriptions_amd/csrc EXPERIMENTS=1)
#include "../../neuron-descriptions_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
using namespace milan;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 3840;
  const int h = 56, w = 56;
  const size_t px = (size_t)n * h * w;
  float *in, *ws, *out, *bias, *zero;
  CK(hipMalloc(&in, px * 256)); CK(hipMalloc(&out, px * 256)); CK(hipMalloc(&ws, 64 * 576 * 4));
  CK(hipMalloc(&bias, 256)); CK(hipMalloc(&zero, 256)); CK(hipMemset(zero, 0, 256)); CK(hipMemset(bias, 0, 256));
  CK(hipMemset(in, 0x2c, px * 256)); CK(hipMemset(ws, 0x2c, 64 * 576 * 4));
  Conv3Args a{}; a.in = in; a.ws = ws; a.bias = bias; a.acc_scale = 0.5f; a.out = out; a.zero = zero; a.n = n; a.h = h; a.w = w;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
  const int dbg[] = {0, 1, 2, 3, 4, 8, 9, 11, 15, 14, 6};
  for (int d : dbg) {
    a.debug = d;
    for (int r = 0; r < 2; ++r) if (launch_conv3_p64(a, 0)) { printf("fail %s\n", milan_last_error()); return 1; }
    hipEventRecord(e0, 0); for (int r = 0; r < 5; ++r) launch_conv3_p64(a, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("debug %2d (%s%s%s%s): %.3f ms\n", d, d & 1 ? "noMFMA " : "", d & 2 ? "noEpilogue " : "", d & 4 ? "noDMA " : "", d & 8 ? "noHandover " : "", ms / 5);
  }
  {
    long long* pr; CK(hipMalloc(&pr, 128)); CK(hipMemset(pr, 0, 128));
    a.debug = 0; a.prof = pr; launch_conv3_p64(a, 0); CK(hipDeviceSynchronize()); a.prof = nullptr;
    long long hp[8]; CK(hipMemcpy(hp, pr, 64, hipMemcpyDeviceToHost));
    const double steps = 2.0 * ((n + 7) / 8 * 28 / 32) + 2;
    printf("workgroup 0, cycles per step: first-half wave {MFMA + hand-over %.0f, DMA issue + wait %.0f, barrier %.0f}\n"
           "                              second-half wave {epilogue %.0f, hand-over + MFMA %.0f, barrier %.0f}\n",
           hp[0] / steps, hp[1] / steps, hp[2] / steps, hp[4] / steps, hp[5] / steps, hp[6] / steps);
  }
  // the implicit GEMM on the same shape
  GemmArgs g{};
  g.A = in; g.W = ws; g.bias = bias; g.C = out; g.M = (int)px; g.N = 64; g.K = 576; g.Kp = 576; g.ldc = 64; g.ldaux = 64;
  g.H = h; g.Wd = w; g.Cin = 64; g.Ho = h; g.Wo = w; g.KH = 3; g.KW = 3; g.stride = 1; g.pad = 1;
  g.a_pix_stride = 64; g.a_img_stride = (long)h * w * 64; g.epilogue = EPI_BIAS_RELU; g.zero = zero;
  g.a_split = 1; g.out_split = 1; g.acc_scale = 0.5f;
  for (int r = 0; r < 2; ++r) if (launch_gemm(g, 0)) { printf("gemm fail %s\n", milan_last_error()); return 1; }
  hipEventRecord(e0, 0); for (int r = 0; r < 5; ++r) launch_gemm(g, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1); printf("implicit GEMM: %.3f ms\n", ms / 5);
  return 0;
}
