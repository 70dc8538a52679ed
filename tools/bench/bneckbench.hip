// layer1 bottleneck tail: conv3_p64 + chain (two launches) against the chain launch with the
// 3x3 conv in front (chain_kernel<.., CONV>, round 6): bitwise check, timing, and -- with
// BNECK_PROF=1 -- the in-kernel phase profile of the fused kernel.
//   bneckbench [images=1280] [reps=20]         (56 x 56 x 64 -> 256 -> 64, the l1.x geometry)
#include "../../neuron-descriptions_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace milan;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void fill(float* p, long n, unsigned seed, float lo, float hi) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = lo + (hi - lo) * (x >> 8) * (1.f / 16777216.f);
  }
}
static float* dalloc(size_t floats) { float* p; CK(hipMalloc((void**)&p, floats * 4)); CK(hipMemset(p, 0, floats * 4)); return p; }
static float* rnd(size_t n, unsigned seed, float lo, float hi) { float* p = dalloc(n); hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, p, (long)n, seed, lo, hi); return p; }
static float* to_split(const float* src, long rows, int K) { float* d = dalloc((size_t)rows * K); if (launch_f32_to_split(src, K, d, K, rows, K, 1.f, 0)) exit(2); return d; }
static long diff(const float* a, const float* b, size_t n) {
  std::vector<float> ha(n), hb(n);
  CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
  long bad = 0; for (size_t i = 0; i < n; ++i) if (memcmp(&ha[i], &hb[i], 4)) ++bad;
  return bad;
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1280, reps = argc > 2 ? atoi(argv[2]) : 20;
  const int h = 56, w = 56, P = 64, N3 = 256;
  const long M = (long)n * h * w;
  float* zero = dalloc(64);
  float* T1in = to_split(rnd((size_t)M * P, 1, 0.f, 2.f), M, P);
  float* W2 = to_split(rnd((size_t)P * 576, 9, -0.04f, 0.04f), P, 576);
  float* W3 = to_split(rnd((size_t)N3 * P, 2, -0.06f, 0.06f), N3, P);
  float* W1 = to_split(rnd((size_t)P * N3, 3, -0.03f, 0.03f), P, N3);
  float* R = to_split(rnd((size_t)M * N3, 4, 0.f, 3.f), M, N3);
  float *b2 = rnd(P, 7, -0.5f, 0.5f), *b3 = rnd(N3, 5, -0.5f, 0.5f), *b1 = rnd(P, 6, -0.5f, 0.5f);
  float *T2 = dalloc((size_t)M * P), *Xr = dalloc((size_t)M * N3), *T1r = dalloc((size_t)M * P);
  float *X = dalloc((size_t)M * N3), *T1 = dalloc((size_t)M * P);
  Conv3Args ca{}; ca.in = T1in; ca.ws = W2; ca.bias = b2; ca.acc_scale = 0.5f; ca.out = T2; ca.zero = zero; ca.n = n; ca.h = h; ca.w = w;
  ChainArgs c{}; c.T2 = T2; c.W3 = W3; c.bias3 = b3; c.R = R; c.X = Xr; c.W1 = W1; c.bias1 = b1; c.T1 = T1r; c.M = (int)M; c.P = P; c.scale3 = 0.5f; c.scale1 = 0.25f;
  ChainArgs f = c; f.T2 = nullptr; f.X = X; f.T1 = T1; f.C2in = T1in; f.W2 = W2; f.bias2 = b2; f.scale2 = 0.5f; f.ch = h; f.cw = w;
  if (launch_conv3_p64(ca, 0) || launch_chain(c, 0) || launch_chain(f, 0)) { printf("launch failed: %s\n", milan_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  printf("n=%d M=%ld: X mismatches %ld, T1 mismatches %ld\n", n, M, diff(X, Xr, (size_t)M * N3), diff(T1, T1r, (size_t)M * P));
  if (getenv("BNECK_PROF")) {
    const long nwg = (M + 255) / 256;
    long long* pr; CK(hipMalloc((void**)&pr, nwg * 64)); CK(hipMemset(pr, 0, nwg * 64));
    f.prof = pr; launch_chain(f, 0); CK(hipDeviceSynchronize()); f.prof = nullptr;
    std::vector<long long> hp(nwg * 8); CK(hipMemcpy(hp.data(), pr, nwg * 64, hipMemcpyDeviceToHost));
    const char* nm[8] = {"prologue (region + first tiles land)", "chain: tile issue", "chain: mfma steps", "chain: expand epilogue", "chain: ring wait", "chain: barrier", "chain: reduce epilogue", "3x3 phase"};
    double tot = 0; for (int k = 0; k < 8; ++k) { double s = 0; for (long i = 0; i < nwg; ++i) s += hp[i * 8 + k]; printf("  %-40s %8.0f cycles per workgroup\n", nm[k], s / nwg); tot += s / nwg; }
    printf("  total %.0f cycles per 256-pixel workgroup (wave 0)\n", tot);
  }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms, ms3, msc;
  for (int r = 0; r < 3; ++r) { launch_conv3_p64(ca, 0); launch_chain(c, 0); launch_chain(f, 0); }
  hipEventRecord(a, 0); for (int r = 0; r < reps; ++r) launch_conv3_p64(ca, 0); hipEventRecord(b, 0); hipEventSynchronize(b); hipEventElapsedTime(&ms3, a, b);
  hipEventRecord(a, 0); for (int r = 0; r < reps; ++r) launch_chain(c, 0); hipEventRecord(b, 0); hipEventSynchronize(b); hipEventElapsedTime(&msc, a, b);
  hipEventRecord(a, 0); for (int r = 0; r < reps; ++r) launch_chain(f, 0); hipEventRecord(b, 0); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
  const double bytes = 4.0 * ((double)M * P * 2 + (double)M * N3 * 2);
  printf("  conv3_p64 %.3f ms + chain %.3f ms = %.3f ms;  fused %.3f ms (%.3f of the two; %.2f TB/s algorithmic)\n", ms3 / reps, msc / reps, (ms3 + msc) / reps, ms / reps, ms / (ms3 + msc), bytes / (ms / reps) / 1e9);
  return 0;
}
