// bitwise check + timing of launch_chain against the two separate launches (scratch)
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
  long bad = 0; for (size_t i = 0; i < n; ++i) if (memcmp(&ha[i], &hb[i], 4)) { if (bad < 5) printf("  diff at %zu (row %zu): %08x vs %08x\n", i, i, *(unsigned*)&ha[i], *(unsigned*)&hb[i]); ++bad; }
  return bad;
}
int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 256;
  const long M = argc > 2 ? atol(argv[2]) : 128 * 20;
  const int reps = argc > 3 ? atoi(argv[3]) : 1;
  const int N3 = 4 * P;
  float* zero = dalloc(64);
  float* t2f = rnd((size_t)M * P, 1, 0.f, 2.f);
  float* w3f = rnd((size_t)N3 * P, 2, -0.06f, 0.06f);
  float* w1f = rnd((size_t)P * N3, 3, -0.03f, 0.03f);
  float* rf = rnd((size_t)M * N3, 4, 0.f, 3.f);
  float* b3 = rnd(N3, 5, -0.5f, 0.5f);
  float* b1 = rnd(P, 6, -0.5f, 0.5f);
  float* T2 = to_split(t2f, M, P); float* W3 = to_split(w3f, N3, P); float* W1 = to_split(w1f, P, N3); float* R = to_split(rf, M, N3);
  float* Xr = dalloc((size_t)M * N3); float* T1r = dalloc((size_t)M * P);
  float* X = dalloc((size_t)M * N3); float* T1 = dalloc((size_t)M * P);
  GemmArgs g3 = linear_args(T2, P, W3, b3, Xr, N3, (int)M, N3, P, EPI_BIAS_RES_RELU, zero, R, N3);
  g3.a_split = 1; g3.out_split = 1; g3.aux_split = 1; g3.acc_scale = 0.5f;
  // conv-like geometry (1x1 over a 14x14 image) so the launcher treats it as the trunk does
  GemmArgs g1 = linear_args(Xr, N3, W1, b1, T1r, P, (int)M, P, N3, EPI_BIAS_RELU, zero);
  g1.a_split = 1; g1.out_split = 1; g1.acc_scale = 0.25f;
  const int KD = argc > 4 ? atoi(argv[4]) : 0;
  ChainArgs c{}; c.T2 = T2; c.W3 = W3; c.bias3 = b3; c.R = R; c.X = X; c.W1 = W1; c.bias1 = b1; c.T1 = T1; c.M = (int)M; c.P = P; c.scale3 = 0.5f; c.scale1 = 0.25f;
  if (KD) {
    // two-source expand: [T2 | A2] with K-concatenated weights, no residual
    float* a2f = rnd((size_t)M * KD, 7, 0.f, 2.f); float* A2 = to_split(a2f, M, KD);
    float* w3cf = rnd((size_t)N3 * (P + KD), 8, -0.06f, 0.06f); float* W3c = to_split(w3cf, N3, P + KD);
    g3 = linear_args(T2, P, W3c, b3, Xr, N3, (int)M, N3, P + KD, EPI_BIAS_RELU, zero);
    g3.a_split = 1; g3.out_split = 1; g3.acc_scale = 0.5f;
    g3.Cin = P; g3.A2 = A2; g3.K1 = P; g3.H2 = 1; g3.W2d = 1; g3.stride2 = 1; g3.a2_pix_stride = KD; g3.a2_img_stride = KD;
    c.W3 = W3c; c.R = nullptr; c.A2 = A2; c.KD = KD;
  }
  if (launch_gemm(g3, 0) || launch_gemm(g1, 0)) { printf("ref launch failed: %s\n", milan_last_error()); return 1; }
  if (launch_chain(c, 0)) { printf("chain launch failed: %s\n", milan_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  const long dx = diff(X, Xr, (size_t)M * N3), dt = diff(T1, T1r, (size_t)M * P);
  printf("P=%d M=%ld: X mismatches %ld / %ld, T1 mismatches %ld / %ld\n", P, M, dx, M * N3, dt, M * P);
  if (getenv("CHAIN_PROF")) {
    // chainw_kernel: 8 counters for wave 0 and for wave 4 of every workgroup
    // (chain3_kernel is persistent: at most one workgroup per CU; counters are sums over its tiles)
    const long ntile = (M + 127) / 128;
    const long nwg = getenv("MILAN_CHAIN3") && atoi(getenv("MILAN_CHAIN3")) == 0 ? ntile : (ntile < 256 ? ntile : 256);
    long long* pr; CK(hipMalloc((void**)&pr, nwg * 128)); CK(hipMemset(pr, 0, nwg * 128));
    c.prof = pr; launch_chain(c, 0); CK(hipDeviceSynchronize()); c.prof = nullptr;
    std::vector<long long> h(nwg * 16); CK(hipMemcpy(h.data(), pr, nwg * 128, hipMemcpyDeviceToHost));
    // chain3_kernel's counters (MILAN_CHAIN3=0: chainw_kernel's -- prologue, dma issue, expand
    // mfma, hand-over / strip, reduce mfma, wait + barrier, epilogue B, final epilogue)
    const char* nm[8] = {"prologue", "dma issue", "mfma half-slots", "raw tile->strip", "epilogue items", "wait+barrier", "residual prefetch", "final epilogue"};
    for (int w = 0; w < 2; ++w) {
      double sum[8] = {0}; for (long i = 0; i < nwg; ++i) for (int k = 0; k < 8; ++k) sum[k] += h[(i * 2 + w) * 8 + k];
      double tot = 0; for (int k = 0; k < 8; ++k) tot += sum[k];
      printf(" wave %d:", 4 * w);
      for (int k = 0; k < 8; ++k) printf(" %s %.0f", nm[k], sum[k] / ntile);
      printf(" | total %.0f cycles per 128-pixel tile\n", tot / ntile);
    }
  }
  if (reps > 1) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms;
    for (int r = 0; r < 3; ++r) { launch_gemm(g3, 0); launch_gemm(g1, 0); launch_chain(c, 0); }
    hipEventRecord(a, 0); for (int r = 0; r < reps; ++r) { launch_gemm(g3, 0); launch_gemm(g1, 0); } hipEventRecord(b, 0); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); printf("  separate: %.3f ms per pair\n", ms / reps);
    hipEventRecord(a, 0); for (int r = 0; r < reps; ++r) launch_chain(c, 0); hipEventRecord(b, 0); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    const double bytes = 4.0 * ((double)M * P * 2 + (double)M * N3 * 2);
    printf("  chain:    %.3f ms  (%.2f TB/s algorithmic, %.1f TF-eq)\n", ms / reps, bytes / (ms / reps) / 1e9, 2.0 * M * N3 * P * 2 / (ms / reps) / 1e9);
  }
  return 0;
}
