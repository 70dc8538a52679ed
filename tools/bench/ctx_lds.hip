// Diagnosis of the round-2 "context_kernel is not reproducible when two processes share
// the GPU" finding.  Standalone: the removed LDS-staged kernel,
// verbatim, plus variants, run in a loop and compared bit for bit with a reference
// computed before the load process starts.  Run one instance with `check` while another
// process keeps the GPU busy (tools/bench/ubench in a loop, or a second instance).
//   ctx_lds <seconds> [variant]     variant: 0 old kernel (static 256 B LDS)
//                                            1 same, LDS array padded to 4 KB
//                                            2 same, dynamic LDS (256 B)
//                                            3 no LDS: scalar loads (the shipped form)
//                                            4 old kernel, ds_read one float at a time (volatile)
#ifdef WITH_MILAN
#include "../../neuron-descriptions_amd/csrc/common.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <thread>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int PAD, bool VOL>
__global__ __launch_bounds__(256) void context_lds(const float* __restrict__ att,
                                                   const float* __restrict__ feat, int rpn,
                                                   int k, int F, float* __restrict__ ctx) {
  __shared__ float a[PAD];
  const int r = blockIdx.x;
  if (threadIdx.x < k) a[threadIdx.x] = att[(long)r * k + threadIdx.x];
  __syncthreads();
  const float4* f4 = reinterpret_cast<const float4*>(feat + (long)(r / rpn) * k * F);
  const int F4 = F >> 2;
  for (int f = blockIdx.y * 256 + threadIdx.x; f < F4; f += gridDim.y * 256) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < k; ++j) {
      const float4 v = f4[(long)j * F4 + f];
      const float aj = VOL ? ((volatile float*)a)[j] : a[j];
      s.x += aj * v.x; s.y += aj * v.y; s.z += aj * v.z; s.w += aj * v.w;
    }
    reinterpret_cast<float4*>(ctx + (long)r * F)[f] = s;
  }
}
__global__ __launch_bounds__(256) void context_dyn(const float* __restrict__ att,
                                                   const float* __restrict__ feat, int rpn,
                                                   int k, int F, float* __restrict__ ctx) {
  extern __shared__ float a[];
  const int r = blockIdx.x;
  if (threadIdx.x < k) a[threadIdx.x] = att[(long)r * k + threadIdx.x];
  __syncthreads();
  const float4* f4 = reinterpret_cast<const float4*>(feat + (long)(r / rpn) * k * F);
  const int F4 = F >> 2;
  for (int f = blockIdx.y * 256 + threadIdx.x; f < F4; f += gridDim.y * 256) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < k; ++j) {
      const float4 v = f4[(long)j * F4 + f];
      const float aj = a[j];
      s.x += aj * v.x; s.y += aj * v.y; s.z += aj * v.z; s.w += aj * v.w;
    }
    reinterpret_cast<float4*>(ctx + (long)r * F)[f] = s;
  }
}
__global__ __launch_bounds__(256) void context_scalar(const float* __restrict__ att,
                                                      const float* __restrict__ feat, int rpn,
                                                      int k, int F, float* __restrict__ ctx) {
  const int r = blockIdx.x;
  const float* ar = att + (long)r * k;
  const float4* f4 = reinterpret_cast<const float4*>(feat + (long)(r / rpn) * k * F);
  const int F4 = F >> 2;
  for (int f = blockIdx.y * 256 + threadIdx.x; f < F4; f += gridDim.y * 256) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < k; ++j) {
      const float4 v = f4[(long)j * F4 + f];
      const float aj = ar[j];
      s.x += aj * v.x; s.y += aj * v.y; s.z += aj * v.z; s.w += aj * v.w;
    }
    reinterpret_cast<float4*>(ctx + (long)r * F)[f] = s;
  }
}
// Library-independent aggressor: a bare v_mfma_f32_32x32x16_f16 loop at
// the split16 GEMM's occupancy -- 512 threads = two waves per SIMD, ~230 VGPRs (fourteen
// accumulator tiles), 128 KB of LDS claimed so that one workgroup owns a CU and the
// victim's small workgroups fill the rest -- with NO LDS-DMA, no LDS traffic, no inline
// asm: registers and the kernel descriptor are entirely the compiler's.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(512, 2) void mfma_aggressor(float* __restrict__ out, int iters) {
  extern __shared__ float lds[];
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(0.001f * (float)((threadIdx.x + e) & 15));
    b[e] = (_Float16)(0.002f * (float)((threadIdx.x * 3 + e) & 7));
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) lds[threadIdx.x] = s;  // (keeps the LDS claim and the sums alive)
  out[(long)blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ void fill(float* p, long n, unsigned seed) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (x >> 8) * (1.f / 16777216.f);
  }
}
int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 10;
  const int variant = argc > 2 ? atoi(argv[2]) : 0;
  const int rows = 64 * 50, rpn = 50, k = 15, F = 3904, n = rows / rpn;
  float *att, *feat, *ctx;
  CK(hipMalloc((void**)&att, (size_t)rows * k * 4)); CK(hipMalloc((void**)&feat, (size_t)n * k * F * 4));
  CK(hipMalloc((void**)&ctx, (size_t)rows * F * 4));
  hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, att, (long)rows * k, 1u);
  hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, feat, (long)n * k * F, 2u);
  const dim3 grid(rows, (F / 4 + 255) / 256);
  const size_t dyn_lds = argc > 3 ? (size_t)atol(argv[3]) : 256;
  auto launch = [&]() {
    switch (variant) {
      case 0: hipLaunchKernelGGL((context_lds<64, false>), grid, dim3(256), 0, 0, att, feat, rpn, k, F, ctx); break;
      case 1: hipLaunchKernelGGL((context_lds<1024, false>), grid, dim3(256), 0, 0, att, feat, rpn, k, F, ctx); break;
      case 2: hipLaunchKernelGGL(context_dyn, grid, dim3(256), dyn_lds, 0, att, feat, rpn, k, F, ctx); break;
      case 3: hipLaunchKernelGGL(context_scalar, grid, dim3(256), 0, 0, att, feat, rpn, k, F, ctx); break;
      default: hipLaunchKernelGGL((context_lds<64, true>), grid, dim3(256), 0, 0, att, feat, rpn, k, F, ctx); break;
    }
  };
  // optional IN-PROCESS neighbour: a second host thread keeps a kernel running on its own
  // stream.  argv[4]: 4 = the bare MFMA loop above (14 accumulator tiles, ~230 VGPRs),
  // 5 = the same with 6 tiles (~110 VGPRs); with -DWITH_MILAN (needs the experiments
  // build of the library, which exports its launchers) also 1 split16 256x256,
  // 2 split16 256x128x3, 3 fp32 igemm 128x128x2 of libmilan_hip.
  const int neighbour = argc > 4 ? atoi(argv[4]) : 0;
  std::atomic<bool> stop{false};
  std::thread load;
  if (neighbour >= 4) {
    load = std::thread([&] {
      hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
      float* o; CK(hipMalloc((void**)&o, (size_t)1024 * 512 * 4));
      const size_t lds = 128 * 1024;
      if (neighbour == 4) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_aggressor<14>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      else CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_aggressor<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      while (!stop.load()) {
        for (int i = 0; i < 8; ++i) {
          if (neighbour == 4) hipLaunchKernelGGL(mfma_aggressor<14>, dim3(1024), dim3(512), lds, s2, o, 400);
          else hipLaunchKernelGGL(mfma_aggressor<6>, dim3(1024), dim3(512), lds, s2, o, 900);
        }
        CK(hipStreamSynchronize(s2));
      }
    });
  }
#ifdef WITH_MILAN
  else if (neighbour) {
    load = std::thread([&] {
      using namespace milan;
      hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
      const int M = 1881600 / 4, N = 256, K = 2304;
      float *A, *W, *C, *zero;
      CK(hipMalloc((void**)&A, (size_t)M * K * 4)); CK(hipMalloc((void**)&W, (size_t)N * K * 4));
      CK(hipMalloc((void**)&C, (size_t)M * N * 4)); CK(hipMalloc((void**)&zero, 256));
      CK(hipMemsetAsync(A, 0x3c, (size_t)M * K * 4, s2)); CK(hipMemsetAsync(W, 0x3c, (size_t)N * K * 4, s2));
      CK(hipMemsetAsync(zero, 0, 256, s2));
      GemmArgs g = linear_args(A, K, W, nullptr, C, N, M, N, K, EPI_BIAS_RELU, zero);
      if (neighbour != 3) { g.a_split = 1; g.out_split = 1; g.acc_scale = 1.f; }
      if (neighbour == 2) g.tile_hint = 4;
      while (!stop.load()) { for (int i = 0; i < 8; ++i) launch_gemm(g, s2); CK(hipStreamSynchronize(s2)); }
    });
  }
#endif
  // reference: the scalar kernel (no LDS)
  std::vector<float> ref((size_t)rows * F), got((size_t)rows * F);
  hipLaunchKernelGGL(context_scalar, grid, dim3(256), 0, 0, att, feat, rpn, k, F, ctx);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(ref.data(), ctx, ref.size() * 4, hipMemcpyDeviceToHost));
  long iters = 0, bad_iters = 0, shown = 0;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    CK(hipMemset(ctx, 0xff, got.size() * 4));
    for (int rep = 0; rep < 4; ++rep) launch();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got.data(), ctx, got.size() * 4, hipMemcpyDeviceToHost));
    ++iters;
    if (memcmp(got.data(), ref.data(), got.size() * 4) == 0) continue;
    ++bad_iters;
    if (shown++ < 6) {
      long bad = 0, comp[4] = {0, 0, 0, 0}, row16[16] = {0}; long first_row = -1, last_row = -1;
      size_t first = 0;
      for (size_t i = 0; i < got.size(); ++i)
        if (memcmp(&got[i], &ref[i], 4)) {
          if (!bad) first = i;
          ++bad; ++comp[(i % F) % 4]; ++row16[((i % F) / 4 % 256) / 16];
          if (first_row < 0) first_row = i / F;
          last_row = i / F;
        }
      printf("  iter %ld: %ld wrong elements, rows %ld..%ld, components x/y/z/w %ld/%ld/%ld/%ld; first: row %zu col %zu thread %zu got %g want %g; per 16-lane row of the block:",
             iters, bad, first_row, last_row, comp[0], comp[1], comp[2], comp[3], first / F, first % F, (first % F) / 4 % 256, got[first], ref[first]);
      for (int q = 0; q < 16; ++q) printf(" %ld", row16[q]);
      printf("\n");
    }
  }
  stop.store(true);
  if (load.joinable()) load.join();
  printf("variant %d: %ld iterations, %ld with mismatches\n", variant, iters, bad_iters);
  return bad_iters ? 1 : 0;
}
