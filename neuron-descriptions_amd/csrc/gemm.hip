// Implicit-GEMM convolution / linear layer on the gfx950 fp32 matrix cores.
//
// One kernel serves every dense contraction on the MILAN path: the ResNet
// trunk's convolutions (torchvision resnet101, reference call site
// src/milan/encoders.py:298) and every nn.Linear / LSTM gate product of the
// decoder and LM (src/milan/decoders.py:304-323,576-634; src/milan/lms.py:
// 47-56).  Arithmetic is v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate,
// bit-for-bit an fmaf chain over K (MI355X guide §3), so results match the
// reference's fp32 to summation-order round-off.
//
// Tiling (wave64, 4 waves / 256 threads per workgroup):
//   block tile BM x BN, k-tile 32 floats; each wave owns WM x WN made of
//   32x32 MFMA tiles.  A (implicit im2col rows, NHWC so a (kh,kw) tap is a
//   contiguous Cin run) and W ([N][Kp], K contiguous) are streamed
//   HBM/L2 -> LDS with global_load_lds_dwordx4 (16 B/lane, no VGPR staging),
//   double buffered.  LDS rows are 128 B; the 16-B chunk p of row r holds
//   k-chunk p ^ ((r>>1)&7) (swizzle applied on the per-lane SOURCE address,
//   undone on the ds_read_b128), which makes both the lane-linear DMA write
//   and the row-per-lane fragment read bank-conflict free.
//   A fragment read: lane l takes 4 consecutive k of row (l&31); lanes 0-31
//   take k-chunk 2g, lanes 32-63 chunk 2g+1, so one ds_read_b128 per operand
//   feeds four K=2 MFMAs.
//   Workgroup -> tile mapping is XCD-aware: the 8 XCDs each walk a contiguous
//   range of tiles (n fastest), so the blocks sharing an A row-panel hit the
//   same 4 MiB L2.
#include "common.h"
#include <type_traits>
#include <utility>
#include <vector>

namespace milan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

struct RowInfo {
  const float* base;
  int hi0, wi0;
};

template <int BM, int BN, int WM, int WN, bool CIN32>
__global__ __launch_bounds__(256) void igemm_f32_kernel(GemmArgs g, int tiles_m,
                                                         int tiles_n) {
  constexpr int BK = 32;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * WAVES_N == 4, "4 waves per workgroup");
  constexpr int A_ITERS = BM / 32, B_ITERS = BN / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][BM*32]
  float* Bs = smem + 2 * BM * BK;   // [2][BN*32]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware bijective remap (block b runs on XCD b % 8).
  int tile;
  {
    const int T = tiles_m * tiles_n;
    const int b = blockIdx.x;
    const int q = T >> 3, r = T & 7;
    const int xcd = b & 7, idx = b >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;

  // ---- loader setup ---------------------------------------------------------
  const int lrow = tid >> 3;                       // 0..31
  const int kc = (tid & 7) ^ ((tid >> 4) & 7);     // swizzled source chunk
  RowInfo ra[A_ITERS];
  const int HoWo = g.Ho * g.Wo;
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    int m = tile_m * BM + it * 32 + lrow;
    m = m < g.M ? m : g.M - 1;
    const int img = m / HoWo;
    const int rem = m - img * HoWo;
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    ra[it].base = g.A + (long)img * g.a_img_stride;
    ra[it].hi0 = ho * g.stride - g.pad;
    ra[it].wi0 = wo * g.stride - g.pad;
  }
  const float* rb[B_ITERS];
#pragma unroll
  for (int it = 0; it < B_ITERS; ++it) {
    int n = tile_n * BN + it * 32 + lrow;
    n = n < g.N ? n : g.N - 1;
    rb[it] = g.W + (long)n * g.Kp + kc * 4;
  }

  auto stage = [&](int buf, int kt) {
    int kh, kw, cin;
    bool kvalid = true;
    if constexpr (CIN32) {
      const int kbase = kt * BK;
      const int tap = kbase / g.Cin;  // wave-uniform
      cin = kbase - tap * g.Cin + kc * 4;
      kh = tap / g.KW;
      kw = tap - kh * g.KW;
    } else {
      const int k0 = kt * BK + kc * 4;
      const int tap = k0 / g.Cin;
      cin = k0 - tap * g.Cin;
      kh = tap / g.KW;
      kw = tap - kh * g.KW;
      kvalid = k0 < g.K;
    }
    float* adst = As + buf * (BM * BK) + wave * (8 * BK);
    float* bdst = Bs + buf * (BN * BK) + wave * (8 * BK);
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      const int hi = ra[it].hi0 + kh, wi = ra[it].wi0 + kw;
      const bool inb = kvalid && hi >= 0 && hi < g.H && wi >= 0 && wi < g.Wd;
      const float* src =
          inb ? ra[it].base + ((long)hi * g.Wd + wi) * g.a_pix_stride + cin
              : g.zero;
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src,
                                       (LDS_AS void*)(adst + it * (32 * BK)),
                                       16, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      __builtin_amdgcn_global_load_lds(
          (const GLOBAL_AS void*)(rb[it] + kt * BK),
          (LDS_AS void*)(bdst + it * (32 * BK)), 16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.Kp / BK;
  // fragment read offsets (floats) within a buffer, for g = 0
  const int frow = lane & 31, fhalf = lane >> 5;
  int aoff[TM], boff[TN], aswz[TM], bswz[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * WM + i * 32 + frow;
    aoff[i] = row * BK;
    aswz[i] = (row >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * WN + j * 32 + frow;
    boff[j] = row * BK;
    bswz[j] = (row >> 1) & 7;
  }

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const float* Ab = As + cur * (BM * BK);
    const float* Bb = Bs + cur * (BN * BK);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int chunk = 2 * gq + fhalf;
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const f32x4*>(Ab + aoff[i] +
                                               ((chunk ^ aswz[i]) << 2));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const f32x4*>(Bb + boff[j] +
                                               ((chunk ^ bswz[j]) << 2));
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------
  // D layout (32x32): col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  auto epilogue = [&](auto epi_tag) {
    constexpr int EPI = decltype(epi_tag)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = tile_n * BN + wn * WN + j * 32 + (lane & 31);
      if (n >= g.N) continue;
      const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mbase = tile_m * BM + wm * WM + i * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2);
          if (m >= g.M) continue;
          float v = acc[i][j][r] + bias;
          if constexpr (EPI == EPI_BIAS_RELU) v = fmaxf(v, 0.f);
          if constexpr (EPI == EPI_BIAS_RES_RELU)
            v = fmaxf(v + g.aux[(long)m * g.ldaux + n], 0.f);
          if constexpr (EPI == EPI_BIAS_ADD) v = v + g.aux[(long)m * g.ldaux + n];
          if constexpr (EPI == EPI_BIAS_TANH) v = tanhf(v);
          if constexpr (EPI == EPI_BIAS_SIGMUL)
            v = (1.f / (1.f + expf(-v))) * g.aux[(long)m * g.ldaux + n];
          g.C[(long)m * g.ldc + n] = v;
        }
      }
    }
  };
  // Vectorised form: each wave transposes its accumulators through its own
  // 8 KB slice of the (now idle) LDS, 32 rows x 64 cols at a time, and then
  // moves whole 256-B row segments with 16-B per-lane loads/stores (residual
  // read + output write are the HBM-bound part of the small-K 1x1 convs).
  auto epilogue_vec = [&](auto epi_tag) {
    constexpr int EPI = decltype(epi_tag)::value;
    static_assert(WN == 64, "vector epilogue assumes 64-wide wave tiles");
    float* stage = smem + wave * (32 * 64);
    const int col4 = (lane & 15) * 4;
    const int n = tile_n * BN + wn * WN + col4;
    const bool n_ok = n < g.N;  // N % 4 == 0 guaranteed by the caller
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (g.bias && n_ok) bias4 = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          stage[row * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
        }
      // same-wave LDS ops complete in order: no barrier needed
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + (lane >> 4);
        const int m = tile_m * BM + wm * WM + i * 32 + row;
        f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * 64 + col4);
        if (m < g.M && n_ok) {
          v += bias4;
          if constexpr (EPI == EPI_BIAS_RES_RELU || EPI == EPI_BIAS_ADD ||
                        EPI == EPI_BIAS_SIGMUL) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(
                g.aux + (long)m * g.ldaux + n);
            if constexpr (EPI == EPI_BIAS_SIGMUL) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                v[e] = (1.f / (1.f + expf(-v[e]))) * a[e];
            } else {
              v += a;
            }
          }
          if constexpr (EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_RES_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if constexpr (EPI == EPI_BIAS_TANH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
          }
          *reinterpret_cast<f32x4*>(g.C + (long)m * g.ldc + n) = v;
        }
      }
    }
  };
  const bool vec_ok =
      (g.N % 4 == 0) && (g.ldc % 4 == 0) &&
      ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0) &&
      (g.bias == nullptr || (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0) &&
      (g.aux == nullptr ||
       ((g.ldaux % 4 == 0) && (reinterpret_cast<uintptr_t>(g.aux) & 15) == 0));
  if (vec_ok) {
    switch (g.epilogue) {
      case EPI_BIAS_RELU:
        epilogue_vec(std::integral_constant<int, EPI_BIAS_RELU>{});
        break;
      case EPI_BIAS_RES_RELU:
        epilogue_vec(std::integral_constant<int, EPI_BIAS_RES_RELU>{});
        break;
      case EPI_BIAS_TANH:
        epilogue_vec(std::integral_constant<int, EPI_BIAS_TANH>{});
        break;
      case EPI_BIAS_SIGMUL:
        epilogue_vec(std::integral_constant<int, EPI_BIAS_SIGMUL>{});
        break;
      case EPI_BIAS_ADD:
        epilogue_vec(std::integral_constant<int, EPI_BIAS_ADD>{});
        break;
      default:
        epilogue_vec(std::integral_constant<int, EPI_BIAS>{});
        break;
    }
    return;
  }
  switch (g.epilogue) {
    case EPI_BIAS_RELU:
      epilogue(std::integral_constant<int, EPI_BIAS_RELU>{});
      break;
    case EPI_BIAS_RES_RELU:
      epilogue(std::integral_constant<int, EPI_BIAS_RES_RELU>{});
      break;
    case EPI_BIAS_TANH:
      epilogue(std::integral_constant<int, EPI_BIAS_TANH>{});
      break;
    case EPI_BIAS_SIGMUL:
      epilogue(std::integral_constant<int, EPI_BIAS_SIGMUL>{});
      break;
    case EPI_BIAS_ADD:
      epilogue(std::integral_constant<int, EPI_BIAS_ADD>{});
      break;
    default:
      epilogue(std::integral_constant<int, EPI_BIAS>{});
      break;
  }
}

template <int BM, int BN, int WM, int WN, bool CIN32>
static int launch_cfg(const GemmArgs& g, hipStream_t s) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const size_t lds = size_t(2) * (BM + BN) * 32 * sizeof(float);
  auto kern = igemm_f32_kernel<BM, BN, WM, WN, CIN32>;
  static bool attr_set = false;
  if (!attr_set) {
    MILAN_CHECK_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(kern),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, s, g,
                     tiles_m, tiles_n);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- live kernel timing (bench.py's roofline leg) ---------------------------
// HIP events are recorded on the launch stream right before/after each GEMM
// launch while profiling is enabled; durations are read back after a sync.
struct GemmProfiler {
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  size_t used = 0;
  double flops = 0.0;
};
static GemmProfiler g_prof;

int gemm_profile_enable(int enable) {
  g_prof.on = enable != 0;
  g_prof.used = 0;
  g_prof.flops = 0.0;
  return 0;
}

int gemm_profile_read(double* ms, double* flops, long long* launches) {
  MILAN_CHECK_HIP(hipDeviceSynchronize());
  double total = 0.0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    float t = 0.f;
    MILAN_CHECK_HIP(hipEventElapsedTime(&t, g_prof.ev[i].first, g_prof.ev[i].second));
    total += t;
  }
  if (ms) *ms = total;
  if (flops) *flops = g_prof.flops;
  if (launches) *launches = (long long)g_prof.used;
  return 0;
}

static int launch_gemm_impl(const GemmArgs& g, hipStream_t s) {
  const bool cin32 = (g.Cin % 32 == 0);
  if (g.N <= 64) {
    return cin32 ? launch_cfg<256, 64, 64, 64, true>(g, s)
                 : launch_cfg<256, 64, 64, 64, false>(g, s);
  }
  return cin32 ? launch_cfg<128, 128, 64, 64, true>(g, s)
               : launch_cfg<128, 128, 64, 64, false>(g, s);
}

int launch_gemm(const GemmArgs& g, hipStream_t s) {
  MILAN_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, MILAN_ERR_SHAPE,
                "gemm: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
  MILAN_REQUIRE(g.Cin % 4 == 0 && g.a_pix_stride % 4 == 0 &&
                    g.a_img_stride % 4 == 0 && g.Kp % 32 == 0,
                MILAN_ERR_SHAPE,
                "gemm: Cin=%d / strides must be multiples of 4 floats", g.Cin);
  if (!g_prof.on) return launch_gemm_impl(g, s);
  if (g_prof.used == g_prof.ev.size()) {
    hipEvent_t a, b;
    MILAN_CHECK_HIP(hipEventCreate(&a));
    MILAN_CHECK_HIP(hipEventCreate(&b));
    g_prof.ev.emplace_back(a, b);
  }
  auto& e = g_prof.ev[g_prof.used++];
  MILAN_CHECK_HIP(hipEventRecord(e.first, s));
  const int r = launch_gemm_impl(g, s);
  MILAN_CHECK_HIP(hipEventRecord(e.second, s));
  g_prof.flops += 2.0 * (double)g.M * (double)g.N * (double)(g.flop_k > 0 ? g.flop_k : g.K);
  return r;
}

}  // namespace milan
