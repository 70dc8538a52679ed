// Implicit-GEMM convolution / linear layer on the gfx950 matrix cores.
//
// One kernel family serves every dense contraction on the MILAN path: the
// ResNet trunk's convolutions (torchvision resnet101, reference call site
// src/milan/encoders.py:298) and every nn.Linear / LSTM gate product of the
// decoder and LM (src/milan/decoders.py:304-323,576-634; src/milan/lms.py:
// 47-56).  Two arithmetic modes share the tiling, the loader and the epilogue:
//
//   F32   v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-for-bit an fmaf
//         chain over K (MI355X guide section 3).  157 TFLOP/s roof.  Differs
//         from the reference's fp32 only by summation order.
//   SPLIT v_mfma_f32_32x32x16_f16 on operands stored as (hi, lo) f16 pairs:
//         x = hi + lo with hi = f16(x), lo = f16(x - hi) (22 significant bits).
//         A.B ~= Ah.Bh + (Ah.Bl + Al.Bh), three f16 MFMAs with exact products
//         and f32 accumulation (the 2^-22-relative Al.Bl term is dropped); the
//         cross terms get their own accumulator so they are not rounded into
//         the large sum one at a time.  Error is of fp32-GEMM class (measured
//         in tests/test_gpu_parity.py), rate is 1/3 of the f16 MFMA roof =
//         833 TFLOP/s-equivalent, 5.3x the F32 mode.
//
// Split storage format ("f16x2-interleaved"): a row of C channels occupies the
// same 4*C bytes as fp32; every 8 channels form a 32-byte group [hi x8 | lo x8].
// A 16-byte chunk is therefore 4 fp32 channels in F32 mode or one half-group
// in SPLIT mode, the byte addressing is identical, and the HBM->LDS loader
// below does not know which mode it is feeding.
//
// Tiling (wave64, 4 waves / 256 threads per workgroup):
//   block tile BM x BN, k-tile = 32 channel slots (128 B per row); each wave
//   owns 64 x 64 = 2 x 2 MFMA 32x32 tiles.  A (implicit im2col rows, NHWC so
//   a (kh,kw) tap is a contiguous Cin run) and W ([N][Kp], K contiguous) go
//   HBM/L2 -> LDS with global_load_lds_dwordx4 (16 B/lane, no VGPR staging),
//   double buffered.  The 16-B chunk p of LDS row r holds k-chunk
//   p ^ ((r>>1)&7): the swizzle is applied on the per-lane SOURCE address
//   (the DMA destination is lane-linear) and undone on the ds_read_b128, which
//   makes both the DMA write and the row-per-lane fragment read conflict free.
//   Workgroup -> tile mapping is XCD-aware: each of the 8 XCDs walks a
//   contiguous range of tiles (n fastest), so blocks sharing an A row panel
//   share one 4 MiB L2.
//   Epilogue: accumulators are transposed through the (idle) LDS so that HBM
//   sees 16-byte per-lane accesses covering whole 256-B row segments.
#include "common.h"

#include <map>
#include <mutex>
#include <utility>

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

#ifndef MILAN_ABLATE_BUILD
#define MILAN_ABLATE_BUILD 0
#endif
// Kernels and knobs that were measured and LOST (the LDS-strip 3x3 conv, the
// two-problem "pair" launch, issue-slot shaping, forced tile configurations) are
// compiled only into an experiments build (make EXPERIMENTS=1); the product library
// carries the winning configuration per layer shape and nothing else.  Table of knobs:
// DESIGN.md section 5.
#ifndef MILAN_EXPERIMENTS
#define MILAN_EXPERIMENTS 0
#endif

namespace milan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// ---- per-device launch state (a process may drive several GPUs: ADVICE r3) ------------
// Opt a kernel in to `bytes` of dynamic LDS on the CURRENT device, once per (kernel, device).
int ensure_lds_attr(const void* kern, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> done;
  int dev = 0;
  MILAN_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  int& have = done[{kern, dev}];
  if (have < bytes) {
    MILAN_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    have = bytes;
  }
  return 0;
}
// CUs of the current device, rounded down to whole XCD octets (persistent grids)
int device_cus8(int* out) {
  static std::mutex mu;
  static std::map<int, int> cus;
  int dev = 0;
  MILAN_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  int& n = cus[dev];
  if (!n) {
    MILAN_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    n = n < 8 ? 8 : n / 8 * 8;
  }
  *out = n;
  return 0;
}

struct RowInfo {
  const float* base;
  int hi0, wi0;
};

__device__ inline f16x8 as_f16x8(f32x4 v) {
  return __builtin_bit_cast(f16x8, v);
}

// 8 fp32 <-> (hi, lo) f16x8 pair: common.h (split8_rne / join8_exact)
__device__ inline void split8(const float* v, f32x4* hi_out, f32x4* lo_out, float* sat = nullptr) {
  split8_rne(v, hi_out, lo_out, sat);
}
__device__ inline void join8(f32x4 hi, f32x4 lo, float* v) { join8_exact(hi, lo, v); }

// ---------------------------------------------------------------------------
// epilogue shared by all tile configurations (wave tile = TM x TN MFMA tiles,
// rows row0.., columns col0..; TN == 2, i.e. 64 columns per wave)
// ---------------------------------------------------------------------------
// max / sum over the 16 lanes of a DPP row, result in every lane: quad xor 1, quad xor 2, mirror
// inside each half row, mirror of the row -- four VALU instructions (a ds_bpermute butterfly is
// four dependent LDS round trips).  s_nop: a DPP operand written by the previous VALU instruction
// needs two wait states, which the assembler does not insert inside inline asm.
__device__ __forceinline__ float row16_max(float x) {
  asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
               : "+v"(x));
  return x;
}
__device__ __forceinline__ float row16_sum(float x) {
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
               : "+v"(x));
  return x;
}

template <int TM, int TN, bool SWZ64 = false>
__device__ __forceinline__ void run_epilogue(const GemmArgs& g,
                                             f32x16 (&acc)[TM][TN], float* smem,
                                             int wave, int lane, int row0,
                                             int col0) {
  static_assert(TN == 2, "epilogues assume 64-column wave tiles");
  // ---- epilogues ----------------------------------------------------------------
  // D layout (32x32): col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // Staging rows: 64 floats + 4 of padding (the padding staggers consecutive rows over the
  // banks), or -- SWZ64, 8 KB per wave: the persistent ping-pong kernel has exactly 64 KB of
  // LDS to spare while its ring holds the next tile's prefetch -- 64 floats with the 16-byte
  // chunk index XORed by the row's parity, which keeps the two rows of a split-format
  // 16-lane read group on disjoint banks.
  constexpr int SROW = SWZ64 ? 64 : 68;
  auto sw = [](int row, int col) { return SWZ64 ? (col ^ ((row & 1) << 2)) : col; };
  float* stage_out = smem + wave * (32 * SROW);
  auto to_stage = [&](int i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        stage_out[row * SROW + sw(row, j * 32 + (lane & 31))] = acc[i][j][r];
      }
    // same-wave LDS ops complete in order: no barrier needed
  };

  // (1) fp32 out, fp32 aux, float4 per lane.
  auto epilogue_vec4 = [&](auto epi_tag) {
    constexpr int EPI = decltype(epi_tag)::value;
    const int col4 = (lane & 15) * 4;
    const int n = col0 + col4;
    const bool n_ok = n < g.N;  // N % 4 == 0 checked by the launcher
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (g.bias && n_ok) bias4 = *reinterpret_cast<const f32x4*>(g.bias + n);
    // aux rows are fetched one row-tile ahead of the stores (see the split8
    // epilogue below for why)
    constexpr bool AUX = (EPI == EPI_BIAS_RES_RELU || EPI == EPI_BIAS_ADD ||
                          EPI == EPI_BIAS_SIGMUL);
    f32x4 res[AUX ? 2 : 1][8];
    auto load_res = [&](int i, f32x4 (&r)[8]) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int m = row0 + i * 32 + it * 4 + (lane >> 4);
        r[it] = (m < g.M && n_ok)
                    ? *reinterpret_cast<const f32x4*>(g.aux + (long)m * g.ldaux + n)
                    : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    if constexpr (AUX) load_res(0, res[0]);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (AUX) {
        if (i + 1 < TM) load_res(i + 1, res[(i + 1) & 1]);
      }
      to_stage(i);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + (lane >> 4);
        const int m = row0 + i * 32 + row;
        f32x4 v = *reinterpret_cast<const f32x4*>(stage_out + row * SROW + sw(row, col4));
        if (m < g.M && n_ok) {
          v += bias4;
          if constexpr (AUX) {
            const f32x4 a = res[i & 1][it];
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

  // (1b) EPI_LSE: the row statistics of a log-softmax instead of the 4 N bytes per row of x.
  // The 16 lanes that share a staged row cover this wave's 64 columns: maximum and sum of
  // exponentials are reduced over them (xor 1, 2, 4, 8 stays inside the 16-lane group).
  auto epilogue_lse = [&]() {
    const int col4 = (lane & 15) * 4;
    const int n = col0 + col4;
    const bool n_ok = n < g.N;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (g.bias && n_ok) bias4 = *reinterpret_cast<const f32x4*>(g.bias + n);
    const int cb = col0 >> 6;
    // target columns one row-tile ahead of the stores (C and lse_tgt are not provably distinct:
    // a load behind a store would wait for it -- see epilogue_split8)
    int tg[2][8];
    auto load_tgt = [&](int i, int (&t)[8]) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int m = row0 + i * 32 + it * 4 + (lane >> 4);
        t[it] = m < g.M ? (int)g.lse_tgt[(long)m * g.lse_tgt_stride] : -1;
      }
    };
    load_tgt(0, tg[0]);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (i + 1 < TM) load_tgt(i + 1, tg[(i + 1) & 1]);
      to_stage(i);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + (lane >> 4);
        const int m = row0 + i * 32 + row;
        f32x4 v = *reinterpret_cast<const f32x4*>(stage_out + row * SROW + sw(row, col4));
        v += bias4;
        const bool ok = m < g.M && n_ok;
        float mx = ok ? fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])) : -INFINITY;
        mx = row16_max(mx);
        // (exp on v_exp_f32: x - mx <= 0, the error of the sum is that of its largest terms)
        float e = ok ? (__expf(v[0] - mx) + __expf(v[1] - mx)) + (__expf(v[2] - mx) + __expf(v[3] - mx)) : 0.f;
        e = row16_sum(e);
        if (m < g.M && col0 < g.N) {
          if ((lane & 15) == 0) {
            float* d = g.C + (long)m * g.ldc + 2 * cb;
            d[0] = mx;
            d[1] = e;
          }
          const int tgt = tg[i & 1][it];
          if (n_ok && tgt >= n && tgt < n + 4) {
            const int q = tgt - n;
            g.lse_x[m] = q == 0 ? v[0] : (q == 1 ? v[1] : (q == 2 ? v[2] : v[3]));
          }
        }
      }
    }
  };

  // (2) split-format out (and aux): 8 channels = one 32-B group per lane.
  auto epilogue_split8 = [&](auto epi_tag) {
    constexpr int EPI = decltype(epi_tag)::value;
    const int col8 = (lane & 7) * 8;
    const int n = col0 + col8;
    const bool n_ok = n < g.N;  // N % 8 == 0 checked by the launcher
    float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (g.bias && n_ok) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.bias + n);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(g.bias + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
    }
    // Residual groups are fetched one row-tile ahead, before any store of the
    // current one: C and aux are not provably distinct, so without the explicit
    // early loads every load would wait behind the previous store and the
    // epilogue would pay one HBM round trip per 8 rows.
    // RES: split-format residual; SIGMUL: fp32 multiplicand (the decoder's
    // gated context, written straight into the LSTM's split-format input)
    constexpr bool SIG = EPI == EPI_BIAS_SIGMUL;
    constexpr bool RES = (EPI == EPI_BIAS_RES_RELU || EPI == EPI_BIAS_ADD || SIG);
    float sat = 0.f;  // running max of the clamped magnitudes (common.h: saturation is loud)
    f32x4 res[RES ? 2 : 1][4][2];
    auto load_res = [&](int i, f32x4 (&r)[4][2]) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = row0 + i * 32 + it * 8 + (lane >> 3);
        if (m < g.M && n_ok) {
          const float* ap = g.aux + (long)m * g.ldaux + n;
          r[it][0] = *reinterpret_cast<const f32x4*>(ap);
          r[it][1] = *reinterpret_cast<const f32x4*>(ap + 4);
        } else {
          r[it][0] = f32x4{0.f, 0.f, 0.f, 0.f};
          r[it][1] = r[it][0];
        }
      }
    };
    if constexpr (RES) load_res(0, res[0]);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (RES) {
        if (i + 1 < TM) load_res(i + 1, res[(i + 1) & 1]);
      }
      to_stage(i);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3);
        const int m = row0 + i * 32 + row;
        const f32x4 v0 =
            *reinterpret_cast<const f32x4*>(stage_out + row * SROW + sw(row, col8));
        const f32x4 v1 =
            *reinterpret_cast<const f32x4*>(stage_out + row * SROW + sw(row, col8 + 4));
        if (m < g.M && n_ok) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = v0[e] + bias8[e];
            v[4 + e] = v1[e] + bias8[4 + e];
          }
          if constexpr (SIG) {
            const f32x4 a0 = res[i & 1][it][0], a1 = res[i & 1][it][1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = (1.f / (1.f + expf(-v[e]))) * a0[e];
              v[4 + e] = (1.f / (1.f + expf(-v[4 + e]))) * a1[e];
            }
          } else if constexpr (RES) {
            float a[8];
            join8(res[i & 1][it][0], res[i & 1][it][1], a);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += a[e];
          }
          f32x4 hi, lo;
          if constexpr (EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_RES_RELU)
            split8_relu_rne(v, &hi, &lo, &sat);   // (ReLU folded into the clamp)
          else
            split8(v, &hi, &lo, &sat);
          float* cp = g.C + (long)m * g.ldc + n;
          *reinterpret_cast<f32x4*>(cp) = hi;
          *reinterpret_cast<f32x4*>(cp + 4) = lo;
        }
      }
    }
    report_saturation(g.status, sat);
  };

  // (2b) fast mode: plain f16 out (and aux), 8 channels = 16 bytes per lane
  auto epilogue_f16 = [&](auto epi_tag) {
    constexpr int EPI = decltype(epi_tag)::value;
    typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
    const int col8 = (lane & 7) * 8;
    const int n = col0 + col8;
    const bool n_ok = n < g.N;
    float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (g.bias && n_ok) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.bias + n);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(g.bias + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
    }
    constexpr bool RES = EPI == EPI_BIAS_RES_RELU;
    float sat = 0.f;
    f32x4 res[RES ? 2 : 1][4];
    auto load_res = [&](int i, f32x4 (&r)[4]) {   // one row-tile ahead (see epilogue_split8)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = row0 + i * 32 + it * 8 + (lane >> 3);
        r[it] = (m < g.M && n_ok)
                    ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(g.aux) +
                                                      ((long)m * g.ldaux * 4 + n * 2))
                    : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    if constexpr (RES) load_res(0, res[0]);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (RES) {
        if (i + 1 < TM) load_res(i + 1, res[(i + 1) & 1]);
      }
      to_stage(i);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3);
        const int m = row0 + i * 32 + row;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(stage_out + row * SROW + sw(row, col8));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(stage_out + row * SROW + sw(row, col8 + 4));
        if (m < g.M && n_ok) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = v0[e] + bias8[e];
            v[4 + e] = v1[e] + bias8[4 + e];
          }
          if constexpr (RES) {
            const f16x8v a = __builtin_bit_cast(f16x8v, res[i & 1][it]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)a[e];
          }
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if constexpr (EPI == EPI_BIAS)
              asm("v_med3_f32 %0, %1, %2, %3" : "=v"(x[e]) : "v"(v[e]), "v"(-65504.f), "v"(65504.f));
            else
              asm("v_med3_f32 %0, %1, 0, %2" : "=v"(x[e]) : "v"(v[e]), "v"(65504.f));
          }
          sat = sat_fold8(x, sat);
          f32x4 o;
#pragma unroll
          for (int d = 0; d < 4; ++d) o[d] = cvt_pk_f16(x[2 * d], x[2 * d + 1]);
          *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(g.C) + ((long)m * g.ldc * 4 + n * 2)) = o;
        }
      }
    }
    report_saturation(g.status, sat);
  };

  // (4) LSTM cell on gate-interleaved columns: the wave's 64 columns are
  // [i x16 | f x16 | g x16 | o x16] of hidden units col0/4 .. col0/4 + 15; a lane
  // takes the four gates of 4 consecutive units from the staged tile.
  auto epilogue_lstm = [&]() {
    const int q4 = (lane & 3) * 4;
    const int u = (col0 >> 2) + q4;  // first of this lane's 4 hidden units
    const bool n_ok = col0 < g.N;
    f32x4 bias4[4];
#pragma unroll
    for (int gate = 0; gate < 4; ++gate)
      bias4[gate] = (g.bias && n_ok)
                        ? *reinterpret_cast<const f32x4*>(g.bias + col0 + gate * 16 + q4)
                        : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 cin[2][2];
    auto load_c = [&](int i, f32x4 (&c)[2]) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int m = row0 + i * 32 + it * 16 + (lane >> 2);
        c[it] = (m < g.M && n_ok)
                    ? *reinterpret_cast<const f32x4*>(g.aux + (long)m * g.ldaux + u)
                    : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    load_c(0, cin[0]);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (i + 1 < TM) load_c(i + 1, cin[(i + 1) & 1]);
      to_stage(i);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = it * 16 + (lane >> 2);
        const int m = row0 + i * 32 + row;
        f32x4 p[4];
#pragma unroll
        for (int gate = 0; gate < 4; ++gate)
          p[gate] = *reinterpret_cast<const f32x4*>(stage_out + row * SROW +
                                                    sw(row, gate * 16 + q4)) + bias4[gate];
        if (m < g.M && n_ok) {
          f32x4 h4, c4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float hh, cc;
            lstm_cell(p[0][e], p[1][e], p[2][e], p[3][e], cin[i & 1][it][e], &hh, &cc);
            h4[e] = hh; c4[e] = cc;
          }
          *reinterpret_cast<f32x4*>(g.C + (long)m * g.ldc + u) = h4;
          *reinterpret_cast<f32x4*>(g.C2 + (long)m * g.ldc + u) = c4;
          if (g.Cs) {
            // split format: 8 channels = 32 B [hi x8 | lo x8]; this lane owns
            // one half (4 channels) of its group
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float x = fminf(fmaxf(h4[e], -65504.f), 65504.f);
              const _Float16 hh = (_Float16)x;
              hi[e] = hh;
              lo[e] = (_Float16)(x - (float)hh);
            }
            char* base = reinterpret_cast<char*>(g.Cs + (long)m * g.ldc) +
                         (u >> 3) * 32 + ((u >> 2) & 1) * 8;
            *reinterpret_cast<f16x4*>(base) = hi;
            *reinterpret_cast<f16x4*>(base + 16) = lo;
          }
        }
      }
    }
  };

  // (3) scalar fallback (unaligned fp32 outputs of odd-sized test models).
  auto epilogue_scalar = [&](auto epi_tag) {
    constexpr int EPI = decltype(epi_tag)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = col0 + j * 32 + (lane & 31);
      if (n >= g.N) continue;
      const float bias = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mbase = row0 + i * 32 + 4 * (lane >> 5);
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

  auto dispatch = [&](auto&& fn) {
    switch (g.epilogue) {
      case EPI_BIAS_RELU: fn(std::integral_constant<int, EPI_BIAS_RELU>{}); break;
      case EPI_BIAS_RES_RELU: fn(std::integral_constant<int, EPI_BIAS_RES_RELU>{}); break;
      case EPI_BIAS_TANH: fn(std::integral_constant<int, EPI_BIAS_TANH>{}); break;
      case EPI_BIAS_SIGMUL: fn(std::integral_constant<int, EPI_BIAS_SIGMUL>{}); break;
      case EPI_BIAS_ADD: fn(std::integral_constant<int, EPI_BIAS_ADD>{}); break;
      default: fn(std::integral_constant<int, EPI_BIAS>{}); break;
    }
  };
  if (g.epilogue == EPI_LSTM) {
    epilogue_lstm();
  } else if (g.epilogue == EPI_LSE) {
    epilogue_lse();
  } else if (g.out_mode == OUT_F16) {
    switch (g.epilogue) {
      case EPI_BIAS_RELU: epilogue_f16(std::integral_constant<int, EPI_BIAS_RELU>{}); break;
      case EPI_BIAS_RES_RELU: epilogue_f16(std::integral_constant<int, EPI_BIAS_RES_RELU>{}); break;
      default: epilogue_f16(std::integral_constant<int, EPI_BIAS>{}); break;
    }
  } else if (g.out_mode == OUT_SPLIT8) {
    // only the conv epilogues exist in split form
    switch (g.epilogue) {
      case EPI_BIAS_RELU: epilogue_split8(std::integral_constant<int, EPI_BIAS_RELU>{}); break;
      case EPI_BIAS_RES_RELU: epilogue_split8(std::integral_constant<int, EPI_BIAS_RES_RELU>{}); break;
      case EPI_BIAS_ADD: epilogue_split8(std::integral_constant<int, EPI_BIAS_ADD>{}); break;
      case EPI_BIAS_SIGMUL: epilogue_split8(std::integral_constant<int, EPI_BIAS_SIGMUL>{}); break;
      default: epilogue_split8(std::integral_constant<int, EPI_BIAS>{}); break;
    }
  } else if (g.out_mode == OUT_VEC4) {
    dispatch(epilogue_vec4);
  } else {
    dispatch(epilogue_scalar);
  }
}

// tiles_m of a launch whose row count lives on the device: the launcher's value is an upper
// bound; workgroups beyond tiles_m(live) x tiles_n exit, xcd_tile() maps over the live count
__device__ __forceinline__ int live_tiles_m(GemmArgs& g, int tiles_m, int BM) {
  if (g.m_live == nullptr) return tiles_m;
  g.M = live_rows(g.m_live, g.m_live_mul, g.M);
  return (g.M + BM - 1) / BM;
}

template <int BM, int BN, int STAGES, bool CIN32, bool SPLIT>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64, 2) void igemm_kernel(
    GemmArgs g, int tiles_m, int tiles_n) {
  constexpr int BK = 32;
  constexpr int WM = 64, WN = 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int WAVES_N = BN / WN;
  constexpr int NW = (BM / WM) * WAVES_N;    // waves per workgroup (4 or 8)
  constexpr int LROWS = NW * 8;              // tile rows covered per loader pass
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");
  static_assert(STAGES == 2 || STAGES == 3, "2- or 3-deep LDS ring");
  constexpr int A_ITERS = BM / LROWS, B_ITERS = BN / LROWS;
  constexpr int LOADS = A_ITERS + B_ITERS;   // DMA instructions per wave per stage
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [STAGES][BM*32]
  float* Bs = smem + STAGES * BM * BK;   // [STAGES][BN*32]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware bijective remap (block b runs on XCD b % 8).
  tiles_m = live_tiles_m(g, tiles_m, BM);
  if ((int)blockIdx.x >= tiles_m * tiles_n) return;
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
  const int lrow = tid >> 3;                       // 0..LROWS-1
  const int kc = (tid & 7) ^ ((tid >> 4) & 7);     // swizzled source chunk
  RowInfo ra[A_ITERS];
  const int HoWo = g.Ho * g.Wo;
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    int m = tile_m * BM + it * LROWS + lrow;
    m = m < g.M ? m : g.M - 1;
    const int img = m / HoWo;
    const int rem = m - img * HoWo;
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    ra[it].base = g.A + (long)img * g.a_img_stride;
    ra[it].hi0 = ho * g.stride - g.pad;
    ra[it].wi0 = g.aniso ? wo * g.stride_w - g.pad_w : wo * g.stride - g.pad;
  }
  const float* rb[B_ITERS];
#pragma unroll
  for (int it = 0; it < B_ITERS; ++it) {
    int n = tile_n * BN + it * LROWS + lrow;
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
                                       (LDS_AS void*)(adst + it * (LROWS * BK)),
                                       16, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      __builtin_amdgcn_global_load_lds(
          (const GLOBAL_AS void*)(rb[it] + kt * BK),
          (LDS_AS void*)(bdst + it * (LROWS * BK)), 16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
  f32x16 accx[SPLIT ? TM : 1][SPLIT ? TN : 1];  // cross terms (SPLIT only)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0.f;
        if constexpr (SPLIT) accx[i][j][r] = 0.f;
      }

  const int nk = g.Kp / BK;
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

  auto compute = [&](int cur) {
    const float* Ab = As + cur * (BM * BK);
    const float* Bb = Bs + cur * (BN * BK);
    if constexpr (!SPLIT) {
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
    } else {
      // two K=16 slabs per k-tile; lanes 0-31 feed channels 0-7 of the slab,
      // lanes 32-63 channels 8-15: group q = 2*slab + half, hi chunk 2q, lo 2q+1
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int chi = 2 * (2 * sl + fhalf), clo = chi + 1;
        f32x4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ah[i] = *reinterpret_cast<const f32x4*>(Ab + aoff[i] +
                                                  ((chi ^ aswz[i]) << 2));
          al[i] = *reinterpret_cast<const f32x4*>(Ab + aoff[i] +
                                                  ((clo ^ aswz[i]) << 2));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[j] = *reinterpret_cast<const f32x4*>(Bb + boff[j] +
                                                  ((chi ^ bswz[j]) << 2));
          bl[j] = *reinterpret_cast<const f32x4*>(Bb + boff[j] +
                                                  ((clo ^ bswz[j]) << 2));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                as_f16x8(ah[i]), as_f16x8(bh[j]), acc[i][j], 0, 0, 0);
            accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                as_f16x8(ah[i]), as_f16x8(bl[j]), accx[i][j], 0, 0, 0);
            accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                as_f16x8(al[i]), as_f16x8(bh[j]), accx[i][j], 0, 0, 0);
          }
      }
    }
  };

  if constexpr (STAGES == 2) {
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
      compute(cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    // 3-deep ring, two k-tiles in flight: the DMA of tile kt+2 is issued
    // before the MFMAs of tile kt, and only tile kt+1 is waited for (counted
    // vmcnt + raw s_barrier, so the newest LOADS stay in flight ACROSS the
    // barrier instead of being drained by a __syncthreads fence).
    stage(0, 0);
    if (nk > 1) {
      stage(1, 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      int nxt2 = cur + 2;
      nxt2 = nxt2 >= 3 ? nxt2 - 3 : nxt2;
      if (kt + 2 < nk && !(g.debug & 2)) stage(nxt2, kt + 2);
      if (!(g.debug & 1)) compute(cur);
      if (kt + 2 < nk) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      cur = cur + 1 == 3 ? 0 : cur + 1;
    }
  }
  if constexpr (SPLIT) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = (acc[i][j] + accx[i][j]) * g.acc_scale;
  }

  run_epilogue<TM, TN>(g, acc, smem, wave, lane, tile_m * BM + wm * WM,
                       tile_n * BN + wn * WN);
}

// ---------------------------------------------------------------------------
// SPLIT-mode large tile: 256 x 256 per workgroup, 8 waves (2 x 4), wave tile
// 128 x 64.  Written for the K-heavy convolutions, where the 256x128 kernel is
// bound by the HBM/L2 -> LDS DMA rate rather than by the matrix cores (MFMA-only
// and DMA-only ablations each took ~half of the combined time): a 256x256 tile
// moves 2/3 of the bytes per MFMA.  k-tile = 16 channel slots (one K=16 MFMA
// slab, 64-byte LDS rows), 5-deep ring with four tiles in flight (all 160 KB of LDS).
// One accumulator set (the hl/lh cross terms are added into the main sum).
// LDS row r holds 4 chunks [hi g0 | lo g0 | hi g1 | lo g1]; chunk p of row r
// stores k-chunk p ^ ((r>>2)&3) (conflict-free for the DMA write and for the
// row-per-lane ds_read_b128).
// ---------------------------------------------------------------------------
// XCD-aware bijective remap (block b runs on XCD b % 8): each XCD walks a
// contiguous range of the T tiles.
__device__ __forceinline__ int xcd_tile(int b, int T) {
  const int q = T >> 3, r = T & 7;
  const int xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int BM, int BN, int STAGES, int SHAPE, int TMW = 4>
__device__ __forceinline__ void split16_tile(const GemmArgs& g, int tile_m,
                                             int tile_n, int tid_in = -1) {
  constexpr int BK = 16;
  constexpr int TM = TMW, TN = 2;             // wave tile (TM*32) x 64: 128 x 64,
                                              // or 64 x 64 for the N = 64 layers
  constexpr int WROWS = TM * 32;
  constexpr int WAVES_N = BN / 64;
  constexpr int NT = (BM / WROWS) * WAVES_N * 64;
  constexpr int LROWS = NT / 4;               // rows per loader pass
  constexpr int A_ITERS = BM / LROWS, B_ITERS = BN / LROWS;
  constexpr int LOADS = A_ITERS + B_ITERS;
  static_assert(STAGES >= 3 && STAGES <= 5, "3- to 5-deep ring");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                        // [STAGES][BM*16]
  float* Bs = smem + STAGES * BM * BK;     // [STAGES][BN*16]

  const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

#if MILAN_EXPERIMENTS
  // phase stagger: every second first-round workgroup of an XCD starts late, so that half
  // of the CUs are in their (HBM-bound) epilogue while the other half multiplies
  if (g.stagger > 0 && blockIdx.x < (NT >= 512 ? 256 : 512) && ((blockIdx.x >> 3) & 3)) {
    // four phases per XCD: 0, 1, 2, 3 x stagger microseconds
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    const unsigned long long dt = (unsigned long long)g.stagger * 100ull * ((blockIdx.x >> 3) & 3);
    while (wall_clock64() - t0 < dt) __builtin_amdgcn_s_sleep(8);
  }
#endif
  // loader: NT threads cover NT/4 rows x 4 chunks per pass
  const int lrow = tid >> 2;                        // 0..LROWS-1
  const int kc = (tid & 3) ^ ((tid >> 4) & 3);      // (row>>2)&3 == (tid>>4)&3
  RowInfo ra[A_ITERS];
  const float* ra2[A_ITERS];
  const int k1 = g.A2 ? g.K1 : 0x7fffffff;
  const int HoWo = g.Ho * g.Wo;
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    int m = tile_m * BM + it * LROWS + lrow;
    m = m < g.M ? m : g.M - 1;
    const int img = m / HoWo;
    const int rem = m - img * HoWo;
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    ra[it].hi0 = ho * g.stride - g.pad;
    ra[it].wi0 = wo * g.stride - g.pad;
    // pointer to the (possibly virtual, never dereferenced when out of the
    // image) pixel (hi0, wi0), at this lane's chunk
    ra[it].base = g.A + (long)img * g.a_img_stride +
                  ((long)ra[it].hi0 * g.Wd + ra[it].wi0) * g.a_pix_stride +
                  kc * 4;
    // second source (1x1, stride2, always in bounds), or the same pointer
    ra2[it] = g.A2 ? g.A2 + (long)img * g.a2_img_stride +
                         ((long)(ho * g.stride2) * g.W2d + wo * g.stride2) *
                             g.a2_pix_stride +
                         kc * 4 - g.K1
                   : ra[it].base;
  }
  const float* rb[B_ITERS];
#pragma unroll
  for (int it = 0; it < B_ITERS; ++it) {
    int n = tile_n * BN + it * LROWS + lrow;
    n = n < g.N ? n : g.N - 1;
    rb[it] = g.W + (long)n * g.Kp + kc * 4;
  }

  // One DMA piece (1 KB per wave) of k-tile kt into ring slot buf: pieces
  // 0..A_ITERS-1 are A rows, the rest W rows.  Branch-free address generation
  // so the pieces can be interleaved with the MFMAs of the current tile.
  const int nk_total = g.Kp / BK;
  // (kh, kw, cin0) of the k-tile that will be issued next, advanced
  // incrementally (no integer divisions in the main loop).
  int is_kh = 0, is_kw = 0, is_cin0 = 0;
  int is_kt = 0;  // k-tile described by (is_kh, is_kw, is_cin0)
  auto advance_tap = [&]() {
    if (is_kt + 1 >= nk_total) return;  // dummy tiles re-read the last tile
    ++is_kt;
    if constexpr (SHAPE == 3) return;
    if constexpr (SHAPE == 2) {
      // chunk-major k order (GemmArgs::chunk_major): the KH x KW taps of a 16-channel chunk
      // in consecutive k-tiles.  A compile-time variant: the extra scalar work of a run-time
      // choice in this loop cost 6 % on every layer.
      if (++is_kw == g.KW) {
        is_kw = 0;
        if (++is_kh == g.KH) { is_kh = 0; is_cin0 += BK; }
      }
    } else {
      is_cin0 += BK;
      if (is_cin0 >= g.Cin) {
        is_cin0 = 0;
        if (++is_kw == g.KW) { is_kw = 0; ++is_kh; }
      }
    }
  };
  // Past the end of K the loader keeps re-reading the last k-tile into ring
  // slots nobody will consume again, so that every iteration issues exactly
  // LOADS pieces (constant vmcnt counts, branch-free loop body that the
  // scheduler can interleave with the MFMAs).
  // timing experiments (-DMILAN_ABLATE_BUILD=1 + MILAN_ABLATE=bits); compiled out
  // of the production library
  const bool abl_dma = MILAN_ABLATE_BUILD && (g.debug & 16);
  const bool abl_bar = MILAN_ABLATE_BUILD && (g.debug & 32);
  const bool abl_mfma = MILAN_ABLATE_BUILD && (g.debug & 64);
  auto issue_piece = [&](int buf, int piece) {
    if (abl_dma) { if (piece == LOADS - 1) advance_tap(); return; }
    if (piece < A_ITERS) {
      const int it = piece;
      if constexpr (SHAPE == 3) {
        // 1x1 / pad 0 / one source (most of the trunk, every Linear): the A row of a
        // k-tile is base + 16 kt -- no tap arithmetic, no bounds test in the loop
        float* adst = As + buf * (BM * BK) + wave * (16 * BK) + it * (LROWS * BK);
        __builtin_amdgcn_global_load_lds(
            (const GLOBAL_AS void*)(ra[it].base + is_kt * BK), (LDS_AS void*)adst, 16, 0, 0);
        return;
      }
      const long toff = ((long)is_kh * g.Wd + is_kw) * g.a_pix_stride + is_cin0;
      const int hi = ra[it].hi0 + is_kh, wi = ra[it].wi0 + is_kw;
      const bool inb = (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.Wd;
      const float* src = inb ? ra[it].base + toff : g.zero;
      if (is_kt * BK >= k1) src = ra2[it] + is_kt * BK;  // wave-uniform
      float* adst = As + buf * (BM * BK) + wave * (16 * BK) + it * (LROWS * BK);
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src,
                                       (LDS_AS void*)adst, 16, 0, 0);
    } else {
      const int it = piece - A_ITERS;
      const float* src = rb[it] + is_kt * BK;
      float* bdst = Bs + buf * (BN * BK) + wave * (16 * BK) + it * (LROWS * BK);
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src,
                                       (LDS_AS void*)bdst, 16, 0, 0);
    }
    if (piece == LOADS - 1) advance_tap();
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int p = 0; p < LOADS; ++p) issue_piece(buf, p);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.Kp / BK;
  const int frow = lane & 31, fhalf = lane >> 5;
  int aoff[TM], boff[TN], aswz[TM], bswz[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * WROWS + i * 32 + frow;
    aoff[i] = row * BK;
    aswz[i] = (row >> 2) & 3;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * 64 + j * 32 + frow;
    boff[j] = row * BK;
    bswz[j] = (row >> 2) & 3;
  }

  // MFMAs of tile `cur`, with the DMA pieces of tile `ktn` (ring slot `nxt`,
  // when do_stage) issued between the MFMA groups so that their issue cost
  // and address arithmetic hide under the matrix pipe.
  // MFMAs of tile `cur`; the DMA pieces of tile `ktn` (ring slot `nxt`) are
  // issued between the MFMA groups and the A fragments of row-tile i+1 are
  // fetched while row-tile i multiplies, so address arithmetic, DMA issue and
  // LDS latency hide under the matrix pipe.
  // DMA_POS: 0 the DMA pieces between the MFMA row groups; 1 all of them ahead of the
  // MFMAs, 2 all behind.  In the 8-wave tile the two waves of a SIMD run 1 and 2 (see
  // igemm_split16_linp_kernel: an LDS-DMA instruction holds its wave for 150-200 cycles;
  // with the same order in both waves the stalls coincide and the matrix pipe idles).
  auto compute = [&](int cur, int nxt, auto pos_tag) {
    constexpr int DMA_POS = decltype(pos_tag)::value;
    const float* Ab = As + cur * (BM * BK);
    const float* Bb = Bs + cur * (BN * BK);
    // lanes 0-31: group 0 (channels 0-7), lanes 32-63: group 1 (channels 8-15)
    const int chi = 2 * fhalf, clo = chi + 1;
    f32x4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[j] = *reinterpret_cast<const f32x4*>(Bb + boff[j] + ((chi ^ bswz[j]) << 2));
      bl[j] = *reinterpret_cast<const f32x4*>(Bb + boff[j] + ((clo ^ bswz[j]) << 2));
    }
    ah[0] = *reinterpret_cast<const f32x4*>(Ab + aoff[0] + ((chi ^ aswz[0]) << 2));
    al[0] = *reinterpret_cast<const f32x4*>(Ab + aoff[0] + ((clo ^ aswz[0]) << 2));
    constexpr int PER = (LOADS + TM - 1) / TM;  // pieces per MFMA group
    if constexpr (DMA_POS == 1) {
#pragma unroll
      for (int q = 0; q < LOADS; ++q) issue_piece(nxt, q);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (i + 1 < TM) {
        ah[i + 1] = *reinterpret_cast<const f32x4*>(Ab + aoff[i + 1] +
                                                    ((chi ^ aswz[i + 1]) << 2));
        al[i + 1] = *reinterpret_cast<const f32x4*>(Ab + aoff[i + 1] +
                                                    ((clo ^ aswz[i + 1]) << 2));
      }
      if constexpr (DMA_POS == 0) {
#pragma unroll
        for (int q = 0; q < PER; ++q)
          if (i * PER + q < LOADS) issue_piece(nxt, i * PER + q);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
            as_f16x8(ah[i]), as_f16x8(bl[j]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
            as_f16x8(al[i]), as_f16x8(bh[j]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
            as_f16x8(ah[i]), as_f16x8(bh[j]), acc[i][j], 0, 0, 0);
      }
    }
    if constexpr (DMA_POS == 2) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < LOADS; ++q) issue_piece(nxt, q);
    }
    if constexpr (MILAN_EXPERIMENTS && SHAPE == 1) {
      // issue-slot shaping: fragments of the first row-tile, then one MFMA
      // followed by a few of the remaining non-MFMA instructions, repeated
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);   // DS_READ
#pragma unroll
      for (int m = 0; m < TM * TN * 3; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // MFMA
        if (m < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); // VALU
        __builtin_amdgcn_sched_group_barrier(0x004, 2, 0); // SALU
        if ((m & 3) == 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // VMEM
      }
    }
  };

#if MILAN_EXPERIMENTS
  long long pt_last = g.prof ? clock64() : 0;
  long long pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define G_STAMP(k) do { if (g.prof) { const long long t1 = clock64(); pt_acc[k] += t1 - pt_last; pt_last = t1; } } while (0)
#else
#define G_STAMP(k) do {} while (0)
#endif
  // prologue: STAGES-1 tiles in flight (dummy ones if K is short), wait for
  // the first
  constexpr int AHEAD = STAGES - 1;
#pragma unroll
  for (int t = 0; t < AHEAD; ++t) stage(t);
  G_STAMP(0);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * LOADS) : "memory");
  __builtin_amdgcn_s_barrier();
  G_STAMP(1);
  int cur = 0;
  const int nk_run = (MILAN_ABLATE_BUILD && (g.debug & 4)) ? 0 : nk;  // epilogue only
  auto k_loop = [&](auto pos_tag) {
    for (int kt = 0; kt < nk_run; ++kt) {
      int nxt = cur + AHEAD;
      nxt = nxt >= STAGES ? nxt - STAGES : nxt;
      if (!abl_mfma) compute(cur, nxt, pos_tag);
      else {
#pragma unroll
        for (int q = 0; q < LOADS; ++q) issue_piece(nxt, q);
      }
      G_STAMP(5);  // (experiments, with GemmArgs::prof: compute / DMA wait / barrier inside the loop)
      // tile kt+1 must have landed; the AHEAD-1 younger tiles stay in flight
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * LOADS) : "memory");
      G_STAMP(6);
      if (!abl_bar) __builtin_amdgcn_s_barrier();
      G_STAMP(7);
      cur = cur + 1 == STAGES ? 0 : cur + 1;
    }
  };
  // (the two waves of a SIMD issuing their pieces at opposite ends of the iteration -- what
  // igemm_split16_linp_kernel does -- measured SLOWER here: layer3 119.8 -> 123.9 ms per 256
  // neurons; this kernel's pieces carry the tap arithmetic and stay between the MFMA groups)
  k_loop(std::integral_constant<int, 0>{});
  G_STAMP(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the dummy tiles
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = acc[i][j] * g.acc_scale;
  G_STAMP(3);

  if (MILAN_ABLATE_BUILD && (g.debug & 8)) {  // main loop only
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 123.456f) g.C[0] = t;
    return;
  }
  run_epilogue<TM, TN>(g, acc, smem, wave, lane, tile_m * BM + wm * WROWS,
                       tile_n * BN + wn * 64);
  G_STAMP(4);
#if MILAN_EXPERIMENTS
  if (g.prof && lane == 0)  // 8 counters per wave: waves 1.. at prof[8 * wave + k]
    for (int k = 0; k < 8; ++k)
      atomicAdd((unsigned long long*)g.prof + 8 * wave + k, (unsigned long long)pt_acc[k]);
#endif
}

template <int BM, int BN, int STAGES, int SHAPE>
__global__ __launch_bounds__((BM / 128) * (BN / 64) * 64, 2) void igemm_split16_kernel(
    GemmArgs g, int tiles_m, int tiles_n) {
  // one tile per workgroup (gridDim.x == tiles), or persistent workgroups walking
  // tiles q = blockIdx.x, + gridDim.x, ... (gridDim.x a multiple of 8: q stays on its XCD)
  tiles_m = live_tiles_m(g, tiles_m, BM);
  const int T = tiles_m * tiles_n;
  for (int q = blockIdx.x; q < T; q += gridDim.x) {
    const int tile = xcd_tile(q, T);
    const int tile_m = tile / tiles_n;
    // opaque per iteration: nothing derived from the thread index is kept live across
    // tiles (the 256 x 256 kernel has no registers to spare)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    split16_tile<BM, BN, STAGES, SHAPE>(g, tile_m, tile - tile_m * tiles_n, tid);
    // the epilogue's staging reads are done before the next tile's DMA lands there
    __builtin_amdgcn_s_barrier();
  }
}

// Persistent form of the 256 x 256 tile for 1x1 convolutions / Linear layers (one A
// source, pad 0): a workgroup walks tiles q = blockIdx.x, + gridDim.x, ... and the ring
// keeps rolling across them.  A tile of the one-launch-per-tile kernel starts with ~4 us
// in which nothing multiplies (16 DMA instructions per wave to fill the ring, then the
// HBM latency of the first A rows: 10 % of an expand-conv tile, in-kernel profile in
// profiles/r3_experiments.txt P).  Here the last AHEAD iterations of a tile's main loop,
// which have nothing left to fetch for it, fetch the A rows of the NEXT tile's first
// AHEAD k-tiles instead -- they land under the epilogue -- and only the W rows (L2 hits)
// are fetched after it.  The epilogue stages through the W half of the ring, which is
// idle by then; the A half holds the prefetch.  Same instruction stream per k-tile, same
// bits as igemm_split16_kernel.
template <int STAGES>
__global__ __launch_bounds__(512, 2) void igemm_split16_linp_kernel(GemmArgs g,
                                                                    int tiles_m,
                                                                    int tiles_n) {
  constexpr int BM = 256, BN = 256, BK = 16, TM = 4, TN = 2;
  constexpr int WROWS = TM * 32, WAVES_N = BN / 64, LROWS = 128;
  constexpr int A_ITERS = BM / LROWS, B_ITERS = BN / LROWS, LOADS = A_ITERS + B_ITERS;
  constexpr int AHEAD = STAGES - 1, NT_WAVES = 8;
  static_assert(STAGES == 5 && LOADS == TM, "one DMA piece per MFMA row group");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                        // [STAGES][BM*16]
  float* Bs = smem + STAGES * BM * BK;     // [STAGES][BN*16]; the epilogue's staging
  const int HoWo = g.Ho * g.Wo;
  tiles_m = live_tiles_m(g, tiles_m, BM);
  const int T = tiles_m * tiles_n;
  const int nk = g.Kp / BK;  // >= AHEAD (launcher)

  // A row pointers of a tile at the lane's 16-byte chunk (k-tile 0)
  auto a_rows = [&](int tile_m, int lrow, int kc, const float* (&ra)[A_ITERS]) {
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      int m = tile_m * BM + it * LROWS + lrow;
      m = m < g.M ? m : g.M - 1;
      const int img = m / HoWo;
      const int rem = m - img * HoWo;
      const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
      ra[it] = g.A + (long)img * g.a_img_stride +
               ((long)(ho * g.stride) * g.Wd + wo * g.stride) * g.a_pix_stride + kc * 4;
    }
  };

  int q = blockIdx.x;
  if (q >= T) return;
  int tile = xcd_tile(q, T);
  int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  const float* ra[A_ITERS];
  int cur = 0;  // ring slot of the tile's k-tile 0
  {
    const int tid0 = threadIdx.x;
    const int wave0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    a_rows(tile_m, tid0 >> 2, (tid0 & 3) ^ ((tid0 >> 4) & 3), ra);
#pragma unroll
    for (int t = 0; t < AHEAD; ++t)
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it)
        __builtin_amdgcn_global_load_lds(
            (const GLOBAL_AS void*)(ra[it] + t * BK),
            (LDS_AS void*)(As + t * (BM * BK) + wave0 * (16 * BK) + it * (LROWS * BK)), 16, 0, 0);
  }

  for (;;) {
    // opaque per tile: nothing derived from the thread index stays live across the
    // epilogue (the kernel has no registers to spare)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lrow = tid >> 2;
    const int kc = (tid & 3) ^ ((tid >> 4) & 3);
    const int frow = lane & 31, fhalf = lane >> 5;
    int aoff[TM], boff[TN], aswz[TM], bswz[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * WROWS + i * 32 + frow;
      aoff[i] = row * BK;
      aswz[i] = (row >> 2) & 3;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn * 64 + j * 32 + frow;
      boff[j] = row * BK;
      bswz[j] = (row >> 2) & 3;
    }
    const float *ra_n[A_ITERS], *rb[B_ITERS];
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      int n = tile_n * BN + it * LROWS + lrow;
      n = n < g.N ? n : g.N - 1;
      rb[it] = g.W + (long)n * g.Kp + kc * 4;
    }
    auto issue_a = [&](const float* const (&rows)[A_ITERS], int it, int kt, int slot) {
      __builtin_amdgcn_global_load_lds(
          (const GLOBAL_AS void*)(rows[it] + kt * BK),
          (LDS_AS void*)(As + slot * (BM * BK) + wave * (16 * BK) + it * (LROWS * BK)), 16, 0, 0);
    };
    auto issue_b = [&](int it, int kt, int slot) {
      __builtin_amdgcn_global_load_lds(
          (const GLOBAL_AS void*)(rb[it] + kt * BK),
          (LDS_AS void*)(Bs + slot * (BN * BK) + wave * (16 * BK) + it * (LROWS * BK)), 16, 0, 0);
    };
    // the W rows of k-tiles 0 .. AHEAD-1 first (the A rows are in flight or landed); the
    // next tile's addresses and the accumulator reset are computed while they fly
    {
      int slot = cur;
#pragma unroll
      for (int t = 0; t < AHEAD; ++t) {
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) issue_b(it, t, slot);
        slot = slot + 1 == STAGES ? 0 : slot + 1;
      }
    }
    const bool has_next = q + (int)gridDim.x < T;
    int ntile_m = tile_m, ntile_n = tile_n;
    if (has_next) {
      const int nt = xcd_tile(q + gridDim.x, T);
      ntile_m = nt / tiles_n;
      ntile_n = nt - ntile_m * tiles_n;
    }
    a_rows(ntile_m, lrow, kc, ra_n);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // MFMAs of the k-tile in ring slot `cur`; one DMA piece between the MFMA groups:
    // STEADY the A and W rows of this tile's k-tile `kt_issue`; otherwise (tail) the A
    // rows of the next tile's k-tile `kt_issue`, if there is a next tile
    // DMA_POS (steady state): 0 one piece between the MFMA row groups, 1 all four pieces
    // ahead of the MFMAs, 2 all four behind them.  The two waves of a SIMD run 1 and 2:
    // an LDS-DMA instruction holds its wave for 150-200 cycles, and with the same
    // instruction order in both waves those stalls coincide and the matrix pipe idles
    // (in-kernel profile: the second wave of each SIMD finished 540 cycles late).
    auto compute = [&](int cur_slot, int nxt_slot, int kt_issue, auto steady_tag, auto pos_tag) {
      constexpr bool STEADY = decltype(steady_tag)::value;
      constexpr int DMA_POS = decltype(pos_tag)::value;
      const float* Ab = As + cur_slot * (BM * BK);
      const float* Bb = Bs + cur_slot * (BN * BK);
      const int chi = 2 * fhalf, clo = chi + 1;
      f32x4 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const f32x4*>(Bb + boff[j] + ((chi ^ bswz[j]) << 2));
        bl[j] = *reinterpret_cast<const f32x4*>(Bb + boff[j] + ((clo ^ bswz[j]) << 2));
      }
      ah[0] = *reinterpret_cast<const f32x4*>(Ab + aoff[0] + ((chi ^ aswz[0]) << 2));
      al[0] = *reinterpret_cast<const f32x4*>(Ab + aoff[0] + ((clo ^ aswz[0]) << 2));
      if constexpr (STEADY && DMA_POS == 1) {
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) issue_a(ra, it, kt_issue, nxt_slot);
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) issue_b(it, kt_issue, nxt_slot);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (i + 1 < TM) {
          ah[i + 1] = *reinterpret_cast<const f32x4*>(Ab + aoff[i + 1] +
                                                      ((chi ^ aswz[i + 1]) << 2));
          al[i + 1] = *reinterpret_cast<const f32x4*>(Ab + aoff[i + 1] +
                                                      ((clo ^ aswz[i + 1]) << 2));
        }
        if constexpr (STEADY) {
          if constexpr (DMA_POS == 0) {
            if (i < A_ITERS) issue_a(ra, i, kt_issue, nxt_slot);
            else issue_b(i - A_ITERS, kt_issue, nxt_slot);
          }
        } else {
          if (i < A_ITERS && has_next) issue_a(ra_n, i, kt_issue, nxt_slot);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
              as_f16x8(ah[i]), as_f16x8(bl[j]), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
              as_f16x8(al[i]), as_f16x8(bh[j]), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
              as_f16x8(ah[i]), as_f16x8(bh[j]), acc[i][j], 0, 0, 0);
        }
      }
      if constexpr (STEADY && DMA_POS == 2) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) issue_a(ra, it, kt_issue, nxt_slot);
#pragma unroll
        for (int it = 0; it < B_ITERS; ++it) issue_b(it, kt_issue, nxt_slot);
      }
    };
    auto advance = [&]() { cur = cur + 1 == STAGES ? 0 : cur + 1; };
    auto slot_ahead = [&]() { const int n = cur + AHEAD; return n >= STAGES ? n - STAGES : n; };

    // steady state: k-tile kt + AHEAD of this tile is issued under k-tile kt
    auto steady_loop = [&](auto pos_tag) {
      for (int kt = 0; kt + AHEAD < nk; ++kt) {
        compute(cur, slot_ahead(), kt + AHEAD, std::true_type{}, pos_tag);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * LOADS) : "memory");
        __builtin_amdgcn_s_barrier();
        advance();
      }
    };
    if (wave < NT_WAVES / 2) steady_loop(std::integral_constant<int, 1>{});
    else steady_loop(std::integral_constant<int, 2>{});
    // tail: the last AHEAD k-tiles.  Before k-tile (current + 1) starts it must have
    // landed; what was issued after it -- the rest of this tile, the next tile's A rows
    // -- may still fly (the last tile of a workgroup issues nothing and drains)
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) {
      compute(cur, slot_ahead(), t, std::false_type{}, std::integral_constant<int, 0>{});
      if (t + 1 < AHEAD) {
        if (has_next) {
          if (t == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS + A_ITERS) : "memory");
          else if (t == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS + 2 * A_ITERS) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * A_ITERS) : "memory");
        } else {
          if (t == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
          else if (t == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
      }
      advance();
    }
    // every wave is done with the W half of the ring before it becomes staging
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = acc[i][j] * g.acc_scale;
    run_epilogue<TM, TN>(g, acc, Bs, wave, lane, tile_m * BM + wm * WROWS,
                         tile_n * BN + wn * 64);
    if (!has_next) break;
    // staging reads are done before the next tile's W rows land there
    __builtin_amdgcn_s_barrier();
    q += gridDim.x;
    tile_m = ntile_m; tile_n = ntile_n;
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) ra[it] = ra_n[it];
    // `cur` already points at the slot of the next tile's k-tile 0 (the ring kept rolling)
  }
}

// ---------------------------------------------------------------------------
// Round 4: the 256 x 256 split16 tile as a TWO-GROUP PING-PONG (igemm_split16_pp_kernel).
//
// The lockstep loop above keeps the matrix pipe ~65 % busy in cycles: all eight waves come
// out of the k-tile barrier together, all read their fragments, and the two waves of a SIMD
// stall on the same things at the same time.  Here the workgroup is two groups of four
// waves (one wave of each group per SIMD; group = row half of the tile) that alternate
// roles every phase:
//
//   phase 2t     group 0 multiplies k-tile t (24 MFMAs, nothing else in its stream)
//                group 1 reads its 12 fragments of k-tile t from LDS, issues the 4 DMA
//                pieces of k-tile t + 3 and waits for its own pieces of k-tile t + 1
//   phase 2t + 1 group 0 reads k-tile t + 1 / issues k-tile t + 4, group 1 multiplies t
//
// with one s_barrier per phase.  What makes the read phase fit under the partner's 768
// MFMA cycles is that it holds NO vector-ALU instruction (two waves of a SIMD share its
// VALU port, and beside an MFMA stream every VALU op of the partner costs ~10-20 cycles):
//   * operands go L2 -> LDS by buffer_load ... lds: descriptor (SGPRs) + a loop-invariant
//     per-lane 32-bit row offset + a SCALAR k offset; rows past M / N and out-of-image
//     taps carry an out-of-range offset, for which the hardware writes zeros into LDS (no
//     zero page, no per-lane select per piece);
//   * the ring has 4 slots and the k loop is unrolled 4 x, so every ds_read_b128 is one
//     base register + an immediate;
//   * the per-lane tap bounds test runs once per (kh, kw), not per k-tile.
// Prototype and measurements: tools/bench/pp.hip, profiles/r4_mainloop_prototype.txt
// (1570-1650 cycles per k-tile against 2210-2280 for the lockstep loop; ideal 1536).
// Same k order and the same (hl, lh, hh) order per accumulator as the kernels above:
// bitwise the same results.
// ---------------------------------------------------------------------------
// GemmArgs re-read from the kernel-argument segment (it is the first kernel argument:
// offset 0) through a pointer the compiler cannot see through.  Values loaded this way
// are not hoisted above the point of the call: split16_pp_tile uses it so that neither
// the next tile's set-up inputs nor the epilogue's arguments occupy scalar registers
// during the main loop (they spilled into vector registers and from there to scratch --
// VALU and VMEM traffic inside a loop whose waits count VMEM operations).
__device__ __forceinline__ GemmArgs reload_gemm_args() {
  typedef const int __attribute__((address_space(4)))* KArgWords;
  KArgWords kw = (KArgWords)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kw));
  GemmArgs g;
  int words[sizeof(GemmArgs) / 4];
#pragma unroll
  for (unsigned i = 0; i < sizeof(GemmArgs) / 4; ++i) words[i] = kw[i];
  __builtin_memcpy(&g, words, sizeof(GemmArgs));
  g.M = live_rows(g.m_live, g.m_live_mul, g.M);   // (GemmArgs::m_live: row count on the device)
  return g;
}


__device__ __forceinline__ void split16_pp_tile(int tile_m, int tile_n, int tid) {
  const GemmArgs g = reload_gemm_args();
  constexpr int BM = 256, BN = 256, BK = 16, S = 4, TM = 4, TN = 2, LOADS = 4;
  constexpr unsigned OOB = 0x80000000u;  // >= num_records of every descriptor below
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [S][BM * 16]
  float* Bs = smem + S * BM * BK;   // [S][BN * 16]
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- loader state --------------------------------------------------------------
  const int lrow = tid >> 2;                        // 0..127
  const int kc = (tid & 3) ^ ((tid >> 4) & 3);      // source chunk of this lane's LDS slot
  const int HoWo = g.Ho * g.Wo;
  const int img0 = (tile_m * BM) / HoWo;            // first image of the tile (scalar)
  // descriptors: A from (image img0, pixel (-pad, -pad)) so that every per-lane offset is
  // non-negative; W from the tile's first row
  const long bias = ((long)g.pad * g.Wd + g.pad) * g.a_pix_stride;
  const __amdgpu_buffer_rsrc_t srd_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.A + (long)img0 * g.a_img_stride - bias), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_a2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.A2 ? g.A2 + (long)img0 * g.a2_img_stride : g.A), 0, 0x7fffffff,
      0x00020000);
  const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.W + (long)tile_n * BN * g.Kp), 0, 0x7fffffff, 0x00020000);
  unsigned row_off[2], row_off2[2], voff[2], vb[2];
  int hi0[2], wi0[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int m = tile_m * BM + it * 128 + lrow;
    const bool ok = m < g.M;
    const int mm = ok ? m : g.M - 1;
    const int img = mm / HoWo;
    const int rem = mm - img * HoWo;
    const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
    hi0[it] = ho * g.stride - g.pad;
    wi0[it] = wo * g.stride - g.pad;
    row_off[it] = ok ? (unsigned)(((long)(img - img0) * g.a_img_stride +
                                   ((long)hi0[it] * g.Wd + wi0[it]) * g.a_pix_stride + bias +
                                   kc * 4) * 4)
                     : OOB;
    row_off2[it] = (ok && g.A2)
                       ? (unsigned)(((long)(img - img0) * g.a2_img_stride +
                                     ((long)(ho * g.stride2) * g.W2d + wo * g.stride2) *
                                         g.a2_pix_stride + kc * 4) * 4)
                       : OOB;
    const int n = tile_n * BN + it * 128 + lrow;
    vb[it] = n < g.N ? (unsigned)(((long)(it * 128 + lrow) * g.Kp + kc * 4) * 4) : OOB;
  }
  const int nk = g.Kp / BK;
  const int k1_tiles = g.A2 ? g.K1 / BK : 0x7fffffff;
  // The k-tile the loader issues next: index, tap, first channel.  ONE code path per piece:
  // the A source of the moment is (descriptor cur_srd, per-lane offsets voff[], scalar
  // offset seg_soff + 4 * is_cin0); a tap change or the switch to the second source (once
  // per tile) rewrites that state.
  int is_kt = 0, is_kh = 0, is_kw = 0, is_cin0 = 0, seg_soff = 0;
  int cin_limit = g.Cin;
  __amdgpu_buffer_rsrc_t cur_srd = srd_a;
  auto set_tap = [&]() {  // per-lane bounds of tap (is_kh, is_kw): once per tap, not per k-tile
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const bool inb = (unsigned)(hi0[it] + is_kh) < (unsigned)g.H &&
                       (unsigned)(wi0[it] + is_kw) < (unsigned)g.Wd;
      voff[it] = inb ? row_off[it] : OOB;
    }
    seg_soff = (int)((((long)is_kh * g.Wd + is_kw) * g.a_pix_stride) * 4);
  };
  set_tap();
  auto issue_tile = [&](auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    // (opaque: sixteen loop-invariant LDS addresses are not worth sixteen scalar registers)
    int woff = wave * (16 * BK);
    asm volatile("" : "+s"(woff));
    float* adst = As + SLOT * (BM * BK) + woff;
    float* bdst = Bs + SLOT * (BN * BK) + woff;
    const int soff = seg_soff + is_cin0 * 4;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(cur_srd, (LDS_AS void*)adst, 16, voff[0], soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(cur_srd, (LDS_AS void*)(adst + 128 * BK), 16, voff[1],
                                             soff, 0, 0);
    const int woffk = is_kt * (BK * 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (LDS_AS void*)bdst, 16, vb[0], woffk, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_w, (LDS_AS void*)(bdst + 128 * BK), 16, vb[1],
                                             woffk, 0, 0);
    // advance; past the end of K the last k-tile is fetched again into slots nobody reads
    // (every phase issues exactly LOADS pieces: constant vmcnt counts)
    if (is_kt + 1 < nk) {
      ++is_kt;
      is_cin0 += BK;
      if (is_kt == k1_tiles) {
        // second source from here on: a 1x1 / stride2 input, always in bounds, k from K1
        cur_srd = srd_a2;
        voff[0] = row_off2[0];
        voff[1] = row_off2[1];
        seg_soff = 0;
        is_cin0 = 0;
        cin_limit = 0x7fffffff;
      } else if (is_cin0 >= cin_limit) {
        is_cin0 = 0;
        if (++is_kw == g.KW) { is_kw = 0; ++is_kh; }
        set_tap();
      }
    }
  };

  // ---- fragments -------------------------------------------------------------------
  const int frow = lane & 31, fhalf = lane >> 5;
  const int swz = (frow >> 2) & 3;
  const int chi = (2 * fhalf) ^ swz, clo = (2 * fhalf + 1) ^ swz;
  const float* a_hi = As + (wm * 128 + frow) * BK + (chi << 2);
  const float* a_lo = As + (wm * 128 + frow) * BK + (clo << 2);
  const float* b_hi = Bs + (wn * 64 + frow) * BK + (chi << 2);
  const float* b_lo = Bs + (wn * 64 + frow) * BK + (clo << 2);
  f32x4 ah[TM], al[TM], bh[TN], bl[TN];
  auto read_frags = [&](auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[j] = *reinterpret_cast<const f32x4*>(b_hi + SLOT * (BN * BK) + j * 32 * BK);
      bl[j] = *reinterpret_cast<const f32x4*>(b_lo + SLOT * (BN * BK) + j * 32 * BK);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[i] = *reinterpret_cast<const f32x4*>(a_hi + SLOT * (BM * BK) + i * 32 * BK);
      al[i] = *reinterpret_cast<const f32x4*>(a_lo + SLOT * (BM * BK) + i * 32 * BK);
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // the 24 MFMAs of a k-tile, product-major (every accumulator once per product; per
  // accumulator the order is hl, lh, hh as in the kernels above)
  auto multiply = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(ah[i]), as_f16x8(bl[j]),
                                                           acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(al[i]), as_f16x8(bh[j]),
                                                           acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(ah[i]), as_f16x8(bh[j]),
                                                           acc[i][j], 0, 0, 0);
  };
  // a phase boundary: nothing is scheduled across it (an MFMA is register-only, the
  // "memory" clobber of the waits would not hold it)
  auto phase_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // read phase of the k-tile in ring slot SLOT: fragments, then the pieces of the k-tile
  // three ahead into the slot both groups are done with
  auto read_phase = [&](auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    read_frags(slot_tag);
    issue_tile(std::integral_constant<int, (SLOT + S - 1) % S>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  // prologue: k-tiles 0 .. S-2 in flight, k-tile 0 landed
  issue_tile(std::integral_constant<int, 0>{});
  issue_tile(std::integral_constant<int, 1>{});
  issue_tile(std::integral_constant<int, 2>{});
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LOADS) : "memory");
  phase_barrier();
  if (wm == 0) {
    read_phase(std::integral_constant<int, 0>{});
    phase_barrier();
    auto step = [&](auto slot_tag) {  // k-tile kt in slot SLOT: multiply it, then read kt + 1
      constexpr int SLOT = decltype(slot_tag)::value;
      multiply();
      __builtin_amdgcn_sched_barrier(0);
      // every wave has its own pieces of k-tile kt + 1 before the barrier that precedes
      // the first read of it; two younger k-tiles stay in flight
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LOADS) : "memory");
      phase_barrier();
      read_phase(std::integral_constant<int, (SLOT + 1) % S>{});
      phase_barrier();
    };
    int kt = 0;
    for (; kt + 4 <= nk; kt += 4) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
    }
    if (kt < nk) {  // Kp is a multiple of 32: nk is even
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
    }
  } else {
    phase_barrier();
    auto step = [&](auto slot_tag) {  // k-tile kt in slot SLOT: read it, then multiply it
      read_phase(slot_tag);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LOADS) : "memory");
      phase_barrier();
      multiply();
      phase_barrier();
    };
    int kt = 0;
    for (; kt + 4 <= nk; kt += 4) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
    }
    if (kt < nk) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
    }
  }
  // the re-fetched tail k-tiles have landed and every wave is done with the ring before it
  // becomes the epilogue's staging
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const GemmArgs ge = reload_gemm_args();  // the epilogue's arguments, loaded here
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = acc[i][j] * ge.acc_scale;
  run_epilogue<TM, TN>(ge, acc, smem, wave, lane, tile_m * BM + wm * 128, tile_n * BN + wn * 64);
}

// ---------------------------------------------------------------------------
// The same ping-pong with the operands staged in k-tile PAIRS = whole 128-byte lines per
// row (igemm_split16_pp32_kernel; needs Cin % 32 == 0).  A DMA piece of the kernel above
// is 16 rows x 64 B: half a cache line per row, the other half asked for one k-tile later.
// Measured (tools/bench/pp.hip E2): such pieces stream 4.0 TB/s from HBM and 66 B/clk/CU
// from L2 where 8 rows x 128 B pieces reach 6.0 TB/s and 102 B/clk/CU -- and the 1x1 reduce
// convs of layer3 (every workgroup streams its own 1 MB A panel) sat at 3.7 TB/s.
//   * LDS rows of 128 B = two 16-slot slabs; the 16-byte chunk q = 4 * slab + 2 * group +
//     (hi | lo) of row r lives at position q ^ ((r >> 1) & 7): conflict-free for the DMA
//     write (lane-linear, the XOR is applied on the source address) and for the row-per-lane
//     ds_read_b128 of either slab;
//   * A ring of 3 pair slots (96 KB), W ring of 2 (64 KB): all 160 KB.  Per pair T:
//       phase 4T     group 0 multiplies slab 2T      | group 1 reads slab 2T, issues W(T+1)
//       phase 4T + 1 group 0 reads slab 2T+1, A(T+2) | group 1 multiplies slab 2T
//       phase 4T + 2 group 0 multiplies slab 2T+1    | group 1 reads slab 2T+1, issues A(T+2)
//                    both: vmcnt(4) -- W(T+1), A(T+1) landed, A(T+2) in flight
//       phase 4T + 3 group 0 reads slab 2T+2, W(T+2) | group 1 multiplies slab 2T+1
//     The loop is unrolled over 6 pairs (slot indices 3 x 2), every ds_read address is a
//     base register + immediate.
// Same k order, same bits.
// ---------------------------------------------------------------------------
// Loader state of one tile of the pair-staged ping-pong kernel (all per-lane offsets are
// bytes relative to the tile's descriptors).
struct PpLoader {
  __amdgpu_buffer_rsrc_t srd_a, srd_a2, srd_w, cur_srd;
  unsigned row_off[4], row_off2[4], voff[4], vb[4];
  int hw0[4];  // (hi0 << 16) | (wi0 & 0xffff): the row's first input pixel (may be negative)
  unsigned mask[4];  // TAPI: bit kh * KW + kw = that tap of the row is inside the image
  int ia, iw, is_kh, is_kw, is_cin0, seg_soff, cin_limit;
};

// Persistent workgroups: a workgroup walks tiles q = blockIdx.x, + gridDim.x, ...  After a
// tile's main loop the first pairs of the NEXT tile -- A(0), W(0), A(1): exactly the kernel's
// prologue -- are requested before the epilogue runs and land under it; the epilogue stages
// through the two ring slots the prefetch does not use (A slot 2 + W slot 1: 64 KB).
// LDS: [A0 32K][A1 32K][W0 32K][A2 32K][W1 32K].
// BNW: columns of the output tile.  256: wave tiles 128 x 64 (2 x 4 waves).  128 (round 4, the
// N = 128 layers of layer2 and the odd-width products): wave tiles 64 x 64 (4 x 2 waves), the
// SAME loader minus the W pieces of rows 128..255, which no wave tile reads (round 5; they had
// been issued with out-of-range offsets = zero fill): every counted wait names the A pieces
// issued behind them, so the waits are the 256-column kernel's.
// TAPI (GemmArgs::tap_inner): k runs (32-channel slice, tap, channel).  The KH * KW shifted
// copies of a slice are then requested in consecutive pairs -- the second to ninth find their
// lines in L2 (tap-major, a line is asked for again 8 pairs = 8 x 32 workgroups x 32 KB later
// and has left it: layer3's 3x3 convs pulled their input 8.3 x over the fabric).  The in-image
// test of a row's taps is a 9-bit mask made once per tile, the per-pair step is 12 VALU + a
// few scalar instructions in the read phase.  NOT the tap-major bits (the order of the fp32
// additions differs); compile-time so that the other launches keep their loop.
template <int BNW, bool F16 = false, bool TAPI = false>
__device__ __forceinline__ void split16_pp32_tile(int tile_m, int tile_n, int tid, bool prefetched,
                                                  bool has_next, int ntile_m, int ntile_n) {
  const GemmArgs g = reload_gemm_args();
  constexpr int BM = 256, BN = 256, BKP = 32, TM = BNW == 256 ? 4 : 2, TN = 2;
  static_assert(BNW == 256 || BNW == 128, "ping-pong tile: 256 or 128 columns");
  constexpr int SLOTF = BM * BKP;  // floats per 32 KB slot
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  auto a_slot = [&](int sl) { return smem + (sl < 2 ? sl : 3) * SLOTF; };
  auto w_slot = [&](int sl) { return smem + (sl == 0 ? 2 : 4) * SLOTF; };
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = BNW == 256 ? wave >> 2 : wave >> 1, wn = BNW == 256 ? wave & 3 : wave & 1;
  const int group = wave >> 2;   // ping-pong group: the row half of the tile
  const int HoWo = g.Ho * g.Wo;
  const int np = g.Kp / BKP;  // k-tile pairs
  const int k1_pairs = g.A2 ? g.K1 / BKP : 0x7fffffff;

  // ---- loader: piece `it` of this wave = rows it * 64 + wave * 8 .. + 7, one 128-B line each
  auto set_tap = [&](PpLoader& L) {  // per-lane bounds of tap (is_kh, is_kw): once per tap
    if constexpr (TAPI) {
      const int t = L.is_kh * g.KW + L.is_kw;
#pragma unroll
      for (int it = 0; it < 4; ++it) L.voff[it] = (L.mask[it] >> t) & 1u ? L.row_off[it] : OOB;
      L.seg_soff = (int)((((long)L.is_kh * g.Wd + L.is_kw) * g.a_pix_stride) * 4);
      return;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int hi = (L.hw0[it] >> 16) + L.is_kh, wi = (int)(short)(L.hw0[it] & 0xffff) + L.is_kw;
      const bool inb = (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.Wd;
      L.voff[it] = inb ? L.row_off[it] : OOB;
    }
    L.seg_soff = (int)((((long)L.is_kh * g.Wd + L.is_kw) * g.a_pix_stride) * 4);
  };
  auto setup = [&](PpLoader& L, int tm, int tn) {
    const int prow = lane >> 3, pos = lane & 7;
    const int m0 = tm * BM;
    const int img0 = m0 / HoWo;       // scalar
    const int rem0 = m0 - img0 * HoWo;
    // descriptors: A from (image img0, pixel (-pad, -pad)) so that every per-lane offset is
    // non-negative; W from the tile's first row.  Out-of-range offsets read as zeros.
    const long bias = ((long)g.pad * g.Wd + g.pad) * g.a_pix_stride;
    L.srd_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.A + (long)img0 * g.a_img_stride - bias), 0, 0x7fffffff, 0x00020000);
    L.srd_a2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.A2 ? g.A2 + (long)img0 * g.a2_img_stride : g.A), 0, 0x7fffffff,
        0x00020000);
    L.srd_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W + (long)tn * BNW * g.Kp), 0,
                                                0x7fffffff, 0x00020000);
    // (rows of a tile are consecutive: the per-lane divisions run on rem0 + row < HoWo + 256,
    // exact in float arithmetic with one correction step -- a tenth of an integer division)
    const float r_howo = 1.0f / (float)HoWo, r_wo = 1.0f / (float)g.Wo;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 64 + wave * 8 + prow;
      const int q = pos ^ ((row >> 1) & 7);  // source chunk of this lane's LDS position
      const bool ok = m0 + row < g.M;
      const int x = rem0 + (ok ? row : 0);
      int dimg = (int)((float)x * r_howo);
      dimg -= (dimg * HoWo > x);
      dimg += ((dimg + 1) * HoWo <= x);
      const int rem = x - dimg * HoWo;
      int ho = (int)((float)rem * r_wo);
      ho -= (ho * g.Wo > rem);
      ho += ((ho + 1) * g.Wo <= rem);
      const int wo = rem - ho * g.Wo;
      const int hi0 = ho * g.stride - g.pad, wi0 = wo * g.stride - g.pad;
      L.hw0[it] = (hi0 << 16) | (wi0 & 0xffff);
      if constexpr (TAPI) {
        unsigned m = 0;
        for (int kh = 0, t = 0; kh < g.KH; ++kh)
          for (int kw = 0; kw < g.KW; ++kw, ++t)
            m |= (unsigned)((unsigned)(hi0 + kh) < (unsigned)g.H &&
                            (unsigned)(wi0 + kw) < (unsigned)g.Wd) << t;
        L.mask[it] = m;
      }
      L.row_off[it] = ok ? (unsigned)(((long)dimg * g.a_img_stride +
                                       ((long)hi0 * g.Wd + wi0) * g.a_pix_stride + bias + q * 4) * 4)
                         : OOB;
      L.row_off2[it] = (ok && g.A2)
                           ? (unsigned)(((long)dimg * g.a2_img_stride +
                                         ((long)(ho * g.stride2) * g.W2d + wo * g.stride2) *
                                             g.a2_pix_stride + q * 4) * 4)
                           : OOB;
      L.vb[it] = (row < BNW && tn * BNW + row < g.N) ? (unsigned)(((long)row * g.Kp + q * 4) * 4) : OOB;
    }
    L.ia = L.iw = L.is_kh = L.is_kw = L.is_cin0 = 0;
    L.cin_limit = g.Cin;
    L.cur_srd = L.srd_a;
    set_tap(L);
  };
  // the A rows of pair L.ia into A slot SLOT (`really`: a tile whose first pairs were
  // prefetched only replays the state changes), then advance
  auto issue_a = [&](PpLoader& L, auto slot_tag, bool really) {
    constexpr int SLOT = decltype(slot_tag)::value;
    if (really) {
      int woff = wave * (8 * BKP);
      asm volatile("" : "+s"(woff));  // (not sixteen hoisted LDS addresses in scalar registers)
      float* dst = a_slot(SLOT) + woff;
      const int soff = L.seg_soff + L.is_cin0 * 4;
#pragma unroll
      for (int it = 0; it < 4; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(L.cur_srd, (LDS_AS void*)(dst + it * (64 * BKP)), 16,
                                                 L.voff[it], soff, 0, 0);
    }
    if constexpr (TAPI) {
      if (L.ia + 1 < np) {
        ++L.ia;
        if (++L.is_kw == g.KW) {
          L.is_kw = 0;
          if (++L.is_kh == g.KH) { L.is_kh = 0; L.is_cin0 += BKP; }
        }
        set_tap(L);
      }
      return;
    }
    if (L.ia + 1 < np) {  // past the end the last pair is fetched again (constant vmcnt counts)
      ++L.ia;
      L.is_cin0 += BKP;
      if (L.ia == k1_pairs) {
        // second source from here on: a 1x1 / stride2 input, always in bounds, k from K1
        L.cur_srd = L.srd_a2;
#pragma unroll
        for (int it = 0; it < 4; ++it) L.voff[it] = L.row_off2[it];
        L.seg_soff = 0;
        L.is_cin0 = 0;
        L.cin_limit = 0x7fffffff;
      } else if (L.is_cin0 >= L.cin_limit) {
        L.is_cin0 = 0;
        if (++L.is_kw == g.KW) { L.is_kw = 0; ++L.is_kh; }
        set_tap(L);
      }
    }
  };
  auto issue_w = [&](PpLoader& L, auto slot_tag, bool really) {
    constexpr int SLOT = decltype(slot_tag)::value;
    if (really) {
      int woff = wave * (8 * BKP);
      asm volatile("" : "+s"(woff));
      float* dst = w_slot(SLOT) + woff;
      const int soff = L.iw * (BKP * 4);
      // (128-column tiles: W rows 128..255 of a slot are never read -- wave tiles end at row 127 --
      // so their two pieces are not issued; every counted wait names the A pieces behind them)
#pragma unroll
      for (int it = 0; it < (BNW == 256 ? 4 : 2); ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(L.srd_w, (LDS_AS void*)(dst + it * (64 * BKP)), 16,
                                                 L.vb[it], soff, 0, 0);
    }
    if (L.iw + 1 < np) ++L.iw;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  PpLoader L;
  setup(L, tile_m, tile_n);

  // ---- fragments: per lane four chunk positions (slab x hi/lo); one base register set per
  // slot group a ds_read immediate (< 64 KB) can reach
  const int frow = lane & 31, fhalf = lane >> 5;
  const int fsw = (frow >> 1) & 7;
  const float* a_p[2][2];  // A slots 0, 1
  const float* a_q[2][2];  // A slot 2
  const float* b_p[2][2];  // W slot 0
  const float* b_q[2][2];  // W slot 1
#pragma unroll
  for (int sl = 0; sl < 2; ++sl)
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
      const int off = ((sl * 4 + fhalf * 2 + hl) ^ fsw) * 4;
      a_p[sl][hl] = a_slot(0) + (wm * (TM * 32) + frow) * BKP + off;
      a_q[sl][hl] = a_slot(2) + (wm * (TM * 32) + frow) * BKP + off;
      b_p[sl][hl] = w_slot(0) + (wn * 64 + frow) * BKP + off;
      b_q[sl][hl] = w_slot(1) + (wn * 64 + frow) * BKP + off;
    }
  f32x4 ah[TM], al[TM], bh[TN], bl[TN];
  auto read_frags = [&](auto sa_tag, auto sw_tag, auto slab_tag) {
    constexpr int SLA = decltype(sa_tag)::value, SLW = decltype(sw_tag)::value;
    constexpr int SLAB = decltype(slab_tag)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[j] = *reinterpret_cast<const f32x4*>((SLW ? b_q : b_p)[SLAB][0] + j * 32 * BKP);
      bl[j] = *reinterpret_cast<const f32x4*>((SLW ? b_q : b_p)[SLAB][1] + j * 32 * BKP);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (SLA == 2) {
        ah[i] = *reinterpret_cast<const f32x4*>(a_q[SLAB][0] + i * 32 * BKP);
        al[i] = *reinterpret_cast<const f32x4*>(a_q[SLAB][1] + i * 32 * BKP);
      } else {
        ah[i] = *reinterpret_cast<const f32x4*>(a_p[SLAB][0] + SLA * SLOTF + i * 32 * BKP);
        al[i] = *reinterpret_cast<const f32x4*>(a_p[SLAB][1] + SLA * SLOTF + i * 32 * BKP);
      }
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto multiply = [&]() {
    if constexpr (F16) {
      // fast mode: the "hi" and "lo" chunks of a pretended split row are two different k
      // ranges of a plain f16 row -- hi . hi + lo . lo = one MFMA per 16 real k
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(ah[i]), as_f16x8(bh[j]),
                                                             acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(al[i]), as_f16x8(bl[j]),
                                                             acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(ah[i]), as_f16x8(bl[j]),
                                                           acc[i][j], 0, 0, 0);
#if !MILAN_DROP_CROSS_TERM  // (sensitivity check of the parity suite: a build WITHOUT the
                            // Al . Bh product must fail it -- profiles/r4_experiments.txt I)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(al[i]), as_f16x8(bh[j]),
                                                           acc[i][j], 0, 0, 0);
#endif
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(ah[i]), as_f16x8(bh[j]),
                                                           acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto phase_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // read phase of the even slab of the pair in slots (SLA, SLW): fragments, then the W rows
  // of the NEXT pair into the other W slot
  auto read_even = [&](auto sa_tag, auto sw_tag) {
    constexpr int SLW = decltype(sw_tag)::value;
    read_frags(sa_tag, sw_tag, I0{});
    issue_w(L, std::integral_constant<int, SLW ^ 1>{}, true);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // ... of the odd slab: fragments, then the A rows of the pair two ahead (the A slot that
  // held the previous pair)
  auto read_odd = [&](auto sa_tag, auto sw_tag) {
    constexpr int SLA = decltype(sa_tag)::value;
    read_frags(sa_tag, sw_tag, I1{});
    issue_a(L, std::integral_constant<int, (SLA + 2) % 3>{}, true);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  // prologue: A(0), W(0), A(1) in flight (or already here: prefetched by the previous tile,
  // which also waited for them); A(0), W(0) landed
  issue_a(L, I0{}, !prefetched);
  issue_w(L, I0{}, !prefetched);
  issue_a(L, I1{}, !prefetched);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  phase_barrier();
  int left = np;
  if (group == 0) {
    read_even(I0{}, I0{});
    phase_barrier();
    auto pair = [&](auto sa_tag, auto sw_tag) {
      constexpr int SLA = decltype(sa_tag)::value, SLW = decltype(sw_tag)::value;
      multiply();
      phase_barrier();
      read_odd(sa_tag, sw_tag);
      phase_barrier();
      multiply();
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      phase_barrier();
      read_even(std::integral_constant<int, (SLA + 1) % 3>{}, std::integral_constant<int, SLW ^ 1>{});
      phase_barrier();
    };
    for (;;) {
      pair(std::integral_constant<int, 0>{}, I0{});
      if (--left == 0) break;
      pair(std::integral_constant<int, 1>{}, I1{});
      if (--left == 0) break;
      pair(std::integral_constant<int, 2>{}, I0{});
      if (--left == 0) break;
      pair(std::integral_constant<int, 0>{}, I1{});
      if (--left == 0) break;
      pair(std::integral_constant<int, 1>{}, I0{});
      if (--left == 0) break;
      pair(std::integral_constant<int, 2>{}, I1{});
      if (--left == 0) break;
    }
  } else {
    phase_barrier();
    auto pair = [&](auto sa_tag, auto sw_tag) {
      read_even(sa_tag, sw_tag);
      phase_barrier();
      multiply();
      phase_barrier();
      read_odd(sa_tag, sw_tag);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      phase_barrier();
      multiply();
      phase_barrier();
    };
    for (;;) {
      pair(std::integral_constant<int, 0>{}, I0{});
      if (--left == 0) break;
      pair(std::integral_constant<int, 1>{}, I1{});
      if (--left == 0) break;
      pair(std::integral_constant<int, 2>{}, I0{});
      if (--left == 0) break;
      pair(std::integral_constant<int, 0>{}, I1{});
      if (--left == 0) break;
      pair(std::integral_constant<int, 1>{}, I0{});
      if (--left == 0) break;
      pair(std::integral_constant<int, 2>{}, I1{});
      if (--left == 0) break;
    }
  }
  // the re-fetched tail pairs have landed and every wave is done with the ring
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (has_next) {
    // the next tile's prologue, requested now; its loader state is rebuilt after the epilogue
    // (keeping it alive across the epilogue would cost 20 vector registers there)
    PpLoader N;
    setup(N, ntile_m, ntile_n);
    issue_a(N, I0{}, true);
    issue_w(N, I0{}, true);
    issue_a(N, I1{}, true);
  }
  const GemmArgs ge = reload_gemm_args();  // the epilogue's arguments, loaded here
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = acc[i][j] * ge.acc_scale;
  // staging: A slot 2 + W slot 1 (contiguous 64 KB; the prefetch goes to A0, A1, W0)
  run_epilogue<TM, TN, true>(ge, acc, a_slot(2), wave, lane, tile_m * BM + wm * (TM * 32),
                             tile_n * BNW + wn * 64);
}

template <int BNW>
__global__ __launch_bounds__(512, 2) void igemm_split16_pp32n_kernel(GemmArgs g, int tiles_m,
                                                                     int tiles_n) {
  tiles_m = live_tiles_m(g, tiles_m, 256);
  const int T = tiles_m * tiles_n;
  int q = blockIdx.x;
  if (q >= T) return;
  bool prefetched = false;
  for (;;) {
    const int tile = xcd_tile(q, T);
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const bool has_next = q + (int)gridDim.x < T;
    int ntile_m = tile_m, ntile_n = tile_n;
    if (has_next) {
      const int nt = xcd_tile(q + gridDim.x, T);
      ntile_m = nt / tiles_n;
      ntile_n = nt - ntile_m * tiles_n;
    }
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    split16_pp32_tile<BNW>(tile_m, tile_n, tid, prefetched, has_next, ntile_m, ntile_n);
    if (!has_next) break;
    // the prefetch has landed, the epilogue's stores are out and its staging reads are done
    // before the next tile's DMA reuses the slots
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    q += gridDim.x;
    prefetched = true;
  }
}

__global__ __launch_bounds__(512, 2) void igemm_split16_pp32_kernel(GemmArgs g, int tiles_m,
                                                                    int tiles_n) {
  // (the 256-column form keeps its round-4 name: profiles and tools key on it)
  tiles_m = live_tiles_m(g, tiles_m, 256);
  const int T = tiles_m * tiles_n;
  const int q = blockIdx.x;
  if (q >= T) return;
  const int tile = xcd_tile(q, T);
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  split16_pp32_tile<256>(tile_m, tile_n, tid, false, false, tile_m, tile_n);
}

// k x k convs in (slice, tap, channel) order (GemmArgs::tap_inner)
template <int BNW>
__global__ __launch_bounds__(512, 2) void igemm_split16_pp32t_kernel(GemmArgs g, int tiles_m,
                                                                     int tiles_n) {
  tiles_m = live_tiles_m(g, tiles_m, 256);
  const int T = tiles_m * tiles_n;
  const int q = blockIdx.x;
  if (q >= T) return;
  const int tile = xcd_tile(q, T);
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  split16_pp32_tile<BNW, false, true>(tile_m, tile_n, tid, false, false, tile_m, tile_n);
}

// fast mode (GemmArgs::f16): the same tile with one MFMA per 16 real k
template <int BNW, bool TAPI>
__global__ __launch_bounds__(512, 2) void igemm_f16_pp32_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  tiles_m = live_tiles_m(g, tiles_m, 256);
  const int T = tiles_m * tiles_n;
  const int q = blockIdx.x;
  if (q >= T) return;
  const int tile = xcd_tile(q, T);
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  split16_pp32_tile<BNW, true, TAPI>(tile_m, tile_n, tid, false, false, tile_m, tile_n);
}

__global__ __launch_bounds__(512, 2) void igemm_split16_pp_kernel(GemmArgs g, int tiles_m,
                                                                  int tiles_n) {
  // one tile per workgroup (gridDim.x == tiles), or workgroups walking tiles q = blockIdx.x,
  // + gridDim.x, ... (gridDim.x a multiple of 8: q stays on its XCD)
  tiles_m = live_tiles_m(g, tiles_m, 256);
  const int T = tiles_m * tiles_n;
  for (int q = blockIdx.x; q < T; q += gridDim.x) {
    const int tile = xcd_tile(q, T);
    const int tile_m = tile / tiles_n;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    split16_pp_tile(tile_m, tile - tile_m * tiles_n, tid);
    // the epilogue's staging reads are done before the next tile's DMA lands there
    __builtin_amdgcn_s_barrier();
  }
}

// The same pipeline with 64 x 64 wave tiles (4 waves per 256 x 64 block, two
// blocks per CU): the N = 64 layers of layer1.
template <int BM, int BN, int STAGES>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64, 2) void igemm_split16_tm2_kernel(
    GemmArgs g, int tiles_m, int tiles_n) {
  tiles_m = live_tiles_m(g, tiles_m, BM);
  if ((int)blockIdx.x >= tiles_m * tiles_n) return;
  const int tile = xcd_tile(blockIdx.x, tiles_m * tiles_n);
  const int tile_m = tile / tiles_n;
  split16_tile<BM, BN, STAGES, 0, 2>(g, tile_m, tile - tile_m * tiles_n);
}


#if MILAN_EXPERIMENTS
// ---------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with the input strip resident in LDS.
//
// The generic kernels stream the implicit-im2col operand tap by tap: every
// input pixel crosses L2 -> LDS nine times, and because a tap comes back only
// Cin/16 k-tiles later the re-reads miss L2 as well (rocprofv3 FETCH_SIZE of
// layer3's 3x3 convs: 6.4 GB per launch for a 0.77 GB input).  Here K runs
// channel-chunk-major: for each chunk of 16 channels (one 64-byte split-format
// row per pixel) the strip of input pixels the 256 output rows of the tile can
// touch -- rows m0-(W+1) .. m0+255+(W+1) of the flat (image, y, x) pixel list,
// images are stacked so the strip is one contiguous range -- is DMA'd into LDS
// ONCE, and the nine taps read their A fragments from it at a per-tap row
// offset; taps that fall outside the image read a zero row instead (address
// select per lane).  A-side DMA instructions per 9 k-steps: 3 per wave instead
// of 72.  The weights stream through the same 4-deep ring as before, packed
// chunk-major ([n][chunk][tap][16]).  Accumulation order over K differs from
// the generic kernel (chunk-major), so the layer -> kernel choice stays a pure
// function of the layer (never of the batch).
// ---------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(2 * (BN / 64) * 64, 2) void conv3x3_split16_kernel(
    GemmArgs g, int tiles_m, int tiles_n) {
  constexpr int BM = 256, BK = 16, TM = 4, TN = 2, STAGES = 4, AHEAD = 3;
  constexpr int WAVES_N = BN / 64;
  constexpr int NT = 2 * WAVES_N * 64;
  constexpr int LROWS = NT / 4;                 // rows per DMA pass (4 lanes per row)
  constexpr int B_ITERS = BN / LROWS;           // = 2
  constexpr int SROWS = 384;                    // strip rows incl. padding
  constexpr int A_PIECES = SROWS / LROWS;       // 3 (8 waves) or 6 (4 waves)
  static_assert(B_ITERS == 2 && A_PIECES <= 9, "tile configuration");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Bs = smem;                               // [STAGES][BN*16]
  float* As = smem + STAGES * BN * BK;            // [2][SROWS*16]
  float* Zs = As + 2 * SROWS * BK;                // 16 floats of zeros
  float* Ds = Zs + 16;                            // sink of the dummy DMA pieces

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int tile = xcd_tile(blockIdx.x, tiles_m * tiles_n);
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  if (tid < 16) Zs[tid] = 0.f;

  const int Wd = g.Wd, H = g.H, Cin = g.Cin;
  const long m0 = (long)tile_m * BM;
  const long total_px = (long)g.M;               // output pixels == input pixels
  const int halo = Wd + 1;
  const int nchunks = Cin / BK;
  const int nk = nchunks * 9;

  // ---- DMA lane roles --------------------------------------------------------
  const int lrow = tid >> 2;                        // row within a pass
  const int lchunk = tid & 3;
  // chunk position lchunk of (strip or tile) row r holds k-chunk
  // lchunk ^ ((r >> 2) & 3); LROWS is a multiple of 16, so the swizzle of a
  // lane is the same in every pass
  const int kc = lchunk ^ ((lrow >> 2) & 3);
  const float* a_lane = g.A + (m0 - halo + lrow) * g.a_pix_stride + kc * 4;
  const float* rb[B_ITERS];
#pragma unroll
  for (int it = 0; it < B_ITERS; ++it) {
    int n = tile_n * BN + it * LROWS + lrow;
    n = n < g.N ? n : g.N - 1;
    rb[it] = g.W3 + (long)n * g.Kp + kc * 4;
  }
  // strip piece `piece` = strip rows piece*LROWS + lrow of channel chunk `chunk`
  // Every k-step issues exactly ONE strip-side piece (so the vmcnt bookkeeping
  // is a constant and the k-loop has no data-dependent branches): pieces
  // 0..A_PIECES-1 fill the next chunk's strip, the rest are dummies that read
  // the zero page into a sink.
  auto issue_a = [&](int buf, int piece, int chunk) {
    const int j = piece * LROWS + lrow;
    const long P = m0 - halo + j;
    const bool real = piece < A_PIECES;
    const bool ok = real && P >= 0 && P < total_px && j < BM + 2 * halo;
    const float* src =
        ok ? a_lane + (long)piece * LROWS * g.a_pix_stride + chunk * BK : g.zero;
    float* dst = real ? As + buf * (SROWS * BK) + piece * (LROWS * BK) + wave * (16 * BK)
                      : Ds + wave * (16 * BK);
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src, (LDS_AS void*)dst,
                                     16, 0, 0);
  };
  auto issue_b = [&](int slot, int kt) {
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      float* dst = Bs + slot * (BN * BK) + it * (LROWS * BK) + wave * (16 * BK);
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(rb[it] + kt * BK),
                                       (LDS_AS void*)dst, 16, 0, 0);
    }
  };

  // ---- fragment addressing ---------------------------------------------------
  const int frow = lane & 31, fhalf = lane >> 5;
  const int chi = 2 * fhalf;
  int srow[TM];        // strip row of this lane's output pixel, centre tap
  unsigned vmask[TM];  // 9 validity bits (tap = 3*dy + dx)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * 128 + i * 32 + frow;
    long m = m0 + r;
    m = m < total_px ? m : total_px - 1;
    const int pix = (int)(m % ((long)H * Wd));
    const int y = pix / Wd, x = pix - y * Wd;
    unsigned mk = 0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const bool ok = (unsigned)(y + dy - 1) < (unsigned)H &&
                        (unsigned)(x + dx - 1) < (unsigned)Wd;
        mk |= (ok ? 1u : 0u) << (3 * dy + dx);
      }
    vmask[i] = mk;
    srow[i] = r + halo;
  }
  int boff[TN], bswz[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * 64 + j * 32 + frow;
    boff[j] = row * BK;
    bswz[j] = (row >> 2) & 3;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue --------------------------------------------------------------
  // same issue pattern as the steady state ([strip-side piece, weight piece,
  // weight piece] per k-step) so that ONE vmcnt constant is right from the
  // first iteration on
#pragma unroll
  for (int p = 0; p < A_PIECES; ++p) issue_a(0, p, 0);
#pragma unroll
  for (int t = 0; t < AHEAD; ++t) {
    if (t > 0) issue_a(0, A_PIECES, 0);  // dummy
    issue_b(t, t < nk ? t : nk - 1);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * (B_ITERS + 1)) : "memory");
  __builtin_amdgcn_s_barrier();

  int slot = 0, kt = 0;
  for (int c = 0; c < nchunks; ++c) {
    const float* Ab = As + (c & 1) * (SROWS * BK);
    const int cn = c + 1 < nchunks ? c + 1 : nchunks - 1;  // dummy refill at the end
    // The tap loop is deliberately NOT unrolled: unrolled, the compiler hoists
    // the 72 per-(tap, row-tile) fragment addresses out of the chunk loop and
    // spills; with a runtime tap they are a handful of VALU ops per fragment.
    int toff = -halo;  // (dy - 1) * W + (dx - 1), advanced incrementally
    int dx = 0;
    f32x4 ah[TM], al[TM];
    const int zoff = (int)(Zs - Ab);
    auto load_a = [&](int i, int tap, int tap_off) {
      const int sr = srow[i] + tap_off;
      const int live = sr * BK + ((chi ^ ((sr >> 2) & 3)) << 2);
      const int off = ((vmask[i] >> tap) & 1u) ? live : zoff;
      ah[i] = *reinterpret_cast<const f32x4*>(Ab + off);
      // lo half: chunk chi + 1 = the neighbouring 16 bytes (positions differ in
      // bit 0 only); the zero row is 64 B wide so the same flip stays inside it
      al[i] = *reinterpret_cast<const f32x4*>(Ab + (off ^ 4));
    };
    load_a(0, 0, toff);  // the strip of this chunk has landed (barrier above)
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
      // DMA pieces of this k-step (one strip piece of the next chunk during the
      // first A_PIECES taps, two weight pieces for the tile AHEAD k-steps from
      // now) are issued BETWEEN the MFMA groups, and the A fragments of row-tile
      // i+1 are fetched under row-tile i's MFMAs.
      int nslot = slot + AHEAD;
      nslot = nslot >= STAGES ? nslot - STAGES : nslot;
      const int ktn = kt + AHEAD < nk ? kt + AHEAD : nk - 1;
      const float* Bb = Bs + slot * (BN * BK);
      f32x4 bh[TN], bl[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const f32x4*>(Bb + boff[j] + ((chi ^ bswz[j]) << 2));
        bl[j] = *reinterpret_cast<const f32x4*>(Bb + boff[j] + (((chi + 1) ^ bswz[j]) << 2));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        // matrix pipe first: the three MFMAs of (i, 0) are in flight before the
        // next fragment's address arithmetic / the DMA issue of this group
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
            as_f16x8(ah[i]), as_f16x8(bl[0]), acc[i][0], 0, 0, 0);
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
            as_f16x8(al[i]), as_f16x8(bh[0]), acc[i][0], 0, 0, 0);
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
            as_f16x8(ah[i]), as_f16x8(bh[0]), acc[i][0], 0, 0, 0);
        if (i + 1 < TM) load_a(i + 1, t, toff);
        if (i == 0) issue_a((c + 1) & 1, t, cn);
        if (i == 1 || i == 2) {
          const int it = i - 1;
          float* dst = Bs + nslot * (BN * BK) + it * (LROWS * BK) + wave * (16 * BK);
          __builtin_amdgcn_global_load_lds(
              (const GLOBAL_AS void*)(rb[it] + ktn * BK), (LDS_AS void*)dst, 16, 0, 0);
        }
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
            as_f16x8(ah[i]), as_f16x8(bl[1]), acc[i][1], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
            as_f16x8(al[i]), as_f16x8(bh[1]), acc[i][1], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
            as_f16x8(ah[i]), as_f16x8(bh[1]), acc[i][1], 0, 0, 0);
      }
      // next tap's offset; its first A fragments can be fetched BEFORE the
      // barrier (the strip is complete for the whole chunk -- only the weight
      // tile needs the barrier), which takes them off the post-barrier bubble
      if (++dx == 3) { dx = 0; toff += Wd - 2; } else { ++toff; }
      if (t < 8) load_a(0, t + 1, toff);
      // tile kt+1 (and, at the last tap, the next strip) must have landed; what
      // may stay in flight is what was issued after tile kt+1's pieces: the
      // three pieces of each of the last AHEAD-1 k-steps
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * (B_ITERS + 1)) : "memory");
      __builtin_amdgcn_s_barrier();
      slot = slot + 1 == STAGES ? 0 : slot + 1;
      ++kt;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = acc[i][j] * g.acc_scale;
  run_epilogue<TM, TN>(g, acc, smem, wave, lane, tile_m * BM + wm * 128,
                       tile_n * BN + wn * 64);
}

template <int BN>
static int launch_conv3x3(const GemmArgs& g, hipStream_t s) {
  constexpr int NT = 2 * (BN / 64) * 64;
  const int tiles_m = (g.M + 255) / 256, tiles_n = (g.N + BN - 1) / BN;
  size_t lds = sizeof(float) * (size_t)(4 * BN * 16 + 2 * 384 * 16 + 16 + (NT / 64) * 256);
  const size_t stage_bytes = size_t(NT / 64) * 32 * 68 * sizeof(float);
  if (lds < stage_bytes) lds = stage_bytes;
  auto kern = conv3x3_split16_kernel<BN>;
  MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(NT), lds, s, g, tiles_m,
                     tiles_n);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

#endif  // MILAN_EXPERIMENTS (LDS-strip 3x3 kernel)

// ---------------------------------------------------------------------------
// fp32 <-> split-format conversion of a row-major matrix (rows x K, K % 8 == 0)
// ---------------------------------------------------------------------------
__global__ void f32_to_split_kernel(const float* __restrict__ src, long lds_,
                                    float* __restrict__ dst, long ldd, long rows,
                                    int K, float scale, unsigned* status) {
  const int g8 = K >> 3;
  const long total = rows * g8;
  float sat = 0.f;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / g8;
    const int q = idx - r * g8;
    const float* s = src + r * lds_ + q * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(s);
    const f32x4 b = *reinterpret_cast<const f32x4*>(s + 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e] * scale; v[4 + e] = b[e] * scale; }
    f32x4 hi, lo;
    split8(v, &hi, &lo, &sat);
    float* d = dst + r * ldd + q * 8;
    *reinterpret_cast<f32x4*>(d) = hi;
    *reinterpret_cast<f32x4*>(d + 4) = lo;
  }
  report_saturation(status, sat);
}

int launch_f32_to_split(const float* src, long ld_src, float* dst, long ld_dst,
                        long rows, int K, float scale, hipStream_t s) {
  MILAN_REQUIRE(K % 8 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0, MILAN_ERR_SHAPE,
                "f32_to_split: K=%d must be a multiple of 8", K);
  const long total = rows * (K >> 3);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(f32_to_split_kernel, dim3((int)blocks), dim3(256), 0, s, src,
                     ld_src, dst, ld_dst, rows, K, scale, status_word());
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------
template <int BM, int BN, int STAGES, bool CIN32, bool SPLIT>
static int launch_cfg(const GemmArgs& g, hipStream_t s) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  constexpr int NT = (BM / 64) * (BN / 64) * 64;
  size_t lds = size_t(STAGES) * (BM + BN) * 32 * sizeof(float);
  const size_t stage_bytes = size_t(NT / 64) * 32 * 68 * sizeof(float);
  if (lds < stage_bytes) lds = stage_bytes;
  auto kern = igemm_kernel<BM, BN, STAGES, CIN32, SPLIT>;
  MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(NT), lds, s, g,
                     tiles_m, tiles_n);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- live kernel timing (bench.py's roofline leg) ---------------------------
// While profiling is enabled, HIP events are recorded on the launch stream
// right before/after every GEMM launch and around every stage region
// (StageScope, common.h); durations are read back after a sync.  A GEMM record
// carries the stage it was launched in.
struct ProfRec {
  hipEvent_t a = nullptr, b = nullptr;
  int stage = 0;
  int kernel = 0;     // MILAN_KERNEL_* family of a GEMM-class launch
  bool gemm = false;
  double flops = 0.0;
  double bytes = 0.0;  // algorithmic HBM bytes of a GEMM launch
};

// What a launch has to move if every operand crosses HBM exactly once: the
// input pixels it reads (whole images for a k x k conv, the visited pixels for
// a 1x1), the second source, the weights, the residual / multiplicand and the
// output (fp32 and split format are both 4 B per element).
static double gemm_algorithmic_bytes(const GemmArgs& g) {
  const double M = g.M, N = g.N;
  const int k1 = g.A2 ? g.K1 : g.K;
  double a;
  if (g.KH == 1 && g.KW == 1) {
    a = M * (double)k1;
  } else {
    const double images = M / ((double)g.Ho * g.Wo);
    a = images * (double)g.H * g.Wd * g.Cin;
  }
  if (g.A2) a += M * (double)(g.K - g.K1);
  double c = M * N;
  double aux = g.aux ? M * N : 0.0;
  if (g.epilogue == EPI_LSTM) {  // N = 4 H: c read; h', c' (and split h') written
    const double H = N / 4;
    c = M * H * (g.Cs ? 3.0 : 2.0);
    aux = M * H;
  }
  if (g.epilogue == EPI_LSE) c = M * ((double)g.ldc + 1.0);  // statistics + the target's x
  if (g.f16) { c *= 0.5; aux *= 0.5; }   // (2-byte outputs; the K side is already in 4-byte units)
  return 4.0 * (a + (double)g.N * g.Kp + c + aux);
}
struct Profiler {
  bool on = false;
  std::vector<ProfRec> rec;
  size_t used = 0;
  int stage = MILAN_STAGE_OTHER;
  ProfRec* open = nullptr;  // the GEMM-class record being launched (profile_tag_kernel)
};
static Profiler g_prof;

// ---- status word of the calling context (common.h) ------------------------------------
static thread_local unsigned* t_status_word = nullptr;
__global__ void zero_fill_kernel(unsigned* __restrict__ p, size_t words) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    uint4* q = reinterpret_cast<uint4*>(p);
    const size_t quads = words >> 2;
    for (size_t j = i; j < quads; j += stride) q[j] = make_uint4(0u, 0u, 0u, 0u);
    for (size_t j = (quads << 2) + i; j < words; j += stride) p[j] = 0u;
  } else {
    for (; i < words; i += stride) p[i] = 0u;
  }
}

int launch_zero_fill(void* p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return 0;
  MILAN_REQUIRE(p != nullptr && (bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 3) == 0,
                MILAN_ERR_ARG, "zero_fill: %zu bytes at %p", bytes, p);
  const size_t words = bytes >> 2;
  const size_t want = (words / 4 + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
  hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<unsigned*>(p), words);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

void set_status_word(unsigned* word) { t_status_word = word; }
unsigned* status_word() { return t_status_word; }
void profile_tag_kernel(int family) {
  if (g_prof.on && g_prof.open) g_prof.open->kernel = family;
}

static ProfRec* prof_next() {
  if (g_prof.used == g_prof.rec.size()) {
    ProfRec r;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess)
      return nullptr;
    g_prof.rec.push_back(r);
  }
  return &g_prof.rec[g_prof.used++];
}

// hooks for launches made outside this file (chain.hip): a GEMM-class record
void* gemm_profile_begin(double flops, double bytes, hipStream_t s) {
  if (!g_prof.on) return nullptr;
  ProfRec* e = prof_next();
  if (!e) return nullptr;
  e->stage = g_prof.stage; e->gemm = true; e->flops = flops; e->bytes = bytes;
  e->kernel = MILAN_KERNEL_OTHER;
  g_prof.open = e;
  (void)hipEventRecord(e->a, s);
  return e;
}
void gemm_profile_end(void* rec, hipStream_t s) {
  if (rec) (void)hipEventRecord(static_cast<ProfRec*>(rec)->b, s);
  g_prof.open = nullptr;
}

int gemm_profile_enable(int enable) {
  g_prof.on = enable != 0;
  g_prof.used = 0;
  g_prof.stage = MILAN_STAGE_OTHER;
  g_prof.open = nullptr;
  return 0;
}

bool gemm_profile_active() { return g_prof.on; }

StageScope::StageScope(int stage, hipStream_t s) : stream_(s) {
  prev_ = g_prof.stage;
  g_prof.stage = stage;
  idx_ = -1;
  if (!g_prof.on) return;
  ProfRec* r = prof_next();
  if (!r) return;
  r->stage = stage; r->gemm = false; r->flops = 0.0; r->bytes = 0.0;
  idx_ = (long)(g_prof.used - 1);
  (void)hipEventRecord(r->a, s);
}

StageScope::~StageScope() {
  g_prof.stage = prev_;
  if (idx_ >= 0 && (size_t)idx_ < g_prof.used)
    (void)hipEventRecord(g_prof.rec[idx_].b, stream_);
}

int gemm_profile_read(double* ms, double* flops, long long* launches) {
  MILAN_CHECK_HIP(hipDeviceSynchronize());
  double total = 0.0, fl = 0.0;
  long long n = 0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    const ProfRec& r = g_prof.rec[i];
    if (!r.gemm) continue;
    float t = 0.f;
    MILAN_CHECK_HIP(hipEventElapsedTime(&t, r.a, r.b));
    total += t; fl += r.flops; ++n;
  }
  if (ms) *ms = total;
  if (flops) *flops = fl;
  if (launches) *launches = n;
  return 0;
}

// table[family][0..3] = ms, algorithmic flops, launches, algorithmic HBM bytes
int profile_read_kernels(double* table) {
  MILAN_CHECK_HIP(hipDeviceSynchronize());
  for (int i = 0; i < MILAN_KERNEL_COUNT * 4; ++i) table[i] = 0.0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    const ProfRec& r = g_prof.rec[i];
    if (!r.gemm || r.kernel < 0 || r.kernel >= MILAN_KERNEL_COUNT) continue;
    float t = 0.f;
    MILAN_CHECK_HIP(hipEventElapsedTime(&t, r.a, r.b));
    double* row = table + r.kernel * 4;
    row[0] += t; row[1] += r.flops; row[2] += 1.0; row[3] += r.bytes;
  }
  return 0;
}

// table[stage][0..4] = region ms, region count, gemm ms, gemm flops, gemm launches
int profile_read_stages(double* table) {
  MILAN_CHECK_HIP(hipDeviceSynchronize());
  for (int i = 0; i < MILAN_STAGE_COUNT * 6; ++i) table[i] = 0.0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    const ProfRec& r = g_prof.rec[i];
    if (r.stage < 0 || r.stage >= MILAN_STAGE_COUNT) continue;
    float t = 0.f;
    MILAN_CHECK_HIP(hipEventElapsedTime(&t, r.a, r.b));
    double* row = table + r.stage * 6;
    if (r.gemm) { row[2] += t; row[3] += r.flops; row[4] += 1.0; row[5] += r.bytes; }
    else { row[0] += t; row[1] += 1.0; }
  }
  return 0;
}

template <int BM, int BN, int STAGES, int SHAPE>
static int launch_split16_impl(const GemmArgs& g, hipStream_t s) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  constexpr int NT = (BM / 128) * (BN / 64) * 64;
  size_t lds = size_t(STAGES) * (BM + BN) * 16 * sizeof(float);
  const size_t stage_bytes = size_t(NT / 64) * 32 * 68 * sizeof(float);
  if (lds < stage_bytes) lds = stage_bytes;
  auto kern = igemm_split16_kernel<BM, BN, STAGES, SHAPE>;
  MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(kern), (int)lds));
  int grid = tiles_m * tiles_n;
#if MILAN_EXPERIMENTS
  // MILAN_PERSIST=<workgroups>: persistent grid (a multiple of 8)
  static int persist = -1;
  if (persist < 0) { const char* e = getenv("MILAN_PERSIST"); persist = e ? atoi(e) : 0; }
  // (4-wave tiles run two workgroups per CU)
  const int pgrid = persist / 8 * 8 * (NT == 256 ? 2 : 1);
  if (persist >= 8 && grid > pgrid) grid = pgrid;
#endif
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, g, tiles_m, tiles_n);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

template <int BM, int BN, int STAGES>
static int launch_split16_tm2(const GemmArgs& g, hipStream_t s) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  constexpr int NT = (BM / 64) * (BN / 64) * 64;
  size_t lds = size_t(STAGES) * (BM + BN) * 16 * sizeof(float);
  const size_t stage_bytes = size_t(NT / 64) * 32 * 68 * sizeof(float);
  if (lds < stage_bytes) lds = stage_bytes;
  auto kern = igemm_split16_tm2_kernel<BM, BN, STAGES>;
  MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(NT), lds, s, g, tiles_m,
                     tiles_n);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

// the ping-pong form of the 256 x 256 tile (4-slot ring: 128 KB of LDS)
// ... staged in k-tile pairs (whole 128-byte lines): A ring 3 x 32 KB + W ring 2 x 32 KB
// k x k convs whose weights exist in (slice, tap, channel) order run in that order
// (MILAN_TAP_INNER=0: tap-major, the round-4 bits; A/B timing and traffic)
static bool tap_inner_wanted(const GemmArgs& g) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("MILAN_TAP_INNER"); on = e ? atoi(e) : 1; }
  // (Cin in 4-byte units: whole 32-slot slices, which is what slice_major_kernel packed)
  return on && g.Wt && !g.A2 && g.KH * g.KW > 1 && g.KH * g.KW <= 32 && g.K == g.Kp &&
         g.Cin % 32 == 0;
}

template <int BNW>
static int launch_split16_pp32(const GemmArgs& g, hipStream_t s) {
  const int tiles_m = (g.M + 255) / 256, tiles_n = (g.N + BNW - 1) / BNW;
  const size_t lds = size_t(5) * 256 * 32 * sizeof(float);
  if (tap_inner_wanted(g)) {
    GemmArgs t = g;
    t.W = g.Wt;
    t.tap_inner = 1;
    profile_tag_kernel(BNW == 256 ? MILAN_KERNEL_PP32T_256 : MILAN_KERNEL_PP32_128);
    auto kern = igemm_split16_pp32t_kernel<BNW>;
    MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, s, t, tiles_m, tiles_n);
    MILAN_CHECK_HIP(hipGetLastError());
    return 0;
  }
  int ncus = 0;
  MILAN_TRY(device_cus8(&ncus));
  int grid = tiles_m * tiles_n;
  // MILAN_PP_PERSIST=1: persistent workgroups with the next tile's first pairs prefetched
  // under the epilogue (=2: only the K <= 512 launches).  Measured slower in every form --
  // static walk, short-K only, dynamic tile queue (profiles/r4_experiments.txt B, H, K): one
  // tile per workgroup (hardware dispatch) is the default.
  static int persist = -1;
  if (persist < 0) { const char* e = getenv("MILAN_PP_PERSIST"); persist = e ? atoi(e) : 0; }
  const bool walk = persist && (persist != 2 || g.K <= 512) && grid > ncus;
  profile_tag_kernel(BNW == 256 ? MILAN_KERNEL_PP32_256 : MILAN_KERNEL_PP32_128);
  if (BNW == 256 && !walk) {
    auto kern = igemm_split16_pp32_kernel;
    MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, g, tiles_m, tiles_n);
  } else {
    auto kern = igemm_split16_pp32n_kernel<BNW>;
    MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(kern), (int)lds));
    if (walk) grid = ncus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, g, tiles_m, tiles_n);
  }
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

template <int BNW>
static int launch_f16_pp32(const GemmArgs& g, hipStream_t s) {
  const int tiles_m = (g.M + 255) / 256, tiles_n = (g.N + BNW - 1) / BNW;
  const size_t lds = size_t(5) * 256 * 32 * sizeof(float);
  profile_tag_kernel(MILAN_KERNEL_F16);
  if (tap_inner_wanted(g)) {
    GemmArgs t = g;
    t.W = g.Wt;
    t.tap_inner = 1;
    auto kern = igemm_f16_pp32_kernel<BNW, true>;
    MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(kern), (int)lds));
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, s, t, tiles_m, tiles_n);
    MILAN_CHECK_HIP(hipGetLastError());
    return 0;
  }
  auto kern = igemm_f16_pp32_kernel<BNW, false>;
  MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, s, g, tiles_m, tiles_n);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

static int launch_split16_pp(const GemmArgs& g, hipStream_t s) {
  const int tiles_m = (g.M + 255) / 256, tiles_n = (g.N + 255) / 256;
  const size_t lds = size_t(4) * (256 + 256) * 16 * sizeof(float);
  static_assert(size_t(4) * 512 * 16 * sizeof(float) >= size_t(8) * 32 * 68 * sizeof(float),
                "the epilogue stages through the ring");
  auto kern = igemm_split16_pp_kernel;
  MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(kern), (int)lds));
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, s, g, tiles_m, tiles_n);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

// per-launch limits of the ping-pong kernel's 32-bit buffer offsets: a tile's rows span
// less than 2 GB of the input, the weight panel of a tile less than 2 GB
static bool pp_eligible(const GemmArgs& g) {
  if (g.Cin % 16 != 0 || g.aniso) return false;
  if (g.A2 && (g.K1 % 16 != 0)) return false;
  const long howo = (long)g.Ho * g.Wo;
  const long imgs = 256 / (howo > 0 ? howo : 1) + 2;  // images a 256-row tile can touch
  const long a_span = imgs * g.a_img_stride * 4 + 64;
  const long a2_span = g.A2 ? imgs * g.a2_img_stride * 4 + 64 : 0;
  const long w_span = 256L * g.Kp * 4;
  return a_span < 0x7fffffffL && a2_span < 0x7fffffffL && w_span < 0x7fffffffL;
}

template <int BM, int BN, int STAGES>
static int launch_split16(const GemmArgs& g, hipStream_t s) {
  if constexpr (BM == 256 && BN == 128 && STAGES == 3) {
    // MILAN_PP128=0: the round-3 lockstep 256 x 128 kernel (same bits; A/B timing)
    static int pp128 = -1;
    if (pp128 < 0) { const char* e = getenv("MILAN_PP128"); pp128 = e ? atoi(e) : 1; }
    if (pp128 && !g.chunk_major && pp_eligible(g) && g.Cin % 32 == 0 &&
        (!g.A2 || g.K1 % 32 == 0) && ((long)g.H + g.pad) < 32768 && ((long)g.Wd + g.pad) < 32768)
      return launch_split16_pp32<128>(g, s);
  }
  if constexpr (BM == 256 && BN == 256 && STAGES == 5) {
    static int pp = -1;  // MILAN_PP=0: the round-3 lockstep kernels (same bits; A/B timing)
    if (pp < 0) { const char* e = getenv("MILAN_PP"); pp = e ? atoi(e) : 1; }
    if (pp && !g.chunk_major && pp_eligible(g)) {
      // pp == 2: the 16-slot form everywhere (A/B timing)
      if (pp != 2 && g.Cin % 32 == 0 && (!g.A2 || g.K1 % 32 == 0) &&
          ((long)g.H + g.pad) < 32768 && ((long)g.Wd + g.pad) < 32768)
        return launch_split16_pp32<256>(g, s);
      return launch_split16_pp(g, s);
    }
  }
#if MILAN_EXPERIMENTS
  static int shape = -1;
  if (shape < 0) { const char* e = getenv("MILAN_SCHED"); shape = e ? atoi(e) : 0; }
  if (shape == 1) return launch_split16_impl<BM, BN, STAGES, 1>(g, s);
#endif
#if MILAN_EXPERIMENTS
  if (g.chunk_major) return launch_split16_impl<BM, BN, STAGES, 2>(g, s);
  static int lin = -1;
  if (lin < 0) { const char* e = getenv("MILAN_LINEAR_KERNEL"); lin = e ? atoi(e) : 1; }
  if (!lin) return launch_split16_impl<BM, BN, STAGES, 0>(g, s);
#endif
  // 1x1 convolutions and Linear layers: the loop variant without tap arithmetic
  if (g.KH == 1 && g.KW == 1 && g.pad == 0 && g.A2 == nullptr) {
    if constexpr (BM == 256 && BN == 256 && STAGES == 5) {
      // more tiles than CUs: persistent workgroups, the next tile's A rows prefetched
      // under the epilogue (igemm_split16_linp_kernel)
      int cus = 0;
      MILAN_TRY(device_cus8(&cus));
      bool on = true;
#if MILAN_EXPERIMENTS
      static int plin = -1;
      if (plin < 0) { const char* e = getenv("MILAN_PERSIST_LIN"); plin = e ? atoi(e) : 1; }
      on = plin != 0;
#endif
      const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
      if (on && g.Kp / 16 >= STAGES - 1 && tiles_m * tiles_n > cus) {
        const size_t lds = size_t(STAGES) * (BM + BN) * 16 * sizeof(float);
        auto kern = igemm_split16_linp_kernel<STAGES>;
        MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(kern), (int)lds));
        hipLaunchKernelGGL(kern, dim3(cus), dim3(512), lds, s, g, tiles_m, tiles_n);
        MILAN_CHECK_HIP(hipGetLastError());
        return 0;
      }
    }
    return launch_split16_impl<BM, BN, STAGES, 3>(g, s);
  }
  return launch_split16_impl<BM, BN, STAGES, 0>(g, s);
}

static bool aligned16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}

#if MILAN_EXPERIMENTS
static int env_tile_hint() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MILAN_TILE_HINT");  // experiments only
    v = e ? atoi(e) : 0;
  }
  return v;
}

// experiments only: MILAN_TILE_OVERRIDE="N:K:hint,N:K:hint" forces a tile
// configuration for the layers with that (N, K)
static int env_tile_override(int N, int K) {
  static std::vector<int> table;
  static bool parsed = false;
  if (!parsed) {
    parsed = true;
    const char* e = getenv("MILAN_TILE_OVERRIDE");
    while (e && *e) {
      int n = 0, k = 0, h = 0, used = 0;
      if (sscanf(e, "%d:%d:%d%n", &n, &k, &h, &used) == 3) {
        table.push_back(n); table.push_back(k); table.push_back(h);
        e += used;
        if (*e == ',') ++e;
      } else {
        break;
      }
    }
  }
  for (size_t i = 0; i + 2 < table.size(); i += 3)
    if (table[i] == N && table[i + 1] == K) return table[i + 2];
  return 0;
}

// The LDS-strip 3x3 kernel is OFF by default (MILAN_CONV3X3=1 or tile_hint 8
// turn it on).  MEASURED in round 2 (profiles/r2_conv3x3_strip_experiment.txt):
// it cuts the HBM fetch of layer3's 3x3 convs from 6.4 GB to under 1 GB per
// launch and the A-side DMA instructions by 4x, results pass the same parity
// tests -- and the layers run 8-13 % SLOWER (l3.x.c2 51.8 vs 46.7 ms per
// pass): the generic kernel's loop-invariant fragment addresses keep its
// MFMA / ds_read stream tighter than the per-tap address arithmetic here, and
// the L2/MALL absorbs the re-reads well enough that the saved traffic was not
// on the critical path.
static bool conv3x3_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MILAN_CONV3X3"); v = e ? atoi(e) != 0 : 0; }
  return v != 0;
}

#else
static int env_tile_hint() { return 0; }
static int env_tile_override(int, int) { return 0; }
#endif

static int launch_gemm_impl(GemmArgs g, hipStream_t s) {
  const bool cin32 = (g.Cin % 32 == 0);
  if (g.status == nullptr) g.status = status_word();
  if (g.tile_hint == 0) g.tile_hint = env_tile_hint();
  if (const int o = env_tile_override(g.N, g.K)) g.tile_hint = o;
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("MILAN_ABLATE"); dbg = e ? atoi(e) : 0; }
    g.debug = dbg;  // 1: skip MFMA phase, 2: skip DMA (timing experiments only)
#if MILAN_EXPERIMENTS
    static int stg = -1;
    if (stg < 0) { const char* e = getenv("MILAN_STAGGER"); stg = e ? atoi(e) : 0; }
    // the expand convs (residual / two-source, epilogue-heavy launches)
    static int stg_all = -1;
    if (stg_all < 0) { const char* e = getenv("MILAN_STAGGER_ALL"); stg_all = e ? atoi(e) : 0; }
    g.stagger = (stg_all || (g.out_split && g.N >= 256 && g.KH == 1 && g.H > 1 && (g.aux || g.A2))) ? stg : 0;
#endif
  }
  // pick the epilogue form
  if (g.epilogue == EPI_LSE) {
    MILAN_REQUIRE(!g.out_split && !g.aux_split && !g.f16 && g.aux == nullptr && g.N % 4 == 0 &&
                      g.ldc == 2 * ((g.N + 63) / 64) && g.C && g.lse_tgt && g.lse_x &&
                      (g.bias == nullptr || aligned16(g.bias)),
                  MILAN_ERR_SHAPE, "gemm: bad log-sum-exp epilogue geometry (N=%d ldc=%d)", g.N,
                  g.ldc);
    g.out_mode = OUT_VEC4;
  } else if (g.epilogue == EPI_LSTM) {
    MILAN_REQUIRE(!g.out_split && !g.aux_split && g.N % 64 == 0 && g.ldc % 8 == 0 &&
                      g.ldaux % 4 == 0 && g.C && g.C2 && g.aux && aligned16(g.C) &&
                      aligned16(g.C2) && aligned16(g.aux) &&
                      (g.Cs == nullptr || aligned16(g.Cs)) &&
                      (g.bias == nullptr || aligned16(g.bias)),
                  MILAN_ERR_SHAPE, "gemm: bad LSTM epilogue geometry (N=%d ldc=%d)",
                  g.N, g.ldc);
    g.out_mode = OUT_VEC4;
  } else if (g.f16) {
    // fast mode: plain f16 operands through the ping-pong tile only (K-side quantities in
    // 4-byte units, see GemmArgs::f16)
    MILAN_REQUIRE(g.a_split && !g.out_split && !g.aux_split && g.N % 8 == 0 && g.ldc % 4 == 0 &&
                      aligned16(g.C) && (g.bias == nullptr || aligned16(g.bias)) &&
                      (g.aux == nullptr || (g.ldaux % 4 == 0 && aligned16(g.aux))) &&
                      (g.epilogue == EPI_BIAS || g.epilogue == EPI_BIAS_RELU ||
                       g.epilogue == EPI_BIAS_RES_RELU) &&
                      g.Cin % 32 == 0 && (!g.A2 || g.K1 % 32 == 0) && g.N >= 128 &&
                      !g.chunk_major && ((long)g.H + g.pad) < 32768 && ((long)g.Wd + g.pad) < 32768,
                  MILAN_ERR_SHAPE, "gemm: unsupported fast-mode (f16) layer N=%d Cin=%d epi=%d",
                  g.N, g.Cin, g.epilogue);
    g.out_mode = OUT_F16;
  } else if (g.out_split || g.aux_split) {
    MILAN_REQUIRE(g.out_split && g.N % 8 == 0 && g.ldc % 8 == 0 && aligned16(g.C) &&
                      (g.bias == nullptr || aligned16(g.bias)) &&
                      (g.aux == nullptr ||
                       ((g.aux_split || g.epilogue == EPI_BIAS_SIGMUL) &&
                        g.ldaux % 8 == 0 && aligned16(g.aux))) &&
                      (g.epilogue == EPI_BIAS || g.epilogue == EPI_BIAS_RELU ||
                       g.epilogue == EPI_BIAS_RES_RELU ||
                       g.epilogue == EPI_BIAS_ADD ||
                       (g.epilogue == EPI_BIAS_SIGMUL && !g.aux_split)),
                  MILAN_ERR_SHAPE,
                  "gemm: unsupported split-format epilogue (N=%d ldc=%d epi=%d)",
                  g.N, g.ldc, g.epilogue);
    g.out_mode = OUT_SPLIT8;
  } else {
    const bool vec_ok = (g.N % 4 == 0) && (g.ldc % 4 == 0) && aligned16(g.C) &&
                        (g.bias == nullptr || aligned16(g.bias)) &&
                        (g.aux == nullptr ||
                         ((g.ldaux % 4 == 0) && aligned16(g.aux)));
    g.out_mode = vec_ok ? OUT_VEC4 : OUT_SCALAR;
  }
  MILAN_REQUIRE(g.A2 == nullptr || (g.a_split && g.N > 64 &&
                                    g.KH == 1 && g.KW == 1 && g.K1 % 16 == 0 &&
                                    (g.tile_hint == 0 || g.tile_hint >= 3)),
                MILAN_ERR_SHAPE, "gemm: two-source A needs the split16 kernels");
  if (g.a_split) {
    if (g.acc_scale == 0.f) g.acc_scale = 1.f;
    if (g.f16) {
      MILAN_REQUIRE(pp_eligible(g), MILAN_ERR_SHAPE, "gemm: fast-mode layer exceeds the 32-bit "
                    "buffer offsets of the ping-pong kernel");
      return g.N % 256 == 0 ? launch_f16_pp32<256>(g, s) : launch_f16_pp32<128>(g, s);
    }
    if (!cin32) {
      // per-lane taps (8-slot groups): only the narrow-N tile is built for it
      MILAN_REQUIRE(g.Cin % 8 == 0 && g.N <= 64 && !g.A2, MILAN_ERR_SHAPE,
                    "gemm: split-f16 operands with Cin %% 32 != 0 need "
                    "Cin %% 8 == 0 and N <= 64 (Cin=%d N=%d)", g.Cin, g.N);
      return launch_cfg<256, 64, 2, false, true>(g, s);
    }
    // (N <= 64 on the split16 pipeline -- 512x64x3 / 256x64x4 tiles -- measured
    // 10-20 % SLOWER in round 2: these layers are bound by the L2 -> LDS operand
    // traffic of a 64-column tile, not by the pipeline depth.)
    // N <= 64: the k x k convs (layer1's 3x3) run on the split16 pipeline with
    // 64 x 64 wave tiles (256x64 block, 3-deep ring, 2 blocks per CU): round 2,
    // same-box A/B: l1.x.c2 10.6 -> 9.7 ms per pass.  The 1x1 layers measured
    // 3-8 % SLOWER on it (and a 128x64 block slower still), so they stay on the
    // 2-stage kernel; tile_hint 6 / 10 force either.
    if (g.N <= 64 && g.tile_hint != 10 && !g.A2 &&
        (g.tile_hint == 6 || (g.tile_hint == 0 && g.KH * g.KW > 1)))
      return launch_split16_tm2<256, 64, 3>(g, s);
    if (g.N <= 64) return launch_cfg<256, 64, 2, true, true>(g, s);
    MILAN_REQUIRE(!g.chunk_major || (MILAN_EXPERIMENTS && g.tile_hint == 0 && !g.A2 && g.Cin % 16 == 0),
                  MILAN_ERR_SHAPE, "gemm: chunk-major k order: experiments build, split16 kernels");
#if MILAN_EXPERIMENTS
    // 3x3 / stride 1 with the chunk-major weight copy: input strip in LDS
    if (g.W3 && g.KH == 3 && g.KW == 3 && g.stride == 1 && g.pad == 1 && !g.A2 &&
        g.Cin % 16 == 0 && g.a_pix_stride == g.Cin && g.Ho == g.H && g.Wo == g.Wd &&
        g.a_img_stride == (long)g.H * g.Wd * g.Cin && g.Wd + 1 <= 64 &&
        (g.tile_hint == 8 || (g.tile_hint == 0 && conv3x3_enabled()))) {
      if (g.N % 256 == 0) return launch_conv3x3<256>(g, s);
      if (g.N % 128 == 0) return launch_conv3x3<128>(g, s);
    }
#endif
    // Measured on the 4096-neuron workload (profiles/): the 4-wave 256x128
    // tile with 16-slot k-tiles, a 3-deep ring and DMA pieces interleaved with
    // the MFMA groups (2 workgroups per CU) is the fastest split-mode
    // configuration for every N > 64 layer; the others stay reachable through
    // tile_hint for experiments.
    // The kernel is chosen from the LAYER (N, K) only, never from the number
    // of rows: different tile configurations accumulate in different orders,
    // and a description must not depend on how many neurons shared its launch
    // (chunk size, world size).  Short M just leaves tile rows masked.
#if MILAN_EXPERIMENTS
    if (g.tile_hint == 3) return launch_split16<256, 256, 4>(g, s);
    if (g.tile_hint == 7) return launch_split16<256, 256, 5>(g, s);
    if (g.tile_hint == 4) return launch_split16<256, 128, 3>(g, s);
    if (g.tile_hint == 5) return launch_split16<128, 256, 3>(g, s);
    if (g.tile_hint == 1) return launch_cfg<256, 128, 3, true, true>(g, s);
    if (g.tile_hint == 2) return launch_cfg<128, 128, 2, true, true>(g, s);
#endif
    // 256x256 tile, 5-deep ring = all 160 KB of LDS, four k-tiles in flight (round 3,
    // same-box A/B against the 4-deep ring: +0.5 % end to end, same bits)
    if (g.N % 256 == 0) return launch_split16<256, 256, 5>(g, s);
    {
      // wide outputs that are not a multiple of 256 (the vocabulary GEMMs, N =
      // 5004): the 256-column tile when it pads no more than the 128-column one
      // would (same-box A/B: decode stage -0.5 ms; N = 3904 pads 4.9 % vs 1.6 %
      // and is slower on it)
      const int pad256 = (g.N + 255) / 256 * 256 - g.N;
      const int pad128 = (g.N + 127) / 128 * 128 - g.N;
      if (g.N > 2048 && pad256 <= pad128)
        return launch_split16<256, 256, 5>(g, s);
    }
    return launch_split16<256, 128, 3>(g, s);
  }
  if (g.N <= 64) {
    return cin32 ? launch_cfg<256, 64, 2, true, false>(g, s)
                 : launch_cfg<256, 64, 2, false, false>(g, s);
  }
  // (the 8-wave 3-stage tile measured 4% slower than 128x128x2 in F32 mode,
  // where one k-tile is 4096 MFMA cycles and latency is already hidden)
#if MILAN_EXPERIMENTS
  if (cin32 && g.tile_hint == 1)
    return launch_cfg<256, 128, 3, true, false>(g, s);
#endif
  return cin32 ? launch_cfg<128, 128, 2, true, false>(g, s)
               : launch_cfg<128, 128, 2, false, false>(g, s);
}

int launch_gemm(const GemmArgs& g, hipStream_t s) {
  MILAN_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, MILAN_ERR_SHAPE,
                "gemm: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
  MILAN_REQUIRE(g.Cin % 4 == 0 && g.a_pix_stride % 4 == 0 &&
                    g.a_img_stride % 4 == 0 && g.Kp % 32 == 0 && aligned16(g.A) &&
                    aligned16(g.W),
                MILAN_ERR_SHAPE,
                "gemm: Cin=%d / strides must be multiples of 4 floats and "
                "operands 16-byte aligned", g.Cin);
  if (!g_prof.on) return launch_gemm_impl(g, s);
  ProfRec* e = prof_next();
  MILAN_REQUIRE(e != nullptr, MILAN_ERR_STATE, "profiler: cannot create events");
  e->stage = g_prof.stage;
  e->gemm = true;
  e->kernel = g.a_split ? MILAN_KERNEL_SPLIT_OTHER : MILAN_KERNEL_F32;
  e->flops = 2.0 * (double)g.M * (double)g.N * (double)(g.flop_k > 0 ? g.flop_k : g.K);
  e->bytes = gemm_algorithmic_bytes(g);
  g_prof.open = e;
  MILAN_CHECK_HIP(hipEventRecord(e->a, s));
  const int r = launch_gemm_impl(g, s);
  g_prof.open = nullptr;
  MILAN_CHECK_HIP(hipEventRecord(e->b, s));
  return r;
}

}  // namespace milan
