// Fused ResNet stem for the split-f16 mode: conv1 (7x7 / 2) + bn1 + ReLU +
// maxpool (3x3 / 2, pad 1) in ONE persistent launch.
//
// Reference call sites: torchvision ResNet.forward's conv1 / bn1 / relu / maxpool as
// driven by src/milan/encoders.py:286-320 (PyramidConvEncoder.forward; the RAW conv1
// output is pyramid level 0, the pooled tensor feeds layer1).
//
// As three launches (encoder.hip: implicit-GEMM conv1 -> fp32 raw tensor, masked
// pooling of it, bn+relu+maxpool -> split format) the 112 x 112 x 64 fp32 conv1
// output of every image is written (12.5 x the input bytes) and read back in full,
// and the GEMM gathers every input group 7 x 4 times through the texture path.
// Here a workgroup owns a 7 x 8 tile of POOLED pixels: it stages the 35 x 20 input
// groups behind it in LDS once (global_load_lds), computes the 15 x 17 conv1 pixels
// under the pooling windows on the matrix cores with the A fragments read straight
// from that LDS tile (no im2col), parks the fp32 result in an LDS staging tile,
// writes (a) the raw conv1 rows the mask-weighted pooling of level 0 will read --
// only inside the bounding box of the image's non-zero mask weights -- and (b) the
// bn + ReLU + max-pooled tile in split format.  The 64 x 224 weights live in
// registers as MFMA B fragments for the whole kernel (persistent workgroups, one
// per CU; 8 waves = 4 pixel-block pairs x 2 channel halves).
//
// Arithmetic is the igemm_kernel<.., SPLIT> sequence (gemm.hip): two accumulator
// sets (hi*hi | hi*lo + lo*hi), k ascending, (acc + accx) * scale, + bias; then the
// bn / ReLU / max sequence of bn_relu_maxpool_split_kernel -- so raw and pooled
// outputs are bitwise those of the three launches (tests/test_gpu_stem.py).
#include "common.h"

namespace milan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

namespace {

constexpr int kPoolR = 7, kPoolC = 8;            // pooled pixels per tile
constexpr int kConvR = 2 * kPoolR + 1;           // 15 conv rows (one halo row above)
constexpr int kConvC = 2 * kPoolC + 1;           // 17 conv columns (one halo column left)
constexpr int kInR = 2 * (kConvR - 1) + 7;       // 35 input rows
constexpr int kInC = kConvC + 3;                 // 20 pixel-pair groups per input row
constexpr int kPieces = kInR * kInC;             // 16-byte pieces per plane (hi / lo)
constexpr int kDma = (2 * kPieces + 63) / 64;    // wave-wide DMA instructions per tile
constexpr int kInBytes = kDma * 1024;            // one input buffer
constexpr int kSRow = 68;                        // staging row stride (floats)
constexpr int kSlabs = 14;                       // K = 224 slots = 14 x 16
static_assert(kConvR * kConvC <= 256, "conv tile = 8 MFMA row blocks");

__device__ inline f16x8 h8(f32x4 v) { return __builtin_bit_cast(f16x8, v); }

__device__ inline void stem_split8(const float* v, f32x4* hi_out, f32x4* lo_out) {
  f16x8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = fminf(fmaxf(v[e], -65504.f), 65504.f);
    const _Float16 hh = (_Float16)x;
    h[e] = hh;
    l[e] = (_Float16)(x - (float)hh);
  }
  *hi_out = __builtin_bit_cast(f32x4, h);
  *lo_out = __builtin_bit_cast(f32x4, l);
}

}  // namespace

__global__ __launch_bounds__(512, 1) void stem_fused_kernel(StemArgs a) {
  // roundings as written (the unfused path rounds acc * scale before the bias add
  // because an LDS round trip sits between them); the one fused multiply-add of the
  // bn step is spelled __builtin_fmaf
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) char stem_smem[];
  float* stg = reinterpret_cast<float*>(stem_smem + 2 * kInBytes);  // [256][kSRow]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mp = wave >> 1, nb = wave & 1;  // pixel-block pair, channel half
  const int half = lane >> 5;

  // ---- weights: B fragments of this wave's 32 output channels, all 14 slabs ----
  f32x4 bh[kSlabs], bl[kSlabs];
  {
    const float* wrow = a.ws + (long)(nb * 32 + (lane & 31)) * 224 + half * 8;
#pragma unroll
    for (int s = 0; s < kSlabs; ++s) {
      bh[s] = *reinterpret_cast<const f32x4*>(wrow + s * 16);
      bl[s] = *reinterpret_cast<const f32x4*>(wrow + s * 16 + 4);
    }
  }
  const float bias_n = a.bias ? a.bias[nb * 32 + (lane & 31)] : 0.f;

  // ---- loader: DMA instruction d = wave + 8 k moves pieces 64 d .. 64 d + 63 ----
  int ld_row[3], ld_col[3], ld_off[3];
  bool ld_ok[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int i = (wave + 8 * k) * 64 + lane;
    const int plane = i >= kPieces ? 1 : 0;
    const int gi = i - plane * kPieces;
    ld_ok[k] = i < 2 * kPieces;
    ld_row[k] = gi / kInC;
    ld_col[k] = gi - ld_row[k] * kInC;
    ld_off[k] = plane * 4;
  }
  const int tiles = a.tiles_y * a.tiles_x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  // tile sequence of this workgroup: XCD x walks images x, x + 8, ... tile by tile,
  // its workgroups taking consecutive tiles (halo rows / columns meet in that L2)
  auto decode = [&](int it, int* img, int* ta, int* tb) -> bool {
    const long q = (long)it * nslots + slot;
    const int gidx = (int)(q / tiles);
    const int ti = (int)(q - (long)gidx * tiles);
    *img = gidx * 8 + xcd;
    *ta = ti / a.tiles_x;
    *tb = ti - *ta * a.tiles_x;
    return *img < a.n;
  };
  auto issue = [&](int img, int ta, int tb, int buf) {
    const int iy0 = 2 * (2 * kPoolR * ta - 1) - 3, ig0 = 2 * kPoolC * tb - 2;
    char* dst = stem_smem + buf * kInBytes;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (wave + 8 * k < kDma) {
        const int iy = iy0 + ld_row[k], ig = ig0 + ld_col[k];
        const bool ok = ld_ok[k] && iy >= 0 && iy < a.H && ig >= 0 && ig < a.G;
        const float* src =
            ok ? a.in + (((long)img * a.H + iy) * a.G + ig) * 8 + ld_off[k] : a.zero;
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src,
                                         (LDS_AS void*)(dst + (wave + 8 * k) * 1024),
                                         16, 0, 0);
      }
    }
  };

  // ---- A fragment addresses: pixel t = 32 (2 mp + i) + lane % 32 of the tile ----
  int abase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int t = (2 * mp + i) * 32 + (lane & 31);
    t = t < kConvR * kConvC ? t : kConvR * kConvC - 1;
    const int pr = t / kConvC, pc = t - pr * kConvC;
    abase[i] = ((2 * pr) * kInC + pc + 2 * 0 + half) * 16;
  }
  // slab s = k-groups 2s, 2s+1: kernel row s/2, group column 2 (s%2) + half

  // ---- store-phase constants ------------------------------------------------
  // raw rows: iteration it handles pixel 32 it + tid/16, channels 4 (tid%16) ..
  unsigned raw_rc[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int t = it * 32 + (tid >> 4);
    const int pr = t / kConvC, pc = t - pr * kConvC;
    const bool own = t < kConvR * kConvC && pr >= 1 && pc >= 1;
    raw_rc[it] = own ? (unsigned)(pr << 8 | pc) : 0xffffu;
  }
  const int po = tid >> 3, c8 = tid & 7;
  const int py = po >> 3, px = po & 7;  // po < 56 for the pooling threads
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = a.scale[c8 * 8 + e];
    sh[e] = a.shift[c8 * 8 + e];
  }

  int img, ta, tb;
  bool have = decode(0, &img, &ta, &tb);
  if (have) issue(img, ta, tb, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int it = 0; have; ++it) {
    const int buf = it & 1;
    int nimg, nta, ntb;
    const bool nhave = decode(it + 1, &nimg, &nta, &ntb);
    if (nhave) issue(nimg, nta, ntb, buf ^ 1);

    // ---- conv1 on the matrix cores ----
    f32x16 acc[2], accx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accx[i][r] = 0.f; }
    const char* cur = stem_smem + buf * kInBytes;
#pragma unroll
    for (int s = 0; s < kSlabs; ++s) {
      const int off = ((s >> 1) * kInC + 2 * (s & 1)) * 16;
      f32x4 ah[2], al[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const f32x4*>(cur + abase[i] + off);
        al[i] = *reinterpret_cast<const f32x4*>(cur + abase[i] + off + kPieces * 16);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(ah[i]), h8(bh[s]), acc[i], 0, 0, 0);
        accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(ah[i]), h8(bl[s]), accx[i], 0, 0, 0);
        accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(al[i]), h8(bh[s]), accx[i], 0, 0, 0);
      }
    }

    // staging tile free again (every wave is past the previous tile's store phase)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const float v = (acc[i][r] + accx[i][r]) * a.acc_scale + bias_n;
        stg[((2 * mp + i) * 32 + row) * kSRow + nb * 32 + (lane & 31)] = v;
      }
    // next tile's input has had the whole MFMA loop to land
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- (a) raw conv1 rows for the level-0 masked pooling ----
    const int cr0 = 2 * kPoolR * ta - 1, cc0 = 2 * kPoolC * tb - 1;
    if (a.raw) {
      int y0 = 0, y1 = a.h1 - 1, x0 = 0, x1 = a.w1 - 1;
      if (a.bbox) {
        const int* bb = a.bbox + (long)img * 4;
        y0 = bb[0]; y1 = bb[1]; x0 = bb[2]; x1 = bb[3];
      }
      if (cr0 + kConvR > y0 && cr0 <= y1 && cc0 + kConvC > x0 && cc0 <= x1) {
        y1 = y1 < a.h1 - 1 ? y1 : a.h1 - 1;
        x1 = x1 < a.w1 - 1 ? x1 : a.w1 - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int pr = raw_rc[j] >> 8, pc = raw_rc[j] & 255;
          const int r = cr0 + pr, c = cc0 + pc;
          if (raw_rc[j] != 0xffffu && r >= y0 && r <= y1 && c >= x0 && c <= x1) {
            const int t = j * 32 + (tid >> 4);
            const f32x4 v =
                *reinterpret_cast<const f32x4*>(stg + t * kSRow + (tid & 15) * 4);
            *reinterpret_cast<f32x4*>(a.raw + (((long)img * a.h1 + r) * a.w1 + c) * 64 +
                                      (tid & 15) * 4) = v;
          }
        }
      }
    }
    // ---- (b) bn1 + ReLU + 3x3/2 max pooling -> split format ----
    {
      const int ho = kPoolR * ta + py, wo = kPoolC * tb + px;
      if (po < kPoolR * kPoolC && ho < a.hp && wo < a.wp) {
        float best[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int hi = ho * 2 - 1 + dy;
          if (hi < 0 || hi >= a.h1) continue;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int wi = wo * 2 - 1 + dx;
            if (wi < 0 || wi >= a.w1) continue;
            const float* p = stg + ((2 * py + dy) * kConvC + 2 * px + dx) * kSRow + c8 * 8;
            const f32x4 u = *reinterpret_cast<const f32x4*>(p);
            const f32x4 w = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              best[e] = fmaxf(best[e], fmaxf(__builtin_fmaf(u[e], sc[e], sh[e]), 0.f));
              best[4 + e] =
                  fmaxf(best[4 + e], fmaxf(__builtin_fmaf(w[e], sc[4 + e], sh[4 + e]), 0.f));
            }
          }
        }
        f32x4 hi4, lo4;
        stem_split8(best, &hi4, &lo4);
        float* d = a.y + ((((long)img * a.hp + ho) * a.wp + wo) * 8 + c8) * 8;
        *reinterpret_cast<f32x4*>(d) = hi4;
        *reinterpret_cast<f32x4*>(d + 4) = lo4;
      }
    }
    have = nhave; img = nimg; ta = nta; tb = ntb;
  }
}

bool stem_fused_supported(int cout, int Kp) { return cout == 64 && Kp == 224; }

int launch_stem_fused(const StemArgs& a0, hipStream_t s) {
  StemArgs a = a0;
  MILAN_REQUIRE(a.n > 0 && a.H > 0 && a.G > 0 && a.in && a.ws && a.y && a.scale &&
                    a.shift && a.zero,
                MILAN_ERR_ARG, "stem: missing operand");
  a.tiles_y = (a.hp + kPoolR - 1) / kPoolR;
  a.tiles_x = (a.wp + kPoolC - 1) / kPoolC;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    MILAN_CHECK_HIP(hipGetDevice(&dev));
    MILAN_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus = cus < 8 ? 8 : cus / 8 * 8;
  }
  const size_t lds = 2 * (size_t)kInBytes + sizeof(float) * 256 * kSRow;
  static bool attr_set = false;
  if (!attr_set) {
    MILAN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stem_fused_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
    attr_set = true;
  }
  // algorithmic work: 7x7x3 taps per conv1 output; bytes: input groups once, raw
  // fp32 out, pooled split out
  const double px1 = (double)a.n * a.h1 * a.w1, pxp = (double)a.n * a.hp * a.wp;
  void* rec = gemm_profile_begin(
      2.0 * px1 * 64 * 147,
      32.0 * a.n * a.H * a.G + (a.raw ? 256.0 * px1 : 0.0) + 256.0 * pxp, s);
  hipLaunchKernelGGL(stem_fused_kernel, dim3(cus), dim3(512), lds, s, a);
  gemm_profile_end(rec, s);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace milan
