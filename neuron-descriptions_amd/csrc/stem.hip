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

#include <cstdint>
#include <cstdlib>

namespace milan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// timing ablations (StemArgs::debug) exist in the experiments build only
#if MILAN_EXPERIMENTS
#define STEM_ABLATE(bit) (a.debug & (bit))
#else
#define STEM_ABLATE(bit) false
#endif

namespace {

constexpr int kSRow = 68;    // staging row stride (floats)
constexpr int kSlabs = 14;   // K = 224 slots = 14 x 16

// Tile geometry: PR x PC pooled pixels per workgroup of NW waves.
//   <7, 8, 8>: 15 x 17 conv pixels (255 of 256 MFMA rows), one workgroup per CU
//   <3, 8, 4>:  7 x 17 conv pixels (119 of 128), 59 KB of LDS: two workgroups per CU,
//              whose store phases (VALU / LDS / HBM) run under each other's MFMA phase
template <int PR, int PC, int NW>
struct StemTile {
  static constexpr int kPoolR = PR, kPoolC = PC;
  static constexpr int kConvR = 2 * PR + 1;            // one halo row above
  static constexpr int kConvC = 2 * PC + 1;            // one halo column left
  static constexpr int kPix = kConvR * kConvC;
  static constexpr int kMB = (kPix + 31) / 32;         // 32-pixel MFMA row blocks
  static constexpr int kMBW = kMB / (NW / 2);          // row blocks per wave
  static constexpr int kInR = 2 * (kConvR - 1) + 7;    // input rows
  static constexpr int kInC = kConvC + 3;              // pixel-pair groups per input row
  static constexpr int kPieces = kInR * kInC;          // 16-byte pieces per plane (hi / lo)
  static constexpr int kDma = (2 * kPieces + 63) / 64; // wave-wide DMA instructions per tile
  static constexpr int kDmaW = (kDma + NW - 1) / NW;   // per wave
  static constexpr int kInBytes = kDma * 1024;         // one input buffer
  static constexpr int kRawIt = kMB * 32 * 16 / (NW * 64);
  static constexpr size_t kLds = 2 * (size_t)kInBytes + sizeof(float) * kMB * 32 * kSRow;
  static_assert(kMB % (NW / 2) == 0, "row blocks split evenly over the waves");
  static_assert(PR * PC * 8 <= NW * 64, "one pooling work item per thread");
};

__device__ inline f16x8 h8(f32x4 v) { return __builtin_bit_cast(f16x8, v); }

__device__ inline void stem_split8(const float* v, f32x4* hi_out, f32x4* lo_out, float* sat) {
  split8_rne(v, hi_out, lo_out, sat);  // common.h
}

}  // namespace

template <int PR, int PC, int NW, bool U8>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void stem_fused_kernel(StemArgs a) {
  // roundings as written (the unfused path rounds acc * scale before the bias add
  // because an LDS round trip sits between them); the one fused multiply-add of the
  // bn step is spelled __builtin_fmaf
#pragma clang fp contract(off)
  using T = StemTile<PR, PC, NW>;
  constexpr int kPoolR = T::kPoolR, kPoolC = T::kPoolC, kConvR = T::kConvR, kConvC = T::kConvC;
  constexpr int kInC = T::kInC, kPieces = T::kPieces, kDma = T::kDma, kInBytes = T::kInBytes;
  constexpr int kMBW = T::kMBW;
  extern __shared__ __attribute__((aligned(16))) char stem_smem[];
  float* stg = reinterpret_cast<float*>(stem_smem + 2 * kInBytes);  // [32 kMB][kSRow]
  // U8: the byte -> (hi, lo) table behind the staging tile (kLutWords words)
  unsigned* lut = reinterpret_cast<unsigned*>(stem_smem + T::kLds);
  float sat = 0.f;  // (common.h: saturation of the split clamp is loud)

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

  // ---- loader: DMA instruction d = wave + NW k moves pieces 64 d .. 64 d + 63 ----
  int ld_row[T::kDmaW], ld_col[T::kDmaW], ld_off[T::kDmaW];
  bool ld_ok[T::kDmaW];
#pragma unroll
  for (int k = 0; k < T::kDmaW; ++k) {
    const int i = (wave + NW * k) * 64 + lane;
    const int plane = i >= kPieces ? 1 : 0;
    const int gi = i - plane * kPieces;
    ld_ok[k] = i < 2 * kPieces;
    ld_row[k] = gi / kInC;
    ld_col[k] = gi - ld_row[k] * kInC;
    ld_off[k] = plane * 4;
  }
  a.n = live_rows(a.n_live, 1, a.n);   // (image count on the device, GemmArgs::m_live)
  const int tiles = a.tiles_y * a.tiles_x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  // tile sequence of this workgroup: XCD x walks images x, x + 8, ... tile by tile,
  // its workgroups taking consecutive tiles (halo rows / columns meet in that L2).
  // (seq_g, seq_t) = (image group, tile in image) of the NEXT tile to decode.
  int seq_g = slot / tiles, seq_t = slot - seq_g * tiles;
  auto decode = [&](int* img, int* ta, int* tb) -> bool {
    *img = seq_g * 8 + xcd;
    *ta = seq_t / a.tiles_x;
    *tb = seq_t - *ta * a.tiles_x;
    seq_t += nslots;
    while (seq_t >= tiles) { seq_t -= tiles; ++seq_g; }
    return *img < a.n;
  };
  auto issue = [&](int img, int ta, int tb, int buf) {
    const int iy0 = 2 * (2 * kPoolR * ta - 1) - 3, ig0 = 2 * kPoolC * tb - 2;
    char* dst = stem_smem + buf * kInBytes;
#pragma unroll
    for (int k = 0; k < T::kDmaW; ++k) {
      if (wave + NW * k < kDma) {
        const int iy = iy0 + ld_row[k], ig = ig0 + ld_col[k];
        const bool ok = ld_ok[k] && iy >= 0 && iy < a.H && ig >= 0 && ig < a.G;
        const float* src =
            ok ? a.in + (((long)img * a.H + iy) * a.G + ig) * 8 + ld_off[k] : a.zero;
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src,
                                         (LDS_AS void*)(dst + (wave + NW * k) * 1024),
                                         16, 0, 0);
      }
    }
  };

  // ---- U8: thread t < kInR * kInC / 2 builds the pixel-pair groups 2j, 2j + 1 of tile row
  // t / (kInC / 2): their four pixels X0 + 4j .. + 3 (X0 = 3 mod 4) are byte 3 of one aligned
  // dword of the image row and bytes 0..2 of the next -- 6 dword loads (2 x 3 planes), 12 table
  // look-ups, the hi pieces at g, the lo pieces at kPieces + g (the layout the DMA produces).
  // Needs W % 4 == 0 and a 4-byte aligned image pointer (launcher).
  static_assert(kInC % 2 == 0, "pixel-pair groups come in pairs");
  constexpr int kHalfC = kInC / 2;
  const int u_row = tid / kHalfC, u_j = tid - u_row * kHalfC;
  const bool u_ok = U8 && tid < T::kInR * kHalfC;
  // ub[0..2] = dword A of planes R, G, B, ub[3..5] = dword B, ub[6] = validity (bit 0: A, bit 1: B)
  auto fetch = [&](int img, int ta, int tb, unsigned (&ub)[7]) {
    const int iy0 = 2 * (2 * kPoolR * ta - 1) - 3;
    const int xa = 4 * kPoolC * tb - 8 + 4 * u_j;  // first pixel of dword A; pixel P0 = xa + 3
    const int iy = iy0 + u_row;
    const long hw = (long)a.H * a.W;
    const bool row_ok = u_ok && iy >= 0 && iy < a.H;
    const bool ok_a = row_ok && xa >= 0 && xa < a.W, ok_b = row_ok && xa + 4 >= 0 && xa + 4 < a.W;
    const unsigned char* p = a.in_u8 + (long)(a.order ? a.order[img] : img) * 3 * hw + (long)iy * a.W + xa;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      ub[ch] = ok_a ? *reinterpret_cast<const unsigned*>(p + ch * hw) : 0u;
      ub[3 + ch] = ok_b ? *reinterpret_cast<const unsigned*>(p + ch * hw + 4) : 0u;
    }
    ub[6] = (ok_a ? 1u : 0u) | (ok_b ? 2u : 0u);
  };
  auto convert = [&](const unsigned (&ub)[7], int buf) {
    if (!u_ok) return;
    char* dst = stem_smem + buf * kInBytes;
    const bool ok_a = ub[6] & 1u, ok_b = ub[6] & 2u;
    unsigned e[4][3];  // table entries of pixels P0..P3, channels R G B (768 = 0.0: outside the image)
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      e[0][ch] = lut[ok_a ? ch * 256u + (ub[ch] >> 24) : 768u];
      e[1][ch] = lut[ok_b ? ch * 256u + (ub[3 + ch] & 255u) : 768u];
      e[2][ch] = lut[ok_b ? ch * 256u + ((ub[3 + ch] >> 8) & 255u) : 768u];
      e[3][ch] = lut[ok_b ? ch * 256u + ((ub[3 + ch] >> 16) & 255u) : 768u];
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      // group = (R0 G0 B0 0 R1 G1 B1 0): halfword j of a piece = value j
      const unsigned* p0 = e[2 * q];
      const unsigned* p1 = e[2 * q + 1];
      u32x4 hi, lo;
      hi[0] = (p0[0] & 0xffffu) | (p0[1] << 16); hi[1] = p0[2] & 0xffffu;
      hi[2] = (p1[0] & 0xffffu) | (p1[1] << 16); hi[3] = p1[2] & 0xffffu;
      lo[0] = (p0[0] >> 16) | (p0[1] & 0xffff0000u); lo[1] = p0[2] >> 16;
      lo[2] = (p1[0] >> 16) | (p1[1] & 0xffff0000u); lo[3] = p1[2] >> 16;
      const int g = u_row * kInC + 2 * u_j + q;
      *reinterpret_cast<u32x4*>(dst + g * 16) = hi;
      *reinterpret_cast<u32x4*>(dst + (kPieces + g) * 16) = lo;
    }
  };

  // ---- A fragment addresses: pixel t = 32 (2 mp + i) + lane % 32 of the tile ----
  int abase[kMBW];
#pragma unroll
  for (int i = 0; i < kMBW; ++i) {
    int t = (kMBW * mp + i) * 32 + (lane & 31);
    t = t < kConvR * kConvC ? t : kConvR * kConvC - 1;
    const int pr = t / kConvC, pc = t - pr * kConvC;
    abase[i] = ((2 * pr) * kInC + pc + half) * 16;
  }
  // slab s = k-groups 2s, 2s+1: kernel row s/2, group column 2 (s%2) + half

  // ---- store-phase constants ------------------------------------------------
  // raw rows: iteration it handles pixel 32 it + tid/16, channels 4 (tid%16) ..
  unsigned raw_rc[T::kRawIt];
#pragma unroll
  for (int it = 0; it < T::kRawIt; ++it) {
    const int t = it * (NW * 4) + (tid >> 4);
    const int pr = t / kConvC, pc = t - pr * kConvC;
    const bool own = t < kConvR * kConvC && pr >= 1 && pc >= 1;
    raw_rc[it] = own ? (unsigned)(pr << 8 | pc) : 0xffffu;
  }
  const int po = tid >> 3, c8 = tid & 7;
  const int py = po / kPoolC, px = po - py * kPoolC;  // po < PR * PC for the pooling threads
  float sc[8], sh[8], sg[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = a.scale[c8 * 8 + e];
    sh[e] = a.shift[c8 * 8 + e];
    sg[e] = sc[e] < 0.f ? -1.f : 1.f;
  }

  int img, ta, tb;
  bool have = decode(&img, &ta, &tb);
  unsigned ub[7];
  if constexpr (U8) {
    for (int i = tid; i < 769; i += NW * 64) lut[i] = a.lut[i];
    __syncthreads();
    if (have) {
      fetch(img, ta, tb, ub);
      convert(ub, 0);
    }
    __syncthreads();
  } else {
    if (have) issue(img, ta, tb, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  for (int it = 0; have; ++it) {
    const int buf = it & 1;
    int nimg, nta, ntb;
    const bool nhave = decode(&nimg, &nta, &ntb);
    if constexpr (!U8)
      if (nhave && !STEM_ABLATE(4)) issue(nimg, nta, ntb, buf ^ 1);

    // ---- conv1 on the matrix cores ----
    f32x16 acc[kMBW], accx[kMBW];
#pragma unroll
    for (int i = 0; i < kMBW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accx[i][r] = 0.f; }
    const char* cur = stem_smem + buf * kInBytes;
    if (!STEM_ABLATE(1))
#pragma unroll
    for (int s = 0; s < kSlabs; ++s) {
      const int off = ((s >> 1) * kInC + 2 * (s & 1)) * 16;
      f32x4 ah[kMBW], al[kMBW];
#pragma unroll
      for (int i = 0; i < kMBW; ++i) {
        ah[i] = *reinterpret_cast<const f32x4*>(cur + abase[i] + off);
        al[i] = *reinterpret_cast<const f32x4*>(cur + abase[i] + off + kPieces * 16);
      }
#pragma unroll
      for (int i = 0; i < kMBW; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(ah[i]), h8(bh[s]), acc[i], 0, 0, 0);
        accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(ah[i]), h8(bl[s]), accx[i], 0, 0, 0);
        accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(al[i]), h8(bh[s]), accx[i], 0, 0, 0);
      }
    }

    // staging tile free again (every wave is past the previous tile's store phase)
    __builtin_amdgcn_s_barrier();
    if (!STEM_ABLATE(8))
#pragma unroll
    for (int i = 0; i < kMBW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const float v = (acc[i][r] + accx[i][r]) * a.acc_scale + bias_n;
        stg[((kMBW * mp + i) * 32 + row) * kSRow + nb * 32 + (lane & 31)] = v;
      }
    // next tile's input has had the whole MFMA loop to land
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // U8: the next tile's bytes travel under the store phase and are converted after it
    if constexpr (U8)
      if (nhave) fetch(nimg, nta, ntb, ub);

    // ---- (a) raw conv1 rows for the level-0 masked pooling ----
    const int cr0 = 2 * kPoolR * ta - 1, cc0 = 2 * kPoolC * tb - 1;
    if (a.raw && !STEM_ABLATE(2)) {
      int y0 = 0, y1 = a.h1 - 1, x0 = 0, x1 = a.w1 - 1;
      if (a.bbox) {
        const int* bb = a.bbox + (long)img * 4;
        y0 = bb[0]; y1 = bb[1]; x0 = bb[2]; x1 = bb[3];
      }
      if (cr0 + kConvR > y0 && cr0 <= y1 && cc0 + kConvC > x0 && cc0 <= x1) {
        y1 = y1 < a.h1 - 1 ? y1 : a.h1 - 1;
        x1 = x1 < a.w1 - 1 ? x1 : a.w1 - 1;
#pragma unroll
        for (int j = 0; j < T::kRawIt; ++j) {
          const int pr = raw_rc[j] >> 8, pc = raw_rc[j] & 255;
          const int r = cr0 + pr, c = cc0 + pc;
          if (raw_rc[j] != 0xffffu && r >= y0 && r <= y1 && c >= x0 && c <= x1) {
            const int t = j * (NW * 4) + (tid >> 4);
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
      if (po < kPoolR * kPoolC && ho < a.hp && wo < a.wp && !STEM_ABLATE(2)) {
        // max_i relu(round(v_i * sc + sh)) == relu(round(sel_i(v_i) * sc + sh)) with sel =
        // max for sc >= 0, min for sc < 0: x -> round(x * sc + sh) is monotone, so the
        // window maximum can be taken BEFORE bn (on sg * v, sg = +-1 exact) -- two thirds
        // of the VALU work of the per-pixel form, same bits
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
              best[e] = fmaxf(best[e], u[e] * sg[e]);
              best[4 + e] = fmaxf(best[4 + e], w[e] * sg[4 + e]);
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
          best[e] = fmaxf(__builtin_fmaf(best[e] * sg[e], sc[e], sh[e]), 0.f);
        f32x4 hi4, lo4;
        stem_split8(best, &hi4, &lo4, &sat);
        float* d = a.y + ((((long)img * a.hp + ho) * a.wp + wo) * 8 + c8) * 8;
        *reinterpret_cast<f32x4*>(d) = hi4;
        *reinterpret_cast<f32x4*>(d + 4) = lo4;
      }
    }
    if constexpr (U8) {
      if (nhave) convert(ub, buf ^ 1);
      __syncthreads();  // (the DMA form publishes the next tile at the barrier after the staging)
    }
    have = nhave; img = nimg; ta = nta; tb = ntb;
  }
  report_saturation(a.status, sat);
}

bool stem_fused_supported(int cout, int Kp) { return cout == 64 && Kp == 224; }

template <int PR, int PC, int NW, bool U8>
static int launch_stem_cfg(StemArgs a, int cus, hipStream_t s) {
  using T = StemTile<PR, PC, NW>;
  a.tiles_y = (a.hp + PR - 1) / PR;
  a.tiles_x = (a.wp + PC - 1) / PC;
  auto kern = stem_fused_kernel<PR, PC, NW, U8>;
  const size_t lds = T::kLds + (U8 ? 772 * sizeof(unsigned) : 0);
  MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(kern), (int)lds));
  // persistent: one workgroup per CU (8 waves) or two (4 waves each)
  hipLaunchKernelGGL(kern, dim3(cus * (NW == 8 ? 1 : 2)), dim3(NW * 64), lds, s, a);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_stem_fused(const StemArgs& a, hipStream_t s) {
  MILAN_REQUIRE(a.in_u8 == nullptr || (a.W % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in_u8) & 3) == 0),
                MILAN_ERR_ARG, "stem: uint8 input needs W %% 4 == 0 and a 4-byte aligned pointer");
  MILAN_REQUIRE(a.n > 0 && a.H > 0 && a.G > 0 && (a.in || (a.in_u8 && a.lut && a.W > 0)) && a.ws &&
                    a.y && a.scale && a.shift && a.zero,
                MILAN_ERR_ARG, "stem: missing operand");
  int cus = 0;
  MILAN_TRY(device_cus8(&cus));
  StemArgs aa = a;
  if (aa.status == nullptr) aa.status = status_word();
  // algorithmic work: 7x7x3 taps per conv1 output; bytes: input groups once, raw
  // fp32 out, pooled split out
  const double px1 = (double)a.n * a.h1 * a.w1, pxp = (double)a.n * a.hp * a.wp;
  void* rec = gemm_profile_begin(
      2.0 * px1 * 64 * 147,
      (a.in_u8 ? 3.0 * a.n * a.H * a.W : 32.0 * a.n * a.H * a.G) + (a.raw ? 256.0 * px1 : 0.0) +
          256.0 * pxp, s);
  profile_tag_kernel(MILAN_KERNEL_STEM);
  int r;
#if MILAN_EXPERIMENTS
  // MILAN_STEM_TILE=0: 3 x 8 pooled pixels per 4-wave workgroup, two per CU (measured
  // slower: 5.8 against 5.3 ms per 256 neurons, profiles/r3_experiments.txt)
  static const int variant = getenv("MILAN_STEM_TILE") ? atoi(getenv("MILAN_STEM_TILE")) : 1;
  if (variant == 0 && !aa.in_u8) r = launch_stem_cfg<3, 8, 4, false>(aa, cus, s);
  else
#endif
  r = aa.in_u8 ? launch_stem_cfg<7, 8, 8, true>(aa, cus, s) : launch_stem_cfg<7, 8, 8, false>(aa, cus, s);
  gemm_profile_end(rec, s);
  return r;
}

}  // namespace milan
