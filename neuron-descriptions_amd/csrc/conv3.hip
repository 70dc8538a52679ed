// 3x3 / stride 1 / pad 1 convolution with 64 -> 64 channels (layer1's conv2 of every
// torchvision Bottleneck, reference call site src/milan/encoders.py:298) for the
// split-f16 mode, as a persistent kernel with the weights resident in registers.
//
// The implicit-GEMM kernel (gemm.hip) gathers every input pixel nine times through
// the L2 -> LDS DMA path (once per tap, 64-byte pieces) and streams the 147 KB of
// weights through LDS for every 256 x 64 tile: at 56 x 56 x 64 the layer is bound by
// that traffic, not by the matrix cores or HBM.  Here
//   * a workgroup owns an 8 x 14 tile of output pixels; the 10 x 16 input pixels
//     behind it are staged in LDS ONCE (global_load_lds, whole 256-byte pixels, a
//     three-deep ring of tiles) and the A fragments of all nine taps are read
//     straight from that tile (16-byte slots of a pixel XOR-swizzled by 14 row + column:
//     lane l of a fragment read is output pixel 14 pr + pc = 32 block + l, so for every
//     tap the sixteen lanes of a ds_read_b128 group carry sixteen different keys --
//     round 5; keyed by the column alone two lanes of every group met in a slot, 41 % of
//     the kernel's LDS cycles);
//   * the 64 x 576 weights never touch LDS: wave (nb, kh) keeps the B fragments of
//     output channels 32 nb .. 32 nb + 31 for HALF of K (18 of the 36 k-slabs, 144
//     registers) for the whole kernel;
//   * the accumulation order of the implicit-GEMM kernel (one accumulator, k
//     ascending, hl / lh / hh per slab) is kept by handing the accumulator from the
//     first-half wave to the second-half wave through LDS: in step s the first-half
//     waves multiply block s while the second-half waves first store block s - 2
//     (scale, + bias, ReLU, split: VALU / LDS / HBM work under the other waves' MFMAs)
//     and then finish block s - 1.  One s_barrier per step.
// Results are bitwise those of launch_gemm on the same layer (tests/test_gpu_conv3.py).
#include "common.h"

namespace milan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// timing ablations (Conv3Args::debug) exist in the experiments build only
#if MILAN_EXPERIMENTS
#define C3_ABLATE(bit) (a.debug & (bit))
#else
#define C3_ABLATE(bit) false
#endif

namespace {

constexpr int kTR = 8, kTC = 14;           // output pixels per tile (112 of 128 MFMA rows)
constexpr int kIR = kTR + 2, kIC = kTC + 2;  // input tile 10 x 16 pixels
constexpr int kInBytes = kIR * kIC * 256;  // 64 channels x (hi, lo) f16 = 256 B per pixel
constexpr int kRing = 3;                   // input tiles in LDS
constexpr int kBatch = 5;                  // DMA instructions per first-half wave and step
constexpr int kHalfSlabs = 18;             // k-slabs (16 slots) per K half; K = 576 = 36 slabs
constexpr int kHandBytes = 4096;           // one accumulator: 16 registers x 64 lanes
constexpr size_t kLds = (size_t)kRing * kInBytes + 4 * 2 * kHandBytes + 4 * 2048;  // = 160 KB
static_assert(kIR * kIC * 16 == 4 * 2 * kBatch * 64, "two batches of the four loader waves = one tile");

__device__ inline f16x8 h8(f32x4 v) { return __builtin_bit_cast(f16x8, v); }

__device__ inline void c3_split8(const float* v, f32x4* hi_out, f32x4* lo_out, float* sat) {
  split8_rne(v, hi_out, lo_out, sat);  // common.h
}

// 18 k-slabs of one 32-pixel block: slab sg = 18 KH + s covers k-groups 2 sg (+1 for
// the upper half-wave) = tap sg / 4, channel groups (2 sg) % 8 + half
template <int KH>
__device__ __forceinline__ void c3_mfma(const char* buf, int pbase, int pc, int half,
                                        const f32x4 (&bh)[kHalfSlabs],
                                        const f32x4 (&bl)[kHalfSlabs], f32x16& acc) {
  // Fragments of slab s + 1 are fetched before the MFMAs of slab s.  Hand-placed LDS
  // reads and waits (the compiler drains lgkmcnt to 0 in front of every MFMA group,
  // which would put the latency of the prefetch back on the critical path); the wait
  // names the fragment registers, which keeps the MFMAs behind it.
  f32x4 ah[3], al[3];
  auto fetch = [&](int s, int b3) {
    const int sg = KH * kHalfSlabs + s;
    const int tap = sg >> 2, kh = tap / 3, kw = tap - kh * 3;
    const int jh = ((4 * sg) & 15) + 2 * half;  // 16-byte slot of the hi piece
    const int x = pc + kTC * kh + kw;           // swizzle key of input pixel (pr + kh, pc + kw)
    const char* p = buf + pbase + (kh * kIC + kw) * 256;
    asm volatile("ds_read_b128 %0, %1"
                 : "=v"(ah[b3]) : "v"((LDS_AS const char*)(p + (((jh ^ x) & 15) << 4))) : "memory");
    asm volatile("ds_read_b128 %0, %1"
                 : "=v"(al[b3]) : "v"((LDS_AS const char*)(p + ((((jh + 1) ^ x) & 15) << 4))) : "memory");
  };
  // two slabs ahead: a wave alone on its SIMD (its partner is in the epilogue) must
  // cover the whole LDS latency by itself
  fetch(0, 0);
  fetch(1, 1);
#pragma unroll
  for (int s = 0; s < kHalfSlabs; ++s) {
    const int b3 = s % 3;
    if (s + 2 < kHalfSlabs) {
      fetch(s + 2, (s + 2) % 3);
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[b3]), "+v"(al[b3]) :: "memory");
    } else if (s + 1 < kHalfSlabs) {
      asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[b3]), "+v"(al[b3]) :: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[b3]), "+v"(al[b3]) :: "memory");
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(ah[b3]), h8(bl[s]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(al[b3]), h8(bh[s]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(ah[b3]), h8(bh[s]), acc, 0, 0, 0);
  }
}

}  // namespace

__global__ __launch_bounds__(512, 1) void conv3_p64_kernel(Conv3Args a) {
  extern __shared__ __attribute__((aligned(16))) char c3_smem[];
  char* hand = c3_smem + kRing * kInBytes;  // [pair 0..3][parity][kHandBytes]
  float sat = 0.f;  // (common.h: saturation of the split clamp is loud)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh2 = wave >> 2;                 // 0: k-slabs 0..17, 1: k-slabs 18..35
  const int pair = wave & 3;                 // (mg, nb): the two K halves of a pair share a SIMD
  const int nb = pair & 1, mg = pair >> 1;   // channel half, pixel blocks 2 mg, 2 mg + 1
  const int half = lane >> 5;

  // ---- weights: B fragments of channels 32 nb .. +31, this wave's 18 slabs ----
  f32x4 bh[kHalfSlabs], bl[kHalfSlabs];
  {
    const float* wrow = a.ws + (long)(nb * 32 + (lane & 31)) * 576 + kh2 * (kHalfSlabs * 16) + half * 8;
#pragma unroll
    for (int s = 0; s < kHalfSlabs; ++s) {
      bh[s] = *reinterpret_cast<const f32x4*>(wrow + s * 16);
      bl[s] = *reinterpret_cast<const f32x4*>(wrow + s * 16 + 4);
    }
  }

  // ---- A fragment addresses of the pair's two pixel blocks ----
  int pbase[2], pcol[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    int t = (mg * 2 + b) * 32 + (lane & 31);
    t = t < kTR * kTC ? t : kTR * kTC - 1;
    const int pr = t / kTC, pc = t - pr * kTC;
    pbase[b] = (pr * kIC + pc) * 256;
    pcol[b] = kTC * pr + pc;  // (c3_mfma adds the tap's 14 kh + kw: key = 14 row + column)
  }

  // ---- tile sequence (XCD x walks images x, x + 8, ...; see stem.hip) ----
  a.n = live_rows(a.n_live, 1, a.n);   // (image count on the device, GemmArgs::m_live)
  const int tiles = a.tiles_y * a.tiles_x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  auto decode = [&](int i, int* img, int* ty, int* tx) -> bool {
    const int q = i * nslots + slot;
    const int gidx = q / tiles, ti = q - gidx * tiles;
    *img = gidx * 8 + xcd;
    *ty = ti / a.tiles_x;
    *tx = ti - *ty * a.tiles_x;
    return *img < a.n;
  };
  // number of tiles of this workgroup
  int ntile = 0;
  {
    const int per_xcd_imgs = (a.n - xcd + 7) / 8;          // images of this XCD
    const long total = (long)(per_xcd_imgs > 0 ? per_xcd_imgs : 0) * tiles;
    ntile = total > slot ? (int)((total - slot + nslots - 1) / nslots) : 0;
  }

  // DMA batch g = tile g / 2, rows 5 (g % 2) .. + 4 of the 10-row input tile: loader wave
  // `pair`, instruction k moves row 5 (g % 2) + k, columns 4 pair .. 4 pair + 3
  // (4 pixels x 16 slots); LDS slot jj of pixel (r, c) holds memory piece jj ^ (14 r + c)
  const int ld_c = 4 * pair + (lane >> 4);
  auto issue_batch = [&](int g) -> bool {
    const int i = g >> 1;
    int img, ty, tx;
    if (i >= ntile || !decode(i, &img, &ty, &tx) || C3_ABLATE(4)) return false;
    char* dst = c3_smem + (i % kRing) * kInBytes;
    const int ix = tx * kTC - 1 + ld_c;
    const bool xok = ix >= 0 && ix < a.w;
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int r = (g & 1) * kBatch + k;
      const int iy = ty * kTR - 1 + r;
      const bool ok = xok && iy >= 0 && iy < a.h;
      const int ld_j = ((lane ^ (kTC * r + ld_c)) & 15) * 4;  // floats
      const float* src = ok ? a.in + (((long)img * a.h + iy) * a.w + ix) * 64 + ld_j : a.zero;
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)src,
                                       (LDS_AS void*)(dst + (r * kIC + 4 * pair) * 256), 16,
                                       0, 0);
    }
    return true;
  };

  // epilogue constants: lane -> (row it * 16 + lane / 4, channels 8 (lane % 4) ..)
  const int e_g = lane & 3;
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = a.bias ? a.bias[nb * 32 + e_g * 8 + e] : 0.f;

  if (kh2 == 0) {
    issue_batch(0); issue_batch(1);
    if (issue_batch(2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kBatch) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  // step s: first-half waves multiply block s and hand it over; second-half waves store
  // block s - 2 (epilogue, under the first-half waves' MFMAs) and then finish block s - 1
  const int nblk = 2 * ntile;
  const int nsteps = nblk + 2;
#if MILAN_EXPERIMENTS
  long long pt[4] = {0, 0, 0, 0}, t0 = clock64();
#define C3_STAMP(k) do { const long long t1 = clock64(); pt[k] += t1 - t0; t0 = t1; } while (0)
#else
#define C3_STAMP(k) do {} while (0)
#endif
  if (kh2 == 0) {
    for (int s = 0; s < nsteps; ++s) {
      if (s < nblk) {
        const int ti = s >> 1, b = s & 1;
        const char* buf = c3_smem + (ti % kRing) * kInBytes;
        char* hslot = hand + (pair * 2 + b) * kHandBytes;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (!C3_ABLATE(1)) c3_mfma<0>(buf, pbase[b], pcol[b], half, bh, bl, acc);
        if (!C3_ABLATE(8))
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *reinterpret_cast<f32x4*>(hslot + q * 1024 + lane * 16) = v;
          }
      }
      C3_STAMP(0);
      // every batch but the one issued in this step has landed: the tile the next step
      // starts is whole
      if (issue_batch(s + 3)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kBatch) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      C3_STAMP(1);
      __builtin_amdgcn_s_barrier();
      C3_STAMP(2);
    }
  } else {
    float* stg = reinterpret_cast<float*>(hand + 4 * 2 * kHandBytes + pair * 2048);  // [16][32]
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = 0; s < nsteps; ++s) {
      if (s >= 2 && s - 2 < nblk && !C3_ABLATE(2)) {
        // ---- epilogue of block s - 2: scale, + bias, ReLU, split, store ----
        const int ti = (s - 2) >> 1, b = (s - 2) & 1;
        int img, ty, tx;
        decode(ti, &img, &ty, &tx);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int lrow = (r & 3) + 8 * (r >> 2) + 4 * half;  // rows 16 it .. 16 it + 15
            stg[lrow * 32 + (lane & 31)] = acc[8 * it + r] * a.acc_scale;
          }
          const int row = it * 16 + (lane >> 2);
          const int t = (mg * 2 + b) * 32 + row;
          const int pr = t / kTC, pc = t - pr * kTC;
          const int oy = ty * kTR + pr, ox = tx * kTC + pc;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + (lane >> 2) * 32 + e_g * 8);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + (lane >> 2) * 32 + e_g * 8 + 4);
          if (t < kTR * kTC && oy < a.h && ox < a.w) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = fmaxf(v0[e] + bias8[e], 0.f);
              v[4 + e] = fmaxf(v1[e] + bias8[4 + e], 0.f);
            }
            f32x4 hi, lo;
            c3_split8(v, &hi, &lo, &sat);
            float* d = a.out + (((long)img * a.h + oy) * a.w + ox) * 64 + nb * 32 + e_g * 8;
            *reinterpret_cast<f32x4*>(d) = hi;
            *reinterpret_cast<f32x4*>(d + 4) = lo;
          }
        }
      }
      C3_STAMP(0);
      if (s >= 1 && s - 1 < nblk) {
        const int ti = (s - 1) >> 1, b = (s - 1) & 1;
        const char* buf = c3_smem + (ti % kRing) * kInBytes;
        const char* hslot = hand + (pair * 2 + b) * kHandBytes;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (!C3_ABLATE(8))
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(hslot + q * 1024 + lane * 16);
            acc[4 * q] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
          }
        if (!C3_ABLATE(1)) c3_mfma<1>(buf, pbase[b], pcol[b], half, bh, bl, acc);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      C3_STAMP(1);
      __builtin_amdgcn_s_barrier();
      C3_STAMP(2);
    }
  }
#if MILAN_EXPERIMENTS
  if (a.prof && blockIdx.x == 0 && lane == 0 && pair == 0)
    for (int k = 0; k < 4; ++k) a.prof[kh2 * 4 + k] = pt[k];
#endif
  report_saturation(a.status, sat);
}

bool conv3_p64_supported(int cin, int cout, int kh, int kw, int stride, int pad) {
  return cin == 64 && cout == 64 && kh == 3 && kw == 3 && stride == 1 && pad == 1;
}

int launch_conv3_p64(const Conv3Args& a0, hipStream_t s) {
  Conv3Args a = a0;
  MILAN_REQUIRE(a.n > 0 && a.h > 0 && a.w > 0 && a.in && a.ws && a.out && a.zero,
                MILAN_ERR_ARG, "conv3: missing operand");
  a.tiles_y = (a.h + kTR - 1) / kTR;
  a.tiles_x = (a.w + kTC - 1) / kTC;
  if (a.status == nullptr) a.status = status_word();
  int cus = 0;
  MILAN_TRY(device_cus8(&cus));
  MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(conv3_p64_kernel), (int)kLds));
  const double px = (double)a.n * a.h * a.w;
  void* rec = gemm_profile_begin(2.0 * px * 64 * 576, 4.0 * (px * 64 * 2 + 64 * 576), s);
  profile_tag_kernel(MILAN_KERNEL_CONV3);
  hipLaunchKernelGGL(conv3_p64_kernel, dim3(cus), dim3(512), kLds, s, a);
  gemm_profile_end(rec, s);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace milan
