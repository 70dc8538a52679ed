// Fused expand -> reduce chain of two 1x1 convolutions on the gfx950 matrix cores
// (split-f16 mode only).
//
// torchvision's Bottleneck (reference call site src/milan/encoders.py:298) ends
// with   x' = relu(bn3(conv3(t2)) + x)              (1x1, P -> 4P channels)
// and the next block starts with
//        t1' = relu(bn1(conv1(x')))                 (1x1, 4P -> P channels).
// As two launches the 4P-channel tensor x' is written by the first and read back
// by the second -- in layer3 that read is a quarter of the block's HBM traffic.
// Here ONE launch does both: the expand output never leaves the chip on its way
// into the reduce convolution (it is still written once, for the residual path
// and the pyramid tap).
//
// Layout of the computation ("pixel per lane"): a wave owns 32 pixels for the
// whole kernel and every MFMA is issued TRANSPOSED -- weights as the A operand
// (rows = output channels), pixels as the B operand (columns = pixels) -- so that
// an accumulator lane holds output channels of ONE pixel:
//     D[n][pixel]: lane = (pixel & 31, half), register r <-> channel
//     (r & 3) + 8 (r >> 2) + 4 half.
// A and B fragments have the same register format (lane & 31 = row / column,
// lane >> 5 = which 8 of the 16 k), so the expand result of a pixel, converted to
// (hi, lo) f16 pairs, is a B fragment of the reduce product for the same lane set
// after one trip through the wave's private LDS strip (which the epilogue needs
// anyway to turn channel-per-register into the row-contiguous 16-byte accesses
// HBM wants).  Per wave:
//   t2 fragments of its 32 pixels       P / 2 registers, loaded once
//   reduce accumulators 32 px x P       P / 2 registers, live for the whole kernel
//   expand accumulators, one 64-channel slab at a time (32 registers)
// (P = 256 -- layer3 -- would need ~400 registers per wave: it runs on the role ping-pong
// of chain3.hip instead.)  Weights stream HBM/L2 -> LDS by global_load_lds_dwordx4 in
// 16 KB tiles (64 weight rows x 64 k) through a 4-deep ring shared by the waves;
// a tile is 24 MFMAs per wave.  Per 64-channel slab j of the expand output:
//   S1  expand:  acc3 = W3[slab j] . t2          (P/64 tiles)
//   S2  epilogue: scale, + bias, + residual, ReLU, split -> HBM (x') and -> LDS
//   S3  reduce:  acc1 += W1[:, slab j] . x'[slab j]   (P/64 tiles)
// Accumulation order over k and the (hl, lh, hh) order of the three f16 products
// are those of igemm_split16_kernel, and the epilogue arithmetic is the same
// sequence of roundings, so x' and t1' are bitwise what the two separate launches
// produce (tests/test_gpu_chain.py).
#include "common.h"

#include <cstdlib>

namespace milan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

namespace {

__device__ inline f16x8 h8(f32x4 v) { return __builtin_bit_cast(f16x8, v); }

// 8 fp32 <-> (hi, lo) f16x8 pair, hi saturating: common.h
__device__ inline void chain_split8(const float* v, f32x4* hi_out, f32x4* lo_out) {
  split8_rne(v, hi_out, lo_out);
}
__device__ inline void chain_join8(f32x4 hi, f32x4 lo, float* v) { join8_exact(hi, lo, v); }

constexpr int kSRow = 68;        // floats per row of a wave's LDS strip (64 + 4)
constexpr int kTileFloats = 64 * 64;  // one weight tile: 64 rows x 64 k (16 KB)
// chain_kernel<.., CONV>: pixels of the 3x3 conv's input region in LDS -- 256 output pixels
// + (w + 1) before and after, w <= 56, rounded up to whole 4-pixel DMA pieces
constexpr int kConvMaxW = 56;
constexpr int kConvSlots = (256 + 2 * (kConvMaxW + 1) + 3) / 4 * 4;   // 372

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace

// RES_LDS: the residual slab and the bias reach the epilogue through LDS (DMA) instead
// of registers; needs 8 KB more LDS per wave, so only the 4-wave configuration has it.
// KD: channels of the second expand source (0 = none; then there is a residual).
// XACC: the reduce product keeps its cross terms (hl + lh) in a second accumulator,
// like igemm_kernel<SPLIT> does for N <= 64 layers -- same bits as that kernel.
// NR: output channels of the reduce conv: P inside a stage, 2 P when the next block opens
// the next stage (its conv1 still runs at this stage's resolution).
//
// CONV (round 6; planes 64, 8 waves): the bottleneck's 3x3 convolution c2 runs IN FRONT of the
// chain, in the same pixel-per-lane form -- W2 as the A operand (one 16 KB tile of the weight
// stream per tap, nine tiles ahead of the W3 / W1 tiles, same ring, same waits), the pixels
// of c2's input as the B operand, read from an LDS copy of the 256 + 2 (w + 1) input pixels
// behind the workgroup's 256 consecutive output pixels (pixel p of the region at 256 p bytes,
// its 16-byte chunks XOR-swizzled by p & 15; taps outside the image read a zero line).  The
// accumulator of a pixel's 64 output channels then sits in the lane that owns the pixel; one
// v_permlane32_swap per register pair turns it into whole 8-channel groups = the (hi, lo) B
// fragments the expand product needs (after scale, + bias, ReLU, split): t2 is never
// written and never read -- 512 bytes per pixel less through HBM, and the 3x3's MFMA work
// runs inside a kernel whose matrix pipe is otherwise 70 % idle (conv3_p64 by itself is bound
// by its hand-over / epilogue chain, not by HBM).  The region aliases the strips (used only
// after it), LDS = 64 KB ring + 93 KB.  k order (tap, channel), (hl, lh, hh) per k-step and the
// epilogue's roundings are conv3_p64's / the implicit GEMM's: same bits.
//
// C1 (with CONV; layer1.0, 64-channel block input): the block's own 1x1 reduce conv runs in
// front of the 3x3 -- the region is DMA'd from the block INPUT, one more 16 KB tile (W0) heads
// the weight stream, and before the first tap every wave turns one or two 32-pixel blocks of
// the region into t1 = relu(x W0^T * scale0 + bias0) IN PLACE (fragments out of the region,
// 24 MFMAs, lane swap, split, eight 16-byte writes back into the pixel's own slots; with the
// separate cross-term accumulator of igemm_kernel<SPLIT>, the kernel that runs this layer
// unfused -- same bits).  Halo pixels are computed by every workgroup that needs them (1.45 x
// of a 64 x 64 product).  The whole bottleneck in one launch: t1 and t2 stay on chip.
template <int P, int NW, bool RES_LDS, int KD, bool XACC, bool PROF = false, int NR = P,
          bool CONV = false, bool C1 = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void chain_kernel(ChainArgs g) {
  long long tprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = 0;
  auto stamp = [&](int k) {
    if constexpr (PROF) {
      const long long t = __builtin_readcyclecounter();
      tprof[k] += t - tlast;
      tlast = t;
    }
  };
  if constexpr (PROF) tlast = __builtin_readcyclecounter();
  constexpr int N3 = 4 * P, N1 = NR;
  constexpr int K3 = P + KD;        // K of the expand product
  constexpr int KS3 = K3 / 16;      // its k-steps (16 channels)
  constexpr int NSLAB = N3 / 64;    // 64-channel slabs of the expand output
  constexpr int T3 = K3 / 64;       // W3 tiles per slab (64 k each)
  constexpr bool RES = KD == 0;     // residual epilogue (else bias + ReLU only)
  static_assert(!(RES_LDS && !RES) && K3 % 64 == 0, "chain: configuration");
  constexpr int T1 = N1 / 64;       // W1 tiles per slab (64 output rows each)
  constexpr int L = T3 + T1;        // weight tiles per slab
  constexpr int NT1 = N1 / 32;      // reduce-output MFMA tiles
  // Ring depth.  Every workgroup of an XCD walks the weight stream in step, so a tile
  // is an L2 miss for the first one and ~2 us away; what hides that is BYTES IN
  // FLIGHT.  The 4-wave configuration spends its LDS on a 7-slot ring (5 tiles = 80 KB
  // in flight); the residual slab then lands in the wave's strip itself (it is idle
  // between two epilogues).
  constexpr int STAGES = RES_LDS ? 7 : 4, AHEAD = STAGES - 1;
  constexpr int INFL = AHEAD - 2;   // tiles still in flight behind a complete tile q + 2
  constexpr int PIECES = 16 / NW;   // 1 KB DMA pieces per wave per tile
  // VMEM ops of one epilogue that the ring waits may count on being outstanding: its 8
  // residual DMA pieces, or its 8 residual loads + 2 bias loads.  The 8 stores are issued
  // ahead of them and are NOT counted: a wait that allowed for them would be
  // too lax if a toolchain merged or split a store, or if stores retired out of order --
  // allowing for fewer operations than are in flight only waits a little longer.
  constexpr int C_OPS = RES_LDS ? 8 : (RES ? 10 : 2);
  static_assert(P % 64 == 0 && (NW == 4 || NW == 8), "chain: configuration");
  static_assert(!CONV || (P == 64 && NW == 8 && !RES_LDS), "chain: the conv front needs planes 64");
  static_assert(!C1 || CONV, "chain: c1 in front needs the conv front");
  constexpr int Q1 = C1 ? 1 : 0;       // weight tile of the block's own c1 at the very front
  constexpr int QC = CONV ? 9 + Q1 : 0;  // ... then the nine taps of the 3x3 conv
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 31, half = lane >> 5;
  float sat = 0.f;  // (common.h: saturation of the split clamp is loud)
  float* ring = smem;                                         // [STAGES][16 KB]
  float* strip = smem + STAGES * kTileFloats + wave * (32 * kSRow);
  // residual slab of the wave's 32 pixels (32 rows x 64 channels, DMA target: read
  // into registers at the start of the epilogue, before the accumulators overwrite
  // it) and the layer's expand bias
  float* rstrip = strip;  // 32 rows x 64 floats, unpadded, at the start of the strip
  float* biasl = smem + STAGES * kTileFloats + NW * (32 * kSRow);

  // (row count on the device: workgroups whose pixels lie beyond it exit, the one that
  // straddles it runs its ragged-tail path -- GemmArgs::m_live)
  g.M = live_rows(g.m_live, g.m_live_mul, g.M);
  if ((long)blockIdx.x * (NW * 32) >= g.M) return;
  const long m0 = (long)blockIdx.x * (NW * 32) + wave * 32;   // wave's first pixel
  const long mfrag = (m0 + px < g.M) ? m0 + px : (long)g.M - 1;
  const bool tail = (long)(blockIdx.x + 1) * (NW * 32) > g.M;  // workgroup-uniform

  // ---- weight-tile DMA ---------------------------------------------------------
  // tile q of the stream: slab j = q / L; qq = q % L < T3: W3 rows 64 j.., k 64 qq..;
  // else W1 rows 64 (qq - T3).., k 64 j...  LDS row rr holds logical 16-byte chunk c
  // at position c ^ (rr & 15) (conflict-free row-per-lane ds_read_b128).
  const int lrow = lane >> 4;                 // row within a 4-row piece
  const int lpos = lane & 15;
  auto issue_tile = [&](int q) {
    constexpr int NTILES = QC + NSLAB * L;
    q = q < NTILES ? q : NTILES - 1;          // dummy re-read past the end
    const float* base;
    long ld;
    if (C1 && q == 0) {
      base = g.W0; ld = 64;                   // c1: W0 rows 0..63, k 0..63
    } else if (CONV && q < QC) {
      base = g.W2 + 64 * (q - Q1); ld = 9 * 64;   // tap q - Q1: W2 rows 0..63, k 64 tap ..
    } else {
      const int qc = q - QC;
      const int j = qc / L, qq = qc - j * L;
      if (qq < T3) { base = g.W3 + (long)(64 * j) * K3 + 64 * qq; ld = K3; }
      else { base = g.W1 + (long)(64 * (qq - T3)) * N3 + 64 * j; ld = N3; }
    }
    float* slot = ring + (q % STAGES) * kTileFloats;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int pc = wave * PIECES + i;       // piece = rows 4 pc .. 4 pc + 3
      const int rr = 4 * pc + lrow;
      const int c = lpos ^ (rr & 15);
      __builtin_amdgcn_global_load_lds(
          (const GLOBAL_AS void*)(base + (long)rr * ld + c * 4),
          (LDS_AS void*)(slot + pc * 256), 16, 0, 0);
    }
  };

  // ---- prologue: t2 fragments, first residual / bias, three tiles in flight ------
  f32x4 t2h[KS3], t2l[KS3];
  // CONV: the input region of the 3x3 conv, pixels wg0 - (w + 1) .. wg0 + NW * 32 + w of the
  // flat (image, y, x) list, 256 bytes each (rows beyond either end of the batch repeat the
  // first / last row: only taps outside the image point there, and those read `czero`)
  float* const cin = smem + STAGES * kTileFloats;           // aliases the strips
  float* const czero = cin + kConvSlots * 64;               // 256 bytes of zeros
  f32x4 bias2v[CONV ? 8 : 1], bias0v[C1 ? 8 : 1];
  if constexpr (C1) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      bias0v[2 * s4] = *reinterpret_cast<const f32x4*>(g.bias0 + 16 * s4 + 8 * half);
      bias0v[2 * s4 + 1] = *reinterpret_cast<const f32x4*>(g.bias0 + 16 * s4 + 8 * half + 4);
    }
  }
  if constexpr (CONV) {
    const long wg0 = (long)blockIdx.x * (NW * 32);
    const int nslots = NW * 32 + 2 * (g.cw + 1);
    const int npieces = (nslots + 3) >> 2;
    for (int pc = wave; pc < npieces; pc += NW) {
      const int sl = 4 * pc + lrow;
      long m = wg0 - (g.cw + 1) + sl;
      m = m < 0 ? 0 : (m < g.M ? m : (long)g.M - 1);
      const int c = lpos ^ (sl & 15);
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(g.C2in + m * 64 + c * 4),
                                       (LDS_AS void*)(cin + pc * 256), 16, 0, 0);
    }
    if (tid < 16) *reinterpret_cast<f32x4*>(czero + tid * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias of the lane's channels after the lane swap: k-step s4 = 2 t + sigma, group half
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      bias2v[2 * s4] = *reinterpret_cast<const f32x4*>(g.bias2 + 16 * s4 + 8 * half);
      bias2v[2 * s4 + 1] = *reinterpret_cast<const f32x4*>(g.bias2 + 16 * s4 + 8 * half + 4);
    }
  }
  {
    if constexpr (!CONV) {
    const float* tp = g.T2 + mfrag * P + half * 8;
#pragma unroll
    for (int s = 0; s < P / 16; ++s) {
      t2h[s] = *reinterpret_cast<const f32x4*>(tp + s * 16);
      t2l[s] = *reinterpret_cast<const f32x4*>(tp + s * 16 + 4);
    }
    }
    if constexpr (KD > 0) {
      const float* ap = g.A2 + mfrag * KD + half * 8;
#pragma unroll
      for (int s = 0; s < KD / 16; ++s) {
        t2h[P / 16 + s] = *reinterpret_cast<const f32x4*>(ap + s * 16);
        t2l[P / 16 + s] = *reinterpret_cast<const f32x4*>(ap + s * 16 + 4);
      }
    }
  }
  // row-major epilogue roles: 8 lanes cover the 64 channels of a row, 8 rows per pass
  const int erow = lane >> 3, ecol = (lane & 7) * 8;
  // Residual slab j of the wave's pixels: 8 DMA pieces (4 rows x 256 B each) into the
  // wave's own LDS strip -- no registers, and no compiler-visible pending loads that
  // would make it drain the weight stream.  Row r holds logical chunk c at position
  // c ^ (r & 1) (conflict-free for the two-rows-per-group row-major reads).
  f32x4 res[RES_LDS ? 1 : 4][2], bias3v[2];
  auto load_res = [&](int j) {
    if constexpr (!RES_LDS) {
      const int n = 64 * j + ecol;
      bias3v[0] = *reinterpret_cast<const f32x4*>(g.bias3 + n);
      bias3v[1] = *reinterpret_cast<const f32x4*>(g.bias3 + n + 4);
      if constexpr (!RES) return;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        long m = m0 + it * 8 + erow;
        m = m < g.M ? m : (long)g.M - 1;
        const float* rp = g.R + m * N3 + n;
        res[it][0] = *reinterpret_cast<const f32x4*>(rp);
        res[it][1] = *reinterpret_cast<const f32x4*>(rp + 4);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 4 * i + lrow;
      long m = m0 + r;
      m = m < g.M ? m : (long)g.M - 1;
      const int c = lpos ^ (r & 1);
      __builtin_amdgcn_global_load_lds(
          (const GLOBAL_AS void*)(g.R + m * N3 + 64 * j + c * 4),
          (LDS_AS void*)(rstrip + i * 256), 16, 0, 0);
    }
  };
  if constexpr (RES_LDS) {
    for (int i = tid; i < N3; i += NW * 64) biasl[i] = g.bias3[i];
  }
  load_res(0);
#pragma unroll
  for (int t = 0; t < AHEAD; ++t) issue_tile(t);

  f32x16 acc1[NT1], acc1x[XACC ? NT1 : 1];
#pragma unroll
  for (int u = 0; u < NT1; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc1[u][r] = 0.f;
      if constexpr (XACC) acc1x[u][r] = 0.f;
    }

  // fragment of MFMA tile t (rows 32 t + px) of the weight tile in `slot`, chunk c
  const int fsw = px & 15;
  auto wfrag = [&](const float* slot, int t, int c) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(slot + (32 * t + px) * 64 + ((c ^ fsw) << 2));
  };

  // accumulators -> strip (channel-per-register to row-major): tile t of a 64-column
  // slab lands in columns 32 t .. 32 t + 31 of the wave's 32 x 64 strip
  auto to_strip = [&](const f32x16& a, int t) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      f32x4 v = {a[4 * qd], a[4 * qd + 1], a[4 * qd + 2], a[4 * qd + 3]};
      *reinterpret_cast<f32x4*>(strip + px * kSRow + 32 * t + 8 * qd + 4 * half) = v;
    }
  };

  // Ring protocol: at the top of iteration q tiles q and q + 1 are complete in LDS
  // (so the first fragments of tile q + 1 can be fetched BEFORE the barrier that ends
  // iteration q), tile q + 2 is in flight and tile q + 3 is issued into the slot tile
  // q - 1 has just left.
  wait_vmcnt<INFL * PIECES>();   // t2, residual, tiles 0 and 1 have landed
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // bias in LDS
  __builtin_amdgcn_s_barrier();
  stamp(0);
  // the t2 fragments have landed (wait above): tell the compiler, so that it does not
  // drain the weight DMA in front of their first use inside the loop
#pragma unroll
  for (int s = CONV ? P / 16 : 0; s < KS3; ++s)   // (CONV: t2 is computed below, A2 is loaded)
    asm volatile("" : "+v"(t2h[s]), "+v"(t2l[s]));
  if constexpr (CONV) {
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(bias2v[s]));
  }
  if constexpr (C1) {
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(bias0v[s]));
  }

  // weight fragments of one k-step (both 32-row MFMA tiles), double buffered: the
  // fragments of step s + 1 are fetched while step s multiplies.  With one wave per
  // SIMD (NW == 4) nothing else hides the LDS latency, and the compiler's own
  // lgkmcnt(0) in front of a step would also drain the reads just issued for the
  // next one -- so that configuration issues the reads and the counted wait itself
  // (the wait names the fragment registers, which keeps the MFMAs behind it).
  f32x4 wh[2][2], wl[2][2];
  auto load_w = [&](const float* slot, int s, int buf) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if constexpr (NW == 4) {
        const float* ph = slot + (32 * t + px) * 64 + (((4 * s + 2 * half) ^ fsw) << 2);
        const float* pl = slot + (32 * t + px) * 64 + (((4 * s + 2 * half + 1) ^ fsw) << 2);
        asm volatile("ds_read_b128 %0, %1" : "=v"(wh[buf][t]) : "v"((LDS_AS const float*)ph) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(wl[buf][t]) : "v"((LDS_AS const float*)pl) : "memory");
      } else {
        wh[buf][t] = wfrag(slot, t, 4 * s + 2 * half);
        wl[buf][t] = wfrag(slot, t, 4 * s + 2 * half + 1);
      }
    }
  };
  // fragments of buffer `buf` have landed; the four reads issued after them may fly
  auto wait_w = [&](int buf) {
    if constexpr (NW == 4)
      asm volatile("s_waitcnt lgkmcnt(4)"
                   : "+v"(wh[buf][0]), "+v"(wh[buf][1]), "+v"(wl[buf][0]), "+v"(wl[buf][1])
                   :: "memory");
  };
  load_w(ring, 0, 0);

  if constexpr (CONV) {
    // ================= 3x3 conv in front: nine weight tiles, one per tap =================
    const int hw = g.ch * g.cw;
    const int img = (int)(mfrag / hw);
    const int rem = (int)(mfrag - (long)img * hw);
    const int y = rem / g.cw, x = rem - y * g.cw;
    // the lane's pixel inside the region (tail lanes: the clamped pixel's, a valid duplicate)
    const int slot0 = (int)(mfrag - (long)blockIdx.x * (NW * 32)) + g.cw + 1;
    LDS_AS const char* const cin_b = (LDS_AS const char*)cin;
    LDS_AS const char* const czero_b = (LDS_AS const char*)czero;
    f32x16 acc2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
    // B fragments of one k-step (hi, lo), double buffered like the weight fragments
    f32x4 bxh[2], bxl[2];
    LDS_AS const char* tbase = czero_b;
    unsigned tkey = 0;
    auto set_tap = [&](int tap) {
      const int kh = tap / 3, kw = tap - 3 * kh;
      const int yy = y + kh - 1, xx = x + kw - 1;
      const bool ok = (unsigned)yy < (unsigned)g.ch && (unsigned)xx < (unsigned)g.cw;
      const int sl = slot0 + (kh - 1) * g.cw + (kw - 1);
      tkey = (unsigned)(sl & 15) << 4;
      tbase = ok ? cin_b + sl * 256 : czero_b;
    };
    auto load_b = [&](int s, int buf) {
      const unsigned ch = (unsigned)((4 * s) << 4) + ((unsigned)(2 * half) << 4);
      bxh[buf] = *reinterpret_cast<LDS_AS const f32x4*>(tbase + (ch ^ tkey));
      bxl[buf] = *reinterpret_cast<LDS_AS const f32x4*>(tbase + ((ch + 16u) ^ tkey));
    };
    if constexpr (C1) {
      // ============ the block's own c1 over the whole region, in place (tile 0 = W0) ============
      issue_tile(AHEAD);
      const int nslots = NW * 32 + 2 * (g.cw + 1);
      const int nblk = (nslots + 31) >> 5;
      const float* w0 = ring;   // (slot 0 of the ring)
      for (int blk = wave; blk < nblk; blk += NW) {
        int sl = 32 * blk + px;
        const bool own = sl < nslots;           // (the last block reaches beyond the region:
        sl = own ? sl : nslots - 1;             //  those lanes repeat its last pixel, unstored)
        LDS_AS char* const pb = (LDS_AS char*)cin + sl * 256;
        const unsigned key = (unsigned)(sl & 15) << 4;
        f32x4 xh[4], xl[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const unsigned ch = (unsigned)((4 * s + 2 * half) << 4);
          xh[s] = *reinterpret_cast<LDS_AS const f32x4*>(pb + (ch ^ key));
          xl[s] = *reinterpret_cast<LDS_AS const f32x4*>(pb + ((ch + 16u) ^ key));
        }
        // (one 32-channel tile at a time: two accumulators of 16 registers live, not four)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x16 a0, a0x;
#pragma unroll
          for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a0x[r] = 0.f; }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const f32x4 ah = wfrag(w0, t, 4 * s + 2 * half);
            const f32x4 al = wfrag(w0, t, 4 * s + 2 * half + 1);
            // (hh in one accumulator, hl + lh in the other: igemm_kernel<SPLIT>'s order)
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(ah), h8(xh[s]), a0, 0, 0, 0);
            a0x = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(al), h8(xh[s]), a0x, 0, 0, 0);
            a0x = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(ah), h8(xl[s]), a0x, 0, 0, 0);
          }
          a0 = a0 + a0x;
#pragma unroll
          for (int sg = 0; sg < 2; ++sg) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a0[8 * sg + i]),
                                                        __float_as_uint(a0[8 * sg + 4 + i]),
                                                        false, false);
              v[i] = __uint_as_float(r[0]);
              v[4 + i] = __uint_as_float(r[1]);
            }
            const int s4 = 2 * t + sg;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = __fmul_rn(v[e], g.scale0) + bias0v[2 * s4][e];
              v[4 + e] = __fmul_rn(v[4 + e], g.scale0) + bias0v[2 * s4 + 1][e];
            }
            f32x4 hi, lo;
            split8_relu_rne(v, &hi, &lo, &sat);
            // the lane's 8-channel group 2 s4 + half of its pixel: chunks 2 g, 2 g + 1
            const unsigned ch = (unsigned)((4 * s4 + 2 * half) << 4);
            if (own) {
              *reinterpret_cast<LDS_AS f32x4*>(pb + (ch ^ key)) = hi;
              *reinterpret_cast<LDS_AS f32x4*>(pb + ((ch + 16u) ^ key)) = lo;
            }
          }
        }
      }
      // tile 2 has landed (only the pieces of tile 3 were issued behind it); every wave's
      // region writes are done before the first tap reads them
      if (tail) wait_vmcnt<0>();
      else wait_vmcnt<INFL * PIECES>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      load_w(ring + (Q1 % STAGES) * kTileFloats, 0, 0);
    }
    set_tap(0);
    load_b(0, 0);
#pragma unroll
    for (int q = Q1; q < QC; ++q) {
      issue_tile(q + AHEAD);
      const float* slot = ring + (q % STAGES) * kTileFloats;
      const float* next_slot = ring + ((q + 1) % STAGES) * kTileFloats;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s < 3) {
          load_w(slot, s + 1, (s + 1) & 1);
          load_b(s + 1, (s + 1) & 1);
        } else {
          load_w(next_slot, 0, 0);
          if (q + 1 < QC) { set_tap(q + 1 - Q1); load_b(0, 0); }
        }
        const int b = s & 1;
        wait_w(b);
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wl[b][0]), h8(bxh[b]), acc2[0], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wl[b][1]), h8(bxh[b]), acc2[1], 0, 0, 0);
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][0]), h8(bxl[b]), acc2[0], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][1]), h8(bxl[b]), acc2[1], 0, 0, 0);
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][0]), h8(bxh[b]), acc2[0], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][1]), h8(bxh[b]), acc2[1], 0, 0, 0);
      }
      // tile q + 2 must have landed: only the pieces of tile q + 3 were issued behind it
      if (tail) wait_vmcnt<0>();
      else wait_vmcnt<INFL * PIECES>();
      __builtin_amdgcn_s_barrier();
    }
    // ---- t2 = relu(acc2 * scale2 + bias2) as the expand product's B fragments ------------
    // register r of tile t in lane (px, half) is channel 32 t + (r & 3) + 8 (r >> 2) + 4 half;
    // swapping the upper half-wave of registers 8 sigma + i with the lower half-wave of
    // registers 8 sigma + 4 + i leaves lane (px, half) with channels 32 t + 16 sigma + 8 half
    // + 0..7 in order: k-step 2 t + sigma of the lane's pixel
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int sg = 0; sg < 2; ++sg) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc2[t][8 * sg + i]),
                                                    __float_as_uint(acc2[t][8 * sg + 4 + i]),
                                                    false, false);
          v[i] = __uint_as_float(r[0]);
          v[4 + i] = __uint_as_float(r[1]);
        }
        const int s4 = 2 * t + sg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = __fmul_rn(v[e], g.scale2) + bias2v[2 * s4][e];
          v[4 + e] = __fmul_rn(v[4 + e], g.scale2) + bias2v[2 * s4 + 1][e];
        }
        split8_relu_rne(v, &t2h[s4], &t2l[s4], &sat);
      }
    stamp(7);   // (in-kernel profile: the whole 3x3 phase)
  }

  for (int j = 0; j < NSLAB; ++j) {
    f32x16 acc3[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[t][r] = 0.f;
    f32x4 xh[4], xl[4];
#pragma unroll
    for (int qq = 0; qq < L; ++qq) {
      const int q = QC + j * L + qq;
      issue_tile(q + AHEAD);
      stamp(1);
      const float* slot = ring + (q % STAGES) * kTileFloats;
      const float* next_slot = ring + ((q + 1) % STAGES) * kTileFloats;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        // (4 k-steps per tile: buffer parity is the same in every tile)
        if (s < 3) load_w(slot, s + 1, (s + 1) & 1);
        else load_w(next_slot, 0, 0);
        const int b = s & 1;
        wait_w(b);
        if (qq < T3) {
          // ---- S1: expand, k-step 4 qq + s ---------------------------------------
          const int ks = 4 * qq + s;
          acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wl[b][0]), h8(t2h[ks]), acc3[0], 0, 0, 0);
          acc3[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wl[b][1]), h8(t2h[ks]), acc3[1], 0, 0, 0);
          acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][0]), h8(t2l[ks]), acc3[0], 0, 0, 0);
          acc3[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][1]), h8(t2l[ks]), acc3[1], 0, 0, 0);
          acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][0]), h8(t2h[ks]), acc3[0], 0, 0, 0);
          acc3[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][1]), h8(t2h[ks]), acc3[1], 0, 0, 0);
        } else {
          // ---- S3: reduce, output rows 64 (qq - T3) .., k = slab j, step s ----------
          const int u0 = 2 * (qq - T3);
          if constexpr (XACC) {
            acc1[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][0]), h8(xh[s]), acc1[u0], 0, 0, 0);
            acc1[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][1]), h8(xh[s]), acc1[u0 + 1], 0, 0, 0);
            acc1x[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wl[b][0]), h8(xh[s]), acc1x[u0], 0, 0, 0);
            acc1x[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wl[b][1]), h8(xh[s]), acc1x[u0 + 1], 0, 0, 0);
            acc1x[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][0]), h8(xl[s]), acc1x[u0], 0, 0, 0);
            acc1x[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][1]), h8(xl[s]), acc1x[u0 + 1], 0, 0, 0);
          } else {
            acc1[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wl[b][0]), h8(xh[s]), acc1[u0], 0, 0, 0);
            acc1[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wl[b][1]), h8(xh[s]), acc1[u0 + 1], 0, 0, 0);
            acc1[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][0]), h8(xl[s]), acc1[u0], 0, 0, 0);
            acc1[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][1]), h8(xl[s]), acc1[u0 + 1], 0, 0, 0);
            acc1[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][0]), h8(xh[s]), acc1[u0], 0, 0, 0);
            acc1[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wh[b][1]), h8(xh[s]), acc1[u0 + 1], 0, 0, 0);
          }
        }

      }
      stamp(2);
      if (qq == T3 - 1) {
        // ---- S2: epilogue of slab j -------------------------------------------------
        const int n = 64 * j + ecol;
        f32x4 bias0, bias1, rh4[4], rl4[4];
        if constexpr (RES_LDS) {
          // residual slab j was issued L iterations ago: only the weight pieces issued
          // since then may still be in flight
          wait_vmcnt<(L * PIECES < 63 ? L * PIECES : 63)>();
          bias0 = *reinterpret_cast<const f32x4*>(biasl + n);
          bias1 = *reinterpret_cast<const f32x4*>(biasl + n + 4);
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + erow;
            const float* rp = rstrip + row * 64;
            const int rsw = row & 1;
            rh4[it] = *reinterpret_cast<const f32x4*>(rp + (((ecol >> 2)) ^ rsw) * 4);
            rl4[it] = *reinterpret_cast<const f32x4*>(rp + (((ecol >> 2) + 1) ^ rsw) * 4);
          }
        } else {
          bias0 = bias3v[0]; bias1 = bias3v[1];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc3[t] = acc3[t] * g.scale3;
          to_strip(acc3[t], t);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 8 + erow;
          const long m = m0 + row;
          float* sp = strip + row * kSRow + ecol;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(sp + 4);
          f32x4 rh, rl;
          if constexpr (RES_LDS) {
            rh = rh4[it]; rl = rl4[it];
          } else if constexpr (RES) {
            rh = res[it][0]; rl = res[it][1];
          }
          float v[8], a[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = v0[e] + bias0[e];
            v[4 + e] = v1[e] + bias1[e];
          }
          if constexpr (RES) {
            chain_join8(rh, rl, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] + a[e];
          }
          f32x4 hi, lo;
          split8_relu_rne(v, &hi, &lo, &sat);   // (ReLU folded into the clamp)
          if (m < g.M) {
            float* xp = g.X + m * N3 + n;
            *reinterpret_cast<f32x4*>(xp) = hi;
            *reinterpret_cast<f32x4*>(xp + 4) = lo;
          }
          *reinterpret_cast<f32x4*>(sp) = hi;      // x' in split form: a B fragment
          *reinterpret_cast<f32x4*>(sp + 4) = lo;  // source for the reduce product
        }
        if constexpr (!RES_LDS) load_res(j + 1 < NSLAB ? j + 1 : j);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float* fp = strip + px * kSRow + (2 * s + half) * 8;
          xh[s] = *reinterpret_cast<const f32x4*>(fp);
          xl[s] = *reinterpret_cast<const f32x4*>(fp + 4);
        }
        if constexpr (RES_LDS) {
          // the next residual slab lands in the strip: its last readers (the x'
          // fragment reads above) must have returned first
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]),
                         "+v"(xl[0]), "+v"(xl[1]), "+v"(xl[2]), "+v"(xl[3])
                       :: "memory");
          load_res(j + 1 < NSLAB ? j + 1 : j);
        }
      }
      stamp(3);
      // Tile q + 2 must have landed.  VMEM ops issued after its pieces: the pieces of
      // tiles q + 3 .. q + AHEAD, plus the epilogue ops of iterations q - INFL .. q.
      {
        constexpr auto has_c = [](int i) { return ((i % L) + L) % L == T3 - 1; };
        int c_later = 0, c_first = 0;
#pragma unroll
        for (int d = 0; d <= INFL; ++d) {
          c_later += has_c(qq - d) ? 1 : 0;
          c_first += (qq - d >= 0 && has_c(qq - d)) ? 1 : 0;  // first slab
        }
        const int cnt = (j == 0) ? c_first : c_later;
        constexpr int B = INFL * PIECES;
        if (tail) {
          // (a partly masked epilogue issues fewer stores than counted below)
          wait_vmcnt<0>();
        } else {
          switch (cnt) {
            case 0: wait_vmcnt<B>(); break;
            case 1: wait_vmcnt<(B + C_OPS < 63 ? B + C_OPS : 63)>(); break;
            case 2: wait_vmcnt<(B + 2 * C_OPS < 63 ? B + 2 * C_OPS : 63)>(); break;
            default: wait_vmcnt<(B + 3 * C_OPS < 63 ? B + 3 * C_OPS : 63)>(); break;
          }
        }
      }
      stamp(4);
      __builtin_amdgcn_s_barrier();
      stamp(5);
    }
  }
  wait_vmcnt<0>();  // drain the dummy tiles before the LDS can be re-allocated
  __builtin_amdgcn_s_barrier();

  // ---- reduce epilogue: t1' = relu(acc1 * scale + bias) in split form --------------
#pragma unroll
  for (int cidx = 0; cidx < NT1 / 2; ++cidx) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if constexpr (XACC) acc1[2 * cidx + t] = acc1[2 * cidx + t] + acc1x[2 * cidx + t];
      acc1[2 * cidx + t] = acc1[2 * cidx + t] * g.scale1;
      to_strip(acc1[2 * cidx + t], t);
    }
    const int n = 64 * cidx + ecol;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.bias1 + n);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(g.bias1 + n + 4);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + erow;
      const long m = m0 + row;
      const float* sp = strip + row * kSRow + ecol;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(sp + 4);
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = v0[e] + b0[e];
        v[4 + e] = v1[e] + b1[e];
      }
      f32x4 hi, lo;
      split8_relu_rne(v, &hi, &lo, &sat);   // (ReLU folded into the clamp)
      if (m < g.M) {
        float* tp = g.T1 + m * N1 + n;
        *reinterpret_cast<f32x4*>(tp) = hi;
        *reinterpret_cast<f32x4*>(tp + 4) = lo;
      }
    }
  }
  report_saturation(g.status, sat);
  if constexpr (PROF) {
    stamp(6);
    if (tid == 0 && g.prof)
      for (int k = 0; k < 8; ++k) g.prof[(long)blockIdx.x * 8 + k] = tprof[k];
  }
}

// Planes 256 (layer3) run on the role ping-pong of chain3.hip (round 5).  Its predecessors
// lived here and are gone: round 3's one-wave form (t2 fragments + reduce accumulators of 32
// pixels = ~400 registers, one wave per SIMD: 8.17 ms against 7.27 for the two launches) and a
// producer / consumer form (8.6 ms); round 4's lockstep two-wave form with a K-split expand
// and an accumulator hand-over (bitwise, 2.83 against 2.79 ms: parity, because the two waves
// of a SIMD multiplied one after the other and the epilogue ran with the pipe idle).  Numbers:
// DESIGN.md section 4.4, profiles/r3_experiments.txt, profiles/r4_experiments.txt E.

bool chain_conv_supported(int P, int KD, int NR, int h, int w) {
  return P == 64 && chain_supported(P, KD, NR) && h >= 1 && w >= 1 && w <= kConvMaxW;
}

bool chain_conv_c1_supported(int P, int KD, int NR, int h, int w) {
  return chain_conv_supported(P, KD, NR, h, w) && KD == 64 && NR == 64;   // layer1.0
}

bool chain_supported(int P, int KD, int NR) {
  if (NR == 2 * P) return KD == 0 && P == 64;  // stage boundary layer1 -> layer2
  if (NR != P) return false;
  if (KD == 0) return P == 64 || P == 128 || P == 256;
  return P == 64 && KD == 64;
}

template <int P, int NW, bool RES_LDS, int KD, bool XACC, bool PROF = false, int NR = P,
          bool CONV = false, bool C1 = false>
static int launch_chain_cfg(const ChainArgs& a, hipStream_t s) {
  size_t lds = sizeof(float) * (size_t)((RES_LDS ? 7 : 4) * kTileFloats +
                                        NW * 32 * kSRow + (RES_LDS ? 4 * P : 0));
  if (CONV) {   // the conv's input region (+ zero line) aliases the strips
    const size_t with_region = sizeof(float) * (size_t)(4 * kTileFloats + kConvSlots * 64 + 64);
    lds = lds > with_region ? lds : with_region;
  }
  auto kern = chain_kernel<P, NW, RES_LDS, KD, XACC, PROF, NR, CONV, C1>;
  MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(kern), (int)lds));
  const int rows = NW * 32;
  const int grid = (a.M + rows - 1) / rows;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, a);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_chain3(const ChainArgs& a, hipStream_t s);   // chain3.hip: the role ping-pong (round 5)

int launch_chain(const ChainArgs& a0, hipStream_t s) {
  ChainArgs a = a0;
  if (a.status == nullptr) a.status = status_word();
  const int NR = a.NR ? a.NR : a.P;
  MILAN_REQUIRE(chain_supported(a.P, a.KD, NR) && a.M > 0, MILAN_ERR_SHAPE,
                "chain: unsupported planes %d (+%d) -> %d", a.P, a.KD, NR);
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool conv = a.C2in != nullptr;
  MILAN_REQUIRE(!conv || (chain_conv_supported(a.P, a.KD, NR, a.ch, a.cw) && a.W2 && a.bias2 &&
                          al(a.C2in) && al(a.W2) && al(a.bias2) && a.C2in != a.T1 &&
                          a.M % (a.ch * a.cw) == 0),
                MILAN_ERR_SHAPE, "chain: unsupported 3x3 front (%d x %d images)", a.ch, a.cw);
  const bool c1 = conv && a.W0 != nullptr;
  MILAN_REQUIRE(!c1 || (chain_conv_c1_supported(a.P, a.KD, NR, a.ch, a.cw) && a.bias0 && al(a.W0) &&
                        al(a.bias0)),
                MILAN_ERR_SHAPE, "chain: unsupported c1 front");
  MILAN_REQUIRE((conv || al(a.T2)) && al(a.W3) && al(a.bias3) && al(a.X) && al(a.W1) &&
                    al(a.bias1) && al(a.T1) &&
                    (a.KD ? (a.A2 && al(a.A2) && !a.R) : (a.R && al(a.R))),
                MILAN_ERR_SHAPE, "chain: operands must be 16-byte aligned");
  const double M = a.M, P = a.P, K3 = a.P + a.KD, R1 = NR;
  // (conv front: + the 3x3's 2 M 64 576 flops; t2 is neither written nor read -- the operand
  // that crosses HBM is c2's input, M x 64)
  void* rec = gemm_profile_begin(
      2.0 * M * (4 * P) * (K3 + R1) + (conv ? 2.0 * M * 64 * 576 : 0.0) + (c1 ? 2.0 * M * 64 * 64 : 0.0),
      4.0 * (M * K3 + M * 4 * P * (a.KD ? 1 : 2) + M * R1 + 4 * P * (K3 + R1) +
             (conv ? 64.0 * 576 : 0.0)), s);
  profile_tag_kernel(a.P == 256 ? MILAN_KERNEL_CHAIN_WIDE
                                : (conv ? MILAN_KERNEL_BNECK : MILAN_KERNEL_CHAIN));
  int r;
#if MILAN_EXPERIMENTS
  if (conv && a.KD == 0 && NR == 64 && a.prof)   // in-kernel phase profile (tools/bench/bneckbench.hip)
    r = launch_chain_cfg<64, 8, false, 0, true, true, 64, true>(a, s);
  else
#endif
  if (conv && NR == 128) r = launch_chain_cfg<64, 8, false, 0, false, false, 128, true>(a, s);
  else if (conv && a.KD == 0) r = launch_chain_cfg<64, 8, false, 0, true, false, 64, true>(a, s);
  else if (c1) r = launch_chain_cfg<64, 8, false, 64, true, false, 64, true, true>(a, s);
  else if (conv) r = launch_chain_cfg<64, 8, false, 64, true, false, 64, true>(a, s);
  else if (a.P == 256) r = launch_chain3(a, s);
  else if (a.P == 128) r = launch_chain_cfg<128, 8, false, 0, false>(a, s);
  // the 128-channel reduce conv runs on the single-accumulator kernel when unfused
  else if (NR == 128) r = launch_chain_cfg<64, 8, false, 0, false, false, 128>(a, s);
  else if (a.KD == 0) r = launch_chain_cfg<64, 8, false, 0, true>(a, s);
  else r = launch_chain_cfg<64, 8, false, 64, true>(a, s);
  gemm_profile_end(rec, s);
  return r;
}

}  // namespace milan
