// Layer3's expand -> reduce chain (planes 256) as a ROLE ping-pong of the two waves of a SIMD
// (round 5; replaces the lockstep two-wave form of round 4, which only reached parity).
//
// torchvision's Bottleneck (reference call site src/milan/encoders.py:298) ends with
//     x' = relu(bn3(conv3(t2)) + x)        1x1, 256 -> 1024 channels
// and the next block starts with
//     t1' = relu(bn1(conv1(x')))           1x1, 1024 -> 256 channels.
// One launch does both for 128 pixels per workgroup: x' is written once (residual path,
// pyramid tap) and reaches the reduce product on chip -- 10 KB of HBM traffic per pixel
// instead of 14.
//
// What sank the earlier forms: every wave did the same thing at the same time,
// so the MFMA phases of the two waves of a SIMD ran one after the other and the epilogue
// (HBM round trips, LDS transposes) ran with the matrix pipe idle.  Here the two waves
// (p, 0) and (p, 1) of SIMD p share pixel block p (32 pixels) but have different JOBS, half
// a slab apart, so that in every slot exactly one of them multiplies:
//
//   wave E = (p, 0): holds the t2 fragments of its 32 pixels for the whole K = 256 (128
//            registers).  Slot 2s: E(s) = the 64-channel slab s of the expand product, 96
//            MFMAs, one accumulator per tile, k ascending -- the bits of the unfused kernel.
//            Slot 2s+1: epilogue of channels 0..31 of slab s (scale, + bias, + residual,
//            ReLU, split -> HBM and -> the block's LDS strip).
//   wave R = (p, 1): holds the reduce accumulators 32 px x 256 channels (128 registers).
//            Slot 2s+2: epilogue of channels 32..63 of slab s (the raw tile came through
//            the strip).  Slot 2s+3: R(s) = acc1 += W1[:, slab s] . x'[slab s], 96 MFMAs,
//            x' fragments from the strip.
//
//   slot      0      1       2       3       4       5    ...
//   E-wave  E(0)  epiA(0)  E(1)  epiA(1)  E(2)  epiA(2)
//   R-wave   --     --    epiB(0)  R(0)  epiB(1)  R(1)
//
// A slot is two half-slots of one weight-tile PAIR each (2 x 16 KB, 48 MFMAs per multiplying
// wave); one s_barrier per half-slot.  Weights stream L2 -> LDS through a ring of three
// pairs (96 KB); the pair of half-slot q + 2 is issued at the top of half-slot q by the four
// waves that are NOT multiplying (8 x 1 KB buffer_load ... lds each: no vector ALU, no
// registers).  The strips are double buffered by slab parity (2 x 4 x 8 KB): x'(s) is read
// by R(s) while epiA(s + 1) rewrites the other buffer.  LDS = 64 + 96 = all 160 KB.
// HBM traffic per slot and CU: 16 KB of residual loads (issued one slab ahead, 16 registers
// per wave) + 16 KB of x' stores against ~3100 cycles of MFMA = 10.3 B/clk -- the chip's HBM
// share: the kernel is balanced between the two roofs instead of serialising them.
//
// Arithmetic: k order, the (hl, lh, hh) order of the three f16 products and every rounding of
// the epilogue are those of igemm_split16_pp32_kernel + run_epilogue, so X and T1 are bitwise
// the two separate launches (tools/bench/chainbench.hip, tests/test_gpu_chain.py).
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace milan {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

namespace {

__device__ inline f16x8 h8(f32x4 v) { return __builtin_bit_cast(f16x8, v); }

// 16 bytes from (uniform base) + (per-lane byte offset).  A PLAIN load: the compiler sees it
// pending and orders every use -- and every copy or spill of its destination -- behind its own
// counted wait.  (An inline-asm load hides the pending state: the allocator may then move or
// spill the destination registers before the data has landed.  That passed every test on an
// idle GPU and corrupted a value now and then once a second process stretched the HBM latency:
// tests/test_gpu_concurrent.py, profiles/r5_experiments.txt.)
__device__ __forceinline__ f32x4 load16(const float* base, unsigned off) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off);
}
__device__ __forceinline__ float clamp_relu(float u) {  // min(max(u, 0), 65504)
  float r;
  asm("v_med3_f32 %0, %1, 0, %2" : "=v"(r) : "v"(u), "v"(65504.f));
  return r;
}
// 8 values already clamped to [0, 65504] -> (hi, lo) f16x8 pair (the roundings of split8_rne)
__device__ __forceinline__ void split8_clamped(const float* x, f32x4* hi_out, f32x4* lo_out) {
  f32x4 hi, lo;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    hi[d] = cvt_pk_f16(x[2 * d], x[2 * d + 1]);
    lo[d] = cvt_pk_f16(mix_sub_f16<0>(x[2 * d], hi[d]), mix_sub_f16<1>(x[2 * d + 1], hi[d]));
  }
  *hi_out = hi;
  *lo_out = lo;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int kTileFloats = 64 * 64;   // one weight tile: 64 rows x 64 k (16 KB)
constexpr int kStripFloats = 32 * 64;  // one pixel block's strip: 32 px x 64 channels (8 KB)

}  // namespace

template <bool PROF>
__global__ __launch_bounds__(512, 2) void chain3_kernel(ChainArgs g, int ntiles) {
  // in-kernel phase profile (ChainArgs::prof): cycles of wave 0 (E) / wave 4 (R) in
  // 0 prologue, 2 busy part of its MFMA half-slots, 4 busy part of its epilogue half-slots
  // (DMA issue, items, prefetch), 5 / 6 waiting at the barrier that ends an MFMA / epilogue
  // half-slot, 7 final epilogue.  Stamps sit only where the wave has drained its LDS
  // counter anyway (s_memtime returns through lgkmcnt), so they do not distort the phases.
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
  // (row count on the device, GemmArgs::m_live: the persistent walk covers the live tiles)
  if (g.m_live != nullptr) {
    g.M = live_rows(g.m_live, g.m_live_mul, g.M);
    ntiles = (g.M + 127) / 128;
    if ((int)blockIdx.x >= ntiles) return;
  }
  constexpr int P = 256, N3 = 1024, N1 = 256, NSLAB = 16;
  constexpr int NPAIR = 4 * (NSLAB + 1);  // weight-tile pairs = half-slots
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = wave & 3, h = wave >> 2;   // pixel block = SIMD; role 0 = E, 1 = R
  const int px = lane & 31, half = lane >> 5;
  float* strips = smem;                         // [2 buffers][4 blocks][8 KB]
  float* ring = smem + 2 * 4 * kStripFloats;    // [3 pairs][2 tiles][16 KB]
  float* strip0 = strips + p * kStripFloats;    // this block's strip, buffer 0 (+32 KB: buffer 1)
  float sat = 0.f;                              // (common.h: saturation of the split clamp is loud)

  // Persistent workgroups: tile = 128 pixels, tiles blockIdx.x, + gridDim.x, ...  The weight
  // stream, the residual prefetch and the t2 fragments of the NEXT tile are requested while
  // the current one finishes; `qbase` = pairs issued by earlier tiles (mod 3: the ring phase).
  int tile = blockIdx.x;
  int qbase = 0;

  // (Registers: 254 of 256.  The build keeps 4 scratch operations -- a two-register spill whose
  // store / reload sit on the persistent tile loop, once per tile, none inside the slab loop;
  // tools/check_isa.py pins them and fails on any scratch operation inside a further loop.
  // Every vmcnt(N) below is derived WITHOUT counting stores or scratch traffic: both can only
  // make a wait longer than needed, never shorter.)
  // ---- weight-pair DMA ----------------------------------------------------------------
  // A pair = two 16 KB tiles (64 weight rows x 64 k, 256-byte LDS rows, 16-byte chunk c of
  // row r at position c ^ (r & 15)).  A wave of the issuing group moves pieces i = 0..3 of
  // BOTH tiles: rows 16 i + 4 p + (lane >> 4) -- (row & 15) is the same for the four, so one
  // per-lane offset per matrix serves all of them and the rest of the address is scalar.
  // (the per-lane offset is recomputed from the lane number at every issue: five VALU
  // instructions instead of two registers held through the MFMA phases -- the kernel sits at
  // the 256-register limit, and a spill RELOAD is a scratch load = a full VMEM drain)
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
  auto voff_of = [&](int ld) {
    const int l = opaque(lane);
    const int wrow = 4 * p + (l >> 4), lpos = l & 15;
    return (unsigned)((wrow * ld + ((lpos ^ wrow) << 2)) * 4);
  };
  const __amdgpu_buffer_rsrc_t srd3 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.W3), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.W1), 0, 0x7fffffff, 0x00020000);
  // pair q: q >> 2 = step s, q & 3 = c; c < 2: W3 rows 64 s.., k tiles 2 c, 2 c + 1;
  // c >= 2: W1 rows 64 (2 (c - 2) + j).., k = slab s - 1.  Pairs nobody multiplies (R(-1),
  // E(16)) re-read a valid tile: constant VMEM counts per half-slot.
  auto issue_pair = [&](int q) {
    float* dst = ring + (((qbase + q) % 3) * 2) * kTileFloats + p * 256;
    q = q < NPAIR ? q : q - NPAIR;   // pairs 68, 69 = the next tile's 0, 1 (same weights)
    const int s = q >> 2, c = q & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (c < 2) {
        const unsigned voff3 = voff_of(P);
        const int sl = s < NSLAB ? s : NSLAB - 1;
        const int soff = ((64 * sl) * P + 64 * (2 * c + j)) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srd3, (LDS_AS void*)(dst + j * kTileFloats + i * 1024),
                                                   16, voff3, soff + i * (16 * P * 4), 0, 0);
      } else {
        const unsigned voff1 = voff_of(N3);
        const int sl = s < 1 ? 0 : s - 1;
        const int soff = ((64 * (2 * (c - 2) + j)) * N3 + 64 * sl) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(srd1, (LDS_AS void*)(dst + j * kTileFloats + i * 1024),
                                                   16, voff1, soff + i * (16 * N3 * 4), 0, 0);
      }
    }
  };

  // ---- epilogue roles: this wave's half = channels 32 h .. 32 h + 31 of a slab; a lane
  // takes one 8-channel group (32 bytes of split format) of rows erow and 16 + erow
  const int erow = lane >> 2, eg = lane & 3;
  const int G = 4 * h + eg;                      // group within the 64-channel slab
  // per-tile geometry: first pixel of the tile / of the block (scalars); a lane's epilogue
  // row `it` of tile t as a byte offset into X / R (tail: clamped to the last valid row,
  // stores masked) is recomputed where it is used
  long wg_m0 = 0, m0 = 0;
  float* xbase = g.X;
  auto row_off = [&](int t, int it) {
    const int l = opaque(lane);
    const long w0 = (long)t * 128;
    long r = p * 32 + 16 * it + (l >> 2);
    r = w0 + r < g.M ? r : (long)g.M - 1 - w0;
    return (unsigned)((r * N3 + 8 * (4 * h + (l & 3))) * 4);
  };
  auto set_geometry = [&](int t) {
    wg_m0 = (long)t * 128;
    m0 = wg_m0 + p * 32;
    xbase = g.X + wg_m0 * N3;
  };
  f32x4 res[2][2], bias3v[2];
  // bias of slab j (2 loads) and residual rows of slab j of tile t (4 loads from HBM), both
  // requested one slab ahead and held through the MFMA half-slots in between (requesting the
  // bias at the top of the half-slot that uses it cost its L2 round trip every slab)
  auto load_bias = [&](int j) {
    j = j < 0 ? 0 : (j < NSLAB ? j : NSLAB - 1);
    const unsigned off = (unsigned)((4 * h + (opaque(lane) & 3)) * 32);
    bias3v[0] = load16(g.bias3 + 64 * j, off);
    bias3v[1] = load16(g.bias3 + 64 * j + 4, off);
  };
  auto load_rows = [&](int t, int j) {
    j = j < NSLAB ? j : NSLAB - 1;
    const float* rb = g.R + (long)t * 128 * N3 + 64 * j;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const unsigned off = row_off(t, it);
      res[it][0] = load16(rb, off);
      res[it][1] = load16(rb + 4, off);
    }
  };
  auto res_landed = [&]() {
    asm volatile("" : "+v"(res[0][0]), "+v"(res[0][1]), "+v"(res[1][0]), "+v"(res[1][1]),
                      "+v"(bias3v[0]), "+v"(bias3v[1]));
  };

  // ---- LDS addresses (bytes), loop-invariant, no vector ALU in the MFMA phases -----------
  const int fsw = px & 15;
  auto lds_addr = [](const float* q) { return (unsigned)(uintptr_t)(LDS_AS const float*)q; };
  // fa[2 s4 + e]: this lane's (hi | lo = e) chunk of k-step s4 inside a 32 x 64 block with
  // 256-byte rows -- the x' fragments in the strip, and (+ tile base) the weight fragments
  unsigned fa[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    fa[k] = (unsigned)((px * 64 + (((4 * (k >> 1) + 2 * half + (k & 1)) ^ fsw) << 2)) * 4);
  const unsigned strip_b = lds_addr(strip0);
  const unsigned ring_b = lds_addr(ring);
  f32x4 wf[2][4];   // [buffer][hi t0, lo t0, hi t1, lo t1]
  unsigned wa[8];
  auto set_pair = [&](int q) {   // fragment addresses of tile 0 of pair q
    int soff = (int)ring_b + (((qbase + q) % 3) * 2) * (kTileFloats * 4);
    asm volatile("" : "+s"(soff));
#pragma unroll
    for (int k = 0; k < 8; ++k) wa[k] = fa[k] + (unsigned)soff;
  };
  // Weight fragments of step i = 0..7 of the current pair (tile i >> 2, k-step i & 3; wa[] points
  // at tile 0, tile 1 is +16 KB and the second 32-row MFMA tile +8 KB: immediates), the "lo"
  // and the "hi" pair separately.  A step multiplies lo . h first and hi . l, hi . h after it, so
  // the lo registers of a buffer are free after the step's second MFMA and the hi registers
  // after its sixth: each pair is re-requested for step i + 2 right there -- 1.3-1.7 steps of
  // look-ahead out of two buffers (requesting a whole step at the top of the previous one gave
  // 1.0: ~60 cycles of LDS stall per step, profiles/r5_experiments.txt A).
  auto rd_lo = [&](int buf, int i) {
    const int s4 = i & 3;
    if (i < 4) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(wf[buf][1]) : "v"(wa[2 * s4 + 1]) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(wf[buf][3]) : "v"(wa[2 * s4 + 1]) : "memory");
    } else {
      asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(wf[buf][1]) : "v"(wa[2 * s4 + 1]) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:24576" : "=v"(wf[buf][3]) : "v"(wa[2 * s4 + 1]) : "memory");
    }
  };
  auto rd_hi = [&](int buf, int i) {
    const int s4 = i & 3;
    if (i < 4) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(wf[buf][0]) : "v"(wa[2 * s4]) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(wf[buf][2]) : "v"(wa[2 * s4]) : "memory");
    } else {
      asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(wf[buf][0]) : "v"(wa[2 * s4]) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:24576" : "=v"(wf[buf][2]) : "v"(wa[2 * s4]) : "memory");
    }
  };
  // Requests are issued in the order lo(0) hi(0) lo(1) hi(1) | lo(2) hi(2) lo(3) ... (LDS returns in
  // order): before step i's first MFMA everything up to lo(i) has landed, before its third
  // everything up to hi(i); `n` = the reads allowed to be still in flight (6 6 / .. / 4 / 2 0)
  auto wait_lo = [&](int buf, int n) {
    switch (n) {
      case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wf[buf][1]), "+v"(wf[buf][3]) :: "memory"); break;
      case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wf[buf][1]), "+v"(wf[buf][3]) :: "memory"); break;
      default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[buf][1]), "+v"(wf[buf][3]) :: "memory"); break;
    }
  };
  auto wait_hi = [&](int buf, int n) {
    switch (n) {
      case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wf[buf][0]), "+v"(wf[buf][2]) :: "memory"); break;
      case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(wf[buf][0]), "+v"(wf[buf][2]) :: "memory"); break;
      default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[buf][0]), "+v"(wf[buf][2]) :: "memory"); break;
    }
  };
  // (reads in flight behind lo(i) / hi(i) at the two wait points of step i)
  auto lo_inflight = [](int i) { return i <= 6 ? 6 : 2; };
  auto hi_inflight = [](int i) { return i < 6 ? 6 : (i == 6 ? 4 : 0); };
  // x' fragments of a slab (four k-steps x (hi, lo)): read ONCE per slab and kept for its
  // eight output tiles (re-reading them per tile pair was a third of the reduce phase's LDS
  // traffic, and the latency of the weight reads rides on that traffic)
  f32x4 xs[4][2];
  auto rd_xs = [&](int buf) {
    int soff = (int)strip_b + buf * (4 * kStripFloats * 4);
    asm volatile("" : "+s"(soff));
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const unsigned a0 = fa[2 * s4] + (unsigned)soff, a1 = fa[2 * s4 + 1] + (unsigned)soff;
      asm volatile("ds_read_b128 %0, %1" : "=v"(xs[s4][0]) : "v"(a0) : "memory");
      asm volatile("ds_read_b128 %0, %1" : "=v"(xs[s4][1]) : "v"(a1) : "memory");
    }
  };
  auto xs_landed = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(xs[0][0]), "+v"(xs[0][1]), "+v"(xs[1][0]), "+v"(xs[1][1]),
                   "+v"(xs[2][0]), "+v"(xs[2][1]), "+v"(xs[3][0]), "+v"(xs[3][1]) :: "memory");
  };
  // accumulator tile t (32 channels) of a slab -> the strip, channel-per-register to
  // row-major: row = pixel, fp32 chunk 8 t + 2 qd + half, swizzled by the row
  auto to_strip = [&](float* st, const f32x16& a, int t) {
    const int sw = opaque(fsw);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      f32x4 v = {a[4 * qd], a[4 * qd + 1], a[4 * qd + 2], a[4 * qd + 3]};
      *reinterpret_cast<f32x4*>(st + px * 64 + (((8 * t + 2 * qd + half) ^ sw) << 2)) = v;
    }
  };
  auto row_chunk = [&](float* st, int row, int c) -> float* {
    return st + row * 64 + ((c ^ opaque(row & 15)) << 2);
  };
  // One epilogue item = row 16 it + erow of a slab, this lane's 8-channel group, in two
  // steps so that the weight DMA of the half-slot is issued while the raw values travel:
  // epi_read -- the raw accumulators from the strip; epi_finish -- (* scale, + bias, +
  // residual, ReLU, split) -> HBM and -> the strip (x' fragments of the reduce product).
  // (__fmul_rn: the scale is rounded on its own, as where the unfused kernel applies it.)
  auto epi_read = [&](float* st, int it, f32x4* v0, f32x4* v1) {
    const int row = 16 * it + erow;
    *v0 = *reinterpret_cast<const f32x4*>(row_chunk(st, row, 2 * G));
    *v1 = *reinterpret_cast<const f32x4*>(row_chunk(st, row, 2 * G + 1));
  };
  auto epi_finish = [&](float* st, int it, int slab, f32x4 v0, f32x4 v1) {
    const int row = 16 * it + erow;
    float* sp0 = row_chunk(st, row, 2 * G);
    float* sp1 = row_chunk(st, row, 2 * G + 1);
    float v[8];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      v[2 * d] = clamp_relu(__fmul_rn(v0[2 * d], g.scale3) + bias3v[0][2 * d] + mix_add_f16<0>(res[it][0][d], res[it][1][d]));
      v[2 * d + 1] = clamp_relu(__fmul_rn(v0[2 * d + 1], g.scale3) + bias3v[0][2 * d + 1] + mix_add_f16<1>(res[it][0][d], res[it][1][d]));
      v[4 + 2 * d] = clamp_relu(__fmul_rn(v1[2 * d], g.scale3) + bias3v[1][2 * d] + mix_add_f16<0>(res[it][0][2 + d], res[it][1][2 + d]));
      v[5 + 2 * d] = clamp_relu(__fmul_rn(v1[2 * d + 1], g.scale3) + bias3v[1][2 * d + 1] + mix_add_f16<1>(res[it][0][2 + d], res[it][1][2 + d]));
    }
    sat = sat_fold8(v, sat);
    f32x4 ehi, elo;
    split8_clamped(v, &ehi, &elo);
    if (m0 + row < g.M) {
      float* xp = xbase + 64 * slab + (row_off(tile, it) >> 2);
      *reinterpret_cast<f32x4*>(xp) = ehi;
      *reinterpret_cast<f32x4*>(xp + 4) = elo;
    }
    *reinterpret_cast<f32x4*>(sp0) = ehi;
    *reinterpret_cast<f32x4*>(sp1) = elo;
  };
  // LDS writes / reads of this half-slot done, then the barrier (kind: 2 MFMA, 4 epilogue)
  auto end_half = [&](int kind) {
    wait_lgkm0();
    stamp(kind);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stamp(kind == 2 ? 5 : 6);
  };

  // ---- end of a tile: t1' = relu(acc1 * scale + bias) in split form, 64 channels of the
  // block's 32 pixels from one strip buffer (row-major: 8 lanes cover a row, 8 rows per pass)
  auto final_items = [&](float* fst, int n0, long bm0, f32x4 b0, f32x4 b1) {
    const int frow = opaque(lane) >> 3, fcol8 = opaque(lane) & 7;
    const int n = n0 + fcol8 * 8;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int row = ps * 8 + frow;
      const long m = bm0 + row;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(row_chunk(fst, row, 2 * fcol8));
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(row_chunk(fst, row, 2 * fcol8 + 1));
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = clamp_relu(v0[e] + b0[e]);
        v[4 + e] = clamp_relu(v1[e] + b1[e]);
      }
      sat = sat_fold8(v, sat);
      f32x4 hi, lo;
      split8_clamped(v, &hi, &lo);
      if (m < g.M) {
        float* tp = g.T1 + m * N1 + n;
        *reinterpret_cast<f32x4*>(tp) = hi;
        *reinterpret_cast<f32x4*>(tp + 4) = lo;
      }
    }
  };
  if (h == 0) {
    // =============================== E-wave ===============================================
    f32x4 t2h[16], t2l[16];
    auto load_t2 = [&](int t) {
      const long mf = ((long)t * 128 + p * 32 + px < g.M) ? (long)t * 128 + p * 32 + px : (long)g.M - 1;
      const float* tp = g.T2 + mf * P + half * 8;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        t2h[s] = *reinterpret_cast<const f32x4*>(tp + s * 16);
        t2l[s] = *reinterpret_cast<const f32x4*>(tp + s * 16 + 4);
      }
    };
    set_geometry(tile);
    load_t2(tile);
    load_rows(tile, 0);
    load_bias(0);
    wait_vmcnt<0>();
    res_landed();
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(t2h[s]), "+v"(t2l[s]));
    __builtin_amdgcn_s_barrier();   // pairs 0 and 1 (issued by the R-waves) have landed
    stamp(0);
    for (;;) {
      const bool has_next = tile + (int)gridDim.x < ntiles;
      const int ntile = has_next ? tile + (int)gridDim.x : tile;
      for (int s = 0; s <= NSLAB; ++s) {
        float* st = strip0 + (s & 1) * (4 * kStripFloats);
        f32x16 acc3[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc3[t][r] = 0.f;
        // ---- half-slots 4 s, 4 s + 1: E(s), k-steps 0..7 and 8..15 --------------------------
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int q = 4 * s + hh;
          if (s < NSLAB) {
            // eight k-steps (two tiles) in one software pipeline: the fragments of step i + 1
            // are requested before step i multiplies
            __builtin_amdgcn_s_setprio(1);   // the multiplying wave outranks its partner's VALU / DMA issue
            set_pair(q);
            rd_lo(0, 0); rd_hi(0, 0); rd_lo(1, 1); rd_hi(1, 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int ks = 8 * hh + i, b = i & 1;
              wait_lo(b, lo_inflight(i));
              acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wf[b][1]), h8(t2h[ks]), acc3[0], 0, 0, 0);
              acc3[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wf[b][3]), h8(t2h[ks]), acc3[1], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              if (i < 6) rd_lo(b, i + 2);
              wait_hi(b, hi_inflight(i));
              acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wf[b][0]), h8(t2l[ks]), acc3[0], 0, 0, 0);
              acc3[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wf[b][2]), h8(t2l[ks]), acc3[1], 0, 0, 0);
              acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wf[b][0]), h8(t2h[ks]), acc3[0], 0, 0, 0);
              acc3[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wf[b][2]), h8(t2h[ks]), acc3[1], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              if (i < 6) rd_hi(b, i + 2);
            }
          }
          // own pieces of pair q + 1 (issued at the top of half-slot 4 s - 1): only the six
          // residual / bias loads behind them may still fly
          if (hh == 0) wait_vmcnt<6>();
          // the last step has nothing to multiply: the NEXT tile's t2 fragments are requested
          // here (E(15) was the last reader of the current ones) and travel under the R-wave's
          // last epilogue / product
          if (hh == 1 && s == NSLAB && has_next) load_t2(ntile);
          __builtin_amdgcn_s_setprio(0);
          end_half(2);
        }
        // ---- half-slots 4 s + 2, 4 s + 3: weights for E(s + 1); epilogue of channels 0..31 ----
        // (the raw tiles leave the registers here, not inside the MFMA half-slots; channels
        // 32..63 go to the R-wave through the strip: its epilogue runs two half-slots later)
        f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
        // (an inline-asm load's destination must stay live until it lands: no load is
        // issued whose result nothing will read -- the allocator would reuse the registers)
        if (s < NSLAB) {
          to_strip(st, acc3[0], 0);
          epi_read(st, 0, &r0, &r1);
        }
        // weights for E(s + 1): step 15 has no successor in this tile (pairs 64, 65 do not
        // exist), step 16 requests the next tile's pairs 0, 1
        const bool feed = s < NSLAB - 1 || (s == NSLAB && has_next);
        if (feed) issue_pair(4 * s + 4);
        if (s < NSLAB) {
          // the residual / bias of slab s were requested before the pieces of this half-slot
          if (feed) wait_vmcnt<8>();
          else wait_vmcnt<0>();
          res_landed();
          epi_finish(st, 0, s, r0, r1);
        }
        end_half(4);
        if (s < NSLAB) {
          to_strip(st, acc3[1], 1);
          epi_read(st, 1, &r0, &r1);
        }
        if (feed) issue_pair(4 * s + 5);
        if (s < NSLAB) epi_finish(st, 1, s, r0, r1);
        // (an asm load nobody reads is never issued: the allocator would reuse its registers)
        if (s < NSLAB - 1) {
          load_rows(tile, s + 1);
          load_bias(s + 1);
        } else if (s == NSLAB && has_next) {   // the next tile's slab 0
          load_rows(ntile, 0);
          load_bias(0);
        }
        // pair 4 s + 4 (top of the previous half-slot): 8 pieces + 6 loads behind it
        if (feed) wait_vmcnt<14>();
        end_half(4);
      }
      // (the reduce epilogue is the R-wave's: it runs under this wave's first product of the
      // next tile)
      stamp(7);
      if (!has_next) break;
      tile = ntile;
      set_geometry(tile);
      qbase = (qbase + NPAIR) % 3;
    }
  } else {
    // =============================== R-wave ===============================================
    f32x16 acc1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[u][r] = 0.f;
    set_geometry(tile);
    issue_pair(0);
    issue_pair(1);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    stamp(0);
    // The reduce epilogue of a tile -- t1' = relu(acc1 * scale + bias) in split form -- runs in
    // the four half-slots of the NEXT tile's step 0, where this wave has nothing else to do
    // (no slab behind it yet) and the E-wave multiplies / runs its first epilogue: a quarter
    // (two accumulator tiles = 64 channels) per half-slot through strip buffer 1 of this block,
    // which nothing else touches during step 0; same-wave LDS ordering, no extra barrier.
    // 128 KB of stores per workgroup leave under the next tile's MFMAs instead of after them.
    bool pending = false;
    long pm0 = 0;
    auto final_quarter = [&](int k) {
      // (gfx9 counts stores in vmcnt: the bias is requested in front of the quarter's stores)
      const int fcol8 = opaque(lane) & 7;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.bias1 + 64 * k + fcol8 * 8);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(g.bias1 + 64 * k + fcol8 * 8 + 4);
      float* fst = strip0 + 4 * kStripFloats;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc1[2 * k + t] = acc1[2 * k + t] * g.scale1;
        to_strip(fst, acc1[2 * k + t], t);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[2 * k + t][e] = 0.f;
      }
      final_items(fst, 64 * k, pm0, b0, b1);
    };
    for (;;) {
      const bool has_next = tile + (int)gridDim.x < ntiles;
      for (int s = 0; s <= NSLAB; ++s) {
        // slab s - 1 lives in strip buffer (s - 1) & 1
        float* st = strip0 + ((s + 1) & 1) * (4 * kStripFloats);
        // ---- half-slots 4 s, 4 s + 1: weights for R(s - 1); epilogue of channels 32..63 -------
        // (step 0 has no slab behind it: pairs 2, 3 do not exist)
        f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
        if (s >= 1) {
          epi_read(st, 0, &r0, &r1);
          issue_pair(4 * s + 2);
          wait_vmcnt<8>();
          res_landed();
          epi_finish(st, 0, s - 1, r0, r1);
        } else if (pending) {
          final_quarter(0);
        }
        end_half(4);
        if (s >= 1) {
          epi_read(st, 1, &r0, &r1);
          issue_pair(4 * s + 3);
          epi_finish(st, 1, s - 1, r0, r1);
        } else if (pending) {
          final_quarter(1);
        }
        // pair 4 s + 2 (top of the previous half-slot): 8 pieces (+ 6 residual / bias loads) behind it;
        // the last step has no slab to prefetch for
        if (s < NSLAB) {
          load_rows(tile, s);
          load_bias(s);
        }
        if (s >= 1) {
          if (s < NSLAB) wait_vmcnt<14>();
          else wait_vmcnt<8>();
        }
        end_half(4);
        // ---- half-slots 4 s + 2, 4 s + 3: R(s - 1), output rows 0..127 and 128..255 ------------
        if (s >= 1) {
          rd_xs((s + 1) & 1);
          xs_landed();
        }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int q = 4 * s + 2 + hh;
          if (s >= 1) {
            __builtin_amdgcn_s_setprio(1);   // the multiplying wave outranks its partner's VALU / DMA issue
            set_pair(q);
            rd_lo(0, 0); rd_hi(0, 0); rd_lo(1, 1); rd_hi(1, 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int b = i & 1, u0 = 2 * (2 * hh + (i >> 2)), k4 = i & 3;
              wait_lo(b, lo_inflight(i));
              acc1[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wf[b][1]), h8(xs[k4][0]), acc1[u0], 0, 0, 0);
              acc1[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wf[b][3]), h8(xs[k4][0]), acc1[u0 + 1], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              if (i < 6) rd_lo(b, i + 2);
              wait_hi(b, hi_inflight(i));
              acc1[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wf[b][0]), h8(xs[k4][1]), acc1[u0], 0, 0, 0);
              acc1[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wf[b][2]), h8(xs[k4][1]), acc1[u0 + 1], 0, 0, 0);
              acc1[u0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wf[b][0]), h8(xs[k4][0]), acc1[u0], 0, 0, 0);
              acc1[u0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8(wf[b][2]), h8(xs[k4][0]), acc1[u0 + 1], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              if (i < 6) rd_hi(b, i + 2);
            }
          }
          if (s == 0 && pending) final_quarter(2 + hh);
          if (hh == 0 && s >= 1) {   // own pieces of pair 4 s + 3
            if (s < NSLAB) wait_vmcnt<6>();
            else wait_vmcnt<0>();
          }
          __builtin_amdgcn_s_setprio(0);
          end_half(2);
        }
      }
      pending = true;
      pm0 = m0;
      stamp(7);
      if (!has_next) break;
      tile += (int)gridDim.x;
      set_geometry(tile);
      qbase = (qbase + NPAIR) % 3;
    }
    // the last tile's reduce epilogue
#pragma unroll
    for (int k = 0; k < 4; ++k) final_quarter(k);
  }
  report_saturation(g.status, sat);
  if constexpr (PROF) {
    if (lane == 0 && (wave == 0 || wave == 4) && g.prof)
      for (int k = 0; k < 8; ++k) g.prof[((long)blockIdx.x * 2 + h) * 8 + k] = tprof[k];
  }
}

int launch_chain3(const ChainArgs& a, hipStream_t s) {
  constexpr int lds = 160 * 1024;
  auto kern = a.prof ? chain3_kernel<true> : chain3_kernel<false>;
  MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(kern), lds));
  const int ntiles = (a.M + 127) / 128;
  int cus = 0;
  MILAN_TRY(device_cus8(&cus));
  // persistent: one workgroup per CU walks tiles b, b + grid, ...; ChainArgs::prof (timing
  // experiments) needs 16 counters per workgroup
  hipLaunchKernelGGL(kern, dim3(ntiles < cus ? ntiles : cus), dim3(512), lds, s, a, ntiles);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace milan
