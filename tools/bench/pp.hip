// Round-4 main-loop prototypes for the 256 x 256 split16 tile (standalone: no libmilan_hip).
//
//   E1  MFMA issue rate of dependent chains (v_mfma_f32_32x32x16_f16), 1 or 2 waves per SIMD
//   E2  L2 -> LDS DMA rate per CU (global_load_lds_dwordx4), 64-B and 128-B row pieces, and
//       the same bytes as plain global_load_dwordx4
//   E3  C = A . W^T on split-format operands, 1x1 geometry: the lockstep loop of
//       igemm_split16_linp_kernel (VAR 0) against ping-pong schedules (VAR >= 1)
//
// hipcc --offload-arch=gfx950 -O3 -std=c++17 pp.hip -o pp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__device__ inline f16x8 as_f16x8(f32x4 v) { return __builtin_bit_cast(f16x8, v); }

// ------------------------------------------------------------------------------------------
// E1
// ------------------------------------------------------------------------------------------
template <int DIST>
__global__ __launch_bounds__(512) void e1_kernel(float* out, long long* cyc, int iters) {
  f32x16 acc[DIST];
#pragma unroll
  for (int d = 0; d < DIST; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  f32x4 a = {1.f, 2.f, 3.f, (float)threadIdx.x}, b = {0.5f, 0.25f, 1.f, 2.f};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 24 / DIST; ++rep)
#pragma unroll
      for (int d = 0; d < DIST; ++d)
        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(a), as_f16x8(b), acc[d], 0, 0, 0);
  }
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < DIST; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[d][r];
  if (s == 123.456f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int DIST>
static void run_e1(int threads) {
  float* out; long long* cyc;
  CK(hipMalloc((void**)&out, 64)); CK(hipMalloc((void**)&cyc, 64));
  const int iters = 2000;
  hipLaunchKernelGGL(e1_kernel<DIST>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a, 0));
  hipLaunchKernelGGL(e1_kernel<DIST>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double mfmas = (double)iters * (24 / DIST) * DIST;
  printf("E1 dist %d, %d waves/SIMD: %.1f cycles per MFMA per wave (clock64), %.1f per SIMD-MFMA; %.3f ms -> %.0f TF f16\n",
         DIST, threads / 256, c / mfmas, c / mfmas / (threads / 256), ms,
         mfmas * (threads / 64) * 256 * 32768.0 / ms / 1e9);
  hipFree(out); hipFree(cyc);
}

// ------------------------------------------------------------------------------------------
// E2: every workgroup (512 threads) streams k-tiles of 256 A rows + 256 W rows into a 5-slot ring,
// leaving 3 k-tiles in flight; nothing is consumed.  A panel per workgroup group (tiles_n
// workgroups share one), W panel shared by all.
//   MODE 0: 16 rows x 64 B per piece (the split16 kernels' pattern), k-tile = 64 B per row
//   MODE 1: 8 rows x 128 B per piece, k-tile = 128 B per row (whole lines), half the k-tiles
//   MODE 2: as 0 with plain global_load_dwordx4 into registers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int xcd_tile_(int b, int T) {
  const int q = T >> 3, r = T & 7;
  const int xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
template <int MODE>
__global__ __launch_bounds__(512) void e2_kernel(const float* A, const float* W, int K, int tiles_n,
                                                 long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int panel = xcd_tile_(blockIdx.x, gridDim.x) / tiles_n;
  const float* Ab = A + (long)panel * 256 * K;
  constexpr int ROWB = MODE == 1 ? 32 : 16;  // floats per row per k-tile
  constexpr int LR = MODE == 1 ? 64 : 128;   // rows per pass of 512 threads
  constexpr int ITERS = 256 / LR;
  const int lrow = MODE == 1 ? tid >> 3 : tid >> 2;
  const int kc = MODE == 1 ? ((tid & 7) ^ ((tid >> 3) & 7)) : ((tid & 3) ^ ((tid >> 4) & 3));
  const float* ra[ITERS]; const float* rb[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    ra[it] = Ab + (long)(it * LR + lrow) * K + kc * 4;
    rb[it] = W + (long)(it * LR + lrow) * K + kc * 4;
  }
  const int nk = K / ROWB;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const long long t0 = clock64();
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (MODE == 2) {
      f32x4 v[2 * ITERS];
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        v[2 * it] = *reinterpret_cast<const f32x4*>(ra[it] + kt * ROWB);
        v[2 * it + 1] = *reinterpret_cast<const f32x4*>(rb[it] + kt * ROWB);
      }
#pragma unroll
      for (int it = 0; it < 2 * ITERS; ++it) acc += v[it];
    } else {
      float* base = smem + slot * (512 * ROWB);
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(ra[it] + kt * ROWB),
                                         (LDS_AS void*)(base + (it * 8 + wave) * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(rb[it] + kt * ROWB),
                                         (LDS_AS void*)(base + (ITERS * 8 + it * 8 + wave) * 256), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * 2 * ITERS) : "memory");
      slot = slot + 1 == (MODE == 1 ? 2 : 5) ? 0 : slot + 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = clock64();
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
  if (tid == 0) atomicAdd((unsigned long long*)cyc, (unsigned long long)(t1 - t0));
}

template <int MODE>
static void run_e2(const float* A, const float* W, int K, int tiles_n, int grid, const char* what) {
  long long* cyc; float* sink;
  CK(hipMalloc((void**)&cyc, 64)); CK(hipMalloc((void**)&sink, 64));
  CK(hipMemset(cyc, 0, 64));
  auto kern = e2_kernel<MODE>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 160 * 1024, 0, A, W, K, tiles_n, cyc, sink);
  CK(hipDeviceSynchronize());
  CK(hipMemset(cyc, 0, 64));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a, 0));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 160 * 1024, 0, A, W, K, tiles_n, cyc, sink);
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double bytes = (double)grid * 512.0 * K * 4.0;
  printf("E2 mode %d (%s) K=%d tiles_n=%d grid=%d: %.1f B/clk/CU (clock64), %.3f ms -> %.2f TB/s into LDS/regs\n",
         MODE, what, K, tiles_n, grid, 512.0 * K * 4.0 / ((double)c / grid), ms, bytes / ms / 1e9);
  hipFree(cyc); hipFree(sink);
}

// ------------------------------------------------------------------------------------------
// E4: what one LDS-DMA instruction costs the issuing wave.  NT/64 waves per workgroup (one
// workgroup per CU); every wave issues bursts of BURST pieces (16 rows x 64 B, or 8 rows x 128 B
// with LINE128), then idles `gap` s_sleep units; at most 12 pieces in flight per wave.  Sources
// are L2-resident (every workgroup walks the same 1 MB panel).
//   DEST 0: ring destination (M0 rewritten per piece), 1: one fixed destination
//   BUF 1: raw_buffer_load_lds (descriptor + 32-bit voffset + scalar soffset) instead of global_load_lds
// ------------------------------------------------------------------------------------------
template <int NT, int BURST, int DEST, int LINE128, int BUF>
__global__ __launch_bounds__(NT) void e4_kernel(const float* A, int K, int bursts, int gap, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = LINE128 ? lane >> 3 : lane >> 2;
  const int kc = LINE128 ? ((lane & 7) ^ (lrow & 7)) : ((lane & 3) ^ ((lane >> 4) & 3));
  const int rowb = LINE128 ? 32 : 16;
  const float* src = A + (long)(wave * 16 + lrow) * K + kc * 4;
  const unsigned voff = (unsigned)(((long)(wave * 16 + lrow) * K + kc * 4) * 4);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
  const int nk = K / rowb;
  long long busy = 0;
  int kt = 0, slot = 0;
  for (int b = 0; b < bursts; ++b) {
    const long long t0 = clock64();
#pragma unroll
    for (int q = 0; q < BURST; ++q) {
      float* dst = smem + wave * (16 * 256) + (DEST ? 0 : slot * 256);
      if constexpr (BUF) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)dst, 16, voff, kt * rowb * 4, 0, 0);
      } else {
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + kt * rowb), (LDS_AS void*)dst, 16, 0, 0);
      }
      kt = kt + 1 == nk ? 0 : kt + 1;
      slot = (slot + 1) & 15;
    }
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    busy += clock64() - t0;
    for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(8);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) atomicAdd((unsigned long long*)cyc, (unsigned long long)busy);
}

template <int NT, int BURST, int DEST, int LINE128, int BUF>
static void run_e4(const float* A, int K, int gap) {
  long long* cyc; CK(hipMalloc((void**)&cyc, 64));
  auto kern = e4_kernel<NT, BURST, DEST, LINE128, BUF>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int bursts = 2048 / BURST;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(cyc, 0, 64));
    hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 160 * 1024, 0, A, K, bursts, gap, cyc);
    CK(hipDeviceSynchronize());
  }
  long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double per = (double)c / (256.0 * (NT / 64) * bursts * BURST);
  printf("E4 %d waves/CU burst %d %s %s %s gap %d: %.0f cycles per piece per wave (incl. clock64 overhead / burst) -> %.1f B/clk/CU while issuing\n",
         NT / 64, BURST, DEST ? "fixed-dest" : "ring-dest", LINE128 ? "128B-rows" : "64B-rows", BUF ? "buffer_load" : "global_load",
         gap, per, 1024.0 * (NT / 64) / per);
  hipFree(cyc);
}

// ------------------------------------------------------------------------------------------
// E3
// ------------------------------------------------------------------------------------------
struct P {
  const float* A;   // [M][K] split format (K slots of 4 B: groups of 8 channels [hi x8 | lo x8])
  const float* W;   // [N][K]
  float* C;         // [M][N] fp32
  int M, N, K;
  int tiles_m, tiles_n;
  long long* prof;  // [8 waves][8]
  long long* total; // [2]: main-loop cycles of wave 0 / wave 4 summed over workgroups (always on)
  int store;        // 0: main loop only (checksum guard), 1: store C
  int zrows;        // pp2 buffer mode: A rows with m % 7 == 3 are fetched out of range (must read as zeros)
  int amod;         // A row m is read from row m % amod (an L2-resident operand), 0 = off
};

__device__ __forceinline__ int xcd_tile(int b, int T) {
  const int q = T >> 3, r = T & 7;
  const int xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// VAR 0: lockstep (all waves: read fragments, 24 MFMAs with the DMA pieces ahead (waves 0-3) or
//        behind (waves 4-7), counted wait, barrier) -- the round-3 loop
// VAR 1: ping-pong, one phase per k-tile and group, MFMAs product-major (dependent distance 8)
// VAR 2: ping-pong, MFMAs accumulator-major (dependent distance 1: hl, lh, hh back to back)
// VAR 3: ping-pong, product-major, s_setprio 1 over the MFMA phase
// VAR 4: lockstep with all 12 fragments up front and product-major MFMAs
template <int VAR, int S = 5>
__global__ __launch_bounds__(512, 2) void pp_kernel(P p) {
  constexpr int BM = 256, BN = 256, BK = 16, TM = 4, TN = 2, LOADS = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + S * BM * BK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int T = p.tiles_m * p.tiles_n;
  const int tile = xcd_tile(blockIdx.x, T);
  const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
  const int lrow = tid >> 2, kc = (tid & 3) ^ ((tid >> 4) & 3);
  const float *ra[2], *rb[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    int m = tile_m * BM + it * 128 + lrow; m = m < p.M ? m : p.M - 1;
    int n = tile_n * BN + it * 128 + lrow; n = n < p.N ? n : p.N - 1;
    if (p.amod) m %= p.amod;
    ra[it] = p.A + (long)m * p.K + kc * 4;
    rb[it] = p.W + (long)n * p.K + kc * 4;
  }
  const int nk = p.K / BK;
  auto issue_tile = [&](int kt, int slot) {
    const int k = (kt < nk ? kt : nk - 1) * BK;
#pragma unroll
    for (int it = 0; it < 2; ++it)
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(ra[it] + k),
          (LDS_AS void*)(As + slot * (BM * BK) + wave * (16 * BK) + it * (128 * BK)), 16, 0, 0);
#pragma unroll
    for (int it = 0; it < 2; ++it)
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(rb[it] + k),
          (LDS_AS void*)(Bs + slot * (BN * BK) + wave * (16 * BK) + it * (128 * BK)), 16, 0, 0);
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int swz = (frow >> 2) & 3;
  const int chi = (2 * fhalf) ^ swz, clo = (2 * fhalf + 1) ^ swz;
  const int a_hi = (wm * 128 + frow) * BK + (chi << 2), a_lo = (wm * 128 + frow) * BK + (clo << 2);
  const int b_hi = (wn * 64 + frow) * BK + (chi << 2), b_lo = (wn * 64 + frow) * BK + (clo << 2);
  f32x4 ah[TM], al[TM], bh[TN], bl[TN];
  if constexpr (VAR == 6 || VAR == 7) {
#pragma unroll
    for (int i = 0; i < TM; ++i) { ah[i] = f32x4{1.f, 2.f, 3.f, (float)lane}; al[i] = ah[i]; }
#pragma unroll
    for (int j = 0; j < TN; ++j) { bh[j] = f32x4{1.f, 2.f, 3.f, (float)lane}; bl[j] = bh[j]; }
  }
  auto read_frags = [&](int slot) {
    const float* Ab = As + slot * (BM * BK);
    const float* Bb = Bs + slot * (BN * BK);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[j] = *reinterpret_cast<const f32x4*>(Bb + b_hi + j * 32 * BK);
      bl[j] = *reinterpret_cast<const f32x4*>(Bb + b_lo + j * 32 * BK);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[i] = *reinterpret_cast<const f32x4*>(Ab + a_hi + i * 32 * BK);
      al[i] = *reinterpret_cast<const f32x4*>(Ab + a_lo + i * 32 * BK);
    }
  };
  auto mfmas = [&](auto order_tag) {
    constexpr int ORDER = decltype(order_tag)::value;
    if constexpr (ORDER == 0) {  // product-major: every accumulator once per product
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(ah[i]), as_f16x8(bl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(al[i]), as_f16x8(bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(ah[i]), as_f16x8(bh[j]), acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(ah[i]), as_f16x8(bl[j]), acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(al[i]), as_f16x8(bh[j]), acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(ah[i]), as_f16x8(bh[j]), acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  };
  long long pt_last = p.prof ? clock64() : 0;
  long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(k) do { if (p.prof) { const long long t1_ = clock64(); pt[k] += t1_ - pt_last; pt_last = t1_; } } while (0)

  const long long tt0 = clock64();
  // prologue: tiles 0 .. S-2 in flight, tile 0 landed
#pragma unroll
  for (int t = 0; t < S - 1; ++t) issue_tile(t, t);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LOADS) : "memory");
  __builtin_amdgcn_s_barrier();
  STAMP(0);

  if constexpr (VAR == 0 || VAR == 4) {
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      int nxt = cur + S - 1; nxt = nxt >= S ? nxt - S : nxt;
      read_frags(cur);
      if (wave < 4) { issue_tile(kt + S - 1, nxt); __builtin_amdgcn_sched_barrier(0); }
      if constexpr (VAR == 0) mfmas(std::integral_constant<int, 1>{});
      else mfmas(std::integral_constant<int, 0>{});
      if (wave >= 4) { __builtin_amdgcn_sched_barrier(0); issue_tile(kt + S - 1, nxt); }
      STAMP(1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LOADS) : "memory");
      STAMP(2);
      __builtin_amdgcn_s_barrier();
      STAMP(3);
      cur = cur + 1 == S ? 0 : cur + 1;
    }
  } else {
    // phase 2t: group 0 multiplies k-tile t, group 1 reads its fragments of k-tile t and issues
    // the DMA of k-tile t + S - 1; phase 2t + 1: the roles swap (group 0 reads k-tile t + 1).
    // Before the barrier that ends an even phase every wave has waited for its own pieces of
    // k-tile t + 1.
    auto R = [&](int kt, int slot) {  // slot = kt % S
      if constexpr (VAR != 6 && VAR != 7) read_frags(slot);
      else {
#pragma unroll
        for (int i = 0; i < TM; ++i) { asm volatile("" : "+v"(ah[i])); asm volatile("" : "+v"(al[i])); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { asm volatile("" : "+v"(bh[j])); asm volatile("" : "+v"(bl[j])); }
      }
      int nxt = slot + S - 1; nxt = nxt >= S ? nxt - S : nxt;
      if constexpr (VAR != 5 && VAR != 7) issue_tile(kt + S - 1, nxt);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto M = [&]() {
      if constexpr (VAR >= 3) __builtin_amdgcn_s_setprio(1);
      if constexpr (VAR == 8) {
#pragma unroll
        for (int i = 0; i < TM; ++i) { asm volatile("" :: "v"(ah[i]), "v"(al[i])); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { asm volatile("" :: "v"(bh[j]), "v"(bl[j])); }
      } else if constexpr (VAR == 2) mfmas(std::integral_constant<int, 1>{});
      else mfmas(std::integral_constant<int, 0>{});
      if constexpr (VAR >= 3) __builtin_amdgcn_s_setprio(0);
    };
    if (wm == 0) {
      int slot = 0;
      R(0, 0);                       // phase -1
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      STAMP(1);
      for (int kt = 0; kt < nk; ++kt) {
        M();                         // phase 2 kt
        __builtin_amdgcn_sched_barrier(0);
        STAMP(2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LOADS) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3);
        slot = slot + 1 == S ? 0 : slot + 1;
        R(kt + 1, slot);             // phase 2 kt + 1 (the last one reads a dummy tile)
        __builtin_amdgcn_sched_barrier(0);
        STAMP(4);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        STAMP(5);
      }
    } else {
      int slot = 0;
      __builtin_amdgcn_s_barrier();  // phase -1: idle
      __builtin_amdgcn_sched_barrier(0);
      STAMP(1);
      for (int kt = 0; kt < nk; ++kt) {
        R(kt, slot);                 // phase 2 kt
        __builtin_amdgcn_sched_barrier(0);
        STAMP(4);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LOADS) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        STAMP(5);
        M();                         // phase 2 kt + 1
        __builtin_amdgcn_sched_barrier(0);
        STAMP(2);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3);
        slot = slot + 1 == S ? 0 : slot + 1;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.total && lane == 0 && (wave & 3) == 0)
    atomicAdd((unsigned long long*)p.total + (wave >> 2), (unsigned long long)(clock64() - tt0));
  STAMP(6);
  if (p.store) {
    const int row0 = tile_m * BM + wm * 128, col0 = tile_n * BN + wn * 64;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int n = col0 + j * 32 + (lane & 31);
          if (m < p.M && n < p.N) p.C[(long)m * p.N + n] = acc[i][j][r];
        }
  } else {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 123.456f) p.C[0] = t;
  }
  STAMP(7);
  if (p.prof && lane == 0)
    for (int k = 0; k < 8; ++k) atomicAdd((unsigned long long*)p.prof + 8 * wave + k, (unsigned long long)pt[k]);
}


// ------------------------------------------------------------------------------------------
// pp32: ping-pong over 16-slot k-slabs, operands staged in 32-slot k-tiles = whole 128-B lines
// per row (8 rows x 128 B per DMA piece).  A ring 3 x 32 KB, W ring 2 x 32 KB.
//   phase 4T     group 0 multiplies slab 2T      | group 1 reads slab 2T, issues W(T+1)
//   phase 4T + 1 group 0 reads slab 2T+1, A(T+2) | group 1 multiplies slab 2T
//   phase 4T + 2 group 0 multiplies slab 2T+1    | group 1 reads slab 2T+1, issues A(T+2); vmcnt(4)
//   phase 4T + 3 group 0 reads slab 2T+2, W(T+2) | group 1 multiplies slab 2T+1
// PVAR 0: setprio 1 over the MFMA phase; 1: no setprio; 2: DMA issue before the fragment reads
// ------------------------------------------------------------------------------------------
template <int PVAR>
__global__ __launch_bounds__(512, 2) void pp32_kernel(P p) {
  constexpr int BM = 256, BN = 256, BK = 32, SA = 3, SW = 2, TM = 4, TN = 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + SA * BM * BK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int T = p.tiles_m * p.tiles_n;
  const int tile = xcd_tile(blockIdx.x, T);
  const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
  const float *ra[4], *rb[4];
  {
    const int prow = lane >> 3, pos = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 64 + wave * 8 + prow;
      const int kc = pos ^ ((row >> 1) & 7);
      int m = tile_m * BM + row; m = m < p.M ? m : p.M - 1;
      int n = tile_n * BN + row; n = n < p.N ? n : p.N - 1;
      if (p.amod) m %= p.amod;
      ra[it] = p.A + (long)m * p.K + kc * 4;
      rb[it] = p.W + (long)n * p.K + kc * 4;
    }
  }
  const int nkt = p.K / BK;
  auto issue_A = [&](int kt, int slot) {
    const int k = (kt < nkt ? kt : nkt - 1) * BK;
#pragma unroll
    for (int it = 0; it < 4; ++it)
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(ra[it] + k),
          (LDS_AS void*)(As + slot * (BM * BK) + (it * 64 + wave * 8) * BK), 16, 0, 0);
  };
  auto issue_W = [&](int kt, int slot) {
    const int k = (kt < nkt ? kt : nkt - 1) * BK;
#pragma unroll
    for (int it = 0; it < 4; ++it)
      __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(rb[it] + k),
          (LDS_AS void*)(Bs + slot * (BN * BK) + (it * 64 + wave * 8) * BK), 16, 0, 0);
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int sw = (frow >> 1) & 7;
  int off[2][2];  // [slab][hi / lo]: float offset of the lane's 16-byte chunk inside its row
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) off[s][hl] = ((s * 4 + fhalf * 2 + hl) ^ sw) * 4;
  const int a_base = (wm * 128 + frow) * BK, b_base = (wn * 64 + frow) * BK;
  f32x4 ah[TM], al[TM], bh[TN], bl[TN];
  auto read_frags = [&](int slotA, int slotW, auto slab_tag) {
    constexpr int SLAB = decltype(slab_tag)::value;
    const float* Ab = As + slotA * (BM * BK) + a_base;
    const float* Bb = Bs + slotW * (BN * BK) + b_base;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[j] = *reinterpret_cast<const f32x4*>(Bb + off[SLAB][0] + j * 32 * BK);
      bl[j] = *reinterpret_cast<const f32x4*>(Bb + off[SLAB][1] + j * 32 * BK);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[i] = *reinterpret_cast<const f32x4*>(Ab + off[SLAB][0] + i * 32 * BK);
      al[i] = *reinterpret_cast<const f32x4*>(Ab + off[SLAB][1] + i * 32 * BK);
    }
  };
  auto M = [&]() {
    if constexpr (PVAR != 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(ah[i]), as_f16x8(bl[j]), acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(al[i]), as_f16x8(bh[j]), acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(ah[i]), as_f16x8(bh[j]), acc[i][j], 0, 0, 0);
    if constexpr (PVAR != 1) __builtin_amdgcn_s_setprio(0);
  };
  long long pt_last = p.prof ? clock64() : 0;
  long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
  // ring slot of tile kt: A kt % 3, W kt % 2, tracked incrementally
  auto inc3 = [](int x) { return x + 1 == 3 ? 0 : x + 1; };
  // R of the even slab of tile kt: reads, then the W rows of tile kt + 1
  auto R_even = [&](int kt, int sA, int sW) {
    if constexpr (PVAR == 2) issue_W(kt + 1, sW ^ 1);
    read_frags(sA, sW, std::integral_constant<int, 0>{});
    if constexpr (PVAR != 2) issue_W(kt + 1, sW ^ 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // R of the odd slab of tile kt: reads, then the A rows of tile kt + 2 (slot (kt + 2) % 3)
  auto R_odd = [&](int kt, int sA, int sW) {
    const int s2 = sA == 0 ? 2 : sA - 1;  // (sA + 2) % 3
    if constexpr (PVAR == 2) issue_A(kt + 2, s2);
    read_frags(sA, sW, std::integral_constant<int, 1>{});
    if constexpr (PVAR != 2) issue_A(kt + 2, s2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  const long long tt0 = clock64();
  issue_A(0, 0); issue_W(0, 0); issue_A(1, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  BAR();
  STAMP(0);
  if (wm == 0) {
    int sA = 0, sW = 0;
    R_even(0, 0, 0);
    BAR();
    STAMP(1);
    for (int kt = 0; kt < nkt; ++kt) {
      M();
      __builtin_amdgcn_sched_barrier(0);
      STAMP(2);
      BAR();
      STAMP(3);
      R_odd(kt, sA, sW);
      __builtin_amdgcn_sched_barrier(0);
      STAMP(4);
      BAR();
      STAMP(5);
      M();
      __builtin_amdgcn_sched_barrier(0);
      STAMP(2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      BAR();
      STAMP(3);
      sA = inc3(sA); sW ^= 1;
      R_even(kt + 1, sA, sW);
      __builtin_amdgcn_sched_barrier(0);
      STAMP(4);
      BAR();
      STAMP(5);
    }
  } else {
    int sA = 0, sW = 0;
    BAR();
    STAMP(1);
    for (int kt = 0; kt < nkt; ++kt) {
      R_even(kt, sA, sW);
      __builtin_amdgcn_sched_barrier(0);
      STAMP(4);
      BAR();
      STAMP(5);
      M();
      __builtin_amdgcn_sched_barrier(0);
      STAMP(2);
      BAR();
      STAMP(3);
      R_odd(kt, sA, sW);
      __builtin_amdgcn_sched_barrier(0);
      STAMP(4);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      BAR();
      STAMP(5);
      M();
      __builtin_amdgcn_sched_barrier(0);
      STAMP(2);
      BAR();
      STAMP(3);
      sA = inc3(sA); sW ^= 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.total && lane == 0 && (wave & 3) == 0)
    atomicAdd((unsigned long long*)p.total + (wave >> 2), (unsigned long long)(clock64() - tt0));
  STAMP(6);
  if (p.store) {
    const int row0 = tile_m * BM + wm * 128, col0 = tile_n * BN + wn * 64;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int n = col0 + j * 32 + (lane & 31);
          if (m < p.M && n < p.N) p.C[(long)m * p.N + n] = acc[i][j][r];
        }
  } else {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 123.456f) p.C[0] = t;
  }
  STAMP(7);
  if (p.prof && lane == 0)
    for (int k = 0; k < 8; ++k) atomicAdd((unsigned long long*)p.prof + 8 * wave + k, (unsigned long long)pt[k]);
}


// ------------------------------------------------------------------------------------------
// pp2: ping-pong with NO vector ALU work in the read phase and an early hand-over.
//   * operands through buffer_load ... lds: descriptor + per-lane 32-bit row offset (loop
//     invariant) + SCALAR k offset -- no 64-bit pointer arithmetic per piece;
//   * 4-slot ring walked by a loop unrolled 4 x, so every ds_read is base register + immediate;
//   * the multiplying group arrives at the phase barrier TAIL MFMAs before its last one: the
//     other group's first MFMAs queue behind them and the matrix pipe does not idle through
//     the barrier latency.
// ------------------------------------------------------------------------------------------
template <int TAIL, int PRIO, int BUF = 0>
__global__ __launch_bounds__(512, 2) void pp2_kernel(P p) {
  constexpr int BM = 256, BN = 256, BK = 16, S = 4, TM = 4, TN = 2, LOADS = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + S * BM * BK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int T = p.tiles_m * p.tiles_n;
  const int tile = xcd_tile(blockIdx.x, T);
  const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
  const int lrow = tid >> 2, kc = (tid & 3) ^ ((tid >> 4) & 3);
  unsigned va[2], vb[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    int m = tile_m * BM + it * 128 + lrow;
    const bool mok = m < p.M && !(p.zrows && m % 7 == 3);
    if (p.amod) m %= p.amod;
    const int n = tile_n * BN + it * 128 + lrow;
    va[it] = (mok || !BUF) ? (unsigned)(((long)(mok ? m : 0) * p.K + kc * 4) * 4) : 0x80000000u;
    vb[it] = (n < p.N || !BUF) ? (unsigned)(((long)(n < p.N ? n : 0) * p.K + kc * 4) * 4) : 0x80000000u;
  }
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
  const int nk = p.K / BK;
  auto issue_tile = [&](int kt, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const int koff = (kt < nk ? kt : nk - 1) * (BK * 4);
    if constexpr (BUF) {
#pragma unroll
      for (int it = 0; it < 2; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (LDS_AS void*)(As + SLOT * (BM * BK) + wave * (16 * BK) + it * (128 * BK)),
                                                 16, va[it], koff, 0, 0);
#pragma unroll
      for (int it = 0; it < 2; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (LDS_AS void*)(Bs + SLOT * (BN * BK) + wave * (16 * BK) + it * (128 * BK)),
                                                 16, vb[it], koff, 0, 0);
    } else {
      // scalar base (+ k offset) + 32-bit per-lane offset: the saddr form of global_load_lds
      const char* sa = reinterpret_cast<const char*>(p.A) + koff;
      const char* sb = reinterpret_cast<const char*>(p.W) + koff;
#pragma unroll
      for (int it = 0; it < 2; ++it)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(sa + (size_t)va[it]),
            (LDS_AS void*)(As + SLOT * (BM * BK) + wave * (16 * BK) + it * (128 * BK)), 16, 0, 0);
#pragma unroll
      for (int it = 0; it < 2; ++it)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(sb + (size_t)vb[it]),
            (LDS_AS void*)(Bs + SLOT * (BN * BK) + wave * (16 * BK) + it * (128 * BK)), 16, 0, 0);
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
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
  // MFMA n of a k-tile, product-major: n = 8 * product + 2 * i + j; products hl, lh, hh
  auto mfma_n = [&](auto n_tag) {
    constexpr int N_ = decltype(n_tag)::value;
    constexpr int PR = N_ / 8, I = (N_ % 8) / 2, J = N_ % 2;
    if constexpr (PR == 0)
      acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(ah[I]), as_f16x8(bl[J]), acc[I][J], 0, 0, 0);
    else if constexpr (PR == 1)
      acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(al[I]), as_f16x8(bh[J]), acc[I][J], 0, 0, 0);
    else
      acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(ah[I]), as_f16x8(bh[J]), acc[I][J], 0, 0, 0);
  };
  auto mfma_range = [&](auto lo_tag, auto hi_tag) {
    constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
    [&]<int... Ns>(std::integer_sequence<int, Ns...>) {
      (mfma_n(std::integral_constant<int, LO + Ns>{}), ...);
    }(std::make_integer_sequence<int, HI - LO>{});
  };
  using I0 = std::integral_constant<int, 0>;
  using IH = std::integral_constant<int, 24 - TAIL>;
  using IE = std::integral_constant<int, 24>;
  const long long tt0 = clock64();
  long long pt_last = p.prof ? clock64() : 0;
  long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // prologue: tiles 0 .. S-2 in flight, tile 0 landed
  issue_tile(0, std::integral_constant<int, 0>{});
  issue_tile(1, std::integral_constant<int, 1>{});
  issue_tile(2, std::integral_constant<int, 2>{});
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LOADS) : "memory");
  BAR();
  STAMP(0);
  // one k-tile of a group; SLOT = kt % S (compile time), WAIT: this is the phase pair's even barrier
  // group 0: [M(kt) | wait | bar | tail] [R(kt + 1) | bar]
  // group 1: [R(kt) | wait | bar] [M(kt) | bar | tail]
  auto R = [&](int kt, auto slot_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    read_frags(slot_tag);
    issue_tile(kt + S - 1, std::integral_constant<int, (SLOT + S - 1) % S>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  if (wm == 0) {
    R(0, std::integral_constant<int, 0>{});
    BAR();
    auto step = [&](int kt, auto slot_tag) {
      constexpr int SLOT = decltype(slot_tag)::value;
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
      mfma_range(I0{}, IH{});
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LOADS) : "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mfma_range(IH{}, IE{});
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      R(kt + 1, std::integral_constant<int, (SLOT + 1) % S>{});
      BAR();
    };
    for (int kt = 0; kt < nk; kt += 4) {
      step(kt, std::integral_constant<int, 0>{});
      step(kt + 1, std::integral_constant<int, 1>{});
      step(kt + 2, std::integral_constant<int, 2>{});
      step(kt + 3, std::integral_constant<int, 3>{});
    }
  } else {
    BAR();
    auto step = [&](int kt, auto slot_tag) {
      R(kt, slot_tag);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LOADS) : "memory");
      BAR();
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
      mfma_range(I0{}, IH{});
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mfma_range(IH{}, IE{});
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt = 0; kt < nk; kt += 4) {
      step(kt, std::integral_constant<int, 0>{});
      step(kt + 1, std::integral_constant<int, 1>{});
      step(kt + 2, std::integral_constant<int, 2>{});
      step(kt + 3, std::integral_constant<int, 3>{});
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (p.total && lane == 0 && (wave & 3) == 0)
    atomicAdd((unsigned long long*)p.total + (wave >> 2), (unsigned long long)(clock64() - tt0));
  if (p.store) {
    const int row0 = tile_m * BM + wm * 128, col0 = tile_n * BN + wn * 64;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int n = col0 + j * 32 + (lane & 31);
          if (m < p.M && n < p.N) p.C[(long)m * p.N + n] = acc[i][j][r];
        }
  } else {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 123.456f) p.C[0] = t;
  }
  if (p.prof && lane == 0)
    for (int k = 0; k < 8; ++k) atomicAdd((unsigned long long*)p.prof + 8 * wave + k, (unsigned long long)pt[k]);
}

// reference: fp32 dot of (hi + lo) values, one thread per output
__global__ void ref_kernel(const float* A, const float* W, float* C, int M, int N, int K, int rows, int zrows) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = blockIdx.y;
  if (n >= N || m >= rows) return;
  if (zrows && m % 7 == 3) { C[(long)m * N + n] = 0.f; return; }
  const _Float16* a = reinterpret_cast<const _Float16*>(A + (long)m * K);
  const _Float16* w = reinterpret_cast<const _Float16*>(W + (long)n * K);
  double s = 0.0;
  for (int g = 0; g < K / 8; ++g)
    for (int e = 0; e < 8; ++e) {
      const double x = (double)(float)a[g * 16 + e] + (double)(float)a[g * 16 + 8 + e];
      const double y = (double)(float)w[g * 16 + e] + (double)(float)w[g * 16 + 8 + e];
      s += x * y;
    }
  C[(long)m * N + n] = (float)s;
}

static void fill_split(std::vector<float>& buf, size_t rows, int K, unsigned seed) {
  buf.resize(rows * K);
  _Float16* h = reinterpret_cast<_Float16*>(buf.data());
  unsigned s = seed;
  for (size_t r = 0; r < rows; ++r)
    for (int g = 0; g < K / 8; ++g)
      for (int e = 0; e < 8; ++e) {
        s = s * 1664525u + 1013904223u;
        const float x = ((s >> 8) & 0xffff) / 65536.0f * 2.f - 1.f;
        const _Float16 hi = (_Float16)x;
        h[(r * K + g * 8) * 2 + e] = hi;
        h[(r * K + g * 8) * 2 + 8 + e] = (_Float16)(x - (float)hi);
      }
}

static int g_amod = 0, g_zrows = 0;
template <int VAR, typename KERN>
static void run_e3k(KERN kern, const char* name, const float* dA, const float* dW, float* dC, const float* dRef, int M, int N, int K, int check_rows) {
  P p; p.A = dA; p.W = dW; p.C = dC; p.M = M; p.N = N; p.K = K;
  p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256; p.prof = nullptr; p.store = 1;
  long long* tot; CK(hipMalloc((void**)&tot, 64)); p.total = nullptr; p.amod = g_amod; p.zrows = g_zrows;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int grid = p.tiles_m * p.tiles_n;
  CK(hipMemset(dC, 0, (size_t)M * N * 4));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 160 * 1024, 0, p);
  CK(hipDeviceSynchronize());
  // check the first rows
  std::vector<float> c((size_t)check_rows * N), r((size_t)check_rows * N);
  CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r.data(), dRef, r.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0, scale = 0;
  const size_t ncheck = g_amod ? (size_t)(g_amod < check_rows ? g_amod : check_rows) * N : c.size();
  for (size_t i = 0; i < ncheck; ++i) { maxerr = fmax(maxerr, fabs((double)c[i] - r[i])); scale = fmax(scale, fabs((double)r[i])); }
  p.store = 0;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 160 * 1024, 0, p);
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int reps = 5;
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 160 * 1024, 0, p);
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  CK(hipMemset(tot, 0, 64)); p.total = tot;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 160 * 1024, 0, p);
  CK(hipDeviceSynchronize());
  long long ht[2]; CK(hipMemcpy(ht, tot, 16, hipMemcpyDeviceToHost)); p.total = nullptr;
  long long* pr; CK(hipMalloc((void**)&pr, 512)); CK(hipMemset(pr, 0, 512));
  p.prof = pr;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 160 * 1024, 0, p);
  CK(hipDeviceSynchronize());
  long long hp[64]; CK(hipMemcpy(hp, pr, 512, hipMemcpyDeviceToHost));
  const double kts = (double)grid * (K / 16);
  const double cyc_tile = (double)ht[0] / grid;
  printf("E3 %s amod=%d M=%d N=%d K=%d: %.3f ms main loop only -> %.0f TF-eq (%.0f TF f16 MFMA); max err %.3g (scale %.3g); unperturbed: %.0f cycles per k-tile incl. prologue (wave 4: %.0f) -> clock %.2f GHz\n",
         name, g_amod, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, 6.0 * M * N * K / ms / 1e9, maxerr, scale,
         (double)ht[0] / kts, (double)ht[1] / kts, cyc_tile * ((double)grid / 256.0) / (ms * 1e6));
  for (int wv = 0; wv < 8; wv += 4) {
    if (VAR == 0 || VAR == 4)
      printf("   wave %d per k-tile: compute+issue %.0f, dma wait %.0f, barrier %.0f | prologue %.0f per tile\n", wv,
             hp[8 * wv + 1] / kts, hp[8 * wv + 2] / kts, hp[8 * wv + 3] / kts, hp[8 * wv + 0] / (double)grid);
    else
      printf("   wave %d per k-tile: M %.0f, barrier after M %.0f, R %.0f, barrier after R %.0f | prologue %.0f per tile\n", wv,
             hp[8 * wv + 2] / kts, hp[8 * wv + 3] / kts, hp[8 * wv + 4] / kts, hp[8 * wv + 5] / kts,
             (hp[8 * wv + 0] + hp[8 * wv + 1]) / (double)grid);
  }
  hipFree(pr);
}
template <int VAR, int S = 5>
static void run_e3(const float* dA, const float* dW, float* dC, const float* dRef, int M, int N, int K, int check_rows) {
  char name[32]; snprintf(name, sizeof name, "var %d S=%d", VAR, S);
  run_e3k<VAR>(pp_kernel<VAR, S>, name, dA, dW, dC, dRef, M, N, K, check_rows);
}
template <int TAIL, int PRIO, int BUF = 0>
static void run_pp2(const float* dA, const float* dW, float* dC, const float* dRef, int M, int N, int K, int check_rows) {
  char name[48]; snprintf(name, sizeof name, "pp2 tail %d prio %d buf %d", TAIL, PRIO, BUF);
  run_e3k<1>(pp2_kernel<TAIL, PRIO, BUF>, name, dA, dW, dC, dRef, M, N, K, check_rows);
}
template <int PVAR>
static void run_e32(const float* dA, const float* dW, float* dC, const float* dRef, int M, int N, int K, int check_rows) {
  char name[32]; snprintf(name, sizeof name, "pp32 %d", PVAR);
  run_e3k<1>(pp32_kernel<PVAR>, name, dA, dW, dC, dRef, M, N, K, check_rows);
}

int main(int argc, char** argv) {
  const int which = argc > 1 ? atoi(argv[1]) : 7;
  g_amod = argc > 2 ? atoi(argv[2]) : 0;
  g_zrows = argc > 3 ? atoi(argv[3]) : 0;
  if (which & 1) {
    run_e1<1>(256); run_e1<2>(256); run_e1<3>(256); run_e1<4>(256); run_e1<8>(256);
    run_e1<1>(512); run_e1<2>(512); run_e1<4>(512); run_e1<8>(512);
  }
  if (which & 2) {
    const int K = 1024;
    float *A, *W;
    CK(hipMalloc((void**)&A, (size_t)256 * 256 * K * 4)); CK(hipMemset(A, 0x3c, (size_t)256 * 256 * K * 4));
    CK(hipMalloc((void**)&W, (size_t)256 * K * 4)); CK(hipMemset(W, 0x3c, (size_t)256 * K * 4));
    // tiles_n 1: every workgroup streams its own A panel (HBM); 4: four share one; 256: all share one (L2)
    for (int tn : {1, 4, 256}) {
      run_e2<0>(A, W, K, tn, 256, "lds-dma 64 B rows");
      run_e2<1>(A, W, K, tn, 256, "lds-dma 128 B rows");
      run_e2<2>(A, W, K, tn, 256, "global_load 64 B rows");
    }
    hipFree(A); hipFree(W);
  }
  if (which & 8) {
    const int K = 1024;
    float* A; CK(hipMalloc((void**)&A, (size_t)256 * K * 4)); CK(hipMemset(A, 0x3c, (size_t)256 * K * 4));
    for (int gap : {0, 12}) {
      run_e4<256, 1, 0, 0, 0>(A, K, gap); run_e4<256, 4, 0, 0, 0>(A, K, gap); run_e4<256, 4, 1, 0, 0>(A, K, gap);
      run_e4<256, 4, 0, 1, 0>(A, K, gap); run_e4<256, 4, 0, 0, 1>(A, K, gap); run_e4<256, 4, 0, 1, 1>(A, K, gap);
      run_e4<512, 4, 0, 0, 0>(A, K, gap); run_e4<512, 4, 0, 1, 0>(A, K, gap); run_e4<512, 4, 0, 0, 1>(A, K, gap);
      run_e4<128, 4, 0, 0, 0>(A, K, gap); run_e4<64, 4, 0, 0, 0>(A, K, gap); run_e4<64, 16, 0, 0, 0>(A, K, gap);
    }
    hipFree(A);
  }
  if (which & 4) {
    struct Shape { int M, N, K; };
    // M = 256 * 1024 rows: four rounds of 256 workgroups at N = 256
    for (Shape s : {Shape{262144, 256, 1024}, Shape{262144, 256, 2304}, Shape{65536, 1024, 256}, Shape{65536, 2048, 4544}}) {
      std::vector<float> hA, hW;
      fill_split(hA, (size_t)4096, s.K, 1u);  // 4096 distinct rows, tiled over M
      fill_split(hW, (size_t)s.N, s.K, 2u);
      float *dA, *dW, *dC, *dRef;
      CK(hipMalloc((void**)&dA, (size_t)s.M * s.K * 4));
      for (size_t r = 0; r < (size_t)s.M; r += 4096)
        CK(hipMemcpy(dA + r * s.K, hA.data(), (size_t)4096 * s.K * 4, hipMemcpyHostToDevice));
      CK(hipMalloc((void**)&dW, (size_t)s.N * s.K * 4));
      CK(hipMemcpy(dW, hW.data(), (size_t)s.N * s.K * 4, hipMemcpyHostToDevice));
      CK(hipMalloc((void**)&dC, (size_t)s.M * s.N * 4));
      const int check_rows = 512;
      CK(hipMalloc((void**)&dRef, (size_t)check_rows * s.N * 4));
      hipLaunchKernelGGL(ref_kernel, dim3((s.N + 255) / 256, check_rows), dim3(256), 0, 0, dA, dW, dRef, s.M, s.N, s.K, check_rows, g_zrows);
      CK(hipDeviceSynchronize());
      run_e3<0>(dA, dW, dC, dRef, s.M, s.N, s.K, check_rows);
      run_e3<3, 4>(dA, dW, dC, dRef, s.M, s.N, s.K, check_rows);
      run_pp2<0, 1, 0>(dA, dW, dC, dRef, s.M, s.N, s.K, check_rows);
      run_pp2<0, 1, 1>(dA, dW, dC, dRef, s.M, s.N, s.K, check_rows);
      run_pp2<0, 0, 1>(dA, dW, dC, dRef, s.M, s.N, s.K, check_rows);
      run_pp2<1, 1, 1>(dA, dW, dC, dRef, s.M, s.N, s.K, check_rows);
      hipFree(dA); hipFree(dW); hipFree(dC); hipFree(dRef);
    }
  }
  return 0;
}
