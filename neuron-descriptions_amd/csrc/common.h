// Shared host-side declarations for libmilan_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <map>

#include "../../include/milan_hip.h"

// experiments build (make EXPERIMENTS=1): measured-and-rejected kernels and the
// knobs that force tile configurations; see gemm.hip and DESIGN.md section 5
#ifndef MILAN_EXPERIMENTS
#define MILAN_EXPERIMENTS 0
#endif

namespace milan {

// ---- error plumbing (no C++ exceptions cross the C ABI) --------------------
void set_error(const char* fmt, ...);
#define MILAN_CHECK_HIP(expr)                                              \
  do {                                                                     \
    hipError_t _e = (expr);                                                \
    if (_e != hipSuccess) {                                                \
      milan::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,       \
                       hipGetErrorString(_e));                             \
      return (int)_e;                                                      \
    }                                                                      \
  } while (0)
#define MILAN_REQUIRE(cond, code, ...)                                     \
  do {                                                                     \
    if (!(cond)) {                                                         \
      milan::set_error(__VA_ARGS__);                                       \
      return (code);                                                       \
    }                                                                      \
  } while (0)
#define MILAN_TRY(expr)                                                    \
  do {                                                                     \
    int _r = (expr);                                                       \
    if (_r != 0) return _r;                                                \
  } while (0)

// ---- implicit-GEMM (conv / linear) -----------------------------------------
enum Epilogue : int {
  EPI_BIAS = 0,           // C = acc + bias
  EPI_BIAS_RELU = 1,      // C = relu(acc + bias)
  EPI_BIAS_RES_RELU = 2,  // C = relu(acc + bias + aux)
  EPI_BIAS_TANH = 3,      // C = tanh(acc + bias)
  EPI_BIAS_SIGMUL = 4,    // C = sigmoid(acc + bias) * aux
  EPI_BIAS_ADD = 5,       // C = acc + bias + aux
  EPI_LSTM = 6,           // acc + bias = an LSTM's gate pre-activations, columns
                          // gate-interleaved per 16 units ([i16 f16 g16 o16] per 64):
                          // C = h' (ldc), C2 = c' (ldc), aux = c (ldaux), Cs =
                          // optional split-format copy of h' (ldc); N = 4 H
  EPI_LSE = 7,            // x = acc + bias is NOT written: per row m and 64-column block cb,
                          // C[m * ldc + 2 cb] = { max x, sum exp(x - max) } (ldc = 2 ceil(N / 64)),
                          // and lse_x[m] = x[m][lse_tgt[m * lse_tgt_stride]] -- what a log-softmax
                          // read at one target column needs (the LM's rerank pass, decoder.hip)
};

// C[M][N] (row stride ldc) = epi( A (*) W^T + bias ), fp32 MFMA, exact f32.
// A is an NHWC activation tensor addressed as an implicit im2col matrix:
//   row m = (img, ho, wo), column kk = (kh, kw, cin) with cin fastest.
// A plain linear layer is the 1x1 case with H = W = 1 and `a_pix_stride`
// the row stride of the (possibly strided) input matrix.
struct GemmArgs {
  const float* A;
  const float* W;     // packed [N][Kp], Kp = round_up(K, 32), zero padded
  const float* bias;  // [N] or nullptr
  const float* aux;   // residual / multiplicand, row stride ldaux, or nullptr
  float* C;
  int M, N, K, Kp;
  int ldc, ldaux;
  // conv geometry
  int H, Wd, Cin, Ho, Wo, KH, KW, stride, pad;
  long a_pix_stride;  // floats between consecutive input pixels (>= Cin)
  long a_img_stride;  // floats between consecutive images
  int epilogue;
  const float* zero;  // >= 64 B of zeros (device)
  int flop_k;         // algorithmic K for FLOP accounting (0 = K)
  // split-f16 mode (see gemm.hip): A and W hold (hi,lo) f16 pairs
  int a_split;        // 1: run the 3xf16 MFMA main loop
  int out_split;      // 1: write C in split format (N % 8 == 0)
  int aux_split;      // 1: aux is in split format
  float acc_scale;    // accumulators are multiplied by this (1 / weight scale)
  int out_mode;       // filled by the launcher (OUT_*)
  // optional second operand source (split16 kernels only): k >= K1 reads a
  // 1x1 / stride `stride2` convolution input A2 (H2 x W2d pixels, no padding).
  // Used to fold a bottleneck's downsample conv into its c3 (K-concatenation).
  const float* W3;    // 3x3 convs: split weights packed chunk-major
                      // [N][Cin/16][9 taps][16 slots] (conv3x3_split16_kernel)
  const float* A2;
  int K1, H2, W2d, stride2;
  long a2_pix_stride, a2_img_stride;
  // anisotropic geometry (general igemm path only): when aniso != 0 the
  // horizontal stride / padding are stride_w / pad_w instead of stride / pad.
  // Used by the pixel-pair stem (encoder.hip).
  int aniso, stride_w, pad_w;
  int debug;          // MILAN_ABLATE timing experiments (0 in production): 1 no MFMA,
                      // 2 no DMA (igemm_kernel); 4 epilogue only, 8 main loop only (split16)
  int tile_hint;      // 0 auto (the kernel follows from the layer's N, K, kernel size);
                      // experiments: 1 / 2 the round-1 256x128x3 / 128x128x2 kernels, 3 / 4 / 5
                      // split16 256x256x4 / 256x128x3 / 128x256x3, 6 / 10 N <= 64 on / off the
                      // 64x64-wave-tile split16 kernel, 8 the LDS-strip 3x3 kernel
  int stagger;        // experiments build: half of the first-round workgroups of an expand conv
                      // start this many microseconds late (MILAN_STAGGER)
  long long* prof;    // experiments build: per-phase cycle sums over all workgroups (16 counters)
  int chunk_major;    // experiments build, split16 kernels, k x k convs: k runs (16-channel chunk,
                      // tap) with the taps INNER -- the KH*KW pieces of a pixel are requested in
                      // consecutive k-tiles and hit L2 -- and W is packed to match (ConvW::ws3)
  float* C2;          // EPI_LSTM: new cell state
  float* Cs;          // EPI_LSTM: h' once more in split format, or nullptr
  unsigned* status;   // device status word (MILAN_STATUS_* bits) or nullptr; filled by the
                      // launcher from the calling context (set_status_word)
  int f16;            // fast mode (MILAN_PRECISION_F16): A, W, aux and C hold PLAIN f16 values, 2
                      // bytes per element.  Every K-side quantity above (Cin, K, Kp, K1, pixel /
                      // image strides, ldc, ldaux) is then given in 4-byte units, i.e. HALF the
                      // element count: a row of C f16 channels is addressed exactly like a
                      // split-format row of C / 2 channels, and the ping-pong kernel multiplies
                      // "hi . hi + lo . lo" of that pretended row = one f16 MFMA per 16 real k
  const float* Wt;    // k x k convs, Cin % 32 == 0: the rows of W with k running (32-channel slice, tap,
                      // channel) instead of (tap, channel) -- ConvW::wst.  The ping-pong kernel then
                      // asks for the KH * KW shifted copies of a slice in CONSECUTIVE k-tile pairs:
                      // they hit L2 instead of crossing the fabric once per tap (gemm.hip, TAPI)
  int tap_inner;      // filled by the launcher: this launch runs in that order
  const long long* lse_tgt;  // EPI_LSE: target column of row m at lse_tgt[m * lse_tgt_stride]
  long lse_tgt_stride;
  float* lse_x;              // EPI_LSE: [M] the target column's x
  // Row count known only on the DEVICE (round 6; encoder.hip, MILAN_FUSE_SKIP_EMPTY): when
  // m_live != nullptr the launch is sized for M rows but only the first min(M, *m_live *
  // m_live_mul) hold work -- the kernel reads the word with a scalar load, workgroups whose
  // tiles lie beyond it exit, the tile that straddles it masks its rows as for a ragged M.
  // No host read-back, no hipStreamSynchronize (live_rows() below).
  const int* m_live;
  int m_live_mul;
};

enum OutMode : int { OUT_SCALAR = 0, OUT_VEC4 = 1, OUT_SPLIT8 = 2, OUT_F16 = 3 };

int launch_gemm(const GemmArgs& g, hipStream_t s);
// Status word of the context whose entry point is running on this thread (api.hip sets it
// on entry): every kernel that writes split format ORs MILAN_STATUS_SATURATED into it when
// its clamp to +-65504 was hit (split8_* below).  nullptr = nobody is listening.
void set_status_word(unsigned* word);
unsigned* status_word();
// kernel family of the GEMM-class launch being recorded (milan_profile_read_kernels)
void profile_tag_kernel(int family);
// rows x K fp32 (row stride ld_src) -> split format (row stride ld_dst), x scale
// Zero `bytes` (a multiple of 4, 4-byte aligned) on the stream with a KERNEL.  Not
// hipMemsetAsync: as a node of a captured graph (round 6: whole passes can be captured) the
// runtime's fill replayed a recycled 16-byte pattern from the second replay on (ROCm 7.2; the
// feature rows of empty-mask exemplars and the LM's initial state came back as garbage:
// tools/graph_describe.py, tests/test_gpu_skip_empty.py).
int launch_zero_fill(void* p, size_t bytes, hipStream_t s);
int launch_f32_to_split(const float* src, long ld_src, float* dst, long ld_dst,
                        long rows, int K, float scale, hipStream_t s);
int gemm_profile_enable(int enable);
// per-device launch state (gemm.hip): dynamic-LDS opt-in of a kernel, CU count in octets
int ensure_lds_attr(const void* kern, int bytes);
int device_cus8(int* out);
void* gemm_profile_begin(double flops, double bytes, hipStream_t s);
void gemm_profile_end(void* rec, hipStream_t s);

// ---- fused expand -> reduce chain (chain.hip) --------------------------------
// One launch = a bottleneck's 1x1 expand conv c3 (+ residual + ReLU) AND the next
// bottleneck's 1x1 reduce conv c1 (+ ReLU): X = relu(T2 W3^T + b3 + R),
// T1 = relu(X W1^T + b1); all tensors split-format, dense rows.  P = planes.
// Optional second expand source (a bottleneck with a stride-1 downsample branch,
// layer1.0): X = relu([T2 | A2] W3^T + b3) with W3 = [c3 | downsample] along K, no
// residual (R == nullptr).
struct ChainArgs {
  const float* T2;     // [M][P]
  const float* W3;     // [4P][P] split weights (scaled), K contiguous
  const float* bias3;  // [4P]
  const float* R;      // [M][4P] residual
  float* X;            // [M][4P]
  const float* W1;     // [NR][4P] split weights (scaled)
  const float* bias1;  // [NR]
  float* T1;           // [M][NR]
  int M, P;
  const float* A2;     // [M][KD] or nullptr
  int KD;              // channels of A2 (0 without)
  float scale3, scale1;  // 1 / weight scale of W3, W1
  long long* prof;       // timing experiments: per-workgroup phase cycles (8 per WG)
  int NR;                // output channels of the reduce conv (0 = P; 2 P at a stage boundary)
  unsigned* status;      // filled by the launcher (status_word())
  const int* m_live;     // device-side row count, see GemmArgs::m_live (nullptr: M rows)
  int m_live_mul;
  // Round 6, the bottleneck's 3x3 conv c2 IN FRONT of the chain (chain_kernel<.., CONV>; planes
  // 64, stride 1, pad 1): when C2in != nullptr the t2 operand is not read from T2 but computed
  // in the kernel from c2's input C2in [M][64] (pixels (img, y, x) of ch x cw images, split
  // format) -- t2 = relu(conv3x3(C2in; W2) * scale2 + bias2) never exists in memory.
  const float* C2in;
  const float* W2;       // [64][576] split weights (scaled), k = tap * 64 + channel
  const float* bias2;    // [64]
  float scale2;          // 1 / weight scale of W2
  int ch, cw;            // image height / width of C2in (cw <= 56)
  // ... and the block's OWN 1x1 reduce conv c1 in front of that (chain_kernel<.., CONV, C1>;
  // layer1.0, whose input has 64 channels): when W0 != nullptr, C2in is the BLOCK INPUT x
  // [M][64] and t1 = relu(x W0^T * scale0 + bias0) is computed in place in the LDS copy of
  // the input region -- neither t1 nor t2 exists in memory (the whole bottleneck in one launch).
  const float* W0;       // [64][64] split weights (scaled)
  const float* bias0;    // [64]
  float scale0;
};
bool chain_supported(int P, int KD, int NR = 0);
bool chain_conv_supported(int P, int KD, int NR, int h, int w);
bool chain_conv_c1_supported(int P, int KD, int NR, int h, int w);
int launch_chain(const ChainArgs& a, hipStream_t s);
// ---- fused split-f16 stem (stem.hip): conv1 7x7/2 + bn1 + ReLU + maxpool 3x3/2 ----
struct StemArgs {
  const float* in;      // pixel-pair groups [n][H][G][hi x8 | lo x8] (encoder.hip)
  const float* ws;      // split weights [64][224] (pack_stem_pairs), scaled
  const float* bias;    // [64] or nullptr
  float acc_scale;      // 1 / weight scale
  const float* scale;   // bn1 folded: y = x * scale + shift
  const float* shift;
  float* raw;           // [n][h1][w1][64] fp32 conv1 output (pyramid level 0) or nullptr
  float* y;             // [n][hp][wp][64] split format
  const int* bbox;      // [n][4] = y0, y1, x0, x1 (inclusive) of the conv1 pixels the
                        // level-0 pooling reads, or nullptr = write every raw pixel
  const float* zero;    // >= 64 B of zeros
  int n, H, G, h1, w1, hp, wp;
  int tiles_y, tiles_x;  // filled by the launcher
  int debug;             // timing experiments: 1 no MFMA, 2 no store phase, 4 no DMA, 8 no staging
  unsigned* status;      // filled by the launcher (status_word())
  // uint8 images straight into the stem (round 5): when in_u8 != nullptr the kernel reads the
  // NCHW bytes of its input tile itself and turns them into pixel-pair groups through `lut`
  // (3 x 256 + 1 entries: hi | lo << 16 of the normalised value of byte v in channel c at
  // c * 256 + v, entry 768 = 0 for the padding) -- bit for bit what preprocess_pairs_kernel
  // writes, without the 32-bytes-per-pixel-pair tensor `in` (unused then)
  const unsigned char* in_u8;
  const unsigned* lut;
  const int* order;      // batch slot -> image number (encoder.hip, MILAN_FUSE_SKIP_EMPTY) or nullptr
  int W;
  const int* n_live;     // device-side image count (<= n) or nullptr, see GemmArgs::m_live
};
bool stem_fused_supported(int cout, int Kp);
int launch_stem_fused(const StemArgs& a, hipStream_t s);
// ---- 3x3 / 1 / 1 convolution, 64 -> 64 channels, weights in registers (conv3.hip) ----
struct Conv3Args {
  const float* in;     // [n][h][w][64] split format
  const float* ws;     // split weights [64][576] (k = tap * 64 + channel), scaled
  const float* bias;   // [64] or nullptr
  float acc_scale;     // 1 / weight scale
  float* out;          // [n][h][w][64] split format: relu(conv + bias)
  const float* zero;   // >= 64 B of zeros
  int n, h, w;
  int tiles_y, tiles_x;  // filled by the launcher
  int debug;             // timing experiments: 1 no MFMA, 2 no epilogue, 4 no DMA, 8 no hand-over
  long long* prof;       // experiments build: per-phase cycle sums of workgroup 0 (16 values)
  unsigned* status;      // filled by the launcher (status_word())
  const int* n_live;     // device-side image count (<= n) or nullptr, see GemmArgs::m_live
};
bool conv3_p64_supported(int cin, int cout, int kh, int kw, int stride, int pad);
int launch_conv3_p64(const Conv3Args& a, hipStream_t s);
bool gemm_profile_active();
int gemm_profile_read(double* ms, double* flops, long long* launches);
int profile_read_stages(double* table /* [MILAN_STAGE_COUNT][6] */);
int profile_read_kernels(double* table /* [MILAN_KERNEL_COUNT][4] */);
// RAII bracket of one stage region on a stream: labels the GEMM launches made
// inside it and, while profiling is on, times the region with two HIP events.
class StageScope {
 public:
  StageScope(int stage, hipStream_t s);
  ~StageScope();
  StageScope(const StageScope&) = delete;
  StageScope& operator=(const StageScope&) = delete;
 private:
  hipStream_t stream_;
  int prev_;
  long idx_;
};

inline GemmArgs linear_args(const float* A, long lda, const float* W,
                            const float* bias, float* C, int ldc, int M, int N,
                            int K, int epi, const float* zero,
                            const float* aux = nullptr, int ldaux = 0) {
  GemmArgs g{};
  g.A = A; g.W = W; g.bias = bias; g.aux = aux; g.C = C;
  g.M = M; g.N = N; g.K = K; g.Kp = (K + 31) / 32 * 32;
  g.ldc = ldc; g.ldaux = ldaux;
  g.H = 1; g.Wd = 1; g.Cin = K; g.Ho = 1; g.Wo = 1; g.KH = 1; g.KW = 1;
  g.stride = 1; g.pad = 0;
  g.a_pix_stride = lda; g.a_img_stride = lda;
  g.epilogue = epi; g.zero = zero;
  return g;
}

// torch's LSTM cell, gate order i, f, g, o (one definition for the pointwise
// kernel and the fused GEMM epilogue, roundings spelled out, so that both give
// the same bits)
__device__ __forceinline__ void lstm_cell(float pi, float pf, float pg, float po,
                                          float c_in, float* h, float* c) {
  const float gi = 1.f / (1.f + expf(-pi));
  const float gf = 1.f / (1.f + expf(-pf));
  const float gg = tanhf(pg);
  const float go = 1.f / (1.f + expf(-po));
  const float c2 = __fmaf_rn(gi, gg, __fmul_rn(gf, c_in));
  *c = c2;
  *h = __fmul_rn(go, tanhf(c2));
}
// row of gate `gate` (0..3 = i, f, g, o), hidden unit u, in the gate-interleaved
// order EPI_LSTM expects: 64-row groups [i x16 | f x16 | g x16 | o x16]
__host__ __device__ inline int lstm_interleaved_row(int gate, int u) {
  return (u >> 4) * 64 + gate * 16 + (u & 15);
}


// ---- packed weights ----------------------------------------------------------
struct ConvW {
  float* w = nullptr;     // [Cout][Kp]
  float* ws = nullptr;    // same, split-f16 format, scaled by 1/ws_inv
  float* ws3 = nullptr;   // 3x3 only, experiments build: ws re-ordered chunk-major (gemm.hip)
  float ws_inv = 1.f;     // exact power of two
  float* bias = nullptr;  // folded BN shift, [Cout] (nullptr for raw stem)
  float* bias_s = nullptr;  // bias x activation scale (split-f16 trunk, see milan_ctx::act_scale)
  float* wf = nullptr;      // fast mode: the hi halves of `ws` as plain f16 rows [Cout][Kp] (2 B each)
  float* wst = nullptr;     // k x k convs: `ws` with k in (32-channel slice, tap, channel) order (GemmArgs::Wt)
  float* wft = nullptr;     // ... `wf` in (64-channel slice, tap, channel) order
  int cout = 0, cin = 0, kh = 0, kw = 0, stride = 1, pad = 0, K = 0, Kp = 0;
  int cin_real = 0;  // channels before padding to a multiple of 4
};
struct Bottleneck {
  ConvW c1, c2, c3, down;
  ConvW c3d;  // [c3 | downsample] concatenated along K (split mode fusion)
  bool has_down = false;
  bool basic = false;  // torchvision BasicBlock: c1 3x3 (stride), c2 3x3, no c3
};
struct LinearW {
  float* w = nullptr;  // [N][Kp]
  float* ws = nullptr; // split-f16 copy (nullptr when K % 32 != 0)
  float ws_inv = 1.f;
  float* b = nullptr;
  int n = 0, k = 0, kp = 0;
  bool gate_interleaved = false;  // rows permuted for EPI_LSTM (cat weights)
};

struct Tensor {
  std::vector<int64_t> shape;
  const float* dev = nullptr;  // caller-owned device pointer (until finalize)
  int64_t numel() const {
    int64_t n = 1;
    for (auto d : shape) n *= d;
    return n;
  }
};

// bump allocator over a caller-provided workspace
struct Arena {
  char* base = nullptr;
  size_t size = 0, off = 0;
  bool dry = false;  // dry run: only count
  template <typename T>
  T* get(size_t n) {
    size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
    size_t o = off;
    off += bytes;
    if (dry) return nullptr;
    if (off > size) return nullptr;
    return reinterpret_cast<T*>(base + o);
  }
};

// ---- device-side row / image counts (GemmArgs::m_live) --------------------------------
// min(bound, *live * mul) through a SCALAR load (constant address space: the word was
// written by an earlier kernel of the same stream and does not change while this one runs;
// the scalar cache is invalidated at every dispatch, as for the kernel arguments).
__device__ __forceinline__ int live_rows(const int* live, int mul, int bound) {
  if (live == nullptr) return bound;
  typedef const int __attribute__((address_space(4)))* KWord;
  const long v = (long)*(KWord)(uintptr_t)live * mul;
  return v < bound ? (int)v : bound;
}

// ---- split-f16 storage format, device side (one implementation for every kernel) -----
// 8 fp32 -> (hi, lo) f16x8 pair; hi saturates instead of overflowing to inf:
//   x = min(max(v, -65504), 65504);  hi = f16(x);  lo = f16(x - float(hi))
// spelled with the instructions that do two of these steps at once -- v_med3_f32 (the
// clamp), v_cvt_pk_f16_f32 (two roundings to f16, RNE), v_fma_mix_f32 (x - float(hi): the
// f16 -> f32 conversion inside it is exact, so the one rounding is the subtraction's) --
// 20 instructions per 8 values instead of 48, the same bits (tests/test_gpu_chain.py and
// the token hash of the bench workload pin that).  V4 = any 4 x 32-bit vector type.
__device__ __forceinline__ float cvt_pk_f16(float a, float b) {
  float r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int SEL>   // x - float(f16 half SEL of hpair)
__device__ __forceinline__ float mix_sub_f16(float x, float hpair) {
  float r;
  if constexpr (SEL == 0)
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(x));
  else
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(x));
  return r;
}
template <int SEL>   // float(f16 half SEL of hpair) + float(f16 half SEL of lpair)
__device__ __forceinline__ float mix_add_f16(float hpair, float lpair) {
  float r;
  if constexpr (SEL == 0)
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hpair), "v"(lpair));
  else
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hpair), "v"(lpair));
  return r;
}
// Saturation is LOUD (round 5): every thread keeps the running maximum `sat` of the clamped
// magnitudes it has split (four v_max3_f32 per 8 values, no compare, no scalar traffic);
// `sat >= 65504` at the end of the kernel means a value hit the clamp (or was exactly the
// largest f16) and the kernel ORs MILAN_STATUS_SATURATED into the context's status word
// (report_saturation).  The host raises / falls back to f32 (milan_amd/hip.py).
// NaN: v_med3_f32 returns min3 when an operand is NaN, so a NaN accumulator leaves as 0 (ReLU
// form) or -65504 (plain form).  A NaN cannot ARISE inside the trunk without a saturation
// first (|x| > 65504 / act_scale is flagged long before fp32 overflows to Inf - Inf), and a
// non-finite INPUT pixel poisons its whole image in the reference (every pyramid level pools
// NaN x mask) -- that is reproduced at image level: preprocess_* marks the image, the pooling
// / spatial read-out write NaN for it (encoder.hip, MILAN_STATUS_NONFINITE_INPUT).
__device__ __forceinline__ float max3_abs(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float sat_fold8(const float* x, float sat) {
  return max3_abs(max3_abs(x[0], x[1], x[2]), max3_abs(x[3], x[4], x[5]), max3_abs(x[6], x[7], sat));
}
__device__ __forceinline__ void report_saturation(unsigned* status, float sat) {
  if (status != nullptr && sat >= 65504.f) atomicOr(status, 1u /* MILAN_STATUS_SATURATED */);
}
template <typename V4>
__device__ __forceinline__ void split8_rne(const float* v, V4* hi_out, V4* lo_out, float* sat = nullptr) {
  V4 hi, lo;
  float x[8];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(x[2 * d]) : "v"(v[2 * d]), "v"(-65504.f), "v"(65504.f));
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(x[2 * d + 1]) : "v"(v[2 * d + 1]), "v"(-65504.f), "v"(65504.f));
    hi[d] = cvt_pk_f16(x[2 * d], x[2 * d + 1]);
    lo[d] = cvt_pk_f16(mix_sub_f16<0>(x[2 * d], hi[d]), mix_sub_f16<1>(x[2 * d + 1], hi[d]));
  }
  if (sat) *sat = sat_fold8(x, *sat);
  *hi_out = hi;
  *lo_out = lo;
}
// relu(v) and the clamp in one v_med3_f32: med3(v, 0, 65504) == min(max(max(v, 0), -65504),
// 65504) for every finite input (-0 -> +0 like v_max_f32; NaN -> 0, see above)
template <typename V4>
__device__ __forceinline__ void split8_relu_rne(const float* v, V4* hi_out, V4* lo_out, float* sat = nullptr) {
  V4 hi, lo;
  float x[8];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    asm("v_med3_f32 %0, %1, 0, %2" : "=v"(x[2 * d]) : "v"(v[2 * d]), "v"(65504.f));
    asm("v_med3_f32 %0, %1, 0, %2" : "=v"(x[2 * d + 1]) : "v"(v[2 * d + 1]), "v"(65504.f));
    hi[d] = cvt_pk_f16(x[2 * d], x[2 * d + 1]);
    lo[d] = cvt_pk_f16(mix_sub_f16<0>(x[2 * d], hi[d]), mix_sub_f16<1>(x[2 * d + 1], hi[d]));
  }
  if (sat) *sat = sat_fold8(x, *sat);
  *hi_out = hi;
  *lo_out = lo;
}
template <typename V4>   // (hi, lo) -> 8 fp32, v[e] = float(hi[e]) + float(lo[e])
__device__ __forceinline__ void join8_exact(V4 hi, V4 lo, float* v) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    v[2 * d] = mix_add_f16<0>(hi[d], lo[d]);
    v[2 * d + 1] = mix_add_f16<1>(hi[d], lo[d]);
  }
}

}  // namespace milan

struct milan_ctx {
  int device = 0;
  milan_dims d{};
  bool finalized = false;
  int precision = 0;  // MILAN_PRECISION_F32 / MILAN_PRECISION_SPLIT_F16 (also in fast mode:
                      // decoder, LM and the front of the trunk stay split)
  int trunk_f16 = 0;  // MILAN_PRECISION_F16: layer3 / layer4 of a bottleneck trunk on plain f16
  int fusion = MILAN_FUSE_CHAIN | MILAN_FUSE_CHAIN_WIDE | MILAN_FUSE_STEM | MILAN_FUSE_CONV3 |
               MILAN_FUSE_SKIP_EMPTY | MILAN_FUSE_BNECK | MILAN_FUSE_SPARSE_TAIL;  // milan_set_fusion
  // hipGraph cache of whole decode passes (milan_set_graph_capture)
  int graph_capture = 0;
  struct GraphEntry { std::vector<char> key; hipGraphExec_t exec = nullptr; int seen = 0; };
  std::vector<GraphEntry> graphs;
  long graph_replays = 0, graph_captures = 0;
  float* scratch = nullptr;   // per-call split-conversion scratch (workspace)
  size_t scratch_floats = 0;
  std::map<std::string, milan::Tensor> raw;  // named reference tensors
  // weight arena (library-owned, freed in milan_destroy)
  std::vector<void*> owned;
  float* zero = nullptr;
  // encoder
  milan::ConvW stem;
  milan::ConvW stem_pair;  // split-f16 stem over pixel-pair groups (encoder.hip)
  milan::ConvW alex[5];    // 'alexnet' config: features.0/3/6/8/10
  float *bn1_scale = nullptr, *bn1_shift = nullptr;
  // Split-f16 trunk: every activation tensor in split format is stored as x * act_scale
  // (an exact power of two; MILAN_ACT_SCALE_LOG2, default 5).  `lo = f16(x - f16(x))`
  // leaves the f16 normal range for |x| < 0.125, and its absolute floor (2^-24) then
  // costs accuracy relative to x; stored 32 x larger, the same happens only below 0.004,
  // while `hi` saturates at 2047 instead of 65504.  Scaling by a power of two commutes
  // with every fp32 operation of the epilogues, so the only places that know about it
  // are: the stem's folded bn1 (scale, shift) and every conv's folded bias (pre-multiplied
  // copies below / ConvW::bias_s), and the consumers that leave the split domain (the
  // mask-weighted pooling and the spatial read-out divide by it).
  float act_scale = 32.f;
  int act_scale_log2 = 5;
  float *bn1_scale_s = nullptr, *bn1_shift_s = nullptr;
  unsigned* stem_lut = nullptr;  // byte -> split value table of the uint8 stem (StemArgs::lut), lazily
  // device status word (MILAN_STATUS_* bits, milan_status): ORed by the split epilogues when
  // a value hit the +-65504 clamp and by the input conversion when a pixel was not finite
  unsigned* status = nullptr;
  unsigned* calib = nullptr;   // != nullptr only inside milan_encoder_absmax
  std::vector<milan::Bottleneck> blocks[4];
  float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  // decoder
  milan::LinearW init_h, init_c, q2h, k2h, gate, lstm_ih, lstm_hh, out;
  milan::LinearW lstm_cat;                   // [W_ih | W_hh] along K (split mode)
  float *att_w = nullptr, *att_b = nullptr;  // attend.output.0: [A], [1]
  float* embedding = nullptr;                // [V][E]
  // language model
  std::vector<milan::LinearW> lm_ih, lm_hh;  // per layer
  std::vector<milan::LinearW> lm_cat;        // [W_ih | W_hh] per layer
  milan::LinearW lm_out;
  float* lm_embedding = nullptr;
};

namespace milan {
int dev_alloc(milan_ctx* c, void** p, size_t bytes);
// api.hip: split-f16 copy of a packed [n][kp] fp32 weight; picks a power-of-two
// scale from max|w| (host sync) and returns its inverse.
int make_split_weight(milan_ctx* c, const float* w, int n, int kp, float** ws,
                      float* ws_inv, hipStream_t s);
// encoder.hip
int pack_conv_plain(const float* w_oihw, int cout, int cin, int kh, int kw,
                    float* wp, hipStream_t s);
int encoder_finalize(milan_ctx* c, hipStream_t s);
int encoder_rescale(milan_ctx* c, hipStream_t s);
// split 3x3 weights [n][tap][Cin] -> chunk-major [n][Cin/16][tap][16]
// packed conv rows [cout][taps][cin] -> [cout][cin / (8 gps)][taps][8 gps] in 8-channel groups of
// `gb` 16-byte pieces (2: split format, 1: plain f16) -- GemmArgs::Wt (gps 4 / 8)
int make_slice_major(const float* ws, int cout, int taps, int cin, int gps, int gb, float* dst,
                     hipStream_t s);
int make_chunk_major(const float* ws, int cout, int cin, float* dst,
                     hipStream_t s);
size_t encoder_workspace(const milan_ctx* c, int n_images, int H, int W);
int encoder_run_spatial(milan_ctx* c, const void* images, int image_dtype,
                        const void* masks, int mask_dtype, int n, int H, int W,
                        float* out, milan::Arena& ws, hipStream_t s);
int encoder_run(milan_ctx* c, const void* images, int image_dtype,
                const void* masks, int mask_dtype, int n_images, int H, int W,
                float* features, Arena& ws, hipStream_t s);
// decoder.hip
int decoder_finalize(milan_ctx* c, hipStream_t s);
size_t decoder_workspace(const milan_ctx* c, int n, int k, int beam, int length);
int decoder_init_state(milan_ctx* c, const float* features, int n, int k,
                       float* h, float* cc, Arena& ws, hipStream_t s);
int decoder_step(milan_ctx* c, const float* features, int rows, int k,
                 const int64_t* tokens, const float* h, const float* cc,
                 float* h_lm, float* c_lm, float temperature,
                 float* predictions, float* attentions, float* h_out,
                 float* c_out, Arena& ws, hipStream_t s);
int decoder_decode(milan_ctx* c, const float* features, int n, int k,
                   int strategy, int length, int beam, int mi,
                   float temperature, int group_size, int64_t* tokens,
                   float* scores, float* predictions, float* attentions,
                   int64_t* beam_tokens, float* beam_scores, int32_t* out_len,
                   Arena& ws, hipStream_t s);
int decoder_lm_logprobs(milan_ctx* c, const int64_t* seqs, int rows, int L,
                        float* out, milan::Arena& ws, hipStream_t s);
int decoder_lm_score(milan_ctx* c, const int64_t* seqs, int rows, int L,
                     const int32_t* seq_len, float* out, Arena& ws,
                     hipStream_t s);
}  // namespace milan
