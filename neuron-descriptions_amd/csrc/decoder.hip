// Attention-LSTM caption decoder, beam search and LM (PMI) rerank.
//
// Mirrors src/milan/decoders.py (init_state :548-574, Attention.forward
// :57-73, step :576-634, greedy loop :431-463, allennlp BeamSearch call
// :467-484, rerank epilogue :495-512) and src/milan/lms.py:58-101.
//
// Work split: every dense product goes through the fp32-MFMA GEMM of
// gemm.hip; what is here is the glue that is HBM/latency bound: the k-way
// additive attention (wave reductions), the context sum, LSTM cell
// pointwise, per-row log-softmax + top-k, the beam merge, index-only beam
// reordering (the (rows,k,F) features never move: a row reads its neuron's
// features through row / rows_per_neuron) and the LM gather/mask sum.
//
// The key projection W_k f_k (decoders.py:71) is hoisted: it depends on the
// neuron only, so it is computed once per neuron instead of once per step per
// beam row (68% of the reference decoder's FLOPs).
#include "common.h"
#include <limits>
#include <optional>

namespace milan {

static constexpr float kFloatMin = -3.402823466e+38f;  // torch.finfo(f32).min

static const Tensor* find(milan_ctx* c, const std::string& name) {
  auto it = c->raw.find(name);
  return it == c->raw.end() ? nullptr : &it->second;
}

// ---------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------
__global__ void pack_rows_kernel(const float* __restrict__ src, int n, int k,
                                 int kp, float* __restrict__ dst) {
  const long total = (long)n * kp;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int r = idx / kp, col = idx - (long)r * kp;
    dst[idx] = col < k ? src[(long)r * k + col] : 0.f;
  }
}

// hidden > 0: output row i holds source row (gate, unit) of the gate-interleaved
// order (lstm_interleaved_row); 0: identity
__device__ inline int lstm_source_row(int i, int hidden) {
  if (!hidden) return i;
  const int gate = (i & 63) >> 4, u = (i >> 6) * 16 + (i & 15);
  return gate * hidden + u;
}
__global__ void add_vec_kernel(const float* a, const float* b, int n,
                               float* out, int hidden = 0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int r = lstm_source_row(i, hidden);
    out[i] = a[r] + (b ? b[r] : 0.f);
  }
}

static int copy_vec(milan_ctx* c, const Tensor* t, float** out, hipStream_t s) {
  MILAN_TRY(dev_alloc(c, (void**)out, sizeof(float) * t->numel()));
  MILAN_CHECK_HIP(hipMemcpyAsync(*out, t->dev, sizeof(float) * t->numel(),
                                 hipMemcpyDeviceToDevice, s));
  return 0;
}

static int pack_linear(milan_ctx* c, const std::string& wname,
                       const std::string& bname, int n, int k, LinearW* out,
                       hipStream_t s) {
  const Tensor* w = find(c, wname);
  MILAN_REQUIRE(w, MILAN_ERR_STATE, "missing weight %s", wname.c_str());
  MILAN_REQUIRE(w->shape.size() == 2 && w->shape[0] == n && w->shape[1] == k,
                MILAN_ERR_SHAPE, "%s: expected (%d,%d), got (%lld,%lld)",
                wname.c_str(), n, k, (long long)w->shape[0],
                (long long)(w->shape.size() > 1 ? w->shape[1] : -1));
  MILAN_REQUIRE(k % 4 == 0, MILAN_ERR_SHAPE,
                "%s: inner dimension %d must be a multiple of 4", wname.c_str(),
                k);
  out->n = n; out->k = k; out->kp = (k + 31) / 32 * 32;
  MILAN_TRY(dev_alloc(c, (void**)&out->w, sizeof(float) * (size_t)n * out->kp));
  const long total = (long)n * out->kp;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_rows_kernel, dim3(blocks), dim3(256), 0, s, w->dev, n,
                     k, out->kp, out->w);
  MILAN_CHECK_HIP(hipGetLastError());
  if (k % 32 == 0)
    MILAN_TRY(make_split_weight(c, out->w, n, out->kp, &out->ws, &out->ws_inv, s));
  if (!bname.empty()) {
    const Tensor* b = find(c, bname);
    MILAN_REQUIRE(b && b->numel() == n, MILAN_ERR_STATE, "missing/bad bias %s",
                  bname.c_str());
    MILAN_TRY(copy_vec(c, b, &out->b, s));
  }
  return 0;
}

// [W_a | W_b] along K, bias b_a + b_b: an LSTM's two gate products
// x W_ih^T + h W_hh^T as ONE GEMM over the concatenated operand [x | h]
// (split-f16 mode, where both operands are converted into one scratch matrix
// anyway).  Leaves out->ws null for shapes the split path does not cover.
__global__ void cat_rows_kernel(const float* __restrict__ a, int ka,
                                const float* __restrict__ b, int kb, int n,
                                float* __restrict__ out, int hidden) {
  const int kt = ka + kb;
  const long total = (long)n * kt;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int ro = idx / kt, col = idx - (long)ro * kt;
    const int r = lstm_source_row(ro, hidden);
    out[idx] = col < ka ? a[(long)r * ka + col] : b[(long)r * kb + (col - ka)];
  }
}
// LSTM weights (n = 4 H, H % 16 == 0): rows are stored gate-interleaved per 16
// hidden units so that the GEMM epilogue sees the four gates of a unit inside
// one wave tile and applies the cell itself (EPI_LSTM) -- the (rows, 4H)
// pre-activation matrix is never written.
static int cat_linear(milan_ctx* c, const LinearW& a, const LinearW& b,
                      LinearW* out, hipStream_t s) {
  *out = LinearW();
  if (a.n != b.n || a.k % 32 || b.k % 32 || a.kp != a.k || b.kp != b.k ||
      !a.b || !b.b)
    return 0;
  out->n = a.n; out->k = out->kp = a.k + b.k;
  MILAN_TRY(dev_alloc(c, (void**)&out->w, sizeof(float) * (size_t)out->n * out->kp));
  MILAN_TRY(dev_alloc(c, (void**)&out->b, sizeof(float) * out->n));
  const long total = (long)out->n * out->kp;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  // MILAN_LSTM_FUSE=0 (read when the weights are packed) keeps the plain row
  // order and the separate pointwise kernel, for A/B timing
  static const bool fuse = [] {
    const char* e = getenv("MILAN_LSTM_FUSE");
    return !e || atoi(e) != 0;
  }();
  out->gate_interleaved = fuse && out->n % 64 == 0;
  const int hidden = out->gate_interleaved ? out->n / 4 : 0;
  hipLaunchKernelGGL(cat_rows_kernel, dim3(blocks), dim3(256), 0, s, a.w, a.k, b.w,
                     b.k, out->n, out->w, hidden);
  hipLaunchKernelGGL(add_vec_kernel, dim3((out->n + 255) / 256), dim3(256), 0, s,
                     a.b, b.b, out->n, out->b, hidden);
  MILAN_CHECK_HIP(hipGetLastError());
  return make_split_weight(c, out->w, out->n, out->kp, &out->ws, &out->ws_inv, s);
}

static int lm_finalize(milan_ctx* c, hipStream_t s);

int decoder_finalize(milan_ctx* c, hipStream_t s) {
  // encoder-only context, or a standalone LanguageModel (lm.* weights only)
  if (!find(c, "lstm.weight_ih")) return lm_finalize(c, s);
  const milan_dims& d = c->d;
  const int F = d.feature_size, H = d.hidden_size, E = d.embedding_size,
            A = d.attention_size, V = d.vocab_size;
  MILAN_TRY(pack_linear(c, "init_h.0.weight", "init_h.0.bias", H, F, &c->init_h, s));
  MILAN_TRY(pack_linear(c, "init_c.0.weight", "init_c.0.bias", H, F, &c->init_c, s));
  MILAN_TRY(pack_linear(c, "attend.query_to_hidden.weight",
                        "attend.query_to_hidden.bias", A, H, &c->q2h, s));
  MILAN_TRY(pack_linear(c, "attend.key_to_hidden.weight",
                        "attend.key_to_hidden.bias", A, F, &c->k2h, s));
  MILAN_TRY(pack_linear(c, "feature_gate.0.weight", "feature_gate.0.bias", F, H,
                        &c->gate, s));
  MILAN_TRY(pack_linear(c, "lstm.weight_ih", "lstm.bias_ih", 4 * H, E + F,
                        &c->lstm_ih, s));
  MILAN_TRY(pack_linear(c, "lstm.weight_hh", "lstm.bias_hh", 4 * H, H,
                        &c->lstm_hh, s));
  MILAN_TRY(pack_linear(c, "output.1.weight", "output.1.bias", V, H, &c->out, s));
  MILAN_TRY(cat_linear(c, c->lstm_ih, c->lstm_hh, &c->lstm_cat, s));
  {
    const Tensor *w = find(c, "attend.output.0.weight"),
                 *b = find(c, "attend.output.0.bias"),
                 *e = find(c, "embedding.weight");
    MILAN_REQUIRE(w && b && e, MILAN_ERR_STATE,
                  "missing attend.output.0.* or embedding.weight");
    MILAN_REQUIRE(w->numel() == A && e->numel() == (int64_t)V * E,
                  MILAN_ERR_SHAPE, "attend.output / embedding shape mismatch");
    MILAN_TRY(copy_vec(c, w, &c->att_w, s));
    MILAN_TRY(copy_vec(c, b, &c->att_b, s));
    MILAN_TRY(copy_vec(c, e, &c->embedding, s));
  }
  return lm_finalize(c, s);
}

// LanguageModel weights (src/milan/lms.py:47-56): present in a Decoder with an
// LM, or on their own for a standalone LanguageModel.
static int lm_finalize(milan_ctx* c, hipStream_t s) {
  const milan_dims& d = c->d;
  const int V = d.vocab_size;
  if (!d.has_lm || !find(c, "lm.lstm.weight_ih_l0")) return 0;
  {
    const int Hl = d.lm_hidden_size, El = d.lm_embedding_size;
    c->lm_ih.resize(d.lm_layers);
    c->lm_hh.resize(d.lm_layers);
    c->lm_cat.resize(d.lm_layers);
    for (int l = 0; l < d.lm_layers; ++l) {
      const std::string sfx = "_l" + std::to_string(l);
      MILAN_TRY(pack_linear(c, "lm.lstm.weight_ih" + sfx, "lm.lstm.bias_ih" + sfx,
                            4 * Hl, l == 0 ? El : Hl, &c->lm_ih[l], s));
      MILAN_TRY(pack_linear(c, "lm.lstm.weight_hh" + sfx, "lm.lstm.bias_hh" + sfx,
                            4 * Hl, Hl, &c->lm_hh[l], s));
      MILAN_TRY(cat_linear(c, c->lm_ih[l], c->lm_hh[l], &c->lm_cat[l], s));
    }
    MILAN_TRY(pack_linear(c, "lm.output.0.weight", "lm.output.0.bias", V, Hl,
                          &c->lm_out, s));
    const Tensor* e = find(c, "lm.embedding.weight");
    MILAN_REQUIRE(e && e->numel() == (int64_t)V * El, MILAN_ERR_STATE,
                  "missing/bad lm.embedding.weight");
    MILAN_TRY(copy_vec(c, e, &c->lm_embedding, s));
  }
  return 0;
}

// ---------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------
// pooled[n][f] = mean_k features[n][k][f]      (decoders.py:564)
__global__ void mean_k_kernel(const float* __restrict__ feat, int n, int k,
                              int F, float* __restrict__ pooled) {
  const long total = (long)n * F;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long i = idx / F;
    const int f = idx - i * F;
    float s = 0.f;
    for (int j = 0; j < k; ++j) s += feat[(i * k + j) * F + f];
    pooled[idx] = s / (float)k;
  }
}

// att[r][:] = softmax_k( w . tanh(q[r] + keys[neuron][k]) + b )
// (decoders.py:70-73).  One wave per row; neuron = r / rpn.
__global__ __launch_bounds__(256) void attend_kernel(
    const float* __restrict__ q, const float* __restrict__ keys,
    const float* __restrict__ w, const float* __restrict__ b, int rows, int rpn,
    int k, int A, float* __restrict__ att) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* qr = q + (long)r * A;
  const float* kr = keys + (long)(r / rpn) * k * A;
  float mine = 0.f;  // lane j keeps score j
  float mx = -INFINITY;
  for (int j = 0; j < k; ++j) {
    float s = 0.f;
    for (int a = lane; a < A; a += 64) s += w[a] * tanhf(qr[a] + kr[(long)j * A + a]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    s += b[0];
    if (lane == (j & 63)) mine = s;  // k <= 64 enforced by the host
    mx = fmaxf(mx, s);
  }
  float e = lane < k ? expf(mine - mx) : 0.f;
  float sum = e;
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane < k) att[(long)r * k + lane] = e / sum;
}

// The same for k <= 16 exemplars and A <= 512 attention units (MILAN: 15, 512), one
// workgroup per NEURON: a wave keeps the neuron's k x A keys in registers (120 per lane) and
// walks every fourth of its rpn beam rows, so a row costs its 2 KB of q instead of 32 KB of
// keys out of L2.  The launch of every decode step (32 000 rows x 15 x 512 tanh at beam 50 x
// 640 neurons) was bound by that and by the ~30 VALU instructions of libm's tanhf:
// tanh(x) = 1 - 2 / (e^{2x} + 1) on the hardware's exp2 / rcp (1 ulp each) has an absolute
// error <= 2e-7, the size of one fp32 rounding of the 512-term sum it feeds.  The k wave sums
// are reduced together, halving the live values per butterfly step (25 cross-lane moves per
// row instead of 102); lane L ends with the score of exemplar L / 4.
constexpr int kAttMaxK = 16, kAttMaxE = 8;
__device__ __forceinline__ float tanh_exp2(float x) {
  const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // e^{2x}
  return 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
}
__global__ __launch_bounds__(256) void attend16_kernel(
    const float* __restrict__ q, const float* __restrict__ keys,
    const float* __restrict__ w, const float* __restrict__ b, int rows, int rpn,
    int k, int A, float* __restrict__ att) {
  const int neuron = blockIdx.x;
  // (blockIdx.y, wave): which of the neuron's rows, stride 4 gridDim.y
  const int wave = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int stride = 4 * gridDim.y;
  const int r0 = neuron * rpn;
  const int r1 = r0 + rpn < rows ? r0 + rpn : rows;
  if (r0 + wave >= r1) return;
  const float* kr = keys + (long)neuron * k * A;
  const int ne = (A + 63) / 64;
  float wv[kAttMaxE], kv[kAttMaxK][kAttMaxE];
  int av[kAttMaxE];
#pragma unroll
  for (int e = 0; e < kAttMaxE; ++e) {
    const int a = lane + 64 * e;
    const bool ok = a < A;
    av[e] = ok ? a : 0;
    wv[e] = ok ? w[a] : 0.f;  // (a lane past A adds w * tanh = 0 * finite)
  }
#pragma unroll
  for (int j = 0; j < kAttMaxK; ++j)
#pragma unroll
    for (int e = 0; e < kAttMaxE; ++e)
      kv[j][e] = (j < k && e < ne) ? kr[(long)j * A + av[e]] : 0.f;
  const float b0 = b[0];
  for (int r = r0 + wave; r < r1; r += stride) {
    const float* qr = q + (long)r * A;
    float qv[kAttMaxE];
#pragma unroll
    for (int e = 0; e < kAttMaxE; ++e) qv[e] = e < ne ? qr[av[e]] : 0.f;
    float t[kAttMaxK];
#pragma unroll
    for (int j = 0; j < kAttMaxK; ++j) {
      float s = 0.f;
      if (j < k) {
#pragma unroll
        for (int e = 0; e < kAttMaxE; ++e)
          if (e < ne) s += wv[e] * tanh_exp2(qv[e] + kv[j][e]);
      }
      t[j] = s;
    }
    // lanes whose bit X is clear keep the lower half of the live values, the others the upper
#pragma unroll
    for (int n = kAttMaxK, X = 32; n > 1; n >>= 1, X >>= 1) {
      const bool hi = (lane & X) != 0;
#pragma unroll
      for (int i = 0; i < n / 2; ++i) {
        const float recv = __shfl_xor(hi ? t[i] : t[i + n / 2], X);
        t[i] = (hi ? t[i + n / 2] : t[i]) + recv;
      }
    }
    float s = t[0] + __shfl_xor(t[0], 2);
    s += __shfl_xor(s, 1);
    s += b0;
    const int j = lane >> 2;  // (32 -> 8, 16 -> 4, 8 -> 2, 4 -> 1)
    const bool valid = j < k;
    float mx = valid ? s : -INFINITY;
#pragma unroll
    for (int X = 4; X < 64; X <<= 1) mx = fmaxf(mx, __shfl_xor(mx, X));
    const bool writer = valid && (lane & 3) == 0;
    const float ev = writer ? expf(s - mx) : 0.f;
    float sum = ev;
#pragma unroll
    for (int X = 4; X < 64; X <<= 1) sum += __shfl_xor(sum, X);
    if (writer) att[(long)r * k + j] = ev / sum;
  }
}

static void launch_attend(const float* q, const float* keys, const float* w, const float* b,
                          int rows, int rpn, int k, int A, float* att, hipStream_t s) {
  // MILAN_ATTEND_FAST=0: libm tanhf (A/B timing)
  static const bool fast = !(getenv("MILAN_ATTEND_FAST") && atoi(getenv("MILAN_ATTEND_FAST")) == 0);
  if (fast && k <= kAttMaxK && A <= 64 * kAttMaxE) {
    // enough waves to fill the chip (>= 8 per CU), each with as many rows as that leaves
    const int neurons = (rows + rpn - 1) / rpn;
    int rs = (4096 + neurons * 4 - 1) / (neurons * 4);
    const int rs_max = (rpn + 3) / 4;
    rs = rs < 1 ? 1 : (rs > rs_max ? rs_max : rs);
    hipLaunchKernelGGL(attend16_kernel, dim3(neurons, rs), dim3(256), 0, s, q, keys, w, b, rows,
                       rpn, k, A, att);
  } else {
    hipLaunchKernelGGL(attend_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, q, keys, w, b,
                       rows, rpn, k, A, att);
  }
}

// ctx[r][f] = sum_k att[r][k] * features[neuron][k][f]   (decoders.py:613)
// One workgroup per (neuron, 1024-column slice): each thread keeps its float4
// column of the neuron's k feature rows in registers and walks the neuron's rpn
// beam rows, so the features cross L2 once per neuron instead of once per row
// (3 GB -> 60 MB per launch at beam 50).  The k attention weights of a row are
// wave-uniform and read with scalar loads -- no LDS, no barrier (an earlier
// LDS-staged variant was not reproducible under multi-process GPU sharing, see
// DESIGN.md section 6).  Accumulation order per element: k ascending, fmaf.
constexpr int kCtxMaxK = 16;  // register-resident exemplars (k = 15 in MILAN)
template <bool RESIDENT>
__global__ __launch_bounds__(256) void context_kernel(
    const float* __restrict__ att, const float* __restrict__ feat, int rows,
    int rpn, int k, int F, float* __restrict__ ctx) {
  const int neuron = blockIdx.x;
  const int f = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (f >= F) return;
  const float* fr = feat + (long)neuron * k * F + f;
  float4 v[RESIDENT ? kCtxMaxK : 1];
  if constexpr (RESIDENT) {
#pragma unroll
    for (int j = 0; j < kCtxMaxK; ++j)
      v[j] = j < k ? *reinterpret_cast<const float4*>(fr + (long)j * F)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int r0 = neuron * rpn;
  const int r1 = r0 + rpn < rows ? r0 + rpn : rows;
  for (int r = r0; r < r1; ++r) {
    const float* ar = att + (long)r * k;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if constexpr (RESIDENT) {
#pragma unroll
      for (int j = 0; j < kCtxMaxK; ++j) {
        if (j < k) {
          const float aj = ar[j];
          s0 = fmaf(aj, v[j].x, s0); s1 = fmaf(aj, v[j].y, s1);
          s2 = fmaf(aj, v[j].z, s2); s3 = fmaf(aj, v[j].w, s3);
        }
      }
    } else {
      for (int j = 0; j < k; ++j) {
        const float4 w = *reinterpret_cast<const float4*>(fr + (long)j * F);
        const float aj = ar[j];
        s0 = fmaf(aj, w.x, s0); s1 = fmaf(aj, w.y, s1);
        s2 = fmaf(aj, w.z, s2); s3 = fmaf(aj, w.w, s3);
      }
    }
    *reinterpret_cast<float4*>(ctx + (long)r * F + f) = make_float4(s0, s1, s2, s3);
  }
}

// dst[r][0:E] = table[tok[r]]
__global__ void embed_kernel(const float* __restrict__ table,
                             const int64_t* __restrict__ tok, int rows, int E,
                             float* __restrict__ dst, int ldd) {
  const long total = (long)rows * E;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / E;
    const int e = idx - r * E;
    dst[r * ldd + e] = table[tok[r] * E + e];
  }
}

// same, written in split format (groups of 8 channels [hi x8 | lo x8])
__global__ void embed_split_kernel(const float* __restrict__ table,
                                   const int64_t* __restrict__ tok, int rows, int E,
                                   float* __restrict__ dst, int ldd,
                                   const int* __restrict__ live = nullptr) {
  if (live) rows = min(rows, *live);   // (row count on the device: lm_score_dedup)
  const int g8 = E >> 3;
  const long total = (long)rows * g8;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / g8;
    const int q = idx - r * g8;
    const float* sp = table + tok[r] * E + q * 8;
    _Float16 hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = fminf(fmaxf(sp[e], -65504.f), 65504.f);
      hi[e] = (_Float16)x;
      lo[e] = (_Float16)(x - (float)hi[e]);
    }
    float4* d = reinterpret_cast<float4*>(dst + r * ldd + q * 8);
    d[0] = *reinterpret_cast<const float4*>(hi);
    d[1] = *reinterpret_cast<const float4*>(lo);
  }
}

// torch LSTM cell pointwise, gate order i,f,g,o.
__global__ void lstm_pointwise_kernel(const float* __restrict__ gates,
                                      const float* __restrict__ c_in, int rows,
                                      int H, float* __restrict__ h_out,
                                      float* __restrict__ c_out) {
  const long total = (long)rows * H;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / H;
    const int j = idx - r * H;
    const float* g = gates + r * 4 * H;
    float h2, c2;
    lstm_cell(g[j], g[H + j], g[2 * H + j], g[3 * H + j], c_in[idx], &h2, &c2);
    c_out[idx] = c2;
    h_out[idx] = h2;
  }
}

// ---- block helpers ----------------------------------------------------------
__device__ inline float block_max(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ inline float block_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// Select the k largest of vals[0..n) (LDS, destroyed); ties -> lowest index.
// 256 threads.  Thread 0 writes out_v/out_i.
__device__ inline void block_topk(float* vals, int n, int k, float* out_v,
                                  int* out_i, float* red_v, int* red_i) {
  const int tid = threadIdx.x;
  float bv; int bi;
  auto rescan = [&]() {
    bv = -INFINITY; bi = 0x7fffffff;
    for (int i = tid; i < n; i += 256) {
      const float v = vals[i];
      if (v > bv) { bv = v; bi = i; }
    }
  };
  rescan();
  for (int j = 0; j < k; ++j) {
    float v = bv; int i = bi;
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o);
      const int oi = __shfl_xor(i, o);
      if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
    if ((tid & 63) == 0) { red_v[tid >> 6] = v; red_i[tid >> 6] = i; }
    __syncthreads();
    float wv = red_v[0]; int wi = red_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float ov = red_v[w]; const int oi = red_i[w];
      if (ov > wv || (ov == wv && oi < wi)) { wv = ov; wi = oi; }
    }
    if (tid == 0) { out_v[j] = wv; out_i[j] = wi; }
    if (wi != 0x7fffffff && (wi & 255) == tid) {
      vals[wi] = -INFINITY;
      rescan();
    }
    __syncthreads();
  }
}

// ---- exact top-k by threshold search + sort ----------------------------------
// The k-th largest value is found by bisection on an order-preserving integer
// key (<= 32 block-wide counts), the winners are gathered and a bitonic sort
// orders them by (value desc, index asc) -- the tie rule of the oracle.  About
// 10x faster than k rounds of block argmax for k = 50, V = 5004.
constexpr int kGatherCap = 512;  // rank-count top-k: gathered candidates

__device__ inline unsigned fkey(float x) {
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// vals: LDS [n] (preserved); 2 <= k <= 256, k <= n; scratch: 528 LDS words.
__device__ inline void block_topk_fast(const float* vals, int n, int k,
                                       float* out_v, int* out_i,
                                       unsigned* scratch) {
  unsigned* cnt = scratch;                          // [8]
  float* cv = reinterpret_cast<float*>(scratch + 8);      // [256]
  int* ci = reinterpret_cast<int*>(scratch + 8 + 256);    // [256]
  const int tid = threadIdx.x;
  {
    // Rank counting first (see row_select_reg_kernel): T0 = min over the waves
    // of the wave's ceil(k/4)-th largest per-thread maximum bounds the k-th
    // value from below; the elements >= T0 are gathered and ranked (key desc,
    // index asc).  The key-bisection path below is the fallback.
    __shared__ unsigned gk[kGatherCap];
    __shared__ int gi[kGatherCap];
    __shared__ unsigned wave_t[4], gcount;
    unsigned tmax = 0u;  // keys of real values are > 0
    for (int i = tid; i < n; i += 256) tmax = max(tmax, fkey(vals[i]));
    const int lane = tid & 63;
    int rank = 0;
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      const unsigned m = (unsigned)__builtin_amdgcn_readlane((int)tmax, u);
      rank += (m > tmax) || (m == tmax && u < lane);
    }
    __syncthreads();  // previous users of the shared arrays are done
    if (tid == 0) gcount = 0u;
    if (rank == ((k + 3) >> 2) - 1) wave_t[tid >> 6] = tmax;
    __syncthreads();
    const unsigned T0 = min(min(wave_t[0], wave_t[1]), min(wave_t[2], wave_t[3]));
    if (T0 != 0u) {
      int c = 0;
      for (int i = tid; i < n; i += 256) c += fkey(vals[i]) >= T0;
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
      if ((tid & 63) == 0) cnt[tid >> 6] = (unsigned)c;
      __syncthreads();
      const int total = (int)(cnt[0] + cnt[1] + cnt[2] + cnt[3]);
      if (total <= kGatherCap) {  // block-uniform
        for (int i = tid; i < n; i += 256) {
          const unsigned key = fkey(vals[i]);
          if (key >= T0) {
            const unsigned q = atomicAdd(&gcount, 1u);
            gk[q] = key; gi[q] = i;
          }
        }
        __syncthreads();
        for (int c0 = tid; c0 < total; c0 += 256) {
          const unsigned mk = gk[c0];
          const int mi = gi[c0];
          int r2 = 0;
          for (int u = 0; u < total; ++u) {
            const unsigned ok = gk[u];
            r2 += (ok > mk) || (ok == mk && gi[u] < mi);
          }
          if (r2 < k) { out_v[r2] = vals[mi]; out_i[r2] = mi; }
        }
        return;
      }
      __syncthreads();
    }
  }
  unsigned lo = 0u, hi = 0xFFFFFFFFu;  // invariant: count(key >= lo) >= k
  for (int it = 0; it < 32 && lo < hi; ++it) {
    const unsigned mid = lo + ((hi - lo) >> 1) + ((hi - lo) & 1u);
    int c = 0;
    for (int i = tid; i < n; i += 256) c += fkey(vals[i]) >= mid;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    __syncthreads();
    if ((tid & 63) == 0) cnt[tid >> 6] = (unsigned)c;
    __syncthreads();
    const int total = (int)(cnt[0] + cnt[1] + cnt[2] + cnt[3]);
    if (total >= k) lo = mid; else hi = mid - 1u;
  }
  const unsigned T = lo;
  __syncthreads();
  if (tid == 0) { cnt[4] = 0u; cnt[5] = 0u; }
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    const float v = vals[i];
    const unsigned key = fkey(v);
    if (key > T) {
      const unsigned p = atomicAdd(&cnt[4], 1u);
      cv[p] = v; ci[p] = i;
    } else if (key == T) {
      atomicAdd(&cnt[5], 1u);
    }
  }
  __syncthreads();
  const int ngt = (int)cnt[4], neq = (int)cnt[5], need = k - ngt;
  __syncthreads();
  if (neq == need) {
    for (int i = tid; i < n; i += 256) {
      const float v = vals[i];
      if (fkey(v) == T) {
        const unsigned p = atomicAdd(&cnt[4], 1u);
        cv[p] = v; ci[p] = i;
      }
    }
  } else if (tid == 0) {  // ties across the boundary: lowest indices win (rare)
    int got = 0;
    for (int i = 0; i < n && got < need; ++i)
      if (fkey(vals[i]) == T) { cv[ngt + got] = vals[i]; ci[ngt + got] = i; ++got; }
  }
  __syncthreads();
  int P = 64;
  while (P < k) P <<= 1;
  if (tid >= k && tid < P) { cv[tid] = -INFINITY; ci[tid] = 0x7fffffff; }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int partner = tid ^ stride;
      if (tid < P && partner > tid) {
        const float av = cv[tid], bv = cv[partner];
        const int ai = ci[tid], bi = ci[partner];
        const bool a_first = av > bv || (av == bv && ai < bi);
        const bool up = (tid & size) == 0;
        if (up ? !a_first : a_first) {
          cv[tid] = bv; ci[tid] = bi; cv[partner] = av; ci[partner] = ai;
        }
      }
      __syncthreads();
    }
  if (tid < k) { out_v[tid] = cv[tid]; out_i[tid] = ci[tid]; }
}

// Per row: pred = log_softmax(logits) [- lambda * log_softmax(lm_logits)]
// (decoders.py:621,624-630); optionally store pred; then the top-k of pred
// (k = beam for allennlp's per-node topk, k = 1 for greedy argmax).  Rows whose
// last token is <stop> get allennlp's forced distribution instead (0 at stop,
// finfo.min elsewhere).  One workgroup per row, the row lives in LDS.
__global__ __launch_bounds__(256) void row_select_kernel(
    const float* __restrict__ logits, const float* __restrict__ lm_logits,
    float lambda, int V, int k, const int64_t* __restrict__ last_tok, int stop,
    float* __restrict__ cand_v, int* __restrict__ cand_i,
    float* __restrict__ pred_out, long pred_stride) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* pred = sm;            // [V]
  float* red = sm + V;         // [4]
  int* redi = (int*)(red + 4); // [4]
  const int r = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (long)r * V;
  float mx = -INFINITY;
  for (int i = tid; i < V; i += 256) { const float v = x[i]; pred[i] = v; mx = fmaxf(mx, v); }
  mx = block_max(mx, red);
  float s = 0.f;
  for (int i = tid; i < V; i += 256) s += expf(pred[i] - mx);
  s = block_sum(s, red);
  const float ls = logf(s);
  if (lm_logits) {
    const float* y = lm_logits + (long)r * V;
    float my = -INFINITY;
    for (int i = tid; i < V; i += 256) my = fmaxf(my, y[i]);
    my = block_max(my, red);
    float sy = 0.f;
    for (int i = tid; i < V; i += 256) sy += expf(y[i] - my);
    sy = block_sum(sy, red);
    const float lsy = logf(sy);
    for (int i = tid; i < V; i += 256)
      pred[i] = ((pred[i] - mx) - ls) - lambda * ((y[i] - my) - lsy);
  } else {
    for (int i = tid; i < V; i += 256) pred[i] = (pred[i] - mx) - ls;
  }
  if (pred_out) {
    float* po = pred_out + (long)r * pred_stride;
    for (int i = tid; i < V; i += 256) po[i] = pred[i];
  }
  if (k <= 0) return;
  __syncthreads();
  if (last_tok && last_tok[r] == stop) {
    for (int j = tid; j < k; j += 256) {
      cand_v[(long)r * k + j] = j == 0 ? 0.f : kFloatMin;
      cand_i[(long)r * k + j] = j == 0 ? stop : (j - 1 < stop ? j - 1 : j);
    }
    return;
  }
  if (k >= 2 && k <= 256)
    block_topk_fast(pred, V, k, cand_v + (long)r * k, cand_i + (long)r * k,
                    reinterpret_cast<unsigned*>(sm + V + 8));
  else
    block_topk(pred, V, k, cand_v + (long)r * k, cand_i + (long)r * k, red, redi);
}

// Register-resident variant of row_select_kernel for V <= 24*256: the row, its
// log-softmax and the order-preserving keys never leave the VGPRs (24 values
// per thread); LDS holds only reduction scratch and the <= 256 winners.  Same
// results as row_select_kernel (same float operations, same tie rule).
constexpr int kRowRegs = 24;

__device__ inline float funkey(unsigned key) {
  return __uint_as_float((key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
}

__global__ __launch_bounds__(256) void row_select_reg_kernel(
    const float* __restrict__ logits, const float* __restrict__ lm_logits,
    float lambda, int V, int k, const int64_t* __restrict__ last_tok, int stop,
    float* __restrict__ cand_v, int* __restrict__ cand_i,
    float* __restrict__ pred_out, long pred_stride, int rank_select) {
  __shared__ float red[8];
  __shared__ unsigned scratch[528];
  const int r = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (long)r * V;
  float p[kRowRegs];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < kRowRegs; ++j) {
    const int i = tid + 256 * j;
    p[j] = i < V ? x[i] : -INFINITY;
    mx = fmaxf(mx, p[j]);
  }
  mx = block_max(mx, red);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < kRowRegs; ++j)
    if (tid + 256 * j < V) s += expf(p[j] - mx);
  s = block_sum(s, red);
  const float ls = logf(s);
  if (lm_logits) {
    const float* y = lm_logits + (long)r * V;
    float q[kRowRegs];
    float my = -INFINITY;
#pragma unroll
    for (int j = 0; j < kRowRegs; ++j) {
      const int i = tid + 256 * j;
      q[j] = i < V ? y[i] : -INFINITY;
      my = fmaxf(my, q[j]);
    }
    my = block_max(my, red);
    float sy = 0.f;
#pragma unroll
    for (int j = 0; j < kRowRegs; ++j)
      if (tid + 256 * j < V) sy += expf(q[j] - my);
    sy = block_sum(sy, red);
    const float lsy = logf(sy);
#pragma unroll
    for (int j = 0; j < kRowRegs; ++j)
      p[j] = ((p[j] - mx) - ls) - lambda * ((q[j] - my) - lsy);
  } else {
#pragma unroll
    for (int j = 0; j < kRowRegs; ++j) p[j] = (p[j] - mx) - ls;
  }
  if (pred_out) {
    float* po = pred_out + (long)r * pred_stride;
#pragma unroll
    for (int j = 0; j < kRowRegs; ++j)
      if (tid + 256 * j < V) po[tid + 256 * j] = p[j];
  }
  if (k <= 0) return;
  float* out_v = cand_v + (long)r * k;
  int* out_i = cand_i + (long)r * k;
  if (last_tok && last_tok[r] == stop) {
    for (int j = tid; j < k; j += 256) {
      out_v[j] = j == 0 ? 0.f : kFloatMin;
      out_i[j] = j == 0 ? stop : (j - 1 < stop ? j - 1 : j);
    }
    return;
  }
  unsigned key[kRowRegs];
#pragma unroll
  for (int j = 0; j < kRowRegs; ++j)
    key[j] = tid + 256 * j < V ? fkey(p[j]) : 0u;
  unsigned* cnt = scratch;
  float* cv = reinterpret_cast<float*>(scratch + 8);
  int* ci = reinterpret_cast<int*>(scratch + 8 + 256);
  // Exact top-k without a search over the key range.  A lower bound T0 that is
  // itself close to the k-th largest element comes from the per-thread maxima
  // (see below); the few elements >= T0 are gathered and each one's rank (key
  // descending, index ascending among equals: the order the threshold path
  // produces) is counted against the others -- ranks < k are the answer, already
  // in output order.  One barrier-free wave pass and two LDS passes instead of
  // ~25 bisection rounds with two barriers each.  Falls through to the threshold
  // path when a wave holds fewer than ceil(k/4) elements or the gather overflows
  // (long runs of equal log-probs).
  if (k >= 2) {
    __shared__ unsigned gk[kGatherCap];
    __shared__ float gp[kGatherCap];
    __shared__ int gi[kGatherCap];
    __shared__ unsigned wave_t[4], gcount;
    unsigned tmax = 0u;
#pragma unroll
    for (int j = 0; j < kRowRegs; ++j) tmax = max(tmax, key[j]);  // invalid = 0
    // T0 = min over the 4 waves of the wave's m-th largest thread maximum,
    // m = ceil(k / 4): every wave then holds >= m elements >= T0, >= k in all.
    // The rank of a lane's maximum inside its wave needs no LDS (readlane).
    const int m_need = (k + 3) >> 2;
    const int lane = tid & 63;
    int rank = 0;
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      const unsigned m = (unsigned)__builtin_amdgcn_readlane((int)tmax, u);
      rank += (m > tmax) || (m == tmax && u < lane);
    }
    if (tid == 0) gcount = 0u;
    if (rank == m_need - 1) wave_t[tid >> 6] = rank_select ? tmax : 0u;
    __syncthreads();
    const unsigned T0 = min(min(wave_t[0], wave_t[1]), min(wave_t[2], wave_t[3]));
    if (T0 != 0u) {  // block-uniform
#pragma unroll
      for (int j = 0; j < kRowRegs; ++j)
        if (key[j] >= T0) {  // invalid keys are 0 < T0
          const unsigned q = atomicAdd(&gcount, 1u);
          if (q < (unsigned)kGatherCap) { gk[q] = key[j]; gp[q] = p[j]; gi[q] = tid + 256 * j; }
        }
      __syncthreads();
      const int total = (int)gcount;
      if (total <= kGatherCap) {  // block-uniform
        for (int c0 = tid; c0 < total; c0 += 256) {
          const unsigned mk = gk[c0];
          const int mi = gi[c0];
          int r2 = 0;
          for (int u = 0; u < total; ++u) {
            const unsigned ok = gk[u];
            r2 += (ok > mk) || (ok == mk && gi[u] < mi);
          }
          if (r2 < k) { out_v[r2] = gp[c0]; out_i[r2] = mi; }
        }
        return;
      }
    }
    __syncthreads();
  }
  // greedy argmax and the threshold path work from the key range
  unsigned kmax = 0u, kmin = 0xFFFFFFFFu;
#pragma unroll
  for (int j = 0; j < kRowRegs; ++j)
    if (tid + 256 * j < V) { kmax = max(kmax, key[j]); kmin = min(kmin, key[j]); }
  for (int o = 32; o > 0; o >>= 1) {
    kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, o));
    kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o));
  }
  __syncthreads();
  if ((tid & 63) == 0) { cnt[tid >> 6] = kmax; cnt[4 + (tid >> 6)] = kmin; }
  __syncthreads();
  unsigned hi = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
  unsigned lo = min(min(cnt[4], cnt[5]), min(cnt[6], cnt[7]));
  if (k == 1) {  // greedy argmax: the maximum, lowest index among equals
    int best = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < kRowRegs; ++j)
      if (key[j] == hi && tid + 256 * j < V) best = min(best, tid + 256 * j);
    for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
    __syncthreads();
    if ((tid & 63) == 0) cnt[tid >> 6] = (unsigned)best;
    __syncthreads();
    if (tid == 0) {
      out_i[0] = (int)min(min(cnt[0], cnt[1]), min(cnt[2], cnt[3]));
      out_v[0] = funkey(hi);
    }
    return;
  }
  // largest T with count(key >= T) >= k   (count(key >= lo) = V >= k)
  for (int it = 0; it < 32 && lo < hi; ++it) {
    const unsigned mid = lo + ((hi - lo) >> 1) + ((hi - lo) & 1u);
    int c = 0;
#pragma unroll
    for (int j = 0; j < kRowRegs; ++j) c += key[j] >= mid;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    __syncthreads();
    if ((tid & 63) == 0) cnt[tid >> 6] = (unsigned)c;
    __syncthreads();
    const int total = (int)(cnt[0] + cnt[1] + cnt[2] + cnt[3]);
    if (total >= k) lo = mid; else hi = mid - 1u;
  }
  const unsigned T = lo;
  __syncthreads();
  if (tid == 0) { cnt[4] = 0u; cnt[5] = 0u; }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kRowRegs; ++j) {
    const int i = tid + 256 * j;
    if (i < V) {
      if (key[j] > T) {
        const unsigned q = atomicAdd(&cnt[4], 1u);
        cv[q] = p[j]; ci[q] = i;
      } else if (key[j] == T) {
        atomicAdd(&cnt[5], 1u);
      }
    }
  }
  __syncthreads();
  const int ngt = (int)cnt[4], neq = (int)cnt[5], need = k - ngt;
  __syncthreads();
  if (neq == need) {
#pragma unroll
    for (int j = 0; j < kRowRegs; ++j) {
      const int i = tid + 256 * j;
      if (i < V && key[j] == T) {
        const unsigned q = atomicAdd(&cnt[4], 1u);
        cv[q] = p[j]; ci[q] = i;
      }
    }
  } else {
    // ties straddle the boundary: the `need` lowest indices win.  Picked one
    // at a time by a block-min over the tied indices above the previous pick
    // (rare: needs equal log-probs exactly at the k-th value).
    int prev = -1;
    for (int round = 0; round < need; ++round) {
      int best = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < kRowRegs; ++j) {
        const int i = tid + 256 * j;
        if (i < V && key[j] == T && i > prev) best = min(best, i);
      }
      for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o));
      __syncthreads();
      if ((tid & 63) == 0) cnt[tid >> 6] = (unsigned)best;
      __syncthreads();
      prev = (int)min(min(cnt[0], cnt[1]), min(cnt[2], cnt[3]));
      if (tid == 0) { cv[ngt + round] = funkey(T); ci[ngt + round] = prev; }
    }
  }
  __syncthreads();
  int P = 64;
  while (P < k) P <<= 1;
  if (tid >= k && tid < P) { cv[tid] = -INFINITY; ci[tid] = 0x7fffffff; }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int partner = tid ^ stride;
      if (tid < P && partner > tid) {
        const float av = cv[tid], bv = cv[partner];
        const int ai = ci[tid], bi = ci[partner];
        const bool a_first = av > bv || (av == bv && ai < bi);
        const bool up = (tid & size) == 0;
        if (up ? !a_first : a_first) {
          cv[tid] = bv; ci[tid] = bi; cv[partner] = av; ci[partner] = ai;
        }
      }
      __syncthreads();
    }
  if (tid < k) { out_v[tid] = cv[tid]; out_i[tid] = ci[tid]; }
}

// allennlp beam restriction: per neuron, top-`beam` of the beam_prev*beam
// summed candidates; backpointer = flat index / beam (trunc).
__global__ __launch_bounds__(256) void beam_merge_kernel(
    const float* __restrict__ cand_v, const int* __restrict__ cand_i,
    const float* __restrict__ last_lp, int beam_prev, int beam,
    float* __restrict__ new_lp, int* __restrict__ new_tok,
    int* __restrict__ new_bp) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int nc = beam_prev * beam;
  float* vals = sm;                 // [nc]
  float* outv = sm + nc;            // [beam]
  int* outi = (int*)(outv + beam);  // [beam]
  float* red = (float*)(outi + beam);
  int* redi = (int*)(red + 4);
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < nc; i += 256) {
    const float lp = last_lp ? last_lp[(long)n * beam_prev + i / beam] : 0.f;
    vals[i] = cand_v[(long)n * nc + i] + lp;
  }
  __syncthreads();
  if (beam >= 2 && beam <= 256)
    block_topk_fast(vals, nc, beam, outv, outi,
                    reinterpret_cast<unsigned*>(redi + 4));
  else
    block_topk(vals, nc, beam, outv, outi, red, redi);
  __syncthreads();
  for (int j = tid; j < beam; j += 256) {
    const int idx = outi[j];
    new_lp[(long)n * beam + j] = outv[j];
    new_tok[(long)n * beam + j] = cand_i[(long)n * nc + idx];
    new_bp[(long)n * beam + j] = idx / beam;
  }
}

// Index-only beam reorder (allennlp _update_state without moving features):
// new row (n,j) takes h/c (and LM state) of old row n*beam_prev + bp, and its
// next input token.  Also appends to the per-step history.
__global__ void beam_reorder_kernel(
    const int* __restrict__ new_tok, const int* __restrict__ new_bp, int n,
    int beam_prev, int beam, int H, const float* __restrict__ h_src,
    const float* __restrict__ c_src, float* __restrict__ h_dst,
    float* __restrict__ c_dst, int lm_layers, int Hl, long lm_rows_src,
    long lm_rows_dst, const float* __restrict__ hl_src,
    const float* __restrict__ cl_src, float* __restrict__ hl_dst,
    float* __restrict__ cl_dst, int64_t* __restrict__ tok_dst,
    int* __restrict__ hist_tok, int* __restrict__ hist_bp) {
  const int r = blockIdx.x;  // new row
  const int nn = r / beam;
  const int src = nn * beam_prev + new_bp[r];
  for (int j = threadIdx.x; j < H; j += blockDim.x) {
    h_dst[(long)r * H + j] = h_src[(long)src * H + j];
    c_dst[(long)r * H + j] = c_src[(long)src * H + j];
  }
  for (int l = 0; l < lm_layers; ++l)
    for (int j = threadIdx.x; j < Hl; j += blockDim.x) {
      hl_dst[((long)l * lm_rows_dst + r) * Hl + j] =
          hl_src[((long)l * lm_rows_src + src) * Hl + j];
      cl_dst[((long)l * lm_rows_dst + r) * Hl + j] =
          cl_src[((long)l * lm_rows_src + src) * Hl + j];
    }
  if (threadIdx.x == 0) {
    tok_dst[r] = new_tok[r];
    hist_tok[r] = new_tok[r];
    hist_bp[r] = new_bp[r];
  }
}

// Back-trace (allennlp _reconstruct_sequences) + T' per neuron group.
__global__ void beam_finalize_kernel(const int* __restrict__ hist_tok,
                                     const int* __restrict__ hist_bp, int n,
                                     int beam, int T,
                                     int64_t* __restrict__ beam_tokens) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n * beam) return;
  const int nn = r / beam;
  int j = r - nn * beam;
  const long R = (long)n * beam;
  for (int t = T - 1; t >= 0; --t) {
    beam_tokens[(long)r * T + t] = hist_tok[t * R + nn * beam + j];
    j = hist_bp[t * R + nn * beam + j];
  }
}

// T'[g] = 1 + first step t whose chosen tokens are all <stop> over the group's
// neurons (allennlp's early exit, evaluated per reference forward batch).
__global__ void group_len_kernel(const int* __restrict__ hist_tok, int n,
                                 int beam, int T, int group, int stop,
                                 int32_t* __restrict__ out_len) {
  const int g = blockIdx.x;
  const int n0 = g * group, n1 = min(n, n0 + group);
  const long R = (long)n * beam;
  __shared__ int len;
  if (threadIdx.x == 0) len = T;
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    int all = 1;
    for (int i = n0 * beam + threadIdx.x; i < n1 * beam; i += blockDim.x)
      all &= hist_tok[t * R + i] == stop;
    all = __syncthreads_and(all);
    if (all) {
      if (threadIdx.x == 0) len = t + 1;
      break;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out_len[g] = len;
}

// greedy bookkeeping (decoders.py:447,456-463)
__global__ void greedy_record_kernel(const float* __restrict__ cand_v,
                                     const int* __restrict__ cand_i,
                                     const float* __restrict__ att, int n, int k,
                                     int t, int T, int64_t* __restrict__ tokens,
                                     float* __restrict__ scores,
                                     float* __restrict__ attentions,
                                     int64_t* __restrict__ next_tok) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  tokens[(long)r * T + t] = cand_i[r];
  next_tok[r] = cand_i[r];
  scores[r] = (t == 0 ? 0.f : scores[r]) + cand_v[r];
  if (attentions)
    for (int j = 0; j < k; ++j)
      attentions[((long)r * T + t) * k + j] = att[(long)r * k + j];
}

// strategy=<tensor>: the forced token is the next input, its log-prob the score
__global__ void forced_record_kernel(const float* __restrict__ pred,
                                     long pred_stride, int V,
                                     const float* __restrict__ att, int n, int k,
                                     int t, int T,
                                     const int64_t* __restrict__ tokens,
                                     float* __restrict__ scores,
                                     float* __restrict__ attentions,
                                     int64_t* __restrict__ next_tok) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int64_t tok = tokens[(long)r * T + t];
  tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);  // callers validate the range
  next_tok[r] = tok;
  scores[r] = (t == 0 ? 0.f : scores[r]) + pred[(long)r * pred_stride + tok];
  if (attentions)
    for (int j = 0; j < k; ++j)
      attentions[((long)r * T + t) * k + j] = att[(long)r * k + j];
}

__global__ void fill_i64_kernel(int64_t* p, long n, int64_t v) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// LM: total[r] += alive[r] * valid * logp(target);  alive *= (input != stop)
// (lms.py:93-100 including the j+1 off-by-one: `alive` lags one step).
__global__ __launch_bounds__(256) void lm_accumulate_kernel(
    const float* __restrict__ logits, int V, const int64_t* __restrict__ seqs,
    long lds, int t, const int32_t* __restrict__ seq_len, int len_div, int stop,
    float* __restrict__ alive, float* __restrict__ total) {
  __shared__ float red[4];
  const int r = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (long)r * V;
  float mx = -INFINITY;
  float s = 0.f;
  if (V <= kRowRegs * 256) {
    // the row is read once and stays in registers between the two passes (same
    // operations in the same order as the streaming form below)
    float p[kRowRegs];
#pragma unroll
    for (int j = 0; j < kRowRegs; ++j) {
      const int i = tid + 256 * j;
      p[j] = i < V ? x[i] : -INFINITY;
      mx = fmaxf(mx, p[j]);
    }
    mx = block_max(mx, red);
#pragma unroll
    for (int j = 0; j < kRowRegs; ++j)
      if (tid + 256 * j < V) s += expf(p[j] - mx);
  } else {
    for (int i = tid; i < V; i += 256) mx = fmaxf(mx, x[i]);
    mx = block_max(mx, red);
    for (int i = tid; i < V; i += 256) s += expf(x[i] - mx);
  }
  s = block_sum(s, red);
  if (tid == 0) {
    const int64_t in = seqs[(long)r * lds + t], tgt = seqs[(long)r * lds + t + 1];
    const bool valid = seq_len == nullptr || (t + 1) < seq_len[r / len_div];
    const float a = t == 0 ? 1.f : alive[r];
    float tot = t == 0 ? 0.f : total[r];
    if (valid) tot += a * ((x[tgt] - mx) - logf(s));
    total[r] = tot;
    alive[r] = a * (in != stop ? 1.f : 0.f);
  }
}

// The same from the statistics the vocabulary GEMM's EPI_LSE epilogue leaves (gemm.hip): per
// row and 64-column block {max, sum exp(x - max)}, and the target column's x -- the 4 V bytes
// per row of logits are neither written nor read back (1.28 GB per step at 32 000 rows).
// logsumexp = M + log(sum_b s_b exp(m_b - M)); one wave per row.
__global__ __launch_bounds__(256) void lm_accumulate_lse_kernel(
    const float* __restrict__ part, int nb, const float* __restrict__ xt, int rows,
    const int64_t* __restrict__ seqs, long lds, int t, const int32_t* __restrict__ seq_len,
    int len_div, int stop, float* __restrict__ alive, float* __restrict__ total,
    const int* __restrict__ map = nullptr) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  // (map: the statistics of row r live at row map[r] -- rows that share their prefix AND their
  // next token share one row of the vocabulary product, lm_score_dedup)
  const int sr = map ? map[r] : r;
  const float* p = part + (long)sr * nb * 2;
  float mx = -INFINITY;
  for (int b = lane; b < nb; b += 64) mx = fmaxf(mx, p[2 * b]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float s = 0.f;
  for (int b = lane; b < nb; b += 64) s += p[2 * b + 1] * expf(p[2 * b] - mx);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    const int64_t in = seqs[(long)r * lds + t];
    const bool valid = seq_len == nullptr || (t + 1) < seq_len[r / len_div];
    const float a = t == 0 ? 1.f : alive[r];
    float tot = t == 0 ? 0.f : total[r];
    if (valid) tot += a * ((xt[sr] - mx) - logf(s));
    total[r] = tot;
    alive[r] = a * (in != stop ? 1.f : 0.f);
  }
}

// seqs[r] = [start, beam_tokens[r][0..T)]
__global__ void build_lm_seqs_kernel(const int64_t* __restrict__ beam_tokens,
                                     long rows, int T, int64_t start,
                                     int64_t* __restrict__ seqs) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= rows * (T + 1)) return;
  const long r = idx / (T + 1);
  const int t = idx - r * (T + 1);
  seqs[idx] = t == 0 ? start : beam_tokens[r * T + t - 1];
}

__global__ void len_plus_one_kernel(const int32_t* in, int n, int32_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + 1;
}

// decoders.py:507-512 (rerank) / :493-494 (beam): pick per neuron.
__global__ void rerank_select_kernel(const float* __restrict__ beam_scores,
                                     const float* __restrict__ lm_scores,
                                     float lambda, int n, int beam, int T,
                                     const int64_t* __restrict__ beam_tokens,
                                     int64_t* __restrict__ tokens,
                                     float* __restrict__ scores) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int best = 0;
  float bs = beam_scores[(long)i * beam];
  if (lm_scores) {
    bs = bs - lambda * lm_scores[(long)i * beam];
    for (int j = 1; j < beam; ++j) {
      const float s = beam_scores[(long)i * beam + j] -
                      lambda * lm_scores[(long)i * beam + j];
      if (s > bs) { bs = s; best = j; }
    }
  }
  scores[i] = bs;
  for (int t = 0; t < T; ++t)
    tokens[(long)i * T + t] = beam_tokens[((long)i * beam + best) * T + t];
}

// ---------------------------------------------------------------------------
// host drivers
// ---------------------------------------------------------------------------
static inline int nblk(long total, int cap = 16384) {
  long b = (total + 255) / 256;
  return (int)(b < cap ? (b < 1 ? 1 : b) : cap);
}

// y = epi(A W^T + b): fp32 MFMA, or -- in split-f16 mode, when the layer has a
// split weight copy -- A is first rewritten as (hi,lo) f16 pairs into the
// context's scratch and the 3xf16 MFMA kernel runs (outputs stay fp32).
static int lin(milan_ctx* c, const float* A, long lda, const LinearW& w, float* C,
               int ldc, int M, int epi, hipStream_t s,
               const float* aux = nullptr, int ldaux = 0) {
  GemmArgs g = linear_args(A, lda, w.w, w.b, C, ldc, M, w.n, w.k, epi, c->zero,
                           aux, ldaux);
  if (c->precision == MILAN_PRECISION_SPLIT_F16 && w.ws && c->scratch &&
      (size_t)M * w.k <= c->scratch_floats) {
    MILAN_TRY(launch_f32_to_split(A, lda, c->scratch, w.k, M, w.k, 1.f, s));
    g.A = c->scratch;
    g.a_pix_stride = g.a_img_stride = w.k;
    g.W = w.ws;
    g.a_split = 1;
    g.acc_scale = w.ws_inv;
  }
  return launch_gemm(g, s);
}

// One LSTM layer step: (h', c') = cell(A1 W1^T + A2 W2^T + b1 + b2, c).  Split
// mode with a concatenated weight: both operands are converted side by side
// into one scratch matrix [A1 | A2] and multiplied by [W1 | W2] in ONE launch
// whose epilogue applies the cell (gate-interleaved rows, EPI_LSTM) -- the gate
// pre-activations never reach HBM; otherwise two GEMMs, the second accumulating
// into the first, and the pointwise kernel.
static int lstm_layer(milan_ctx* c, const float* A1, long lda1, const LinearW& w1,
                      const float* A2, long lda2, const LinearW& w2,
                      const LinearW& cat, float* gates, const float* c_in,
                      float* h_out, float* c_out, int M, hipStream_t s) {
  const int H = w1.n / 4;
  if (c->precision == MILAN_PRECISION_SPLIT_F16 && cat.ws && c->scratch &&
      (size_t)M * cat.k <= c->scratch_floats) {
    MILAN_TRY(launch_f32_to_split(A1, lda1, c->scratch, cat.k, M, w1.k, 1.f, s));
    MILAN_TRY(launch_f32_to_split(A2, lda2, c->scratch + w1.k, cat.k, M, w2.k,
                                  1.f, s));
    GemmArgs g = linear_args(c->scratch, cat.k, cat.ws, cat.b, gates, 4 * H, M,
                             cat.n, cat.k, EPI_BIAS, c->zero);
    g.a_split = 1;
    g.acc_scale = cat.ws_inv;
    if (cat.gate_interleaved && H % 8 == 0) {
      g.epilogue = EPI_LSTM;
      g.C = h_out; g.C2 = c_out; g.ldc = H;
      g.aux = c_in; g.ldaux = H;
      return launch_gemm(g, s);
    }
    MILAN_REQUIRE(!cat.gate_interleaved, MILAN_ERR_STATE, "lstm: weight layout");
    MILAN_TRY(launch_gemm(g, s));
  } else {
    MILAN_TRY(lin(c, A1, lda1, w1, gates, 4 * H, M, EPI_BIAS, s));
    MILAN_TRY(lin(c, A2, lda2, w2, gates, 4 * H, M, EPI_BIAS_ADD, s, gates, 4 * H));
  }
  hipLaunchKernelGGL(lstm_pointwise_kernel, dim3(nblk((long)M * H)), dim3(256), 0, s,
                     gates, c_in, M, H, h_out, c_out);
  return 0;
}

struct LmState {  // [layers][rows][Hl]
  float *h = nullptr, *c = nullptr;
  long rows = 0;
};

// One LM token step: returns logits (rows,V) in `logits`.
static int lm_step(milan_ctx* c, const int64_t* tok, int rows, LmState& st,
                   LmState& nx, float* emb, float* gates, float* logits,
                   hipStream_t s) {
  const milan_dims& d = c->d;
  const int Hl = d.lm_hidden_size, El = d.lm_embedding_size, V = d.vocab_size;
  hipLaunchKernelGGL(embed_kernel, dim3(nblk((long)rows * El)), dim3(256), 0, s,
                     c->lm_embedding, tok, rows, El, emb, El);
  const float* in = emb;
  int in_dim = El;
  for (int l = 0; l < d.lm_layers; ++l) {
    const float* hl = st.h + (long)l * st.rows * Hl;
    const float* cl = st.c + (long)l * st.rows * Hl;
    float* hn = nx.h + (long)l * nx.rows * Hl;
    float* cn = nx.c + (long)l * nx.rows * Hl;
    MILAN_TRY(lstm_layer(c, in, in_dim, c->lm_ih[l], hl, Hl, c->lm_hh[l],
                         c->lm_cat[l], gates, cl, hn, cn, rows, s));
    in = hn;
    in_dim = Hl;
  }
  MILAN_TRY(lin(c, in, Hl, c->lm_out, logits, V, rows, EPI_BIAS, s));
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

struct DecBuf {
  float *keys, *pooled, *x, *h, *cc, *hn, *cn, *q, *att, *ctx, *gates, *logits;
  float *lm_logits, *lm_emb, *lm_gates;
  LmState lm[2];
  float* cand_v; int* cand_i;
  float* last_lp[2]; int *new_tok, *new_bp, *hist_tok, *hist_bp;
  int64_t *tok, *seqs;
  float *lm_alive, *lm_total;
  float* lm_dedup; size_t lm_dedup_floats;   // scratch of the prefix-class rerank pass (lm_score_dedup)
  int32_t* len1;
  float* scratch;
  size_t scratch_floats;
};

static size_t lm_dedup_floats(const milan_dims& d, size_t R, int L, int G);

static void dec_plan(const milan_ctx* c, int n, int k, int beam, int T, bool lm,
                     Arena& a, DecBuf* b) {
  const milan_dims& d = c->d;
  const size_t R = (size_t)n * beam;
  const int F = d.feature_size, H = d.hidden_size, E = d.embedding_size,
            A = d.attention_size, V = d.vocab_size;
  b->keys = a.get<float>((size_t)n * k * A);
  b->pooled = a.get<float>((size_t)n * F);
  b->x = a.get<float>(R * (E + F));
  b->h = a.get<float>(R * H); b->cc = a.get<float>(R * H);
  b->hn = a.get<float>(R * H); b->cn = a.get<float>(R * H);
  b->q = a.get<float>(R * A);
  b->att = a.get<float>(R * k);
  b->ctx = a.get<float>(R * F);
  b->gates = a.get<float>(R * 4 * H);
  b->logits = a.get<float>(R * V);
  b->cand_v = a.get<float>(R * beam); b->cand_i = a.get<int>(R * beam);
  b->last_lp[0] = a.get<float>(R); b->last_lp[1] = a.get<float>(R);
  b->new_tok = a.get<int>(R); b->new_bp = a.get<int>(R);
  b->hist_tok = a.get<int>(R * T); b->hist_bp = a.get<int>(R * T);
  b->tok = a.get<int64_t>(R);
  b->seqs = a.get<int64_t>(R * (T + 1));
  b->len1 = a.get<int32_t>(2 * (size_t)n + 2);
  {
    const size_t rows = R > (size_t)n * k ? R : (size_t)n * k;
    // widest split-format A operand: the LSTM's concatenated [x | h]
    b->scratch_floats = rows * (size_t)(E + F + d.hidden_size);
    b->scratch = a.get<float>(b->scratch_floats);
  }
  b->lm_logits = nullptr;
  b->lm_dedup = nullptr; b->lm_dedup_floats = 0;
  if (lm) {
    const int Hl = d.lm_hidden_size, El = d.lm_embedding_size;
    b->lm_logits = a.get<float>(R * V);
    b->lm_emb = a.get<float>(R * El);
    b->lm_gates = a.get<float>(R * 4 * Hl);
    for (int i = 0; i < 2; ++i) {
      b->lm[i].h = a.get<float>(R * Hl * d.lm_layers);
      b->lm[i].c = a.get<float>(R * Hl * d.lm_layers);
      b->lm[i].rows = (long)R;
    }
    b->lm_alive = a.get<float>(R);
    b->lm_total = a.get<float>(R);
    // the rerank pass's prefix classes + gathered states (~2.6 k floats per row at Hl = 512)
    b->lm_dedup_floats = beam > 1 ? lm_dedup_floats(d, R, T + 1, n) : 0;
    b->lm_dedup = b->lm_dedup_floats ? a.get<float>(b->lm_dedup_floats) : nullptr;
  }
}

size_t decoder_workspace(const milan_ctx* c, int n, int k, int beam, int length) {
  Arena a; a.dry = true;
  DecBuf b;
  dec_plan(c, n, k, beam < 1 ? 1 : beam, length, c->d.has_lm != 0, a, &b);
  return a.off;
}

static int check_dims(const milan_ctx* c, int k) {
  MILAN_REQUIRE(c->finalized && c->lstm_ih.w != nullptr, MILAN_ERR_STATE,
                "decoder weights not uploaded/finalized");
  MILAN_REQUIRE(k >= 1 && k <= 64, MILAN_ERR_SHAPE,
                "number of exemplars k=%d must be in 1..64", k);
  MILAN_REQUIRE(c->d.feature_size % 4 == 0, MILAN_ERR_SHAPE,
                "feature_size must be a multiple of 4");
  return 0;
}

// keys = key_to_hidden(features): (n*k, F) x (A, F)^T
static int project_keys(milan_ctx* c, const float* features, int nk, float* keys,
                        hipStream_t s) {
  const milan_dims& d = c->d;
  return lin(c, features, d.feature_size, c->k2h, keys, d.attention_size, nk, EPI_BIAS, s);
}

static int init_state_impl(milan_ctx* c, const float* features, int n, int k,
                           float* pooled, float* h, float* cc, hipStream_t s) {
  const milan_dims& d = c->d;
  const int F = d.feature_size, H = d.hidden_size;
  hipLaunchKernelGGL(mean_k_kernel, dim3(nblk((long)n * F)), dim3(256), 0, s,
                     features, n, k, F, pooled);
  MILAN_TRY(lin(c, pooled, F, c->init_h, h, H, n, EPI_BIAS_TANH, s));
  MILAN_TRY(lin(c, pooled, F, c->init_c, cc, H, n, EPI_BIAS_TANH, s));
  return 0;
}

int decoder_init_state(milan_ctx* c, const float* features, int n, int k,
                       float* h, float* cc, Arena& ws, hipStream_t s) {
  MILAN_TRY(check_dims(c, k));
  float* pooled = ws.get<float>((size_t)n * c->d.feature_size);
  MILAN_REQUIRE(pooled, MILAN_ERR_WORKSPACE, "init_state: workspace too small");
  c->scratch_floats = (size_t)n * c->d.feature_size;
  c->scratch = ws.get<float>(c->scratch_floats);  // may be null: fp32 path
  return init_state_impl(c, features, n, k, pooled, h, cc, s);
}

static void launch_context(const float* att, const float* features, int rows,
                           int rpn, int k, int F, float* ctx, hipStream_t s) {
  const int gy = (F / 4 + 255) / 256;
  const int neurons = (rows + rpn - 1) / rpn;
  if (k <= kCtxMaxK)
    hipLaunchKernelGGL(context_kernel<true>, dim3(neurons, gy), dim3(256), 0, s,
                       att, features, rows, rpn, k, F, ctx);
  else
    hipLaunchKernelGGL(context_kernel<false>, dim3(neurons, gy), dim3(256), 0, s,
                       att, features, rows, rpn, k, F, ctx);
}

// Everything of Decoder.step up to the vocabulary logits, for `rows` rows that
// share features in groups of `rpn`.  x = [emb | gated] is built in b->x.
static int step_core(milan_ctx* c, const float* features, const float* keys,
                     int rows, int rpn, int k, const int64_t* tok,
                     const float* h, const float* cc, float* hn, float* cn,
                     DecBuf* b, hipStream_t s) {
  const milan_dims& d = c->d;
  const int F = d.feature_size, H = d.hidden_size, E = d.embedding_size,
            A = d.attention_size, V = d.vocab_size;
  const int ldx = E + F;
  // Split-f16 mode with every weight in split form: h is converted ONCE (it
  // feeds three GEMMs), and the LSTM input x = [emb | gated context] is produced
  // directly in split format -- the gate GEMM's epilogue writes its half, the
  // embedding gather the other -- so the (rows, E+F) matrix is never converted
  // (470 -> 50 MB of conversion traffic per step).  Same bits as converting.
  const bool fused = c->precision == MILAN_PRECISION_SPLIT_F16 && c->scratch &&
                     c->q2h.ws && c->gate.ws && c->lstm_cat.ws && c->out.ws &&
                     E % 16 == 0 && F % 8 == 0 && H % 16 == 0 &&
                     (size_t)rows * (ldx + H) <= c->scratch_floats;
  if (fused) {
    float* xs = c->scratch;                       // (rows, E+F) split
    float* hs = c->scratch + (size_t)rows * ldx;  // (rows, H) split
    float* hs2 = b->gates;  // h' split (vocabulary GEMM); the gate matrix is unused here
    MILAN_TRY(launch_f32_to_split(h, H, hs, H, rows, H, 1.f, s));
    auto split_lin = [&](const LinearW& w, float* C, int ldc, int epi,
                         const float* aux, int ldaux) {
      GemmArgs g = linear_args(hs, H, w.ws, w.b, C, ldc, rows, w.n, w.k, epi,
                               c->zero, aux, ldaux);
      g.a_split = 1;
      g.acc_scale = w.ws_inv;
      return g;
    };
    MILAN_TRY(launch_gemm(split_lin(c->q2h, b->q, A, EPI_BIAS, nullptr, 0), s));
    launch_attend(b->q, keys, c->att_w, c->att_b, rows, rpn, k, A, b->att, s);
    launch_context(b->att, features, rows, rpn, k, F, b->ctx, s);
    {
      GemmArgs g = split_lin(c->gate, xs + E, ldx, EPI_BIAS_SIGMUL, b->ctx, F);
      g.out_split = 1;  // aux stays fp32
      MILAN_TRY(launch_gemm(g, s));
    }
    hipLaunchKernelGGL(embed_split_kernel, dim3(nblk((long)rows * (E / 8))),
                       dim3(256), 0, s, c->embedding, tok, rows, E, xs, ldx);
    {
      // gates = [x | h] [W_ih | W_hh]^T: x from xs (k < E+F), h from hs
      const LinearW& w = c->lstm_cat;
      // the epilogue applies the cell and writes h' in both forms (fp32 state +
      // split operand of the vocabulary GEMM); hs is this GEMM's own A2 source,
      // so the split copy goes to a second buffer
      const bool cell = w.gate_interleaved;  // false only under MILAN_LSTM_FUSE=0
      GemmArgs g = cell ? linear_args(xs, ldx, w.ws, w.b, hn, H, rows, w.n, w.k,
                                      EPI_LSTM, c->zero, cc, H)
                        : linear_args(xs, ldx, w.ws, w.b, b->gates, 4 * H, rows,
                                      w.n, w.k, EPI_BIAS, c->zero);
      if (cell) { g.C2 = cn; g.Cs = hs2; }
      g.a_split = 1;
      g.acc_scale = w.ws_inv;
      g.Cin = ldx;  // geometry of source 1
      g.A2 = hs; g.K1 = ldx; g.H2 = 1; g.W2d = 1; g.stride2 = 1;
      g.a2_pix_stride = H; g.a2_img_stride = H;
      MILAN_TRY(launch_gemm(g, s));
      if (!cell) {
        hipLaunchKernelGGL(lstm_pointwise_kernel, dim3(nblk((long)rows * H)),
                           dim3(256), 0, s, b->gates, cc, rows, H, hn, cn);
        hs2 = hs;
        MILAN_TRY(launch_f32_to_split(hn, H, hs2, H, rows, H, 1.f, s));
      }
    }
    hs = hs2;
    MILAN_TRY(launch_gemm(split_lin(c->out, b->logits, V, EPI_BIAS, nullptr, 0), s));
    MILAN_CHECK_HIP(hipGetLastError());
    return 0;
  }
  MILAN_TRY(lin(c, h, H, c->q2h, b->q, A, rows, EPI_BIAS, s));
  launch_attend(b->q, keys, c->att_w, c->att_b, rows, rpn, k, A, b->att, s);
  launch_context(b->att, features, rows, rpn, k, F, b->ctx, s);
  // gated = sigmoid(W_g h + b_g) * ctx  -> x[:, E:]
  MILAN_TRY(lin(c, h, H, c->gate, b->x + E, ldx, rows, EPI_BIAS_SIGMUL, s, b->ctx, F));
  hipLaunchKernelGGL(embed_kernel, dim3(nblk((long)rows * E)), dim3(256), 0, s,
                     c->embedding, tok, rows, E, b->x, ldx);
  MILAN_TRY(lstm_layer(c, b->x, ldx, c->lstm_ih, h, H, c->lstm_hh, c->lstm_cat,
                       b->gates, cc, hn, cn, rows, s));
  MILAN_TRY(lin(c, hn, H, c->out, b->logits, V, rows, EPI_BIAS, s));
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

static size_t row_select_lds(int V) { return sizeof(float) * (V + 8 + 528); }

static int launch_row_select(const float* logits, const float* lm_logits,
                             float lambda, int rows, int V, int k,
                             const int64_t* last_tok, int stop, float* cand_v,
                             int* cand_i, float* pred_out, long pred_stride,
                             hipStream_t s) {
  if (V <= kRowRegs * 256 && k <= 256 && (k <= 1 || k <= V)) {
    // ties straddling the k-th value are resolved for up to 16 picks in the
    // register kernel; larger tie groups are impossible for distinct logits
    // MILAN_ROW_SELECT=threshold keeps the key-bisection path (A/B timing)
    static const int rank_select = [] {
      const char* e = getenv("MILAN_ROW_SELECT");
      return (e && e[0] == 't') ? 0 : 1;
    }();
    hipLaunchKernelGGL(row_select_reg_kernel, dim3(rows), dim3(256), 0, s,
                       logits, lm_logits, lambda, V, k, last_tok, stop, cand_v,
                       cand_i, pred_out, pred_stride, rank_select);
    MILAN_CHECK_HIP(hipGetLastError());
    return 0;
  }
  const size_t lds = row_select_lds(V);
  MILAN_REQUIRE(lds <= 160 * 1024, MILAN_ERR_SHAPE,
                "vocab_size %d too large for the row-select kernel", V);
  if (lds > 64 * 1024)  // per (kernel, device): a process may drive several GPUs
    MILAN_TRY(ensure_lds_attr(reinterpret_cast<const void*>(row_select_kernel), (int)lds));
  hipLaunchKernelGGL(row_select_kernel, dim3(rows), dim3(256), lds, s, logits,
                     lm_logits, lambda, V, k, last_tok, stop, cand_v, cand_i,
                     pred_out, pred_stride);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

int decoder_step(milan_ctx* c, const float* features, int rows, int k,
                 const int64_t* tokens, const float* h, const float* cc,
                 float* h_lm, float* c_lm, float temperature, float* predictions,
                 float* attentions, float* h_out, float* c_out, Arena& ws,
                 hipStream_t s) {
  MILAN_TRY(check_dims(c, k));
  const bool mi = h_lm != nullptr;
  MILAN_REQUIRE((h_lm == nullptr) == (c_lm == nullptr), MILAN_ERR_ARG,
                "state must have both h_lm and c_lm or neither");
  MILAN_REQUIRE(!mi || c->d.has_lm, MILAN_ERR_NO_LM,
                "state has h_lm or c_lm, but decoder has no lm");
  DecBuf b;
  dec_plan(c, rows, k, 1, 1, mi, ws, &b);
  c->scratch = ws.off <= ws.size ? b.scratch : nullptr;
  c->scratch_floats = b.scratch_floats;
  // per-row keys: the public step takes per-row features (un-hoisted API)
  MILAN_REQUIRE(ws.off <= ws.size, MILAN_ERR_WORKSPACE,
                "step: workspace too small (%zu needed)", ws.off);
  MILAN_TRY(project_keys(c, features, rows * k, b.keys, s));
  MILAN_TRY(step_core(c, features, b.keys, rows, 1, k, tokens, h, cc, h_out,
                      c_out, &b, s));
  const float* lm_logits = nullptr;
  if (mi) {
    LmState st{h_lm, c_lm, rows};
    MILAN_TRY(lm_step(c, tokens, rows, st, b.lm[0], b.lm_emb, b.lm_gates,
                      b.lm_logits, s));
    const size_t bytes = sizeof(float) * (size_t)rows * c->d.lm_hidden_size *
                         c->d.lm_layers;
    MILAN_CHECK_HIP(hipMemcpyAsync(h_lm, b.lm[0].h, bytes, hipMemcpyDeviceToDevice, s));
    MILAN_CHECK_HIP(hipMemcpyAsync(c_lm, b.lm[0].c, bytes, hipMemcpyDeviceToDevice, s));
    lm_logits = b.lm_logits;
  }
  MILAN_TRY(launch_row_select(b.logits, lm_logits, temperature, rows,
                              c->d.vocab_size, 0, nullptr, 0, nullptr, nullptr,
                              predictions, c->d.vocab_size, s));
  if (attentions)
    MILAN_CHECK_HIP(hipMemcpyAsync(attentions, b.att,
                                   sizeof(float) * (size_t)rows * k,
                                   hipMemcpyDeviceToDevice, s));
  return 0;
}

__global__ void gather_col_kernel(const int64_t* __restrict__ seqs, long rows,
                                  long lds, int t, int64_t* __restrict__ out) {
  const long r = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (r < rows) out[r] = seqs[r * lds + t];
}

// The LM token step of the scoring loop with every operand already in split
// format: the embedding gather writes split rows, each layer's GEMM reads
// [input | h] as two split sources and its epilogue (EPI_LSTM) leaves h' in
// split form for the next layer, the vocabulary GEMM and the next step -- no
// fp32 -> split conversion launches (5 per step otherwise).  `hs` holds the
// split hidden states, [2 slots][layers][rows][Hl]; `emb_s` (rows, El) split.
// Same bits as lm_step.
static bool lm_split_ok(const milan_ctx* c, int rows) {
  const milan_dims& d = c->d;
  if (c->precision != MILAN_PRECISION_SPLIT_F16 || !c->scratch || !c->lm_out.ws ||
      d.lm_layers < 1 || d.lm_layers > 2 || d.lm_embedding_size % 16 ||
      d.lm_hidden_size % 16 ||
      (size_t)rows * d.lm_embedding_size > c->scratch_floats)
    return false;
  for (int l = 0; l < d.lm_layers; ++l)
    if (!c->lm_cat[l].ws || !c->lm_cat[l].gate_interleaved) return false;
  return true;
}

// lse_tgt != nullptr: the vocabulary GEMM leaves log-sum-exp statistics (EPI_LSE) in `logits`
// ((rows, 2 nb) floats, then (rows) target values) instead of the logits themselves
static int lm_step_split(milan_ctx* c, const int64_t* tok, int rows, LmState& st,
                         LmState& nx, float* hs, int cur, float* logits,
                         hipStream_t s, const int64_t* lse_tgt = nullptr,
                         long lse_tgt_stride = 0) {
  const milan_dims& d = c->d;
  const int Hl = d.lm_hidden_size, El = d.lm_embedding_size, V = d.vocab_size;
  float* emb_s = c->scratch;
  hipLaunchKernelGGL(embed_split_kernel, dim3(nblk((long)rows * (El / 8))),
                     dim3(256), 0, s, c->lm_embedding, tok, rows, El, emb_s, El);
  const size_t layer_sz = (size_t)rows * Hl;
  const float* in = emb_s;
  int in_dim = El;
  for (int l = 0; l < d.lm_layers; ++l) {
    const float* h_prev = hs + ((size_t)cur * d.lm_layers + l) * layer_sz;
    float* h_next = hs + ((size_t)(cur ^ 1) * d.lm_layers + l) * layer_sz;
    const float* cl = st.c + (long)l * st.rows * Hl;
    float* hn = nx.h + (long)l * nx.rows * Hl;
    float* cn = nx.c + (long)l * nx.rows * Hl;
    const LinearW& w = c->lm_cat[l];
    GemmArgs g = linear_args(in, in_dim, w.ws, w.b, hn, Hl, rows, w.n, w.k,
                             EPI_LSTM, c->zero, cl, Hl);
    g.C2 = cn; g.Cs = h_next;
    g.a_split = 1;
    g.acc_scale = w.ws_inv;
    g.Cin = in_dim;
    g.A2 = h_prev; g.K1 = in_dim; g.H2 = 1; g.W2d = 1; g.stride2 = 1;
    g.a2_pix_stride = Hl; g.a2_img_stride = Hl;
    MILAN_TRY(launch_gemm(g, s));
    in = h_next;
    in_dim = Hl;
  }
  GemmArgs g = linear_args(in, Hl, c->lm_out.ws, c->lm_out.b, logits, V, rows,
                           c->lm_out.n, c->lm_out.k, EPI_BIAS, c->zero);
  g.a_split = 1;
  g.acc_scale = c->lm_out.ws_inv;
  if (lse_tgt) {
    const int nb = (V + 63) / 64;
    g.epilogue = EPI_LSE;
    g.ldc = 2 * nb;
    g.lse_tgt = reinterpret_cast<const long long*>(lse_tgt);
    g.lse_tgt_stride = lse_tgt_stride;
    g.lse_x = logits + (size_t)rows * 2 * nb;
  }
  MILAN_TRY(launch_gemm(g, s));
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// Rerank pass without the redundant rows (round 6).
//
// The reference scores all B x beam sequences with the LM, row by row (decoders.py:495-512,
// lms.py:58-101).  But the rows of one neuron are the leaves of a beam-search tree: at LM step
// t the LSTM state of a row is a function of its PREFIX seqs[r][0..t] only, and most of the 50
// beams share theirs (measured on the benchmark workload: 2.5 distinct prefixes of 50 after
// the first token, 16 at t = 5, 31 at t = 10 -- 40 % of the rows on average; profiles/
// r6_experiments.txt F).  Like the key projection the reference recomputes per row and step
// (hoisted in round 1), the duplicate rows are work the result does not depend on:
//   * lm_prefix_*: per group of `beam` rows and prefix length p = 0..L, rows with the same
//     prefix form a class; classes are numbered densely ACROSS the groups (cls[p][r]), with
//     their representative row (urow[p][j]) and the class of their one-shorter prefix
//     (par[p][j]); the class counts U[p] stay on the device (GemmArgs::m_live);
//   * step t multiplies the U[t + 1] classes of seqs[..t]: embedding of the class's token,
//     state gathered from the parent class, EPI_LSTM as before;
//   * the vocabulary product runs over the U[t + 2] classes of seqs[..t + 1] (same prefix AND
//     same target token), its A rows gathered from the parent class's top-layer state,
//     EPI_LSE with one target per class; lm_accumulate_lse_kernel reads row cls[t + 2][r].
// A row's result does not depend on where in a launch it is computed (every kernel choice
// follows from the layer shape), so the scores are bitwise those of the row-by-row pass
// (tests/test_gpu_lm_lse.py, the rerank goldens).
__global__ __launch_bounds__(256) void lm_prefix_classes_kernel(
    const int64_t* __restrict__ seqs, int L, int group, int R, int G,
    int* __restrict__ rep, int* __restrict__ lrank, int* __restrict__ cnt) {
  __shared__ int s_rep[256];
  __shared__ long long s_tok[256];
  __shared__ int s_isrep[256];
  const int g = blockIdx.x, b = threadIdx.x;
  const int r = g * group + b;
  const bool valid = b < group && r < R;
  int myrep = 0;   // local index of the class representative (smallest row of the class)
  if (valid) { rep[r] = g * group; lrank[r] = 0; }
  if (b == 0) cnt[g] = 1;
  for (int p = 1; p <= L; ++p) {
    const long long mytok = valid ? seqs[(long)r * L + p - 1] : -1 - b;
    s_rep[b] = valid ? myrep : -1;
    s_tok[b] = mytok;
    __syncthreads();
    int newrep = b;
    if (valid)
      for (int b2 = 0; b2 < b; ++b2)
        if (s_rep[b2] == myrep && s_tok[b2] == mytok) { newrep = b2; break; }
    __syncthreads();
    myrep = newrep;
    s_isrep[b] = (valid && newrep == b) ? 1 : 0;
    __syncthreads();
    int lr = 0, tot = 0;
    for (int b2 = 0; b2 < group; ++b2) {
      lr += b2 < b ? s_isrep[b2] : 0;
      tot += s_isrep[b2];
    }
    if (valid) { rep[(long)p * R + r] = g * group + myrep; lrank[(long)p * R + r] = lr; }
    if (b == 0) cnt[(long)p * G + g] = tot;
    __syncthreads();
  }
}

// off[p][g] = classes of prefix length p in the groups before g; U[p] = their total
__global__ __launch_bounds__(256) void lm_prefix_scan_kernel(const int* __restrict__ cnt, int G,
                                                             int* __restrict__ off,
                                                             int* __restrict__ U) {
  __shared__ int tmp[256];
  const int p = blockIdx.x, tid = threadIdx.x;
  int base = 0;
  for (int g0 = 0; g0 < G; g0 += 256) {
    const int g = g0 + tid;
    const int v = g < G ? cnt[(long)p * G + g] : 0;
    tmp[tid] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const int add = tid >= o ? tmp[tid - o] : 0;
      __syncthreads();
      tmp[tid] += add;
      __syncthreads();
    }
    if (g < G) off[(long)p * G + g] = base + tmp[tid] - v;
    base += tmp[255];
    __syncthreads();
  }
  if (tid == 0) U[p] = base;
}

__global__ void lm_prefix_assign_kernel(const int* __restrict__ rep, const int* __restrict__ lrank,
                                        const int* __restrict__ off, int L, int group, int R,
                                        int G, int* __restrict__ cls, int* __restrict__ urow,
                                        int* __restrict__ par) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int g = r / group;
  int prev = 0;
  for (int p = 0; p <= L; ++p) {
    const int rr = rep[(long)p * R + r];
    const int c = off[(long)p * G + g] + lrank[(long)p * R + rr];
    cls[(long)p * R + r] = c;
    if (rr == r) {
      urow[(long)p * R + c] = r;
      par[(long)p * R + c] = prev;   // the class of the representative's one-shorter prefix
    }
    prev = c;
  }
}

// out[j] = seqs[urow[j]][col] for the live classes j < *live
__global__ void lm_gather_tok_kernel(const int64_t* __restrict__ seqs, long L, int col,
                                     const int* __restrict__ urow, const int* __restrict__ live,
                                     int64_t* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < *live) out[j] = seqs[(long)urow[j] * L + col];
}

// dst[j][:] = src[idx[j]][:] (rows of w4 x 16 bytes) for j < *live
__global__ void lm_gather_rows_kernel(const float4* __restrict__ src, const int* __restrict__ idx,
                                      const int* __restrict__ live, int w4,
                                      float4* __restrict__ dst) {
  const long total = (long)*live * w4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long j = i / w4;
    const int q = (int)(i - j * w4);
    dst[i] = src[(long)idx[j] * w4 + q];
  }
}

struct LmDedup {
  int *cls, *urow, *par, *U;       // [L + 1][R] x 3, [L + 1]
  float *gh, *gc, *gv;             // gathered states [layers][R][Hl] x 2, top-layer rows [R][Hl]
  int64_t *tok_c, *tgt_c;          // [R]
};

// floats of scratch the dedup pass needs (dec_plan)
static size_t lm_dedup_floats(const milan_dims& d, size_t R, int L, int G) {
  const size_t maps = 5 * (size_t)(L + 1) * R + 2 * (size_t)(L + 1) * G + (size_t)(L + 1) + 64;
  const size_t state = (2 * (size_t)d.lm_layers + 1) * R * d.lm_hidden_size;
  return maps + state + 4 * R + 2048;   // (+ the 256-byte alignment of thirteen pieces)
}

static int lm_score_dedup(milan_ctx* c, const int64_t* seqs, int rows, int L, int group,
                          const int32_t* seq_len, int len_div, float* total, DecBuf* b,
                          float* scratch, hipStream_t s) {
  const milan_dims& d = c->d;
  const int Hl = d.lm_hidden_size, El = d.lm_embedding_size, V = d.vocab_size;
  const int R = rows, G = (rows + group - 1) / group, nb = (V + 63) / 64;
  // ---- carve the scratch ----
  char* base = reinterpret_cast<char*>(scratch);
  size_t off = 0;
  auto take = [&](size_t bytes) { void* q = base + off; off += (bytes + 255) & ~size_t(255); return q; };
  int* rep = (int*)take(sizeof(int) * (size_t)(L + 1) * R);
  int* lrank = (int*)take(sizeof(int) * (size_t)(L + 1) * R);
  int* cnt = (int*)take(sizeof(int) * (size_t)(L + 1) * G);
  int* offs = (int*)take(sizeof(int) * (size_t)(L + 1) * G);
  LmDedup q;
  q.cls = (int*)take(sizeof(int) * (size_t)(L + 1) * R);
  q.urow = (int*)take(sizeof(int) * (size_t)(L + 1) * R);
  q.par = (int*)take(sizeof(int) * (size_t)(L + 1) * R);
  q.U = (int*)take(sizeof(int) * (size_t)(L + 1));
  q.gh = (float*)take(sizeof(float) * (size_t)d.lm_layers * R * Hl);
  q.gc = (float*)take(sizeof(float) * (size_t)d.lm_layers * R * Hl);
  q.gv = (float*)take(sizeof(float) * (size_t)R * Hl);
  q.tok_c = (int64_t*)take(sizeof(int64_t) * (size_t)R);
  q.tgt_c = (int64_t*)take(sizeof(int64_t) * (size_t)R);
  // ---- prefix classes ----
  hipLaunchKernelGGL(lm_prefix_classes_kernel, dim3(G), dim3(256), 0, s, seqs, L, group, R, G,
                     rep, lrank, cnt);
  hipLaunchKernelGGL(lm_prefix_scan_kernel, dim3(L + 1), dim3(256), 0, s, cnt, G, offs, q.U);
  hipLaunchKernelGGL(lm_prefix_assign_kernel, dim3(nblk(R)), dim3(256), 0, s, rep, lrank, offs,
                     L, group, R, G, q.cls, q.urow, q.par);
  MILAN_CHECK_HIP(hipGetLastError());
  // ---- zero state in front of step 0 (its classes have no parent state to gather) ----
  const size_t st_floats = (size_t)d.lm_layers * R * Hl;
  MILAN_TRY(launch_zero_fill(q.gh, sizeof(float) * st_floats, s));
  MILAN_TRY(launch_zero_fill(q.gc, sizeof(float) * st_floats, s));
  float* hs = b->lm_gates;           // split hidden states [2 slots][layers][R][Hl]
  float* emb_s = c->scratch;
  const size_t layer_sz = (size_t)R * Hl;
  int cur = 0;
  for (int t = 0; t + 1 < L; ++t) {
    const int p1 = t + 1, p2 = t + 2;
    const int* U1 = q.U + p1;
    const int* U2 = q.U + p2;
    LmState& st = b->lm[cur];
    LmState& nx = b->lm[cur ^ 1];
    // token of every class of seqs[..t], its embedding in split form
    hipLaunchKernelGGL(lm_gather_tok_kernel, dim3(nblk(R)), dim3(256), 0, s, seqs, (long)L, t,
                       q.urow + (size_t)p1 * R, U1, q.tok_c);
    hipLaunchKernelGGL(embed_split_kernel, dim3(nblk((long)R * (El / 8))), dim3(256), 0, s,
                       c->lm_embedding, q.tok_c, R, El, emb_s, El, U1);
    if (t > 0) {
      // the parent class's state of every layer: h (split form) and c
      for (int l = 0; l < d.lm_layers; ++l) {
        hipLaunchKernelGGL(lm_gather_rows_kernel, dim3(nblk((long)R * (Hl / 4), 8192)), dim3(256),
                           0, s, (const float4*)(hs + ((size_t)cur * d.lm_layers + l) * layer_sz),
                           q.par + (size_t)p1 * R, U1, Hl / 4, (float4*)(q.gh + l * layer_sz));
        hipLaunchKernelGGL(lm_gather_rows_kernel, dim3(nblk((long)R * (Hl / 4), 8192)), dim3(256),
                           0, s, (const float4*)(st.c + (long)l * st.rows * Hl),
                           q.par + (size_t)p1 * R, U1, Hl / 4, (float4*)(q.gc + l * layer_sz));
      }
    }
    const float* in = emb_s;
    int in_dim = El;
    for (int l = 0; l < d.lm_layers; ++l) {
      float* h_next = hs + ((size_t)(cur ^ 1) * d.lm_layers + l) * layer_sz;
      float* hn = nx.h + (long)l * nx.rows * Hl;
      float* cn = nx.c + (long)l * nx.rows * Hl;
      const LinearW& w = c->lm_cat[l];
      GemmArgs g = linear_args(in, in_dim, w.ws, w.b, hn, Hl, R, w.n, w.k, EPI_LSTM, c->zero,
                               q.gc + l * layer_sz, Hl);
      g.C2 = cn; g.Cs = h_next;
      g.a_split = 1;
      g.acc_scale = w.ws_inv;
      g.Cin = in_dim;
      g.A2 = q.gh + l * layer_sz; g.K1 = in_dim; g.H2 = 1; g.W2d = 1; g.stride2 = 1;
      g.a2_pix_stride = Hl; g.a2_img_stride = Hl;
      g.m_live = U1; g.m_live_mul = 1;
      MILAN_TRY(launch_gemm(g, s));
      in = h_next;
      in_dim = Hl;
    }
    // vocabulary product over the classes of seqs[..t + 1]: A = the parent's top-layer state,
    // one target (the class's last token) per row
    hipLaunchKernelGGL(lm_gather_rows_kernel, dim3(nblk((long)R * (Hl / 4), 8192)), dim3(256), 0, s,
                       (const float4*)in, q.par + (size_t)p2 * R, U2, Hl / 4, (float4*)q.gv);
    hipLaunchKernelGGL(lm_gather_tok_kernel, dim3(nblk(R)), dim3(256), 0, s, seqs, (long)L, t + 1,
                       q.urow + (size_t)p2 * R, U2, q.tgt_c);
    {
      GemmArgs g = linear_args(q.gv, Hl, c->lm_out.ws, c->lm_out.b, b->lm_logits, V, R,
                               c->lm_out.n, c->lm_out.k, EPI_BIAS, c->zero);
      g.a_split = 1;
      g.acc_scale = c->lm_out.ws_inv;
      g.epilogue = EPI_LSE;
      g.ldc = 2 * nb;
      g.lse_tgt = reinterpret_cast<const long long*>(q.tgt_c);
      g.lse_tgt_stride = 1;
      g.lse_x = b->lm_logits + (size_t)R * 2 * nb;
      g.m_live = U2; g.m_live_mul = 1;
      MILAN_TRY(launch_gemm(g, s));
    }
    hipLaunchKernelGGL(lm_accumulate_lse_kernel, dim3((R + 3) / 4), dim3(256), 0, s,
                       b->lm_logits, nb, b->lm_logits + (size_t)R * 2 * nb, R, seqs, (long)L, t,
                       seq_len, len_div, d.stop_index, b->lm_alive, total,
                       q.cls + (size_t)p2 * R);
    cur ^= 1;
  }
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

// LanguageModel.forward(reduce=True), lms.py:58-101.  seq_len (device, may be
// null) is indexed by row / len_div.  `group` > 1 with `dedup_scratch`: rows come in groups
// (the beams of a neuron) whose shared prefixes are multiplied once (lm_score_dedup).
static int lm_score_impl(milan_ctx* c, const int64_t* seqs, int rows, int L,
                         const int32_t* seq_len, int len_div, float* total,
                         DecBuf* b, hipStream_t s, int group = 1,
                         float* dedup_scratch = nullptr, size_t dedup_floats = 0) {
  const milan_dims& d = c->d;
  MILAN_REQUIRE(d.has_lm && c->lm_out.w, MILAN_ERR_NO_LM,
                "cannot use MI/rerank decoding without an LM");
  const size_t st_bytes =
      sizeof(float) * (size_t)b->lm[0].rows * d.lm_hidden_size * d.lm_layers;
  MILAN_TRY(launch_zero_fill(b->lm[0].h, st_bytes, s));
  MILAN_TRY(launch_zero_fill(b->lm[0].c, st_bytes, s));
  if (L < 2) {
    MILAN_TRY(launch_zero_fill(total, sizeof(float) * rows, s));
    return 0;
  }
  int cur = 0;
  // split hidden states live in the gate matrix, which the fused cell never writes
  const bool split = lm_split_ok(c, rows) && (long)rows == b->lm[0].rows;
  // MILAN_LM_LSE=0: logits through HBM + lm_accumulate_kernel (A/B timing)
  static const bool lse_on = !(getenv("MILAN_LM_LSE") && atoi(getenv("MILAN_LM_LSE")) == 0);
  const bool lse = lse_on && d.vocab_size % 4 == 0 && d.vocab_size >= 256;
  if (split)
    MILAN_TRY(launch_zero_fill(b->lm_gates, sizeof(float) * (size_t)rows * d.lm_hidden_size * d.lm_layers, s));
  // MILAN_LM_DEDUP=0: every row through the LM, as the reference does (A/B timing)
  static const bool dedup_on = !(getenv("MILAN_LM_DEDUP") && atoi(getenv("MILAN_LM_DEDUP")) == 0);
  if (dedup_on && split && lse && group > 1 && group <= 256 && rows % group == 0 && dedup_scratch &&
      d.lm_hidden_size % 4 == 0 &&
      lm_dedup_floats(d, (size_t)rows, L, rows / group) <= dedup_floats)
    return lm_score_dedup(c, seqs, rows, L, group, seq_len, len_div, total, b, dedup_scratch, s);
  for (int t = 0; t + 1 < L; ++t) {
    hipLaunchKernelGGL(gather_col_kernel, dim3(nblk(rows)), dim3(256), 0, s, seqs,
                       (long)rows, (long)L, t, b->tok);
    if (split && lse) {
      // the vocabulary GEMM leaves the log-softmax statistics of every row, not the logits
      const int nb = (d.vocab_size + 63) / 64;
      MILAN_TRY(lm_step_split(c, b->tok, rows, b->lm[cur], b->lm[cur ^ 1],
                              b->lm_gates, cur, b->lm_logits, s, seqs + t + 1, (long)L));
      hipLaunchKernelGGL(lm_accumulate_lse_kernel, dim3((rows + 3) / 4), dim3(256), 0, s,
                         b->lm_logits, nb, b->lm_logits + (size_t)rows * 2 * nb, rows, seqs,
                         (long)L, t, seq_len, len_div, d.stop_index, b->lm_alive, total);
      cur ^= 1;
      continue;
    }
    if (split)
      MILAN_TRY(lm_step_split(c, b->tok, rows, b->lm[cur], b->lm[cur ^ 1],
                              b->lm_gates, cur, b->lm_logits, s));
    else
      MILAN_TRY(lm_step(c, b->tok, rows, b->lm[cur], b->lm[cur ^ 1], b->lm_emb,
                        b->lm_gates, b->lm_logits, s));
    hipLaunchKernelGGL(lm_accumulate_kernel, dim3(rows), dim3(256), 0, s,
                       b->lm_logits, d.vocab_size, seqs, (long)L, t, seq_len,
                       len_div, d.stop_index, b->lm_alive, total);
    cur ^= 1;
  }
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

int decoder_lm_score(milan_ctx* c, const int64_t* seqs, int rows, int L,
                     const int32_t* seq_len, float* out, Arena& ws,
                     hipStream_t s) {
  MILAN_REQUIRE(c->finalized, MILAN_ERR_STATE, "weights not finalized");
  MILAN_REQUIRE(c->d.has_lm && c->lm_out.w, MILAN_ERR_NO_LM,
                "cannot use MI/rerank decoding without an LM");
  MILAN_REQUIRE(rows > 0 && L >= 1, MILAN_ERR_SHAPE, "lm_score: empty input");
  DecBuf b;
  dec_plan(c, rows, 1, 1, 1, true, ws, &b);
  c->scratch = ws.off <= ws.size ? b.scratch : nullptr;
  c->scratch_floats = b.scratch_floats;
  MILAN_REQUIRE(ws.off <= ws.size, MILAN_ERR_WORKSPACE,
                "lm_score: workspace too small (%zu needed)", ws.off);
  return lm_score_impl(c, seqs, rows, L, seq_len, 1, out, &b, s);
}

// LanguageModel.forward(inputs, reduce=False): log-probs at every position.
int decoder_lm_logprobs(milan_ctx* c, const int64_t* seqs, int rows, int L,
                        float* out, Arena& ws, hipStream_t s) {
  MILAN_REQUIRE(c->finalized, MILAN_ERR_STATE, "weights not finalized");
  MILAN_REQUIRE(c->d.has_lm && c->lm_out.w, MILAN_ERR_NO_LM,
                "cannot use MI/rerank decoding without an LM");
  MILAN_REQUIRE(rows > 0 && L >= 1, MILAN_ERR_SHAPE, "lm_logprobs: empty input");
  const milan_dims& d = c->d;
  DecBuf b;
  dec_plan(c, rows, 1, 1, 1, true, ws, &b);
  c->scratch = ws.off <= ws.size ? b.scratch : nullptr;
  c->scratch_floats = b.scratch_floats;
  MILAN_REQUIRE(ws.off <= ws.size, MILAN_ERR_WORKSPACE,
                "lm_logprobs: workspace too small (%zu needed)", ws.off);
  const size_t st_bytes =
      sizeof(float) * (size_t)b.lm[0].rows * d.lm_hidden_size * d.lm_layers;
  MILAN_TRY(launch_zero_fill(b.lm[0].h, st_bytes, s));
  MILAN_TRY(launch_zero_fill(b.lm[0].c, st_bytes, s));
  int cur = 0;
  for (int t = 0; t < L; ++t) {
    hipLaunchKernelGGL(gather_col_kernel, dim3(nblk(rows)), dim3(256), 0, s, seqs,
                       (long)rows, (long)L, t, b.tok);
    MILAN_TRY(lm_step(c, b.tok, rows, b.lm[cur], b.lm[cur ^ 1], b.lm_emb,
                      b.lm_gates, b.lm_logits, s));
    // log-softmax of position t into out[:, t, :]
    MILAN_TRY(launch_row_select(b.lm_logits, nullptr, 0.f, rows, d.vocab_size, 0,
                                nullptr, 0, nullptr, nullptr,
                                out + (long)t * d.vocab_size,
                                (long)L * d.vocab_size, s));
    cur ^= 1;
  }
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

int decoder_decode(milan_ctx* c, const float* features, int n, int k,
                   int strategy, int length, int beam, int mi, float temperature,
                   int group_size, int64_t* tokens, float* scores,
                   float* predictions, float* attentions, int64_t* beam_tokens,
                   float* beam_scores, int32_t* out_len, Arena& ws,
                   hipStream_t s) {
  MILAN_TRY(check_dims(c, k));
  const milan_dims& d = c->d;
  const int H = d.hidden_size, V = d.vocab_size;
  MILAN_REQUIRE(strategy == MILAN_GREEDY || strategy == MILAN_FORCED ||
                    strategy == MILAN_BEAM || strategy == MILAN_RERANK,
                MILAN_ERR_ARG, "unknown strategy: %d", strategy);
  MILAN_REQUIRE(!(mi && strategy == MILAN_RERANK), MILAN_ERR_ARG,
                "cannot set `mi=` decoding when reranking");
  MILAN_REQUIRE(!(mi || strategy == MILAN_RERANK) || d.has_lm, MILAN_ERR_NO_LM,
                "cannot use MI/rerank decoding without an LM");
  MILAN_REQUIRE(n > 0 && length > 0, MILAN_ERR_SHAPE, "decode: empty batch");
  const bool forced = strategy == MILAN_FORCED;
  const bool greedy = strategy == MILAN_GREEDY || forced;
  if (greedy) beam = 1;
  MILAN_REQUIRE(!forced || predictions, MILAN_ERR_ARG,
                "teacher forcing needs the predictions output");
  MILAN_REQUIRE(beam >= 1 && beam <= V, MILAN_ERR_ARG,
                "beam_size %d must be in 1..vocab_size", beam);
  MILAN_REQUIRE(greedy || (beam_tokens && beam_scores), MILAN_ERR_ARG,
                "beam search needs beam_tokens and beam_scores outputs");
  MILAN_REQUIRE(tokens && scores, MILAN_ERR_ARG, "tokens/scores outputs required");
  if (group_size <= 0) group_size = n;
  const bool need_lm = mi || strategy == MILAN_RERANK;
  DecBuf b;
  dec_plan(c, n, k, beam, length, need_lm, ws, &b);
  c->scratch = ws.off <= ws.size ? b.scratch : nullptr;
  c->scratch_floats = b.scratch_floats;
  MILAN_REQUIRE(ws.off <= ws.size, MILAN_ERR_WORKSPACE,
                "decode: workspace too small (%zu needed, %zu given)", ws.off,
                ws.size);
  const int R = n * beam;
  const int Hl = d.lm_hidden_size;

  {
    StageScope scope(MILAN_STAGE_DEC_INIT, s);
    MILAN_TRY(project_keys(c, features, n * k, b.keys, s));
    MILAN_TRY(init_state_impl(c, features, n, k, b.pooled, b.h, b.cc, s));
  }
  hipLaunchKernelGGL(fill_i64_kernel, dim3(nblk(n)), dim3(256), 0, s, b.tok,
                     (long)n, (int64_t)d.start_index);
  int lmcur = 0;
  if (mi) {
    const size_t st_bytes = sizeof(float) * (size_t)R * Hl * d.lm_layers;
    MILAN_TRY(launch_zero_fill(b.lm[0].h, st_bytes, s));
    MILAN_TRY(launch_zero_fill(b.lm[0].c, st_bytes, s));
  }

  if (greedy) {
    StageScope scope(MILAN_STAGE_DEC_SEARCH, s);
    float *h = b.h, *cc = b.cc, *hn = b.hn, *cn = b.cn;
    for (int t = 0; t < length; ++t) {
      MILAN_TRY(step_core(c, features, b.keys, n, 1, k, b.tok, h, cc, hn, cn, &b, s));
      const float* lm_logits = nullptr;
      if (mi) {
        MILAN_TRY(lm_step(c, b.tok, n, b.lm[lmcur], b.lm[lmcur ^ 1], b.lm_emb,
                          b.lm_gates, b.lm_logits, s));
        lmcur ^= 1;
        lm_logits = b.lm_logits;
      }
      MILAN_TRY(launch_row_select(
          b.logits, lm_logits, temperature, n, V, 1, nullptr, 0, b.cand_v,
          b.cand_i, predictions ? predictions + (long)t * V : nullptr,
          (long)length * V, s));
      if (forced)
        hipLaunchKernelGGL(forced_record_kernel, dim3(nblk(n)), dim3(256), 0, s,
                           predictions + (long)t * V, (long)length * V, V, b.att,
                           n, k, t, length, tokens, scores, attentions, b.tok);
      else
        hipLaunchKernelGGL(greedy_record_kernel, dim3(nblk(n)), dim3(256), 0, s,
                           b.cand_v, b.cand_i, b.att, n, k, t, length, tokens,
                           scores, attentions, b.tok);
      float* tmp = h; h = hn; hn = tmp;
      tmp = cc; cc = cn; cn = tmp;
    }
    MILAN_CHECK_HIP(hipGetLastError());
    return 0;
  }

  // ---- beam search (allennlp 2.10 semantics, fixed `length` steps) ----------
  const size_t merge_lds = sizeof(float) * ((size_t)beam * beam + 2 * beam + 16 + 528);
  MILAN_REQUIRE(merge_lds <= 64 * 1024, MILAN_ERR_ARG,
                "beam_size %d too large for the merge kernel", beam);
  int beam_prev = 1, rows = n, lpcur = 0;
  std::optional<StageScope> search_scope;
  search_scope.emplace(MILAN_STAGE_DEC_SEARCH, s);
  for (int t = 0; t < length; ++t) {
    MILAN_TRY(step_core(c, features, b.keys, rows, beam_prev, k, b.tok, b.h, b.cc,
                        b.hn, b.cn, &b, s));
    const float* lm_logits = nullptr;
    if (mi) {
      b.lm[lmcur].rows = b.lm[lmcur ^ 1].rows = R;
      MILAN_TRY(lm_step(c, b.tok, rows, b.lm[lmcur], b.lm[lmcur ^ 1], b.lm_emb,
                        b.lm_gates, b.lm_logits, s));
      lm_logits = b.lm_logits;
    }
    MILAN_TRY(launch_row_select(b.logits, lm_logits, temperature, rows, V, beam,
                                t == 0 ? nullptr : b.tok, d.stop_index, b.cand_v,
                                b.cand_i, nullptr, 0, s));
    hipLaunchKernelGGL(beam_merge_kernel, dim3(n), dim3(256), merge_lds, s,
                       b.cand_v, b.cand_i, t == 0 ? nullptr : b.last_lp[lpcur],
                       beam_prev, beam, b.last_lp[lpcur ^ 1], b.new_tok, b.new_bp);
    lpcur ^= 1;
    // reorder: (hn,cn)[src] -> (h,cc)[r]; LM new state -> the other LM buffer
    hipLaunchKernelGGL(
        beam_reorder_kernel, dim3(R), dim3(128), 0, s, b.new_tok, b.new_bp, n,
        beam_prev, beam, H, b.hn, b.cn, b.h, b.cc, mi ? d.lm_layers : 0, Hl,
        (long)R, (long)R, mi ? b.lm[lmcur ^ 1].h : nullptr,
        mi ? b.lm[lmcur ^ 1].c : nullptr, mi ? b.lm[lmcur].h : nullptr,
        mi ? b.lm[lmcur].c : nullptr, b.tok, b.hist_tok + (long)t * R,
        b.hist_bp + (long)t * R);
    beam_prev = beam;
    rows = R;
  }
  hipLaunchKernelGGL(beam_finalize_kernel, dim3(nblk(R)), dim3(256), 0, s,
                     b.hist_tok, b.hist_bp, n, beam, length, beam_tokens);
  MILAN_CHECK_HIP(hipMemcpyAsync(beam_scores, b.last_lp[lpcur],
                                 sizeof(float) * R, hipMemcpyDeviceToDevice, s));
  const int groups = (n + group_size - 1) / group_size;
  int32_t* lens = out_len ? out_len : b.len1 + n + 1;  // scratch if not wanted
  hipLaunchKernelGGL(group_len_kernel, dim3(groups), dim3(256), 0, s, b.hist_tok,
                     n, beam, length, group_size, d.stop_index, lens);
  search_scope.reset();
  StageScope lm_scope(MILAN_STAGE_DEC_LM, s);
  const float* lm_scores = nullptr;
  if (strategy == MILAN_RERANK) {
    hipLaunchKernelGGL(build_lm_seqs_kernel, dim3(nblk((long)R * (length + 1))),
                       dim3(256), 0, s, beam_tokens, (long)R, length,
                       (int64_t)d.start_index, b.seqs);
    hipLaunchKernelGGL(len_plus_one_kernel, dim3(nblk(groups)), dim3(256), 0, s,
                       lens, groups, b.len1);
    b.lm[0].rows = b.lm[1].rows = R;
    MILAN_TRY(lm_score_impl(c, b.seqs, R, length + 1, b.len1, beam * group_size,
                            b.lm_total, &b, s, beam, b.lm_dedup, b.lm_dedup_floats));
    lm_scores = b.lm_total;
  }
  hipLaunchKernelGGL(rerank_select_kernel, dim3(nblk(n)), dim3(256), 0, s,
                     beam_scores, lm_scores, temperature, n, beam, length,
                     beam_tokens, tokens, scores);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace milan
