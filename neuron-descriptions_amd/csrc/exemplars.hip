// Exemplar computation kernels: the stage that
// produces the images.npy / masks.npy the MILAN path consumes.
//
// Mirrors src/exemplars/compute.py:27-246 and its vendored netdissect pieces:
//   RunningTopK      src/deps/netdissect/runningstats.py:31-151
//   RunningQuantile  src/deps/netdissect/runningstats.py:274-627 (KLL sketch)
//   ImageVisualizer  src/deps/netdissect/imgviz.py:185-210, upsample.py:6-156,
//                    src/deps/ext/netdissect/imgviz.py:56-81
// The model that produces the activations stays the caller's (the reference
// takes it as two black-box functions); everything downstream of the
// activation tensor runs here.  All of it is HBM/latency-bound integer, compare
// and byte work -- no MFMA: coalesced streaming, LDS sorts, one pass per datum.
// The sketch's control flow (which level is compacted when, which random bit
// is used) is host logic in milan_amd/exemplars.py, exactly as it is Python in
// the reference; the kernels below are its tensor operations.
#include "common.h"

#include <cmath>
#include <mutex>
#include <algorithm>

#include <vector>
#include <hipcub/hipcub.hpp>

namespace milan {

// ---------------------------------------------------------------------------
// tally: spatial max + running top-k
// ---------------------------------------------------------------------------
// pooled[u][b] = max_s hiddens[b][unit(u)][s]     (compute.py:331)
// one wave per (b, u)
__global__ __launch_bounds__(256) void exemplar_pool_kernel(
    const float* __restrict__ hid, int batch, int channels, int hw,
    const int32_t* __restrict__ units, int n_units, float* __restrict__ pooled) {
  const long item = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (item >= (long)batch * n_units) return;
  const int b = item / n_units, u = item - (long)b * n_units;
  const int ch = units ? units[u] : u;
  const float* x = hid + ((long)b * channels + ch) * hw;
  float m = -INFINITY;
  for (int s = lane; s < hw; s += 64) m = fmaxf(m, x[s]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) pooled[(long)u * batch + b] = m;
}

// RunningTopK.add + result: merge the batch into the per-unit list, sorted by
// value descending, equal values by dataset index ascending.  One workgroup
// per unit, bitonic sort of (filled + batch) <= 2048 candidates in LDS.
__global__ __launch_bounds__(256) void exemplar_topk_merge_kernel(
    const float* __restrict__ pooled, int batch, int64_t first_index, int k,
    int filled, float* __restrict__ top_values, int64_t* __restrict__ top_index) {
  __shared__ float v[2048];
  __shared__ int64_t ix[2048];
  const int u = blockIdx.x, tid = threadIdx.x;
  const int n = filled + batch;
  int P = 2;
  while (P < n) P <<= 1;
  for (int i = tid; i < P; i += 256) {
    if (i < filled) {
      v[i] = top_values[(long)u * k + i];
      ix[i] = top_index[(long)u * k + i];
    } else if (i < n) {
      v[i] = pooled[(long)u * batch + (i - filled)];
      ix[i] = first_index + (i - filled);
    } else {
      v[i] = -INFINITY;
      ix[i] = 0x7fffffffffffffffLL;
    }
  }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < P; t += 256) {
        const int partner = t ^ stride;
        if (partner > t) {
          const float av = v[t], bv = v[partner];
          const int64_t ai = ix[t], bi = ix[partner];
          const bool a_first = av > bv || (av == bv && ai < bi);
          const bool up = (t & size) == 0;
          if (up ? !a_first : a_first) {
            v[t] = bv; ix[t] = bi; v[partner] = av; ix[partner] = ai;
          }
        }
      }
      __syncthreads();
    }
  const int keep = n < k ? n : k;
  for (int i = tid; i < keep; i += 256) {
    top_values[(long)u * k + i] = v[i];
    top_index[(long)u * k + i] = ix[i];
  }
}

// ---------------------------------------------------------------------------
// KLL sketch
// ---------------------------------------------------------------------------
// level0[u][column + i] = activations[first + i][unit(u)], where activation row
// p is position (img, y, x) of hiddens.permute(0,2,3,1).reshape(-1, C)
// (compute.py:329-330, runningstats.py:380-383).
__global__ void sketch_append_kernel(const float* __restrict__ hid, int channels,
                                     int hw, const int32_t* __restrict__ units,
                                     long first, long count,
                                     float* __restrict__ level0, long capacity,
                                     long column) {
  const int u = blockIdx.y;
  const int ch = units ? units[u] : u;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count;
       i += (long)gridDim.x * blockDim.x) {
    const long p = first + i;
    const long img = p / hw;
    const int s = p - img * hw;
    level0[(long)u * capacity + column + i] = hid[(img * channels + ch) * hw + s];
  }
}

__global__ void segment_offsets_kernel(int n_units, long stride, long n,
                                       int* __restrict__ begin,
                                       int* __restrict__ end) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u < n_units) {
    begin[u] = (int)(u * stride);
    end[u] = (int)(u * stride + n);
  }
}

// dst[u][position + j] = sorted[u][offset + 2 j]; extremes from the sorted row
// (runningstats.py:395-404 / :511-516, :415-419).
__global__ void sketch_decimate_kernel(const float* __restrict__ sorted,
                                       long stride, long n, int offset,
                                       float* __restrict__ dst, long dst_capacity,
                                       long position,
                                       float* __restrict__ extremes) {
  const int u = blockIdx.y;
  const long m = (n - offset + 1) / 2;
  for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < m;
       j += (long)gridDim.x * blockDim.x)
    dst[(long)u * dst_capacity + position + j] =
        sorted[(long)u * stride + offset + 2 * j];
  if (extremes && blockIdx.x == 0 && threadIdx.x == 0 && n > 0) {
    extremes[2 * u] = fminf(extremes[2 * u], sorted[(long)u * stride]);
    extremes[2 * u + 1] =
        fmaxf(extremes[2 * u + 1], sorted[(long)u * stride + n - 1]);
  }
}

// Fused compaction for rows of up to 8192 samples (every level of the r = 4096
// sketch the reference's tally builds): one workgroup per unit sorts its row in
// LDS (bitonic, 32 KB), writes every second element from `offset` on to dst and
// folds the row's ends into the extremes.  One launch instead of a copy, a
// library sort and a gather; dst may alias src (the row is in LDS by then).
constexpr int kSortCap = 8192;
__global__ __launch_bounds__(1024) void sketch_sort_decimate_kernel(
    const float* __restrict__ src, long src_capacity, int n, int offset,
    float* dst, long dst_capacity, long position, float* __restrict__ extremes) {
  __shared__ float v[kSortCap];
  const int u = blockIdx.x, tid = threadIdx.x;
  int P = 2;
  while (P < n) P <<= 1;
  for (int i = tid; i < P; i += 1024)
    v[i] = i < n ? src[(long)u * src_capacity + i] : INFINITY;
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (P >> 1); t += 1024) {
        // t-th compare-exchange of this pass: lower index of the pair
        const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
        const int hi = lo | stride;
        const float a = v[lo], b = v[hi];
        const bool up = (lo & size) == 0;  // ascending run
        if (up ? (a > b) : (a < b)) { v[lo] = b; v[hi] = a; }
      }
      __syncthreads();
    }
  const int m = (n - offset + 1) / 2;
  for (int j = tid; j < m; j += 1024)
    dst[(long)u * dst_capacity + position + j] = v[offset + 2 * j];
  if (extremes && tid == 0) {
    extremes[2 * u] = fminf(extremes[2 * u], v[0]);
    extremes[2 * u + 1] = fmaxf(extremes[2 * u + 1], v[n - 1]);
  }
}

// ---------------------------------------------------------------------------
// Bulk add: the sketch's state machine (which level is compacted when, over how
// many samples, with which random bit) depends on buffer sizes only, never on
// the data.  milan_exemplar_sketch_add therefore plays it forward on the host
// for a whole batch and executes the result level by level: level l sees a
// STREAM S_l = [what the level buffer held | what level l-1 emitted, in order],
// its compactions are consecutive ranges of that stream, all independent -- one
// launch per level, one workgroup per (range, unit) -- and what is left of the
// stream after the last range is the level's new buffer content.  A conv1-sized
// batch (400 k samples per unit, ~5000 compactions) is ~15 launches instead of
// ~10 000.
// ---------------------------------------------------------------------------
struct SketchOp { int start, n, offset, outpos; };

struct SketchSource {  // S_0's tail: activation rows of the batch
  const float* hid; const int32_t* units; int channels, hw; long first;
};

template <bool SOURCE>
__device__ __forceinline__ float sketch_stream_at(
    const float* __restrict__ init, long init_capacity, int init_len,
    const float* __restrict__ in, long in_stride, const SketchSource& src, int u,
    long s) {
  if (s < init_len) return init[(long)u * init_capacity + s];
  long p = s - init_len;
  if (SOURCE) {
    p += src.first;
    const long img = p / src.hw;
    const int sp = (int)(p - img * src.hw);
    const int ch = src.units ? src.units[u] : u;
    return src.hid[(img * src.channels + ch) * src.hw + sp];
  }
  return in[(long)u * in_stride + p];
}

template <bool SOURCE>
__global__ __launch_bounds__(1024) void sketch_level_kernel(
    const float* __restrict__ init, long init_capacity, int init_len,
    const float* __restrict__ in, long in_stride, SketchSource src,
    const SketchOp* __restrict__ ops, float* __restrict__ out, long out_stride) {
  extern __shared__ float sv[];
  const SketchOp op = ops[blockIdx.x];
  const int u = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
  int P = 2;
  while (P < op.n) P <<= 1;
  for (int i = tid; i < P; i += nt)
    sv[i] = i < op.n ? sketch_stream_at<SOURCE>(init, init_capacity, init_len, in,
                                                in_stride, src, u,
                                                (long)op.start + i)
                     : INFINITY;
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (P >> 1); t += nt) {
        const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
        const int hi = lo | stride;
        const float a = sv[lo], b = sv[hi];
        const bool up = (lo & size) == 0;
        if (up ? (a > b) : (a < b)) { sv[lo] = b; sv[hi] = a; }
      }
      __syncthreads();
    }
  const int m = (op.n - op.offset + 1) / 2;
  for (int j = tid; j < m; j += nt)
    out[(long)u * out_stride + op.outpos + j] = sv[op.offset + 2 * j];
}

// extremes over S_0[0, n): everything that went through a level-0 compaction
__global__ __launch_bounds__(256) void sketch_stream_extremes_kernel(
    const float* __restrict__ init, long init_capacity, int init_len,
    SketchSource src, long n, float* __restrict__ extremes) {
  __shared__ float lo[4], hi[4];
  const int u = blockIdx.x, tid = threadIdx.x;
  float a = INFINITY, b = -INFINITY;
  for (long i = tid; i < n; i += 256) {
    const float x = sketch_stream_at<true>(init, init_capacity, init_len, nullptr,
                                           0, src, u, i);
    a = fminf(a, x); b = fmaxf(b, x);
  }
  for (int o = 32; o > 0; o >>= 1) {
    a = fminf(a, __shfl_xor(a, o));
    b = fmaxf(b, __shfl_xor(b, o));
  }
  if ((tid & 63) == 0) { lo[tid >> 6] = a; hi[tid >> 6] = b; }
  __syncthreads();
  if (tid == 0) {
    a = fminf(fminf(lo[0], lo[1]), fminf(lo[2], lo[3]));
    b = fmaxf(fmaxf(hi[0], hi[1]), fmaxf(hi[2], hi[3]));
    extremes[2 * u] = fminf(extremes[2 * u], a);
    extremes[2 * u + 1] = fmaxf(extremes[2 * u + 1], b);
  }
}

// min / max of the first n columns of every row (runningstats.py:409-413)
__global__ __launch_bounds__(256) void sketch_scan_extremes_kernel(
    const float* __restrict__ level, long capacity, long n,
    float* __restrict__ extremes) {
  __shared__ float lo[4], hi[4];
  const int u = blockIdx.x, tid = threadIdx.x;
  float a = INFINITY, b = -INFINITY;
  for (long i = tid; i < n; i += 256) {
    const float x = level[(long)u * capacity + i];
    a = fminf(a, x); b = fmaxf(b, x);
  }
  for (int o = 32; o > 0; o >>= 1) {
    a = fminf(a, __shfl_xor(a, o));
    b = fmaxf(b, __shfl_xor(b, o));
  }
  if ((tid & 63) == 0) { lo[tid >> 6] = a; hi[tid >> 6] = b; }
  __syncthreads();
  if (tid == 0) {
    a = fminf(fminf(lo[0], lo[1]), fminf(lo[2], lo[3]));
    b = fmaxf(fmaxf(hi[0], hi[1]), fmaxf(hi[2], hi[3]));
    extremes[2 * u] = fminf(extremes[2 * u], a);
    extremes[2 * u + 1] = fmaxf(extremes[2 * u + 1], b);
  }
}

// summary / weights rows of _weighted_summary (runningstats.py:531-548)
__global__ void sketch_gather_kernel(const float* __restrict__ level,
                                     long capacity, long n, float weight,
                                     float* __restrict__ keys,
                                     float* __restrict__ weights, long total,
                                     long at) {
  const int u = blockIdx.y;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x) {
    keys[(long)u * total + at + i] = level[(long)u * capacity + i];
    weights[(long)u * total + at + i] = weight;
  }
}

// quantiles(q) for one q (runningstats.py:557-580): with the sorted summary
// [min, x_0..x_{n-1}, max] and weights [0, w.., 0],
//   cum_i = (sum_{j<=i} w_j - w_i / 2) / sum w      (float32, as torch)
//   result = numpy.interp(q, cum, summary)           (float64, cast to float32)
// One thread per unit walks its row once; the float32 sums are exact (integer
// weights) while the total stays below 2^24.
__global__ void sketch_quantile_kernel(const float* __restrict__ keys,
                                       const float* __restrict__ weights,
                                       long total, int n_units,
                                       const float* __restrict__ extremes,
                                       float q, float* __restrict__ out) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_units) return;
  const float* x = keys + (long)u * total;
  const float* w = weights + (long)u * total;
  float sum = 0.f;
  for (long i = 0; i < total; ++i) sum = __fadd_rn(sum, w[i]);
  const double qd = (double)q;
  // left neighbour: the extreme minimum at cum = 0 / sum
  double xp0 = (double)__fdiv_rn(0.f, sum), fp0 = (double)extremes[2 * u];
  float run = 0.f;
  double result = 0.0;
  bool done = false;
  for (long i = 0; i <= total && !done; ++i) {
    float cw, val;
    if (i < total) {
      run = __fadd_rn(run, w[i]);
      cw = __fdiv_rn(__fsub_rn(run, __fmul_rn(w[i], 0.5f)), sum);
      val = x[i];
    } else {  // the extreme maximum at cum = sum / sum
      cw = __fdiv_rn(run, sum);
      val = extremes[2 * u + 1];
    }
    const double xp1 = (double)cw, fp1 = (double)val;
    if (xp1 > qd) {
      // numpy: j = largest index with xp[j] <= q
      if (xp0 == qd) {
        result = fp0;
      } else {
        const double slope = __ddiv_rn(__dsub_rn(fp1, fp0), __dsub_rn(xp1, xp0));
        result = __dadd_rn(__dmul_rn(slope, __dsub_rn(qd, xp0)), fp0);
        if (isnan(result)) {
          result = __dadd_rn(__dmul_rn(slope, __dsub_rn(qd, xp1)), fp1);
          if (isnan(result) && fp0 == fp1) result = fp0;
        }
      }
      done = true;
    }
    xp0 = xp1; fp0 = fp1;
  }
  if (!done) result = fp0;  // q >= 1: the right end
  out[u] = (float)result;
}

// ---------------------------------------------------------------------------
// rendering: mask, image, masked image per (unit, rank) cell
// ---------------------------------------------------------------------------
struct RenderGeom {
  int h, w;            // activation map
  int img_h, img_w;    // dataset image
  int out;             // output_size (square)
  float off_y, mul_y;  // upsample_grid: g = (i - off) * mul - 1   (float32)
  float off_x, mul_x;
  float half_h, half_w;  // (h - 1) / 2, (w - 1) / 2
  float scale_y, scale_x;  // nearest resize: src = floor(dst * scale)
  float mul[3], add[3];    // byte renormalisation (renormalize.py:119-136)
};

// torch's vectorised CPU grid_sample (bilinear, zeros padding, align_corners):
// every product and sum rounded on its own, no contraction.
__device__ inline float bilinear_at(const float* __restrict__ a, const RenderGeom& g,
                                    int oy, int ox) {
  const float gy = __fsub_rn(__fmul_rn(__fsub_rn((float)oy, g.off_y), g.mul_y), 1.f);
  const float gx = __fsub_rn(__fmul_rn(__fsub_rn((float)ox, g.off_x), g.mul_x), 1.f);
  const float y = __fmul_rn(__fadd_rn(gy, 1.f), g.half_h);
  const float x = __fmul_rn(__fadd_rn(gx, 1.f), g.half_w);
  const float xw = floorf(x), yn = floorf(y);
  const float west = __fsub_rn(x, xw), north = __fsub_rn(y, yn);
  const float east = __fsub_rn(1.f, west), south = __fsub_rn(1.f, north);
  const int x0 = (int)xw, y0 = (int)yn;
  auto at = [&](int yy, int xx) -> float {
    return (yy >= 0 && yy < g.h && xx >= 0 && xx < g.w) ? a[yy * g.w + xx] : 0.f;
  };
  float out = __fmul_rn(at(y0, x0), __fmul_rn(south, east));
  out = __fadd_rn(out, __fmul_rn(at(y0, x0 + 1), __fmul_rn(south, west)));
  out = __fadd_rn(out, __fmul_rn(at(y0 + 1, x0), __fmul_rn(north, east)));
  out = __fadd_rn(out, __fmul_rn(at(y0 + 1, x0 + 1), __fmul_rn(north, west)));
  return out;
}

__device__ inline float to_byte(float v) {  // .clamp(0, 255).byte()
  v = fminf(fmaxf(v, 0.f), 255.f);
  return truncf(v);
}

// cells[c] = (batch item, activation channel, unit slot, rank)
__global__ __launch_bounds__(256) void exemplar_render_kernel(
    const float* __restrict__ hid, int channels, const float* __restrict__ images,
    const int32_t* __restrict__ cells, const float* __restrict__ levels, int k,
    RenderGeom g, uint8_t* __restrict__ out_images, uint8_t* __restrict__ out_masks,
    uint8_t* __restrict__ out_masked) {
  const int cell = blockIdx.y;
  const int item = cells[4 * cell], ch = cells[4 * cell + 1],
            slot = cells[4 * cell + 2], rank = cells[4 * cell + 3];
  const float* a = hid + ((long)item * channels + ch) * g.h * g.w;
  const float* img = images + (long)item * 3 * g.img_h * g.img_w;
  const float level = levels[slot];
  const long base = (long)slot * k + rank;
  const int px = g.out * g.out;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < px; p += gridDim.x * 256) {
    const int oy = p / g.out, ox = p - oy * g.out;
    const bool inside = bilinear_at(a, g, oy, ox) > level;
    out_masks[base * px + p] = inside ? 1 : 0;
    int iy = (int)floorf(__fmul_rn((float)oy, g.scale_y));
    int ix = (int)floorf(__fmul_rn((float)ox, g.scale_x));
    iy = iy < g.img_h - 1 ? iy : g.img_h - 1;
    ix = ix < g.img_w - 1 ? ix : g.img_w - 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float raw = img[((long)c * g.img_h + iy) * g.img_w + ix];
      const float b = to_byte(__fadd_rn(__fmul_rn(raw, g.mul[c]), g.add[c]));
      out_images[(base * 3 + c) * px + p] = (uint8_t)b;
      const float m = inside ? b : __fmul_rn(0.25f, b);
      out_masked[(base * 3 + c) * px + p] = (uint8_t)to_byte(m);
    }
  }
}

static inline int grid1(long n, int cap = 4096) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b < cap ? b : cap));
}

}  // namespace milan

using namespace milan;

extern "C" {

int milan_exemplar_topk_update(const float* hiddens, int batch, int channels,
                               int hw, const int32_t* units, int n_units,
                               int64_t first_index, int k, int filled,
                               float* pooled_scratch, float* top_values,
                               int64_t* top_index, milan_stream stream) {
  MILAN_REQUIRE(hiddens && pooled_scratch && top_values && top_index,
                MILAN_ERR_ARG, "exemplar_topk_update: null argument");
  MILAN_REQUIRE(batch > 0 && channels > 0 && hw > 0 && n_units > 0 && k >= 1 &&
                    filled >= 0 && filled <= k,
                MILAN_ERR_SHAPE, "exemplar_topk_update: bad sizes");
  MILAN_REQUIRE(filled + batch <= 2048, MILAN_ERR_SHAPE,
                "exemplar_topk_update: k + batch = %d exceeds 2048 (feed smaller "
                "batches)", filled + batch);
  hipStream_t s = (hipStream_t)stream;
  const long items = (long)batch * n_units;
  hipLaunchKernelGGL(exemplar_pool_kernel, dim3((unsigned)((items + 3) / 4)),
                     dim3(256), 0, s, hiddens, batch, channels, hw, units, n_units,
                     pooled_scratch);
  hipLaunchKernelGGL(exemplar_topk_merge_kernel, dim3(n_units), dim3(256), 0, s,
                     pooled_scratch, batch, first_index, k, filled, top_values,
                     top_index);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

int milan_exemplar_sketch_append(const float* hiddens, int batch, int channels,
                                 int hw, const int32_t* units, int n_units,
                                 int64_t first, int64_t count, float* level0,
                                 int64_t capacity, int64_t column,
                                 milan_stream stream) {
  MILAN_REQUIRE(hiddens && level0, MILAN_ERR_ARG, "sketch_append: null argument");
  MILAN_REQUIRE(first >= 0 && count >= 0 && first + count <= (int64_t)batch * hw &&
                    column >= 0 && column + count <= capacity && n_units > 0,
                MILAN_ERR_SHAPE, "sketch_append: range outside the level buffer");
  if (count == 0) return 0;
  hipLaunchKernelGGL(sketch_append_kernel, dim3(grid1(count, 64), n_units),
                     dim3(256), 0, (hipStream_t)stream, hiddens, channels, hw,
                     units, (long)first, (long)count, level0, (long)capacity,
                     (long)column);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

// workspace: [begin | end offsets] [sorted keys (n_units x n)] [hipcub temp]
static size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

size_t milan_exemplar_sort_workspace(int n_units, int64_t n, int pairs) {
  if (n_units <= 0 || n <= 0) return 0;
  size_t temp = 0;
  const int items = (int)((int64_t)n_units * n);
  if (pairs)
    (void)hipcub::DeviceSegmentedRadixSort::SortPairs(
        nullptr, temp, (const float*)nullptr, (float*)nullptr,
        (const float*)nullptr, (float*)nullptr, items, n_units,
        (const int*)nullptr, (const int*)nullptr);
  else
    (void)hipcub::DeviceSegmentedRadixSort::SortKeys(
        nullptr, temp, (const float*)nullptr, (float*)nullptr, items, n_units,
        (const int*)nullptr, (const int*)nullptr);
  const size_t arrays = pairs ? 4 : 2;  // keys in/out (+ weights in/out)
  return 2 * align256(sizeof(int) * (size_t)n_units) +
         arrays * align256(sizeof(float) * (size_t)items) + align256(temp) + 256;
}

int milan_exemplar_sketch_compact(const float* src, int64_t src_capacity,
                                  int64_t n, int n_units, int offset, float* dst,
                                  int64_t dst_capacity, int64_t position,
                                  float* extremes, void* workspace,
                                  size_t workspace_bytes, milan_stream stream) {
  MILAN_REQUIRE(src && dst && (workspace || n <= 8192), MILAN_ERR_ARG,
                "sketch_compact: null argument");
  MILAN_REQUIRE(n > 0 && n <= src_capacity && n_units > 0 &&
                    (offset == 0 || offset == 1) && position >= 0 &&
                    position + (n - offset + 1) / 2 <= dst_capacity &&
                    (int64_t)n_units * n < (int64_t)1 << 31,
                MILAN_ERR_SHAPE, "sketch_compact: bad geometry");
  hipStream_t s = (hipStream_t)stream;
  if (n <= kSortCap) {  // every level of the reference's r = 4096 sketch
    hipLaunchKernelGGL(sketch_sort_decimate_kernel, dim3(n_units), dim3(1024), 0,
                       s, src, (long)src_capacity, (int)n, offset, dst,
                       (long)dst_capacity, (long)position, extremes);
    MILAN_CHECK_HIP(hipGetLastError());
    return 0;
  }
  MILAN_REQUIRE(workspace_bytes >= milan_exemplar_sort_workspace(n_units, n, 0),
                MILAN_ERR_WORKSPACE, "sketch_compact: workspace too small");
  char* w = (char*)workspace;
  int* begin = (int*)w; w += align256(sizeof(int) * (size_t)n_units);
  int* end = (int*)w; w += align256(sizeof(int) * (size_t)n_units);
  const size_t items = (size_t)n_units * n;
  float* packed = (float*)w; w += align256(sizeof(float) * items);
  float* sorted = (float*)w; w += align256(sizeof(float) * items);
  size_t temp = workspace_bytes - (size_t)(w - (char*)workspace);
  // rows of the level buffer -> contiguous segments of length n
  MILAN_CHECK_HIP(hipMemcpy2DAsync(packed, sizeof(float) * n, src,
                                   sizeof(float) * src_capacity, sizeof(float) * n,
                                   n_units, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(segment_offsets_kernel, dim3((n_units + 255) / 256), dim3(256),
                     0, s, n_units, (long)n, (long)n, begin, end);
  MILAN_CHECK_HIP(hipcub::DeviceSegmentedRadixSort::SortKeys(
      w, temp, packed, sorted, (int)items, n_units, begin, end, 0, 32, s));
  const long m = (n - offset + 1) / 2;
  hipLaunchKernelGGL(sketch_decimate_kernel, dim3(grid1(m, 64), n_units), dim3(256),
                     0, s, sorted, (long)n, (long)n, offset, dst,
                     (long)dst_capacity, (long)position, extremes);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

size_t milan_exemplar_sketch_add_workspace(int n_units, int64_t supplied,
                                           const int64_t* capacities,
                                           int n_levels) {
  if (n_units <= 0 || supplied <= 0 || n_levels <= 0 || !capacities) return 0;
  int64_t caps = 0, smallest = capacities[0];
  for (int l = 0; l < n_levels; ++l) {
    caps += capacities[l];
    smallest = capacities[l] < smallest ? capacities[l] : smallest;
  }
  smallest = smallest < 1 ? 1 : smallest;
  // streams: level l+1 receives at most half of what level l saw, plus one
  // element per odd-sized range
  const int64_t ops = 2 * (supplied / smallest + 1) + 2 * n_levels + 64;
  const int64_t floats = supplied + caps + ops + 64 * (int64_t)n_levels;
  return align256(sizeof(SketchOp) * (size_t)ops) +
         (size_t)n_levels * 256 + sizeof(float) * (size_t)n_units * (size_t)floats;
}

int milan_exemplar_sketch_add(const float* hiddens, int batch, int channels, int hw,
                              const int32_t* units, int n_units, int64_t first,
                              int64_t* consumed, float* const* levels,
                              int64_t* firstfree, const int64_t* capacities,
                              int n_levels, const uint8_t* randbits,
                              int64_t n_randbits, int64_t* currentbit,
                              float* extremes, void* workspace,
                              size_t workspace_bytes, milan_stream stream) {
  MILAN_REQUIRE(hiddens && consumed && levels && firstfree && capacities &&
                    randbits && currentbit && extremes && workspace,
                MILAN_ERR_ARG, "sketch_add: null argument");
  const int64_t total = (int64_t)batch * hw;
  MILAN_REQUIRE(n_units > 0 && n_units <= 65535 && n_levels >= 1 && n_levels <= 64 &&
                    first >= 0 && first <= total && hw > 0 && channels > 0,
                MILAN_ERR_SHAPE, "sketch_add: bad sizes");
  for (int l = 0; l < n_levels; ++l)
    MILAN_REQUIRE(capacities[l] >= 2 && capacities[l] <= kSortCap &&
                      firstfree[l] >= 0 && firstfree[l] <= capacities[l],
                  MILAN_ERR_SHAPE,
                  "sketch_add: level capacity outside [2, 8192] (use the per-op path)");
  *consumed = 0;
  const int64_t supplied = total - first;
  if (supplied == 0) return 0;

  // ---- play the state machine forward (runningstats.py:363-407) -------------
  std::vector<int64_t> ff(firstfree, firstfree + n_levels), trial(n_levels);
  std::vector<int64_t> emitted(n_levels + 1, 0);  // elements level l-1 sent to l
  std::vector<int64_t> done(n_levels, 0);         // S_l consumed by compactions
  std::vector<std::vector<SketchOp>> ops(n_levels);
  std::vector<SketchOp> pending;
  std::vector<int> pending_level;
  int64_t bit = *currentbit, index = 0;
  bool blocked = false;
  while (index < supplied) {
    if (capacities[0] - ff[0] == 0) {
      // one _shift(): tentative, committed only if it needs neither _expand()
      // nor more random bits than the caller's buffer still holds
      trial = ff;
      pending.clear(); pending_level.clear();
      int64_t b = bit;
      int l = 0;
      bool ok = true;
      while (capacities[l] - trial[l] < (l ? (capacities[l - 1] + 1) / 2 : 1)) {
        if (l + 1 >= n_levels || b + 1 >= n_randbits) { ok = false; break; }
        const int offset = randbits[++b] ? 1 : 0;
        const int64_t n = trial[l];
        const int64_t moved = (n - offset + 1) / 2;
        pending.push_back(SketchOp{0, (int)n, offset, 0});
        pending_level.push_back(l);
        trial[l] = 0;
        trial[l + 1] += moved;
        ++l;
      }
      if (!ok) { blocked = true; break; }
      for (size_t i = 0; i < pending.size(); ++i) {
        const int pl = pending_level[i];
        SketchOp op = pending[i];
        op.start = (int)done[pl];
        op.outpos = (int)emitted[pl + 1];
        const int64_t moved = (op.n - op.offset + 1) / 2;
        done[pl] += op.n;
        emitted[pl + 1] += moved;
        if (op.n > 0) ops[pl].push_back(op);
      }
      ff = trial;
      bit = b;
    }
    const int64_t room = capacities[0] - ff[0];
    const int64_t take = room < supplied - index ? room : supplied - index;
    ff[0] += take;
    index += take;
  }
  (void)blocked;  // the caller sees consumed < supplied and runs one _shift()

  // ---- workspace ---------------------------------------------------------------
  size_t n_ops = 0;
  for (int l = 0; l < n_levels; ++l) n_ops += ops[l].size();
  char* w = (char*)workspace;
  SketchOp* d_ops = (SketchOp*)w;
  size_t need = align256(sizeof(SketchOp) * (n_ops ? n_ops : 1));
  std::vector<float*> wl(n_levels + 1, nullptr);
  for (int l = 1; l < n_levels; ++l) {
    wl[l] = (float*)(w + need);
    need += align256(sizeof(float) * (size_t)n_units * (size_t)emitted[l]);
  }
  MILAN_REQUIRE(need <= workspace_bytes, MILAN_ERR_WORKSPACE,
                "sketch_add: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (n_ops) {
    // the plan travels from a pinned staging buffer (a truly asynchronous copy); an
    // event guards its reuse by the next call.  One buffer + event PER DEVICE (an event
    // belongs to the device it was created on), under a lock
    struct PlanStage { SketchOp* buf = nullptr; size_t cap = 0; hipEvent_t free = nullptr; bool busy = false; };
    static std::map<int, PlanStage> stages;
    static std::mutex stages_lock;
    std::lock_guard<std::mutex> hold(stages_lock);
    int device = 0;
    MILAN_CHECK_HIP(hipGetDevice(&device));
    PlanStage& st = stages[device];
    SketchOp*& pinned = st.buf;
    size_t& pinned_cap = st.cap;
    hipEvent_t& pinned_free = st.free;
    bool& pinned_busy = st.busy;
    if (!pinned_free) MILAN_CHECK_HIP(hipEventCreateWithFlags(&pinned_free, hipEventDisableTiming));
    if (pinned_busy) { MILAN_CHECK_HIP(hipEventSynchronize(pinned_free)); pinned_busy = false; }
    if (pinned_cap < n_ops) {
      if (pinned) MILAN_CHECK_HIP(hipHostFree(pinned));
      pinned = nullptr; pinned_cap = 0;
      const size_t cap = n_ops + n_ops / 2 + 1024;
      MILAN_CHECK_HIP(hipHostMalloc((void**)&pinned, sizeof(SketchOp) * cap, hipHostMallocDefault));
      pinned_cap = cap;
    }
    size_t o = 0;
    for (int l = 0; l < n_levels; ++l)
      for (const SketchOp& op : ops[l]) pinned[o++] = op;
    MILAN_CHECK_HIP(hipMemcpyAsync(d_ops, pinned, sizeof(SketchOp) * n_ops,
                                   hipMemcpyHostToDevice, s));
    MILAN_CHECK_HIP(hipEventRecord(pinned_free, s));
    pinned_busy = true;
  }
  const SketchSource src{hiddens, units, channels, hw, (long)first};

  // ---- level by level -----------------------------------------------------------
  size_t at = 0;
  for (int l = 0; l < n_levels; ++l) {
    const size_t count = ops[l].size();
    if (!count) continue;
    int widest = 2;
    for (const SketchOp& op : ops[l]) widest = op.n > widest ? op.n : widest;
    int P = 2;
    while (P < widest) P <<= 1;
    int threads = P / 2;
    threads = threads < 64 ? 64 : (threads > 1024 ? 1024 : threads);
    const size_t lds = sizeof(float) * (size_t)P;
    for (size_t begin = 0; begin < count; begin += 65535u * 64u) {
      const size_t part = count - begin < 65535u * 64u ? count - begin : 65535u * 64u;
      if (l == 0)
        hipLaunchKernelGGL(sketch_level_kernel<true>, dim3((unsigned)part, n_units),
                           dim3(threads), lds, s, levels[0], (long)capacities[0],
                           (int)firstfree[0], (const float*)nullptr, 0L, src,
                           d_ops + at + begin, wl[1], (long)emitted[1]);
      else
        hipLaunchKernelGGL(sketch_level_kernel<false>, dim3((unsigned)part, n_units),
                           dim3(threads), lds, s, levels[l], (long)capacities[l],
                           (int)firstfree[l], (const float*)wl[l], (long)emitted[l],
                           src, d_ops + at + begin, wl[l + 1], (long)emitted[l + 1]);
    }
    at += count;
  }
  MILAN_CHECK_HIP(hipGetLastError());
  if (done[0] > 0)  // runningstats.py:398-399 / :415-419
    hipLaunchKernelGGL(sketch_stream_extremes_kernel, dim3(n_units), dim3(256), 0, s,
                       levels[0], (long)capacities[0], (int)firstfree[0], src,
                       (long)done[0], extremes);

  // ---- what is left of every stream is the level's new content --------------------
  for (int l = 0; l < n_levels; ++l) {
    const int64_t old = firstfree[l];
    const int64_t length = old + (l ? emitted[l] : index);
    const int64_t keep = length - done[l];
    MILAN_REQUIRE(keep == ff[l], MILAN_ERR_SHAPE, "sketch_add: plan inconsistent");
    // untouched prefix stays where it is when nothing was compacted
    const int64_t from = done[l] > 0 ? done[l] : old;  // stream position
    const int64_t column = done[l] > 0 ? 0 : old;
    const int64_t n = length - from;
    if (n <= 0) continue;
    if (l == 0) {
      hipLaunchKernelGGL(sketch_append_kernel, dim3(grid1(n, 64), n_units), dim3(256),
                         0, s, hiddens, channels, hw, units,
                         (long)(first + from - old), (long)n, levels[0],
                         (long)capacities[0], (long)column);
    } else {
      MILAN_CHECK_HIP(hipMemcpy2DAsync(
          levels[l] + column, sizeof(float) * capacities[l], wl[l] + (from - old),
          sizeof(float) * emitted[l], sizeof(float) * n, n_units,
          hipMemcpyDeviceToDevice, s));
    }
  }
  MILAN_CHECK_HIP(hipGetLastError());
  for (int l = 0; l < n_levels; ++l) firstfree[l] = ff[l];
  *currentbit = bit;
  *consumed = index;
  return 0;
}

// min / max over the rows of a (rows, units) matrix, 64 units per workgroup: lanes walk
// the columns (coalesced), the four waves walk the rows, LDS combines them
__global__ __launch_bounds__(256) void rows_extremes_kernel(
    const float* __restrict__ rows, long n_rows, int n_units, float* __restrict__ extremes) {
  __shared__ float lo[4][64], hi[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int unit = blockIdx.x * 64 + lane;
  float mn = INFINITY, mx = -INFINITY;
  if (unit < n_units)
    for (long r = wave; r < n_rows; r += 4) {
      const float v = rows[r * n_units + unit];
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
    }
  lo[wave][lane] = mn;
  hi[wave][lane] = mx;
  __syncthreads();
  if (wave == 0 && unit < n_units) {
    for (int w = 1; w < 4; ++w) {
      mn = fminf(mn, lo[w][lane]);
      mx = fmaxf(mx, hi[w][lane]);
    }
    extremes[2 * unit] = fminf(extremes[2 * unit], mn);
    extremes[2 * unit + 1] = fmaxf(extremes[2 * unit + 1], mx);
  }
}

int milan_exemplar_rows_extremes(const float* rows, int64_t n_rows, int n_units,
                                 float* extremes, milan_stream stream) {
  MILAN_REQUIRE(rows && extremes, MILAN_ERR_ARG, "rows_extremes: null argument");
  MILAN_REQUIRE(n_rows >= 0 && n_units > 0, MILAN_ERR_SHAPE, "rows_extremes: bad sizes");
  if (n_rows == 0) return 0;
  hipLaunchKernelGGL(rows_extremes_kernel, dim3((n_units + 63) / 64), dim3(256), 0,
                     (hipStream_t)stream, rows, (long)n_rows, n_units, extremes);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

// Host-only planner of one "make room in level 0" step of the sketch (see the header).
// Restates RunningQuantile._shift / _expand / _next_capacity of the vendored
// netdissect (src/deps/netdissect/runningstats.py:387-407, 485-529) as a function from
// (capacities, fill counts, random bits) to a list of operations.
int milan_exemplar_sketch_plan_shift(int64_t resolution, int64_t buffersize,
                                     int full_rate, int n_levels,
                                     const int64_t* capacities,
                                     const int64_t* firstfree,
                                     int (*draw_bit)(void*), void* user,
                                     milan_sketch_op* ops, int max_ops, int* n_ops,
                                     int64_t* capacities_out, int64_t* firstfree_out,
                                     int* n_levels_out) {
  MILAN_REQUIRE(capacities && firstfree && draw_bit && ops && n_ops &&
                    capacities_out && firstfree_out && n_levels_out,
                MILAN_ERR_ARG, "sketch_plan_shift: null argument");
  MILAN_REQUIRE(n_levels >= 1 && n_levels <= 64 && max_ops >= 2 * n_levels + 4,
                MILAN_ERR_SHAPE, "sketch_plan_shift: bad sizes");
  std::vector<int64_t> cap(capacities, capacities + n_levels);
  std::vector<int64_t> fill(firstfree, firstfree + n_levels);
  int count = 0;
  auto emit = [&](int kind, int src, int dst, int64_t n, int offset, int64_t position,
                  int extremes, int64_t capacity) {
    ops[count++] = milan_sketch_op{kind, src, dst, offset, extremes, 0, n, position,
                                   capacity};
  };
  // a level must keep room for half of the level below it (level 0: one sample)
  auto wanted = [&](size_t level) -> int64_t {
    return level ? (cap[level - 1] + 1) / 2 : 1;
  };
  auto halved = [](int64_t n, int offset) { return (n - offset + 1) / 2; };
  size_t level = 0;
  bool grow = false;
  while (cap[level] - fill[level] < wanted(level)) {
    if (level + 1 >= cap.size()) { grow = true; break; }
    const int offset = draw_bit(user) ? 1 : 0;
    emit(MILAN_SKETCH_COMPACT, (int)level, (int)level + 1, fill[level], offset,
         fill[level + 1], level == 0 && full_rate, 0);
    fill[level + 1] += halved(fill[level], offset);
    fill[level] = 0;
    ++level;
  }
  if (grow) {
    // capacity of one more level: the resolution shrunk by 0.67 per existing level,
    // in steps of 8, not below the buffer size; under 2 there is no further level
    int64_t fresh = (int64_t)std::ceil((double)resolution *
                                       std::pow(0.67, (double)cap.size()));
    fresh = fresh < 2 ? 0 : std::max<int64_t>(buffersize, (fresh + 7) / 8 * 8);
    if (fresh > 0) {
      emit(MILAN_SKETCH_INSERT, 0, 0, 0, 0, 0, 0, fresh);
      cap.insert(cap.begin(), fresh);
      fill.insert(fill.begin(), 0);
    } else {
      MILAN_REQUIRE(fill[0] == 0, MILAN_ERR_STATE, "sketch_plan_shift: level 0 not empty");
      emit(MILAN_SKETCH_HALVE, 0, 0, 0, 0, 0, 0, 0);
    }
    // every level hands its content one level down where that level has the room,
    // and is compacted in place where it has not
    for (size_t upper = 1; upper < cap.size(); ++upper) {
      const int64_t amount = fill[upper];
      if (amount == 0) continue;
      const size_t lower = upper - 1;
      if (cap[lower] - (fill[lower] + amount) >= wanted(lower)) {
        emit(MILAN_SKETCH_MOVE, (int)upper, (int)lower, amount, 0, fill[lower], 0, 0);
        fill[lower] += amount;
        fill[upper] = 0;
      } else {
        const int offset = draw_bit(user) ? 1 : 0;
        emit(MILAN_SKETCH_COMPACT, (int)upper, (int)upper, amount, offset, 0,
             upper == 1, 0);
        fill[upper] = halved(amount, offset);
      }
    }
  }
  MILAN_REQUIRE(count <= max_ops, MILAN_ERR_SHAPE, "sketch_plan_shift: plan too long");
  *n_ops = count;
  *n_levels_out = (int)cap.size();
  for (size_t l = 0; l < cap.size(); ++l) {
    capacities_out[l] = cap[l];
    firstfree_out[l] = fill[l];
  }
  return 0;
}

int milan_exemplar_sketch_quantile(const float* const* levels,
                                   const int64_t* firstfree,
                                   const int64_t* capacities, int n_levels,
                                   int n_units, float* extremes, float q,
                                   float* out, void* workspace,
                                   size_t workspace_bytes, milan_stream stream) {
  MILAN_REQUIRE(levels && firstfree && capacities && extremes && out && workspace,
                MILAN_ERR_ARG, "sketch_quantile: null argument");
  MILAN_REQUIRE(n_levels >= 1 && n_levels <= 64 && n_units > 0, MILAN_ERR_SHAPE,
                "sketch_quantile: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  int64_t total = 0;
  for (int l = 0; l < n_levels; ++l) total += firstfree[l];
  MILAN_REQUIRE(total > 0 && (int64_t)n_units * total < (int64_t)1 << 31,
                MILAN_ERR_SHAPE, "sketch_quantile: empty or oversized sketch");
  MILAN_REQUIRE(workspace_bytes >= milan_exemplar_sort_workspace(n_units, total, 1),
                MILAN_ERR_WORKSPACE, "sketch_quantile: workspace too small");
  // the level-0 remainder has not been through a sort yet (:532-533)
  if (firstfree[0] > 0)
    hipLaunchKernelGGL(sketch_scan_extremes_kernel, dim3(n_units), dim3(256), 0, s,
                       levels[0], (long)capacities[0], (long)firstfree[0],
                       extremes);
  char* w = (char*)workspace;
  int* begin = (int*)w; w += align256(sizeof(int) * (size_t)n_units);
  int* end = (int*)w; w += align256(sizeof(int) * (size_t)n_units);
  const size_t items = (size_t)n_units * total;
  float* keys = (float*)w; w += align256(sizeof(float) * items);
  float* keys_sorted = (float*)w; w += align256(sizeof(float) * items);
  float* wts = (float*)w; w += align256(sizeof(float) * items);
  float* wts_sorted = (float*)w; w += align256(sizeof(float) * items);
  size_t temp = workspace_bytes - (size_t)(w - (char*)workspace);
  long at = 0;
  for (int l = 0; l < n_levels; ++l) {
    if (firstfree[l] == 0) continue;
    hipLaunchKernelGGL(sketch_gather_kernel, dim3(grid1(firstfree[l], 64), n_units),
                       dim3(256), 0, s, levels[l], (long)capacities[l],
                       (long)firstfree[l], ldexpf(1.f, l), keys, wts, (long)total,
                       at);
    at += firstfree[l];
  }
  hipLaunchKernelGGL(segment_offsets_kernel, dim3((n_units + 255) / 256), dim3(256),
                     0, s, n_units, (long)total, (long)total, begin, end);
  // stable: equal values keep their concatenation order, like torch.sort on CPU
  MILAN_CHECK_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(
      w, temp, keys, keys_sorted, wts, wts_sorted, (int)items, n_units, begin, end,
      0, 32, s));
  hipLaunchKernelGGL(sketch_quantile_kernel, dim3((n_units + 63) / 64), dim3(64), 0,
                     s, keys_sorted, wts_sorted, (long)total, n_units, extremes, q,
                     out);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

int milan_exemplar_render(const float* hiddens, int batch, int channels, int h,
                          int w, const float* images, int img_h, int img_w,
                          const int32_t* cells, int n_cells, const float* levels,
                          const float* mul3, const float* add3, int out_size,
                          int k, uint8_t* out_images, uint8_t* out_masks,
                          uint8_t* out_masked, milan_stream stream) {
  MILAN_REQUIRE(hiddens && images && cells && levels && mul3 && add3 &&
                    out_images && out_masks && out_masked,
                MILAN_ERR_ARG, "exemplar_render: null argument");
  MILAN_REQUIRE(batch > 0 && channels > 0 && h > 0 && w > 0 && img_h > 0 &&
                    img_w > 0 && out_size > 0 && k > 0 && n_cells >= 0,
                MILAN_ERR_SHAPE, "exemplar_render: bad sizes");
  if (n_cells == 0) return 0;
  RenderGeom g;
  g.h = h; g.w = w; g.img_h = img_h; g.img_w = img_w; g.out = out_size;
  // upsample.upsample_grid with scale_offset=None (upsample.py:132-150); the
  // Python arithmetic is double, the tensor arithmetic float32
  const double sy = (double)out_size / h, sx = (double)out_size / w;
  g.off_y = (float)(0.5 * sy - 0.5);
  g.off_x = (float)(0.5 * sx - 0.5);
  g.mul_y = (float)(2.0 / (sy * (h - 1 > 1 ? h - 1 : 1)));
  g.mul_x = (float)(2.0 / (sx * (w - 1 > 1 ? w - 1 : 1)));
  g.half_h = (float)(h - 1) / 2;
  g.half_w = (float)(w - 1) / 2;
  // F.interpolate(..., size=) nearest: scale = in / out in float
  g.scale_y = (float)img_h / (float)out_size;
  g.scale_x = (float)img_w / (float)out_size;
  for (int c = 0; c < 3; ++c) { g.mul[c] = mul3[c]; g.add[c] = add3[c]; }
  const int px = out_size * out_size;
  hipLaunchKernelGGL(exemplar_render_kernel, dim3((px + 255) / 256, n_cells),
                     dim3(256), 0, (hipStream_t)stream, hiddens, channels, images,
                     cells, levels, k, g, out_images, out_masks, out_masked);
  MILAN_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
