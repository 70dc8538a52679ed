/* libmilan_hip -- C ABI of the MI355X-native MILAN inference path.
 *
 * The reference (evandez/neuron-descriptions) is pure Python and has no FFI
 * of its own; the seam this library sits behind is the Python class contract
 * of `src.milan.Decoder`.  Each entry point below is the
 * native counterpart of one reference method; the Python mirror in
 * neuron-descriptions_amd/milan_amd/ binds them with ctypes (INTEGRATION.md
 * shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - Every data pointer is a DEVICE pointer owned by the caller (the Python
 *     side passes torch tensors' data_ptr()).  The library owns only its packed
 *     weight arena, freed by milan_destroy().
 *   - All work is enqueued on `stream` (a hipStream_t); no implicit sync except
 *     where documented (milan_finalize_weights).
 *   - Return 0 on success; negative = MILAN_ERR_*; positive = hipError_t.
 *     milan_last_error() returns a thread-local message for the last failure.
 *   - float = IEEE binary32 everywhere (the reference computes in fp32); token
 *     ids are int64 like torch.long.
 *   - One ctx per device; a ctx is not thread-safe.
 */
#ifndef MILAN_HIP_H
#define MILAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the prototypes of this header
 * are its whole dynamic symbol table (tests/test_host.py checks `nm -D`). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define MILAN_ABI_VERSION 8

enum {
  MILAN_OK = 0,
  MILAN_ERR_ARG = -1,       /* bad argument (-> ValueError in Python)        */
  MILAN_ERR_SHAPE = -2,     /* shape/dim mismatch (-> ValueError)            */
  MILAN_ERR_STATE = -3,     /* call order (weights missing / not finalized)  */
  MILAN_ERR_WORKSPACE = -4, /* workspace too small                           */
  MILAN_ERR_NO_LM = -5      /* MI / rerank requested without an LM
                               (src/milan/decoders.py:397-398)               */
};

enum { MILAN_DTYPE_U8 = 0, MILAN_DTYPE_F32 = 1 };

/* src/milan/decoders.py:217-221; MILAN_FORCED = `strategy=<tensor>` (teacher
 * forcing, decoders.py:444-445). */
enum { MILAN_GREEDY = 0, MILAN_FORCED = 1, MILAN_BEAM = 2, MILAN_RERANK = 3 };

typedef struct milan_ctx milan_ctx;
typedef void* milan_stream; /* hipStream_t */

/* Model geometry.  Mirrors the constructor arguments of
 * src/milan/decoders.py:233-244, src/milan/lms.py:20-25 and the encoder
 * config of src/milan/encoders.py:326-351: all four pyramid configs --
 * 'resnet50'/'resnet101' (bottleneck blocks), 'resnet18' (basic blocks) and
 * 'alexnet'. */
enum {
  MILAN_TRUNK_BOTTLENECK = 0, /* taps conv1 + layer1..4, F = 61 * width      */
  MILAN_TRUNK_BASIC = 1,      /* same taps, expansion 1,   F = 16 * width      */
  MILAN_TRUNK_ALEXNET = 2,    /* taps features.0/3/6/8/10, F = 18 * width      */
  MILAN_TRUNK_NONE = 3        /* decoder-only context (a foreign `Encoder`):
                                 any feature_size % 4 == 0, trunk_* ignored   */
};
typedef struct milan_dims {
  int32_t trunk_width;      /* 64 for the torchvision models                */
  int32_t trunk_blocks[4];  /* {3,4,23,3} = resnet101 (unused for alexnet)   */
  int32_t feature_size;     /* 61 / 16 / 18 * trunk_width by trunk_kind      */
  int32_t hidden_size;      /* 512                                           */
  int32_t embedding_size;   /* 128                                           */
  int32_t attention_size;   /* min(hidden, feature) = 512 (decoders.py:50)   */
  int32_t vocab_size;       /* len(indexer) = |vocab| + 4                    */
  int32_t start_index;      /* |vocab|     (src/utils/lang.py:242-245)       */
  int32_t stop_index;       /* |vocab| + 1                                   */
  int32_t pad_index;        /* |vocab| + 2                                   */
  int32_t has_lm;           /* 0/1                                           */
  int32_t lm_hidden_size;   /* 512                                           */
  int32_t lm_embedding_size;/* 128                                           */
  int32_t lm_layers;        /* 2                                             */
  int32_t trunk_kind;       /* MILAN_TRUNK_* (0 = bottleneck ResNet)         */
} milan_dims;

int milan_abi_version(void);
const char* milan_last_error(void);

/* Lifetime.  Replaces Decoder.__init__ / Decoder.to(device). */
int milan_create(milan_ctx** out, int device, const milan_dims* dims);
void milan_destroy(milan_ctx* ctx);

/* Weight upload.  `name` is the key in the reference's Decoder.state_dict()
 * (src/utils/serialize.py:188-204), e.g. "lstm.weight_ih",
 * "attend.key_to_hidden.weight", "lm.lstm.weight_hh_l1",
 * "encoder.encoder.model.layer3.7.bn2.running_var".  `data` is a device float*
 * that must stay valid until milan_finalize_weights() returns.  Unknown names
 * ("...num_batches_tracked", "...fc.weight": computed-and-discarded by the
 * reference) are accepted and ignored.  Replaces load_state_dict
 * (src/utils/serialize.py:222-252). */
int milan_set_weight(milan_ctx* ctx, const char* name, const float* data,
                     const int64_t* shape, int ndim);
/* Folds eval-mode BatchNorm into the convolutions, repacks every matrix into
 * the kernels' layout inside a library-owned arena, then synchronises
 * `stream` (after which the caller may free the uploaded tensors). */
int milan_finalize_weights(milan_ctx* ctx, milan_stream stream);

/* Bytes of scratch the calls below need for at most `max_neurons` neurons of
 * `k` exemplars at image_size x image_size, `beam_size` beams, `length` steps. */
size_t milan_workspace_bytes(const milan_ctx* ctx, int max_neurons, int k,
                             int image_size, int beam_size, int length);

/* Decoder.encode / PyramidConvEncoder.forward
 * (src/milan/decoders.py:525-546, src/milan/encoders.py:286-320).
 * images: (n_images,3,H,W) NCHW, uint8 (0..255, converted exactly like
 * src/milannotations/datasets.py:191-197) or float in [0,1];
 * masks: (n_images,1,H,W) uint8 {0,1} or float, or NULL (= all ones).
 * features: (n_images, feature_size) float out. */
int milan_encode(milan_ctx* ctx, const void* images, int image_dtype,
                 const void* masks, int mask_dtype, int n_images, int height,
                 int width, float* features, void* workspace,
                 size_t workspace_bytes, milan_stream stream);

/* SpatialConvEncoder.forward (src/milan/encoders.py:158-230, config
 * 'resnet18'): the image is normalised, THEN multiplied by its mask (NULL =
 * ones), run through the ResNet trunk, and the last stage's output is returned
 * position-major: out (n_images, h4*w4, C4) = layer4.permute(0,2,3,1), i.e.
 * (n, 49, 512) for 224x224 / resnet18.  ResNet trunks only. */
int milan_encode_spatial(milan_ctx* ctx, const void* images, int image_dtype,
                         const void* masks, int mask_dtype, int n_images,
                         int height, int width, float* out, void* workspace,
                         size_t workspace_bytes, milan_stream stream);

/* Decoder.init_state (src/milan/decoders.py:548-574).
 * features (n,k,F) -> h,c (n,hidden). */
int milan_init_state(milan_ctx* ctx, const float* features, int n, int k,
                     float* h, float* c, void* workspace,
                     size_t workspace_bytes, milan_stream stream);

/* Decoder.step (src/milan/decoders.py:576-634), eval mode.
 * features (rows,k,F); tokens (rows) int64; h,c (rows,hidden) in;
 * h_lm,c_lm (lm_layers,rows,lm_hidden) in/out or NULL (no MI);
 * predictions (rows,V), attentions (rows,k), h_out,c_out (rows,hidden) out. */
int milan_step(milan_ctx* ctx, const float* features, int rows, int k,
               const int64_t* tokens, const float* h, const float* c,
               float* h_lm, float* c_lm, float temperature, float* predictions,
               float* attentions, float* h_out, float* c_out, void* workspace,
               size_t workspace_bytes, milan_stream stream);

/* Decoder.forward on features (src/milan/decoders.py:335-523), eval mode.
 *   strategy MILAN_GREEDY: tokens (n,length), scores (n); predictions
 *     (n,length,V) and attentions (n,length,k) optional (may be NULL);
 *     beam_* must be NULL.  mi = use the LM per step (decoders.py:624-630).
 *   strategy MILAN_FORCED: as MILAN_GREEDY but the token fed to step t+1 is
 *     tokens[:, t] -- `tokens` (n,length) is an INPUT holding the forced ids
 *     (0 <= id < V, validated by the caller) and is left unchanged, scores
 *     accumulate predictions[b, t, tokens[b, t]] (what Decoder.score sums,
 *     decoders.py:636-711); predictions is required.
 *   strategy MILAN_BEAM / MILAN_RERANK: allennlp-2.10 BeamSearch semantics
 *     (decoders.py:467-484) then best-of-beam / LM rerank (decoders.py:492-512).
 *     beam_tokens (n,beam,length) int64, beam_scores (n,beam),
 *     tokens (n,length), scores (n).  The search always runs `length` steps;
 *     out_len[g] receives T' (the length allennlp would have returned for
 *     neuron group g, groups of `group_size` consecutive neurons = the
 *     reference's forward batch), positions >= T' hold stop_index.  Rerank LM
 *     scores use only the first T' tokens, as the reference does.
 *     out_len is a DEVICE int32 array of ceil(n/group_size) entries. */
int milan_decode(milan_ctx* ctx, const float* features, int n, int k,
                 int strategy, int length, int beam_size, int mi,
                 float temperature, int group_size, int64_t* tokens,
                 float* scores, float* predictions, float* attentions,
                 int64_t* beam_tokens, float* beam_scores, int32_t* out_len,
                 void* workspace, size_t workspace_bytes, milan_stream stream);

/* LanguageModel.forward(inputs, reduce=True) (src/milan/lms.py:58-101),
 * including its stop-mask off-by-one.  seqs (rows,L) int64, first column is
 * the start token; out (rows).  seq_len: optional DEVICE int32 per row giving
 * the number of valid columns (<= L) or NULL (= L). */
int milan_lm_score(milan_ctx* ctx, const int64_t* seqs, int rows, int L,
                   const int32_t* seq_len, float* out, void* workspace,
                   size_t workspace_bytes, milan_stream stream);

/* LanguageModel.forward(inputs, reduce=False) (src/milan/lms.py:58-88): the
 * log-probabilities of every next token at every position.  seqs (rows,L)
 * int64; out (rows,L,V).  Used for custom `masks=` reductions and analysis;
 * the rerank path uses milan_lm_score, which never materialises this. */
int milan_lm_logprobs(milan_ctx* ctx, const int64_t* seqs, int rows, int L,
                      float* out, void* workspace, size_t workspace_bytes,
                      milan_stream stream);

/* The whole hot path for one batch = Decoder.forward(images, masks, ...)
 * (what Decoder.predict calls per batch, src/milan/decoders.py:857-865):
 * encode then decode.  images (n,k,3,H,W), masks (n,k,1,H,W); other
 * arguments as milan_decode.  features_out (n,k,F) optional. */
int milan_describe(milan_ctx* ctx, const void* images, int image_dtype,
                   const void* masks, int mask_dtype, int n, int k, int height,
                   int width, int strategy, int length, int beam_size, int mi,
                   float temperature, int group_size, float* features_out,
                   int64_t* tokens, float* scores, float* predictions,
                   float* attentions, int64_t* beam_tokens, float* beam_scores,
                   int32_t* out_len, void* workspace, size_t workspace_bytes,
                   milan_stream stream);

/* hipGraph capture of the decode stage.  When enabled, milan_decode (and the
 * decode half of milan_describe) records its launches -- the whole
 * Decoder.forward search of src/milan/decoders.py:418-512 for one batch -- into
 * a hipGraph the second time an identical call (same sizes AND same buffer
 * pointers, same stream) is seen, and replays it with one hipGraphLaunch from
 * then on.  Requires a non-default stream and caller-side buffer reuse;
 * otherwise (or while milan_profile_enable is on) calls run as plain launches.
 * milan_graph_stats reports how many graphs were captured / replayed. */
int milan_set_graph_capture(milan_ctx* ctx, int enable);
int milan_graph_stats(const milan_ctx* ctx, long long* captures,
                      long long* replays);

/* Arithmetic mode of every dense contraction (convs, Linear, LSTM gates).
 *   MILAN_PRECISION_F32       fp32-in / fp32-accumulate MFMA: bitwise an fmaf
 *                             chain, the reference's precision (default).
 *   MILAN_PRECISION_SPLIT_F16 operands carried as (hi,lo) f16 pairs (22
 *                             significant bits), A.B = Ah.Bh + Ah.Bl + Al.Bh on
 *                             the f16 matrix cores with fp32 accumulation:
 *                             fp32-GEMM-class error at 1/3 of the f16 MFMA rate.
 * Switchable at any time between calls (both weight packings are kept).
 *
 * In split mode the ResNet trunks keep their activations multiplied by a power of two
 * (2^5; environment MILAN_ACT_SCALE_LOG2=<0..10>, read by milan_create): the `lo` half of
 * an operand then stays in the f16 normal range down to activations of ~0.004 (1e-3-scale
 * networks stay in the fp32 error class), while `hi` saturates at 65504 / 2^k (2047).  The
 * scale is invisible at this interface: features, spatial features and descriptions are
 * returned unscaled, and 2^0 reproduces the unscaled storage bit for bit. */
enum { MILAN_PRECISION_F32 = 0, MILAN_PRECISION_SPLIT_F16 = 1,
       /* "Fast mode" (SURVEY 7 hard part 1): NARROWER than the reference's fp32
        * -- a reported extra, never the default and never the headline.  layer3 / layer4 of a
        * bottleneck ResNet trunk run on plain f16 operands (11 significant bits), one f16 MFMA
        * per product with fp32 accumulation, activations stored as 2 bytes per element;
        * everything else (stem, layer1 / layer2, decoder, LM) stays MILAN_PRECISION_SPLIT_F16.
        * Error class 2^-11 of the operand scale; bench.py reports its caption-flip rate against
        * MILAN_PRECISION_F32 next to its throughput. */
       MILAN_PRECISION_F16 = 2 };

/* Cross-layer fusions of the trunk (split-f16 mode; results are bitwise those of
 * the unfused schedule, so this is a scheduling knob for A/B timing and tests):
 *   MILAN_FUSE_CHAIN  a bottleneck's 1x1 expand conv (+ residual + ReLU) and the next
 *                     bottleneck's 1x1 reduce conv run as one launch
 *                     (csrc/chain.hip; torchvision Bottleneck.forward as called from
 *                     src/milan/encoders.py:298).
 *   MILAN_FUSE_STEM   conv1 7x7/2 + bn1 + ReLU + maxpool 3x3/2 as one persistent
 *                     launch over LDS-resident input tiles (csrc/stem.hip; the raw
 *                     conv1 tensor, pyramid level 0 of encoders.py:303-320, is only
 *                     materialised where the mask-weighted pooling reads it).
 *   MILAN_FUSE_CONV3  the 64 -> 64 channel 3x3 convolutions of layer1 as a persistent
 *                     kernel with register-resident weights and an LDS-resident input
 *                     tile (csrc/conv3.hip) instead of the implicit GEMM.
 *   MILAN_FUSE_SKIP_EMPTY  exemplars whose mask is all zero pool exact zeros at every
 *                     pyramid level whatever the trunk computes (src/milan/encoders.py:
 *                     310-317): they are left out of the trunk pass (uint8 images).  The
 *                     number of images with work stays ON THE DEVICE (round 6): every
 *                     trunk launch is sized for the whole batch and reads the count
 *                     itself, workgroups beyond it exit -- nothing is read back and
 *                     the stream is never synchronised (rounds 5's form did both).
 *   MILAN_FUSE_BNECK  (round 6; needs MILAN_FUSE_CHAIN) layer1's 3x3 convolution -- and for the
 *                     stage's first block its own 1x1 reduce conv -- run inside the block's
 *                     chain launch, from an LDS copy of the input region (csrc/chain.hip,
 *                     chain_kernel<.., CONV[, C1]>): the bottleneck's intermediate tensors
 *                     (torchvision Bottleneck.forward: out of conv1 / conv2) never exist in
 *                     memory.
 *   MILAN_FUSE_SPARSE_TAIL  (round 6) the last stage's output is read by nothing but the
 *                     level-4 pooling of src/milan/encoders.py:303-320, at the pixels under
 *                     the shrunk mask: the last two bottlenecks run only at those pixels and
 *                     the 3x3 neighbourhoods they depend on (row sets built on the device
 *                     from the pooling's own pixel lists; dense when no masks are given).
 * Default: all seven (environment MILAN_CHAIN=<flags> overrides at context creation).
 * Not a flag but the same idea in the decoder: the rerank pass (decoders.py:495-512) scores
 * one LM row per distinct beam PREFIX instead of one per beam (csrc/decoder.hip,
 * lm_score_dedup; MILAN_LM_DEDUP=0 restores every row; same bits). */
enum { MILAN_FUSE_CHAIN = 1,       /* planes <= 128 (layer1, layer2): HBM-bound, wins */
       MILAN_FUSE_CHAIN_WIDE = 2,  /* planes 256 (layer3): the role ping-pong of csrc/chain3.hip
                                      (round 5) */
       MILAN_FUSE_STEM = 4,
       MILAN_FUSE_CONV3 = 8,       /* layer1's 3x3 convs: weights in registers (csrc/conv3.hip) */
       MILAN_FUSE_SKIP_EMPTY = 16,
       MILAN_FUSE_BNECK = 32,      /* layer1: the 3x3 conv in front of the chain launch (round 6) */
       MILAN_FUSE_SPARSE_TAIL = 64 };  /* the last two bottlenecks only at the pixels the level-4
                                          pooling (and their 3x3 neighbourhoods) read (round 6) */
int milan_set_fusion(milan_ctx* ctx, int flags);
int milan_set_precision(milan_ctx* ctx, int precision);
int milan_get_precision(const milan_ctx* ctx);

/* Split-f16 mode fails LOUDLY (round 5).  The reference computes in plain fp32
 * (src/milan/encoders.py:295-320) and never saturates; the (hi, lo) f16 storage clamps
 * |x * 2^act_scale_log2| at 65504.  Every kernel that writes split format keeps the running
 * maximum of what it clamped and ORs MILAN_STATUS_SATURATED into the context's device status
 * word when the clamp was hit; the input conversion ORs MILAN_STATUS_NONFINITE_INPUT when a
 * float pixel was NaN / Inf (the features of such an image are NaN at every pyramid level,
 * exactly as the reference's pooling of NaN x mask yields them).
 *   milan_status            copies the word to *flags (host), optionally clearing it;
 *                           synchronises `stream`.  The Python mirror reads it after every
 *                           encode / describe and raises FloatingPointError on
 *                           MILAN_STATUS_SATURATED (or reruns the call in MILAN_PRECISION_F32
 *                           when Decoder.precision == 'auto').
 *   milan_set_act_scale_log2 activation scale 2^k of the split trunk, k in [0, 10]
 *                           (replaces the MILAN_ACT_SCALE_LOG2 default of 5): `hi` saturates
 *                           at 65504 / 2^k, `lo` leaves the f16 normal range below
 *                           2^-3 / 2^k.  Rewrites the pre-scaled bias vectors in place.
 *   milan_encoder_absmax    calibration: runs the trunk in MILAN_PRECISION_F32 over a sample
 *                           and returns the largest |activation| of every tensor the split
 *                           trunk would store (*absmax, host); Decoder.calibrate picks the
 *                           scale from it.  Synchronises `stream`. */
enum { MILAN_STATUS_SATURATED = 1, MILAN_STATUS_NONFINITE_INPUT = 2 };
int milan_status(milan_ctx* ctx, uint32_t* flags, int clear, milan_stream stream);
int milan_set_act_scale_log2(milan_ctx* ctx, int log2_scale, milan_stream stream);
int milan_get_act_scale_log2(const milan_ctx* ctx);
int milan_encoder_absmax(milan_ctx* ctx, const void* images, int image_dtype,
                         int n_images, int height, int width, float* absmax,
                         void* workspace, size_t workspace_bytes,
                         milan_stream stream);

/* Measurement hook (bench.py's roofline leg): while enabled, every launch of
 * the implicit-GEMM MFMA kernel is bracketed by HIP events on its launch
 * stream.  milan_profile_read synchronises the device and returns the summed
 * kernel time (ms), the algorithmic FLOPs (2*M*N*K, un-padded K) and the launch
 * count since milan_profile_enable(1).  Process-wide; not for production. */
int milan_profile_enable(int enable);
int milan_profile_read(double* gemm_ms, double* gemm_flops,
                       long long* gemm_launches);

/* Per-stage breakdown of the same measurement (north_star: "fraction of
 * HBM/MFMA roofline reported per stage").  While profiling is enabled every
 * stage region of the hot path is also bracketed by two HIP events and every
 * GEMM record carries the stage it ran in.  `table` receives
 * MILAN_STAGE_COUNT rows of 6 doubles: region ms (sum over calls), region
 * count, GEMM ms inside the stage, GEMM algorithmic FLOPs, GEMM launches, GEMM
 * algorithmic HBM bytes (every operand of a launch crossing HBM once: visited
 * input pixels, weights, residual, output -- what bounds the small-K layers). */
enum milan_stage {
  MILAN_STAGE_OTHER = 0,
  MILAN_STAGE_ENC_INPUT = 1,     /* mask pyramid lists + u8 -> normalised input
                                    (datasets.py:191-197, encoders.py:295)   */
  MILAN_STAGE_ENC_STEM = 2,      /* conv1 7x7/2                               */
  MILAN_STAGE_ENC_STEM_TAIL = 3, /* bn1 + relu + maxpool 3x3/2                */
  MILAN_STAGE_ENC_LAYER1 = 4,    /* torchvision layer1..layer4 (convs only)   */
  MILAN_STAGE_ENC_LAYER2 = 5,
  MILAN_STAGE_ENC_LAYER3 = 6,
  MILAN_STAGE_ENC_LAYER4 = 7,
  MILAN_STAGE_ENC_POOL = 8,      /* mask-weighted pooling of the five taps
                                    (encoders.py:303-320)                    */
  MILAN_STAGE_DEC_INIT = 9,      /* hoisted key projection + init_state       */
  MILAN_STAGE_DEC_SEARCH = 10,   /* the T-step greedy / beam loop             */
  MILAN_STAGE_DEC_LM = 11,       /* LM scoring of the beams + rerank select   */
  MILAN_STAGE_COUNT = 12
};
int milan_profile_read_stages(double* table /* [MILAN_STAGE_COUNT][6] */);

/* The same records by kernel family: `table` receives MILAN_KERNEL_COUNT rows of 4
 * doubles -- summed launch time (ms), algorithmic FLOPs, launches, algorithmic HBM bytes --
 * so that bench.py's `roofline` can price the DOMINANT kernel on its own launches. */
enum milan_kernel_family {
  MILAN_KERNEL_OTHER = 0,
  MILAN_KERNEL_PP32_256 = 1,    /* igemm_split16_pp32_kernel (256 x 256 ping-pong tile)  */
  MILAN_KERNEL_PP32_128 = 2,    /* igemm_split16_pp32n_kernel<128>                       */
  MILAN_KERNEL_SPLIT_OTHER = 3, /* the other split-f16 implicit-GEMM tiles               */
  MILAN_KERNEL_F32 = 4,         /* igemm_kernel, v_mfma_f32_32x32x2_f32                  */
  MILAN_KERNEL_CHAIN = 5,       /* chain_kernel (layer1 / layer2 expand -> reduce)       */
  MILAN_KERNEL_CHAIN_WIDE = 6,  /* layer3 expand -> reduce chain                         */
  MILAN_KERNEL_STEM = 7,        /* stem_fused_kernel                                     */
  MILAN_KERNEL_CONV3 = 8,       /* conv3_p64_kernel                                      */
  MILAN_KERNEL_F16 = 9,         /* igemm_f16_pp32_kernel (fast mode)                     */
  MILAN_KERNEL_PP32T_256 = 10,  /* igemm_split16_pp32t_kernel<256>: the same tile function as
                                   PP32_256 on k x k convs in (slice, tap, channel) order  */
  MILAN_KERNEL_BNECK = 11,      /* chain_kernel<.., CONV>: layer1's 3x3 + expand + reduce  */
  MILAN_KERNEL_COUNT = 12
};
int milan_profile_read_kernels(double* table /* [MILAN_KERNEL_COUNT][4] */);

/* ---- exemplar computation ---------------------------
 * The stage that WRITES the images.npy / masks.npy this path reads
 * (src/exemplars/compute.py:27-246).  The dissected model stays the caller's
 * (the reference takes it as two black-box functions, compute.py:27-29);
 * these entry points are the tensor operations of its netdissect
 * dependencies on the activation tensor `hiddens` (batch, channels, h, w)
 * fp32 NCHW, hw = h * w.  `units` (device int32 [n_units], or NULL = all
 * channels) is the `units=` subset of compute.py:186-199.  The control flow of
 * the quantile sketch (which level is compacted when, with which random bit)
 * is host logic, as it is Python in the reference (milan_amd/exemplars.py). */

/* RunningTopK.add + result (src/deps/netdissect/runningstats.py:58-116) on the
 * spatial max of compute.py:331: merges this batch into top_values/top_index
 * [n_units][k], of which `filled` columns are valid; afterwards
 * min(k, filled + batch) are, sorted by value descending (equal values: lower
 * dataset index first).  Image b of the batch has dataset index
 * first_index + b.  pooled_scratch: n_units * batch floats.
 * filled + batch <= 2048. */
int milan_exemplar_topk_update(const float* hiddens, int batch, int channels,
                               int hw, const int32_t* units, int n_units,
                               int64_t first_index, int k, int filled,
                               float* pooled_scratch, float* top_values,
                               int64_t* top_index, milan_stream stream);

/* RunningQuantile._add_every's copy (runningstats.py:363-385): activation rows
 * [first, first + count) of hiddens.permute(0,2,3,1).reshape(-1, C) go,
 * transposed, into level0[n_units][capacity] at columns column.. */
int milan_exemplar_sketch_append(const float* hiddens, int batch, int channels,
                                 int hw, const int32_t* units, int n_units,
                                 int64_t first, int64_t count, float* level0,
                                 int64_t capacity, int64_t column,
                                 milan_stream stream);

/* One compaction of RunningQuantile._shift / _expand (runningstats.py:387-407,
 * 485-521): every row's first n columns of src are sorted ascending, every
 * second element starting at `offset` (the random bit) is written to dst at
 * columns position.., and -- when `extremes` [n_units][2] is given -- the
 * row's minimum / maximum are folded into it (:415-419).  dst may equal src
 * (the in-place "scrunch") with position 0.  Workspace:
 * milan_exemplar_sort_workspace(n_units, n, 0). */
size_t milan_exemplar_sort_workspace(int n_units, int64_t n, int pairs);
int milan_exemplar_sketch_compact(const float* src, int64_t src_capacity,
                                  int64_t n, int n_units, int offset, float* dst,
                                  int64_t dst_capacity, int64_t position,
                                  float* extremes, void* workspace,
                                  size_t workspace_bytes, milan_stream stream);

/* RunningQuantile._add_every for a whole batch (runningstats.py:363-407): the
 * sketch's control flow depends on buffer sizes only, so it is played forward on
 * the host and executed level by level -- one launch per level, every compaction
 * of a level in parallel -- instead of one append + one compaction launch per
 * 128..8192 samples.  Consumes activation rows [first, batch*hw) of `hiddens`
 * (row order as milan_exemplar_sketch_append).  levels / firstfree / capacities:
 * HOST arrays of n_levels entries (device row pointers; fill counts, updated on
 * return; row capacities, each in [2, 8192]).  randbits: HOST array of the
 * caller's random bits (torch's, so that seeded runs match the reference),
 * *currentbit the index of the last one used, updated on return.  Stops early --
 * *consumed < batch*hw - first, level 0 full -- before a _shift() that needs
 * _expand() or more random bits than remain: the caller performs that one
 * _shift() (milan_exemplar_sketch_compact) and calls again.  Same results as
 * the per-operation path, bit for bit. */
size_t milan_exemplar_sketch_add_workspace(int n_units, int64_t supplied,
                                           const int64_t* capacities,
                                           int n_levels);
int milan_exemplar_sketch_add(const float* hiddens, int batch, int channels, int hw,
                              const int32_t* units, int n_units, int64_t first,
                              int64_t* consumed, float* const* levels,
                              int64_t* firstfree, const int64_t* capacities,
                              int n_levels, const uint8_t* randbits,
                              int64_t n_randbits, int64_t* currentbit,
                              float* extremes, void* workspace,
                              size_t workspace_bytes, milan_stream stream);

/* RunningQuantile._scan_extremes (runningstats.py:409-419), used by the sketch's
 * subsampling regime where not every sample enters a buffer: column-wise minimum /
 * maximum of `rows` [n_rows][n_units] (row-major, device) folded into `extremes`
 * [n_units][2] = (min, max). */
int milan_exemplar_rows_extremes(const float* rows, int64_t n_rows, int n_units,
                                 float* extremes, milan_stream stream);

/* What RunningQuantile does when level 0 is full and the bulk call above stopped: one
 * _shift() (runningstats.py:387-407) INCLUDING the _expand() it may end in
 * (:485-529), as a plan -- host arithmetic only, no device work.  The caller executes
 * the operations in order (on its own tensors) and adopts the returned fill counts:
 *   MILAN_SKETCH_COMPACT  sort level `src`'s first n columns, keep every second one
 *                         from `offset`, write them to level `dst` at `position`
 *                         (milan_exemplar_sketch_compact; `extremes` != 0: fold min /
 *                         max into the extremes);
 *   MILAN_SKETCH_INSERT   a new, empty level 0 of `capacity` columns (every later
 *                         level index in the plan counts it);
 *   MILAN_SKETCH_MOVE     level `src`'s first n columns -> level `dst` at `position`;
 *   MILAN_SKETCH_HALVE    the sample rate halves (no room for another level).
 * Level indices are those at the moment the operation runs.  `draw_bit(user)` hands
 * out the caller's random bits in the order upstream consumes them (torch's global
 * generator, so that a seeded run reproduces the reference).  capacities_out /
 * firstfree_out need room for n_levels + 1 entries. */
enum { MILAN_SKETCH_COMPACT = 0, MILAN_SKETCH_INSERT = 1, MILAN_SKETCH_MOVE = 2,
       MILAN_SKETCH_HALVE = 3 };
typedef struct milan_sketch_op {
  int32_t kind, src, dst, offset, extremes, reserved;
  int64_t n, position, capacity;
} milan_sketch_op;
int milan_exemplar_sketch_plan_shift(int64_t resolution, int64_t buffersize,
                                     int full_rate, int n_levels,
                                     const int64_t* capacities,
                                     const int64_t* firstfree,
                                     int (*draw_bit)(void*), void* user,
                                     milan_sketch_op* ops, int max_ops, int* n_ops,
                                     int64_t* capacities_out, int64_t* firstfree_out,
                                     int* n_levels_out);

/* RunningQuantile.quantiles(q) for one q (runningstats.py:531-580): weighted
 * summary of all levels (level l has weight 2^l), stable sort, float32
 * cumulative weights, numpy.interp in float64 -> out [n_units] float32.
 * levels / firstfree / capacities are HOST arrays of n_levels entries (device
 * row pointers, fill counts, row capacities); extremes [n_units][2] is updated
 * with the level-0 remainder like the reference does.  Exact parity needs the
 * total weight (= samples seen) below 2^24 per unit (float32 sums, as in the
 * reference).  Workspace: milan_exemplar_sort_workspace(n_units, sum of
 * firstfree, 1). */
int milan_exemplar_sketch_quantile(const float* const* levels,
                                   const int64_t* firstfree,
                                   const int64_t* capacities, int n_levels,
                                   int n_units, float* extremes, float q,
                                   float* out, void* workspace,
                                   size_t workspace_bytes, milan_stream stream);

/* ImageVisualizer cells (src/deps/ext/netdissect/imgviz.py:56-81): for each of
 * the n_cells rows (batch item, activation channel, unit slot, rank) of `cells`
 * (device int32 [n_cells][4]):
 *   mask   = grid_sample(act, upsample_grid, bilinear, align_corners) > level
 *            (imgviz.py:185-198, upsample.py:6-45,132-156)        -> out_masks
 *   image  = nearest resize of (image * mul + add).clamp(0,255).byte()
 *            (imgviz.py:200-210, renormalize.py:119-136)          -> out_images
 *   masked = image inside the mask, 0.25 * image outside          -> out_masked
 * written at [slot][rank] of the (slots, k, 3|1, out, out) uint8 outputs.
 * images (batch, 3, img_h, img_w) fp32; levels [slots] fp32 (device);
 * mul3 / add3 host float[3]. */
int milan_exemplar_render(const float* hiddens, int batch, int channels, int h,
                          int w, const float* images, int img_h, int img_w,
                          const int32_t* cells, int n_cells, const float* levels,
                          const float* mul3, const float* add3, int out_size,
                          int k, uint8_t* out_images, uint8_t* out_masks,
                          uint8_t* out_masked, milan_stream stream);

/* Building block exposed for kernel-level parity tests: one NHWC fp32
 * convolution through the same implicit-GEMM MFMA kernel the trunk uses
 * (counterpart of torch.nn.functional.conv2d as torchvision's ResNet calls it).
 * x (n,h,w,cin) NHWC with cin % 4 == 0; weight (cout,cin,kh,kw) OIHW as in the
 * reference state dict; bias (cout) or NULL; residual (n,ho,wo,cout) or NULL
 * (residual implies relu(conv + bias + residual)); y (n,ho,wo,cout).
 * precision: MILAN_PRECISION_* (split mode converts x and the weights first).
 * Allocates temporaries and synchronises: not for the hot path. */
int milan_conv2d_nhwc(const float* x, int n, int h, int w, int cin,
                      const float* weight_oihw, const float* bias, int cout,
                      int kh, int kw, int stride, int pad, int relu,
                      const float* residual, float* y, int precision,
                      milan_stream stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* MILAN_HIP_H */
