"""ctypes binding of libmilan_hip.so (the C ABI in include/milan_hip.h).

This is the only place the Python mirror touches native code.  There is NO
fallback: if the library cannot be loaded, or no HIP device is present,
everything that computes raises `HipUnavailableError` -- the product path
never routes through torch ops or the CPU oracle.

Tensors cross the boundary as raw device pointers (`tensor.data_ptr()`) on
torch's current HIP stream; torch only owns memory here.
"""
import ctypes
import os
import pathlib
import weakref
from typing import Dict, Optional, Sequence

import torch

# MILAN_LIB=<path>: load another build of the library (A/B timing of compiler options)
LIB_PATH = pathlib.Path(os.environ.get('MILAN_LIB') or
                       pathlib.Path(__file__).resolve().parent / 'lib' / 'libmilan_hip.so')

GREEDY, FORCED, BEAM, RERANK = 0, 1, 2, 3  # MILAN_GREEDY / _FORCED / _BEAM / _RERANK
PRECISION_F32, PRECISION_SPLIT_F16, PRECISION_F16 = 0, 1, 2
# 'f16' = the fast mode: narrower than the reference's fp32 (include/milan_hip.h), never a default
PRECISIONS = {'f32': PRECISION_F32, 'split_f16': PRECISION_SPLIT_F16, 'f16': PRECISION_F16}
FUSE_CHAIN, FUSE_CHAIN_WIDE, FUSE_STEM, FUSE_CONV3, FUSE_SKIP_EMPTY, FUSE_BNECK, FUSE_SPARSE_TAIL = 1, 2, 4, 8, 16, 32, 64  # milan_set_fusion flags (include/milan_hip.h)
SKETCH_COMPACT, SKETCH_INSERT, SKETCH_MOVE, SKETCH_HALVE = 0, 1, 2, 3
# milan_status bits (include/milan_hip.h)
STATUS_SATURATED, STATUS_NONFINITE_INPUT = 1, 2
# enum milan_kernel_family (milan_profile_read_kernels)
KERNEL_FAMILIES = ('other', 'pp32_256', 'pp32_128', 'split_other', 'f32', 'chain',
                   'chain_wide', 'stem', 'conv3', 'f16', 'pp32t_256', 'bneck')


class SketchOp(ctypes.Structure):
    """`milan_sketch_op` (include/milan_hip.h)."""
    _fields_ = [('kind', ctypes.c_int32), ('src', ctypes.c_int32),
                ('dst', ctypes.c_int32), ('offset', ctypes.c_int32),
                ('extremes', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('n', ctypes.c_int64), ('position', ctypes.c_int64),
                ('capacity', ctypes.c_int64)]


DRAW_BIT = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)
# conv2d_nhwc test hook only: split_f16 with the LDS-strip 3x3 kernel forced
_CONV_PRECISIONS = dict(f32=0, split_f16=1, split_f16_strip=3, split_f16_tap_major=4)
DTYPE_U8, DTYPE_F32 = 0, 1

ERR_ARG, ERR_SHAPE, ERR_STATE, ERR_WORKSPACE, ERR_NO_LM = -1, -2, -3, -4, -5


class HipUnavailableError(RuntimeError):
    """libmilan_hip.so or a HIP device is missing."""


class Dims(ctypes.Structure):
    """struct milan_dims."""
    _fields_ = [
        ('trunk_width', ctypes.c_int32),
        ('trunk_blocks', ctypes.c_int32 * 4),
        ('feature_size', ctypes.c_int32),
        ('hidden_size', ctypes.c_int32),
        ('embedding_size', ctypes.c_int32),
        ('attention_size', ctypes.c_int32),
        ('vocab_size', ctypes.c_int32),
        ('start_index', ctypes.c_int32),
        ('stop_index', ctypes.c_int32),
        ('pad_index', ctypes.c_int32),
        ('has_lm', ctypes.c_int32),
        ('lm_hidden_size', ctypes.c_int32),
        ('lm_embedding_size', ctypes.c_int32),
        ('lm_layers', ctypes.c_int32),
        ('trunk_kind', ctypes.c_int32),
    ]


ABI_VERSION = 8  # MILAN_ABI_VERSION this binding was written against

# milan_dims.trunk_kind and the pyramid width multiplier (F = mult * width)
TRUNK_BOTTLENECK, TRUNK_BASIC, TRUNK_ALEXNET, TRUNK_NONE = 0, 1, 2, 3
_FEATURE_MULT = {TRUNK_BOTTLENECK: 61, TRUNK_BASIC: 16, TRUNK_ALEXNET: 18}


_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_SZ = ctypes.c_size_t
_I64 = ctypes.c_int64

# name -> (restype, argtypes); must list every symbol include/milan_hip.h
# declares (tests/test_host.py checks the two against each other).
SIGNATURES = {
    'milan_abi_version': (_I, []),
    'milan_last_error': (ctypes.c_char_p, []),
    'milan_create': (_I, [ctypes.POINTER(_P), _I,
                          ctypes.POINTER(Dims)]),
    'milan_destroy': (None, [_P]),
    'milan_set_weight':
        (_I, [_P, ctypes.c_char_p, _P,
              ctypes.POINTER(ctypes.c_int64), _I]),
    'milan_finalize_weights': (_I, [_P, _P]),
    'milan_workspace_bytes': (_SZ, [_P, _I, _I, _I, _I, _I]),
    'milan_encode': (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _SZ, _P]),
    'milan_encode_spatial': (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _SZ, _P]),
    'milan_init_state': (_I, [_P, _P, _I, _I, _P, _P, _P, _SZ, _P]),
    'milan_step':
        (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _SZ,
              _P]),
    'milan_decode': (_I, [
        _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P,
        _SZ, _P
    ]),
    'milan_lm_score': (_I, [_P, _P, _I, _I, _P, _P, _P, _SZ, _P]),
    'milan_lm_logprobs': (_I, [_P, _P, _I, _I, _P, _P, _SZ, _P]),
    'milan_describe': (_I, [
        _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P,
        _P, _P, _P, _P, _P, _P, _SZ, _P
    ]),
    'milan_set_graph_capture': (_I, [_P, _I]),
    'milan_graph_stats': (_I, [
        _P, ctypes.POINTER(ctypes.c_longlong),
        ctypes.POINTER(ctypes.c_longlong)
    ]),
    'milan_set_fusion': (_I, [_P, _I]),
    'milan_set_precision': (_I, [_P, _I]),
    'milan_get_precision': (_I, [_P]),
    'milan_profile_enable': (_I, [_I]),
    'milan_profile_read': (_I, [
        ctypes.POINTER(ctypes.c_double),
        ctypes.POINTER(ctypes.c_double),
        ctypes.POINTER(ctypes.c_longlong)
    ]),
    'milan_profile_read_stages': (_I, [ctypes.POINTER(ctypes.c_double)]),
    'milan_profile_read_kernels': (_I, [ctypes.POINTER(ctypes.c_double)]),
    'milan_status': (_I, [_P, ctypes.POINTER(ctypes.c_uint32), _I, _P]),
    'milan_set_act_scale_log2': (_I, [_P, _I, _P]),
    'milan_get_act_scale_log2': (_I, [_P]),
    'milan_encoder_absmax':
        (_I, [_P, _P, _I, _I, _I, _I, ctypes.POINTER(_F), _P, _SZ, _P]),
    'milan_exemplar_topk_update':
        (_I, [_P, _I, _I, _I, _P, _I, _I64, _I, _I, _P, _P, _P, _P]),
    'milan_exemplar_sketch_append':
        (_I, [_P, _I, _I, _I, _P, _I, _I64, _I64, _P, _I64, _I64, _P]),
    'milan_exemplar_sort_workspace': (_SZ, [_I, _I64, _I]),
    'milan_exemplar_sketch_compact':
        (_I, [_P, _I64, _I64, _I, _I, _P, _I64, _I64, _P, _P, _SZ, _P]),
    'milan_exemplar_sketch_add_workspace':
        (_SZ, [_I, _I64, ctypes.POINTER(_I64), _I]),
    'milan_exemplar_sketch_add':
        (_I, [_P, _I, _I, _I, _P, _I, _I64, ctypes.POINTER(_I64),
              ctypes.POINTER(_P), ctypes.POINTER(_I64), ctypes.POINTER(_I64),
              _I, _P, _I64, ctypes.POINTER(_I64), _P, _P, _SZ, _P]),
    'milan_exemplar_rows_extremes': (_I, [_P, _I64, _I, _P, _P]),
    'milan_exemplar_sketch_plan_shift':
        (_I, [_I64, _I64, _I, _I, ctypes.POINTER(_I64), ctypes.POINTER(_I64),
              _P, _P, _P, _I, ctypes.POINTER(_I), ctypes.POINTER(_I64),
              ctypes.POINTER(_I64), ctypes.POINTER(_I)]),
    'milan_exemplar_sketch_quantile':
        (_I, [ctypes.POINTER(_P), ctypes.POINTER(_I64), ctypes.POINTER(_I64),
              _I, _I, _P, _F, _P, _P, _SZ, _P]),
    'milan_exemplar_render':
        (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _P, _I, _P,
              ctypes.POINTER(_F), ctypes.POINTER(_F), _I, _I, _P, _P, _P, _P]),
    'milan_conv2d_nhwc':
        (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I,
              _P]),
}

_lib = None


def load_library(path: Optional[os.PathLike] = None) -> ctypes.CDLL:
    """dlopen libmilan_hip.so and attach signatures.  Loud on failure."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = pathlib.Path(path) if path is not None else LIB_PATH
    if not p.exists():
        raise HipUnavailableError(
            f'{p} not found: build it with `python -c "import __graft_entry__ '
            'as g; g.build()"` (or `make -C neuron-descriptions_amd/csrc`). '
            'There is no CPU fallback for the MILAN hot path.')
    try:
        lib = ctypes.CDLL(str(p))
    except OSError as error:
        raise HipUnavailableError(f'cannot load {p}: {error}') from error
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.milan_abi_version() != ABI_VERSION:
        raise HipUnavailableError(
            f'{p} has C-ABI version {lib.milan_abi_version()}, this binding '
            f'needs {ABI_VERSION} (struct milan_dims differs): rebuild it')
    if path is None:
        _lib = lib
    return lib


def require_device(device: torch.device) -> torch.device:
    device = torch.device(device)
    if device.type != 'cuda' or not torch.cuda.is_available():
        raise HipUnavailableError(
            'milan_amd computes only on an AMD GPU (torch device "cuda" under '
            f'ROCm); got device={device}, cuda.is_available()='
            f'{torch.cuda.is_available()}. There is no CPU fallback.')
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    return device


def _check(code: int) -> None:
    if code == 0:
        return
    msg = load_library().milan_last_error().decode(errors='replace')
    if code in (ERR_ARG, ERR_SHAPE, ERR_NO_LM):
        raise ValueError(msg)  # the reference raises ValueError for these
    raise RuntimeError(f'libmilan_hip error {code}: {msg}')


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _dev(t: torch.Tensor, device: torch.device, dtype=None) -> torch.Tensor:
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.device != device:
        t = t.to(device)
    return t.contiguous()


_LIVE_CONTEXTS: 'weakref.WeakSet' = weakref.WeakSet()


def release_workspaces() -> None:
    """Drop the activation workspace of every live Context (up to 154 GB each at
    chunk_size 640; the next call re-allocates it) and hand torch's cached
    blocks back to the device -- for a process that is about to share its GPU
    with another one."""
    for ctx in list(_LIVE_CONTEXTS):
        ctx._ws = None
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


def _kfd_gpuids(pid) -> set:
    """GPU ids (KFD topology ids) on which process `pid` has compute queues."""
    ids = set()
    try:
        base = f'/sys/class/kfd/kfd/proc/{pid}/queues'
        for q in os.listdir(base):
            try:
                with open(f'{base}/{q}/gpuid') as f:
                    ids.add(f.read().strip())
            except OSError:
                pass
    except OSError:
        pass
    return ids


def other_compute_processes() -> list:
    """PIDs of OTHER processes with compute queues on a GPU this process uses.  Every
    process with a KFD context has a directory /sys/class/kfd/kfd/proc/<pid> whose
    queues/*/gpuid name the devices it runs on; processes on other GPUs of the node (other
    jobs' ranks, other containers) do not count."""
    mine = _kfd_gpuids(os.getpid())
    if not mine:
        return []
    try:
        pids = [int(p) for p in os.listdir('/sys/class/kfd/kfd/proc') if p.isdigit()]
    except OSError:
        return []
    return sorted(p for p in pids if p != os.getpid() and _kfd_gpuids(p) & mine)


_SHARED_WARNED = False


def warn_if_gpu_is_shared() -> None:
    """The library's execution model is one process per GPU and one stream per process
    (INTEGRATION.md): co-resident workgroups of different kernels are not safe on this
    platform (profiles/r4_coresidency_minimal.txt -- a compiler-generated MFMA loop
    corrupts the vector registers of a small kernel sharing its CU).  Ranks of one job
    that were deliberately put on one GPU (tests) see this warning too."""
    global _SHARED_WARNED
    others = other_compute_processes()
    if others and not _SHARED_WARNED:
        _SHARED_WARNED = True
        import warnings
        warnings.warn(
            f'milan_amd: {len(others)} other process(es) have compute queues on a GPU '
            f'this process uses (pids {others[:4]}{"..." if len(others) > 4 else ""}): '
            'kernels of two processes can become co-resident on a CU, which is '
            'unsupported on this platform (see INTEGRATION.md, "One process per GPU")',
            RuntimeWarning, stacklevel=3)


class Context:
    """Owns one `milan_ctx` (packed weights on one GPU) plus a workspace."""

    def __init__(self, dims: Dims, state_dict: Dict[str, torch.Tensor],
                 device: torch.device):
        self.lib = load_library()
        self.device = require_device(device)
        self.dims = dims
        self._h = _P()
        with torch.cuda.device(self.device):
            _check(
                self.lib.milan_create(ctypes.byref(self._h), self.device.index,
                                      ctypes.byref(dims)))
            keep = []
            for name, tensor in state_dict.items():
                if not tensor.dtype.is_floating_point:
                    continue  # e.g. num_batches_tracked
                t = _dev(tensor.detach(), self.device, torch.float32)
                keep.append(t)
                shape = (ctypes.c_int64 * max(1, t.dim()))(*t.shape)
                _check(
                    self.lib.milan_set_weight(self._h, name.encode(),
                                              t.data_ptr(), shape, t.dim()))
            _check(
                self.lib.milan_finalize_weights(self._h,
                                                _stream(self.device)))
            del keep
        self._ws: Optional[torch.Tensor] = None
        # what a saturated split-f16 value does (see `_guarded`): 'raise' (default),
        # 'f32' (rerun the call in the exact-fp32 mode: Decoder.precision = 'auto') or
        # 'ignore' (the caller reads `status()` itself, e.g. once after a timed region)
        self.on_saturation = os.environ.get('MILAN_ON_SATURATION', 'raise')
        self.saturation_fallbacks = 0
        _LIVE_CONTEXTS.add(self)
        warn_if_gpu_is_shared()  # (after the context exists: this process has queues)
        if os.environ.get('MILAN_CHAIN'):  # A/B timing (tools/ab_env.sh): flag bits
            bits = int(os.environ['MILAN_CHAIN'])
            self.set_fusion(chain=bool(bits & 1), wide=bool(bits & 2), stem=bool(bits & 4),
                            conv3=bool(bits & 8), skip_empty=bool(bits & 16),
                            bneck=bool(bits & 32), sparse_tail=bool(bits & 64))
        default = os.environ.get('MILAN_PRECISION')
        if default == 'auto':
            self.set_precision('split_f16')
            self.on_saturation = 'f32'
        elif default:
            self.set_precision(default)

    # -- hipGraph replay of the decode stage ------------------------------------
    def enable_graphs(self, enable: bool = True) -> None:
        """Capture each distinct decode pass into a hipGraph and replay it.

        Pays off for small interactive batches, where the ~450 launches of a
        beam search are latency-bound.  Needs pointer-stable buffers and a
        non-default stream, so outputs then live in per-shape pools (callers get
        clones) and the calls run on a private stream that is joined with the
        caller's stream before and after.
        """
        _check(self.lib.milan_set_graph_capture(self._h, int(enable)))
        self._graphs = bool(enable)
        if enable and getattr(self, '_gstream', None) is None:
            self._gstream = torch.cuda.Stream(device=self.device)
            self._pool = {}

    def graph_stats(self):
        cap, rep = ctypes.c_longlong(), ctypes.c_longlong()
        _check(self.lib.milan_graph_stats(self._h, ctypes.byref(cap),
                                          ctypes.byref(rep)))
        return cap.value, rep.value

    def _pooled(self, key, make):
        buf = self._pool.get(key)
        if buf is None:
            self._pool[key] = buf = make()
        return buf

    def set_precision(self, precision) -> None:
        """'f32' (exact fp32 MFMA, default) or 'split_f16' (3xf16 MFMA)."""
        if isinstance(precision, str):
            if precision not in PRECISIONS:
                raise ValueError(f'unknown precision: {precision}')
            precision = PRECISIONS[precision]
        _check(self.lib.milan_set_precision(self._h, int(precision)))

    def set_fusion(self, chain: bool = True, wide: Optional[bool] = None,
                   stem: bool = True, conv3: bool = True, skip_empty: bool = True,
                   bneck: bool = True, sparse_tail: bool = True) -> None:
        """Cross-layer fusions of the trunk (bitwise-neutral scheduling knob):
        `chain` the expand -> reduce launches of layer1 / layer2, `wide` those of
        layer3 (default: as `chain`), `stem` conv1 + bn1 + ReLU + maxpool as one launch, `conv3` the
        register-resident-weight kernel for layer1's 3x3 convolutions, `skip_empty` exemplars
        with an all-zero mask (exact-zero features by the reference's rule) stay out of the trunk,
        `bneck` (round 6) layer1's 3x3 conv runs in front of its chain launch (needs `chain`),
        `sparse_tail` (round 6) the last two bottlenecks run only at the pixels the level-4 pooling
        -- and their 3x3 neighbourhoods -- read."""
        wide = chain if wide is None else wide
        _check(self.lib.milan_set_fusion(
            self._h, (FUSE_CHAIN if chain else 0) | (FUSE_CHAIN_WIDE if wide else 0) |
            (FUSE_STEM if stem else 0) | (FUSE_CONV3 if conv3 else 0) |
            (FUSE_SKIP_EMPTY if skip_empty else 0) | (FUSE_BNECK if bneck else 0) |
            (FUSE_SPARSE_TAIL if sparse_tail else 0)))

    @property
    def precision(self) -> str:
        code = self.lib.milan_get_precision(self._h)
        return {v: k for k, v in PRECISIONS.items()}[code]

    # -- split-f16 fails loudly ------------------------------------------------------
    def status(self, clear: bool = True) -> int:
        """The context's device status word (STATUS_* bits), read back through a stream
        synchronisation; `clear` resets it."""
        flags = ctypes.c_uint32(0)
        with torch.cuda.device(self.device):
            _check(self.lib.milan_status(self._h, ctypes.byref(flags), int(clear),
                                         _stream(self.device)))
        return int(flags.value)

    @property
    def act_scale_log2(self) -> int:
        return int(self.lib.milan_get_act_scale_log2(self._h))

    def set_act_scale_log2(self, k: int) -> None:
        """Activation scale 2^k of the split-f16 trunk (0..10): `hi` saturates at
        65504 / 2^k, `lo` keeps full precision down to activations of 0.125 / 2^k."""
        with torch.cuda.device(self.device):
            _check(self.lib.milan_set_act_scale_log2(self._h, int(k),
                                                     _stream(self.device)))

    def encoder_absmax(self, images: torch.Tensor) -> float:
        """Largest |activation| any tensor of the ResNet trunk holds for `images`
        ((M,3,H,W), uint8 or float), measured in the exact-fp32 mode."""
        m, ch, h, w = images.shape
        if ch != 3:
            raise ValueError(f'images must have 3 channels, got {ch}')
        idt = DTYPE_U8 if images.dtype == torch.uint8 else DTYPE_F32
        images = _dev(images, self.device, None if idt == DTYPE_U8 else torch.float32)
        ws = self.workspace(m or 1, 1, max(h, w), 1, 1)
        out = _F(0.0)
        with torch.cuda.device(self.device):
            _check(self.lib.milan_encoder_absmax(self._h, images.data_ptr(), idt, m, h, w,
                                                 ctypes.byref(out), ws.data_ptr(),
                                                 ws.numel(), _stream(self.device)))
        return float(out.value)

    def calibrate(self, images: torch.Tensor, headroom: float = 8.0) -> int:
        """Pick the split trunk's activation scale from a sample: the largest power of
        two that keeps `headroom` x the observed maximum below the f16 range.  Returns
        the chosen log2 scale.  (The default 2^5 suits activations up to 2047; a
        network whose residual stream grows beyond that needs a smaller scale, one
        whose activations sit at 1e-3 a larger one -- DESIGN.md section 4.2.)"""
        import math
        amax = self.encoder_absmax(images)
        if not math.isfinite(amax):
            raise FloatingPointError('calibration sample produced non-finite activations')
        k = 10 if amax <= 0 else int(math.floor(math.log2(65504.0 / (amax * headroom))))
        k = max(0, min(10, k))
        self.set_act_scale_log2(k)
        return k

    def _guarded(self, what: str, run, check: bool = True):
        """Run one computing call and act on the status word it leaves behind.

        MILAN_STATUS_SATURATED: a split-format value hit the +-65504 clamp -- the
        reference's fp32 would not have, so the result is NOT returned: raise
        FloatingPointError, or (on_saturation == 'f32') rerun the call in the exact-fp32
        mode.  MILAN_STATUS_NONFINITE_INPUT: a float input pixel was NaN / Inf; the
        features of its image are NaN as in the reference -- a RuntimeWarning says so."""
        out = run()
        if not check or self.on_saturation == 'ignore':
            return out
        flags = self.status(clear=True)
        if flags & STATUS_NONFINITE_INPUT:
            import warnings
            warnings.warn(f'milan_amd: {what}: an input image holds NaN / Inf pixels; its '
                          'features are NaN (as in the reference)', RuntimeWarning,
                          stacklevel=3)
        if flags & STATUS_SATURATED:
            limit = 65504.0 / 2 ** self.act_scale_log2
            msg = (f'milan_amd: {what}: a split-f16 activation exceeded {limit:g} '
                   f'(65504 / 2^{self.act_scale_log2}) and was clamped; the exact-fp32 '
                   "mode (precision='f32'), Decoder.precision = 'auto' or a smaller "
                   'activation scale (Context.calibrate) avoid it')
            if self.on_saturation != 'f32' or self.precision == 'f32':
                raise FloatingPointError(msg)
            import warnings
            warnings.warn(msg + ' -- rerunning this call in f32', RuntimeWarning,
                          stacklevel=3)
            self.saturation_fallbacks += 1
            saved = self.precision
            self.set_precision('f32')
            try:
                out = run()
                self.status(clear=True)
            finally:
                self.set_precision(saved)
        return out

    def close(self) -> None:
        if getattr(self, '_h', None) is not None and self._h:
            self.lib.milan_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter teardown
            pass

    # -- workspace ---------------------------------------------------------
    def workspace(self, neurons: int, k: int, image_size: int, beam: int,
                  length: int) -> torch.Tensor:
        need = int(
            self.lib.milan_workspace_bytes(self._h, neurons, k, image_size,
                                           beam, length))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None  # free before growing
            try:
                self._ws = torch.empty(need, dtype=torch.uint8,
                                       device=self.device)
            except torch.cuda.OutOfMemoryError as error:
                free, total = torch.cuda.mem_get_info(self.device)
                raise torch.cuda.OutOfMemoryError(
                    f'the activation workspace for {neurons} neurons x {k} '
                    f'exemplars needs {need / 2**30:.1f} GiB, {free / 2**30:.1f} '
                    f'of {total / 2**30:.1f} GiB are free: lower '
                    'Decoder.chunk_size (or use Context.fit_neurons)') from error
        return self._ws

    def fit_neurons(self, neurons: int, k: int, image_size: int, beam: int,
                    length: int, multiple: int = 16,
                    headroom: float = 0.9) -> int:
        """Largest neuron count <= `neurons` (a multiple of `multiple`) whose
        workspace fits into `headroom` of the memory this process can still
        get (free memory + what the current workspace already holds).  The
        default chunk of 640 neurons takes 154 GB: fine on an idle 288 GB
        MI355X, not when the GPU is shared."""
        free, _ = torch.cuda.mem_get_info(self.device)
        budget = headroom * (free + (self._ws.numel() if self._ws is not None
                                     else 0))

        def need(n):
            return int(self.lib.milan_workspace_bytes(self._h, n, k, image_size,
                                                      beam, length))

        n = max(1, neurons)
        if need(n) <= budget:
            return n
        lo, hi = 0, n  # need(lo) fits (or lo == 0), need(hi) does not
        while hi - lo > multiple:
            mid = (lo + hi) // 2 // multiple * multiple
            if mid <= lo:
                break
            if need(mid) <= budget:
                lo = mid
            else:
                hi = mid
        if lo <= 0:
            lo = min(multiple, n)
        return lo

    # -- operators ----------------------------------------------------------
    def encode(self, images: torch.Tensor,
               masks: Optional[torch.Tensor], check: bool = True) -> torch.Tensor:
        """(M,3,H,W) [+ (M,1,H,W)] -> (M,F).  uint8 or float inputs.  `check`: read the
        status word afterwards (one stream synchronisation; see `_guarded`)."""
        m, ch, h, w = images.shape
        if ch != 3:
            raise ValueError(f'images must have 3 channels, got {ch}')
        idt = DTYPE_U8 if images.dtype == torch.uint8 else DTYPE_F32
        images = _dev(images, self.device,
                      None if idt == DTYPE_U8 else torch.float32)
        mdt = DTYPE_U8
        if masks is not None:
            if masks.shape != (m, 1, h, w):
                raise ValueError(
                    f'masks shape {tuple(masks.shape)} != {(m, 1, h, w)}')
            mdt = DTYPE_U8 if masks.dtype == torch.uint8 else DTYPE_F32
            masks = _dev(masks, self.device,
                         None if mdt == DTYPE_U8 else torch.float32)
        out = torch.empty(m,
                          self.dims.feature_size,
                          dtype=torch.float32,
                          device=self.device)
        ws = self.workspace((m + 0) or 1, 1, max(h, w), 1, 1)

        def run():
            with torch.cuda.device(self.device):
                _check(
                    self.lib.milan_encode(self._h, images.data_ptr(), idt,
                                          _ptr(masks), mdt, m, h, w,
                                          out.data_ptr(), ws.data_ptr(),
                                          ws.numel(), _stream(self.device)))
            return out

        return self._guarded('encode', run, check)

    def encode_spatial(self, images: torch.Tensor,
                       masks: Optional[torch.Tensor], check: bool = True) -> torch.Tensor:
        """SpatialConvEncoder: (M,3,H,W) [+ (M,1,H,W)] -> (M, h4*w4, C4)."""
        m, ch, h, w = images.shape
        if ch != 3:
            raise ValueError(f'images must have 3 channels, got {ch}')
        idt = DTYPE_U8 if images.dtype == torch.uint8 else DTYPE_F32
        images = _dev(images, self.device,
                      None if idt == DTYPE_U8 else torch.float32)
        mdt = DTYPE_U8
        if masks is not None:
            if masks.shape != (m, 1, h, w):
                raise ValueError(
                    f'masks shape {tuple(masks.shape)} != {(m, 1, h, w)}')
            mdt = DTYPE_U8 if masks.dtype == torch.uint8 else DTYPE_F32
            masks = _dev(masks, self.device,
                         None if mdt == DTYPE_U8 else torch.float32)

        def down(x, k, s, p):
            return (x + 2 * p - k) // s + 1

        h4, w4 = down(h, 7, 2, 3), down(w, 7, 2, 3)
        for _ in range(4):  # maxpool + the three stride-2 stages
            h4, w4 = down(h4, 3, 2, 1), down(w4, 3, 2, 1)
        mult = 8 if self.dims.trunk_kind == TRUNK_BASIC else 32
        out = torch.empty(m, h4 * w4, mult * self.dims.trunk_width,
                          dtype=torch.float32, device=self.device)
        ws = self.workspace(m or 1, 1, max(h, w), 1, 1)

        def run():
            with torch.cuda.device(self.device):
                _check(
                    self.lib.milan_encode_spatial(self._h, images.data_ptr(), idt,
                                                  _ptr(masks), mdt, m, h, w,
                                                  out.data_ptr(), ws.data_ptr(),
                                                  ws.numel(), _stream(self.device)))
            return out

        return self._guarded('encode_spatial', run, check)

    def init_state(self, features: torch.Tensor):
        n, k, _ = features.shape
        features = _dev(features, self.device, torch.float32)
        h = torch.empty(n, self.dims.hidden_size, device=self.device)
        c = torch.empty_like(h)
        ws = self.workspace(n, k, 0, 1, 1)
        with torch.cuda.device(self.device):
            _check(
                self.lib.milan_init_state(self._h, features.data_ptr(), n, k,
                                          h.data_ptr(), c.data_ptr(),
                                          ws.data_ptr(), ws.numel(),
                                          _stream(self.device)))
        return h, c

    def step(self, features, tokens, h, c, h_lm, c_lm, temperature):
        rows, k, _ = features.shape
        features = _dev(features, self.device, torch.float32)
        tokens = _dev(tokens, self.device, torch.long)
        h = _dev(h, self.device, torch.float32)
        c = _dev(c, self.device, torch.float32)
        if h_lm is not None:
            h_lm = _dev(h_lm, self.device, torch.float32).clone()
            c_lm = _dev(c_lm, self.device, torch.float32).clone()
        pred = torch.empty(rows, self.dims.vocab_size, device=self.device)
        att = torch.empty(rows, k, device=self.device)
        h2, c2 = torch.empty_like(h), torch.empty_like(c)
        ws = self.workspace(rows, k, 0, 1, 1)
        with torch.cuda.device(self.device):
            _check(
                self.lib.milan_step(self._h, features.data_ptr(), rows, k,
                                    tokens.data_ptr(), h.data_ptr(),
                                    c.data_ptr(), _ptr(h_lm), _ptr(c_lm),
                                    float(temperature), pred.data_ptr(),
                                    att.data_ptr(), h2.data_ptr(),
                                    c2.data_ptr(), ws.data_ptr(), ws.numel(),
                                    _stream(self.device)))
        return pred, att, h2, c2, h_lm, c_lm

    def _alloc_outputs(self, n, k, strategy, length, beam, want_full,
                       forced=None):
        dev = self.device
        if strategy == FORCED:
            # teacher forcing: `tokens` is the INPUT of milan_decode
            if forced is None or tuple(forced.shape) != (n, length):
                raise ValueError('FORCED decoding needs forced tokens of shape '
                                 f'({n}, {length})')
            tokens = forced.to(device=dev, dtype=torch.long).contiguous().clone()
            want_full = True
        else:
            tokens = torch.empty(n, length, dtype=torch.long, device=dev)
        out = dict(tokens=tokens,
                   scores=torch.empty(n, device=dev),
                   predictions=None,
                   attentions=None,
                   beam_tokens=None,
                   beam_scores=None,
                   out_len=None)
        if strategy in (GREEDY, FORCED):
            if want_full:
                out['predictions'] = torch.empty(n,
                                                 length,
                                                 self.dims.vocab_size,
                                                 device=dev)
                out['attentions'] = torch.empty(n, length, k, device=dev)
        else:
            out['beam_tokens'] = torch.empty(n,
                                             beam,
                                             length,
                                             dtype=torch.long,
                                             device=dev)
            out['beam_scores'] = torch.empty(n, beam, device=dev)
        return out

    def decode(self,
               features: torch.Tensor,
               strategy: int,
               length: int,
               beam: int,
               mi: bool,
               temperature: float,
               group_size: int = 0,
               want_full: bool = True,
               forced: Optional[torch.Tensor] = None):
        n, k, _ = features.shape
        features = _dev(features, self.device, torch.float32)
        if getattr(self, '_graphs', False) and strategy != FORCED:
            return self._decode_graphed(features, strategy, length, beam, mi,
                                        temperature, group_size, want_full)
        out = self._alloc_outputs(n, k, strategy, length, beam, want_full,
                                  forced)
        groups = (n + group_size - 1) // group_size if group_size > 0 else 1
        out['out_len'] = torch.full((groups,),
                                    length,
                                    dtype=torch.int32,
                                    device=self.device)
        ws = self.workspace(n, k, 0, beam, length)
        with torch.cuda.device(self.device):
            _check(
                self.lib.milan_decode(
                    self._h, features.data_ptr(), n, k, strategy, length, beam,
                    int(bool(mi)), float(temperature), group_size,
                    out['tokens'].data_ptr(), out['scores'].data_ptr(),
                    _ptr(out['predictions']), _ptr(out['attentions']),
                    _ptr(out['beam_tokens']), _ptr(out['beam_scores']),
                    out['out_len'].data_ptr(), ws.data_ptr(), ws.numel(),
                    _stream(self.device)))
        return out

    def _decode_graphed(self, features, strategy, length, beam, mi, temperature,
                        group_size, want_full):
        n, k, fsz = features.shape
        sig = (n, k, strategy, length, beam, bool(mi), group_size, want_full)
        groups = (n + group_size - 1) // group_size if group_size > 0 else 1

        def make():
            out = self._alloc_outputs(n, k, strategy, length, beam, want_full)
            out['out_len'] = torch.empty(groups, dtype=torch.int32,
                                         device=self.device)
            out['_features'] = torch.empty(n, k, fsz, device=self.device)
            return out

        cur = torch.cuda.current_stream(self.device)
        gs = self._gstream
        gs.wait_stream(cur)
        with torch.cuda.device(self.device), torch.cuda.stream(gs):
            pool = self._pooled(sig, make)
            ws = self.workspace(n, k, 0, beam, length)
            pool['_features'].copy_(features)
            _check(
                self.lib.milan_decode(
                    self._h, pool['_features'].data_ptr(), n, k, strategy,
                    length, beam, int(bool(mi)), float(temperature), group_size,
                    pool['tokens'].data_ptr(), pool['scores'].data_ptr(),
                    _ptr(pool['predictions']), _ptr(pool['attentions']),
                    _ptr(pool['beam_tokens']), _ptr(pool['beam_scores']),
                    pool['out_len'].data_ptr(), ws.data_ptr(), ws.numel(),
                    gs.cuda_stream))
            out = {key: (v.clone() if isinstance(v, torch.Tensor) else v)
                   for key, v in pool.items() if key != '_features'}
        cur.wait_stream(gs)
        for v in out.values():
            if isinstance(v, torch.Tensor):
                v.record_stream(cur)
        return out

    def lm_score(self, seqs: torch.Tensor) -> torch.Tensor:
        rows, length = seqs.shape
        seqs = _dev(seqs, self.device, torch.long)
        out = torch.empty(rows, device=self.device)
        ws = self.workspace(rows, 1, 0, 1, 1)
        with torch.cuda.device(self.device):
            _check(
                self.lib.milan_lm_score(self._h, seqs.data_ptr(), rows, length,
                                        None, out.data_ptr(), ws.data_ptr(),
                                        ws.numel(), _stream(self.device)))
        return out

    def lm_logprobs(self, seqs: torch.Tensor) -> torch.Tensor:
        """(rows, L) token ids -> (rows, L, V) next-token log-probs."""
        rows, length = seqs.shape
        seqs = _dev(seqs, self.device, torch.long)
        out = torch.empty(rows, length, self.dims.vocab_size,
                          device=self.device)
        ws = self.workspace(rows, 1, 0, 1, 1)
        with torch.cuda.device(self.device):
            _check(
                self.lib.milan_lm_logprobs(self._h, seqs.data_ptr(), rows,
                                           length, out.data_ptr(),
                                           ws.data_ptr(), ws.numel(),
                                           _stream(self.device)))
        return out

    def describe(self,
                 images: torch.Tensor,
                 masks: Optional[torch.Tensor],
                 strategy: int,
                 length: int,
                 beam: int,
                 mi: bool,
                 temperature: float,
                 group_size: int = 0,
                 want_full: bool = False,
                 want_features: bool = False,
                 forced: Optional[torch.Tensor] = None,
                 check: bool = True):
        """Fused hot path on (n,k,3,H,W) images (+ (n,k,1,H,W) masks).  `check=False`
        skips the status read (a stream synchronisation) -- the caller then reads
        `status()` itself, as bench.py does once after its timed region."""
        n, k, ch, h, w = images.shape
        idt = DTYPE_U8 if images.dtype == torch.uint8 else DTYPE_F32
        images = _dev(images, self.device,
                      None if idt == DTYPE_U8 else torch.float32)
        mdt = DTYPE_U8
        if masks is not None:
            mdt = DTYPE_U8 if masks.dtype == torch.uint8 else DTYPE_F32
            masks = _dev(masks, self.device,
                         None if mdt == DTYPE_U8 else torch.float32)
        out = self._alloc_outputs(n, k, strategy, length, beam, want_full,
                                  forced)
        groups = (n + group_size - 1) // group_size if group_size > 0 else 1
        out['out_len'] = torch.full((groups,),
                                    length,
                                    dtype=torch.int32,
                                    device=self.device)
        feats = None
        if want_features:
            feats = torch.empty(n,
                                k,
                                self.dims.feature_size,
                                device=self.device)
        out['features'] = feats
        ws = self.workspace(n, k, max(h, w), beam, length)

        def run():
            with torch.cuda.device(self.device):
                _check(
                    self.lib.milan_describe(
                        self._h, images.data_ptr(), idt, _ptr(masks), mdt, n, k, h,
                        w, strategy, length, beam, int(bool(mi)),
                        float(temperature), group_size, _ptr(feats),
                        out['tokens'].data_ptr(), out['scores'].data_ptr(),
                        _ptr(out['predictions']), _ptr(out['attentions']),
                        _ptr(out['beam_tokens']), _ptr(out['beam_scores']),
                        out['out_len'].data_ptr(), ws.data_ptr(), ws.numel(),
                        _stream(self.device)))
            return out

        return self._guarded('describe', run, check)


def profile_enable(enable: bool) -> None:
    _check(load_library().milan_profile_enable(int(enable)))


def profile_read():
    """-> (gemm_ms, gemm_flops, gemm_launches) since profile_enable(True)."""
    ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
    _check(load_library().milan_profile_read(ctypes.byref(ms), ctypes.byref(fl),
                                             ctypes.byref(n)))
    return ms.value, fl.value, n.value


STAGES = ('other', 'enc_input', 'enc_stem', 'enc_stem_tail', 'enc_layer1',
          'enc_layer2', 'enc_layer3', 'enc_layer4', 'enc_pool', 'dec_init',
          'dec_search', 'dec_lm')  # enum milan_stage


def profile_read_stages() -> Dict[str, Dict[str, float]]:
    """Per-stage HIP-event timings since profile_enable(True): region ms and
    count, plus the GEMM ms / algorithmic FLOPs / launches / algorithmic HBM
    bytes inside the stage."""
    table = (ctypes.c_double * (len(STAGES) * 6))()
    _check(load_library().milan_profile_read_stages(table))
    keys = ('region_ms', 'regions', 'gemm_ms', 'gemm_flops', 'gemm_launches',
            'gemm_bytes')
    return {name: dict(zip(keys, table[i * 6:i * 6 + 6]))
            for i, name in enumerate(STAGES)}


def profile_read_kernels() -> Dict[str, Dict[str, float]]:
    """The same records by kernel family (enum milan_kernel_family): summed launch ms,
    algorithmic FLOPs, launches, algorithmic HBM bytes."""
    table = (ctypes.c_double * (len(KERNEL_FAMILIES) * 4))()
    _check(load_library().milan_profile_read_kernels(table))
    keys = ('ms', 'flops', 'launches', 'bytes')
    return {name: dict(zip(keys, table[i * 4:i * 4 + 4]))
            for i, name in enumerate(KERNEL_FAMILIES)}


def conv2d_nhwc(x: torch.Tensor,
                weight: torch.Tensor,
                bias: Optional[torch.Tensor] = None,
                stride: int = 1,
                padding: int = 0,
                relu: bool = False,
                residual: Optional[torch.Tensor] = None,
                precision: str = 'f32') -> torch.Tensor:
    """Test hook for the implicit-GEMM kernel (milan_conv2d_nhwc)."""
    lib = load_library()
    device = require_device(x.device)
    n, h, w, cin = x.shape
    cout, cin_w, kh, kw = weight.shape
    assert cin_w == cin
    ho = (h + 2 * padding - kh) // stride + 1
    wo = (w + 2 * padding - kw) // stride + 1
    y = torch.empty(n, ho, wo, cout, device=device)
    x, weight = x.contiguous(), weight.contiguous()
    with torch.cuda.device(device):
        _check(
            lib.milan_conv2d_nhwc(x.data_ptr(), n, h, w, cin,
                                  weight.data_ptr(), _ptr(bias), cout, kh, kw,
                                  stride, padding, int(relu), _ptr(residual),
                                  y.data_ptr(), _CONV_PRECISIONS[precision],
                                  _stream(device)))
    return y


def make_dims(state_dict: Dict[str, torch.Tensor],
              n_vocab_tokens: int,
              blocks: Sequence[int] = (3, 4, 23, 3)) -> Dims:
    """Derive `milan_dims` from a reference state dict + vocabulary size."""
    d = Dims()
    sd = state_dict
    pre = 'encoder.encoder.model.'
    width = kind = None
    if pre + 'features.0.weight' in sd:  # 'alexnet' config
        kind, width = TRUNK_ALEXNET, sd[pre + 'features.0.weight'].shape[0]
    elif pre + 'conv1.weight' in sd:
        width = sd[pre + 'conv1.weight'].shape[0]
        kind = (TRUNK_BOTTLENECK if pre + 'layer1.0.conv3.weight' in sd else
                TRUNK_BASIC)
    if 'lstm.weight_hh' in sd:
        hidden = sd['lstm.weight_hh'].shape[1]
        emb = sd['embedding.weight'].shape[1]
        feat = sd['lstm.weight_ih'].shape[1] - emb
        att = sd['attend.query_to_hidden.weight'].shape[0]
        vocab = sd['output.1.weight'].shape[0]
    else:
        hidden, emb, att, vocab = 4, 4, 4, n_vocab_tokens + 4
        # encoder-only context, or a standalone LanguageModel (no trunk either)
        feat = _FEATURE_MULT[kind] * width if kind is not None else 4
    if width is None:
        # decoder-only context (foreign Encoder, like the reference accepts any
        # `feature_shape`): no trunk, feature size only GEMM-aligned
        if feat % 4:
            raise ValueError(f'feature size {feat} must be a multiple of 4')
        kind, width = TRUNK_NONE, 0
    d.trunk_kind = kind
    if vocab != n_vocab_tokens + 4:
        raise ValueError(
            f'output layer has {vocab} classes but the indexer has '
            f'{n_vocab_tokens} tokens + 4 specials')
    d.trunk_width = width
    for i, b in enumerate(blocks):
        d.trunk_blocks[i] = b
    d.feature_size, d.hidden_size, d.embedding_size = feat, hidden, emb
    d.attention_size, d.vocab_size = att, vocab
    d.start_index = n_vocab_tokens
    d.stop_index = n_vocab_tokens + 1
    d.pad_index = n_vocab_tokens + 2
    d.has_lm = int('lm.lstm.weight_hh_l0' in sd)
    if d.has_lm:
        d.lm_hidden_size = sd['lm.lstm.weight_hh_l0'].shape[1]
        d.lm_embedding_size = sd['lm.embedding.weight'].shape[1]
        layers = 0
        while f'lm.lstm.weight_hh_l{layers}' in sd:
            layers += 1
        d.lm_layers = layers
    return d
