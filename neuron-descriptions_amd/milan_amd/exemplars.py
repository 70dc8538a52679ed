"""Exemplar computation on MI355X: the writer of `images.npy` / `masks.npy`.

Mirror of the reference's `src/exemplars/compute.py` (`compute` :27-246,
`discriminative` :263-353, `generative` :356-437) -- same function names,
arguments, output files (`images.npy`, `masks.npy`, `units.npy`, `ids.csv`,
`activations.csv`) and `ValueError`s -- with the netdissect machinery it calls
(`RunningTopK`, `RunningQuantile`, `ImageVisualizer`, vendored under
`src/deps/netdissect`) replaced by the HIP kernels of `csrc/exemplars.hip`.

What stays torch: the dissected model itself (the reference takes it as two
black-box callables) and the `DataLoader`s.  What stays host Python, as in the
reference: the control flow of the KLL quantile sketch -- which level is
compacted when and with which random bit.  The bits come from torch's global
generator in the reference's order, so a seeded run reproduces the reference's
quantile levels bit for bit even in the sketch's randomised regime.
There is no CPU fallback: without the library / a GPU everything here raises.
"""
import ctypes
import math
import pathlib
import shutil
from typing import Any, Callable, Dict, Optional, Sequence, Tuple, Union

import numpy
import torch
from torch import nn
from torch.utils import data

from milan_amd import hip

PathLike = Union[str, pathlib.Path]


# ---------------------------------------------------------------------------
# transforms (src/exemplars/transforms.py)
# ---------------------------------------------------------------------------
def map_location(items: Sequence[Any], device) -> Tuple[Any, ...]:
    return tuple(item.to(device) if isinstance(item, torch.Tensor) and
                 device is not None else item for item in items)


def first(*inputs: Any) -> Tuple[Any, ...]:
    return (inputs[0],)


def identity(inputs):
    return inputs


def identities(*inputs):
    return inputs


def spatialize_vit_mlp(hiddens: torch.Tensor) -> torch.Tensor:
    """transforms.py:56-81: (batch, patches, units) -> (batch, units, s, s)."""
    batch_size, n_patches, n_units = hiddens.shape
    hiddens = hiddens[:, 1:]
    size = math.isqrt(n_patches - 1)
    assert size**2 == n_patches - 1
    return hiddens.permute(0, 2, 1).reshape(batch_size, n_units, size, size)


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _normalise_units(units: Sequence[int], channels: int):
    """Unit indices as `pooled[:, units]` reads them (compute.py:331-333): negative
    ones count from the end, out-of-range ones raise IndexError."""
    out = []
    for unit in units:
        unit = int(unit)
        if not -channels <= unit < channels:
            raise IndexError(f'index {unit} is out of bounds for dimension 1 '
                             f'with size {channels}')
        out.append(unit % channels)
    return out


def _checked_units(owner, units: Optional[torch.Tensor], channels: int):
    """Validate a device tensor of unit indices once per (tensor, channels): the
    kernels index `units[u]` without a bounds check."""
    if units is None:
        return None
    # the entry keeps `units` itself alive (its address cannot be handed to another
    # tensor meanwhile) and `_version` notices in-place edits
    key = (units.data_ptr(), len(units), channels, units._version)
    cache = owner.__dict__.setdefault('_units_checked', {})
    if key not in cache:
        host = units.detach().cpu().tolist()
        norm = _normalise_units(host, channels)
        cache.clear()  # one live units tensor per owner is the use
        cache[key] = (units, units if norm == host else torch.tensor(
            norm, dtype=torch.int32, device=units.device))
    return cache[key][1]


def _as_hiddens(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


# ---------------------------------------------------------------------------
# RunningTopK (src/deps/netdissect/runningstats.py:31-151)
# ---------------------------------------------------------------------------
class RunningTopK:
    """Per-unit k largest values + their dataset indices, on the GPU."""

    def __init__(self, k: int = 100, device=None):
        self.k, self.count = k, 0
        self.device = device
        self.values: Optional[torch.Tensor] = None
        self.index: Optional[torch.Tensor] = None
        self.filled = 0
        self.lib = hip.load_library()

    def size(self) -> int:
        return self.count

    def add_hiddens(self, hiddens: torch.Tensor,
                    units: Optional[torch.Tensor] = None) -> None:
        """`hiddens` (batch, channels, *spatial): spatial max per unit
        (compute.py:331) merged into the running top-k."""
        hiddens = _as_hiddens(hiddens)
        device = hip.require_device(hiddens.device)
        batch, channels = hiddens.shape[:2]
        units = _checked_units(self, units, channels)
        hw = int(numpy.prod(hiddens.shape[2:])) if hiddens.dim() > 2 else 1
        n_units = channels if units is None else len(units)
        if self.values is None:
            self.device = device
            self.values = torch.zeros(n_units, self.k, device=device)
            self.index = torch.zeros(n_units, self.k, dtype=torch.long,
                                     device=device)
        # the merge sorts filled + batch <= 2048 candidates in LDS
        step = max(1, 2048 - self.k)
        for lo in range(0, batch, step):
            part = hiddens[lo:lo + step]
            scratch = torch.empty(n_units * len(part), device=device)
            with torch.cuda.device(device):
                hip._check(self.lib.milan_exemplar_topk_update(
                    part.data_ptr(), len(part), channels, hw, hip._ptr(units),
                    n_units, self.count + lo, self.k, self.filled,
                    scratch.data_ptr(), self.values.data_ptr(),
                    self.index.data_ptr(), _stream(device)))
            self.filled = min(self.k, self.filled + len(part))
        self.count += batch

    def add(self, data_: torch.Tensor) -> None:
        """Reference contract: `data_` (observations, units) already pooled."""
        self.add_hiddens(data_)

    def result(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(values, dataset indices), each (units, min(k, seen)), best first."""
        return (self.values[:, :self.filled].clone(),
                self.index[:, :self.filled].clone())

    def to_(self, device) -> None:  # results are read with .result().cpu()
        pass

    # -- persistence: field names of netdissect's RunningTopK (runningstats.py:118-149)
    def state_dict(self) -> Dict[str, Any]:
        """netdissect's layout (runningstats.py:118-134), loadable by its own class: a
        candidate buffer of max(10, 5 k) columns per unit of which the first `next`
        are filled, and `linear_index` = row offsets into the flattened buffer, shape
        (units, 1) -- its `result()` adds them to a (units, k) index tensor."""
        values, index = self.result()
        units, filled = values.shape
        width = max(10, 5 * self.k)
        top_data = numpy.zeros((units, width), dtype=numpy.float32)
        top_index = numpy.zeros((units, width), dtype=numpy.int64)
        top_data[:, :filled] = values.cpu().numpy()
        top_index[:, :filled] = index.cpu().numpy()
        return dict(constructor=f'{__name__}.RunningTopK()', k=self.k,
                    count=self.count, largest=True, data_shape=(units,),
                    top_data=top_data, top_index=top_index, next=filled,
                    linear_index=(numpy.arange(units, dtype=numpy.int64)
                                  * width)[:, None],
                    perm=None)

    def set_state_dict(self, dic) -> None:
        self.k = int(dic['k'])
        self.count = int(dic['count'])
        width = int(dic['next'])
        data = torch.from_numpy(numpy.asarray(dic['top_data']))[:, :width]
        index = torch.from_numpy(numpy.asarray(dic['top_index']))
        # upstream keeps an unsorted buffer of up to 5k candidates per unit: its
        # `result()` is a top-k over it (value descending), read back through the
        # flat `linear_index` offsets
        keep = min(self.k, width)
        values, where = data.float().topk(keep, dim=1, sorted=True)
        linear = torch.from_numpy(numpy.asarray(dic['linear_index'])).view(-1, 1)
        picked = index.reshape(-1)[(where + linear).reshape(-1)].view_as(where)
        device = self.device or hip.require_device('cuda')
        self.device = device
        self.values = torch.zeros(data.shape[0], self.k, device=device)
        self.index = torch.zeros(data.shape[0], self.k, dtype=torch.long,
                                 device=device)
        self.values[:, :keep] = values.to(device)
        self.index[:, :keep] = picked.long().to(device)
        self.filled = keep


# ---------------------------------------------------------------------------
# RunningQuantile (src/deps/netdissect/runningstats.py:274-627)
# ---------------------------------------------------------------------------
class RunningQuantile:
    """The KLL sketch with the reference's state machine; tensors on the GPU."""

    def __init__(self, r: int = 3 * 1024, buffersize: Optional[int] = None):
        self.depth = None
        self.device = None
        self.resolution = r * 2
        if buffersize is None:
            buffersize = min(128, (self.resolution + 7) // 8)
        self.buffersize = buffersize
        self.samplerate = 1.0
        self.data = None
        self.firstfree = [0]
        self.randbits = torch.ByteTensor(self.resolution)
        self.currentbit = len(self.randbits) - 1
        self.extremes = None
        self.count = 0
        self.batchcount = 0
        self._ws = None
        self.bulk = True  # False: one launch per append / compaction
        self.lib = hip.load_library()

    def size(self) -> int:
        return self.count

    # -- plumbing ------------------------------------------------------------
    def _lazy_init(self, depth: int, device) -> None:
        self.depth, self.device = depth, hip.require_device(device)
        self.data = [torch.zeros(depth, self.resolution, device=self.device)]
        self.extremes = torch.zeros(depth, 2, device=self.device)
        self.extremes[:, 0] = float('inf')
        self.extremes[:, 1] = -float('inf')

    def _workspace(self, n: int, pairs: bool) -> torch.Tensor:
        need = int(self.lib.milan_exemplar_sort_workspace(self.depth, n,
                                                          int(pairs)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _compact(self, src, n, offset, dst, position, extremes: bool) -> int:
        """sort src[:, :n]; dst[:, position:] = sorted[:, offset::2]."""
        # rows of <= 8192 samples are sorted in LDS by one fused kernel; only
        # a sketch built with r > 4096 needs the library sort's workspace
        ws = self._workspace(n, False) if n > 8192 else None
        with torch.cuda.device(self.device):
            hip._check(self.lib.milan_exemplar_sketch_compact(
                src.data_ptr(), src.shape[1], n, self.depth, offset,
                dst.data_ptr(), dst.shape[1], position,
                self.extremes.data_ptr() if extremes else None, hip._ptr(ws),
                0 if ws is None else ws.numel(), _stream(self.device)))
        return (n - offset + 1) // 2

    # -- the reference's state machine ------------------------------------------
    def add_hiddens(self, hiddens: torch.Tensor,
                    units: Optional[torch.Tensor] = None) -> None:
        """Every spatial position of `hiddens` (batch, channels, *spatial) is
        one sample per unit (compute.py:329-330)."""
        hiddens = _as_hiddens(hiddens)
        batch, channels = hiddens.shape[:2]
        units = _checked_units(self, units, channels)
        hw = int(numpy.prod(hiddens.shape[2:])) if hiddens.dim() > 2 else 1
        depth = channels if units is None else len(units)
        if self.depth is None:
            self._lazy_init(depth, hiddens.device)
        assert depth == self.depth, (depth, self.depth)
        supplied = batch * hw
        self.count += supplied
        self.batchcount += 1
        if self.samplerate < 1.0:
            # subsampling regime (runningstats.py:359-367): every item still counts
            # for the extremes, a Bernoulli(samplerate) portion of each chunk enters
            rows = self._rows(hiddens, units)
            self._scan_extremes(rows)
            chunk = int(math.ceil(self.buffersize / self.samplerate))
            for lo in range(0, len(rows), chunk):
                sample = self._sample_portion(rows[lo:lo + chunk])
                if len(sample):
                    self._add_rows(sample)
            return
        index = 0
        while index < supplied:  # runningstats.py:363-385
            if self.bulk and max(d.shape[1] for d in self.data) <= 8192:
                index += self._add_bulk(hiddens, batch, channels, hw, units,
                                        index)
                if index >= supplied:
                    break
                # level 0 is full and the next step needs a new level or a refill
                # of the random bits: that one goes the per-operation way
            ff = self.firstfree[0]
            available = self.data[0].shape[1] - ff
            if available == 0:
                if not self._make_room():
                    # no further level: the rate has just halved; the rest of this
                    # batch goes through the subsampling path
                    self._add_rows(self._rows(hiddens, units)[index:],
                                   rate_just_halved=True)
                    return
                ff = self.firstfree[0]
                available = self.data[0].shape[1] - ff
            copycount = min(available, supplied - index)
            with torch.cuda.device(self.device):
                hip._check(self.lib.milan_exemplar_sketch_append(
                    hiddens.data_ptr(), batch, channels, hw, hip._ptr(units),
                    self.depth, index, copycount, self.data[0].data_ptr(),
                    self.data[0].shape[1], ff, _stream(self.device)))
            self.firstfree[0] += copycount
            index += copycount

    # -- subsampling regime (rows = samples x units matrices on the device) ----------
    @staticmethod
    def _rows(hiddens: torch.Tensor, units: Optional[torch.Tensor]) -> torch.Tensor:
        """(batch, channels, *spatial) -> (samples, units), row order (image, y, x)
        like compute.py:329-330; a gather + copy, no arithmetic."""
        if units is not None:
            hiddens = hiddens.index_select(1, units.long())
        if hiddens.dim() == 2:
            return hiddens.contiguous()
        channels = hiddens.shape[1]
        return hiddens.reshape(hiddens.shape[0], channels, -1).permute(
            0, 2, 1).reshape(-1, channels).contiguous()

    def _scan_extremes(self, rows: torch.Tensor) -> None:
        with torch.cuda.device(self.device):
            hip._check(self.lib.milan_exemplar_rows_extremes(
                rows.data_ptr(), len(rows), self.depth, self.extremes.data_ptr(),
                _stream(self.device)))

    def _sample_portion(self, rows: torch.Tensor) -> torch.Tensor:
        """runningstats.py:1221-1224.  The coin flips come from torch's global CPU
        generator, like a reference run on CPU tensors, so seeded runs agree."""
        bits = torch.bernoulli(torch.zeros(len(rows), dtype=torch.uint8),
                               self.samplerate)
        return rows[bits.bool().to(rows.device)]

    def _add_rows(self, rows: torch.Tensor, rate_just_halved: bool = False) -> None:
        """`_add_every` for a (samples, units) matrix, one append per free stretch
        of level 0; when the sketch cannot grow, the remainder is thinned out."""
        def thin(rest):
            if self.samplerate >= 0.5:  # first time: the source is very large
                self._scan_extremes(rest)
            return self._sample_portion(rest)

        if rate_just_halved:
            rows = thin(rows)
        index, supplied = 0, len(rows)
        while index < supplied:
            room = self.data[0].shape[1] - self.firstfree[0]
            if room == 0:
                if not self._make_room():
                    rows = thin(rows[index:])
                    index, supplied = 0, len(rows)
                    if supplied == 0:
                        break
                room = self.data[0].shape[1] - self.firstfree[0]
            take = min(room, supplied - index)
            rows = rows.contiguous()
            with torch.cuda.device(self.device):
                hip._check(self.lib.milan_exemplar_sketch_append(
                    rows.data_ptr(), supplied, self.depth, 1, None, self.depth,
                    index, take, self.data[0].data_ptr(), self.data[0].shape[1],
                    self.firstfree[0], _stream(self.device)))
            self.firstfree[0] += take
            index += take

    def _add_bulk(self, hiddens, batch, channels, hw, units, first) -> int:
        """As many of the remaining samples as the state machine can take
        without `_expand()` / new random bits, in one call (the control flow
        depends on sizes only: milan_exemplar_sketch_add plays it forward on
        the host and runs every level's compactions as one launch)."""
        n = len(self.data)
        caps = (ctypes.c_int64 * n)(*[d.shape[1] for d in self.data])
        need = int(self.lib.milan_exemplar_sketch_add_workspace(
            self.depth, batch * hw - first, caps, n))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        ptrs = (ctypes.c_void_p * n)(*[d.data_ptr() for d in self.data])
        ff = (ctypes.c_int64 * n)(*self.firstfree)
        consumed = ctypes.c_int64(0)
        bit = ctypes.c_int64(self.currentbit)
        with torch.cuda.device(self.device):
            hip._check(self.lib.milan_exemplar_sketch_add(
                hiddens.data_ptr(), batch, channels, hw, hip._ptr(units),
                self.depth, first, ctypes.byref(consumed), ptrs, ff, caps, n,
                self.randbits.data_ptr(), len(self.randbits), ctypes.byref(bit),
                self.extremes.data_ptr(), self._ws.data_ptr(),
                self._ws.numel(), _stream(self.device)))
        self.firstfree = list(ff)
        self.currentbit = int(bit.value)
        return int(consumed.value)

    def add(self, incoming: torch.Tensor) -> None:
        """Reference contract: `incoming` (samples, units)."""
        assert incoming.dim() == 2
        self.add_hiddens(incoming)

    def _draw(self, _user=None) -> int:
        """Next random bit, from torch's global generator in blocks of `resolution`
        (so that a seeded run consumes it like the reference does).  The bulk call
        reads the same block through `randbits` / `currentbit`."""
        self.currentbit += 1
        if self.currentbit >= len(self.randbits):
            self.randbits.random_(to=2)
            self.currentbit = 0
        return int(self.randbits[self.currentbit])

    def _make_room(self) -> bool:
        """Level 0 is full: ask the library what the sketch does now
        (`milan_exemplar_sketch_plan_shift`: compactions up the levels, possibly a new
        level 0) and carry the plan out on the level tensors.  False once the sketch
        cannot grow any more (the sample rate halves)."""
        n = len(self.data)
        caps = (ctypes.c_int64 * n)(*[d.shape[1] for d in self.data])
        fill = (ctypes.c_int64 * n)(*self.firstfree)
        ops = (hip.SketchOp * (2 * n + 4))()
        n_ops, n_out = ctypes.c_int(0), ctypes.c_int(0)
        caps_out = (ctypes.c_int64 * (n + 1))()
        fill_out = (ctypes.c_int64 * (n + 1))()
        draw = hip.DRAW_BIT(self._draw)
        hip._check(self.lib.milan_exemplar_sketch_plan_shift(
            self.resolution, self.buffersize, int(self.samplerate >= 1.0), n, caps,
            fill, draw, None, ops, len(ops), ctypes.byref(n_ops), caps_out, fill_out,
            ctypes.byref(n_out)))
        grown = True
        for op in ops[:n_ops.value]:
            if op.kind == hip.SKETCH_COMPACT:
                self._compact(self.data[op.src], op.n, op.offset, self.data[op.dst],
                              op.position, extremes=bool(op.extremes))
            elif op.kind == hip.SKETCH_INSERT:
                self.data.insert(0, torch.zeros(self.depth, op.capacity,
                                                device=self.device))
            elif op.kind == hip.SKETCH_MOVE:
                self.data[op.dst][:, op.position:op.position + op.n] = (
                    self.data[op.src][:, :op.n])
            else:  # SKETCH_HALVE
                self.samplerate *= 0.5
                grown = False
        self.firstfree = list(fill_out[:n_out.value])
        assert [d.shape[1] for d in self.data] == list(caps_out[:n_out.value])
        return grown

    def quantiles(self, quantile: float) -> torch.Tensor:
        """`quantiles(q)` for a scalar q -> (units,) float32 (:557-580)."""
        if self.count == 0:
            return torch.full((self.depth or 0,), float('nan'))
        total = sum(self.firstfree)
        ws = self._workspace(total, True)
        n = len(self.data)
        ptrs = (ctypes.c_void_p * n)(*[d.data_ptr() for d in self.data])
        ff = (ctypes.c_int64 * n)(*self.firstfree)
        caps = (ctypes.c_int64 * n)(*[d.shape[1] for d in self.data])
        out = torch.empty(self.depth, device=self.device)
        q32 = float(torch.tensor(quantile, dtype=torch.float32))
        with torch.cuda.device(self.device):
            hip._check(self.lib.milan_exemplar_sketch_quantile(
                ptrs, ff, caps, n, self.depth, self.extremes.data_ptr(), q32,
                out.data_ptr(), ws.data_ptr(), ws.numel(),
                _stream(self.device)))
        return out

    # -- persistence: netdissect's format (runningstats.py:428-471) ------------------
    def state_dict(self) -> Dict[str, Any]:
        levels = [d[:, :f].t().cpu().numpy()
                  for d, f in zip(self.data or [], self.firstfree)]
        packed = numpy.empty(len(levels) + 1, dtype=object)  # trailing None as upstream
        for i, level in enumerate(levels):
            packed[i] = level
        return dict(constructor=f'{__name__}.RunningQuantile()',
                    resolution=self.resolution, depth=self.depth,
                    buffersize=self.buffersize, samplerate=self.samplerate,
                    data=packed, sizes=[d.shape[1] for d in self.data or []],
                    extremes=self.extremes.cpu().numpy(), size=self.count,
                    batchcount=self.batchcount)

    def set_state_dict(self, dic) -> None:
        self.resolution = int(dic['resolution'])
        self.randbits = torch.ByteTensor(self.resolution)
        self.currentbit = len(self.randbits) - 1
        depth = int(dic['depth'])
        self.buffersize = int(dic['buffersize'])
        samplerate = float(dic['samplerate'])
        device = self.device or hip.require_device('cuda')
        self._lazy_init(depth, device)
        self.samplerate = samplerate
        self.data, self.firstfree = [], []
        for level, size in zip(dic['data'], dic['sizes']):
            if level is None:
                continue
            level = numpy.asarray(level, dtype=numpy.float32)  # (filled, depth)
            buf = torch.zeros(depth, int(size), device=device)
            buf[:, :level.shape[0]] = torch.from_numpy(level).t().to(device)
            self.data.append(buf)
            self.firstfree.append(int(level.shape[0]))
        self.extremes = torch.from_numpy(
            numpy.asarray(dic['extremes'], dtype=numpy.float32)).to(device).contiguous()
        self.count = int(dic['size'])
        self.batchcount = int(dic['batchcount']) if 'batchcount' in dic else 0

    def to_(self, device) -> None:
        pass


# ---------------------------------------------------------------------------
# ImageVisualizer pieces
# ---------------------------------------------------------------------------
def _find(source, predicate):
    """Crawl dataset.transform / .transforms like netdissect does."""
    if source is None:
        return None
    if predicate(source):
        return source
    found = _find(getattr(source, 'transform', None), predicate)
    if found is not None:
        return found
    for t in reversed(getattr(source, 'transforms', None) or []):
        found = _find(t, predicate)
        if found is not None:
            return found
    return None


def byte_renormalization(source=None) -> Tuple[numpy.ndarray, numpy.ndarray]:
    """(mul, add) of `renormalize.renormalizer(source=..., target='byte')`
    (src/deps/netdissect/renormalize.py:55-136): undo the dataset's Normalize
    (anything with `.mean` / `.std`, else images are taken to be in [0, 1])
    and scale to 0..255.  float64 like the reference; cast to float32 at use."""
    normalizer = _find(source, lambda t: hasattr(t, 'mean') and
                       hasattr(t, 'std') and not isinstance(t, torch.Tensor))
    if normalizer is not None:
        oldoffset, oldscale = normalizer.mean, normalizer.std
    else:
        oldoffset, oldscale = [0.0, 0.0, 0.0], [1.0, 1.0, 1.0]
    newscale = numpy.array([1.0 / 255] * 3)
    mul = numpy.array(oldscale, dtype=numpy.float64) / newscale
    add = (numpy.array(oldoffset, dtype=numpy.float64) - 0.0) / newscale
    return mul, add


class _Cells:
    """Output arrays of `individual_masked_images_for_topk`, on the GPU."""

    def __init__(self, n_units, k, size, device):
        self.k, self.size, self.device = k, size, device
        self.images = torch.zeros(n_units, k, 3, size, size, dtype=torch.uint8,
                                  device=device)
        self.masks = torch.zeros(n_units, k, 1, size, size, dtype=torch.uint8,
                                 device=device)
        self.masked = torch.zeros(n_units, k, 3, size, size, dtype=torch.uint8,
                                  device=device)

    def render(self, lib, activations, images, cells, levels, mul, add):
        activations = _as_hiddens(activations)
        images = _as_hiddens(images)
        assert activations.dim() == 4 and images.dim() == 4
        batch, channels, h, w = activations.shape
        cells_dev = torch.tensor(cells, dtype=torch.int32,
                                 device=self.device).reshape(-1, 4)
        mul3 = (ctypes.c_float * 3)(*[float(numpy.float32(v)) for v in mul])
        add3 = (ctypes.c_float * 3)(*[float(numpy.float32(v)) for v in add])
        with torch.cuda.device(self.device):
            hip._check(lib.milan_exemplar_render(
                activations.data_ptr(), batch, channels, h, w,
                images.data_ptr(), images.shape[2], images.shape[3],
                cells_dev.data_ptr(), len(cells_dev), levels.data_ptr(), mul3,
                add3, self.size, self.k, self.images.data_ptr(),
                self.masks.data_ptr(), self.masked.data_ptr(),
                _stream(self.device)))


# ---------------------------------------------------------------------------
# compute / discriminative / generative
# ---------------------------------------------------------------------------
ActivationStats = Tuple[RunningTopK, RunningQuantile]


def _pull_prefix(prefix: str, state) -> Dict[str, Any]:
    head = prefix + '.'
    return {key[len(head):]: state[key] for key in state if key.startswith(head)}


def _load_cache(path: Optional[pathlib.Path], args: Dict[str, Any]):
    """netdissect's `load_cached_state` (tally.py:741-756): the file counts only if
    every recorded argument equals the current one."""
    if path is None or not path.exists():
        return None
    try:
        state = dict(numpy.load(path, allow_pickle=True))
    except Exception:  # noqa: BLE001 -- an unreadable cache is a missing cache
        return None
    for key, value in args.items():
        if key not in state:
            return None
        have = state[key]
        have = have.item() if getattr(have, 'shape', None) == () else have
        if isinstance(have, numpy.ndarray) or isinstance(value, numpy.ndarray):
            if not numpy.array_equal(numpy.asarray(have), numpy.asarray(value)):
                return None
        elif have != value:
            return None
    return state


def _save_cache(path: pathlib.Path, state: Dict[str, Any],
                args: Dict[str, Any]) -> None:
    path.parent.mkdir(exist_ok=True, parents=True)
    with open(path, 'wb') as handle:  # (numpy.savez appends .npz to bare names)
        numpy.savez(handle, **state, **args)


def compute(compute_topk_and_quantile: Callable[..., Any],
            compute_activations: Callable[..., Any],
            dataset: data.Dataset,
            units: Optional[Sequence[int]] = None,
            k: int = 15,
            quantile: float = 0.99,
            output_size: int = 224,
            batch_size: int = 128,
            image_size: Optional[int] = None,
            renormalizer=None,
            num_workers: int = 30,
            results_dir: Optional[PathLike] = None,
            viz_dir: Optional[PathLike] = None,
            tally_cache_file: Optional[PathLike] = None,
            masks_cache_file: Optional[PathLike] = None,
            save_results: bool = True,
            save_viz: bool = True,
            clear_cache_files: bool = False,
            clear_results_dir: bool = False,
            clear_viz_dir: bool = False,
            display_progress: bool = True) -> ActivationStats:
    """Find the top-activating images of each unit and their masks
    (reference compute.py:27-246; same arguments).

    `compute_topk_and_quantile(*batch)` returns either the reference's pair
    (pooled (batch, units), activations (samples, units)) or -- cheaper, no
    permuted copy -- the hidden tensor (batch, units, h, w) itself;
    `compute_activations(*batch)` returns hiddens (batch, units, h, w) or
    (hiddens, images).  `tally_cache_file` / `masks_cache_file` work as upstream
    (compute.py:140-146, tally.py:199-222,741-767): the first pass's statistics and
    the second pass's arrays are written there and a later call with the same
    arguments loads them instead of touching the dataset; `clear_cache_files`
    deletes them first.  The tally file uses netdissect's key layout (`rtk.*`,
    `rq.*`, `sample_size`, `k`, `r`).
    """
    if units is not None and not units:
        raise ValueError('when setting `units`, must provide >= 1 unit')
    if k < 1:
        raise ValueError(f'must have k >= 1, got k={k}')
    if quantile <= 0 or quantile >= 1:
        raise ValueError('must have quantile in range (0, 1), '
                         f'got quantile={quantile}')
    if image_size is None and not hasattr(dataset, 'transform'):
        raise ValueError('dataset has no `transform` property so '
                         'image_size= must be set')
    del display_progress, image_size
    lib = hip.load_library()
    caches = [pathlib.Path(f) if f is not None else None
              for f in (tally_cache_file, masks_cache_file)]
    if clear_cache_files:
        for cache in caches:
            if cache is not None and cache.exists():
                cache.unlink()
    tally_cache, masks_cache = caches

    if results_dir is None:
        import os
        results_dir = pathlib.Path(os.environ.get('MILAN_RESULTS_DIR',
                                                  'results')) / 'exemplars'
    results_dir = pathlib.Path(results_dir)
    viz_dir = pathlib.Path(viz_dir) if viz_dir is not None else \
        results_dir / 'viz'
    for save, clear, directory in ((save_results, clear_results_dir,
                                    results_dir),
                                   (save_viz, clear_viz_dir, viz_dir)):
        if not save:
            continue
        if clear and directory.exists():
            shutil.rmtree(directory)
        directory.mkdir(exist_ok=True, parents=True)

    units_dev = None
    if units is not None:
        units = sorted(units)
        if save_results:
            numpy.save(f'{results_dir}/units.npy', numpy.array(units))

    # ---- pass 1: tally (tally.tally_topk_and_quantile, tally.py:199-222) ----
    topk, rq = RunningTopK(k=k), RunningQuantile(r=4096)
    # the unit list is part of the cache key (a tally of ANOTHER unit list of
    # the same length, or a subset tally reused for an all-units run, must not be adopted);
    # `units=None` is recorded as the empty list
    tally_args = dict(sample_size=None, k=k, r=4096,
                      units=numpy.array(units if units is not None else [],
                                        dtype=numpy.int64))
    cached = _load_cache(tally_cache, tally_args)
    if cached is not None and (
            'rtk.top_data' not in cached or
            (units is not None and
             cached['rtk.top_data'].shape[0] != len(units))):
        cached = None  # not a tally this run can adopt: recompute
    if cached is not None:
        topk.set_state_dict(_pull_prefix('rtk', cached))
        rq.set_state_dict(_pull_prefix('rq', cached))
    loader = data.DataLoader(dataset, batch_size=batch_size,
                             num_workers=num_workers)
    for batch in (loader if cached is None else ()):
        batch = batch if isinstance(batch, (list, tuple)) else [batch]
        outputs = compute_topk_and_quantile(*batch)
        if isinstance(outputs, torch.Tensor):
            pooled, samples = outputs, outputs  # hiddens (b, c, h, w)
        else:
            pooled, samples = outputs
        if units is not None and units_dev is None:
            # the reference indexes `pooled[:, units]`: negative units count from
            # the end, anything else out of range is an IndexError -- the kernels
            # index `units[u]` unchecked, so this is decided here, on the host
            channels = _as_hiddens(pooled).shape[1]
            units_norm = _normalise_units(units, channels)
            units_dev = torch.tensor(units_norm, dtype=torch.int32,
                                     device=pooled.device)
        topk.add_hiddens(pooled, units_dev)
        rq.add_hiddens(samples, units_dev)

    if cached is None and tally_cache is not None:
        state = {f'rtk.{key}': v for key, v in topk.state_dict().items()}
        state.update({f'rq.{key}': v for key, v in rq.state_dict().items()})
        _save_cache(tally_cache, state, tally_args)
    if units is not None and units_dev is None:  # tally came from the cache
        units_norm = None

    if not (save_results or save_viz or masks_cache is not None):
        return topk, rq

    # ---- pass 2: render the top images (imgviz.py / tally.gather_topk) --------
    levels = rq.quantiles(quantile).reshape(-1)
    mul, add = byte_renormalization(renormalizer if renormalizer is not None
                                    else dataset)
    _, ids = topk.result()
    ids_host = ids.cpu()
    n_units = ids_host.shape[0]
    cells = _Cells(n_units, k, output_size, topk.device)
    needed: Dict[int, list] = {}
    for unit in range(n_units):
        for rank, imgnum in enumerate(ids_host[unit].tolist()):
            needed.setdefault(imgnum, []).append((unit, rank))
    order = sorted(needed)
    # (upstream's gather_topk keys its cache on `count=topk.count`, tally.py; the digest
    # of the top ids and of the unit list also notices another dataset / layer / units)
    import hashlib
    digest = hashlib.sha256(ids_host.numpy().tobytes() +
                            repr(None if units is None else list(units)).encode()
                            ).hexdigest()[:16]
    masks_args = dict(k=k, quantile=quantile, output_size=output_size,
                      n_units=n_units, count=topk.count, top_ids=digest)
    rendered = _load_cache(masks_cache, masks_args)
    if rendered is not None and any(
            tuple(rendered[name].shape) != tuple(getattr(cells, name).shape)
            for name in ('images', 'masks', 'masked')):
        rendered = None
    if rendered is not None:
        for name in ('images', 'masks', 'masked'):
            getattr(cells, name).copy_(torch.from_numpy(rendered[name]))
    loader = data.DataLoader(dataset, sampler=order, batch_size=batch_size,
                             num_workers=num_workers)
    seen = 0
    for batch in (loader if rendered is None else ()):
        batch = batch if isinstance(batch, (list, tuple)) else [batch]
        outputs = compute_activations(*batch)
        if isinstance(outputs, torch.Tensor):
            activations, images = outputs, batch[0]
        else:
            activations, images = outputs
        todo = []
        for j in range(len(activations)):
            for unit, rank in needed[order[seen + j]]:
                if units is not None and units_norm is None:
                    units_norm = _normalise_units(units, activations.shape[1])
                channel = unit if units is None else units_norm[unit]
                todo += [j, channel, unit, rank]
        cells.render(lib, activations, images.to(activations.device), todo,
                     levels, mul, add)
        seen += len(activations)

    if rendered is None and masks_cache is not None:
        _save_cache(masks_cache,
                    {name: getattr(cells, name).cpu().numpy()
                     for name in ('images', 'masks', 'masked')}, masks_args)
    if save_results:
        numpy.save(f'{results_dir}/images.npy', cells.images.cpu().numpy())
        numpy.save(f'{results_dir}/masks.npy', cells.masks.cpu().numpy())
        activations, ids = topk.result()
        for metadata, name, fmt in ((activations, 'activations', '%.5e'),
                                    (ids, 'ids', '%i')):
            metadata = metadata.view(n_units, -1).cpu().numpy()
            numpy.savetxt(str(results_dir / f'{name}.csv'), metadata,
                          delimiter=',', fmt=fmt)
    if save_viz:
        try:
            from PIL import Image
        except ImportError as error:  # pragma: no cover
            raise RuntimeError('save_viz=True needs PIL') from error
        masked = cells.masked.permute(0, 1, 3, 4, 2).cpu().numpy()
        for unit in range(n_units):
            unit_dir = viz_dir / f'unit_{unit}'
            unit_dir.mkdir(exist_ok=True, parents=True)
            for rank in range(masked.shape[1]):
                Image.fromarray(masked[unit, rank]).save(
                    unit_dir / f'image_{rank}.png')
    topk.cells = cells  # the rendered arrays, for callers that skip the files
    return topk, rq


def _run_model(model: nn.Module, inputs) -> Any:
    if not isinstance(inputs, (tuple, dict)):
        raise ValueError(f'inputs must be a tuple or dict, got {type(inputs)}')
    return model(**inputs) if isinstance(inputs, dict) else model(*inputs)


class _Retain:
    """Forward hook standing in for nethook.InstrumentedModel.retain_layer."""

    def __init__(self, model: nn.Module, layer: Optional[str]):
        self.output = None
        self.handle = None
        if layer is not None:
            modules = dict(model.named_modules())
            if layer not in modules:
                raise KeyError(f'layer not found: {layer}')
            self.handle = modules[layer].register_forward_hook(
                lambda _m, _i, out: setattr(self, 'output', out))

    def close(self):
        if self.handle is not None:
            self.handle.remove()


def discriminative(model: nn.Module,
                   dataset: data.Dataset,
                   layer: Optional[Union[str, int]] = None,
                   device=None,
                   results_dir: Optional[PathLike] = None,
                   viz_dir: Optional[PathLike] = None,
                   transform_inputs=first,
                   transform_hiddens=identity,
                   **kwargs: Any) -> ActivationStats:
    """Exemplars of a model whose inputs are the images (compute.py:263-353)."""
    device = hip.require_device(device or 'cuda')
    model.to(device)

    def resolve(directory):
        if directory is not None:
            directory = pathlib.Path(directory)
            directory /= str(layer) if layer is not None else 'outputs'
        return directory

    layer = str(layer) if layer is not None else None
    hook = _Retain(model, layer)

    def hiddens(*args):
        inputs = transform_inputs(*map_location(args, device))
        with torch.no_grad():
            outputs = _run_model(model, inputs)
        return transform_hiddens(outputs if layer is None else hook.output)

    try:
        return compute(hiddens, hiddens, dataset,
                       results_dir=resolve(results_dir),
                       viz_dir=resolve(viz_dir), **kwargs)
    finally:
        hook.close()


def generative(model: nn.Module,
               dataset: data.Dataset,
               layer: Union[str, int],
               device=None,
               results_dir: Optional[PathLike] = None,
               viz_dir: Optional[PathLike] = None,
               transform_inputs=identities,
               transform_hiddens=identity,
               transform_outputs=identity,
               **kwargs: Any) -> ActivationStats:
    """Exemplars of a generator: the images are its outputs
    (compute.py:356-437)."""
    device = hip.require_device(device or 'cuda')
    if results_dir is not None:
        results_dir = pathlib.Path(results_dir) / str(layer)
    if viz_dir is not None:
        viz_dir = pathlib.Path(viz_dir) / str(layer)
    model.to(device)
    hook = _Retain(model, str(layer))

    def run(*args):
        inputs = transform_inputs(*map_location(args, device))
        with torch.no_grad():
            images = _run_model(model, inputs)
        return transform_hiddens(hook.output), transform_outputs(images)

    try:
        return compute(lambda *a: run(*a)[0], run, dataset,
                       results_dir=results_dir, viz_dir=viz_dir, **kwargs)
    finally:
        hook.close()
