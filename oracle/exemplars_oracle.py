"""CPU oracle for the exemplar computation.  TEST
INFRASTRUCTURE ONLY -- same rules as `milan_oracle.py`: imported by `tests/`
only, never by the product.

A plain torch-CPU / numpy restatement of what
`src/exemplars/compute.py:27-246` does with its netdissect dependencies
(vendored in the reference tree):

  * `RunningTopK` (src/deps/netdissect/runningstats.py:31-151): the k largest
    pooled activations per unit and the dataset indices of their images;
  * `RunningQuantile` (runningstats.py:274-627): the KLL sketch, including its
    randomised regime -- the random bits are drawn from torch's global
    generator exactly as the reference draws them;
  * `ImageVisualizer` (src/deps/netdissect/imgviz.py:185-210, upsample.py:
    6-45,132-156): activation map -> bilinear `grid_sample` (align_corners)
    to the output size -> `> level` mask; image -> byte renormalisation ->
    nearest resize; the masked visualisation of ext/netdissect/imgviz.py:
    70-76 (thickness 0, outside_bright .25).

Pinned by tests/golden/make_golden_exemplars.py (the imported reference run on
the tiny models of its own test), see tests/test_exemplar_goldens.py.
"""
import math
from typing import Callable, List, Optional, Sequence, Tuple

import numpy
import torch
from torch.utils import data


# ---------------------------------------------------------------------------
# RunningTopK
# ---------------------------------------------------------------------------
class TopK:
    """Exact global top-k per unit.  (The reference keeps a 5k-wide buffer
    that it compresses with `topk` when full -- mathematically the same set;
    ties are ordered value desc, then lower dataset index.)"""

    def __init__(self, k: int):
        self.k, self.count = k, 0
        self.values: Optional[torch.Tensor] = None  # (units, <=k)
        self.index: Optional[torch.Tensor] = None

    def add(self, pooled: torch.Tensor) -> None:
        size = pooled.shape[0]
        vals = pooled.t().contiguous()
        idx = (torch.arange(size) + self.count)[None].expand_as(vals)
        if self.values is not None:
            vals = torch.cat([self.values, vals], 1)
            idx = torch.cat([self.index, idx], 1)
        # value descending, dataset index ascending: stable sort of the
        # index-ordered concatenation by descending value
        order = torch.argsort(idx, dim=1, stable=True)
        vals, idx = vals.gather(1, order), idx.gather(1, order)
        order = torch.argsort(vals, dim=1, descending=True, stable=True)
        keep = min(self.k, vals.shape[1])
        self.values = vals.gather(1, order)[:, :keep]
        self.index = idx.gather(1, order)[:, :keep]
        self.count += size

    def result(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.values, self.index


# ---------------------------------------------------------------------------
# RunningQuantile (KLL sketch)
# ---------------------------------------------------------------------------
class QuantileSketch:
    """runningstats.py:274-627, state-for-state.  `r` as passed by
    `tally_topk_and_quantile` (tally.py:199: r=4096)."""

    def __init__(self, r: int = 4096):
        self.resolution = r * 2
        self.buffersize = min(128, (self.resolution + 7) // 8)
        self.samplerate = 1.0
        self.depth = None
        self.data: List[torch.Tensor] = []
        self.firstfree = [0]
        self.randbits = torch.ByteTensor(self.resolution)
        self.currentbit = len(self.randbits) - 1
        self.extremes = None
        self.count = 0

    def add(self, incoming: torch.Tensor) -> None:
        if self.depth is None:
            self.depth = incoming.shape[1]
            self.data = [torch.zeros(self.depth, self.resolution)]
            self.extremes = torch.zeros(self.depth, 2)
            self.extremes[:, 0] = float('inf')
            self.extremes[:, 1] = -float('inf')
        assert incoming.shape[1] == self.depth
        self.count += incoming.shape[0]
        if self.samplerate >= 1.0:
            self._add_every(incoming)
            return
        # subsampling regime (runningstats.py:359-367): every item still counts for
        # the extremes; a Bernoulli(samplerate) portion of each chunk enters the sketch
        self._scan_extremes(incoming)
        chunksize = int(math.ceil(self.buffersize / self.samplerate))
        for index in range(0, len(incoming), chunksize):
            sample = self._sample_portion(incoming[index:index + chunksize])
            if len(sample):
                self._add_every(sample)

    def _scan_extremes(self, incoming: torch.Tensor) -> None:  # :409-413
        self._update_extremes(incoming.min(dim=0)[0], incoming.max(dim=0)[0])

    def _sample_portion(self, vec: torch.Tensor) -> torch.Tensor:  # :1221-1224
        bits = torch.bernoulli(torch.zeros(vec.shape[0], dtype=torch.uint8),
                               self.samplerate)  # torch's GLOBAL (CPU) generator
        return vec[bits.bool()]

    def _add_every(self, incoming: torch.Tensor) -> None:
        supplied, index = len(incoming), 0
        while index < supplied:
            ff = self.firstfree[0]
            available = self.data[0].shape[1] - ff
            if available == 0:
                if not self._shift():
                    # no room for another level: the rate halved (:376-383)
                    incoming = incoming[index:]
                    if self.samplerate >= 0.5:
                        self._scan_extremes(incoming)
                    incoming = self._sample_portion(incoming)
                    index, supplied = 0, len(incoming)
                ff = self.firstfree[0]
                available = self.data[0].shape[1] - ff
            copycount = min(available, supplied - index)
            self.data[0][:, ff:ff + copycount] = incoming[index:index +
                                                          copycount].t()
            self.firstfree[0] += copycount
            index += copycount

    def _randbit(self) -> int:
        self.currentbit += 1
        if self.currentbit >= len(self.randbits):
            self.randbits.random_(to=2)  # torch's GLOBAL generator
            self.currentbit = 0
        return int(self.randbits[self.currentbit])

    def _update_extremes(self, minr, maxr) -> None:
        self.extremes[:, 0] = torch.minimum(self.extremes[:, 0], minr)
        self.extremes[:, 1] = torch.maximum(self.extremes[:, 1], maxr)

    def _shift(self) -> bool:
        index = 0
        while self.data[index].shape[1] - self.firstfree[index] < (
                -(-self.data[index - 1].shape[1] // 2) if index else 1):
            if index + 1 >= len(self.data):
                return self._expand()
            d = self.data[index][:, 0:self.firstfree[index]].sort()[0]
            if index == 0 and self.samplerate >= 1.0:
                self._update_extremes(d[:, 0], d[:, -1])
            offset = self._randbit()
            position = self.firstfree[index + 1]
            subset = d[:, offset::2]
            self.data[index + 1][:, position:position +
                                 subset.shape[1]] = subset
            self.firstfree[index] = 0
            self.firstfree[index + 1] += subset.shape[1]
            index += 1
        return True

    def _next_capacity(self) -> int:
        cap = int(math.ceil(self.resolution * (0.67**len(self.data))))
        if cap < 2:
            return 0
        cap = -8 * (-cap // 8)
        return max(self.buffersize, cap)

    def _expand(self) -> bool:
        cap = self._next_capacity()
        if cap > 0:
            self.data.insert(0, torch.zeros(self.depth, cap))
            self.firstfree.insert(0, 0)
        else:
            assert self.firstfree[0] == 0
            self.samplerate *= 0.5
        for index in range(1, len(self.data)):
            amount = self.firstfree[index]
            if amount == 0:
                continue
            position = self.firstfree[index - 1]
            if self.data[index - 1].shape[1] - (amount + position) >= (
                    -(-self.data[index - 2].shape[1] // 2) if
                (index - 1) else 1):
                self.data[index - 1][:, position:position + amount] = (
                    self.data[index][:, :amount])
                self.firstfree[index - 1] += amount
                self.firstfree[index] = 0
            else:
                d = self.data[index][:, :amount].sort()[0]
                if index == 1:
                    self._update_extremes(d[:, 0], d[:, -1])
                offset = self._randbit()
                scrunched = d[:, offset::2]
                self.data[index][:, :scrunched.shape[1]] = scrunched
                self.firstfree[index] = scrunched.shape[1]
        return cap > 0

    def quantiles(self, q: float, stable: bool = False) -> torch.Tensor:
        """`quantiles(q)` for one scalar q -> (depth,) float32.

        `stable=False` is the reference's call, `torch.sort(summary, dim=-1)`:
        NOT a stable sort (torch's CPU sort reorders equal keys once a row has
        more than ~16 elements; its CUDA sort orders them yet another way), so
        when equal samples sit on different levels -- post-ReLU zeros -- the
        order of their WEIGHTS, and with it the interpolated quantile at the
        edge of the run, is unspecified upstream.  `stable=True` keeps equal
        samples in level order, which is what the HIP kernel defines."""
        if self.firstfree[0]:
            d0 = self.data[0][:, :self.firstfree[0]]
            self._update_extremes(d0.min(dim=1)[0], d0.max(dim=1)[0])
        size = sum(self.firstfree)
        weights = torch.zeros(size)
        summary = torch.zeros(self.depth, size)
        index = 0
        for level, ff in enumerate(self.firstfree):
            if ff == 0:
                continue
            summary[:, index:index + ff] = self.data[level][:, :ff]
            weights[index:index + ff] = 2.0**level
            index += ff
        summary, order = torch.sort(summary, dim=-1, stable=stable)
        weights = weights[order.view(-1)].view(order.shape)
        summary = torch.cat(
            [self.extremes[:, :1], summary, self.extremes[:, 1:]], dim=-1)
        weights = torch.cat([
            torch.zeros(weights.shape[0], 1), weights,
            torch.zeros(weights.shape[0], 1)
        ], dim=-1)
        cumweights = torch.cumsum(weights, dim=-1) - weights / 2
        cumweights /= torch.sum(weights, dim=-1, keepdim=True)
        nq = torch.tensor(q).view(-1).numpy()  # float32, as the reference
        result = torch.zeros(self.depth)
        ncw, nsm = cumweights.numpy(), summary.numpy()
        for d in range(self.depth):
            result[d] = torch.tensor(numpy.interp(nq, ncw[d], nsm[d]),
                                     dtype=torch.float32)[0]
        return result


# ---------------------------------------------------------------------------
# ImageVisualizer: mask and image rendering
# ---------------------------------------------------------------------------
def upsample_coords(data_size: int, target_size: int) -> torch.Tensor:
    """One axis of `upsample_grid` (upsample.py:132-156) with no
    scale_offset: normalised source coordinates in [-1, 1] (float32)."""
    scale = float(target_size) / data_size
    offset = 0.5 * scale - 0.5
    return ((torch.arange(target_size, dtype=torch.float) - offset) *
            (2 / (scale * max(1, (data_size - 1)))) - 1)


def bilinear_upsample(a: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """`grid_sample(a[None,None], grid, 'bilinear', 'zeros',
    align_corners=True)[0,0]` as torch's vectorised CPU kernel evaluates it:
    x = (g + 1) * ((W - 1) / 2); weights from floor; out-of-range corners
    contribute zero; out = nw*v_nw + ne*v_ne + sw*v_sw + se*v_se."""
    h, w = a.shape
    gy, gx = upsample_coords(h, size[0]), upsample_coords(w, size[1])
    y = (gy + 1) * torch.tensor((h - 1) / 2, dtype=torch.float)
    x = (gx + 1) * torch.tensor((w - 1) / 2, dtype=torch.float)
    y, x = y[:, None].expand(size), x[None, :].expand(size)
    x_w, y_n = x.floor(), y.floor()
    west, north = x - x_w, y - y_n
    east, south = 1 - west, 1 - north
    out = torch.zeros(size)
    for dy, dx, wt in ((0, 0, south * east), (0, 1, south * west),
                       (1, 0, north * east), (1, 1, north * west)):
        yi, xi = (y_n + dy).long(), (x_w + dx).long()
        ok = (yi >= 0) & (yi < h) & (xi >= 0) & (xi < w)
        val = a[yi.clamp(0, h - 1), xi.clamp(0, w - 1)] * ok
        out = out + val * wt
    return out


def activation_mask(a: torch.Tensor, level: float,
                    size: Tuple[int, int]) -> torch.Tensor:
    """`ImageVisualizer.pytorch_mask` (imgviz.py:185-198)."""
    return bilinear_upsample(a, size) > level


def byte_image(image: torch.Tensor, size: Tuple[int, int],
               mul: Sequence[float] = (255.0, 255.0, 255.0),
               add: Sequence[float] = (0.0, 0.0, 0.0)) -> torch.Tensor:
    """`ImageVisualizer.pytorch_image` (imgviz.py:200-210): renormalise to
    bytes (`data.mul(mul).add_(add).clamp(0,255).byte()`, renormalize.py:
    119-136), then nearest resize; float (0..255) like the reference returns."""
    m = torch.tensor(mul, dtype=torch.float)[:, None, None]
    b = torch.tensor(add, dtype=torch.float)[:, None, None]
    byte = image.mul(m).add_(b).clamp(0, 255).byte()
    return torch.nn.functional.interpolate(byte.float()[None], size=size)[0]


def render(acts: torch.Tensor, image: torch.Tensor, unit: int, level: float,
           size: Tuple[int, int], mul=(255.0,) * 3, add=(0.0,) * 3
           ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """One (unit, rank) cell of ext/netdissect/imgviz.py:56-81 -> uint8
    masked (3,S,S), image (3,S,S), mask (1,S,S)."""
    mask = activation_mask(acts[unit], level, size)
    scaled = byte_image(image, size, mul, add)
    inside, outside = mask.float(), (~mask).float()
    masked = (scaled * inside + 0.25 * scaled * outside).clamp(0, 255).byte()
    return (masked, scaled.clamp(0, 255).byte(),
            mask[None].float().clamp(0, 255).byte())


# ---------------------------------------------------------------------------
# compute()
# ---------------------------------------------------------------------------
def compute(compute_topk_and_quantile: Callable,
            compute_activations: Callable,
            dataset: data.Dataset,
            units: Optional[Sequence[int]] = None,
            k: int = 15,
            quantile: float = 0.99,
            output_size: int = 224,
            batch_size: int = 128,
            mul=(255.0,) * 3, add=(0.0,) * 3,
            stable_ties: bool = False):
    """`compute` (src/exemplars/compute.py:27-246) minus file output.
    Returns dict(images, masks, masked, ids, activations, levels).
    `stable_ties`: see QuantileSketch.quantiles."""
    if units is not None:
        units = sorted(units)
    topk, sketch = TopK(k), QuantileSketch()
    # tally.tally_topk_and_quantile: one DataLoader over the dataset (its
    # iterator draws a base seed from the global RNG, like the reference's)
    for batch in data.DataLoader(dataset, batch_size=batch_size):
        pooled, activations = compute_topk_and_quantile(*batch)
        if units is not None:
            pooled, activations = pooled[:, units], activations[:, units]
        topk.add(pooled)
        sketch.add(activations)
    levels = sketch.quantiles(quantile, stable=stable_ties)
    acts_top, ids = topk.result()
    n_units = ids.shape[0]
    size = (output_size, output_size)
    images = torch.zeros(n_units, k, 3, *size, dtype=torch.uint8)
    masks = torch.zeros(n_units, k, 1, *size, dtype=torch.uint8)
    masked = torch.zeros(n_units, k, 3, *size, dtype=torch.uint8)
    needed = {}
    for unit in range(n_units):
        for rank, imgnum in enumerate(ids[unit].tolist()):
            needed.setdefault(imgnum, []).append((unit, rank))
    order = sorted(needed)
    loader = data.DataLoader(dataset, sampler=order, batch_size=batch_size)
    seen = 0
    for batch in loader:
        outputs = compute_activations(*batch)
        if isinstance(outputs, tuple):
            activations, batch_images = outputs
        else:
            activations, batch_images = outputs, batch[0]
        if units is not None:
            activations = activations[:, units]
        for j in range(len(activations)):
            for unit, rank in needed[order[seen + j]]:
                m, im, mk = render(activations[j], batch_images[j], unit,
                                   float(levels[unit]), size, mul, add)
                masked[unit, rank], images[unit, rank] = m, im
                masks[unit, rank] = mk
        seen += len(activations)
    return dict(images=images, masks=masks, masked=masked, ids=ids,
                activations=acts_top, levels=levels)


def discriminative(model, dataset, layer: Optional[str] = None, **kwargs):
    """`discriminative` (compute.py:263-353) for an nn.Sequential whose
    `layer` is a direct child (what the goldens use): hiddens = that child's
    output; pooled = spatial max; activations = (B*h*w, C)."""

    def hiddens(images):
        x = images
        with torch.no_grad():
            for name, child in model.named_children():
                x = child(x)
                if layer is not None and name == layer:
                    return x
        return x

    def topk_and_quantile(images):
        h = hiddens(images)
        b, c = h.shape[:2]
        return (h.view(b, c, -1).max(dim=2)[0],
                h.permute(0, 2, 3, 1).reshape(-1, c))

    return compute(topk_and_quantile, hiddens, dataset, **kwargs)


def generative(model, dataset, layer: str, **kwargs):
    """`generative` (compute.py:356-437): the images are the model's outputs."""

    def run(inputs):
        x, hid = inputs, None
        with torch.no_grad():
            for name, child in model.named_children():
                x = child(x)
                if name == layer:
                    hid = x
        return hid, x

    def topk_and_quantile(inputs):
        h, _ = run(inputs)
        b, c = h.shape[:2]
        return (h.view(b, c, -1).max(dim=2)[0],
                h.permute(0, 2, 3, 1).reshape(-1, c))

    return compute(topk_and_quantile, run, dataset, **kwargs)
