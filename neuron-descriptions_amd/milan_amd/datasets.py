"""Input side: exemplar images + masks on disk -> tensors.

On-disk contract (writer `src/exemplars/compute.py:217-218`, reader
`src/milannotations/datasets.py:159-197`): `<root>/<layer>/images.npy` uint8
(units, k, 3, H, W), `masks.npy` uint8 {0,1} (units, k, 1, H, W), optional
`units.npy` int (units,).

The reference inflates the whole dataset to float32 in host RAM (49 GB for 4k
units).  Here the arrays are memory-mapped and stay uint8 until the GPU; the
byte->float conversion (`x * float32(1/255)`) happens inside the encoder's
input kernel.  `__getitem__` still hands out the reference's `TopImages`
tuples (float tensors, images in [0,1]) so foreign code keeps working, and
`slice_uint8` is the zero-copy fast path `Decoder.predict` uses.
"""
import pathlib
from typing import Iterable, NamedTuple, Optional, Tuple, Union

import numpy
import torch
from torch.utils import data


class TopImages(NamedTuple):
    """Top images for a unit (reference datasets.py:20-26)."""
    layer: str
    unit: int
    images: torch.Tensor
    masks: torch.Tensor


class TopImagesDataset(data.Dataset):
    """Top-activating images for individual units, memory-mapped uint8."""

    def __init__(self,
                 root: Union[str, pathlib.Path],
                 name: Optional[str] = None,
                 layers: Optional[Iterable[Union[str, int]]] = None,
                 device: Optional[Union[str, torch.device]] = None,
                 transform_images=None,
                 transform_masks=None,
                 display_progress: bool = True,
                 mmap: bool = True):
        """Same arguments as the reference (datasets.py:96-103) plus `mmap`.
        `device` / `transform_*` are applied when a sample is handed out (the
        arrays themselves stay memory-mapped uint8); `display_progress` is
        accepted for compatibility -- opening memory maps takes no time."""
        self.device = device
        self.transform_images = transform_images
        self.transform_masks = transform_masks
        root = pathlib.Path(root)
        if not root.is_dir():
            raise FileNotFoundError(f'root directory not found: {root}')
        if layers is None:
            layers = [f.name for f in root.iterdir() if f.is_dir()]
        if not layers:
            raise ValueError('no layers given and root has no subdirectories')
        self.root = root
        self.name = name or f'{root.parent.name}/{root.name}'
        self.layers = tuple(sorted(str(layer) for layer in layers))
        self.images_by_layer, self.masks_by_layer, self.units_by_layer = {}, {}, {}
        self._index = []  # (layer, position)
        self._valid = {}  # layer -> samples kept (units file may truncate)
        mode = 'r' if mmap else None
        for layer in self.layers:
            for fname in ('images.npy', 'masks.npy'):
                if not (root / layer / fname).exists():
                    raise FileNotFoundError(f'{layer} is missing {fname}')
            images = numpy.load(root / layer / 'images.npy', mmap_mode=mode)
            masks = numpy.load(root / layer / 'masks.npy', mmap_mode=mode)
            for what, arr in (('images', images), ('masks', masks)):
                if arr.ndim != 5:
                    raise ValueError(f'expected 5D {what}, got {arr.ndim}D '
                                     f'in layer {layer}')
            if images.shape[:2] != masks.shape[:2]:
                raise ValueError(f'layer {layer} masks/images have different '
                                 f'# unit/images: {images.shape[:2]} vs. '
                                 f'{masks.shape[:2]}')
            if images.shape[3:] != masks.shape[3:]:
                raise ValueError(f'layer {layer} masks/images have different '
                                 f'height/width {images.shape[3:]} vs. '
                                 f'{masks.shape[3:]}')
            units_file = root / layer / 'units.npy'
            if units_file.exists():
                units = numpy.load(units_file)
                if units.ndim != 1:
                    raise ValueError(f'expected 1D units, got {units.ndim}D')
            else:
                units = numpy.arange(len(images))
            self.images_by_layer[layer] = images
            self.masks_by_layer[layer] = masks
            self.units_by_layer[layer] = units
            # the reference zips (units, images, masks): a units file shorter
            # than the arrays truncates the layer (datasets.py:201-204)
            self._valid[layer] = min(len(units), len(images))
            self._index += [(layer, i) for i in range(self._valid[layer])]
        shapes = {self.images_by_layer[l].shape[1:] for l in self.layers}
        if len(shapes) != 1:
            raise ValueError(f'layers disagree on (k, 3, H, W): {shapes}')

    def __len__(self) -> int:
        return len(self._index)

    def __getitem__(self, index: int) -> TopImages:
        layer, i = self._index[index]
        images = torch.from_numpy(numpy.array(self.images_by_layer[layer][i]))
        masks = torch.from_numpy(numpy.array(self.masks_by_layer[layer][i]))
        return self._sample(layer, int(self.units_by_layer[layer][i]), images,
                            masks)

    def _sample(self, layer, unit, images, masks, transform=True) -> TopImages:
        # reference datasets.py:191-197: float, images * float32(1/255)
        mul = torch.tensor(1.0 / 255.0, dtype=torch.float64).to(torch.float32)
        images, masks = images.float().mul(mul), masks.float()
        if self.device is not None:
            images, masks = images.to(self.device), masks.to(self.device)
        if transform and self.transform_images is not None:
            images = self.transform_images(images)
        if transform and self.transform_masks is not None:
            masks = self.transform_masks(masks)
        return TopImages(layer=layer, unit=unit, images=images, masks=masks)

    @property
    def samples(self):
        """Lazy view with the reference's `dataset.samples` list interface."""
        return _Samples(self)

    @property
    def k(self) -> int:
        """The "k" in "top-k images"."""
        assert len(self) > 0, 'empty dataset?'
        layer, _ = self._index[0]
        return int(self.images_by_layer[layer].shape[1])

    def unit(self, index: int) -> Tuple[str, int]:
        layer, i = self._index[index]
        return layer, int(self.units_by_layer[layer][i])

    def units(self, indices) -> Tuple[Tuple[str, int], ...]:
        return tuple(self.unit(index) for index in indices)

    def slice_uint8(self, lo: int, hi: int,
                    out: Optional[Tuple[Optional[torch.Tensor],
                                        Optional[torch.Tensor]]] = None
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Samples [lo, hi) as uint8 (n,k,3,H,W) / (n,k,1,H,W) CPU tensors
        (same samples, in the same order, as `self[lo] .. self[hi - 1]`).

        `out` = (images, masks) destination tensors with room for hi - lo
        samples (e.g. pinned staging buffers; either may be None): the rows are
        copied from the memory maps straight into them -- one pass over the
        bytes -- and the returned tensors are views of their first n rows."""
        if not 0 <= lo <= hi <= len(self):
            raise IndexError(f'slice [{lo}, {hi}) outside dataset of '
                             f'{len(self)} samples')
        n = hi - lo
        runs = []
        pos = lo
        while pos < hi:
            layer, i = self._index[pos]
            run = min(hi - pos, self._valid[layer] - i)
            runs.append((layer, i, run))
            pos += run

        def gather(by_layer, dst):
            if dst is None:
                first = by_layer[runs[0][0] if runs else self.layers[0]]
                dst = torch.empty((n,) + tuple(first.shape[1:]),
                                  dtype=torch.uint8)
            dst = dst[:n]
            view = dst.numpy()
            at = 0
            for layer, i, run in runs:
                numpy.copyto(view[at:at + run], by_layer[layer][i:i + run])
                at += run
            return dst

        images_out, masks_out = out if out is not None else (None, None)
        return (gather(self.images_by_layer, images_out),
                gather(self.masks_by_layer, masks_out))

    def lookup(self, layer: Union[str, int], unit: int) -> TopImages:
        layer = str(layer)
        if layer not in self.images_by_layer:
            raise KeyError(f'layer "{layer}" does not exist')
        if unit >= len(self.images_by_layer[layer]):
            raise KeyError(f'layer "{layer}" has no unit {unit}')
        # positional, untransformed, like the reference (datasets.py:252-259)
        return self._sample(
            layer, unit,
            torch.from_numpy(numpy.array(self.images_by_layer[layer][unit])),
            torch.from_numpy(numpy.array(self.masks_by_layer[layer][unit])),
            transform=False)


class _Samples:
    """Sequence view over a TopImagesDataset (materialises one sample at a
    time instead of the reference's list of float tensors)."""

    def __init__(self, dataset: TopImagesDataset):
        self.dataset = dataset

    def __len__(self) -> int:
        return len(self.dataset)

    def __getitem__(self, index):
        if isinstance(index, slice):
            return [self.dataset[i] for i in range(*index.indices(len(self)))]
        return self.dataset[index]

    def __iter__(self):
        return (self.dataset[i] for i in range(len(self)))


def load(key: str, path: Optional[Union[str, pathlib.Path]] = None,
         **kwargs) -> TopImagesDataset:
    """`milannotations.load(key, path=...)` for exemplar directories
    (reference src/milannotations/loaders.py:245-259, local-path branch)."""
    import os
    root = pathlib.Path(path) if path is not None else pathlib.Path(
        os.environ.get('MILAN_DATA_DIR', 'data')) / key
    if not root.exists():
        raise KeyError(f'unknown milannotations set: {key}')
    return TopImagesDataset(root, name=key, **kwargs)


class ShardView(data.Dataset):
    """Samples [lo, hi) of a `TopImagesDataset`, as one rank of a sharded job sees
    them (scripts/compute_milan_descriptions.py).  Unlike `data.Subset` it keeps what
    `Decoder.predict` looks at to choose its ingest path: the dataset-side
    `transform_images` / `transform_masks` / `device` attributes, and
    `slice_uint8(lo, hi, out=)` -- the one-pass copy from the memory maps into the
    prefetcher's pinned staging buffers."""

    def __init__(self, dataset: TopImagesDataset, lo: int, hi: int):
        if not 0 <= lo <= hi <= len(dataset):
            raise IndexError(f'shard [{lo}, {hi}) outside dataset of '
                             f'{len(dataset)} samples')
        self.dataset, self.lo, self.hi = dataset, lo, hi

    # what predict()'s fast-path guard reads
    transform_images = property(lambda self: self.dataset.transform_images)
    transform_masks = property(lambda self: self.dataset.transform_masks)
    device = property(lambda self: self.dataset.device)

    def __len__(self) -> int:
        return self.hi - self.lo

    def __getitem__(self, index: int):
        if not -len(self) <= index < len(self):
            raise IndexError(f'sample index out of bounds: {index}')
        return self.dataset[self.lo + index % len(self)]

    def unit(self, index: int):
        return self.dataset.unit(self.lo + index)

    def slice_uint8(self, lo: int, hi: int, out=None):
        if not 0 <= lo <= hi <= len(self):
            raise IndexError(f'slice [{lo}, {hi}) outside shard of '
                             f'{len(self)} samples')
        return self.dataset.slice_uint8(self.lo + lo, self.lo + hi, out=out)
