"""Parameter containers with the reference's state-dict names, no compute.

The reference modules are `nn.Linear` / `nn.LSTMCell` / torchvision ResNet
instances whose `forward` does the arithmetic.  Here the arithmetic lives in
libmilan_hip, so the Python side only needs objects that OWN tensors under the
same dotted names (so `state_dict()`, `load_state_dict()`, `.to(device)` and
checkpoints stay interchangeable with the reference).  `ParamTree` builds a
nested `nn.Module` tree from a {dotted name: shape} spec; it deliberately has
no `forward`.
"""
from typing import Dict, Iterable, Mapping, Sequence, Tuple

import torch
from torch import nn

BUFFER_SUFFIXES = ('running_mean', 'running_var', 'num_batches_tracked')


class ParamTree(nn.Module):
    """A node holding parameters / buffers and child nodes only."""

    def forward(self, *args, **kwargs):  # pragma: no cover
        raise RuntimeError(
            'ParamTree holds weights only; computation happens in '
            'libmilan_hip through milan_amd.Decoder / Encoder / LanguageModel')


def build(spec: Mapping[str, Tuple[Sequence[int], torch.dtype]],
          root: nn.Module = None,
          buffers: Iterable[str] = ()) -> nn.Module:
    """Create nested ParamTrees under `root` for every dotted name in spec."""
    root = root if root is not None else ParamTree()
    buffers = set(buffers)
    for name, (shape, dtype) in spec.items():
        node = root
        *path, leaf = name.split('.')
        for part in path:
            child = getattr(node, part, None) if part in node._modules else None
            if child is None:
                child = ParamTree()
                node.add_module(part, child)
            node = child
        tensor = torch.zeros(tuple(shape), dtype=dtype)
        if name in buffers or leaf in BUFFER_SUFFIXES:
            node.register_buffer(leaf, tensor)
        else:
            node.register_parameter(leaf,
                                    nn.Parameter(tensor, requires_grad=False))
    return root


def alexnet_spec(width: int = 64,
                 prefix: str = '') -> Dict[str, Tuple[Tuple[int, ...],
                                                      torch.dtype]]:
    """torchvision-0.12 AlexNet state-dict names and shapes (`features.N`,
    `classifier.N`; the classifier is in the checkpoint but never read)."""
    f = torch.float32
    c = [m * width for m in (1, 3, 6, 4, 4)]
    spec: Dict[str, Tuple[Tuple[int, ...], torch.dtype]] = {}
    for idx, cin, cout, k in ((0, 3, c[0], 11), (3, c[0], c[1], 5),
                              (6, c[1], c[2], 3), (8, c[2], c[3], 3),
                              (10, c[3], c[4], 3)):
        spec[f'{prefix}features.{idx}.weight'] = ((cout, cin, k, k), f)
        spec[f'{prefix}features.{idx}.bias'] = ((cout,), f)
    hidden = 64 * width
    for idx, cin, cout in ((1, c[4] * 36, hidden), (4, hidden, hidden),
                           (6, hidden, 1000)):
        spec[f'{prefix}classifier.{idx}.weight'] = ((cout, cin), f)
        spec[f'{prefix}classifier.{idx}.bias'] = ((cout,), f)
    return spec


def resnet_spec(blocks: Sequence[int],
                width: int = 64,
                prefix: str = '',
                basic: bool = False) -> Dict[str, Tuple[Tuple[int, ...],
                                                        torch.dtype]]:
    """torchvision-0.12 ResNet state-dict names and shapes (bottleneck blocks,
    or BasicBlock for resnet18/34)."""
    f, i64 = torch.float32, torch.int64
    spec: Dict[str, Tuple[Tuple[int, ...], torch.dtype]] = {}

    def bn(p, c):
        spec[p + '.weight'] = ((c,), f)
        spec[p + '.bias'] = ((c,), f)
        spec[p + '.running_mean'] = ((c,), f)
        spec[p + '.running_var'] = ((c,), f)
        spec[p + '.num_batches_tracked'] = ((), i64)

    spec[prefix + 'conv1.weight'] = ((width, 3, 7, 7), f)
    bn(prefix + 'bn1', width)
    inplanes = width
    if basic:
        for li, n in enumerate(blocks):
            planes = width * 2**li
            for bi in range(n):
                p = f'{prefix}layer{li + 1}.{bi}.'
                spec[p + 'conv1.weight'] = ((planes, inplanes, 3, 3), f)
                bn(p + 'bn1', planes)
                spec[p + 'conv2.weight'] = ((planes, planes, 3, 3), f)
                bn(p + 'bn2', planes)
                if bi == 0 and li > 0:
                    spec[p + 'downsample.0.weight'] = ((planes, inplanes, 1, 1),
                                                       f)
                    bn(p + 'downsample.1', planes)
                inplanes = planes
        blocks = ()
    for li, n in enumerate(blocks):
        planes = width * 2**li
        for bi in range(n):
            p = f'{prefix}layer{li + 1}.{bi}.'
            spec[p + 'conv1.weight'] = ((planes, inplanes, 1, 1), f)
            bn(p + 'bn1', planes)
            spec[p + 'conv2.weight'] = ((planes, planes, 3, 3), f)
            bn(p + 'bn2', planes)
            spec[p + 'conv3.weight'] = ((planes * 4, planes, 1, 1), f)
            bn(p + 'bn3', planes * 4)
            if bi == 0:
                spec[p + 'downsample.0.weight'] = ((planes * 4, inplanes, 1,
                                                    1), f)
                bn(p + 'downsample.1', planes * 4)
            inplanes = planes * 4
    spec[prefix + 'fc.weight'] = ((1000, inplanes), f)
    spec[prefix + 'fc.bias'] = ((1000,), f)
    return spec
