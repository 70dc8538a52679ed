"""Seeded synthetic weights and exemplar workloads for MILAN.

There is no network in the build/test/bench environments, so neither
`milan-base.pth` (reference `src/milan/loaders.py:9-25`) nor the torchvision
ImageNet ResNet-101 weights (`src/milan/encoders.py:274`) can be fetched.
Everything measured or parity-checked in this repo therefore runs on weights
produced here: same tensor names and shapes as the reference state dict
(SURVEY.md appendix), values drawn from a seeded CPU `torch.Generator` so
that this container (where goldens are generated) and the GPU box (where they
are checked) build bit-identical tensors.

Deviation from SURVEY.md §8(d), deliberate: BN running statistics are NOT
calibrated on data (a data-dependent calibration runs a CPU conv whose
reduction order depends on the host's thread count, so two machines would get
slightly different "weights").  Instead every BN gets seeded, non-trivial
(gamma, beta, mean, var), convs are Kaiming-normal (fan-in), and the last BN
of each bottleneck carries a small gamma so the residual stream stays O(1)
through all 33 blocks.  The purpose (meaningful fp tolerances) is the same.
"""
from typing import Dict, Optional, Sequence, Tuple

import torch

RESNET_BLOCKS = {
    'resnet18': (2, 2, 2, 2),
    'resnet34': (3, 4, 6, 3),
    'resnet50': (3, 4, 6, 3),
    'resnet101': (3, 4, 23, 3),
    'resnet152': (3, 8, 36, 3),
}
BASIC_BLOCK_CONFIGS = ('resnet18', 'resnet34')  # torchvision BasicBlock trunks
# trunk kinds of milan_dims.trunk_kind (include/milan_hip.h)
TRUNK_BOTTLENECK, TRUNK_BASIC, TRUNK_ALEXNET = 0, 1, 2
ALEXNET_CHANNELS = (1, 3, 6, 4, 4)  # x width (64): 64, 192, 384, 256, 256


def trunk_kind(config: str) -> int:
    if config == 'alexnet':
        return TRUNK_ALEXNET
    return TRUNK_BASIC if config in BASIC_BLOCK_CONFIGS else TRUNK_BOTTLENECK


def pyramid_feature_size(config: str, width: int = 64) -> int:
    """Sum of the five tap widths (encoders.py:330-350: 1152 / 1024 / 3904)."""
    kind = trunk_kind(config)
    if kind == TRUNK_ALEXNET:
        return sum(ALEXNET_CHANNELS) * width
    return (16 if kind == TRUNK_BASIC else 61) * width
PYRAMID_FEATURES = 64 + 256 + 512 + 1024 + 2048  # encoders.py:346-350


def _gen(seed: int) -> torch.Generator:
    g = torch.Generator(device='cpu')
    g.manual_seed(seed)
    return g


def _conv(g: torch.Generator, cout: int, cin: int, k: int) -> torch.Tensor:
    std = (2.0 / (cin * k * k))**0.5
    return torch.randn(cout, cin, k, k, generator=g) * std


def _bn(g: torch.Generator,
        sd: Dict[str, torch.Tensor],
        prefix: str,
        c: int,
        gamma: float = 1.0,
        var: float = 1.0) -> None:
    sd[prefix + '.weight'] = gamma * (0.8 + 0.4 * torch.rand(c, generator=g))
    sd[prefix + '.bias'] = 0.1 * torch.randn(c, generator=g)
    sd[prefix + '.running_mean'] = 0.1 * torch.randn(c, generator=g)
    sd[prefix + '.running_var'] = var * (0.8 + 0.4 * torch.rand(c, generator=g))
    sd[prefix + '.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)


def resnet_state_dict(config: str = 'resnet101',
                      seed: int = 0,
                      prefix: str = '',
                      width: int = 64,
                      with_fc: bool = True) -> Dict[str, torch.Tensor]:
    """Synthetic torchvision-style ResNet (bottleneck) state dict.

    Key names follow torchvision 0.12 `resnet.py` (the factory the reference
    calls at `src/milan/encoders.py:346-349`).  `width` < 64 gives a slim
    trunk of the same topology for cheap tests.
    """
    blocks = RESNET_BLOCKS[config]
    g = _gen(seed)
    sd: Dict[str, torch.Tensor] = {}
    sd[prefix + 'conv1.weight'] = _conv(g, width, 3, 7)
    _bn(g, sd, prefix + 'bn1', width, var=3.5)
    inplanes = width
    if config in BASIC_BLOCK_CONFIGS:
        # BasicBlock: 3x3 (carries the stride) -> 3x3, expansion 1; downsample
        # only where the shape changes (not in layer1)
        for li, nblocks in enumerate(blocks):
            planes = width * (2**li)
            for bi in range(nblocks):
                p = f'{prefix}layer{li + 1}.{bi}.'
                sd[p + 'conv1.weight'] = _conv(g, planes, inplanes, 3)
                _bn(g, sd, p + 'bn1', planes)
                sd[p + 'conv2.weight'] = _conv(g, planes, planes, 3)
                _bn(g, sd, p + 'bn2', planes, gamma=0.25)
                if bi == 0 and li > 0:
                    sd[p + 'downsample.0.weight'] = _conv(g, planes, inplanes, 1)
                    _bn(g, sd, p + 'downsample.1', planes)
                inplanes = planes
        blocks = ()
    for li, nblocks in enumerate(blocks):
        planes = width * (2**li)
        for bi in range(nblocks):
            p = f'{prefix}layer{li + 1}.{bi}.'
            sd[p + 'conv1.weight'] = _conv(g, planes, inplanes, 1)
            _bn(g, sd, p + 'bn1', planes)
            sd[p + 'conv2.weight'] = _conv(g, planes, planes, 3)
            _bn(g, sd, p + 'bn2', planes)
            sd[p + 'conv3.weight'] = _conv(g, planes * 4, planes, 1)
            _bn(g, sd, p + 'bn3', planes * 4, gamma=0.25)
            if bi == 0:
                sd[p + 'downsample.0.weight'] = _conv(g, planes * 4, inplanes,
                                                      1)
                _bn(g, sd, p + 'downsample.1', planes * 4)
            inplanes = planes * 4
    if with_fc:
        # Present in the reference checkpoint; computed and discarded by the
        # reference, never read by this build.
        sd[prefix + 'fc.weight'] = torch.randn(1000, inplanes,
                                               generator=g) * 0.01
        sd[prefix + 'fc.bias'] = torch.zeros(1000)
    return sd


def alexnet_state_dict(seed: int = 0,
                       prefix: str = '',
                       width: int = 64,
                       with_classifier: bool = True
                       ) -> Dict[str, torch.Tensor]:
    """Synthetic torchvision-style AlexNet state dict (`features.N.*` keys,
    the factory behind the reference's 'alexnet' config, encoders.py:330-335).
    Convs carry biases; there is no BatchNorm."""
    g = _gen(seed)
    sd: Dict[str, torch.Tensor] = {}
    ch = [c * width for c in ALEXNET_CHANNELS]
    spec = ((0, 3, ch[0], 11), (3, ch[0], ch[1], 5), (6, ch[1], ch[2], 3),
            (8, ch[2], ch[3], 3), (10, ch[3], ch[4], 3))
    for idx, cin, cout, k in spec:
        sd[f'{prefix}features.{idx}.weight'] = _conv(g, cout, cin, k)
        sd[f'{prefix}features.{idx}.bias'] = 0.1 * torch.randn(cout, generator=g)
    if with_classifier:  # computed and discarded by the reference
        for idx, cin, cout in ((1, ch[4] * 36, 64), (4, 64, 64), (6, 64, 1000)):
            sd[f'{prefix}classifier.{idx}.weight'] = torch.randn(
                cout, cin, generator=g) * 0.01
            sd[f'{prefix}classifier.{idx}.bias'] = torch.zeros(cout)
    return sd


def _linear(g: torch.Generator, sd: Dict[str, torch.Tensor], name: str,
            out_f: int, in_f: int, scale: float = 1.0) -> None:
    bound = scale / in_f**0.5
    sd[name + '.weight'] = (torch.rand(out_f, in_f, generator=g) * 2 -
                            1) * bound
    sd[name + '.bias'] = (torch.rand(out_f, generator=g) * 2 - 1) * bound


def decoder_state_dict(vocab_size: int,
                       feature_size: int = PYRAMID_FEATURES,
                       hidden_size: int = 512,
                       embedding_size: int = 128,
                       attention_hidden_size: Optional[int] = None,
                       lm: bool = True,
                       lm_hidden_size: int = 512,
                       lm_embedding_size: int = 128,
                       lm_layers: int = 2,
                       seed: int = 0,
                       logit_scale: float = 4.0) -> Dict[str, torch.Tensor]:
    """Synthetic decoder (+LM) parameters with the reference's key names.

    Shapes: `src/milan/decoders.py:304-323`, `src/milan/lms.py:47-56`.
    Default torch inits (uniform +-1/sqrt(fan_in)) except the two output
    layers, scaled by `logit_scale` so the token distributions are peaked
    like a trained captioner's instead of near-uniform (near-uniform logits
    make every argmax a near-tie, which says nothing about parity).
    """
    g = _gen(seed + 1000)
    a = attention_hidden_size or min(hidden_size, feature_size)
    v = vocab_size
    sd: Dict[str, torch.Tensor] = {}
    _linear(g, sd, 'init_h.0', hidden_size, feature_size)
    _linear(g, sd, 'init_c.0', hidden_size, feature_size)
    sd['embedding.weight'] = torch.randn(v, embedding_size, generator=g)
    _linear(g, sd, 'attend.query_to_hidden', a, hidden_size)
    _linear(g, sd, 'attend.key_to_hidden', a, feature_size)
    _linear(g, sd, 'attend.output.0', 1, a, scale=4.0)
    _linear(g, sd, 'feature_gate.0', feature_size, hidden_size)
    bound = 1.0 / hidden_size**0.5
    in_f = embedding_size + feature_size
    for name, shape in (('lstm.weight_ih', (4 * hidden_size, in_f)),
                        ('lstm.weight_hh', (4 * hidden_size, hidden_size)),
                        ('lstm.bias_ih', (4 * hidden_size,)),
                        ('lstm.bias_hh', (4 * hidden_size,))):
        sd[name] = (torch.rand(*shape, generator=g) * 2 - 1) * bound
    _linear(g, sd, 'output.1', v, hidden_size, scale=logit_scale)
    if lm:
        sd['lm.embedding.weight'] = torch.randn(v,
                                                lm_embedding_size,
                                                generator=g)
        sd['lm.embedding.weight'][v - 2] = 0  # padding_idx row, lms.py:47-49
        bound = 1.0 / lm_hidden_size**0.5
        for layer in range(lm_layers):
            in_l = lm_embedding_size if layer == 0 else lm_hidden_size
            for name, shape in ((f'weight_ih_l{layer}', (4 * lm_hidden_size,
                                                         in_l)),
                                (f'weight_hh_l{layer}',
                                 (4 * lm_hidden_size, lm_hidden_size)),
                                (f'bias_ih_l{layer}', (4 * lm_hidden_size,)),
                                (f'bias_hh_l{layer}', (4 * lm_hidden_size,))):
                sd['lm.lstm.' + name] = (torch.rand(*shape, generator=g) * 2 -
                                         1) * bound
        _linear(g, sd, 'lm.output.0', v, lm_hidden_size, scale=logit_scale)
    return sd


def milan_state_dict(vocab_size: int,
                     config: str = 'resnet101',
                     seed: int = 0,
                     width: int = 64,
                     **decoder_kwargs) -> Dict[str, torch.Tensor]:
    """Full synthetic `Decoder.state_dict()` (SURVEY.md a17 layout)."""
    feature_size = pyramid_feature_size(config, width)
    sd = decoder_state_dict(vocab_size,
                            feature_size=feature_size,
                            seed=seed,
                            **decoder_kwargs)
    if config == 'alexnet':
        sd.update(alexnet_state_dict(seed=seed, width=width,
                                     prefix='encoder.encoder.model.'))
    else:
        sd.update(
            resnet_state_dict(config,
                              seed=seed,
                              prefix='encoder.encoder.model.',
                              width=width))
    # encoders.py:282-284 buffers.
    sd['encoder.mean'] = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    sd['encoder.std'] = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return sd


def vocab_tokens(n: int) -> Tuple[str, ...]:
    """A deterministic vocabulary of `n` tokens with some punctuation."""
    base = ['.', ',', '-', ';', ':', 'the', 'of', 'and', 'dog', 'fur']
    out = list(base[:n])
    i = 0
    while len(out) < n:
        out.append(f'w{i}')
        i += 1
    return tuple(out)


def exemplars(n_neurons: int,
              k: int = 15,
              size: int = 224,
              seed: int = 1,
              device: str = 'cpu',
              zero_every: int = 97) -> Tuple[torch.Tensor, torch.Tensor]:
    """Synthetic uint8 exemplar images and {0,1} masks.

    Images i.i.d. U{0..255}; one axis-aligned rectangle mask per image with
    side lengths U{size/14..size*4/7} (16..128 at 224); every `zero_every`-th
    image gets an all-zero mask (exercises the zero-mask rule,
    `src/milan/encoders.py:311-314`).
    """
    gi = torch.Generator(device=device)
    gi.manual_seed(seed)
    images = torch.randint(0,
                           256, (n_neurons, k, 3, size, size),
                           dtype=torch.uint8,
                           generator=gi,
                           device=device)
    gm = _gen(seed + 1)
    m = n_neurons * k
    lo, hi = max(1, size // 14), max(2, size * 4 // 7)
    hh = torch.randint(lo, hi + 1, (m,), generator=gm)
    ww = torch.randint(lo, hi + 1, (m,), generator=gm)
    y0 = (torch.rand(m, generator=gm) * (size - hh + 1)).long()
    x0 = (torch.rand(m, generator=gm) * (size - ww + 1)).long()
    # rectangle parameters come from the CPU generator (same on every box);
    # the rasterisation happens on the target device, in bounded slabs.
    hh, ww, y0, x0 = (t.to(device).view(-1, 1, 1) for t in (hh, ww, y0, x0))
    ys = torch.arange(size, device=device).view(1, size, 1)
    xs = torch.arange(size, device=device).view(1, 1, size)
    masks = torch.empty(m, size, size, dtype=torch.uint8, device=device)
    slab = 4096
    for lo in range(0, m, slab):
        sl = slice(lo, lo + slab)
        masks[sl] = ((ys >= y0[sl]) & (ys < y0[sl] + hh[sl]) & (xs >= x0[sl]) &
                     (xs < x0[sl] + ww[sl])).to(torch.uint8)
    if zero_every:
        masks[zero_every - 1::zero_every] = 0
    return images, masks.view(n_neurons, k, 1, size, size)


def describe(sd: Dict[str, torch.Tensor]) -> Sequence[str]:
    """Human-readable `name shape` lines (used by docs/tests)."""
    return [f'{k} {tuple(v.shape)}' for k, v in sd.items()]


# ---------------------------------------------------------------------------
# exemplar computation (src/exemplars/compute.py): tiny dissected models
# ---------------------------------------------------------------------------
def exemplar_model(n_units: int = 3,
                   n_layers: int = 2,
                   seed: int = 0,
                   kernel_size: int = 4,
                   padding: int = 2,
                   relu: bool = False) -> 'torch.nn.Sequential':
    """The reference's test model (`tests/exemplars/compute_test.py:47-63`):
    conv_1 (3 -> units) and conv_i (units -> units), kernel 4, padding 2 --
    with weights drawn from a seeded generator so that every box rebuilds the
    same model.  `relu=True` appends a ReLU to every layer's output (ties at
    zero, as in real post-ReLU feature maps)."""
    import collections

    from torch import nn
    g = _gen(seed)
    layers = []
    for index in range(1, n_layers + 1):
        cin = 3 if index == 1 else n_units
        conv = nn.Conv2d(cin, n_units, kernel_size, padding=padding)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) *
                              (1.0 / (cin * kernel_size**2))**0.5)
            conv.bias.copy_(torch.randn(conv.bias.shape, generator=g) * 0.1)
        layers.append((f'conv_{index}',
                       nn.Sequential(conv, nn.ReLU()) if relu else conv))
    return nn.Sequential(collections.OrderedDict(layers)).eval()


def exemplar_images(n_images: int, size: int = 16, seed: int = 0) -> torch.Tensor:
    """`tests/conftest.py:28-31`: a TensorDataset of rand(n, 3, size, size)."""
    return torch.rand(n_images, 3, size, size, generator=_gen(seed))
