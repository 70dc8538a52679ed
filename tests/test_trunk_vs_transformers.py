"""The oracle's ResNet trunk (a7) against an independent third-party
implementation that IS installed here: HuggingFace transformers'
`ResNetModel`, whose released checkpoints are converted from torchvision /
timm weights, i.e. a second statement of the torchvision v1.5 topology
(stride on the 3x3 conv, 1x1-stride shortcut + BN on the first block of every
stage, ReLU after the add, 7x7/2 stem + 3x3/2 max-pool, BN eps 1e-5).

torchvision==0.12.0 itself is not installable offline, so the trunk stays
"unpinned upstream"; this test removes the possibility that the oracle and
the golden generator's stand-in (`tests/golden/make_golden.py::_TVResNet`,
written by the same hand) share a topology mistake."""
import pytest
import torch
import torch.nn.functional as F

from milan_amd import synthetic
from oracle import milan_oracle as O

transformers = pytest.importorskip('transformers')


def hf_resnet(config, width, sd, prefix):
    blocks = synthetic.RESNET_BLOCKS[config]
    basic = config in ('resnet18', 'resnet34')
    expansion = 1 if basic else 4
    cfg = transformers.ResNetConfig(
        num_channels=3, embedding_size=width,
        hidden_sizes=[expansion * width * 2**i for i in range(4)],
        depths=list(blocks),
        layer_type='basic' if basic else 'bottleneck', hidden_act='relu',
        downsample_in_first_stage=False, downsample_in_bottleneck=False)
    model = transformers.ResNetModel(cfg).eval()
    mapped = {}
    for key, value in sd.items():
        if not key.startswith(prefix):
            continue
        k = key[len(prefix):]
        parts = k.split('.')
        if parts[0] == 'conv1':
            new = 'embedder.embedder.convolution.' + parts[1]
        elif parts[0] == 'bn1':
            new = 'embedder.embedder.normalization.' + parts[1]
        elif parts[0].startswith('layer'):
            stage, block = int(parts[0][5:]) - 1, int(parts[1])
            base = f'encoder.stages.{stage}.layers.{block}.'
            if parts[2] == 'downsample':
                kind = 'convolution' if parts[3] == '0' else 'normalization'
                new = base + f'shortcut.{kind}.' + parts[4]
            else:
                idx = int(parts[2][-1]) - 1
                kind = 'convolution' if parts[2].startswith('conv') else \
                    'normalization'
                new = base + f'layer.{idx}.{kind}.' + parts[3]
        else:
            continue  # fc (computed and discarded by the reference)
        mapped[new] = value
    missing, unexpected = model.load_state_dict(mapped, strict=False)
    assert not unexpected, unexpected
    assert all(m.endswith('num_batches_tracked') for m in missing), missing
    return model


@pytest.mark.parametrize('config,width,size', [('resnet50', 16, 64),
                                               ('resnet101', 8, 96),
                                               ('resnet50', 64, 64),
                                               ('resnet18', 16, 64)])
def test_oracle_trunk_equals_transformers_resnet(config, width, size):
    prefix = 'encoder.encoder.model.'
    sd = synthetic.resnet_state_dict(config, seed=21, width=width,
                                     prefix=prefix)
    model = hf_resnet(config, width, sd, prefix)
    x = torch.randn(2, 3, size, size, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        out = model(pixel_values=x, output_hidden_states=True)
        taps = O.resnet_trunk(x, sd, prefix=prefix,
                              blocks=synthetic.RESNET_BLOCKS[config])
    hidden = out.hidden_states  # embedder output, then the four stages
    assert len(hidden) == 5
    # nethook's 'conv1' tap is the RAW stem conv; transformers exposes it only
    # after bn1 + relu + maxpool
    stem = F.max_pool2d(F.relu(O._bn(taps[0], sd, prefix + 'bn1')), 3, 2, 1)
    torch.testing.assert_close(stem, hidden[0], rtol=1e-5, atol=1e-5)
    for level in range(1, 5):
        assert taps[level].shape == hidden[level].shape
        torch.testing.assert_close(taps[level], hidden[level], rtol=1e-4,
                                   atol=1e-5)
