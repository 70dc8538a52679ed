"""The reference's own encoder unit tests (tests/milan/encoders_test.py,
fixtures from tests/milan/conftest.py and tests/conftest.py), re-stated
against `milan_amd.encoders`: same configs ('resnet18', 'alexnet'), same
batch (10 x 3 x 224 x 224, random {0,1} masks), same assertions."""
import numpy
import pytest
import torch

from milan_amd import datasets, encoders

N_LAYERS, N_UNITS_PER_LAYER, N_TOP_IMAGES_PER_UNIT = 2, 3, 5
IMAGE_SIZE_SMALL = 16
FEATURE_SHAPE = (10, 10)

BATCH_SIZE = 10
IMAGE_SIZE = 224
IMAGE_SHAPE = (3, IMAGE_SIZE, IMAGE_SIZE)
MASK_SHAPE = (1, IMAGE_SIZE, IMAGE_SIZE)


class FakeEncoder(encoders.Encoder):
    """A fake Encoder that always returns zeros (tests/milan/conftest.py)."""

    def __init__(self, feature_shape):
        super().__init__()
        self.feature_shape = feature_shape

    def forward(self, images, masks, **kwargs):
        assert not kwargs
        assert images.shape[0] == masks.shape[0]
        assert images.shape[2:] == masks.shape[2:]
        assert images.shape[1] == 3
        assert masks.shape[1] == 1
        return torch.zeros(len(images), *self.feature_shape)


@pytest.fixture
def top_images_dataset(tmp_path):
    g = torch.Generator().manual_seed(0)
    for layer in range(N_LAYERS):
        d = tmp_path / 'root' / f'layer-{layer}'
        d.mkdir(parents=True)
        shape = (N_UNITS_PER_LAYER, N_TOP_IMAGES_PER_UNIT)
        numpy.save(d / 'images.npy', torch.randint(
            256, (*shape, 3, IMAGE_SIZE_SMALL, IMAGE_SIZE_SMALL),
            dtype=torch.uint8, generator=g).numpy())
        numpy.save(d / 'masks.npy', torch.randint(
            2, (*shape, 1, IMAGE_SIZE_SMALL, IMAGE_SIZE_SMALL),
            dtype=torch.uint8, generator=g).numpy())
    return datasets.TopImagesDataset(tmp_path / 'root')


@pytest.mark.parametrize('device', (None, 'cpu', torch.device('cpu')))
def test_encoder_map(top_images_dataset, device):
    """Encoder.map returns a TensorDataset of the right size."""
    encoder = FakeEncoder(FEATURE_SHAPE)
    actual = encoder.map(top_images_dataset, image_index=-2, mask_index=-1,
                         display_progress_as=None, device=device)
    assert len(actual) == len(top_images_dataset) == N_LAYERS * N_UNITS_PER_LAYER
    for (features,) in actual:
        assert features.shape == (N_TOP_IMAGES_PER_UNIT, *FEATURE_SHAPE)
        assert features.eq(0).all()


def test_pyramid_conv_encoder_init_bad_config():
    bad = 'bad-config'
    with pytest.raises(ValueError, match=f'.*{bad}.*'):
        encoders.PyramidConvEncoder(config=bad)


def random_batch(seed=0):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(BATCH_SIZE, *IMAGE_SHAPE, generator=g)
    masks = torch.randint(2, size=(BATCH_SIZE, *MASK_SHAPE), generator=g,
                          dtype=torch.int64).float()
    return images, masks


# The three forward tests of the reference (valid masks / the last two masks
# zeroed / every mask zeroed) differ only in the masks and in which feature
# rows must be exactly zero.
MASK_CASES = {
    'valid': (lambda m: m, slice(0, 0)),
    'some_invalid': (lambda m: torch.cat([m[:-2], torch.zeros_like(m[-2:])]),
                     slice(BATCH_SIZE - 2, BATCH_SIZE)),
    'all_invalid': (torch.zeros_like, slice(0, BATCH_SIZE)),
}


@pytest.mark.gpu
@pytest.mark.parametrize('case', sorted(MASK_CASES))
@pytest.mark.parametrize('config', ('resnet18', 'alexnet'))
def test_pyramid_conv_encoder_forward(config, case):
    edit, zero_rows = MASK_CASES[case]
    images, masks = random_batch()
    encoder = encoders.PyramidConvEncoder(config=config,
                                          pretrained=False).to('cuda')
    actual = encoder(images, edit(masks))
    assert actual.shape == (BATCH_SIZE, *encoder.feature_shape)
    assert not torch.isnan(actual).any()
    expect_zero = torch.zeros(BATCH_SIZE, dtype=torch.bool)
    expect_zero[zero_rows] = True
    row_is_zero = actual.eq(0).all(dim=1).cpu()
    assert torch.equal(row_is_zero, expect_zero)
