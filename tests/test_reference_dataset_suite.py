"""The reference's `TopImagesDataset` unit tests
(tests/milannotations/datasets_test.py:64-250, fixtures tests/conftest.py)
re-stated against `milan_amd.datasets`: same directory layout, same
assertions, same error types and messages."""
import shutil

import numpy
import pytest
import torch

from milan_amd import datasets

N_LAYERS, N_UNITS_PER_LAYER, N_TOP_IMAGES_PER_UNIT = 2, 3, 5
N_SAMPLES = N_LAYERS * N_UNITS_PER_LAYER
IMAGE_SIZE = 16
IMAGE_SHAPE = (3, IMAGE_SIZE, IMAGE_SIZE)
MASK_SHAPE = (1, IMAGE_SIZE, IMAGE_SIZE)


def layer(index):
    return f'layer-{index}'


@pytest.fixture
def top_image_tensors():
    return torch.randint(256, (N_LAYERS, N_UNITS_PER_LAYER,
                               N_TOP_IMAGES_PER_UNIT, *IMAGE_SHAPE),
                         dtype=torch.uint8)


@pytest.fixture
def top_image_masks():
    return torch.randint(2, (N_LAYERS, N_UNITS_PER_LAYER,
                             N_TOP_IMAGES_PER_UNIT, *MASK_SHAPE),
                         dtype=torch.uint8)


@pytest.fixture
def top_images_root(tmp_path, top_image_tensors, top_image_masks):
    root = tmp_path / 'root'
    for index in range(N_LAYERS):
        layer_dir = root / layer(index)
        layer_dir.mkdir(parents=True)
        numpy.save(layer_dir / 'images.npy', top_image_tensors[index].numpy())
        numpy.save(layer_dir / 'masks.npy', top_image_masks[index].numpy())
    return root


def transform_images(images):
    assert images.shape == (N_TOP_IMAGES_PER_UNIT, *IMAGE_SHAPE)
    return images


def transform_masks(masks):
    assert masks.shape == (N_TOP_IMAGES_PER_UNIT, *MASK_SHAPE)
    return masks


@pytest.fixture
def top_images_dataset(top_images_root):
    return datasets.TopImagesDataset(top_images_root,
                                     transform_images=transform_images,
                                     transform_masks=transform_masks,
                                     display_progress=False)


def check_sample_ranges(sample):
    assert sample.images.dtype is torch.float
    assert sample.images.min() >= 0 and sample.images.max() <= 1
    assert sample.masks.dtype is torch.float
    assert sample.masks.min() >= 0 and sample.masks.max() <= 1


@pytest.mark.parametrize('device', (None, 'cpu', torch.device('cpu')))
def test_top_images_dataset_init(top_images_root, device):
    dataset = datasets.TopImagesDataset(top_images_root,
                                        display_progress=False, device=device)
    assert dataset.root == top_images_root
    assert str(top_images_root).endswith(dataset.name)
    assert dataset.layers == tuple(layer(i) for i in range(N_LAYERS))
    assert dataset.device is device
    assert len(dataset.samples) == N_SAMPLES
    for sample in dataset.samples:
        check_sample_ranges(sample)


def test_top_images_dataset_init_with_units_file(top_images_root):
    units = range(N_UNITS_PER_LAYER - 1)
    numpy.save(str(top_images_root / layer(0) / 'units.npy'),
               numpy.array(units))
    dataset = datasets.TopImagesDataset(top_images_root,
                                        display_progress=False)
    assert dataset.layers == tuple(layer(i) for i in range(N_LAYERS))
    assert len(dataset.samples) == N_SAMPLES - 1
    for sample in dataset.samples:
        if sample.layer == layer(0):
            assert sample.unit != N_UNITS_PER_LAYER - 1
        check_sample_ranges(sample)


@pytest.mark.parametrize('subpath,error_pattern', (
    ('', '.*root directory not found.*'),
    (f'{layer(0)}/images.npy', '.*missing images.*'),
    (f'{layer(0)}/masks.npy', '.*missing masks.*'),
))
def test_top_images_dataset_init_missing_files(top_images_root, subpath,
                                               error_pattern):
    path = top_images_root / subpath
    if path.is_dir():
        shutil.rmtree(path)
    else:
        assert path.is_file()
        path.unlink()
    with pytest.raises(FileNotFoundError, match=error_pattern):
        datasets.TopImagesDataset(top_images_root)


@pytest.mark.parametrize('images,masks,error_pattern', (
    ((5, 3, 32, 32), None, '.*5D images.*'),
    (None, (5, 1, 32, 32), '.*5D masks.*'),
    ((10, 5, 3, 32, 32), (8, 5, 1, 32, 32), '.*masks/images.*'),
    ((10, 5, 3, 32, 32), (10, 4, 1, 32, 32), '.*masks/images.*'),
    ((10, 5, 3, 31, 32), (10, 5, 1, 32, 32), '.*height/width.*'),
    ((10, 5, 3, 32, 31), (10, 5, 1, 32, 32), '.*height/width.*'),
))
def test_top_images_dataset_init_bad_images_or_masks(top_images_root,
                                                     top_image_tensors,
                                                     top_image_masks, images,
                                                     masks, error_pattern):
    images = (top_image_tensors[0] if images is None else torch.zeros(
        images, dtype=torch.uint8))
    masks = (top_image_masks[0] if masks is None else torch.zeros(
        masks, dtype=torch.uint8))
    for name, tensor in (('images', images), ('masks', masks)):
        numpy.save(top_images_root / layer(0) / f'{name}.npy', tensor.numpy())
    with pytest.raises(ValueError, match=error_pattern):
        datasets.TopImagesDataset(top_images_root)


@pytest.mark.parametrize('shape,error_pattern', (((), '.*0D.*'),
                                                 ((1, 2), '.*2D.*')))
def test_top_images_dataset_init_bad_units(top_images_root, shape,
                                           error_pattern):
    units = torch.randint(N_UNITS_PER_LAYER, size=shape)
    numpy.save(top_images_root / layer(0) / 'units.npy', units.numpy())
    with pytest.raises(ValueError, match=error_pattern):
        datasets.TopImagesDataset(top_images_root)


def test_top_images_dataset_getitem(top_images_root, top_image_tensors,
                                    top_image_masks):
    dataset = datasets.TopImagesDataset(top_images_root,
                                        display_progress=False, device='cpu')
    for li in range(N_LAYERS):
        for unit in range(N_UNITS_PER_LAYER):
            sample = dataset[li * N_UNITS_PER_LAYER + unit]
            assert sample.layer == layer(li)
            assert sample.unit == unit
            assert sample.images.dtype is torch.float
            assert sample.images.allclose(
                top_image_tensors[li][unit].float() / 255, atol=1e-3)
            assert sample.masks.dtype is torch.float
            assert sample.masks.equal(top_image_masks[li][unit].float())


def test_top_images_dataset_len(top_images_dataset):
    assert len(top_images_dataset) == N_SAMPLES


def test_top_images_dataset_lookup(top_images_dataset, top_image_tensors,
                                   top_image_masks):
    for li in range(N_LAYERS):
        for unit in range(N_UNITS_PER_LAYER):
            actual = top_images_dataset.lookup(layer(li), unit)
            assert actual.layer == layer(li)
            assert actual.unit == unit
            assert actual.images.allclose(top_image_tensors[li][unit] / 255,
                                          atol=1e-3)
            assert actual.masks.equal(top_image_masks[li][unit].float())


@pytest.mark.parametrize('layer_name,unit,error_pattern', (
    ('layer-10000', 0, '.*"layer-10000" does not exist.*'),
    ('layer-0', 100000, '.*unit 100000.*'),
))
def test_top_images_dataset_lookup_bad_key(top_images_dataset, layer_name,
                                           unit, error_pattern):
    with pytest.raises(KeyError, match=error_pattern):
        top_images_dataset.lookup(layer_name, unit)


def test_top_images_dataset_k(top_images_dataset):
    assert top_images_dataset.k == N_TOP_IMAGES_PER_UNIT
