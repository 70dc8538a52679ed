"""Visual encoders: the reference's `src/milan/encoders.py` surface on HIP.

`Encoder` keeps the reference's abstract contract (feature_shape, forward(images,
masks=None) -> (M, *feature_shape), map(dataset) -> TensorDataset;
encoders.py:23-148) so any subclass -- including a caller's own, like the
reference's test FakeEncoder -- still plugs into `Decoder`.
`PyramidConvEncoder` (encoders.py:243-351) is the one pretrained MILAN uses;
its arithmetic runs in libmilan_hip (`milan_encode`).  All four configs of the
reference's table (encoders.py:326-351) are built: 'alexnet', 'resnet18',
'resnet50', 'resnet101' (plus the same-shaped resnet34 / resnet152); unknown
names raise ValueError like the reference (encoders.py:265-267).
`SpatialConvEncoder` (encoders.py:158-234, config 'resnet18') runs on the same
trunk kernels through `milan_encode_spatial`.
"""
from typing import Any, Mapping, Optional, Tuple, Type, Union

import torch
from torch import nn
from torch.utils import data

from milan_amd import hip, params, synthetic

IMAGENET_MEAN = (0.485, 0.456, 0.406)  # src/deps/netdissect/renormalize.py:87
IMAGENET_STD = (0.229, 0.224, 0.225)


class Encoder(nn.Module):
    """Abstract module mapping images (and optionally masks) to features."""

    feature_shape: Tuple[int, ...]

    def forward(self,
                images: torch.Tensor,
                masks: Optional[torch.Tensor] = None,
                **kwargs: Any) -> torch.Tensor:
        raise NotImplementedError

    def properties(self) -> Mapping[str, Any]:
        raise NotImplementedError

    def map(self,
            dataset: data.Dataset,
            mask: bool = True,
            image_index: Union[int, str] = -3,
            mask_index: Union[int, str] = -2,
            batch_size: int = 64,
            num_workers: int = 0,
            device: Optional[Union[str, torch.device]] = None,
            display_progress_as: Union[bool, str] = True,
            **kwargs: Any) -> data.TensorDataset:
        """Featurize an entire dataset (reference encoders.py:61-148)."""
        if device is not None:
            self.to(device)
        mapped = []
        loader = data.DataLoader(dataset,
                                 batch_size=batch_size,
                                 num_workers=num_workers)
        if isinstance(display_progress_as, str) or display_progress_as:
            try:
                from tqdm.auto import tqdm
                desc = (display_progress_as if isinstance(
                    display_progress_as, str) else 'featurize dataset')
                loader = tqdm(loader, desc=desc)
            except ImportError:
                pass
        for batch in loader:
            images = batch[image_index]
            if not isinstance(images, torch.Tensor):
                raise ValueError(f'non-tensor images: {type(images).__name__}')
            inputs = [images.view(-1, *images.shape[-3:])]
            if mask:
                masks = batch[mask_index]
                if not isinstance(masks, torch.Tensor):
                    raise ValueError(
                        f'non-tensor masks: {type(masks).__name__}')
                inputs.append(masks.view(-1, *masks.shape[-3:]))
            with torch.no_grad():
                features = self(*inputs, **kwargs)
            features = features.view(*images.shape[:-3], *self.feature_shape)
            mapped.append(features)
        return data.TensorDataset(torch.cat(mapped))


class _HipTrunkEncoder(Encoder):
    """Shared plumbing of the encoders whose trunk runs in libmilan_hip:
    owns `encoder.model.*` (torchvision key names) + `mean` / `std`, builds a
    HIP context lazily and rebuilds it when device or weights change."""

    def _init_trunk(self, config: str, kwargs: Mapping[str, Any]) -> None:
        self.config = config
        self.kwargs = dict(kwargs)
        self.kwargs.setdefault('pretrained', True)
        self.width = int(self.kwargs.get('width', 64))
        self.encoder = params.ParamTree()
        if config == 'alexnet':
            spec = params.alexnet_spec(self.width, 'model.')
        else:
            spec = params.resnet_spec(
                self.blocks, self.width, 'model.',
                basic=config in synthetic.BASIC_BLOCK_CONFIGS)
        params.build(spec, root=self.encoder)
        if not self.kwargs['pretrained']:
            self._random_init()
        self.register_buffer('mean',
                             torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1))
        self.register_buffer('std', torch.tensor(IMAGENET_STD).view(1, 3, 1, 1))
        self._ctx: Optional[hip.Context] = None
        self._ctx_key = None

    def _random_init(self) -> None:
        """`pretrained=False`: what the torchvision factories leave behind --
        ResNets: Kaiming-normal (fan_out) convs, BatchNorm weight 1 / bias 0 /
        mean 0 / var 1; AlexNet: PyTorch's default Conv2d/Linear init
        (uniform +-1/sqrt(fan_in) for weights and biases).  The random stream
        is torch's global generator, not torchvision's exact draw order.
        (`pretrained=True`, the reference default, downloads ImageNet weights;
        there is no network here: tensors stay zero until a checkpoint is
        loaded.)"""
        resnet = self.config != 'alexnet'
        with torch.no_grad():
            for name, t in list(self.encoder.named_parameters()) + list(
                    self.encoder.named_buffers()):
                leaf = name.rsplit('.', 1)[-1]
                if t.dim() >= 2:  # conv / linear weight
                    fan_in = t[0].numel()
                    if resnet and t.dim() == 4:
                        fan_out = t.shape[0] * t.shape[2] * t.shape[3]
                        t.normal_(0.0, (2.0 / fan_out)**0.5)
                    else:
                        t.uniform_(-fan_in**-0.5, fan_in**-0.5)
                elif leaf == 'running_var' or (leaf == 'weight' and resnet):
                    t.fill_(1)  # BatchNorm scale / variance
                elif leaf == 'bias' and not resnet:
                    weight = dict(self.encoder.named_parameters())[
                        name[:-len('bias')] + 'weight']
                    bound = weight[0].numel()**-0.5
                    t.uniform_(-bound, bound)
                # everything else (BN bias / mean, counters) stays zero

    # -- HIP context (encoder used stand-alone, e.g. Encoder.map) -------------
    def _context(self) -> hip.Context:
        device = hip.require_device(self.mean.device)
        key = (device, tuple(p._version for p in self.parameters()))
        if self._ctx is None or self._ctx_key != key:
            sd = {'encoder.' + k: v for k, v in self.state_dict().items()}
            dims = hip.make_dims(sd, 1, blocks=self.blocks)
            self._ctx = hip.Context(dims, sd, device)
            self._ctx_key = key
        return self._ctx

    def properties(self) -> Mapping[str, Any]:
        return {'config': self.config, **self.kwargs}


class SpatialConvEncoder(_HipTrunkEncoder):
    """Spatial conv features of the masked image (reference
    encoders.py:158-230): normalise, multiply by the mask, ResNet-18, and hand
    out layer4 position-major -- (M, 49, 512) for 224x224 inputs."""

    def __init__(self, config: str = 'resnet18', **kwargs: Any):
        super().__init__()
        configs = SpatialConvEncoder.configs()
        if config not in configs:
            raise ValueError(f'encoder not supported: {config}')
        self.blocks, (self.layer,), n_features, feature_size = configs[config]
        self._init_trunk(config, kwargs)
        self.feature_shape = (n_features, feature_size * self.width // 64)

    def forward(self,
                images: torch.Tensor,
                masks: Optional[torch.Tensor] = None,
                normalize: bool = True,
                **_: Any) -> torch.Tensor:
        if not normalize:
            raise ValueError('normalize=False is not supported by the HIP '
                             'encoder (normalisation is fused into the input '
                             'conversion kernel)')
        return self._context().encode_spatial(images, masks)

    def map(self, *args: Any, **kwargs: Any) -> data.TensorDataset:
        """`Encoder.map` with single-image defaults (reference :214-223)."""
        kwargs.setdefault('mask', False)
        kwargs.setdefault('image_index', 0)
        return super().map(*args, **kwargs)

    @staticmethod
    def configs():
        """name -> (blocks, layers, n_features, feature_size); reference
        encoders.py:230-234 (49 positions hold for 224x224 inputs)."""
        return {'resnet18': (synthetic.RESNET_BLOCKS['resnet18'], ('layer4',),
                             49, 512)}


class PyramidConvEncoder(_HipTrunkEncoder):
    """Masked multi-resolution ResNet features (reference encoders.py:243).

    Owns `encoder.model.*` (torchvision key names), `mean`, `std`.  Extra
    keyword arguments (`pretrained=`, ...) are kept for serialisation only:
    there is no torchvision download here, weights come from the checkpoint.
    """

    def __init__(self, config: str = 'resnet50', **kwargs: Any):
        super().__init__()
        configs = PyramidConvEncoder.configs()
        if config not in configs:
            raise ValueError(f'encoder not supported: {config}')
        self.blocks, self.layers = configs[config]
        self._init_trunk(config, kwargs)
        self.feature_shape = (synthetic.pyramid_feature_size(config,
                                                             self.width),)

    def forward(self,
                images: torch.Tensor,
                masks: Optional[torch.Tensor] = None,
                normalize: bool = True,
                **_: Any) -> torch.Tensor:
        """Construct pyramid features: (M,3,H,W) [+ (M,1,H,W)] -> (M, F).

        Accepts the reference's float tensors (images in [0,1]) or raw uint8
        exemplars (converted on the GPU exactly like the reference's loader).
        """
        if not normalize:
            raise ValueError('normalize=False is not supported by the HIP '
                             'encoder (normalisation is fused into the input '
                             'conversion kernel)')
        return self._context().encode(images, masks)

    @staticmethod
    def configs():
        """name -> (blocks per stage, retained layer names); reference
        encoders.py:326-351."""
        layers = ('conv1', 'layer1', 'layer2', 'layer3', 'layer4')
        table = {
            name: (blocks, layers)
            for name, blocks in synthetic.RESNET_BLOCKS.items()
        }
        table['alexnet'] = ((0, 0, 0, 0),
                            ('features.0', 'features.3', 'features.6',
                             'features.8', 'features.10'))
        return table


def parse(key: str) -> Type[Encoder]:
    """Parse the string key into an encoder type (reference :354-359)."""
    return {Type.__name__: Type
            for Type in (SpatialConvEncoder, PyramidConvEncoder)}[key]


def key(encoder: Encoder) -> str:
    return type(encoder).__name__


KIND_SPATIAL = 'spatial'
KIND_PYRAMID = 'pyramid'


def encoder(kind: str = KIND_PYRAMID, **kwargs: Any) -> Encoder:
    """Create an encoder: 'pyramid', 'spatial' or an exact type name
    (reference :371-391)."""
    if kind == KIND_SPATIAL:
        return SpatialConvEncoder(**kwargs)
    if kind == KIND_PYRAMID:
        return PyramidConvEncoder(**kwargs)
    return parse(kind)(**kwargs)
