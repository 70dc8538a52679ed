"""`milan.pretrained` (reference `src/milan/loaders.py:28-32`).

The reference resolves `<MILAN_MODELS_DIR>/<name>.pth` through its model hub
(`src/utils/hubs.py:142-170`, `src/utils/env.py:7-9`) and downloads the file
when missing.  There is no network on the build/GPU boxes, so a missing file
raises `FileNotFoundError` (the reference's own error when it has no URL,
hubs.py:113-114) instead of downloading.
"""
import os
import pathlib
from typing import Any, Optional

from milan_amd import decoders

ENV_MODELS_DIR = 'MILAN_MODELS_DIR'
GROUPS = ('base', 'classifiers', 'generators', 'imagenet', 'places365',
          'alexnet', 'resnet152', 'biggan')


def models_dir() -> pathlib.Path:
    return pathlib.Path(os.environ.get(ENV_MODELS_DIR, 'models'))


def resolve(config: str = 'base',
            path: Optional[os.PathLike] = None) -> pathlib.Path:
    """Checkpoint file for a hub key (reference hubs.py:142-170), no download."""
    if config.endswith('+clip'):
        raise KeyError(f'no such model in hub: {config} (DecoderWithCLIP needs '
                       'the un-vendored CLIP package; out of scope)')
    if config not in GROUPS:
        raise KeyError(f'no such model in hub: {config}')
    if path is None:
        name = config.replace('/', '_')
        candidates = [models_dir() / f'{name}.pth',
                      models_dir() / f'milan-{name}.pth']
        path = next((c for c in candidates if c.exists()), candidates[0])
    path = pathlib.Path(path)
    if not path.exists():
        raise FileNotFoundError(
            f'model path not found: {path} (no network here: place the '
            f'reference checkpoint milan-{config}.pth there, or set '
            f'{ENV_MODELS_DIR})')
    return path


def pretrained(config: str = 'base',
               path: Optional[os.PathLike] = None,
               **kwargs: Any) -> decoders.Decoder:
    """Return a pretrained MILAN model (eval mode, on CPU until `.to(device)`).

    Keyword arguments go to `torch.load` (`map_location='cpu'` by default like
    the reference's ModelConfig).
    """
    path = resolve(config, path)
    kwargs.setdefault('map_location', 'cpu')
    return decoders.Decoder.load(path, **kwargs).eval()


def pretrained_sharded(config: str = 'base',
                       path: Optional[os.PathLike] = None,
                       device=None,
                       src: int = 0,
                       **kwargs: Any) -> decoders.Decoder:
    """`pretrained` for one-process-per-GPU jobs: only rank `src` reads the
    checkpoint file; the module skeleton (vocabulary, constructor arguments)
    travels as one small object broadcast and the weights as ONE flat RCCL
    broadcast over xGMI (`sharding.broadcast_state_dict`).  Outside a process
    group this is `pretrained(...).to(device)`.
    """
    import torch.distributed as dist

    from milan_amd import serialize, sharding
    if not sharding.is_distributed():
        model = pretrained(config, path=path, **kwargs)
        return model if device is None else model.to(device)
    rank = dist.get_rank()
    # box = [payload, error]: a failure on the reading rank travels with the
    # broadcast and is raised on EVERY rank (otherwise the others would wait in
    # the collective forever)
    box, state_dict = [None, None], None
    if rank == src:
        try:
            kwargs.setdefault('map_location', 'cpu')
            payload = serialize.load_payload(resolve(config, path), **kwargs)
            state_dict = payload.pop('state_dict', None)
            if state_dict is None:
                raise ValueError('checkpoint has no state_dict')
            box[0] = payload
        except Exception as error:  # noqa: BLE001 -- re-raised below, everywhere
            box[1] = error
    dist.broadcast_object_list(box, src=src)
    if box[1] is not None:
        raise box[1]
    model = decoders.Decoder.deserialize(box[0]).eval()
    if device is not None:
        model = model.to(device)
    target = model.device
    shared = sharding.broadcast_state_dict(
        None if state_dict is None else dict(state_dict), target, src=src)
    # the reference loads with strict=False (serialize.py:222-252); here the
    # weights came over a collective, so a key lost on the way must not pass
    # silently: the broadcast has to carry exactly the reading rank's keys
    names = [sorted(state_dict)] if rank == src else [None]
    dist.broadcast_object_list(names, src=src)
    if sorted(shared) != names[0]:
        raise RuntimeError('weight broadcast lost or added keys: '
                           f'{sorted(set(names[0]) ^ set(shared))[:5]}')
    model.load_state_dict(shared, strict=False)
    return model
