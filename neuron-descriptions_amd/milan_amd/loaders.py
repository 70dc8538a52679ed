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


def pretrained(config: str = 'base',
               path: Optional[os.PathLike] = None,
               **kwargs: Any) -> decoders.Decoder:
    """Return a pretrained MILAN model (eval mode, on CPU until `.to(device)`).

    Keyword arguments go to `torch.load` (`map_location='cpu'` by default like
    the reference's ModelConfig).
    """
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
    kwargs.setdefault('map_location', 'cpu')
    return decoders.Decoder.load(path, **kwargs).eval()
