"""Checkpoint format of the reference's `SerializableModule`.

`src/utils/serialize.py:80-118,188-219`: a checkpoint is a plain dict

    {'properties': {...constructor kwargs, nested Serializables as
                    {'properties': ..., 'children': ...}},
     'children':   {'encoder': 'PyramidConvEncoder'},
     'state_dict': OrderedDict(name -> tensor)}      # root object only

saved with `torch.save`.  spaCy `Language` objects inside the tokenizer are
stored as `(config, bytes)` (:104-107); this build never needs them for
inference, so they are carried through untouched -- and if the classes they
pickle to (spaCy/thinc `Config`) are not importable, `load_payload` substitutes
inert placeholders instead of failing.
"""
import io
import pickle
from typing import Any, Dict, Mapping

import torch


class _Placeholder(dict):
    """Stands in for an un-importable pickled class (keeps state, inert)."""
    _milan_placeholder_for = '?'

    def __init__(self, *args, **kwargs):
        try:
            super().__init__(*args, **kwargs)
        except (TypeError, ValueError):
            super().__init__()
            self['args'] = args

    def __setstate__(self, state):
        self['state'] = state

    def __reduce_ex__(self, protocol):
        return (dict, (dict(self),))


def _placeholder(module: str, name: str):
    return type(name, (_Placeholder,),
                {'_milan_placeholder_for': f'{module}.{name}'})


class _TolerantUnpickler(pickle.Unpickler):

    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return _placeholder(module, name)


class _TolerantPickle:
    """`pickle_module` for torch.load that tolerates missing classes."""
    __name__ = 'milan_tolerant_pickle'
    Unpickler = _TolerantUnpickler
    load = staticmethod(lambda f, **kw: _TolerantUnpickler(f, **kw).load())
    loads = staticmethod(
        lambda b, **kw: _TolerantUnpickler(io.BytesIO(b), **kw).load())
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    PickleError = pickle.PickleError
    UnpicklingError = pickle.UnpicklingError


def load_payload(file, **torch_load_kwargs) -> Dict[str, Any]:
    """`torch.load` for a reference checkpoint (src/utils/serialize.py:255-269).

    Keyword arguments are forwarded to `torch.load` like the reference does
    (`map_location='cpu'` is what `milan.pretrained` passes).
    """
    torch_load_kwargs.setdefault('map_location', 'cpu')
    torch_load_kwargs.setdefault('weights_only', False)
    torch_load_kwargs.setdefault('pickle_module', _TolerantPickle)
    payload = torch.load(file, **torch_load_kwargs)
    if not isinstance(payload, Mapping) or 'properties' not in payload:
        raise ValueError('not a serialized MILAN module: expected a dict with '
                         "'properties' / 'children' / 'state_dict'")
    return dict(payload)


def props(node: Any) -> Mapping[str, Any]:
    """Properties of a (possibly nested) serialized object."""
    if isinstance(node, Mapping) and 'properties' in node:
        return node['properties']
    raise ValueError('malformed serialized object: missing "properties"')


def serialized(properties: Mapping[str, Any],
               children: Mapping[str, Any] = None) -> Dict[str, Any]:
    return {'properties': dict(properties), 'children': dict(children or {})}
