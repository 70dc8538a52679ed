"""CPU: the tally cache this repo WRITES is netdissect's format.

`RunningTopK.state_dict()` must load into the reference's own class
(src/deps/netdissect/runningstats.py:118-149) and give the same `result()`: upstream
adds `linear_index` -- row offsets into its flattened candidate buffer, shape
(units, 1) -- to a (units, k) index tensor.  The reference is imported only when it is
present (this container); the format itself is checked everywhere.
"""
import pathlib
import sys

import numpy
import pytest
import torch

from milan_amd import exemplars

REFERENCE = pathlib.Path('/root/reference')


def _filled_topk(units, k, filled, count, seed=0):
    g = torch.Generator().manual_seed(seed)
    top = exemplars.RunningTopK(k=k)
    values, order = torch.rand(units, k, generator=g).sort(dim=1, descending=True)
    top.values = values
    top.index = torch.randint(0, count, (units, k), generator=g)
    top.values[:, filled:] = 0
    top.filled, top.count = filled, count
    return top


@pytest.mark.parametrize('units,k,filled', [(7, 5, 5), (4, 4, 4), (6, 15, 9), (1, 3, 2)])
def test_topk_state_dict_is_netdissect_format(tmp_path, units, k, filled):
    top = _filled_topk(units, k, filled, count=1000 + units)
    state = top.state_dict()
    width = max(10, 5 * k)
    assert state['top_data'].shape == state['top_index'].shape == (units, width)
    assert state['linear_index'].shape == (units, 1)
    assert state['linear_index'][:, 0].tolist() == [u * width for u in range(units)]
    assert state['next'] == filled and state['k'] == k and state['data_shape'] == (units,)
    # through the file, as the tally cache does
    path = tmp_path / 'tally.npz'
    numpy.savez(path, **{k_: v for k_, v in state.items() if v is not None})
    loaded = dict(numpy.load(path, allow_pickle=True))
    want_values, want_index = top.result()
    if REFERENCE.exists():
        # (the module's last lines import statsmodels for functions nothing here uses;
        # absent in this image, stubbed like tests/golden/make_golden_sketch.py does)
        import types
        for name in ('statsmodels', 'statsmodels.stats',
                     'statsmodels.stats.correlation_tools'):
            if name not in sys.modules:
                stub = types.ModuleType(name)
                stub.cov_nearest = stub.corr_nearest = None
                sys.modules[name] = stub
        sys.path.insert(0, str(REFERENCE))
        # (the reference tree is read-only: no __pycache__ directories in it)
        bytecode, sys.dont_write_bytecode = sys.dont_write_bytecode, True
        try:
            from src.deps.netdissect import runningstats
        finally:
            sys.dont_write_bytecode = bytecode
            sys.path.remove(str(REFERENCE))
        ref = runningstats.RunningTopK(state=loaded)
        got_values, got_index = ref.result()
        assert torch.equal(got_values, want_values)
        assert torch.equal(got_index, want_index)
        assert ref.size() == top.size()
    # and the same arithmetic spelled out (what upstream's result() does), so that the
    # check also runs where the reference is absent
    data = torch.from_numpy(loaded['top_data'])[:, :int(loaded['next'])]
    keep = min(k, int(loaded['next']))
    vals, bti = data.topk(keep, sorted=True)
    flat = torch.from_numpy(loaded['top_index']).view(-1)
    idx = flat[(bti + torch.from_numpy(loaded['linear_index'])).view(-1)].view(*bti.shape)
    assert torch.equal(vals, want_values) and torch.equal(idx, want_index)


def test_checked_units_notices_edits_and_new_tensors():
    owner = type('Owner', (), {})()
    units = torch.tensor([0, 3, -1], dtype=torch.int32)
    first = exemplars._checked_units(owner, units, 8)
    assert first.tolist() == [0, 3, 7]
    assert exemplars._checked_units(owner, units, 8) is first  # cached
    units[1] = 5  # in place: the version counter moves
    assert exemplars._checked_units(owner, units, 8).tolist() == [0, 5, 7]
    with pytest.raises(IndexError):
        exemplars._checked_units(owner, torch.tensor([9], dtype=torch.int32), 8)
