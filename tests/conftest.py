"""Shared pytest configuration: paths, markers, golden fixtures."""
import json
import os
import pathlib
import sys

import pytest
import torch

REPO = pathlib.Path(__file__).resolve().parent.parent
for p in (REPO, REPO / 'neuron-descriptions_amd'):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN_DIR = REPO / 'tests' / 'golden'

# torch's CPU convolutions collapse when oversubscribed on many-core hosts (the
# GPU box has 256 hardware threads: 88 s per neuron for the oracle's ResNet-101
# measured in round 1, against 0.5 s on 32 threads).
torch.set_num_threads(min(32, os.cpu_count() or 1))


def pytest_configure(config):
    config.addinivalue_line('markers',
                            'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def goldens():
    """Tensors produced by the imported reference (tests/golden/make_golden.py)."""
    return torch.load(GOLDEN_DIR / 'reference_goldens.pt')


@pytest.fixture(scope='session')
def golden_meta():
    with open(GOLDEN_DIR / 'reference_goldens.json') as f:
        return json.load(f)


@pytest.fixture(scope='session')
def forced_goldens():
    """Teacher-forcing / Decoder.score goldens (make_golden_forced.py)."""
    return torch.load(GOLDEN_DIR / 'reference_goldens_forced.pt')


@pytest.fixture(scope='session')
def forced_meta():
    with open(GOLDEN_DIR / 'reference_goldens_forced.json') as f:
        return json.load(f)


@pytest.fixture(scope='session')
def trunk_goldens():
    """Encoder goldens of the 'resnet18' / 'alexnet' configs
    (make_golden_trunks.py)."""
    return torch.load(GOLDEN_DIR / 'reference_goldens_trunks.pt')


@pytest.fixture(scope='session')
def trunk_meta():
    with open(GOLDEN_DIR / 'reference_goldens_trunks.json') as f:
        return json.load(f)
