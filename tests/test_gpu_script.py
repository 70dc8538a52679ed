"""The drop-in driver: `compute_milan_descriptions.py <model> <dataset>` on a
synthetic checkpoint + exemplar directory, CSV compared with the oracle.

Round 6: the script delivers the benchmarked arithmetic by default --
`--precision auto` = split_f16 with the loud per-call fallback to f32 -- so the CSV of
`auto` must equal the CSV of `f32`, and a network whose activations leave the split
format's range must fall back with a warning (`auto`) or fail (`split_f16`), never
hand back clamped features."""
import csv
import os
import pathlib
import subprocess
import sys

import numpy
import pytest
import torch

from milan_amd import decoders, encoders, lang, lms, synthetic
from oracle import milan_oracle as O

pytestmark = pytest.mark.gpu
REPO = pathlib.Path(__file__).resolve().parent.parent
SCRIPT = REPO / 'neuron-descriptions_amd' / 'scripts' / 'compute_milan_descriptions.py'
NV, K = 60, 5
BLOCKS = synthetic.RESNET_BLOCKS['resnet50']


def _checkpoint(tmp_path, width, sd):
    idx = lang.Indexer(lang.Vocab(synthetic.vocab_tokens(NV)), None, True, True,
                       True, True, 15)
    enc = encoders.PyramidConvEncoder('resnet50', width=width, pretrained=False)
    dec = decoders.Decoder(idx, enc, lms.LanguageModel(idx, 16, 64),
                           embedding_size=16, hidden_size=64, length=8)
    dec.load_state_dict(sd, strict=True)
    models = tmp_path / 'models'
    models.mkdir()
    dec.save(models / 'base.pth')
    return models


def _dataset(tmp_path, size, units, zero_every=None):
    data_root = tmp_path / 'data' / 'alexnet' / 'imagenet'
    all_images, all_masks, rows = [], [], []
    for li, (layer, n) in enumerate(sorted(units.items())):
        kw = {} if zero_every is None else {'zero_every': zero_every}
        images, masks = synthetic.exemplars(n, k=K, size=size, seed=40 + li, **kw)
        (data_root / layer).mkdir(parents=True)
        numpy.save(data_root / layer / 'images.npy', images.numpy())
        numpy.save(data_root / layer / 'masks.npy', masks.numpy())
        all_images.append(images)
        all_masks.append(masks)
        rows += [(layer, str(u)) for u in range(n)]
    return torch.cat(all_images), torch.cat(all_masks), rows


def _run(tmp_path, models, *extra, results='results', check=True, env_extra=None):
    env = dict(os.environ, MILAN_MODELS_DIR=str(models),
               MILAN_DATA_DIR=str(tmp_path / 'data'),
               MILAN_RESULTS_DIR=str(tmp_path / results))
    env.pop('MILAN_PRECISION', None)
    env.pop('MILAN_ON_SATURATION', None)
    env.update(env_extra or {})
    done = subprocess.run([sys.executable, '-W', 'always', str(SCRIPT), 'alexnet', 'imagenet',
                           '--beam-size', '4', '--temperature', '0.2', *extra],
                          check=check, env=env, cwd=tmp_path, capture_output=True, text=True)
    out = tmp_path / results / 'descriptions' / 'alexnet_imagenet.csv'
    got = None
    if out.exists():
        with out.open() as handle:
            got = list(csv.reader(handle))
    return done, got


def test_compute_milan_descriptions_script(tmp_path):
    width, size = 16, 64
    sd = synthetic.milan_state_dict(NV + 4, 'resnet50', seed=5, width=width,
                                    hidden_size=64, embedding_size=16,
                                    lm_hidden_size=64, lm_embedding_size=16)
    models = _checkpoint(tmp_path, width, sd)
    images, masks, rows = _dataset(tmp_path, size, {'conv4': 5, 'conv5': 3})
    _, got = _run(tmp_path, models)
    assert got[0] == ['layer', 'unit', 'description']

    # the script (like the reference's) runs predict with batch_size 16 >= 8
    # neurons: one allennlp group
    feats = O.encode(O.byte_to_float(images), masks.float(), sd, blocks=BLOCKS)
    want = O.forward(feats, sd, NV, 'rerank', length=8, beam_size=4)
    caps = [O.reconstruct(t.tolist(), synthetic.vocab_tokens(NV))
            for t in want['tokens']]
    assert [tuple(r) for r in got[1:]] == [(l, u, c)
                                           for (l, u), c in zip(rows, caps)]

    # the default IS `--precision auto`, and its CSV equals the exact-fp32 mode's
    _, auto = _run(tmp_path, models, '--precision', 'auto', results='r_auto')
    _, f32 = _run(tmp_path, models, '--precision', 'f32', results='r_f32')
    _, split = _run(tmp_path, models, '--precision', 'split_f16', results='r_split')
    assert auto == got and f32 == got and split == got


def test_script_falls_back_loudly_when_the_network_saturates(tmp_path):
    """A homogeneously scaled trunk whose layer4 activations reach ~5000 (beyond
    65504 / 2^5): `auto` warns and writes the f32 CSV, `split_f16` fails instead of
    writing clamped descriptions."""
    from test_gpu_dtype_error import PREFIX, _scaled_network
    width, size = 64, 96
    sd = synthetic.milan_state_dict(NV + 4, 'resnet50', seed=7, width=width,
                                    hidden_size=64, embedding_size=16,
                                    lm_hidden_size=64, lm_embedding_size=16)
    for name, value in _scaled_network(150.0).items():
        assert name.startswith(PREFIX) and name in sd
        sd[name] = value
    models = _checkpoint(tmp_path, width, sd)
    _dataset(tmp_path, size, {'conv5': 3}, zero_every=0)
    done, auto = _run(tmp_path, models, '--precision', 'auto', results='r_auto')
    assert 'rerunning this call in f32' in done.stderr, done.stderr[-2000:]
    _, f32 = _run(tmp_path, models, '--precision', 'f32', results='r_f32')
    assert auto == f32 and len(auto) == 4
    done, split = _run(tmp_path, models, '--precision', 'split_f16', results='r_split',
                       check=False)
    assert done.returncode != 0 and split is None
    assert 'FloatingPointError' in done.stderr and 'clamped' in done.stderr
