"""Compute MILAN descriptions for a model/dataset pair on MI355X.

Drop-in for the reference's `scripts/compute_milan_descriptions.py` (same
positional arguments, flags and CSV output); the only edits are the two
imports.  With `torchrun --nproc-per-node N` the neurons are sharded over N
GPUs (weights broadcast from rank 0, descriptions gathered on rank 0).
"""
import argparse
import csv
import os
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))

import torch  # noqa: E402
from torch.utils import data  # noqa: E402

import milan_amd as milan  # noqa: E402
from milan_amd import datasets as milannotations  # noqa: E402
from milan_amd import sharding  # noqa: E402

parser = argparse.ArgumentParser(description='compute milan descriptions')
parser.add_argument('model', help='model architecture (e.g. alexnet)')
parser.add_argument('dataset', help='dataset model trained on (e.g. imagenet)')
parser.add_argument('--temperature', type=float, default=.2,
                    help='pmi temperature (default: .2)')
parser.add_argument('--beam-size', type=int, default=50,
                    help='beam size to rerank (default: 50)')
parser.add_argument('--data-dir', type=pathlib.Path,
                    help='root dir for datasets (default: $MILAN_DATA_DIR)')
parser.add_argument('--results-dir', type=pathlib.Path,
                    help='root dir for final results')
parser.add_argument('--milan', default='base',
                    help='milan model to use (default: base)')
parser.add_argument('--milan-path', type=pathlib.Path,
                    help='explicit checkpoint path (no network here)')
parser.add_argument('--device', help='manually set device (default: cuda)')
args = parser.parse_args()

rank, world, local = sharding.init_from_env()
device = args.device or f'cuda:{local}'

key = f'{args.model}/{args.dataset}'
data_dir = args.data_dir or pathlib.Path(os.environ.get('MILAN_DATA_DIR', 'data'))
data_root = data_dir / key
results_dir = args.results_dir or pathlib.Path(
    os.environ.get('MILAN_RESULTS_DIR', 'results')) / 'descriptions'

decoder = milan.pretrained(args.milan, path=args.milan_path)
decoder.to(device)

dataset = milannotations.load(key, path=data_root)
lo, hi = sharding.partition(len(dataset), world, rank)
shard = data.Subset(dataset, range(lo, hi)) if world > 1 else dataset
if world > 1:  # keep the mmap fast path for the shard
    shard.slice_uint8 = lambda a, b: dataset.slice_uint8(lo + a, lo + b)

predictions = decoder.predict(shard, strategy='rerank',
                              temperature=args.temperature,
                              beam_size=args.beam_size, device=device)
if world > 1:
    gathered = [None] * world
    torch.distributed.all_gather_object(gathered, list(predictions))
    predictions = [p for part in gathered for p in part]

if rank == 0:
    results_dir.mkdir(exist_ok=True, parents=True)
    rows = [('layer', 'unit', 'description')]
    for index, description in enumerate(predictions):
        layer, pos = dataset._index[index]
        unit = int(dataset.units_by_layer[layer][pos])
        rows.append((str(layer), str(unit), description))
    results_csv_file = results_dir / f'{key.replace("/", "_")}.csv'
    with results_csv_file.open('w') as handle:
        csv.writer(handle).writerows(rows)
sharding.finalize()
