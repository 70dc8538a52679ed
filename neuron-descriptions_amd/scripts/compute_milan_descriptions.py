"""Describe every neuron of a model/dataset pair with MILAN on MI355X.

Command-line contract of the reference's `scripts/compute_milan_descriptions.py`
(positional `model dataset`, `--temperature --beam-size --data-dir
--results-dir --milan --device`, CSV `layer,unit,description` named
`<model>_<dataset>.csv`), so existing job scripts keep working.  Additions:
`--milan-path` (there is no downloader here), `--precision {auto,split_f16,f32}`
(default `auto`: the split-f16 MFMA path the published throughput belongs to, a
call whose activations leave its range is rerun in exact fp32 with a warning;
`f32` = the reference's arithmetic at a third of the speed) and multi-GPU sharding -- with
`--gpus N` (the script starts its own N ranks) or under `torchrun
--nproc-per-node N` every rank describes a contiguous block of the neurons and
rank 0 writes the CSV in the reference's order.
"""
import argparse
import csv
import os
import pathlib
import sys
from typing import List, Sequence, Tuple

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))

import torch  # noqa: E402

import milan_amd as milan  # noqa: E402
from milan_amd import datasets as milannotations  # noqa: E402
from milan_amd import sharding  # noqa: E402


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description='compute milan descriptions')
    p.add_argument('model', help='model architecture (e.g. alexnet)')
    p.add_argument('dataset', help='dataset model trained on (e.g. imagenet)')
    p.add_argument('--milan', default='base',
                   help='milan model to use (default: base)')
    p.add_argument('--milan-path', type=pathlib.Path,
                   help='explicit checkpoint path (no network here)')
    p.add_argument('--temperature', type=float, default=.2,
                   help='pmi temperature (default: .2)')
    p.add_argument('--beam-size', type=int, default=50,
                   help='beam size to rerank (default: 50)')
    p.add_argument('--data-dir', type=pathlib.Path,
                   help='root dir for datasets (default: $MILAN_DATA_DIR)')
    p.add_argument('--results-dir', type=pathlib.Path,
                   help='root dir for final results')
    p.add_argument('--device', help='manually set device (default: cuda)')
    p.add_argument('--precision', choices=('auto', 'split_f16', 'f32'), default=None,
                   help='arithmetic of the HIP path: auto = split_f16 with a loud '
                   'per-call fallback to f32 (default, or $MILAN_PRECISION), '
                   'split_f16 = raise on saturation, f32 = exact fp32 MFMA')
    p.add_argument('--gpus', type=int, default=1,
                   help='GPUs of this node to shard the neurons over: N > 1 '
                   'without torchrun starts N ranks itself (default: 1)')
    return p.parse_args(argv)


def env_dir(value, variable: str, default: str) -> pathlib.Path:
    return value or pathlib.Path(os.environ.get(variable, default))


def describe_shard(decoder, dataset, world: int, rank: int,
                   batch_size: int = 16, **predict_kwargs) -> List[str]:
    """This rank's block of neurons -> captions for the whole dataset (every
    rank returns the full list, in dataset order).  Blocks start on multiples
    of `batch_size`, so the reference's per-batch quantities (allennlp's
    early-exit length) are taken over the same neurons as in a 1-process run."""
    lo, hi = sharding.partition(len(dataset), world, rank, align=batch_size)
    predict_kwargs['batch_size'] = batch_size
    shard = dataset
    if world > 1:
        # keeps the memory-mapped uint8 fast path of `predict` (incl. the one-pass
        # `out=` fill of the pinned staging buffers) for the block
        shard = milannotations.ShardView(dataset, lo, hi)
    mine = list(decoder.predict(shard, **predict_kwargs))
    if world == 1:
        return mine
    parts: List[Sequence[str]] = [()] * world
    torch.distributed.all_gather_object(parts, mine)
    return [caption for part in parts for caption in part]


def csv_rows(dataset, captions: Sequence[str]) -> List[Tuple[str, str, str]]:
    rows = [('layer', 'unit', 'description')]
    for index, caption in enumerate(captions):
        layer, unit = dataset.unit(index)
        rows.append((str(layer), str(unit), caption))
    return rows


def main(argv=None) -> None:
    args = parse_args(argv)
    if args.gpus > 1 and not sharding.launched_by_torchrun():
        # become the launcher of `--gpus` ranks (one per GPU, RCCL)
        sys.exit(sharding.self_launch(
            str(pathlib.Path(__file__).resolve()),
            sys.argv[1:] if argv is None else list(argv), args.gpus))
    rank, world, local = sharding.init_from_env()
    device = args.device or f'cuda:{local % max(1, torch.cuda.device_count())}'
    key = f'{args.model}/{args.dataset}'

    # rank 0 reads the checkpoint; the others get it by RCCL broadcast
    decoder = milan.pretrained_sharded(args.milan, path=args.milan_path,
                                       device=device)
    if args.precision is not None:
        decoder.precision = args.precision
    dataset = milannotations.load(
        key, path=env_dir(args.data_dir, 'MILAN_DATA_DIR', 'data') / key)
    captions = describe_shard(decoder, dataset, world, rank,
                              strategy='rerank',
                              temperature=args.temperature,
                              beam_size=args.beam_size, device=device)
    if rank == 0:
        out_dir = args.results_dir or env_dir(None, 'MILAN_RESULTS_DIR',
                                              'results') / 'descriptions'
        out_dir.mkdir(exist_ok=True, parents=True)
        with (out_dir / f'{key.replace("/", "_")}.csv').open('w') as handle:
            csv.writer(handle).writerows(csv_rows(dataset, captions))
    sharding.finalize()


if __name__ == '__main__':
    main()
