"""Every BASELINE.json config through the product path on the GPU
(VERDICT r1 item 1).  Real dimensions throughout: ResNet-101 pyramid encoder,
k = 15, 224x224, F = 3904, H = 512, V = 5004, length 15, split_f16 precision
(the bench's mode).

  config 1  N = 256 (alexnet conv5 width), greedy, mi=False
            -> `Decoder.predict` from an on-disk uint8 dataset
  config 2  N = 1152 (alexnet: 64+192+384+256+256 units over five layers),
            beam 16 + rerank -> `Decoder.predict` from an on-disk dataset
  config 3  N = 3904 over 8 GPUs: one rank's shard (488 neurons), beam 16
  config 4  N = 4096, beam 50 + PMI rerank: the full 4096 with invariants,
            and one rank's shard (512) through the chunked `predict`
  config 5  65 536 neurons = a 1.2k-unit set replicated: one rank's shard
            (8192 neurons), beam 50
Each config is checked (i) by size-independent properties at full N and (ii)
against the CPU oracle on a fixed 32-neuron subset (near-tie excuses counted,
at most one).  The 2-rank bench run (gloo ranks sharing the one GPU) checks the
sharded path end to end against the 1-rank run.
"""
import json
import os
import pathlib
import subprocess
import sys

import numpy
import pytest
import torch

from milan_amd import datasets, decoders, encoders, hip, lang, lms, sharding
from milan_amd import synthetic
from oracle import milan_oracle as O

pytestmark = pytest.mark.gpu

NV, K, SIZE, LENGTH, LAMBDA = 5000, 15, 224, 15, 0.2
SUBSET = 32
TIE = 1e-3
REPO = pathlib.Path(__file__).resolve().parent.parent


@pytest.fixture(scope='module')
def world():
    hip.load_library()
    hip.require_device('cuda')
    sd = synthetic.milan_state_dict(NV + 4, 'resnet101', seed=0)
    idx = lang.Indexer(lang.Vocab(synthetic.vocab_tokens(NV)), None, True, True,
                       True, True, LENGTH)
    enc = encoders.PyramidConvEncoder('resnet101', pretrained=False)
    dec = decoders.Decoder(idx, enc, lms.LanguageModel(idx, 128, 512),
                           embedding_size=128, hidden_size=512)
    res = dec.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    dec.precision = 'split_f16'
    dec.to('cuda')
    return dec, sd


class Exemplars:
    """In-memory stand-in for a `TopImagesDataset` (same `slice_uint8` /
    `unit` / len contract), optionally replicating a base set."""
    transform_images = transform_masks = device = None

    def __init__(self, images, masks, total=None):
        self.images, self.masks = images, masks
        self.total = len(images) if total is None else total

    def __len__(self):
        return self.total

    def slice_uint8(self, lo, hi):
        base = len(self.images)
        idx = torch.arange(lo, hi) % base
        if hi - lo <= base and int(idx[0]) + (hi - lo) <= base:
            a = int(idx[0])
            return self.images[a:a + hi - lo], self.masks[a:a + hi - lo]
        return self.images[idx], self.masks[idx]


def make_exemplars(n, seed):
    im, mk = synthetic.exemplars(n, k=K, size=SIZE, seed=seed, device='cuda',
                                 zero_every=97)
    return im.cpu(), mk.cpu()


def write_dataset(root, images, masks, layer_sizes):
    lo = 0
    for li, width in enumerate(layer_sizes):
        d = root / f'layer{li}'
        d.mkdir(parents=True)
        numpy.save(d / 'images.npy', images[lo:lo + width].numpy())
        numpy.save(d / 'masks.npy', masks[lo:lo + width].numpy())
        lo += width
    assert lo == len(images)
    return datasets.TopImagesDataset(root)


def subset_indices(n, group=16):
    """Fixed 32-neuron subset (was 8): the whole first batch-of-16
    and the whole last one -- the CPU oracle has to run those two batches anyway."""
    last = ((n - 1) // group) * group
    tail = [last + j for j in range(group) if last + j < n and last + j >= group]
    return list(range(min(group, n)))[:SUBSET - len(tail)] + tail


def oracle_subset(images, masks, sd, idx, strategy, beam, group):
    """Oracle captions for neurons `idx`, each evaluated inside its own
    batch-of-`group` (allennlp's early exit is a per-batch quantity), so only
    those batches are run on the CPU."""
    vocab = synthetic.vocab_tokens(NV)
    out, cache = {}, {}
    for i in idx:
        lo = (i // group) * group
        hi = min(len(images), lo + group)
        if strategy == 'greedy':
            lo, hi = i, i + 1  # greedy has no batch-level coupling
        if (lo, hi) not in cache:
            feats = O.encode(O.byte_to_float(images[lo:hi]),
                             masks[lo:hi].float(), sd, chunk=15)
            cache[lo, hi] = O.forward(feats, sd, NV, strategy, length=LENGTH,
                                      beam_size=beam, temperature=LAMBDA,
                                      mi=False)
        w = cache[lo, hi]
        j = i - lo
        tie = False
        if strategy == 'greedy':
            top2 = w['predictions'][j].topk(2, dim=-1).values
            tie = bool(((top2[:, 0] - top2[:, 1]) < 1e-4).any())
        else:
            tie = (float(w['select_margin'][j]) < TIE or
                   float(w['rerank_margin'][j]) < TIE)
        out[i] = (O.reconstruct(w['tokens'][j].tolist(), vocab), tie)
    return out


def check_subset(captions, want):
    excused = 0
    for i, (caption, tie) in want.items():
        if captions[i] != caption:
            assert tie, (f'neuron {i}: {captions[i]!r} != oracle {caption!r} '
                         'without a near-tie')
            excused += 1
    assert excused <= 1, f'{excused} of {len(want)} took the near-tie excuse'


# --------------------------------------------------------------------------
def test_config1_alexnet_conv5_greedy_from_disk(world, tmp_path):
    dec, sd = world
    n = 256
    images, masks = make_exemplars(n, seed=101)
    ds = write_dataset(tmp_path / 'alexnet' / 'places365', images, masks, [n])
    caps = dec.predict(ds, strategy='greedy', mi=False, device='cuda',
                       display_progress_as=None)
    assert len(caps) == n and all(isinstance(c, str) for c in caps)
    # properties at full N: predict == forward on the same uint8 tensors ...
    out = dec(images, masks, strategy='greedy', mi=False)
    assert list(out.captions) == list(caps)
    assert out.tokens.shape == (n, LENGTH) and out.predictions.shape == (
        n, LENGTH, NV + 4)
    # ... every step's token is the argmax of its log-probs, scores are their sum
    best = out.predictions.max(dim=-1)
    assert torch.equal(best.indices, out.tokens)
    torch.testing.assert_close(best.values.sum(1), out.scores, rtol=1e-5,
                               atol=1e-3)
    torch.testing.assert_close(out.predictions.exp().sum(-1),
                               torch.ones(n, LENGTH, device='cuda'), rtol=1e-4,
                               atol=1e-4)
    # ... and the float inputs the reference's dataset hands out agree
    sample = ds[5]
    one = dec(sample.images[None].cuda(), sample.masks[None].cuda(),
              strategy='greedy', mi=False)
    assert one.captions[0] == caps[5]
    # oracle parity on the fixed subset
    idx = subset_indices(n)
    check_subset(caps, oracle_subset(images, masks, sd, idx, 'greedy', 1, 16))


def test_config2_alexnet_all_units_beam16_from_disk(world, tmp_path):
    dec, sd = world
    layers = [64, 192, 384, 256, 256]  # alexnet conv1..conv5
    n = sum(layers)
    images, masks = make_exemplars(n, seed=102)
    ds = write_dataset(tmp_path / 'alexnet' / 'imagenet', images, masks, layers)
    assert len(ds) == n == 1152
    caps = dec.predict(ds, strategy='rerank', beam_size=16,
                       temperature=LAMBDA, device='cuda',
                       display_progress_as=None)
    assert len(caps) == n
    assert ds.unit(64) == ('layer1', 0) and ds.unit(n - 1) == ('layer4', 255)
    # properties: chunks of 256 = bitwise the same as per-batch-of-16 launches
    for lo in (0, 640, 1136):
        out = dec(images[lo:lo + 16], masks[lo:lo + 16], strategy='rerank',
                  beam_size=16, temperature=LAMBDA)
        assert list(out.captions) == list(caps[lo:lo + 16]), lo
        assert (out.beam_scores[:, :-1] >= out.beam_scores[:, 1:]).all()
        assert out.beam_tokens.shape[:2] == (16, 16)
    idx = subset_indices(n)
    check_subset(caps, oracle_subset(images, masks, sd, idx, 'rerank', 16, 16))


def test_config3_resnet152_shard_beam16(world):
    dec, sd = world
    total, ranks = 3904, 8
    lo, hi = sharding.partition(total, ranks, 3, align=16)
    assert (lo, hi) == (3 * 496, 4 * 496)  # ceil(488 / 16) * 16 per rank
    n = hi - lo
    images, masks = make_exemplars(n, seed=103)
    caps = dec.predict(Exemplars(images, masks), strategy='rerank',
                       beam_size=16, temperature=LAMBDA, device='cuda',
                       display_progress_as=None)
    assert len(caps) == n
    # the last rank's ragged shard: 3904 - 7 * 496 = 432 neurons
    lo7, hi7 = sharding.partition(total, ranks, 7, align=16)
    assert hi7 - lo7 == 432 and hi7 == total
    tail = dec.predict(Exemplars(images[:hi7 - lo7], masks[:hi7 - lo7]),
                       strategy='rerank', beam_size=16, temperature=LAMBDA,
                       device='cuda', display_progress_as=None)
    assert list(tail) == list(caps[:hi7 - lo7])  # same neurons, same captions
    idx = subset_indices(n)
    check_subset(caps, oracle_subset(images, masks, sd, idx, 'rerank', 16, 16))


def test_config4_biggan_4096_beam50_rerank(world):
    """The headline workload at its full size, chunk by chunk through the C
    ABI with the invariants, then one rank's 512-neuron shard through
    `predict` (must reproduce the same captions bit for bit)."""
    dec, sd = world
    ctx = dec._context()
    total, chunk, beam = 4096, 256, 50
    stop, start = NV + 1, NV
    shard_caps = []
    first = None
    for ci in range(total // chunk):
        images, masks = synthetic.exemplars(chunk, k=K, size=SIZE,
                                            seed=400 + ci, device='cuda',
                                            zero_every=97)
        out = ctx.describe(images, masks, hip.RERANK, LENGTH, beam, False,
                           LAMBDA, group_size=16, want_features=True)
        if ci < 2:
            shard_caps += dec.indexer.reconstruct(out['tokens'].tolist())
        if ci == 0:
            first = (images.cpu(), masks.cpu())
        bs, bt = out['beam_scores'], out['beam_tokens']
        assert (bs[:, :-1] >= bs[:, 1:]).all()
        assert torch.isfinite(bs).all() and torch.isfinite(out['scores']).all()
        lens = out['out_len']
        assert lens.shape == (chunk // 16,) and (lens >= 1).all() and (
            lens <= LENGTH).all()
        # once a beam has stopped it stays stopped
        ended = (bt == stop).cumsum(-1) > 0
        assert (bt[ended] == stop).all()
        # all-zero masks -> exactly-zero feature rows (encoders.py:311-314)
        zero = masks.reshape(chunk * K, -1).sum(1) == 0
        feats = out['features'].reshape(chunk * K, -1)
        assert zero.any() and feats[zero].eq(0).all()
        assert feats[~zero].abs().sum(1).gt(0).all()
        # the rerank choice is the PMI argmax over the beams, scored by the
        # same LM kernels (decoders.py:495-512), T' per group of 16
        if ci % 5 == 0:
            tp = int(lens.max())
            seqs = torch.cat([
                torch.full((chunk * beam, 1), start, dtype=torch.long,
                           device='cuda'),
                bt[:, :, :tp].reshape(chunk * beam, tp)], 1)
            lm = ctx.lm_score(seqs).reshape(chunk, beam)
            if bool((lens == tp).all()):
                pmi = bs - LAMBDA * lm
                top = pmi.max(1)
                torch.testing.assert_close(out['scores'], top.values,
                                           rtol=1e-5, atol=1e-3)
    # one rank's shard (rank 0 of 8: neurons 0..511) through predict()
    im0, mk0 = first
    im1, mk1 = synthetic.exemplars(chunk, k=K, size=SIZE, seed=401,
                                   zero_every=97, device='cuda')
    shard = Exemplars(torch.cat([im0, im1.cpu()]), torch.cat([mk0, mk1.cpu()]))
    lo, hi = sharding.partition(total, 8, 0, align=16)
    assert (lo, hi) == (0, 512)
    caps = dec.predict(shard, strategy='rerank', beam_size=beam,
                       temperature=LAMBDA, device='cuda',
                       display_progress_as=None)
    assert list(caps) == shard_caps
    idx = subset_indices(32)  # oracle: 8 neurons of the first two batches
    check_subset(caps, oracle_subset(im0, mk0, sd, idx, 'rerank', beam, 16))


def test_config5_replicated_64k_shard_beam50(world):
    """65 536 neurons = a base set replicated (BASELINE config 5); one rank's
    shard is 8192 neurons.  Replicas must get identical captions."""
    dec, sd = world
    lo, hi = sharding.partition(65536, 8, 5, align=16)
    n, base = hi - lo, 512
    assert n == 8192
    images, masks = make_exemplars(base, seed=105)
    caps = dec.predict(Exemplars(images, masks, total=n), strategy='rerank',
                       beam_size=50, temperature=LAMBDA, device='cuda',
                       display_progress_as=None)
    assert len(caps) == n
    for rep in range(1, n // base):
        assert list(caps[rep * base:(rep + 1) * base]) == list(caps[:base]), rep
    idx = subset_indices(base)   # 32 neurons: the first and the last batch-of-16 of the base set
    check_subset(caps, oracle_subset(images, masks, sd, idx, 'rerank', 50, 16))


# --------------------------------------------------------------------------
def _bench(args, nproc, port):
    env = dict(os.environ, MILAN_DIST_BACKEND='gloo',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    hip.release_workspaces()  # the children share this GPU
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ['-m', 'torch.distributed.run', '--nnodes=1',
                f'--nproc-per-node={nproc}', '--master-addr', '127.0.0.1',
                '--master-port', str(port)]
    cmd += [str(REPO / 'bench.py'), '--gpus', str(nproc)] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900,
                         env=env, cwd=str(REPO))
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    return json.loads(line)


def test_two_rank_bench_equals_one_rank(world):
    """bench.py --gpus 2 --neurons-total 512 as two gloo ranks sharing the one
    GPU: weights broadcast from rank 0, batch-aligned shards, final gather --
    and the gathered tokens equal the 1-rank run's, bit for bit."""
    args = ['--neurons-total', '512', '--chunk', '128', '--warmup', '0',
            '--cpu-sample', '0', '--also-f32-steps', '0', '--other-configs',
            '0', '--from-host-steps', '0', '--beam', '16']
    port = 29600 + os.getpid() % 300
    two = _bench(args, 2, port)
    one = _bench(args, 1, port + 1)
    assert two['n_gpus'] == 2 and one['n_gpus'] == 1
    assert two['scaling'] == one['scaling'] == 'strong'
    assert two['config']['gathered_tokens'] == [512, LENGTH]
    assert two['config']['neurons_total'] == 512
    assert two['steps'] == 2 and one['steps'] == 4
    assert (two['config']['gathered_tokens_sha256'] ==
            one['config']['gathered_tokens_sha256'])
    assert two['roofline']['stages'], 'per-stage roofline missing'


def test_two_rank_bench_weak_mode_all_legs(world):
    """The driver's multi-GPU command line (weak scaling: --steps per rank)
    with every leg on: resident run, PCIe-inclusive leg, f32 leg, gather."""
    args = ['--steps', '2', '--chunk', '32', '--warmup', '1', '--cpu-sample',
            '0', '--also-f32-steps', '1', '--beam', '8']
    out = _bench(args, 2, 29950 + os.getpid() % 40)
    assert out['n_gpus'] == 2 and out['scaling'] == 'weak'
    assert out['steps'] == 2 and out['warmup'] == 1
    assert out['config']['neurons_total'] == 2 * 2 * 32
    assert out['config']['gathered_tokens'] == [128, LENGTH]
    assert out['value'] > 0 and out['pcie_inclusive']['value'] > 0
    assert out['pcie_inclusive']['steps'] == 2
    assert out['f32_mode']['steps'] == 1 and out['f32_mode']['value'] > 0
    assert out['cpu_baseline'] is None and 'other_configs' not in out
    names = [st['stage'] for st in out['roofline']['stages']]
    assert 'encoder.layer3' in names and 'decoder.search' in names


def test_bench_gpus_2_without_torchrun_launches_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` started WITHOUT
    torch.distributed.run must not die on WORLD_SIZE=1 -- the process becomes the
    launcher of two ranks (sharding.self_launch; gloo ranks sharing the one GPU of this
    box, RCCL one-rank-per-GPU on a node that has the GPUs) and rank 0 prints the one
    JSON line.  This is the form a SCALE run without torchrun would take
    (`python3 bench.py --gpus 8 --neurons-total 4096`)."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR',
                        'MASTER_PORT', 'MILAN_DIST_BACKEND')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    hip.release_workspaces()  # the children share this GPU
    cmd = [sys.executable, str(REPO / 'bench.py'), '--gpus', '2', '--steps', '2',
           '--warmup', '0', '--chunk', '64', '--cpu-sample', '0',
           '--also-f32-steps', '0', '--no-profile', '--from-host-steps', '0',
           '--other-configs', '0', '--live-traffic', '0']
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=str(REPO),
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and len(line['per_rank']['neurons']) == 2
    assert line['steps'] == 2 and line['value'] > 0
    assert line['scaling'] in ('weak', 'strong')


def test_bench_gpus_8_strong_mode_is_the_scale_command(world):
    """the exact shape of the SCALE command -- `bench.py --gpus 8
    --neurons-total N` launched WITHOUT torchrun -- as eight gloo ranks sharing this
    box's one GPU: rc 0, one JSON line, n_gpus 8, and the gathered tokens equal the
    1-rank run's bit for bit (shards are batch-aligned, kernels are chosen from the layer
    shape only)."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR',
                        'MASTER_PORT', 'MILAN_DIST_BACKEND')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    hip.release_workspaces()  # the children share this GPU
    common = ['--neurons-total', '512', '--chunk', '64', '--warmup', '0',
              '--cpu-sample', '0', '--also-f32-steps', '0', '--no-profile',
              '--from-host-steps', '0', '--other-configs', '0', '--live-traffic', '0',
              '--beam', '16']
    lines = {}
    for gpus in (8, 1):
        out = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--gpus',
                              str(gpus)] + common, capture_output=True, text=True,
                             env=env, cwd=str(REPO), timeout=1500)
        assert out.returncode == 0, out.stderr[-3000:]
        js = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
        assert len(js) == 1, out.stdout[-2000:]
        lines[gpus] = json.loads(js[0])
    eight, one = lines[8], lines[1]
    assert eight['n_gpus'] == 8 and len(eight['per_rank']['neurons']) == 8
    assert eight['scaling'] == 'strong' and eight['config']['neurons_total'] == 512
    assert eight['ranks_per_gpu'] == 8 and eight['dist_backend'] == 'gloo'
    assert eight['status_flags'] == 0 and one['status_flags'] == 0
    assert (eight['config']['gathered_tokens_sha256'] ==
            one['config']['gathered_tokens_sha256'])
