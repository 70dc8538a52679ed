"""Neuron sharding across the GPUs of one node.

The path is embarrassingly parallel over neurons: a description depends only
on that neuron's k exemplars and the read-only weights.  One process per GPU
(`torch.distributed`, backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU
for tests).  Exactly two collectives exist, neither on the data path:
  * one broadcast of the checkpoint tensors from rank 0 at start-up;
  * one (all-)gather of top-1 token ids + scores at the end.
The reference has no distributed code at all; this is
new, not a translation.
"""
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def partition(n: int, world: int, rank: int, align: int = 1) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of the neuron index owned by `rank`.

    Block size ceil(n / world) rounded up to a multiple of `align` (last ranks
    may get fewer / none); keeps the reference's CSV order when shards are
    concatenated rank by rank.  `align` = `predict`'s `batch_size` makes every
    shard start on a batch boundary of the single-process run, so the
    per-batch quantities of the reference (allennlp's early-exit length T',
    which feeds the rerank LM score, decoders.py:495-512) are evaluated over
    the same groups of neurons whatever the world size.
    """
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f'bad rank/world: {rank}/{world}')
    if align < 1:
        raise ValueError(f'bad align: {align}')
    per = -(-n // world)
    per = -(-per // align) * align
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized()


def launched_by_torchrun() -> bool:
    """True inside a process that `torch.distributed.run` (or any launcher that
    exports the same variables) started as one rank of a job."""
    return 'RANK' in os.environ and 'WORLD_SIZE' in os.environ


def self_launch(script: str, argv: Sequence[str], nproc: int) -> int:
    """Re-run `script argv...` as `nproc` ranks of one node and return the job's
    exit code (the caller is the launcher from then on and must not compute).

    `python bench.py --gpus 8` started WITHOUT torchrun ends up here: the ranks
    are started with `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1` on a free port, one rank per GPU
    over RCCL.  On a box with fewer GPUs than ranks (a 1-GPU test box) the ranks
    share the GPUs (`local_rank % device_count`) and the two collectives are
    staged through host memory (gloo): RCCL cannot put two ranks on one device.
    """
    import socket
    import subprocess
    import sys
    if nproc < 2:
        raise ValueError(f'self_launch needs nproc >= 2, got {nproc}')
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    devices = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if devices < nproc:
        env.setdefault('MILAN_DIST_BACKEND', 'gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={nproc}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), script, *argv]
    return subprocess.call(cmd, env=env)


def init_from_env(expected_world: Optional[int] = None,
                  backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise from torchrun's env (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*).

    Returns (rank, world, local_rank).  Single-process when WORLD_SIZE is
    unset or 1.
    """
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if expected_world is not None and expected_world != world:
        raise RuntimeError(
            f'--gpus {expected_world} but WORLD_SIZE={world}; launch with '
            'python -m torch.distributed.run --nproc-per-node N ...')
    if world > 1 and not is_distributed():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            # MILAN_DIST_BACKEND=gloo: collectives staged through host memory
            # (CPU tests; several ranks sharing one GPU on a 1-GPU box)
            backend = os.environ.get('MILAN_DIST_BACKEND') or (
                'nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def _comm_device(device: torch.device) -> torch.device:
    """Where collective buffers live: the GPU for RCCL, host memory for gloo."""
    if is_distributed() and dist.get_backend() == 'gloo':
        return torch.device('cpu')
    return torch.device(device)


def barrier() -> None:
    if is_distributed():
        dist.barrier()


def finalize() -> None:
    if is_distributed():
        dist.barrier()
        dist.destroy_process_group()


def max_over_ranks(value: float, device: torch.device) -> float:
    if not is_distributed():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value: float, device: torch.device) -> list:
    """`value` of every rank, in rank order, on every rank (one tiny all_gather):
    lets rank 0 report per-rank figures, so a straggler GPU shows in a scaling run."""
    if not is_distributed():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device(device))
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def broadcast_state_dict(sd: Optional[Dict[str, torch.Tensor]],
                         device: torch.device,
                         src: int = 0) -> Dict[str, torch.Tensor]:
    """Give every rank the checkpoint that only `src` has loaded.

    Metadata (names, shapes, dtypes) goes as one object broadcast; the float
    tensors are packed into ONE flat buffer so the payload is a single large
    RCCL broadcast (xGMI is point-to-point; one 280 MB message uses the links
    far better than ~800 small ones).  Integer tensors (num_batches_tracked)
    travel in a second, tiny buffer.
    """
    if not is_distributed():
        assert sd is not None
        return {k: v.to(device) for k, v in sd.items()}
    rank = dist.get_rank()
    meta: List = [None]
    if rank == src:
        assert sd is not None
        meta[0] = [(k, tuple(v.shape), str(v.dtype).replace('torch.', ''))
                   for k, v in sd.items()]
    dist.broadcast_object_list(meta, src=src)
    entries = meta[0]
    out: Dict[str, torch.Tensor] = {}
    for is_float in (True, False):
        names = [(k, s, d) for k, s, d in entries
                 if getattr(torch, d).is_floating_point == is_float]
        if not names:
            continue
        dtype = torch.float32 if is_float else torch.int64
        total = sum(int(torch.Size(s).numel()) for _, s, _ in names)
        flat = torch.empty(total, dtype=dtype, device=_comm_device(device))
        if rank == src:
            off = 0
            for k, s, _ in names:
                n = int(torch.Size(s).numel())
                flat[off:off + n] = sd[k].reshape(-1).to(device=flat.device,
                                                         dtype=dtype)
                off += n
        dist.broadcast(flat, src=src)
        flat = flat.to(device)
        off = 0
        for k, s, d in names:
            n = int(torch.Size(s).numel())
            out[k] = flat[off:off + n].view(s).to(getattr(torch, d))
            off += n
    # preserve the checkpoint's key order
    return {k: out[k] for k, _, _ in entries}


def gather_results(tokens: torch.Tensor,
                   scores: torch.Tensor,
                   dst: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Concatenate per-rank (n_r, T) tokens / (n_r,) scores on `dst` in rank
    order.  Ranks may hold different n_r (ragged last shard)."""
    if not is_distributed():
        return tokens, scores
    world, rank = dist.get_world_size(), dist.get_rank()
    out_device = tokens.device
    tokens = tokens.to(_comm_device(out_device))
    scores = scores.to(_comm_device(out_device))
    n = torch.tensor([tokens.shape[0]], dtype=torch.int64, device=tokens.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts_i = [int(c.item()) for c in counts]
    cap = max(counts_i) if counts_i else 0
    t_len = tokens.shape[1]
    pad_t = tokens.new_zeros(cap, t_len)
    pad_s = scores.new_zeros(cap)
    pad_t[:tokens.shape[0]] = tokens
    pad_s[:scores.shape[0]] = scores
    # all_gather rather than gather: 0.5 MB in total, and the one collective
    # every backend (RCCL, gloo) implements natively
    bufs_t = [torch.empty_like(pad_t) for _ in range(world)]
    bufs_s = [torch.empty_like(pad_s) for _ in range(world)]
    dist.all_gather(bufs_t, pad_t)
    dist.all_gather(bufs_s, pad_s)
    if rank == dst:
        return (torch.cat([b[:c] for b, c in zip(bufs_t, counts_i)
                           ]).to(out_device),
                torch.cat([b[:c] for b, c in zip(bufs_s, counts_i)
                           ]).to(out_device))
    return tokens.to(out_device), scores.to(out_device)


def shard_sequence(items: Sequence, world: int, rank: int) -> Sequence:
    lo, hi = partition(len(items), world, rank)
    return items[lo:hi]
