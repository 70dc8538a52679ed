"""The sharding collectives through RCCL itself (backend "nccl"), on the one GPU
a test box has: a 1-rank process group.

Two ranks cannot share a GPU under RCCL, so the 2-rank tests run over gloo
(tests/test_host.py, tests/test_gpu_configs.py); what they cannot see is the
RCCL leg of the same code -- device-resident collective buffers, object
broadcast and barrier on the nccl backend, bench.py's torchrun entry.  Those run
here with world size 1, launched the way the driver launches an N-GPU run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = '''
import os, sys, torch
sys.path[:0] = {paths!r}
import torch.distributed as dist
from milan_amd import sharding, synthetic
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
assert sharding.is_distributed() and dist.get_backend() == 'nccl'
dev = torch.device('cuda', 0)
assert sharding._comm_device(dev).type == 'cuda'
sd = synthetic.milan_state_dict(68, config='resnet18', seed=0)
out = sharding.broadcast_state_dict(sd, dev, src=0)
assert list(out) == list(sd)
for k, v in sd.items():
    assert out[k].device.type == 'cuda' and out[k].dtype == v.dtype, k
    assert torch.equal(out[k].cpu(), v), k
tokens = torch.arange(7 * 15, device=dev).view(7, 15)
scores = torch.linspace(-3, 0, 7, device=dev)
t, s = sharding.gather_results(tokens, scores, dst=0)
assert torch.equal(t, tokens) and torch.equal(s, scores) and t.device.type == 'cuda'
assert sharding.max_over_ranks(1.25, dev) == 1.25
sharding.barrier()
sharding.finalize()
assert not sharding.is_distributed()
print('rccl ok')
'''


def _free_port():
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def _env(port):
    from milan_amd import hip
    hip.release_workspaces()  # the child shares this GPU
    env = dict(os.environ)
    env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('MILAN_DIST_BACKEND', None)
    return env


def test_sharding_collectives_on_the_nccl_backend():
    code = CHILD.format(paths=[p for p in sys.path if p])
    done = subprocess.run([sys.executable, '-c', code], env=_env(_free_port()),
                          capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-3000:]
    assert 'rccl ok' in done.stdout


def test_bench_under_torchrun_launch_line():
    """The driver's launch line (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node N ... bench.py --gpus N`) with N = 1: RANK / WORLD_SIZE /
    MASTER_* come from torchrun and one JSON line must come out."""
    port = _free_port()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
           '--gpus', '1', '--steps', '1', '--warmup', '0', '--chunk', '16',
           '--beam', '8', '--cpu-sample', '0', '--also-f32-steps', '0',
           '--from-host-steps', '0', '--other-configs', '0']
    done = subprocess.run(cmd, env=_env(port), capture_output=True, text=True,
                          timeout=900, cwd=ROOT)
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [l for l in done.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 1 and line['steps'] == 1
    assert line['config']['neurons_total'] == 16
    assert line['value'] > 0
