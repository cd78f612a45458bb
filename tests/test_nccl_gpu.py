"""The RCCL (`nccl` backend) branch of the multi-GPU path, on the one GPU a test box has: a single-rank group in a fresh child
process (tests/nccl_worker.py).  Everything else about sharding is covered over gloo / thread ranks (test_dist_gloo.py,
test_sampler_gpu.py); this is the branch where the collectives move device tensors, the one an 8-GPU run takes."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from . import util
from .test_sampler_gpu import _assert_samples_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'KPD_GEMM')}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), **extra)
    return env


def test_rccl_single_rank_runs_every_collective_on_device_tensors(cuda, tmp_path):
    from . import sharded_worker as W
    out = str(tmp_path / 'nccl.pt')
    r = subprocess.run([sys.executable, '-m', 'tests.nccl_worker', out], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rep = torch.load(out)
    assert rep['gathered'] == 4 and rep['buckets'] == 2 and rep['ranks_seen'] == 1
    torch.manual_seed(321)
    assert rep['seed'] == int(torch.randint(0, 2 ** 62, (1,)).item())
    # the ligands of the one-rank RCCL job = those of a run with no process group and the same per-complex noise seed
    model = W.build_model(cuda).use_complex_noise(W.SEED)
    ref = model._sample(W.pockets(cuda), W.N_LIG, rec_enc_batch_size=2, diff_batch_size=2)
    _assert_samples_equal(rep['samples'], ref)
    torch.manual_seed(55)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    ref2 = W.build_model(cuda).use_complex_noise(seed)._sample(W.pockets(cuda), W.N_LIG, rec_enc_batch_size=2, diff_batch_size=2)
    _assert_samples_equal(rep['samples_common_seed'], ref2)


def test_bench_line_under_rccl(cuda):
    """bench.py's distributed branch (nccl init, barrier-bracketed regions, MAX all-reduce of the region times, `ranks_seen`,
    per-rank times, the all-gather at the end of each region) with a one-rank RCCL group: what `--gpus 8` runs, minus 7 ranks."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--repeats', '2', '--batch', '4',
                        '--n-rec', '60', '--n-lig', '9', '--no-cpu-baseline', '--no-secondary'], cwd=ROOT,
                       env=_env(KPD_BENCH_DIST_AT_1='1', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1'), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d['ranks_seen'] == 1 and d['collective_backend'] == 'nccl' and len(d['per_rank_ms_per_step']) == 1
    assert d['n_gpus'] == 1 and d['value'] > 0 and d['roofline']['frac'] > 0
    assert d['per_rank_ms_per_step'][0] <= d['ms_per_step'] * 1.01
