"""`python bench.py --gpus N` with no launcher environment must start N ranks itself (fresh processes, before any GPU
call) -- the driver's N = 2 / 4 / 8 scaling runs depend on it.  CPU-only checks of that launcher."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(extra)
    return env


def test_bench_spawns_n_ranks_itself():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '3', '--steps', '2'], capture_output=True, text=True,
                       env=_clean_env(KPD_BENCH_SPAWN_ECHO='1'), timeout=120)
    assert r.returncode == 0, r.stderr
    lines = sorted((json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')), key=lambda d: d['rank'])
    assert [d['rank'] for d in lines] == [0, 1, 2] and [d['local_rank'] for d in lines] == [0, 1, 2]
    assert all(d['world'] == 3 and d['gpus'] == 3 and d['master'] == '127.0.0.1' for d in lines)
    assert len({d['port'] for d in lines}) == 1


def test_bench_spawns_eight_ranks():
    """The driver's N = 8 run: eight rank processes, local ranks 0..7, one rendezvous port (echo mode: no GPU, no torch)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2'], capture_output=True, text=True,
                       env=_clean_env(KPD_BENCH_SPAWN_ECHO='1'), timeout=120)
    assert r.returncode == 0, r.stderr
    lines = sorted((json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')), key=lambda d: d['rank'])
    assert [d['rank'] for d in lines] == list(range(8)) and [d['local_rank'] for d in lines] == list(range(8))
    assert all(d['world'] == 8 and d['gpus'] == 8 and d['master'] == '127.0.0.1' for d in lines) and len({d['port'] for d in lines}) == 1


def test_rank_thread_share():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.rank_cpu_threads(128, 8) == 16 and bench.rank_cpu_threads(16, 8) == 2 and bench.rank_cpu_threads(4, 8) == 1


def test_bench_honours_an_external_launcher():
    """Under torch.distributed.run the environment already names the rank: no second level of spawning."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4'], capture_output=True, text=True,
                       env=_clean_env(KPD_BENCH_SPAWN_ECHO='1', RANK='2', LOCAL_RANK='2', WORLD_SIZE='4', MASTER_ADDR='127.0.0.1',
                                      MASTER_PORT='29511'), timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and lines[0]['rank'] == 2 and lines[0]['world'] == 4


def test_bench_rank_failure_ends_the_job():
    """No GPU here: every rank exits non-zero, the launcher must return that instead of hanging."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2'], capture_output=True, text=True,
                       env=_clean_env(), timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and 'needs a GPU' in r.stderr
