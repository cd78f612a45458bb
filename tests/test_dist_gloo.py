"""Multi-rank path on CPU: complexes sharded over ranks, one all-gather of the ligand tensors
(gloo here; the same code runs over RCCL on the GPU box)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from keypoint_diffusion_amd import graph as G
from keypoint_diffusion_amd import synth
from keypoint_diffusion_amd.dist import all_gather_ligands, allreduce_gradients, shard_complexes

from . import util

N_REC = [20, 35, 12, 28, 16]
N_LIG = [5, 9, 3, 7, 4]


def _plain(o):
    """Tensors by value (numpy) for the result queue: a torch tensor in a multiprocessing Queue travels as a file descriptor that the
    parent fetches from the LIVE child, and a child that has already left (barrier, destroy, exit) made q.get() fail with EOFError --
    one run in ten on a busy host."""
    if isinstance(o, torch.Tensor):
        return o.detach().cpu().numpy().copy()
    if isinstance(o, (list, tuple)):
        return type(o)(_plain(x) for x in o)
    if isinstance(o, dict):
        return {k: _plain(v) for k, v in o.items()}
    return o


def _torchify(o):
    import numpy as np
    if isinstance(o, np.ndarray):
        return torch.from_numpy(o)
    if isinstance(o, (list, tuple)):
        return type(o)(_torchify(x) for x in o)
    if isinstance(o, dict):
        return {k: _torchify(v) for k, v in o.items()}
    return o


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    gs = synth.synth_complexes(N_REC, N_LIG, 4, util.CUTOFFS_ALL_ATOM, seed=40)
    mine = shard_complexes(N_REC, world)[rank]
    g = G.batch([gs[i] for i in mine])
    # stand-in for the sampler's output: mark every ligand with its global complex index
    off = 0
    for i in mine:
        g.nodes['lig'].data['x_0'][off:off + N_LIG[i]] += 100.0 * i
        off += N_LIG[i]
    pos, feat = all_gather_ligands(g)
    q.put(_plain((rank, [p.clone() for p in pos], [f.clone() for f in feat])))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_is_contiguous_and_balanced():
    for world in (1, 2, 3, 5):
        parts = shard_complexes(N_REC, world)
        assert len(parts) == world
        flat = [i for r in parts for i in r]
        assert flat == list(range(len(N_REC)))
        assert all(len(r) >= 1 for r in parts)
    parts = shard_complexes([600, 150, 150, 150, 150], 2)
    assert list(parts[0]) == [0]                         # cost-balanced, not count-balanced


def test_all_gather_ligands_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [_torchify(q.get(timeout=120)) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gs = synth.synth_complexes(N_REC, N_LIG, 4, util.CUTOFFS_ALL_ATOM, seed=40)
    for rank, pos, feat in res:
        assert len(pos) == len(N_LIG)
        for i, (p, f) in enumerate(zip(pos, feat)):
            assert p.shape == (N_LIG[i], 3) and f.shape == (N_LIG[i], 10)
            assert torch.allclose(p, gs[i].nodes['lig'].data['x_0'] + 100.0 * i, atol=1e-5)
            assert torch.allclose(f, gs[i].nodes['lig'].data['h_0'], atol=1e-6)


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((257, 515), (257,), (1, 257), (3, 5))]
    gen = torch.Generator().manual_seed(100 + rank)
    for i, p in enumerate(params):
        p.grad = None if i == 3 else torch.randn(p.shape, generator=gen)      # one frozen parameter
    n_buckets = allreduce_gradients(params, bucket_bytes=300 * 1024)          # forces two buckets
    q.put(_plain((rank, n_buckets, [None if p.grad is None else p.grad.clone() for p in params])))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_gradients_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([_torchify(q.get(timeout=120)) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shapes = ((257, 515), (257,), (1, 257))
    want = []
    for i, s in enumerate(shapes):
        gs = []
        for rank in range(2):
            gen = torch.Generator().manual_seed(100 + rank)
            for j in range(i + 1):
                t = torch.randn(shapes[j], generator=gen)
            gs.append(t)
        want.append((gs[0] + gs[1]) / 2)
    for rank, n_buckets, grads in res:
        assert n_buckets == 2 and grads[3] is None
        for got, ref in zip(grads[:3], want):
            assert torch.allclose(got, ref, atol=1e-6)


def _map_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from keypoint_diffusion_amd.dist import common_seed, sharded_map, sharding_active
    assert sharding_active()
    n_lig = [5, 9, 3, 7, 4, 11, 2]
    costs = [600 + n * n for n in n_lig]
    seen = []

    def fn(mine):                       # stand-in for the reverse loop: ligand c = its global index, sized n_lig[c]
        seen.extend(mine)
        return ([torch.full((n_lig[c], 3), float(c)) for c in mine], [torch.full((n_lig[c], 10), 10.0 * c) for c in mine])

    pos, feat = sharded_map(costs, fn)
    torch.manual_seed(1000 + rank)      # ranks disagree on their generators; the common seed is rank 0's draw
    q.put(_plain((rank, seen, [p.clone() for p in pos], [f.clone() for f in feat], common_seed())))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_map_world3_returns_everything_in_input_order():
    """The plumbing of the sharded sampler (`KeypointDiffusion._sample` under a process group): contiguous cost-balanced shards,
    one gather, every rank ends with every complex in input order; more ranks than some shards need is fine."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_map_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_torchify(q.get(timeout=120)) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_lig = [5, 9, 3, 7, 4, 11, 2]
    assert sorted(c for r in res for c in r[1]) == list(range(len(n_lig)))         # each complex sampled exactly once
    assert len({r[4] for r in res}) == 1                                             # one seed for all ranks
    for rank, seen, pos, feat, _ in res:
        assert list(seen) == sorted(seen) and len(pos) == len(n_lig)
        for c, (p, f) in enumerate(zip(pos, feat)):
            assert p.shape == (n_lig[c], 3) and f.shape == (n_lig[c], 10)
            assert bool((p == float(c)).all()) and bool((f == 10.0 * c).all())


# ---- KeypointDiffusion._sample at world 8 (configs[3] / configs[4] are 8-rank jobs) --------------------------------------
# The reverse loop needs the GPU; here it is replaced by a stand-in that marks every ligand with its global complex id, so the
# test covers everything AROUND it: the flat (pocket, replicate) list, the cost-balanced contiguous shards, which pockets a rank
# encodes, the per-complex ids handed to the noise streams, the common seed, the gather, the regrouping per pocket.
S8_N_REC = [40, 75, 22, 60, 33, 90]
S8_N_LIG = [[5, 9], [3], [7, 4, 6], [8], [2, 2], [11]]               # 10 complexes on 8 ranks; S8_FEW: 3 complexes on 8 ranks
S8_FEW = [[5], [], [7, 4], [], [], []]


def _stub_model():
    from keypoint_diffusion_amd.ligand_diffuser import KeypointDiffusion

    class Stub(KeypointDiffusion):
        def sample_from_encoded_receptors(self, g, visualize=False, init_lig_pos=None, complex_ids=None, use_graph=None):
            assert self._noise_seed is not None and complex_ids is not None
            self.calls.append((complex_ids.tolist(), g.batch_num_nodes('kp').tolist()))
            ns = g.batch_num_nodes('lig').tolist()
            return ([torch.full((n, 3), float(c)) for n, c in zip(ns, complex_ids.tolist())],
                    [torch.full((n, 10), float(self._noise_seed % 1000) + c) for n, c in zip(ns, complex_ids.tolist())])

    m = Stub(10, 10, None, n_timesteps=4, architecture='egnn', rec_encoder_type='fixed',
             graph_config=dict(n_keypoints=20, graph_cutoffs=util.CUTOFFS_ALL_ATOM), dynamics_config=util.EGNN_C2, precision=1e-5)
    m.calls = []
    return m


def _s8_pockets():
    out = []
    for g in synth.synth_complexes(S8_N_REC, [1] * len(S8_N_REC), 20, util.CUTOFFS_ALL_ATOM, seed=3):
        g.remove_nodes(g.nodes('lig'), ntype='lig')
        out.append(g)
    return out


def _s8_rank_body(rank):
    torch.manual_seed(500 + rank)                       # ranks disagree on their generators; the job seed is rank 0's draw
    m = _stub_model()
    full = m._sample(_s8_pockets(), S8_N_LIG, rec_enc_batch_size=2, diff_batch_size=3)
    calls_full = m.calls
    m.calls = []
    few = m._sample(_s8_pockets(), S8_FEW, rec_enc_batch_size=2, diff_batch_size=3)
    return full, calls_full, few, m.calls


def _check_s8(results):
    flat = [(i, n) for i, sizes in enumerate(S8_N_LIG) for n in sizes]
    seen = sorted(c for r in results for ids, _ in r[1] for c in ids)
    assert seen == list(range(len(flat)))                                             # every complex sampled exactly once
    for full, calls, few, calls_few in results:
        for ids, n_kp in calls:
            assert ids == list(range(ids[0], ids[0] + len(ids))) and len(ids) <= 3    # contiguous shard, diff_batch_size honoured
            assert n_kp == [S8_N_REC[flat[c][0]] for c in ids]                        # the right pocket under every complex
        assert len(full) == len(S8_N_LIG)
        c = 0
        tags = set()
        for i, sizes in enumerate(S8_N_LIG):                                          # regrouped per pocket, input order
            assert len(full[i]['positions']) == len(sizes)
            for n, p, f in zip(sizes, full[i]['positions'], full[i]['features']):
                assert p.shape == (n, 3) and f.shape == (n, 10) and bool((p == float(c)).all())
                tags.add(float(f[0, 0]) - c)
                c += 1
        assert len(tags) == 1                                                          # one noise seed for the whole job
        assert [len(s['positions']) for s in few] == [1, 0, 2, 0, 0, 0]                # more ranks than complexes: empty shards
        assert [tuple(p.shape) for s in few for p in s['positions']] == [(5, 3), (7, 3), (4, 3)]
    assert sorted(c for r in results for ids, _ in r[3] for c in ids) == [0, 1, 2]
    assert len({float(r[0][0]['features'][0][0, 0]) for r in results}) == 1           # all ranks hold the same result


def _s8_proc(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    q.put(_plain((rank, _s8_rank_body(rank))))
    dist.barrier()
    dist.destroy_process_group()


def test_sample_shards_over_world8_gloo_processes():
    """`KeypointDiffusion._sample` with EIGHT rank processes over gloo (the shape of the configs[3] / configs[4] jobs)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    world = 8
    procs = [ctx.Process(target=_s8_proc, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_torchify(q.get(timeout=300)) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _check_s8([r[1] for r in res])


def test_sample_shards_over_world8_thread_ranks():
    """The same job with eight THREAD ranks (`util.run_threaded_world`) -- the harness the GPU rehearsal uses, where a box
    admits at most six GPU-holding processes; checked here against the same expectations as the process ranks."""
    _check_s8(util.run_threaded_world(8, _s8_rank_body))
