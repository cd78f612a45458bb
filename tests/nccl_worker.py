"""The RCCL branch of keypoint_diffusion_amd.dist on a one-GPU box: `python -m tests.nccl_worker OUT.pt` starts a ONE-rank
`nccl` process group on cuda:0 in a fresh process and drives every collective the 8-GPU jobs issue through the device-tensor
branch (`_group_device` -> device tensors, `all_gather_into_tensor`, the bucketed `all_reduce`, `broadcast_object_list`):
`gather_ligand_lists`, `all_gather_ligands`, `allreduce_gradients`, `common_seed` and `KeypointDiffusion._sample`
(models/ligand_diffuser.py:292-313 + SURVEY.md 8(e)).  tests/test_nccl_gpu.py compares what it saves with a run without
any process group."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    from keypoint_diffusion_amd import dist as D
    from keypoint_diffusion_amd import graph as G
    from keypoint_diffusion_amd import synth
    from tests import sharded_worker as W
    from tests import util
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29581')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == 'nccl' and D.sharding_active()
    assert D._group_device(dev) == dev                      # the branch under test: collectives move DEVICE tensors
    report = {}

    # 1) the end-of-sampling exchange on device tensors: ragged ligands, rank-major order, host results
    n_rec, n_lig = [20, 35, 12, 50], [5, 9, 3, 17]
    gs = synth.synth_complexes(n_rec, n_lig, 4, util.CUTOFFS_ALL_ATOM, seed=40)
    g = G.batch(gs).to(dev)
    pos, feat = D.all_gather_ligands(g)
    assert [tuple(p.shape) for p in pos] == [(n, 3) for n in n_lig] and all(p.device.type == 'cpu' for p in pos)
    for i in range(len(gs)):
        assert torch.equal(pos[i], gs[i].nodes['lig'].data['x_0']) and torch.equal(feat[i], gs[i].nodes['lig'].data['h_0'])
    p2, f2 = D.gather_ligand_lists([p.to(dev) for p in pos], [f.to(dev) for f in feat])
    assert all(torch.equal(a, b) for a, b in zip(p2, pos)) and all(torch.equal(a, b) for a, b in zip(f2, feat))
    assert D.gather_ligand_lists([], [], device=dev) == ([], [])         # a rank without complexes
    report['gathered'] = len(pos)

    # 2) bucketed gradient all-reduce on device tensors (two buckets, one frozen parameter)
    params = [torch.nn.Parameter(torch.zeros(s, device=dev)) for s in ((257, 515), (257,), (1, 257), (3, 5))]
    gen = torch.Generator(device=dev).manual_seed(7)
    for i, p in enumerate(params):
        p.grad = None if i == 3 else torch.randn(p.shape, generator=gen, device=dev)
    before = [None if p.grad is None else p.grad.clone() for p in params]
    n_buckets = D.allreduce_gradients(params, bucket_bytes=300 * 1024)
    assert n_buckets == 2 and params[3].grad is None
    assert all(torch.equal(a, p.grad) for a, p in zip(before[:3], params[:3]))      # mean over one rank = itself, bit for bit
    assert D.allreduce_gradients(params, average=False) == 1
    assert all(torch.equal(a, p.grad) for a, p in zip(before[:3], params[:3]))
    report['buckets'] = n_buckets

    # 3) the seed every rank agrees on (object broadcast over RCCL)
    torch.manual_seed(321)
    report['seed'] = D.common_seed()

    # 4) the sharded sampler end to end under the RCCL group
    report['samples'] = W.run_rank(dev)
    # ... and once more with the seed drawn through the group (no use_complex_noise by the caller)
    model = W.build_model(dev)
    torch.manual_seed(55)
    report['samples_common_seed'] = model._sample(W.pockets(dev), W.N_LIG, rec_enc_batch_size=2, diff_batch_size=2)
    assert getattr(model, '_noise_seed', None) is None              # the caller's noise mode is restored

    # 5) ranks_seen as bench.py reports it
    ones = torch.ones(1, device=dev, dtype=torch.float64)
    dist.all_reduce(ones)
    report['ranks_seen'] = int(ones.item())
    torch.cuda.synchronize()
    torch.save(report, sys.argv[1])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
