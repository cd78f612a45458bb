"""RCCL smoke of the two collectives on a one-GPU box: a single-rank NCCL group exercises the device-tensor paths
(all_gather_into_tensor of the ligand block, bucketed gradient all-reduce) that the CPU tests cover over gloo."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from keypoint_diffusion_amd import graph as G, synth
from keypoint_diffusion_amd.dist import all_gather_ligands, allreduce_gradients
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29571')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
gs = synth.synth_complexes([20, 35, 12], [5, 9, 3], 4, {'kk': 8, 'kl': 6, 'll': 6, 'rk': 100, 'rr': 3.5}, seed=40)
g = G.batch(gs).to('cuda')
pos, feat = all_gather_ligands(g)
assert [p.shape[0] for p in pos] == [5, 9, 3] and torch.allclose(pos[1], gs[1].nodes['lig'].data['x_0'])
ps = [torch.nn.Parameter(torch.zeros(257, 515, device='cuda')), torch.nn.Parameter(torch.zeros(257, device='cuda'))]
for p in ps:
    p.grad = torch.randn_like(p)
before = [p.grad.clone() for p in ps]
n = allreduce_gradients(ps)          # world 1: returns 0 buckets, grads untouched
assert n == 0 and all(torch.equal(a, p.grad) for a, p in zip(before, ps))
flat = torch.cat([p.grad.reshape(-1) for p in ps])
dist.all_reduce(flat)                # the collective itself on a device tensor
torch.cuda.synchronize()
print('nccl single-rank ok')
dist.destroy_process_group()
