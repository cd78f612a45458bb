"""Wall time of the phases of one training step of a keypoint model (synchronised between phases, so host and GPU time of a phase add up):
python profiles/tools/train_phases.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from keypoint_diffusion_amd import graph as G
from keypoint_diffusion_amd import hip as _hip

_solve = _hip.ot_emd_uniform
_solve_ms = []


def _timed_solve(costs, n_threads=0):
    t0 = time.perf_counter()
    out = _solve(costs, n_threads)
    _solve_ms.append(1e3 * (time.perf_counter() - t0))
    return out


_hip.ot_emd_uniform = _timed_solve

wl = sys.argv[1] if len(sys.argv) > 1 else 'gvp_40kp_train'
dev = torch.device('cuda:0')
model = bench.build_model(dev, wl).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
template = bench.raw_batch(64, 300, 25, 1234, dev, wl).to(dev)


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


acc = {}
for it in range(6):
    g = template.to(dev)
    t0 = sync()
    g = model.normalize(g)
    bi = G.get_batch_idxs(g)
    g = model.rec_encoder(g, bi)
    t1 = sync()
    pend = model.rec_encoder_loss_fn.begin(g, interface_points=None)
    t2 = sync()
    g = model.remove_com(g, bi['lig'], bi['kp'], com='ligand')
    B = g.batch_size
    t = torch.randint(0, model.n_timesteps, size=(B,), device=dev).float() / model.n_timesteps
    eps = {'h': torch.randn(g.nodes['lig'].data['h_0'].shape, device=dev), 'x': torch.randn(g.nodes['lig'].data['x_0'].shape, device=dev)}
    g = model.noised_representation(g, bi['lig'], bi['kp'], eps, model.gamma(t).to(dev))
    eh, ex = model.dynamics(g, t, bi)
    t3 = sync()
    le = pend.finish()
    t4 = sync()
    loss = ((eps['x'] - ex).square().sum() + (eps['h'] - eh).square().sum()) / (eps['x'].numel() + eps['h'].numel()) + 0.1 * le
    opt.zero_grad(set_to_none=True)
    loss.backward()
    t5 = sync()
    torch.nn.utils.clip_grad_value_(model.parameters(), 1.0)
    opt.step()
    t6 = sync()
    if it >= 2:
        for k, v in (('encoder fwd', t1 - t0), ('loss begin (costs + D2H + thread start)', t2 - t1), ('noise + denoiser fwd', t3 - t2),
                     ('loss finish (join + plans H2D)', t4 - t3), ('backward', t5 - t4), ('clip + Adam', t6 - t5), ('step', t6 - t0)):
            acc[k] = acc.get(k, 0.0) + v / 4
for k, v in acc.items():
    print(f'{k:45s} {1e3 * v:8.2f} ms')
print('transport solve on its thread (ms per step):', [round(x, 1) for x in _solve_ms])
