import sys, torch
sys.path.insert(0, '/root/repo')
from tests import test_recenc_train_gpu as T
from keypoint_diffusion_amd import graph as G, synth
cuda = torch.device('cuda:0')
K = 8
model, cut = T._kd_model(cuda, K)
model.eval()
def run(grad):
    torch.manual_seed(77)
    with torch.enable_grad() if grad else torch.no_grad():
        out = model(G.batch(synth.synth_complexes([60, 45, 52], [9, 13, 7], K, cut, seed=11)).to(cuda), None)
    return out['l2']
n = 'rec_encoder.scalar_embed.2.weight'
run(True).backward()
p = dict(model.named_parameters())[n]
g = p.grad.clone()
print('repeat evals', [float(run(False).double()) for _ in range(3)])
gen = torch.Generator().manual_seed(3)
for k in range(5):
    d = torch.randn(p.shape, generator=gen).to(cuda)
    if k == 4: d = torch.zeros_like(d)
    a = float((g.double() * d.double()).sum())
    out = []
    for eps in (1e-3, 1e-4):
        v = []
        with torch.no_grad():
            for sign in (1.0, -1.0):
                p.add_(sign * eps * d); v.append(float(run(False).double())); p.sub_(sign * eps * d)
        out.append((v[0] - v[1]) / (2 * eps))
    print(f'dir {k}: analytic {a:+.5e} numeric {out[0]:+.5e} {out[1]:+.5e}  f+ {v[0]:.8f} f- {v[1]:.8f}')
