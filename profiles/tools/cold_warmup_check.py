"""Does the first-forward deviation of a batched-read build (KPD_H_BATCH_D=true) need a cold DEVICE or a cold KERNEL?
arg 'mm': run 0.3 s of torch matmuls first (clocks and caches busy, the edge kernel never launched);
arg 'tiny': run the same engine type once on a tiny batch first (code objects loaded, kernel launched once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoint_diffusion_amd import graph as G, synth
from keypoint_diffusion_amd.dynamics import LigRecDynamics
from tests import util
dev = torch.device('cuda:0'); cut = util.CUTOFFS_ALL_ATOM; B = 64
mode = sys.argv[1] if len(sys.argv) > 1 else 'none'
gs = synth.synth_complexes([300] * B, [25] * B, 20, cut, seed=5)
g = util.fixed_encode(G.batch(gs)).to(dev)
t = torch.linspace(0.05, 1.0, B, device=dev)
m = synth.fill_state_dict_(LigRecDynamics(10, 10, graph_cutoffs=cut, **util.EGNN_C2), 0).eval().to(dev)
with torch.no_grad():
    if mode == 'mm':
        a = torch.randn(4096, 4096, device=dev)
        for _ in range(60):
            a = (a @ a) * 1e-4
        torch.cuda.synchronize()
    if mode == 'tiny':
        m2 = synth.fill_state_dict_(LigRecDynamics(10, 10, graph_cutoffs=cut, **util.EGNN_C2), 1).eval().to(dev)
        g2 = util.fixed_encode(G.batch(synth.synth_complexes([40], [6], 20, cut, seed=1))).to(dev)
        m2(g2, torch.tensor([0.5], device=dev), None)
        torch.cuda.synchronize()
    keys = []
    for r in range(4):
        h, x = m(g, t, None)
        keys.append(hash((h.cpu().numpy().tobytes(), x.cpu().numpy().tobytes())))
ref = max(set(keys), key=keys.count)
print(f'warm-up={mode}: deviating runs {[i for i, k in enumerate(keys) if k != ref]}')
