"""Run-to-run bit equality of the EGNN denoiser at the contract shape (diagnostic): prints how many output elements differ
between repeated forwards.  KPD_GEMM / KPD_H_PARTS select the kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device('cuda:0')
model = bench.build_model(dev)
g = bench.build_batch(model, 64, 300, 25, 1234, dev)
t = torch.linspace(0.05, 1.0, 64, device=dev)
with torch.no_grad():
    h0, x0 = model.dynamics(g, t, None)
    h0, x0 = h0.clone(), x0.clone()
    n = int(os.environ.get('KPD_REPRO_RUNS', '40'))
    outs = []
    for i in range(n):
        h, x = model.dynamics(g, t, None)
        outs.append((h.clone(), x.clone()))
    # majority output = the reference; count the runs (and complexes) that deviate from it
    keys = [hash((o[0].cpu().numpy().tobytes(), o[1].cpu().numpy().tobytes())) for o in outs]
    ref = max(set(keys), key=keys.count)
    hr, xr = outs[keys.index(ref)]
    bad = [i for i, k in enumerate(keys) if k != ref]
    worst = max([float((outs[i][0] - hr).abs().max()) for i in bad], default=0.0)
    rows = sorted({int(r) // 25 for i in bad for r in (outs[i][0] != hr).any(1).nonzero().flatten().tolist()})
    print(f'{n} forwards: {len(bad)} deviate from the majority output ({len(set(keys))} distinct outputs); max |d eps_h| {worst:.3e}; complexes touched: {rows[:20]}')
