"""Which rows / columns of the node state deviate run to run after ONE EGNN layer (diagnostic; KPD_GEMM / KPD_H_PARTS select the kernels)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device('cuda:0')
model = bench.build_model(dev)
g = bench.build_batch(model, 64, 300, 25, 1234, dev)
t = torch.linspace(0.05, 1.0, 64, device=dev)
eng = model.dynamics.engine()
eng.debug('layers=%s' % os.environ.get('KPD_LAYERS', '1'))
n = {'h_lig': 64 * 25, 'h_kp': 64 * 300, 'x_lig': 64 * 25, 'x_kp': 64 * 300}
width = {'h_lig': 264, 'h_kp': 264, 'x_lig': 3, 'x_kp': 3}
outs = []
with torch.no_grad():
    for i in range(int(os.environ.get('KPD_REPRO_RUNS', '40'))):
        model.dynamics(g, t, None)
        outs.append({k: eng.debug(k, n[k] * width[k]).view(n[k], width[k]).clone() for k in n})
for k in n:
    keys = [hash(o[k].cpu().numpy().tobytes()) for o in outs]
    ref = max(set(keys), key=keys.count)
    r = outs[keys.index(ref)][k]
    bad = [i for i, kk in enumerate(keys) if kk != ref]
    print(f'{k}: {len(bad)} of {len(outs)} runs deviate')
    for i in bad[:6]:
        d = outs[i][k] != r
        rows = d.any(1).nonzero().flatten().tolist()
        cols = d.any(0).nonzero().flatten().tolist()
        print(f'   run {i}: {int(d.sum())} elements, rows {rows[:12]}{"..." if len(rows) > 12 else ""} ({len(rows)}), cols {cols[:8]}..{cols[-3:]} ({len(cols)}), '
              f'max |d| {float((outs[i][k] - r).abs().max()):.3e}')
