"""First forward vs later forwards after ONE EGNN layer: which segment-sum pieces (main / cont, per edge type) differ (diagnostic)."""
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
eng.debug('layers=1')
taps = {}
for et, (nd, ne) in enumerate([(1600, 38400), (1600, 96000), (19200, 96000), (19200, 170000)]):
    taps[f'xnm{et}'] = (nd, 4); taps[f'xnc{et}'] = (ne // 64 + 8, 4); taps[f'hnm{et}'] = (nd, 264); taps[f'hnc{et}'] = (ne // 64 + 8, 264)
outs = []
with torch.no_grad():
    for i in range(4):
        model.dynamics(g, t, None)
        outs.append({k: eng.debug(k, r * c).view(r, c).clone() for k, (r, c) in taps.items()})
print(eng.last_counts())
for k in taps:
    d01 = outs[0][k] != outs[1][k]
    d12 = outs[1][k] != outs[2][k]
    if int(d01.sum()) or int(d12.sum()):
        rows = d01.any(1).nonzero().flatten().tolist()
        print(f'{k}: run0 vs run1 differ in {int(d01.sum())} elements, rows {rows[:16]}; run1 vs run2 in {int(d12.sum())}')
        for r in rows[:4]:
            print('     row', r, 'run0', outs[0][k][r, :4].tolist(), 'run1', outs[1][k][r, :4].tolist())
