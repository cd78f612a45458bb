"""Which rows of which tiles deviate in the first forward of the f16x2 edge kernel, and in which quantity (diagnostic; library
built with -DKPD_EDGE_DBG).  Per row of the coordinate branch the kernel taps: the head dot, the distance it read from LDS,
T[row][256] and T[row][7]."""
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
eng.reserve(g.prepared())
eng.debug('edge_dbg=1')
T = 6193
outs = []
with torch.no_grad():
    for i in range(4):
        model.dynamics(g, t, None)
        outs.append(eng.debug('edge_dbg', T * 64 * 4).view(T, 64, 4).clone())
names = ['dot', 'd', 'T256', 'T7']
for i in (0, 2, 3):
    d = outs[i] != outs[1]
    if not int(d.sum()):
        print(f'run {i} == run 1')
        continue
    tiles = d.any(2).any(1).nonzero().flatten().tolist()
    print(f'run {i} vs run 1: {int(d.sum())} tap values differ in tiles {tiles[:10]}')
    for tl in tiles[:4]:
        rows = d[tl].any(1).nonzero().flatten().tolist()
        print(f'  tile {tl}: rows {rows}')
        for r in rows[:6]:
            print('     row', r, {n: (float(outs[i][tl, r, k]), float(outs[1][tl, r, k])) for k, n in enumerate(names) if d[tl, r, k]})
