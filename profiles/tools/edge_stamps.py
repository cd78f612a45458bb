"""Per-phase cycle shares of k_egnn_edge (diagnostic run; stamps perturb the timing slightly)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from keypoint_diffusion_amd import graph as G

dev = torch.device('cuda:0')
model = bench.build_model(dev)
g = bench.build_batch(model, 64, 300, 25, 1234, dev)
eng = model.dynamics.engine()
t = torch.full((64,), 0.9, device=dev)
names = ['geometry', 'A-build e', 'GEMM e', 'T-store e', 'att dot', 'reduce h', 'A-build c', 'GEMM c', 'T-store c', 'coord dot', 'reduce x']
with torch.no_grad():
    for _ in range(2):
        model.dynamics(g, t, None)
    eng.debug('stamps=1')
    n = 5
    for _ in range(n):
        model.dynamics(g, t, None)
    torch.cuda.synchronize()
    raw = eng.debug('stamps', 64).view(torch.int32).view(-1).view(torch.int64).cpu().tolist()
    vals, nvals = raw[:11], raw[16:27]
c = eng.last_counts()
tiles = c['tiles'] * 6 * n
tot = sum(vals)
print('tiles', c['tiles'], 'per-tile cycles (s_memtime ticks = 100 MHz? see note):')
for nm, v in zip(names, vals):
    print(f'  {nm:36s} {v / tiles:10.1f}  {100 * v / tot:5.1f} %')
print('  total        %10.1f' % (tot / tiles))

nn = ['load h/x', 'GEMM 1a', 'gather hn', 'GEMM 1b', 'T-store', 'GEMM 2', 'resid', 'LN', 'writeback', 'proj GEMMs', 'proj store']
ntiles = (20800 // 32 + 1) * 5 * n          # fused launches with update+proj in the timed forwards (approx.)
tot = sum(nvals)
print('node_layer per-tile cycles (approx. per fused launch tile):')
for nm, v in zip(nn, nvals):
    print(f'  {nm:12s} {v / ntiles:10.1f}  {100 * v / max(tot,1):5.1f} %')
