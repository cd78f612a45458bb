"""Per-phase s_memtime ticks of k_node_update8 (and of k_egnn_edge beside it) on the contract workload: python profiles/tools/node_phases.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

EDGE = ['geometry', 'A-build e', 'GEMM e', 'T-store e', 'att dot', 'reduce h', 'A-build c', 'GEMM c', 'T-store c', 'coord dot', 'reduce x']
NODE = ['x update + load h', 'GEMM 1a + row dot', 'h_neigh gather', 'GEMM 1b + row dot', 'SiLU -> T', 'GEMM 2 + row dot', 'bias + residual', 'LayerNorm stats', 'normalise + store']

dev = torch.device('cuda:0')
model = bench.build_model(dev)
g = bench.build_batch(model, 64, 300, 25, 1234, dev)
eng = model.dynamics.engine()
t = torch.full((64,), 0.9, device=dev)
with torch.no_grad():
    for _ in range(3):
        model.dynamics(g, t, None)
    torch.cuda.synchronize()
    eng.debug('stamps=1')
    n_it = 5
    for _ in range(n_it):
        model.dynamics(g, t, None)
    torch.cuda.synchronize()
    raw = eng.debug('stamps', 64).view(torch.int32).view(-1).view(torch.int64).cpu().tolist()
c = eng.last_counts()
tiles = (5 * c['tiles'] + c['tiles_last']) * n_it
print(f'k_egnn_edge<4>: ticks per 64-edge tile ({tiles} tiles)')
for nm, v in zip(EDGE, raw[:11]):
    print(f'  {nm:22s} {v / tiles:10.0f}')
print(f'  {"total":22s} {sum(raw[:11]) / tiles:10.0f}')
tot = sum(raw[16:25])
print('k_node_update8: share of a 32-node tile\'s residency')
for nm, v in zip(NODE, raw[16:25]):
    print(f'  {nm:22s} {100.0 * v / max(tot, 1):6.1f} %   {v:14d}')
