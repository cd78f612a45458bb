"""Per-phase cycle shares of k_gvp_edge (diagnostic run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device('cuda:0')
wl = sys.argv[1] if len(sys.argv) > 1 else 'gvp_all_atom'
model = bench.build_model(dev, wl)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
g = bench.build_batch(model, B, 300, 25, 1234, dev, workload=wl)
eng = model.dynamics.engine()
t = torch.full((B,), 0.9, device=dev)
with torch.no_grad():
    for _ in range(2):
        model.dynamics(g, t, None)
    eng.debug('stamps=1')
    n = 3
    for _ in range(n):
        model.dynamics(g, t, None)
    torch.cuda.synchronize()
    vals = eng.debug('stamps', 64).view(torch.int32).view(-1).view(torch.int64).cpu().tolist()
if os.environ.get('KPD_GVP_EDGE_STAGED', '0') != '0':
    names = {0: 'geometry+gather', 31: 'segmented sums'}
    for k in range(3):
        for i, nm in enumerate(['W->LDS + vec1', 'GEMM', 'T-store(+gather)', 'gates', 'vec2']):
            names[1 + 5 * k + i] = f'GVP{k} {nm}'
    kernel = 'k_gvp_edge'
else:
    names = dict(enumerate(['gathers + geometry + vec1 (GVP0)', 'GVP0 GEMM chunks', 'GVP0 SiLU + bias preload', 'GVP0 gates',
                            'GVP0 vec2', 'GVP1+ vec1', 'GVP1+ GEMM chunks', 'GVP1+ SiLU + bias preload', 'GVP1+ gates',
                            'GVP1+ vec2', 'messages -> LDS', 'segmented sums']))
    kernel = 'k_gvp_chain'
tot = sum(vals)
print(wl, f'B={B}', kernel, 'phase shares (wave 0 of every workgroup):')
tiles_total = n * sum(eng.last_counts()[k] for k in ('tiles',)) * 5 + n * eng.last_counts()['tiles_last']
print(f'  {tiles_total} tile executions; {tot / tiles_total / 100e6 * 1e6:.1f} us of wave-0 time per tile (100-MHz s_memtime clock)')
for i in sorted(names):
    print(f'  {names[i]:34s} {100 * vals[i] / tot:5.1f} %   {vals[i] / 1e6:10.1f} Mcycles')
