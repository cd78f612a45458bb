"""End-to-end soak: full T-step reverse loops at the BASELINE shapes (random-init weights: ligands are garbage, the point is
500-1000 consecutive steps without faults, NaNs or workspace growth) and wall-clock ligands/min including encoding."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from keypoint_diffusion_amd import graph as G

dev = torch.device('cuda:0')
for wl, T in (('egnn_all_atom', 500), ('gvp_40kp', 500), ('egnn_40kp', 500)):
    model = bench.build_model(dev, wl)
    g = bench.build_batch(model, 64, 300, 25, 1234, dev, workload=wl)
    torch.cuda.synchronize()
    mem0 = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    with torch.no_grad():
        pos, feat = model.sample_from_encoded_receptors(g)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(torch.isfinite(p).all() for p in pos) and all(torch.isfinite(f).all() for f in feat)
    print(f'{wl}: {T} steps x 64 complexes in {dt:.2f} s = {64 * 60 / dt:.0f} ligands/min, {T / dt:.1f} steps/s, finite={ok}, '
          f'torch allocator delta {(torch.cuda.memory_allocated() - mem0) / 1e6:.1f} MB', flush=True)
