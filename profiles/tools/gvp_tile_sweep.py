"""Launch time of k_gvp_chain against the number of 64-edge tiles in the launch (gvp_40kp shapes: 2 720 edges per complex in a conv over
all four edge types): how long does a partly filled LAST round of workgroup slots last?  512 slots (2 workgroups x 256 CUs); B complexes
give 42.5 B tiles.  Times: the library's HIP events around the kernel (kpd_gvp_profile), `convs=5` = the five full convs only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device('cuda:0')
wl = 'gvp_40kp'
model = bench.build_model(dev, wl)
rows = []
for B in (3, 6, 9, 12, 15, 18, 21, 24, 30, 36, 42, 48, 54, 60, 64, 66, 72, 78, 84, 96, 128, 192):
    g = bench.build_batch(model, B, 300, 25, 1234, dev, workload=wl)
    eng = model.dynamics.engine()
    t = torch.full((B,), 0.9, device=dev)
    with torch.no_grad():
        eng.debug('convs=5')
        for _ in range(3):
            model.dynamics(g, t, None)
        torch.cuda.synchronize()
        eng.profile(True)
        for _ in range(8):
            model.dynamics(g, t, None)
        ms, n = eng.profile_read()
        eng.profile(False)
        c = eng.last_counts()
        eng.debug('convs=-1')
    tiles = c['tiles']
    rows.append((B, tiles, ms / n * 1e3))
    print(f'B={B:4d} tiles={tiles:6d} rounds={tiles / 512:6.2f}  k_gvp_chain {ms / n * 1e3:8.1f} us  per tile {ms / n * 1e3 / tiles * 1e3:7.1f} ns', flush=True)
    del g
# model: t = R * floor(rounds) + tail(frac)
full = [r for r in rows if r[1] % 512 == 0]
print('tiles/us by size:', [(r[1], round(r[1] / r[2], 2)) for r in rows])
