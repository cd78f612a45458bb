"""Per-phase cycles of k_egnn_edge with two co-resident workgroups per CU (production) and with one
(KPD_EDGE_LDS_PAD forces one workgroup per CU): how much of the non-GEMM time hides behind the partner's MFMAs."""
import os, subprocess, sys
# (the KPD_* switches of this tool exist only in the TOOLS build: `make -C keypoint-diffusion_amd/csrc tools`)
os.environ.setdefault('KPD_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'keypoint-diffusion_amd', 'csrc', 'tools_build', 'libkpd_hip.so'))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

NAMES = ['geometry', 'A-build e', 'GEMM e', 'T-store e', 'att dot', 'reduce h', 'A-build c', 'GEMM c', 'T-store c', 'coord dot', 'reduce x']


def child():
    import torch
    import bench
    dev = torch.device('cuda:0')
    model = bench.build_model(dev)
    g = bench.build_batch(model, 64, 300, 25, 1234, dev)
    eng = model.dynamics.engine()
    t = torch.full((64,), 0.9, device=dev)
    with torch.no_grad():
        for _ in range(3):
            model.dynamics(g, t, None)
        torch.cuda.synchronize()
        eng.profile(True)
        for _ in range(10):
            model.dynamics(g, t, None)
        ms, n = eng.profile_read()
        eng.profile(False)
        eng.debug('stamps=1')
        n_it = 5
        for _ in range(n_it):
            model.dynamics(g, t, None)
        torch.cuda.synchronize()
        raw = eng.debug('stamps', 64).view(torch.int32).view(-1).view(torch.int64).cpu().tolist()
    c = eng.last_counts()
    tiles = (5 * c['tiles'] + c['tiles_last']) * n_it
    vals = raw[:11]
    print(f'pad={os.environ.get("KPD_EDGE_LDS_PAD", "0")}: edge kernel avg {ms / n:.4f} ms over {n} launches (5 full + 1 pruned per forward)')
    for nm, v in zip(NAMES, vals):
        print(f'  {nm:12s} {v / tiles:10.0f}')
    print(f'  {"total":12s} {sum(vals) / tiles:10.0f}   GEMM {(vals[2] + vals[7]) / tiles:.0f}  non-GEMM {(sum(vals) - vals[2] - vals[7]) / tiles:.0f}')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child()
    else:
        for pad in ('0', '24000'):
            subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, KPD_EDGE_LDS_PAD=pad), check=True)
