"""Edge-sized weight-gradient products C[257, 257] += A[K, 257]^T B[K, 257] (264-float rows, as the EGNN trainer calls them) through
kpd_sgemm: TFLOP/s of the product + its split-K reduction.  KPD_SGEMM_TN256=0 selects the tiled kernel, default the full-output one."""
import os, sys, time
# (the KPD_* switches of this tool exist only in the TOOLS build: `make -C keypoint-diffusion_amd/csrc tools`)
os.environ.setdefault('KPD_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'keypoint-diffusion_amd', 'csrc', 'tools_build', 'libkpd_hip.so'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoint_diffusion_amd import hip
dev = torch.device('cuda:0')
ws = torch.empty(260 * (256 * 256 + 1024), device=dev)
for M, K in ((257, 38372), (257, 96000), (257, 165934), (256, 165934), (256, 400000)):
    a = torch.randn(K, 264, device=dev)[:, :M]
    b = torch.randn(K, 264, device=dev)[:, :M]
    c = torch.zeros(M, M, device=dev)
    for _ in range(3):
        hip.sgemm(a, b, True, False, beta=1.0, out=c, workspace=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        hip.sgemm(a, b, True, False, beta=1.0, out=c, workspace=ws)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'M=N={M} K={K}: {dt * 1e6:8.1f} us  {2.0 * M * M * K / dt / 1e12:6.1f} TFLOP/s  (TN256={os.environ.get("KPD_SGEMM_TN256", "1")})')
