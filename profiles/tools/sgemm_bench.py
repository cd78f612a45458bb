"""TFLOP/s of the library's fp32 GEMM (csrc/sgemm.hip) on the shapes the training engines call it with, torch.matmul (vendor BLAS)
beside it.  Run on the GPU box: python profiles/tools/sgemm_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoint_diffusion_amd import hip

dev = torch.device('cuda:0')
ws = torch.zeros(260 * (256 * 256 + 1024), device=dev)      # as the trainers' split-K scratch (GRAD_PART_FLOATS): the edge-sized TN products need it for their full slice count
SHAPES = [  # name, tA, tB, M, N, K
    ('edge fwd NT', False, True, 166000, 256, 256), ('edge bwd NN', False, False, 166000, 256, 256),
    ('node fwd NT', False, True, 20800, 256, 256), ('node bwd NN', False, False, 20800, 256, 512),
    ('grad TN kk', True, False, 256, 256, 166000), ('grad TN kl', True, False, 256, 256, 96000), ('grad TN ll', True, False, 256, 256, 38000),
    ('grad TN node', True, False, 256, 512, 20800), ('grad TN wide', True, False, 256, 513, 96000),
    ('gvp vec NN', False, False, 1200000, 16, 16), ('gvp vec NT', False, True, 1200000, 16, 16), ('gvp grad vec', True, False, 16, 16, 1200000),
    ('gvp s NT', False, True, 400000, 256, 288), ('gvp gate NT', False, True, 400000, 16, 256), ('gvp grad s', True, False, 256, 288, 400000),
    ('rec NT 128', False, True, 167000, 128, 128), ('kp B', False, True, 64, 5120, 128),
]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for name, tA, tB, M, N, K in SHAPES:
    a = torch.randn((K, M) if tA else (M, K), device=dev)
    b = torch.randn((N, K) if tB else (K, N), device=dev)
    out = torch.zeros(M, N, device=dev)
    t_own = timeit(lambda: hip.sgemm(a, b, tA, tB, out=out, workspace=ws))
    A_, B_ = (a.T if tA else a), (b.T if tB else b)
    t_lib = timeit(lambda: torch.matmul(A_, B_, out=out))
    fl = 2.0 * M * N * K
    print(f'{name:14s} M={M:8d} N={N:4d} K={K:8d}  own {t_own * 1e6:8.1f} us {fl / t_own / 1e12:6.1f} TF   vendor {t_lib * 1e6:8.1f} us {fl / t_lib / 1e12:6.1f} TF', flush=True)
