"""C ABI: the shared library loads without a GPU and exports exactly what include/kpd.h declares."""
import ctypes
import os
import re

from keypoint_diffusion_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'kpd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(kpd_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(hip.LIB_PATH), 'build the HIP extension first (__graft_entry__.build())'
    lib = ctypes.CDLL(hip.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/kpd.h but not exported'
    assert sorted(hip.EXPORTS) == syms, 'hip.EXPORTS out of sync with include/kpd.h'


def test_version_and_error_string_without_gpu():
    lib = hip.lib()
    assert lib.kpd_version() >= 100
    assert isinstance(lib.kpd_last_error(), bytes)


def test_product_build_has_no_ablation_or_ab_switches():
    """The shipped library is the product build: kpd_build_flags() == 0 and the names of the A/B, ablation and LDS-padding switches of
    profiles/tools (read from the environment only by the TOOLS build, `make tools`) are not even in the binary.  What the product does
    read: KPD_GEMM (the opt-in f16x2 mode), KPD_TRAIN_STORE (recompute instead of keeping activations), KPD_POISON (NaN-poisoned
    workspaces for the read-before-write tests) -- bench.py refuses to run with any of them set."""
    assert hip.lib().kpd_build_flags() == 0
    blob = open(hip.LIB_PATH, 'rb').read()
    names = set(re.findall(rb'KPD_[A-Z0-9_]{3,}', blob))
    assert names <= {b'KPD_GEMM', b'KPD_TRAIN_STORE', b'KPD_POISON'}, sorted(names)
    for gone in (b'KPD_EDGE_ABLATE', b'KPD_EDGE_SPLIT', b'KPD_EDGE_LDS_PAD', b'KPD_NODE_LDS_PAD', b'KPD_SGEMM_', b'KPD_TRAIN_FUSED', b'KPD_H_PARTS'):
        assert gone not in blob, gone


def test_product_has_no_cpu_fallback():
    """The product path must fail loudly off-GPU rather than compute on the CPU."""
    import pytest
    import torch
    from keypoint_diffusion_amd.dynamics import LigRecDynamics
    from . import util
    g = util.fixed_encode(util.make_batch([20], [5]))
    m = LigRecDynamics(10, 10, graph_cutoffs=util.CUTOFFS_ALL_ATOM, **util.EGNN_C2).eval()
    if torch.cuda.is_available():
        pytest.skip('GPU present: covered by the gpu tests')
    with torch.no_grad(), pytest.raises((hip.KpdError, RuntimeError, AssertionError)):
        m(g, torch.tensor([0.5]), None)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'keypoint-diffusion_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f
