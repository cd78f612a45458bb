import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    return torch.device('cuda:0')


@pytest.fixture(params=['f32', 'f16x2'])
def gemm_mode(request, monkeypatch):
    """Runs a test once per GEMM mode of the denoiser engines: 'f32' = exact fp32 MFMA (the contract path), 'f16x2' = the opt-in
    split-f16 products (EGNN: edge, projection and node-update GEMMs; GVP: the 256 x 256 products of the message and update chains).
    KPD_GEMM is read when an engine is created, so it must be set before the test builds its model; child processes inherit it.
    The modules that exercise the denoisers request it for every test (`pytestmark`), so the whole parity suite has to hold at
    the same tolerances in both modes."""
    monkeypatch.setenv('KPD_GEMM', request.param)
    return request.param
