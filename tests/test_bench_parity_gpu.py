"""bench.py checks the path it measures against the oracle on the CPU-baseline sample (`product_vs_oracle`) and refuses the line when they
disagree: a kernel can be wrong at an unchanged speed (round 5: a missed hardware hazard in a hand-scheduled SiLU)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('workload', ['egnn_all_atom', 'gvp_all_atom'])
def test_baseline_leg_accepts_the_product_and_refuses_a_wrong_one(cuda, workload):
    from oracle import egnn as oegnn
    from oracle import gvp as ogvp
    w = bench.WORKLOADS[workload]
    model = bench.build_model('cpu', workload)
    sd = {k[len('dynamics.'):]: v.detach().cpu().clone() for k, v in model.state_dict().items() if k.startswith('dynamics.')}
    cfg = dict(w['dyn'], graph_cutoffs=w['cutoffs'])
    fwd = oegnn.egnn_dynamics_forward if w['arch'] == 'egnn' else ogvp.gvp_dynamics_forward
    g = bench.build_batch(model, 2, [90, 60], [12, 7], seed=3, device='cpu', workload=workload)
    rec = bench.product_vs_oracle(model, fwd, sd, cfg, g)
    assert rec['ok'] and rec['rel_err_h'] < bench.PARITY_TOL and rec['rel_err_x'] < bench.PARITY_TOL
    # the same check against an oracle holding other weights = a product that computes something else
    wrong = {k: (v * 1.05 if v.dtype.is_floating_point and v.dim() == 2 else v) for k, v in sd.items()}
    with pytest.raises(bench.BenchRefused):
        bench.product_vs_oracle(model, fwd, wrong, cfg, g)
