"""Backward pass of the GVP keypoint receptor encoder (SURVEY.md 8(f) item 2 for row a8): gradients of every parameter from
kpd_recenc_trainer_* against torch autograd through the CPU oracle (`oracle/rec_encoder.py`), for a loss that reaches the encoder
through all three of its outputs -- keypoint positions, scalars and vectors -- as the denoiser and the encoder loss do."""
import pytest
import torch

from keypoint_diffusion_amd import graph as G
from keypoint_diffusion_amd import synth
from keypoint_diffusion_amd.receptor_encoder_gvp import ReceptorEncoderGVP
from oracle import rec_encoder as orec

from . import util
from .golden.make_golden_cfgs import RECENC_CFGS
from .test_recenc_gpu import RECENC_40KP, RECENC_NORM0, RECENC_RAD

pytestmark = pytest.mark.gpu
CUT = util.CUTOFFS_ALL_ATOM
TOL = 2e-4          # relative to the largest entry of each gradient tensor


def _weights(n_kp, S, seed):
    gen = torch.Generator().manual_seed(seed)
    return (torch.randn(n_kp, 3, generator=gen), torch.randn(n_kp, S, generator=gen) / S ** 0.5, torch.randn(n_kp, 16, 3, generator=gen) / 4)


@pytest.mark.parametrize('cfg,n_rec', [(RECENC_CFGS['recenc_mean'], [33, 21]), (RECENC_CFGS['recenc_norm10'], [33, 21]),
                                       (RECENC_NORM0, [50, 3, 27]), (RECENC_40KP, [150, 90]), (RECENC_RAD, [120, 45])])
def test_encoder_gradients_match_oracle_autograd(cuda, cfg, n_rec):
    cfg = dict(cfg, dropout=0.0)
    kw = dict(cfg, graph_cutoffs=CUT)
    model = synth.fill_state_dict_(ReceptorEncoderGVP(**kw), 61).eval()
    g = util.make_batch(n_rec, [4] * len(n_rec), seed=17, n_keypoints=cfg['n_keypoints'])
    n_kp, S = len(n_rec) * cfg['n_keypoints'], cfg['out_scalar_size']
    w_x, w_h, w_v = _weights(n_kp, S, 5)
    # oracle, differentiated by torch autograd
    sd = {k: v.detach().clone().requires_grad_(v.numel() > 0) for k, v in model.state_dict().items()}
    ref = orec.rec_encoder_gvp_forward(sd, kw, util.to_obatch(g))
    ((ref.x['kp'] * w_x).sum() + (ref.h['kp'] * w_h).sum() + (ref.v['kp'] * w_v).sum()).backward()
    # product, differentiated by the HIP backward pass
    model = model.to(cuda)
    gd = g.to(cuda)
    out = model(gd, G.get_batch_idxs(gd))
    kp = out.nodes['kp'].data
    assert kp['x_0'].requires_grad and kp['h_0'].requires_grad and kp['v_0'].requires_grad
    assert util.rel_err(kp['x_0'].detach(), ref.x['kp'].detach()) < 1e-4 and util.rel_err(kp['h_0'].detach(), ref.h['kp'].detach()) < 1e-4
    assert util.rel_err(kp['v_0'].detach(), ref.v['kp'].detach()) < 1e-4
    rs, rd = out.edges(etype='rk')
    assert torch.equal(rs.cpu(), ref.edges['rk'][0]) and torch.equal(rd.cpu(), ref.edges['rk'][1])
    ((kp['x_0'] * w_x.to(cuda)).sum() + (kp['h_0'] * w_h.to(cuda)).sum() + (kp['v_0'] * w_v.to(cuda)).sum()).backward()
    worst = []
    for n, p in model.named_parameters():
        if p.numel() == 0:
            continue
        r = sd[n].grad
        if r is None or float(r.abs().max()) == 0.0:                 # e.g. keypoint_initializer.norm: built, never used (:36)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        assert p.grad is not None, n
        worst.append(((p.grad.cpu() - r).abs().max().item() / r.abs().max().item(), n))
    worst.sort(reverse=True)
    assert len(worst) > 20 and worst[0][0] < TOL, worst[:8]


def test_encoder_training_step_is_reproducible_and_dropout_runs(cuda):
    """Two forward / backward passes of one batch give bit-identical gradients (no float atomics); with GVPDropout the same seed
    reproduces a step and another seed changes it; train-mode dropout without autograd is refused."""
    cfg = dict(RECENC_CFGS['recenc_mean'])
    kw = dict(cfg, graph_cutoffs=CUT)
    model = synth.fill_state_dict_(ReceptorEncoderGVP(**kw), 61).to(cuda).train()
    g = util.make_batch([40, 25], [4, 4], seed=17, n_keypoints=cfg['n_keypoints'])

    def step(seed):
        torch.manual_seed(seed)
        model.zero_grad(set_to_none=True)
        gd = g.to(cuda)
        kp = model(gd, G.get_batch_idxs(gd)).nodes['kp'].data
        (kp['x_0'].square().sum() + kp['h_0'].square().sum() + kp['v_0'].square().sum()).backward()
        return [p.grad.clone() for p in model.parameters() if p.grad is not None]

    a, b, c = step(1), step(1), step(2)
    assert len(a) > 20 and all(torch.equal(x, y) for x, y in zip(a, b))
    assert any(not torch.equal(x, y) for x, y in zip(a, c))
    with pytest.raises(NotImplementedError), torch.no_grad():
        model(g.to(cuda), None)
