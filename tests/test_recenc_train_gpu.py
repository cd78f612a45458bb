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


def _weights(n_kp, S, seed, V=16):
    gen = torch.Generator().manual_seed(seed)
    return (torch.randn(n_kp, 3, generator=gen), torch.randn(n_kp, S, generator=gen) / S ** 0.5, torch.randn(n_kp, V, 3, generator=gen) / 4)


@pytest.mark.parametrize('cfg,n_rec', [(RECENC_CFGS['recenc_mean'], [33, 21]), (RECENC_CFGS['recenc_norm10'], [33, 21]),
                                       (RECENC_NORM0, [50, 3, 27]), (RECENC_40KP, [150, 90]), (RECENC_RAD, [120, 45]),
                                       # vector_size < 16 (round 4: trained through zero-padded wide copies, csrc/train_ops.h WideSet)
                                       (dict(RECENC_CFGS['recenc_norm10'], vector_size=8), [33, 21]),
                                       (dict(RECENC_CFGS['recenc_mean'], vector_size=5), [33, 21])])
def test_encoder_gradients_match_oracle_autograd(cuda, cfg, n_rec):
    cfg = dict(cfg, dropout=0.0)
    kw = dict(cfg, graph_cutoffs=CUT)
    model = synth.fill_state_dict_(ReceptorEncoderGVP(**kw), 61).eval()
    g = util.make_batch(n_rec, [4] * len(n_rec), seed=17, n_keypoints=cfg['n_keypoints'])
    n_kp, S = len(n_rec) * cfg['n_keypoints'], cfg['out_scalar_size']
    w_x, w_h, w_v = _weights(n_kp, S, 5, cfg.get('vector_size', 16))
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


def _kd_model(cuda, K=8):
    from keypoint_diffusion_amd.ligand_diffuser import KeypointDiffusion
    from .test_gvp_gpu import GVP_40KP
    rec_cfg = dict(out_scalar_size=128, n_message_gvps=2, n_update_gvps=1, vector_size=16, n_rr_convs=2, n_rk_convs=2, message_norm=10.0,
                   k_closest=4, kp_rad=0, dropout=0.0, in_scalar_size=10)
    cut = dict(CUT, kl=8, ll=6.0)
    m = KeypointDiffusion(10, 128, None, n_timesteps=50, architecture='gvp', rec_encoder_type='learned',
                          graph_config=dict(n_keypoints=K, graph_cutoffs=cut), dynamics_config=dict(GVP_40KP, n_convs=2, dropout=0.0),
                          rec_encoder_config=rec_cfg, rec_encoder_loss_config=dict(loss_type='optimal_transport'), precision=1e-5)
    synth.fill_state_dict_(m, 3)
    return m.to(cuda), cut


def test_keypoint_model_trains_end_to_end(cuda):
    """`KeypointDiffusion.forward` of a gvp_40kp-style model (learned GVP encoder -> keypoints -> GVP denoiser + optimal-transport
    encoder loss, models/ligand_diffuser.py:89-175) under autograd: every encoder and denoiser parameter that the reference
    trains receives a finite gradient, and for encoder AND denoiser weights the gradient along a random direction matches the
    central finite difference of the same loss evaluated with the inference engines (same timestep / noise draws)."""
    K = 8
    model, cut = _kd_model(cuda, K)
    model.eval()                                                       # (dropout off: the finite difference needs a fixed function)
    mk = lambda: G.batch(synth.synth_complexes([60, 45, 52], [9, 13, 7], K, cut, seed=11)).to(cuda)

    def loss(with_grad, w_enc=0.0):
        torch.manual_seed(77)                                          # the same t and eps draws in every evaluation
        with torch.enable_grad() if with_grad else torch.no_grad():
            out = model(mk(), None)
        return out['l2'] + w_enc * out['rec_encoder'], out

    # the optimal-transport term on its own: it reaches the encoder through the keypoint positions only
    enc_loss, parts = loss(True, w_enc=1.0)
    assert all(torch.isfinite(v).all() for v in parts.values()) and float(parts['rec_encoder'].detach()) > 0
    (enc_loss - parts['l2']).backward()
    named = dict(model.named_parameters())
    assert float(named['rec_encoder.keypoint_initializer.dst_net.weight'].grad.abs().max()) > 0
    assert float(named['rec_encoder.rk_conv_layers.1.edge_message.0.Wh'].grad.abs().max()) == 0          # downstream of the positions
    model.zero_grad(set_to_none=True)
    # the denoising loss: finite differences are meaningful for it (the transport plan of the other term is piecewise constant,
    # its value piecewise linear: central differences at any practical step size straddle plan changes)
    total, parts = loss(True)
    total.backward()
    params = dict(model.named_parameters())
    missing = [n for n, p in params.items() if p.numel() and p.requires_grad and p.grad is None and 'keypoint_initializer.norm' not in n]
    assert not missing, missing[:5]
    assert all(torch.isfinite(p.grad).all() for p in params.values() if p.grad is not None)
    pick = ['rec_encoder.rr_conv_layers.1.edge_message.0.to_feats_out.0.weight', 'rec_encoder.rk_conv_layers.1.edge_message.0.Wh',
            'rec_encoder.scalar_embed.2.weight', 'rec_encoder.rk_conv_layers.0.node_update.0.to_feats_out.0.weight',
            'dynamics.noise_predictor.conv_layers.0.edge_message_fns.kp_kl_lig.0.to_feats_out.0.weight']
    gen = torch.Generator().manual_seed(3)
    strong = 0
    for n in pick:            # one tensor at a time; two step sizes: the keypoints move with the encoder weights, and a kNN / radius edge
        d = torch.randn(params[n].shape, generator=gen).to(cuda)        # that flips inside the larger step makes that difference a jump
        analytic = float((params[n].grad.double() * d.double()).sum())
        numeric = []
        for eps in (1e-3, 2e-4):
            vals = []
            with torch.no_grad():
                for sign in (1.0, -1.0):
                    params[n].add_(sign * eps * d)
                    vals.append(float(loss(False)[0].double()))
                    params[n].sub_(sign * eps * d)
            numeric.append((vals[0] - vals[1]) / (2 * eps))
        print(f'{n}: analytic {analytic:+.5e} numeric {numeric[0]:+.5e} {numeric[1]:+.5e}')
        assert min(abs(analytic - v) - 5e-2 * max(abs(analytic), abs(v)) for v in numeric) <= 4e-4, (n, analytic, numeric)
        strong += abs(analytic) > 2e-3
    assert strong >= 3            # the check has teeth: most directions carry a derivative far above the floor


def test_keypoint_model_optimizer_step_lowers_the_loss(cuda):
    """A few Adam steps on one batch of the learned-encoder model in training mode (GVPDropout off here) reduce its loss."""
    model, cut = _kd_model(cuda, 6)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=2e-4)
    g = lambda: G.batch(synth.synth_complexes([50, 40], [8, 11], 6, cut, seed=5)).to(cuda)
    hist = []
    for it in range(6):
        torch.manual_seed(9)
        out = model(g(), None)
        l = out['l2'] + 0.1 * out['rec_encoder']
        opt.zero_grad(set_to_none=True)
        l.backward()
        torch.nn.utils.clip_grad_value_(model.parameters(), 1.0)
        opt.step()
        hist.append(float(l.detach()))
    assert all(h == h for h in hist) and hist[-1] < hist[0], hist
