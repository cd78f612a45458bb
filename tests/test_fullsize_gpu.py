"""BASELINE.json full-size workloads on the GPU, checked through size-independent properties
(E(3) equivariance, batch independence, determinism) plus an oracle spot check on a slice."""
import math

import pytest
import torch

from keypoint_diffusion_amd import graph as G
from keypoint_diffusion_amd import synth
from keypoint_diffusion_amd.dynamics import LigRecDynamics
from keypoint_diffusion_amd.dynamics_gvp import LigRecDynamicsGVP
from oracle import egnn as oegnn

from . import util
from .test_gvp_gpu import GVP_ALL_ATOM

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('gemm_mode')]   # both GEMM modes of the EGNN edge kernel (conftest.py)
CUT = util.CUTOFFS_ALL_ATOM


def _rot(seed):
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=torch.Generator().manual_seed(seed)))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def _batch(B, n_rec, n_lig, v=None):
    gs = synth.synth_complexes(n_rec, n_lig, 20, CUT, seed=4242)
    return gs, util.fixed_encode(G.batch(gs), n_vec=v)


def test_c2_full_batch_properties(cuda):
    """configs[1]: egnn_all_atom, B = 64 x (300, 25)."""
    B = 64
    gs, g = _batch(B, [300] * B, [25] * B)
    model = synth.fill_state_dict_(LigRecDynamics(10, 10, graph_cutoffs=CUT, **util.EGNN_C2), 0).eval()
    t = torch.linspace(0.05, 1.0, B)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(cuda)
    gd = g.to(cuda)
    with torch.no_grad():
        h, x = model(gd, t.to(cuda), None)
        h2, x2 = model(gd, t.to(cuda), None)
    assert torch.equal(h, h2) and torch.equal(x, x2)                 # no atomics: bitwise reproducible
    # oracle on the first two complexes only (batch independence makes the slice meaningful)
    sub = util.to_obatch(util.fixed_encode(G.batch(gs[:2])))
    rh, rx = oegnn.egnn_dynamics_forward(sd, dict(util.EGNN_C2, graph_cutoffs=CUT), sub, t[:2])
    assert util.rel_err(h[:50].cpu(), rh) < 1e-4 and util.rel_err(x[:50].cpu(), rx) < 1e-4
    # E(3): rotate + translate every complex, eps_x rotates, eps_h is unchanged
    R, shift = _rot(3).to(cuda), torch.tensor([4.0, -7.0, 2.5], device=cuda)
    for nt in ('lig', 'kp'):
        gd.nodes[nt].data['x_0'] = gd.nodes[nt].data['x_0'] @ R.T + shift
    with torch.no_grad():
        hr, xr = model(gd, t.to(cuda), None)
    assert util.rel_err(hr, h) < 1e-4
    assert util.rel_err(xr, x @ R.T) < 1e-4


def test_c5_ragged_gvp_properties(cuda):
    """configs[4] shape on one GPU: gvp_all_atom, ragged pockets 150-600 atoms, ligands 15-35 atoms."""
    gen = torch.Generator().manual_seed(9)
    B = 24
    n_rec = torch.randint(150, 601, (B,), generator=gen).tolist()
    n_lig = torch.randint(15, 36, (B,), generator=gen).tolist()
    gs, g = _batch(B, n_rec, n_lig, v=16)
    model = synth.fill_state_dict_(LigRecDynamicsGVP(10, 10, graph_cutoffs=CUT, **GVP_ALL_ATOM), 1).eval().to(cuda)
    t = torch.linspace(0.1, 1.0, B).to(cuda)
    gd = g.to(cuda)
    with torch.no_grad():
        h, x = model(gd, t, None)
        # batch independence: complex 5 alone
        i = 5
        g1 = util.fixed_encode(G.batch([gs[i]]), n_vec=16).to(cuda)
        h1, x1 = model(g1, t[i:i + 1], None)
    off = sum(n_lig[:i])
    assert util.rel_err(h1, h[off:off + n_lig[i]]) < 1e-5 and util.rel_err(x1, x[off:off + n_lig[i]]) < 1e-5
    R, shift = _rot(5).to(cuda), torch.tensor([-3.0, 1.0, 8.0], device=cuda)
    for nt in ('lig', 'kp'):
        gd.nodes[nt].data['x_0'] = gd.nodes[nt].data['x_0'] @ R.T + shift
    with torch.no_grad():
        hr, xr = model(gd, t, None)
    assert torch.isfinite(h).all() and torch.isfinite(x).all()
    assert util.rel_err(hr, h) < 1e-4
    assert util.rel_err(xr, x @ R.T) < 1e-4


def test_c4_egnn_b512_properties(cuda):
    """configs[3] at its stated batch on ONE GPU: egnn_all_atom, B = 512 x (300, 25) -- the batch the 8-GPU run shards 64 per
    rank, here in one engine (3.2 M edges per layer): bitwise repeatability, an oracle slice taken from the far end of the batch,
    batch independence against complexes run alone, E(3)."""
    B = 512
    gs, g = _batch(B, [300] * B, [25] * B)
    model = synth.fill_state_dict_(LigRecDynamics(10, 10, graph_cutoffs=CUT, **util.EGNN_C2), 0).eval()
    t = torch.linspace(0.05, 1.0, B)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(cuda)
    gd = g.to(cuda)
    with torch.no_grad():
        h, x = model(gd, t.to(cuda), None)
        h2, x2 = model(gd, t.to(cuda), None)
    assert torch.equal(h, h2) and torch.equal(x, x2)
    assert h.shape == (B * 25, 10) and torch.isfinite(h).all() and torch.isfinite(x).all()
    # oracle on complexes 510, 511 (the end of the flat arrays: offsets beyond every B = 64 test)
    sub = util.to_obatch(util.fixed_encode(G.batch(gs[510:])))
    rh, rx = oegnn.egnn_dynamics_forward(sd, dict(util.EGNN_C2, graph_cutoffs=CUT), sub, t[510:])
    util.assert_parity(h[510 * 25:], rh, [25, 25], 1e-4, 'eps_h')
    util.assert_parity(x[510 * 25:], rx, [25, 25], 1e-4, 'eps_x', atol_rel=1e-5)
    # batch independence: complexes 200 and 447 alone
    for i in (200, 447):
        g1 = util.fixed_encode(G.batch([gs[i]])).to(cuda)
        with torch.no_grad():
            h1, x1 = model(g1, t[i:i + 1].to(cuda), None)
        assert util.rel_err(h1, h[25 * i:25 * (i + 1)]) < 1e-5 and util.rel_err(x1, x[25 * i:25 * (i + 1)]) < 1e-5
    R, shift = _rot(11).to(cuda), torch.tensor([-6.0, 3.0, 9.5], device=cuda)
    for nt in ('lig', 'kp'):
        gd.nodes[nt].data['x_0'] = gd.nodes[nt].data['x_0'] @ R.T + shift
    with torch.no_grad():
        hr, xr = model(gd, t.to(cuda), None)
    assert util.rel_err(hr, h) < 1e-4
    assert util.rel_err(xr, x @ R.T) < 1e-4


def test_c5_ragged_gvp_b512_properties(cuda):
    """configs[4] at its stated batch on ONE GPU: gvp_all_atom, B = 512 ragged pockets (150-600 atoms, ligands 15-35 atoms):
    bitwise repeatability, an oracle slice (the two smallest pockets, wherever they fall in the batch), batch independence, E(3)."""
    from oracle import gvp as ogvp
    gen = torch.Generator().manual_seed(21)
    B = 512
    n_rec = torch.randint(150, 601, (B,), generator=gen).tolist()
    n_lig = torch.randint(15, 36, (B,), generator=gen).tolist()
    gs, g = _batch(B, n_rec, n_lig, v=16)
    model = synth.fill_state_dict_(LigRecDynamicsGVP(10, 10, graph_cutoffs=CUT, **GVP_ALL_ATOM), 1).eval()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(cuda)
    t = torch.linspace(0.1, 1.0, B)
    gd = g.to(cuda)
    off = [0]
    for n in n_lig:
        off.append(off[-1] + n)
    with torch.no_grad():
        h, x = model(gd, t.to(cuda), None)
        h2, x2 = model(gd, t.to(cuda), None)
    assert torch.equal(h, h2) and torch.equal(x, x2)
    assert h.shape == (off[-1], 10) and torch.isfinite(h).all() and torch.isfinite(x).all()
    order = sorted(range(B), key=lambda i: n_rec[i])
    for i in order[:2]:                                               # oracle on the two smallest pockets, each alone
        sub = util.to_obatch(util.fixed_encode(G.batch([gs[i]]), n_vec=16))
        rh, rx = ogvp.gvp_dynamics_forward(sd, dict(GVP_ALL_ATOM, graph_cutoffs=CUT), sub, t[i:i + 1])
        util.assert_parity(h[off[i]:off[i + 1]], rh, [n_lig[i]], 1e-4, f'eps_h[{i}]')
        util.assert_parity(x[off[i]:off[i + 1]], rx, [n_lig[i]], 1e-4, f'eps_x[{i}]', atol_rel=1e-5)
    for i in (order[-1], 300):                                        # batch independence: the largest pocket and one from the middle
        g1 = util.fixed_encode(G.batch([gs[i]]), n_vec=16).to(cuda)
        with torch.no_grad():
            h1, x1 = model(g1, t[i:i + 1].to(cuda), None)
        assert util.rel_err(h1, h[off[i]:off[i + 1]]) < 1e-5 and util.rel_err(x1, x[off[i]:off[i + 1]]) < 1e-5
    R, shift = _rot(13).to(cuda), torch.tensor([5.0, -2.0, 7.0], device=cuda)
    for nt in ('lig', 'kp'):
        gd.nodes[nt].data['x_0'] = gd.nodes[nt].data['x_0'] @ R.T + shift
    with torch.no_grad():
        hr, xr = model(gd, t.to(cuda), None)
    assert util.rel_err(hr, h) < 1e-4
    assert util.rel_err(xr, x @ R.T) < 1e-4


def test_c3_gvp_40kp_full_batch_properties(cuda):
    """configs[2]: gvp_40kp, B = 64 x (300 receptor atoms -> 40 learned keypoints, 25 ligand atoms): learned GVP receptor
    encoder -> GVP denoiser (trained_models/gvp_40kp/config.yml:51-62, 90-102), through the size-independent properties of the
    C2 test: bitwise run-to-run equality, an oracle slice (encoder and denoiser), batch independence, E(3)."""
    from keypoint_diffusion_amd.ligand_diffuser import KeypointDiffusion
    from oracle import gvp as ogvp
    from oracle import rec_encoder as orec
    from .test_gvp_gpu import GVP_40KP
    from .test_recenc_gpu import RECENC_40KP
    B = 64
    cut = dict(CUT, kl=8, ll=6.0)
    rec_cfg = {k: v for k, v in RECENC_40KP.items() if k != 'n_keypoints'}
    model = KeypointDiffusion(10, 128, None, n_timesteps=500, architecture='gvp', rec_encoder_type='learned',
                              graph_config=dict(n_keypoints=40, graph_cutoffs=cut), dynamics_config=GVP_40KP, rec_encoder_config=rec_cfg,
                              precision=1e-5)
    synth.fill_state_dict_(model, 0)
    sd_enc = {k[len('rec_encoder.'):]: v.clone() for k, v in model.state_dict().items() if k.startswith('rec_encoder.')}
    sd_dyn = {k[len('dynamics.'):]: v.clone() for k, v in model.state_dict().items() if k.startswith('dynamics.')}
    model = model.eval().to(cuda)
    raw = lambda: synth.synth_complexes([300] * B, [25] * B, 40, cut, seed=4242)
    t = torch.linspace(0.05, 1.0, B)
    with torch.no_grad():
        g = model.encode_receptors(G.batch(raw()).to(cuda))
        g2 = model.encode_receptors(G.batch(raw()).to(cuda))
        h, x = model.dynamics(g, t.to(cuda), None)
        hb, xb = model.dynamics(g2, t.to(cuda), None)
    for k in ('x_0', 'h_0', 'v_0'):                                   # encoder and denoiser: no atomics, bitwise reproducible
        assert torch.equal(g.nodes['kp'].data[k], g2.nodes['kp'].data[k]), k
    assert torch.equal(h, hb) and torch.equal(x, xb)
    assert g.num_nodes('kp') == 40 * B and torch.isfinite(h).all() and torch.isfinite(x).all()
    # oracle slice, first two complexes: the encoder from the raw pockets, the denoiser from the GPU encoder's keypoints
    sub_raw = util.to_obatch(G.batch(raw()[:2]))
    enc_ref = orec.rec_encoder_gvp_forward(sd_enc, dict(RECENC_40KP, graph_cutoffs=cut), sub_raw)
    kp = g.nodes['kp'].data
    for k, ref in (('x_0', enc_ref.x['kp']), ('h_0', enc_ref.h['kp']), ('v_0', enc_ref.v['kp'])):
        util.assert_parity(kp[k][:80].reshape(80, -1), ref.reshape(80, -1), [40, 40], 1e-4, f'kp {k}', atol_rel=1e-5)
    sub = util.to_obatch(G.batch(G.unbatch(g.to('cpu'))[:2]))
    rh, rx = ogvp.gvp_dynamics_forward(sd_dyn, dict(GVP_40KP, graph_cutoffs=cut), sub, t[:2])
    util.assert_parity(h[:50], rh, [25, 25], 1e-4, 'eps_h')
    util.assert_parity(x[:50], rx, [25, 25], 1e-4, 'eps_x')
    # batch independence: complex 5 alone through encoder + denoiser
    i = 5
    with torch.no_grad():
        g1 = model.encode_receptors(G.batch([raw()[i]]).to(cuda))
        h1, x1 = model.dynamics(g1, t[i:i + 1].to(cuda), None)
    assert util.rel_err(g1.nodes['kp'].data['h_0'], kp['h_0'][40 * i:40 * (i + 1)]) < 1e-5
    assert util.rel_err(h1, h[25 * i:25 * (i + 1)]) < 1e-5 and util.rel_err(x1, x[25 * i:25 * (i + 1)]) < 1e-5
    # E(3): rotate + translate pockets and ligands before the encoder; keypoints and their vectors follow, eps_x rotates
    R, shift = _rot(7).to(cuda), torch.tensor([2.0, -5.0, 3.5], device=cuda)
    gr = G.batch(raw()).to(cuda)
    for nt in ('lig', 'rec'):
        gr.nodes[nt].data['x_0'] = gr.nodes[nt].data['x_0'] @ R.T + shift
    with torch.no_grad():
        gr = model.encode_receptors(gr)
        hr, xr = model.dynamics(gr, t.to(cuda), None)
    assert util.rel_err(gr.nodes['kp'].data['x_0'], kp['x_0'] @ R.T + shift) < 1e-4
    assert util.rel_err(gr.nodes['kp'].data['v_0'], kp['v_0'] @ R.T) < 1e-4
    assert util.rel_err(gr.nodes['kp'].data['h_0'], kp['h_0']) < 1e-4
    assert util.rel_err(hr, h) < 1e-4
    assert util.rel_err(xr, x @ R.T) < 1e-4


@pytest.mark.parametrize('arch', ['egnn', 'gvp'])
def test_full_batch_gradients_directional_derivative(cuda, arch):
    """configs[1] / configs[4]-shape batch through the training engines: the gradient along a random direction in weight
    space matches the central finite difference of the loss computed with the fused inference engine (a size-independent
    property: it exercises the full-size workspaces, split-K weight gradients and edge capacities)."""
    B = 64
    if arch == 'egnn':
        _, g = _batch(B, [300] * B, [25] * B)
        model = synth.fill_state_dict_(LigRecDynamics(10, 10, graph_cutoffs=CUT, **util.EGNN_C2), 0)
        pick = ['egnn.conv_layers.2.edge_mlp.kk.2.weight', 'egnn.conv_layers.0.edge_mlp.kl.0.weight',
                'egnn.conv_layers.4.node_mlp.lig.0.weight', 'egnn.conv_layers.1.coord_mlp.ll.2.weight', 'lig_encoder.2.weight']
    else:
        gen = torch.Generator().manual_seed(8)
        n_rec = torch.randint(150, 601, (B,), generator=gen).tolist()
        n_lig = torch.randint(15, 36, (B,), generator=gen).tolist()
        _, g = _batch(B, n_rec, n_lig, v=16)
        model = synth.fill_state_dict_(LigRecDynamicsGVP(10, 10, graph_cutoffs=CUT, **dict(GVP_ALL_ATOM, dropout=0.0)), 0)
        pick = ['noise_predictor.conv_layers.1.edge_message_fns.kp_kk_kp.1.to_feats_out.0.weight',
                'noise_predictor.conv_layers.0.edge_message_fns.kp_kl_lig.0.to_feats_out.0.weight',
                'noise_predictor.conv_layers.3.node_update_fns.lig.0.Wh', 'noise_predictor.noise_predictor.gvps.1.to_feats_out.0.weight',
                'lig_encoder.0.weight']
    model = model.eval().to(cuda)
    t = torch.linspace(0.05, 1.0, B).to(cuda)
    gd = g.to(cuda)
    n_lig_tot = gd.num_nodes('lig')
    gen = torch.Generator().manual_seed(3)
    w_h, w_x = torch.randn(n_lig_tot, 10, generator=gen).to(cuda), torch.randn(n_lig_tot, 3, generator=gen).to(cuda)

    def loss():
        eh, ex = model(gd, t, None)
        return (eh * w_h).sum() + (ex * w_x).sum()

    loss().backward()
    params = dict(model.named_parameters())
    dirs = {n: torch.randn(params[n].shape, generator=gen).to(cuda) for n in pick}
    analytic = sum(float((params[n].grad.double() * dirs[n].double()).sum()) for n in pick)
    eps = 2e-3
    vals = []
    with torch.no_grad():
        for sign in (1.0, -1.0):
            for n in pick:
                params[n].add_(sign * eps * dirs[n])
            vals.append(float(loss().double()))
            for n in pick:
                params[n].sub_(sign * eps * dirs[n])
    numeric = (vals[0] - vals[1]) / (2 * eps)
    assert all(torch.isfinite(p.grad).all() for p in params.values() if p.grad is not None)
    print(f'directional derivative {arch}: analytic {analytic:.6g} numeric {numeric:.6g}')
    assert abs(analytic - numeric) <= 3e-2 * max(abs(analytic), abs(numeric), 1e-3), (analytic, numeric)


@pytest.mark.parametrize('arch', ['egnn', 'gvp'])
def test_permutation_equivariance_over_atoms(cuda, arch):
    """Relabelling the ligand atoms and the keypoints inside every complex permutes the predicted noise and nothing else
    (a property the architecture guarantees; it crosses the graph builders, whose edge order changes)."""
    B = 6
    n_rec, n_lig = [120, 77, 150, 64, 99, 131], [12, 25, 7, 18, 9, 30]
    gs = synth.synth_complexes(n_rec, n_lig, 20, CUT, seed=77)
    g = util.fixed_encode(G.batch(gs), n_vec=16 if arch == 'gvp' else None)
    if arch == 'egnn':
        model = synth.fill_state_dict_(LigRecDynamics(10, 10, graph_cutoffs=CUT, **dict(util.EGNN_C2, n_layers=3)), 1)
    else:
        model = synth.fill_state_dict_(LigRecDynamicsGVP(10, 10, graph_cutoffs=CUT, **dict(GVP_ALL_ATOM, n_convs=3)), 1)
        g.nodes['kp'].data['v_0'] = 0.3 * torch.randn(g.num_nodes('kp'), 16, 3, generator=torch.Generator().manual_seed(5))
    model = model.eval().to(cuda)
    t = torch.linspace(0.1, 0.9, B)
    gen = torch.Generator().manual_seed(9)

    def perm_of(counts):
        parts, off = [], 0
        for n in counts:
            parts.append(off + torch.randperm(n, generator=gen))
            off += n
        return torch.cat(parts)

    pl, pk = perm_of(n_lig), perm_of(n_rec)
    inv_k = torch.empty_like(pk)
    inv_k[pk] = torch.arange(pk.numel())
    g2 = util.fixed_encode(G.batch(synth.synth_complexes(n_rec, n_lig, 20, CUT, seed=77)), n_vec=16 if arch == 'gvp' else None)
    for key in ('x_0', 'h_0'):
        g2.nodes['lig'].data[key] = g.nodes['lig'].data[key][pl]
        g2.nodes['kp'].data[key] = g.nodes['kp'].data[key][pk]
    if arch == 'gvp':
        g2.nodes['kp'].data['v_0'] = g.nodes['kp'].data['v_0'][pk]
    s, d = g.edges(etype='kk')
    g2.remove_edges(torch.arange(g2.num_edges('kk')), etype='kk')
    g2.add_edges(inv_k[s], inv_k[d], etype='kk')                 # the same keypoint graph under the new labels
    g2.set_batch_num_edges({'kk': g.batch_num_edges('kk')})
    with torch.no_grad():
        eh, ex = model(g.to(cuda), t.to(cuda), None)
        eh2, ex2 = model(g2.to(cuda), t.to(cuda), None)
    assert util.rel_err(eh2.cpu(), eh.cpu()[pl]) < 2e-5, util.rel_err(eh2.cpu(), eh.cpu()[pl])
    assert util.rel_err(ex2.cpu(), ex.cpu()[pl]) < 2e-5, util.rel_err(ex2.cpu(), ex.cpu()[pl])
