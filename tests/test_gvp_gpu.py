"""GPU parity of the HIP GVP denoiser (through the C ABI) against the CPU oracle."""
import pytest
import torch

from keypoint_diffusion_amd import graph as G
from keypoint_diffusion_amd import synth
from keypoint_diffusion_amd.dynamics_gvp import LigRecDynamicsGVP
from oracle import gvp as ogvp

from . import util
from .golden.make_golden_cfgs import GVP_CFGS

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('gemm_mode')]   # both GEMM modes of the denoiser engines (conftest.py)
TOL = 1e-4
CUT = util.CUTOFFS_ALL_ATOM

GVP_40KP = dict(vector_size=16, n_convs=6, n_hidden_scalars=256, message_norm=10.0, update_kp=True, ll_k=0, kl_k=7,
                n_message_gvps=3, n_update_gvps=2, n_noise_gvps=4, dropout=0.1)          # trained_models/gvp_40kp
GVP_ALL_ATOM = dict(GVP_40KP, message_norm='mean')                                       # trained_models/gvp_all_atom


def _check(h, x, rh, rx, n_lig):
    """eps_h / eps_x against the oracle: whole-tensor relative error, the same per complex, and elementwise allclose."""
    util.assert_parity(h, rh, n_lig, TOL, 'eps_h')
    util.assert_parity(x, rx, n_lig, TOL, 'eps_x')


def _run(cuda, cfg, n_rec, n_lig, n_kp_scalars=10, convs=None, seed=7, rand_v=True):
    nv = cfg.get('vector_size', 16)
    g = util.fixed_encode(util.make_batch(n_rec, n_lig, seed=31), n_vec=nv)
    gen = torch.Generator().manual_seed(3)
    if rand_v:
        g.nodes['kp'].data['v_0'] = 0.5 * torch.randn(g.num_nodes('kp'), nv, 3, generator=gen)
    if n_kp_scalars != 10:
        g.nodes['kp'].data['h_0'] = torch.randn(g.num_nodes('kp'), n_kp_scalars, generator=gen)
    model = LigRecDynamicsGVP(10, n_kp_scalars, graph_cutoffs=CUT, **cfg)
    synth.fill_state_dict_(model, seed)
    model.eval()
    B = g.batch_size
    t = (torch.arange(B, dtype=torch.float32) + 1) / (B + 1)
    ocfg = dict(cfg, graph_cutoffs=CUT)
    if convs is not None:
        ocfg['n_convs_run'] = convs
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ref_h, ref_x = ogvp.gvp_dynamics_forward(sd, ocfg, util.to_obatch(g), t)
    model = model.to(cuda)
    gd = g.to(cuda)
    with torch.no_grad():
        eps_h, eps_x = model(gd, t.to(cuda), G.get_batch_idxs(gd))
    torch.cuda.synchronize()
    return (eps_h.cpu(), eps_x.cpu()), (ref_h, ref_x)


@pytest.mark.parametrize('tag', list(GVP_CFGS))
def test_gvp_small_configs(cuda, tag):
    cfg = GVP_CFGS[tag]
    (h, x), (rh, rx) = _run(cuda, cfg, [26, 19], [7, 10], n_kp_scalars=128 if tag == 'gvp_kp' else 10)
    _check(h, x, rh, rx, [7, 10])


@pytest.mark.parametrize('ll_k,kl_k,tag', [(3, 5, 'gvp_norm0'), (0, 0, 'gvp_norm0'), (4, 0, 'gvp_mean')])
def test_gvp_graph_variants(cuda, ll_k, kl_k, tag):
    """kNN lig-lig graph (ll_k > 0) and radius keypoint->ligand graph (kl_k = 0), dynamics_gvp.py:206-225."""
    cfg = dict(GVP_CFGS[tag], ll_k=ll_k, kl_k=kl_k)
    (h, x), (rh, rx) = _run(cuda, cfg, [60, 33], [12, 10])
    _check(h, x, rh, rx, [12, 10])


def test_gvp_40kp_shape(cuda):
    # 40 learned keypoints with 128 scalars + vectors, complete kk graph inside 8 A (C3 shape)
    gs = []
    for i, nl in enumerate([25, 12]):
        gen = torch.Generator().manual_seed(50 + i)
        kp_pos = torch.randn(40, 3, generator=gen) * 2.5
        src, dst = synth.radius_graph_dense(kp_pos, 8.0)
        g = G.heterograph({('kp', 'kk', 'kp'): (src, dst)}, {'rec': 0, 'kp': 40, 'lig': nl})
        g.nodes['kp'].data['x_0'] = kp_pos
        g.nodes['kp'].data['h_0'] = torch.randn(40, 128, generator=gen)
        g.nodes['kp'].data['v_0'] = 0.5 * torch.randn(40, 16, 3, generator=gen)
        lx = torch.randn(nl, 3, generator=gen)
        g.nodes['lig'].data['x_0'] = lx - lx.mean(0, keepdim=True)
        g.nodes['lig'].data['h_0'] = torch.randn(nl, 10, generator=gen)
        g.nodes['rec'].data['x_0'] = torch.zeros(0, 3)
        g.nodes['rec'].data['h_0'] = torch.zeros(0, 10)
        gs.append(g)
    g = G.batch(gs)
    model = synth.fill_state_dict_(LigRecDynamicsGVP(10, 128, graph_cutoffs=CUT, **GVP_40KP), 5).eval()
    t = torch.tensor([0.3, 0.8])
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rh, rx = ogvp.gvp_dynamics_forward(sd, dict(GVP_40KP, graph_cutoffs=CUT), util.to_obatch(g), t)
    model = model.to(cuda)
    gd = g.to(cuda)
    with torch.no_grad():
        h, x = model(gd, t.to(cuda), None)
    _check(h.cpu(), x.cpu(), rh, rx, [25, 12])


def test_gvp_all_atom_ragged(cuda):
    (h, x), (rh, rx) = _run(cuda, GVP_ALL_ATOM, [150, 420, 64], [15, 35, 3], rand_v=False)
    _check(h, x, rh, rx, [15, 35, 3])


@pytest.mark.parametrize('n_rec,n_lig', [([40], [1]), ([40, 55], [1, 2]), ([8], [3]), ([300], [60])])
def test_gvp_degenerate_shapes(cuda, n_rec, n_lig):
    """Single-atom ligands (empty lig-lig graph), pockets smaller than one tile, the largest ligand of the datasets."""
    (h, x), (rh, rx) = _run(cuda, GVP_CFGS['gvp_norm0'], n_rec, n_lig)
    _check(h, x, rh, rx, n_lig)


@pytest.mark.parametrize('width,vectors', [(100, 16), (127, 16), (130, 16), (255, 16), (17, 16), (128, 8), (256, 15), (90, 5), (64, 1)])
def test_other_widths(cuda, width, vectors):
    """n_hidden_scalars and vector_size are free in the reference constructor (models/dynamics_gvp.py:106, defaults 128 / 16): narrower
    models run on the 128- / 256-wide, 16-channel kernels with zero-padded weight blocks and GVPLayerNorms of the true widths."""
    cfg = dict(GVP_40KP, n_hidden_scalars=width, vector_size=vectors, n_convs=3)
    (h, x), (rh, rx) = _run(cuda, cfg, [40, 25, 31], [9, 12, 7])
    _check(h, x, rh, rx, [9, 12, 7])


@pytest.mark.parametrize('tag,n_rec,n_lig,nkp', [('gvp_all_atom', [37, 64, 21], [9, 17, 5], 10), ('gvp_kp', [26, 19], [7, 10], 128),
                                                 ('gvp_s128', [33, 18, 50], [16, 3, 11], 10)])
def test_node_side_kernel_forms(cuda, tag, n_rec, n_lig, nkp):
    """The node update, noise head and per-node projections exist in two forms (gvp_kernels.h): register-chained (one wave = 16 rows x all
    columns, gvp_chain.hip) and cooperative (four waves split the columns of 16 rows, gvp_coop.hip); a launch takes the cooperative one up
    to COOP_ROWS_DEFAULT rows.  Both forms against the oracle on the same inputs (engine switch `coop_rows=`: -1 never, 1 << 30 always), at
    256 and 128 hidden scalars, ragged row counts that leave partly filled 16- and 64-row workgroups; they agree with each other to
    rounding (the gate product is summed in four partials in the cooperative form) and each is bitwise repeatable."""
    cfg = {'gvp_all_atom': GVP_ALL_ATOM, 'gvp_kp': GVP_CFGS['gvp_kp'], 'gvp_s128': dict(GVP_ALL_ATOM, n_hidden_scalars=128, n_convs=3)}[tag]
    nv = cfg.get('vector_size', 16)
    g = util.fixed_encode(util.make_batch(n_rec, n_lig, seed=41), n_vec=nv)
    gen = torch.Generator().manual_seed(5)
    g.nodes['kp'].data['v_0'] = 0.5 * torch.randn(g.num_nodes('kp'), nv, 3, generator=gen)
    if nkp != 10:
        g.nodes['kp'].data['h_0'] = torch.randn(g.num_nodes('kp'), nkp, generator=gen)
    model = LigRecDynamicsGVP(10, nkp, graph_cutoffs=CUT, **cfg)
    synth.fill_state_dict_(model, 9)
    model.eval()
    B = g.batch_size
    t = (torch.arange(B, dtype=torch.float32) + 1) / (B + 1)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rh, rx = ogvp.gvp_dynamics_forward(sd, dict(cfg, graph_cutoffs=CUT), util.to_obatch(g), t)
    model = model.to(cuda)
    gd = g.to(cuda)
    outs = {}
    with torch.no_grad():
        for form, rows in (('chained', -1), ('cooperative', 1 << 30)):
            model.engine().debug(f'coop_rows={rows}')
            h, x = model(gd, t.to(cuda), G.get_batch_idxs(gd))
            h2, x2 = model(gd, t.to(cuda), G.get_batch_idxs(gd))
            assert torch.equal(h, h2) and torch.equal(x, x2), form
            outs[form] = (h.cpu(), x.cpu())
            _check(h.cpu(), x.cpu(), rh, rx, n_lig)
        model.engine().debug('coop_rows=0')
    assert util.rel_err(outs['chained'][0], outs['cooperative'][0]) < 1e-5 and util.rel_err(outs['chained'][1], outs['cooperative'][1]) < 1e-5
