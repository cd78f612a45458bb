"""GPU parity of the HIP EGNN denoiser (through the C ABI) against the CPU oracle."""
import pytest
import torch

from keypoint_diffusion_amd import graph as G
from keypoint_diffusion_amd import hip, synth
from keypoint_diffusion_amd.dynamics import LigRecDynamics
from oracle import egnn as oegnn
from oracle import graph_ops as og

from . import util

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('gemm_mode')]   # both GEMM modes of the EGNN edge kernel (conftest.py)
TOL = 1e-4          # BASELINE.json north_star: "within 1e-4 rel fp32"


def _edges_from_hip(out, n):
    E_ll, E_kl = int(out['counts'][0]), int(out['counts'][1])
    return {
        'll': (out['ll_src'][:E_ll].long().cpu(), out['ll_dst'][:E_ll].long().cpu()),
        'kl': (out['kl_src'][:E_kl].long().cpu(), out['kl_dst'][:E_kl].long().cpu()),
        'lk': (out['lk_src'][:E_kl].long().cpu(), out['lk_dst'][:E_kl].long().cpu()),
    }


def _edge_set(src, dst):
    return set(zip(src.tolist(), dst.tolist()))


@pytest.mark.parametrize('n_rec,n_lig', [([60], [20]), ([300, 150, 420], [25, 4, 37]), ([35, 600], [60, 3])])
def test_lig_graph_build(cuda, n_rec, n_lig):
    g = util.fixed_encode(util.make_batch(n_rec, n_lig)).to(cuda)
    pb = g.prepared()
    lx, kx = g.nodes['lig'].data['x_0'], g.nodes['kp'].data['x_0']
    out = hip.build_lig_graph(pb, lx, kx, 6.0, 5)
    torch.cuda.synchronize()
    e = _edges_from_hip(out, pb)
    ob = util.to_obatch(g)
    ref = oegnn.lig_edges(ob, dict(graph_cutoffs={'ll': 6.0}, kl_k=5, ll_k=0))
    # ll: identical edge set, dst-sorted, rowptr consistent
    assert _edge_set(*e['ll']) == _edge_set(*ref['ll'])
    assert torch.equal(e['ll'][0], ref['ll'][0]) and torch.equal(e['ll'][1], ref['ll'][1])
    deg = torch.bincount(e['ll'][1], minlength=pb.n_lig)
    assert torch.equal(out['ll_rowptr'].long().cpu()[1:] - out['ll_rowptr'].long().cpu()[:-1], deg)
    assert torch.equal(out['ll_per_graph'].long().cpu(), og.edges_per_graph(ref['ll'][1], ob.n['lig']))
    # kl / lk: same pairs (oracle kl is kp-major; HIP kl is lig-major, lk is kp-major nearest first)
    assert _edge_set(*e['kl']) == _edge_set(*ref['kl'])
    assert torch.equal(e['lk'][0], ref['lk'][0]) and torch.equal(e['lk'][1], ref['lk'][1])
    assert bool((e['kl'][1][1:] >= e['kl'][1][:-1]).all())
    deg = torch.bincount(e['kl'][1], minlength=pb.n_lig)
    assert torch.equal(out['kl_rowptr'].long().cpu()[1:] - out['kl_rowptr'].long().cpu()[:-1], deg)


@pytest.mark.parametrize('ll_k,kl_k', [(3, 5), (0, 0), (4, 0), (30, 2)])
@pytest.mark.parametrize('n_rec,n_lig', [([60], [20]), ([300, 150, 420], [25, 4, 37]), ([35, 600], [60, 3])])
def test_lig_graph_variants(cuda, n_rec, n_lig, ll_k, kl_k):
    """ll_k > 0 (kNN lig-lig graph) and kl_k = 0 (radius keypoint->ligand graph), dynamics.py:392-411."""
    ll_k = min(ll_k, 16)
    g = util.fixed_encode(util.make_batch(n_rec, n_lig)).to(cuda)
    pb = g.prepared()
    lx, kx = g.nodes['lig'].data['x_0'], g.nodes['kp'].data['x_0']
    out = hip.build_lig_graph(pb, lx, kx, 6.0, kl_k, ll_k=ll_k, kl_cutoff=7.0)
    torch.cuda.synchronize()
    e = _edges_from_hip(out, pb)
    ob = util.to_obatch(g)
    ref = oegnn.lig_edges(ob, dict(graph_cutoffs={'ll': 6.0, 'kl': 7.0}, kl_k=kl_k, ll_k=ll_k))
    assert torch.equal(e['ll'][0], ref['ll'][0]) and torch.equal(e['ll'][1], ref['ll'][1])          # same edges, same order
    deg = torch.bincount(e['ll'][1], minlength=pb.n_lig)
    assert torch.equal(out['ll_rowptr'].long().cpu()[1:] - out['ll_rowptr'].long().cpu()[:-1], deg)
    assert torch.equal(out['ll_per_graph'].long().cpu(), og.edges_per_graph(ref['ll'][1], ob.n['lig']))
    assert torch.equal(e['lk'][0], ref['lk'][0]) and torch.equal(e['lk'][1], ref['lk'][1])          # kp-major, oracle order
    assert _edge_set(*e['kl']) == _edge_set(*ref['kl']) and e['kl'][0].numel() == ref['kl'][0].numel()
    assert bool((e['kl'][1][1:] >= e['kl'][1][:-1]).all())                                          # dst-sorted
    deg = torch.bincount(e['kl'][1], minlength=pb.n_lig)
    assert torch.equal(out['kl_rowptr'].long().cpu()[1:] - out['kl_rowptr'].long().cpu()[:-1], deg)
    deg = torch.bincount(e['lk'][1], minlength=pb.n_kp)
    assert torch.equal(out['lk_rowptr'].long().cpu()[1:] - out['lk_rowptr'].long().cpu()[:-1], deg)


def _check(h, x, rh, rx, n_lig):
    """eps_h / eps_x against the oracle: whole-tensor relative error, the same per complex, and elementwise allclose.  eps_x is the
    difference of two O(1) coordinates (x_out - x_0), so its absolute floor is that of the coordinates: atol 1e-5 of max |ref|."""
    util.assert_parity(h, rh, n_lig, TOL, 'eps_h')
    util.assert_parity(x, rx, n_lig, TOL, 'eps_x', atol_rel=1e-5)


def _run_pair(cuda, cfg, n_rec, n_lig, seed=3, rec_nf=10, layers=None):
    g = util.fixed_encode(util.make_batch(n_rec, n_lig, n_rec_feat=rec_nf))
    model = LigRecDynamics(10, rec_nf, graph_cutoffs=util.CUTOFFS_ALL_ATOM, **cfg)
    synth.fill_state_dict_(model, seed)
    model.eval()
    B = g.batch_size
    t = (torch.arange(B, dtype=torch.float32) + 1) / (B + 1)
    ob = util.to_obatch(g)
    ocfg = dict(cfg, graph_cutoffs=util.CUTOFFS_ALL_ATOM)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    if layers is not None:
        ocfg['n_layers'] = layers
    ref_h, ref_x = oegnn.egnn_dynamics_forward(sd, ocfg, ob, t)
    model = model.to(cuda)
    gd = g.to(cuda)
    with torch.no_grad():
        if layers is not None:
            model.engine().debug(f'layers={layers}')
        eps_h, eps_x = model(gd, t.to(cuda), G.get_batch_idxs(gd))
    torch.cuda.synchronize()
    return (eps_h.cpu(), eps_x.cpu()), (ref_h, ref_x), model


@pytest.mark.parametrize('ll_k,kl_k,message_norm', [(3, 5, 0), (0, 0, 0), (5, 0, 2.0)])
def test_egnn_graph_variants(cuda, ll_k, kl_k, message_norm):
    """The denoiser on the config-reachable graph variants (kNN lig-lig, radius keypoint->ligand): the per-complex
    normaliser z then depends on counted kl edges."""
    cfg = dict(util.EGNN_C2, ll_k=ll_k, kl_k=kl_k, message_norm=message_norm, n_layers=2)
    (h, x), (rh, rx), _ = _run_pair(cuda, cfg, [300, 150, 40], [25, 9, 3])
    _check(h, x, rh, rx, [25, 9, 3])


@pytest.mark.parametrize('layers', [0, 1, 2, 6])
def test_egnn_c2_shape_layers(cuda, layers):
    (h, x), (rh, rx), _ = _run_pair(cuda, util.EGNN_C2, [300, 150], [25, 9], layers=layers)
    util.assert_parity(h, rh, [25, 9], TOL, 'eps_h')
    if layers > 0:
        util.assert_parity(x, rx, [25, 9], TOL, 'eps_x', atol_rel=1e-5)
    else:
        assert float(x.abs().max()) == 0.0


def test_egnn_ragged_batch(cuda):
    (h, x), (rh, rx), model = _run_pair(cuda, util.EGNN_C2, [150, 600, 35, 300, 64], [15, 35, 3, 60, 25])
    _check(h, x, rh, rx, [15, 35, 3, 60, 25])
    c = model.engine().last_counts()
    assert c['E_kl'] == c['E_lk'] and c['tiles'] > 0


@pytest.mark.parametrize('n_rec,n_lig,regime', [
    ([400, 380, 420], [25, 30, 20], 'whole'),                                # ~390 tiles: 48 per XCD -- more than half its 64 slots: whole tiles only
    ([500, 520, 480, 510], [25, 30, 20, 28], 'rounds+split'),                # ~650 tiles: 81 per XCD -- a full round + 17 tiles that run split
])
def test_edge_tile_regimes(cuda, n_rec, n_lig, regime):
    """k_egnn_edge runs the tiles of an XCD's last round of workgroup slots as two work items (one per branch) when they fit half the
    slots.  Parity in the two regimes the small cases above do not reach: a launch whose remainder is too large to split, and one with
    full rounds followed by a split remainder (n_layers = 2 keeps the CPU oracle short)."""
    cfg = dict(util.EGNN_C2, n_layers=2)
    (h, x), (rh, rx), model = _run_pair(cuda, cfg, n_rec, n_lig)
    _check(h, x, rh, rx, n_lig)
    per_xcd = (model.engine().last_counts()['tiles'] + 7) // 8
    assert (32 < per_xcd % 64 < 64 and per_xcd < 64) if regime == 'whole' else (per_xcd > 64 and 0 < per_xcd % 64 <= 32), per_xcd


def test_egnn_dev_config_no_kp_update(cuda):
    # configs/dev_config.yml: update_kp_feat False, C-alpha pocket (rec_nf 20), ll cutoff from graph section
    (h, x), (rh, rx), _ = _run_pair(cuda, util.EGNN_DEV, [60], [20], rec_nf=20)
    _check(h, x, rh, rx, [20])


def test_egnn_const_message_norm_no_tanh_no_norm(cuda):
    cfg = dict(util.EGNN_C2, message_norm=5.0, use_tanh=False, norm=False, n_layers=3, kl_k=7)
    (h, x), (rh, rx), _ = _run_pair(cuda, cfg, [120, 90], [12, 30])
    _check(h, x, rh, rx, [12, 30])


def test_batch_independence(cuda):
    """Batched output equals per-complex outputs (every graph op is batch-masked)."""
    n_rec, n_lig = [200, 90, 310], [25, 11, 18]
    (h, x), _, model = _run_pair(cuda, util.EGNN_C2, n_rec, n_lig)
    off = 0
    for i, (nr, nl) in enumerate(zip(n_rec, n_lig)):
        gs = synth.synth_complexes(n_rec, n_lig, 20, util.CUTOFFS_ALL_ATOM, seed=1234)
        g1 = util.fixed_encode(G.batch([gs[i]])).to(cuda)
        t = torch.tensor([(i + 1) / (len(n_rec) + 1)], device=cuda)
        with torch.no_grad():
            h1, x1 = model(g1, t, None)
        assert util.rel_err(h1.cpu(), h[off:off + nl]) < 1e-5
        assert util.rel_err(x1.cpu(), x[off:off + nl]) < 1e-5
        off += nl


@pytest.mark.parametrize('n_rec,n_lig', [([40], [1]), ([40, 55], [1, 2]), ([8], [3]), ([300], [60])])
def test_degenerate_shapes(cuda, n_rec, n_lig):
    """Single-atom ligands (empty lig-lig graph), pockets smaller than one tile, the largest ligand of the datasets."""
    cfg = dict(util.EGNN_C2, n_layers=2)
    (h, x), (rh, rx), _ = _run_pair(cuda, cfg, n_rec, n_lig)
    _check(h, x, rh, rx, n_lig)


def test_final_layer_pruning_is_bit_identical(cuda):
    """LigRecEGNN.forward returns (h_lig, x_lig) only (models/dynamics.py:288-294): the final layer's lk / kk messages and
    keypoint update feed nothing.  The engine skips them; eps must be bit-for-bit what the full final layer gives."""
    g = util.fixed_encode(util.make_batch([300, 150, 40], [25, 9, 3]))
    model = LigRecDynamics(10, 10, graph_cutoffs=util.CUTOFFS_ALL_ATOM, **util.EGNN_C2)
    synth.fill_state_dict_(model, 3)
    model = model.eval().to(cuda)
    gd = g.to(cuda)
    t = torch.tensor([0.3, 0.6, 0.9], device=cuda)
    eng = model.engine()
    with torch.no_grad():
        eng.debug('prune=0')
        h0, x0 = model(gd, t, None)
        c0 = eng.last_counts()
        eng.debug('prune=1')
        h1, x1 = model(gd, t, None)
        c1 = eng.last_counts()
    torch.cuda.synchronize()
    assert torch.equal(h0, h1) and torch.equal(x0, x1)
    e_all = c0['E_ll'] + c0['E_kl'] + c0['E_lk'] + c0['E_kk']
    assert c0['E_last'] == e_all and c0['tiles_last'] == c0['tiles']
    assert c1['E_last'] == c1['E_ll'] + c1['E_kl'] and 0 < c1['tiles_last'] < c1['tiles']


@pytest.mark.parametrize('ll_k', [0, 3])
def test_non_finite_coordinates_stay_inside_their_complex(cuda, ll_k):
    """A diverged chain (NaN / Inf ligand coordinates) must behave like the reference -- NaNs propagate inside that complex --
    and never index outside it: the kNN builders used to leave empty best-list slots (index -1) when no distance compared finite."""
    g = util.fixed_encode(util.make_batch([60, 45, 50], [9, 7, 11]))
    cfg = dict(util.EGNN_C2, n_layers=2, ll_k=ll_k)
    model = LigRecDynamics(10, 10, graph_cutoffs=util.CUTOFFS_ALL_ATOM, **cfg)
    synth.fill_state_dict_(model, 3)
    model = model.eval().to(cuda)
    t = torch.tensor([0.3, 0.6, 0.9], device=cuda)
    with torch.no_grad():
        h_ok, x_ok = model(g.to(cuda), t, None)
        bad = g.to(cuda)
        x = bad.nodes['lig'].data['x_0'].clone()
        x[9:16] = float('nan')                         # every atom of complex 1
        x[2, 0] = float('inf')                         # one atom of complex 0
        bad.nodes['lig'].data['x_0'] = x
        out = hip.build_lig_graph(bad.prepared(), x, bad.nodes['kp'].data['x_0'], 6.0, 5, ll_k=ll_k, kl_cutoff=7.0)
        h, xx = model(bad, t, None)
    torch.cuda.synchronize()
    E_ll, E_kl = int(out['counts'][0]), int(out['counts'][1])
    n_lig, n_kp = 27, 155
    for key, hi in (('ll_src', n_lig), ('ll_dst', n_lig)):
        assert int(out[key][:E_ll].min()) >= 0 and int(out[key][:E_ll].max()) < hi
    for key, hi in (('kl_src', n_kp), ('kl_dst', n_lig), ('lk_src', n_lig), ('lk_dst', n_kp)):
        assert int(out[key][:E_kl].min()) >= 0 and int(out[key][:E_kl].max()) < hi
    # the untouched complex is what it was (its edges sit in other tiles now: summation order only); NaNs stay where they were put
    assert torch.isfinite(h[16:]).all() and torch.isfinite(xx[16:]).all()
    assert util.rel_err(h[16:], h_ok[16:]) < 1e-5 and util.rel_err(xx[16:], x_ok[16:]) < 1e-5
    assert torch.isnan(h[9:16]).any()


@pytest.mark.parametrize('hidden_nf,norm,update_kp', [(255, True, True), (64, True, False), (100, False, True)])
def test_other_hidden_widths(cuda, hidden_nf, norm, update_kp):
    """hidden_nf other than 256 -- the reference's own constructor default is 255 (models/dynamics.py:300-302) -- on the same
    kernels: features in columns 0 .. hidden_nf - 1 of the 256-wide layout, timestep in column 256, LayerNorm over hidden_nf + 1."""
    cfg = dict(util.EGNN_C2, hidden_nf=hidden_nf, norm=norm, update_kp_feat=update_kp, n_layers=3)
    (h, x), (rh, rx), model = _run_pair(cuda, cfg, [120, 77, 40], [25, 9, 3])
    _check(h, x, rh, rx, [25, 9, 3])
    assert model.egnn.conv_layers[0].edge_mlp['ll'][0].weight.shape == (hidden_nf + 1, 2 * (hidden_nf + 1) + 1)


def test_reference_default_constructor_builds_and_runs(cuda):
    """LigRecDynamics(atom_nf, rec_nf) with every default of the reference signature (n_layers=4, hidden_nf=255, message_norm=1, ...)."""
    gs = synth.synth_complexes([50, 31], [8, 5], 20, util.CUTOFFS_ALL_ATOM, seed=3)
    g = util.fixed_encode(G.batch(gs))
    model = synth.fill_state_dict_(LigRecDynamics(10, 10, graph_cutoffs=util.CUTOFFS_ALL_ATOM, kl_k=5), 2).eval()
    assert model.hidden_nf == 255 and model.n_layers == 4
    t = torch.tensor([0.3, 0.7])
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    cfg = dict(n_layers=4, hidden_nf=255, use_tanh=False, message_norm=1, update_kp_feat=False, norm=False, ll_k=0, kl_k=5,
               graph_cutoffs=util.CUTOFFS_ALL_ATOM)
    rh, rx = oegnn.egnn_dynamics_forward(sd, cfg, util.to_obatch(g), t)
    model = model.to(cuda)
    with torch.no_grad():
        h, x = model(g.to(cuda), t.to(cuda), None)
    _check(h.cpu(), x.cpu(), rh, rx, [8, 5])
