"""GPU parity of the HIP EGNN keypoint receptor encoder (kpd_recegnn_*, models/receptor_encoder.py) against the CPU oracle,
and the egnn_20kp pipeline (learned EGNN encoder -> EGNN denoiser) end to end."""
import pytest
import torch

from keypoint_diffusion_amd import graph as G
from keypoint_diffusion_amd import synth
from keypoint_diffusion_amd.receptor_encoder import ReceptorEncoder
from oracle import rec_encoder_egnn as orec

from . import util
from .golden.make_golden_cfgs import RECEGNN_CFGS, same_res_feature

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('gemm_mode')]   # both GEMM modes of the EGNN edge kernel (conftest.py)
CUT = util.CUTOFFS_ALL_ATOM
RECEGNN_40KP = dict(RECEGNN_CFGS['recegnn_20kp'], n_keypoints=40)        # trained_models/egnn_40kp/config.yml:59-75
# keypoint features from the receptor atoms within kp_rad (receptor_encoder.py:238-262; configs/dev_config.yml:46 sets kp_rad: 5):
# 5 A holds ~30 atoms, 9 A in a 300-atom pocket more than the cap of 100 (:246)
RECEGNN_RAD5 = dict(RECEGNN_CFGS['recegnn_small'], k_closest=0, kp_rad=5.0)
RECEGNN_RAD9 = dict(RECEGNN_CFGS['recegnn_20kp'], k_closest=0, kp_rad=9.0)


def _edge_set(s, d):
    return set(zip(s.tolist(), d.tolist()))


@pytest.mark.parametrize('cfg,n_rec', [(RECEGNN_CFGS['recegnn_20kp'], [33, 21]), (RECEGNN_CFGS['recegnn_small'], [33, 21]),
                                       (RECEGNN_CFGS['recegnn_fixpos'], [50, 5, 27]), (RECEGNN_40KP, [300, 150]),
                                       (RECEGNN_RAD5, [33, 21, 2]), (RECEGNN_RAD9, [300, 150])])
def test_egnn_receptor_encoder(cuda, cfg, n_rec):
    kw = dict(cfg, graph_cutoffs=CUT)
    model = synth.fill_state_dict_(ReceptorEncoder(**kw), 71).eval()
    g = util.make_batch(n_rec, [4] * len(n_rec), seed=19, n_keypoints=cfg['n_keypoints'])
    src, dst = g.edges(etype='rr')
    a = same_res_feature(src, dst)
    g.edges['rr'].data['same_res'] = a.bool()                  # the dataset stores a bool column
    ref, ref_h, ref_x = orec.rec_encoder_egnn_forward({k: v.clone() for k, v in model.state_dict().items()}, kw, util.to_obatch(g),
                                                      a if cfg['use_sameres_feat'] else None, return_rec=True)
    model = model.to(cuda)
    gd = g.to(cuda)
    with torch.no_grad():
        out = model(gd, G.get_batch_idxs(gd))
    torch.cuda.synchronize()
    rec = out.nodes['rec'].data
    assert util.rel_err(rec['h'], ref_h) < 1e-4
    assert util.rel_err(rec['x'], ref_x) < 1e-4
    kp = out.nodes['kp'].data
    assert util.rel_err(kp['x_0'], ref.x['kp']) < 1e-4
    assert util.rel_err(kp['h_0'], ref.h['kp']) < 1e-4
    rs, rd = out.edges(etype='rk')
    assert torch.equal(rs.cpu(), ref.edges['rk'][0]) and torch.equal(rd.cpu(), ref.edges['rk'][1])
    ks, kd = out.edges(etype='kk')
    assert _edge_set(ks.cpu(), kd.cpu()) == _edge_set(*ref.edges['kk'])
    assert out.batch_num_edges('kk').sum() == ks.numel() and out.batch_num_edges('rk').sum() == rs.numel()
    if cfg.get('kp_rad', 0) > 0:
        deg = torch.bincount(rd.cpu(), minlength=out.num_nodes('kp'))
        assert int(deg.max()) <= 100 and (cfg is not RECEGNN_RAD9 or int(deg.max()) == 100)       # the cap of :246 is exercised
        assert torch.equal(out.batch_num_edges('rk').cpu(), torch.bincount(rd.cpu() // cfg['n_keypoints'], minlength=len(n_rec)))


def test_too_few_receptor_atoms_is_refused(cuda):
    cfg = dict(RECEGNN_CFGS['recegnn_small'], graph_cutoffs=CUT)
    model = synth.fill_state_dict_(ReceptorEncoder(**cfg), 71).eval().to(cuda)
    g = util.make_batch([30, 2], [4, 4], seed=19, n_keypoints=cfg['n_keypoints']).to(cuda)
    from keypoint_diffusion_amd.hip import KpdError
    with torch.no_grad(), pytest.raises(KpdError, match='k_closest'):
        model(g, G.get_batch_idxs(g))


def test_learned_egnn_encoder_feeds_egnn_denoiser(cuda):
    """encode_receptors -> copy per ligand -> EGNN denoiser: the egnn_20kp pipeline end to end."""
    from keypoint_diffusion_amd.ligand_diffuser import KeypointDiffusion
    rec_cfg = {k: v for k, v in RECEGNN_CFGS['recegnn_20kp'].items() if k not in ('in_n_node_feat', 'n_keypoints')}
    dyn = dict(util.EGNN_C2, n_layers=2)
    m = KeypointDiffusion(10, 128, None, n_timesteps=6, architecture='egnn', rec_encoder_type='learned',
                          graph_config=dict(n_keypoints=20, graph_cutoffs=CUT), dynamics_config=dyn,
                          rec_encoder_config=dict(rec_cfg, in_n_node_feat=10), precision=1e-5)
    synth.fill_state_dict_(m, 3)
    m = m.eval().to(cuda)
    pocket = synth.synth_complexes([120], [1], 20, CUT, seed=5)[0].to(cuda)
    pocket.remove_nodes(pocket.nodes('lig'), ntype='lig')
    s, d = pocket.edges(etype='rr')
    pocket.edges['rr'].data['same_res'] = same_res_feature(s.cpu(), d.cpu()).bool().to(cuda)
    pos, feat = m.sample_given_pocket(pocket, torch.tensor([7, 12]), diff_batch_size=2)
    assert [p.shape for p in pos] == [(7, 3), (12, 3)] and all(torch.isfinite(p).all() for p in pos)
