"""GPU parity of the HIP GVP receptor encoder (through the C ABI) against the CPU oracle."""
import pytest
import torch

from keypoint_diffusion_amd import graph as G
from keypoint_diffusion_amd import synth
from keypoint_diffusion_amd.receptor_encoder_gvp import ReceptorEncoderGVP
from oracle import rec_encoder as orec

from . import util
from .golden.make_golden_cfgs import RECENC_CFGS

pytestmark = pytest.mark.gpu
CUT = util.CUTOFFS_ALL_ATOM
RECENC_40KP = dict(in_scalar_size=10, out_scalar_size=128, n_message_gvps=3, n_update_gvps=2, vector_size=16, n_rr_convs=4,
                   n_rk_convs=2, message_norm=10.0, k_closest=5, kp_rad=0, dropout=0.1, n_keypoints=40)   # gvp_40kp
RECENC_NORM0 = dict(RECENC_CFGS['recenc_norm10'], message_norm=0)
# radius rec->kp graph (receptor_encoder_gvp.py:304-306): at most 10 receptor atoms within kp_rad per keypoint; 7 A around a
# keypoint of a 0.059 atoms/A^3 pocket holds ~80 atoms, so the cap is reached, 1.3 A leaves keypoints with no edge at all
RECENC_RAD = dict(RECENC_CFGS['recenc_mean'], k_closest=0, kp_rad=7.0)
RECENC_RAD_SPARSE = dict(RECENC_CFGS['recenc_norm10'], k_closest=0, kp_rad=1.3)


def _edge_set(s, d):
    return set(zip(s.tolist(), d.tolist()))


@pytest.mark.parametrize('cfg,n_rec', [(RECENC_CFGS['recenc_mean'], [33, 21]), (RECENC_CFGS['recenc_norm10'], [33, 21]),
                                       (RECENC_NORM0, [50, 3, 27]), (RECENC_40KP, [300, 150]), (RECENC_RAD, [120, 45]),
                                       (RECENC_RAD_SPARSE, [60, 8]),
                                       # vector_size < 16 (the reference accepts any, receptor_encoder_gvp.py:99-114): the 16-channel kernels run
                                       # the model zero-padded; the rk convs after the first read [x_diff | v_src | v_dst] = 1 + 2 V channels
                                       (dict(RECENC_CFGS['recenc_norm10'], vector_size=8), [33, 21]),
                                       (dict(RECENC_CFGS['recenc_mean'], vector_size=5), [40, 3, 27]),
                                       (dict(RECENC_40KP, vector_size=1), [120, 45])])
def test_receptor_encoder(cuda, cfg, n_rec):
    kw = dict(cfg, graph_cutoffs=CUT)
    model = synth.fill_state_dict_(ReceptorEncoderGVP(**kw), 61).eval()
    g = util.make_batch(n_rec, [4] * len(n_rec), seed=17, n_keypoints=cfg['n_keypoints'])
    ref = orec.rec_encoder_gvp_forward({k: v.clone() for k, v in model.state_dict().items()}, kw, util.to_obatch(g))
    model = model.to(cuda)
    gd = g.to(cuda)
    with torch.no_grad():
        out = model(gd, G.get_batch_idxs(gd))
    torch.cuda.synchronize()
    kp = out.nodes['kp'].data
    assert util.rel_err(kp['x_0'], ref.x['kp']) < 1e-4
    assert util.rel_err(kp['h_0'], ref.h['kp']) < 1e-4
    assert util.rel_err(kp['v_0'], ref.v['kp']) < 1e-4
    rs, rd = out.edges(etype='rk')
    assert torch.equal(rs.cpu(), ref.edges['rk'][0]) and torch.equal(rd.cpu(), ref.edges['rk'][1])
    ks, kd = out.edges(etype='kk')
    assert _edge_set(ks.cpu(), kd.cpu()) == _edge_set(*ref.edges['kk'])
    assert out.batch_num_edges('kk').sum() == ks.numel() and out.batch_num_edges('rk').sum() == rs.numel()
    assert out.batch_size == len(n_rec)
    if cfg.get('kp_rad', 0) > 0:
        deg = torch.bincount(rd.cpu(), minlength=out.num_nodes('kp'))
        assert int(deg.max()) <= 10 and (cfg is not RECENC_RAD or int(deg.max()) == 10)      # the cap of :306 is exercised
        assert cfg is not RECENC_RAD_SPARSE or int(deg.min()) == 0                           # and so are edgeless keypoints
        per_graph = torch.bincount(rd.cpu() // cfg['n_keypoints'], minlength=len(n_rec))
        assert torch.equal(out.batch_num_edges('rk').cpu(), per_graph)


def test_learned_encoder_feeds_gvp_denoiser(cuda):
    """encode_receptors -> copy per ligand -> denoiser forward: the gvp_40kp pipeline end to end."""
    from keypoint_diffusion_amd.ligand_diffuser import KeypointDiffusion
    from .test_gvp_gpu import GVP_40KP
    rec_cfg = {k: v for k, v in RECENC_40KP.items() if k not in ('in_scalar_size', 'n_keypoints')}
    m = KeypointDiffusion(10, 128, None, n_timesteps=6, architecture='gvp', rec_encoder_type='learned',
                          graph_config=dict(n_keypoints=40, graph_cutoffs=CUT), dynamics_config=dict(GVP_40KP, n_convs=2),
                          rec_encoder_config=dict(rec_cfg, in_scalar_size=10), precision=1e-5)
    synth.fill_state_dict_(m, 3)
    m = m.eval().to(cuda)
    pocket = synth.synth_complexes([120], [1], 40, CUT, seed=5)[0].to(cuda)
    pocket.remove_nodes(pocket.nodes('lig'), ntype='lig')
    pos, feat = m.sample_given_pocket(pocket, torch.tensor([7, 12]), diff_batch_size=2)
    assert [p.shape for p in pos] == [(7, 3), (12, 3)] and all(torch.isfinite(p).all() for p in pos)
