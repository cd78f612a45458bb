"""Randomized parity at the SHIPPED size distribution (every other parity input is a hand-picked size).

tests/golden/size_pairs.json holds (n_rec, n_lig) pairs drawn from the joint histogram of the reference's training split
(data/bindingmoad_processed/train_n_node_joint_dist.pkl: 7 .. 661 pocket atoms, 2 .. 60 ligand atoms, means 332.8 / 19.8; generator:
tests/golden/make_size_pairs.py).  Three seeded batches of 32 complexes are drawn from them -- the first one with the four extremes of
the histogram's support forced in (a 7-atom and a 661-atom pocket, a 2-atom and a 60-atom ligand) -- and run through the EGNN denoiser
(egnn_all_atom), the GVP denoiser (gvp_all_atom) and the two learned receptor encoders, each against the CPU oracle with the three parity
measures of the denoiser tests (whole tensor, per complex, elementwise), 1e-4 relative, exact fp32 mode.  The CPU oracle is what takes
the time here (15 - 30 s per denoiser batch of 32), so the second and third batch carry 16 complexes.  The encoders' elementwise floor
is 1e-5 of the largest entry instead of 1e-6: their vector outputs are cancelling sums over up to 661 atoms and sit at 2e-6 of the
largest entry in either kernel form (whole-tensor and per-complex errors: 1e-6 .. 3e-6)."""
import json
import os

import pytest
import torch

from keypoint_diffusion_amd import graph as G
from keypoint_diffusion_amd import synth
from keypoint_diffusion_amd.dynamics import LigRecDynamics
from keypoint_diffusion_amd.dynamics_gvp import LigRecDynamicsGVP
from keypoint_diffusion_amd.receptor_encoder import ReceptorEncoder
from keypoint_diffusion_amd.receptor_encoder_gvp import ReceptorEncoderGVP
from oracle import egnn as oegnn
from oracle import gvp as ogvp
from oracle import rec_encoder as orec_gvp
from oracle import rec_encoder_egnn as orec_egnn

from . import util
from .golden.make_golden_cfgs import same_res_feature
from .test_gvp_gpu import GVP_ALL_ATOM
from .test_recegnn_gpu import RECEGNN_40KP
from .test_recenc_gpu import RECENC_40KP

pytestmark = pytest.mark.gpu
CUT = util.CUTOFFS_ALL_ATOM
TOL = 1e-4
B = 32
PAIRS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'size_pairs.json')))['all_atom']


def draw(batch):
    """(n_rec, n_lig) pairs of batch `batch`: 32 with the extremes of the histogram's support for batch 0, 16 for the others."""
    gen = torch.Generator().manual_seed(900 + batch)
    idx = torch.randperm(len(PAIRS['pairs']), generator=gen)[:B if batch == 0 else B // 2].tolist()
    pairs = [tuple(PAIRS['pairs'][i]) for i in idx]
    if batch == 0:
        (r_lo, r_hi), (l_lo, l_hi) = PAIRS['rec_bounds'], PAIRS['lig_bounds']
        pairs[3], pairs[11], pairs[20], pairs[29] = (r_lo, l_hi), (r_hi, l_lo), (r_hi, l_hi), (r_lo, l_lo)
    return [p[0] for p in pairs], [p[1] for p in pairs]


def test_size_pairs_are_the_shipped_distribution():
    assert PAIRS['rec_bounds'] == [7, 661] and PAIRS['lig_bounds'] == [2, 60] and len(PAIRS['pairs']) == 512
    n_rec, n_lig = draw(0)
    assert min(n_rec) == 7 and max(n_rec) == 661 and min(n_lig) == 2 and max(n_lig) == 60
    mean_rec = sum(p[0] for p in PAIRS['pairs']) / 512
    assert abs(mean_rec - PAIRS['mean_n_rec']) < 15                   # the draw follows the histogram (332.8)


@pytest.mark.parametrize('batch', [0, 1, 2])
@pytest.mark.parametrize('arch', ['egnn', 'gvp'])
def test_denoisers_at_shipped_sizes(cuda, arch, batch):
    n_rec, n_lig = draw(batch)
    if arch == 'egnn':
        cfg, model = util.EGNN_C2, LigRecDynamics(10, 10, graph_cutoffs=CUT, **util.EGNN_C2)
        g = util.fixed_encode(util.make_batch(n_rec, n_lig, seed=300 + batch))
        fwd = oegnn.egnn_dynamics_forward
    else:
        cfg, model = GVP_ALL_ATOM, LigRecDynamicsGVP(10, 10, graph_cutoffs=CUT, **GVP_ALL_ATOM)
        g = util.fixed_encode(util.make_batch(n_rec, n_lig, seed=300 + batch), n_vec=16)
        g.nodes['kp'].data['v_0'] = 0.5 * torch.randn(g.num_nodes('kp'), 16, 3, generator=torch.Generator().manual_seed(batch))
        fwd = ogvp.gvp_dynamics_forward
    synth.fill_state_dict_(model, 21 + batch)
    model.eval()
    nb = len(n_rec)
    t = (torch.arange(nb, dtype=torch.float32) + 1) / (nb + 1)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    rh, rx = fwd(sd, dict(cfg, graph_cutoffs=CUT), util.to_obatch(g), t)
    model = model.to(cuda)
    if hasattr(model, 'gemm_mode'):
        model.gemm_mode = 'f32'
    gd = g.to(cuda)
    with torch.no_grad():
        h, x = model(gd, t.to(cuda), G.get_batch_idxs(gd))
    torch.cuda.synchronize()
    util.assert_parity(h.cpu(), rh, n_lig, TOL, f'{arch} eps_h, batch {batch}')
    util.assert_parity(x.cpu(), rx, n_lig, TOL, f'{arch} eps_x, batch {batch}')


@pytest.mark.parametrize('batch', [0, 1, 2])
@pytest.mark.parametrize('enc', ['gvp', 'egnn'])
def test_learned_encoders_at_shipped_sizes(cuda, enc, batch):
    n_rec, _ = draw(batch)
    cfg = RECENC_40KP if enc == 'gvp' else RECEGNN_40KP
    kw = dict(cfg, graph_cutoffs=CUT)
    model = synth.fill_state_dict_((ReceptorEncoderGVP if enc == 'gvp' else ReceptorEncoder)(**kw), 81 + batch).eval()
    g = util.make_batch(n_rec, [4] * len(n_rec), seed=500 + batch, n_keypoints=cfg['n_keypoints'])
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    if enc == 'gvp':
        ref = orec_gvp.rec_encoder_gvp_forward(sd, kw, util.to_obatch(g))
    else:
        src, dst = g.edges(etype='rr')
        a = same_res_feature(src, dst)
        g.edges['rr'].data['same_res'] = a.bool()
        ref = orec_egnn.rec_encoder_egnn_forward(sd, kw, util.to_obatch(g), a if cfg['use_sameres_feat'] else None)
    model = model.to(cuda)
    gd = g.to(cuda)
    with torch.no_grad():
        out = model(gd, G.get_batch_idxs(gd))
    torch.cuda.synchronize()
    kp = out.nodes['kp'].data
    counts = [cfg['n_keypoints']] * len(n_rec)
    util.assert_parity(kp['x_0'].cpu(), ref.x['kp'], counts, TOL, f'{enc} encoder kp x_0, batch {batch}', atol_rel=1e-5)
    util.assert_parity(kp['h_0'].cpu(), ref.h['kp'], counts, TOL, f'{enc} encoder kp h_0, batch {batch}', atol_rel=1e-5)
    if enc == 'gvp':
        util.assert_parity(kp['v_0'].cpu().flatten(1), ref.v['kp'].flatten(1), counts, TOL, f'{enc} encoder kp v_0, batch {batch}', atol_rel=1e-5)
    rs, rd = out.edges(etype='rk')
    assert torch.equal(rs.cpu(), ref.edges['rk'][0]) and torch.equal(rd.cpu(), ref.edges['rk'][1])
