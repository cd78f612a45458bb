"""The CPU oracle against the golden vectors generated from the reference's own sub-modules
(tests/golden/make_golden.py).  Weights are re-created from the fill rule + seed, so these tests
also pin the state-dict layout of this package's modules to the reference's."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from keypoint_diffusion_amd import synth
from keypoint_diffusion_amd.dynamics import LigRecDynamics
from keypoint_diffusion_amd.dynamics_gvp import LigRecDynamicsGVP
from keypoint_diffusion_amd.receptor_encoder import ReceptorEncoder
from keypoint_diffusion_amd.receptor_encoder_gvp import ReceptorEncoderGVP
from oracle import diffusion as odiff
from oracle import egnn as oegnn
from oracle import gvp as ogvp
from oracle import rec_encoder as orec
from oracle import rec_encoder_egnn as orecegnn

from . import util
from .golden.make_golden_cfgs import GVP_CFGS, RECENC_CFGS, RECEGNN_CFGS, same_res_feature

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CUT = util.CUTOFFS_ALL_ATOM


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, name)).items()}


def close(a, b, tol=2e-5):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = util.rel_err(a, b)
    assert err < tol, f'rel err {err}'


def test_state_dict_layout_matches_reference():
    layout = json.load(open(os.path.join(GOLD, 'state_dict_layout.json')))
    mods = {
        'egnn_c2': LigRecDynamics(10, 10, graph_cutoffs=CUT, **util.EGNN_C2),
        'egnn_dev': LigRecDynamics(10, 20, graph_cutoffs=CUT, **util.EGNN_DEV),
        'gvp_kp': LigRecDynamicsGVP(10, 128, graph_cutoffs=CUT, **GVP_CFGS['gvp_kp']),
        'gvp_mean': LigRecDynamicsGVP(10, 10, graph_cutoffs=CUT, **GVP_CFGS['gvp_mean']),
        'gvp_norm0': LigRecDynamicsGVP(10, 10, graph_cutoffs=CUT, **GVP_CFGS['gvp_norm0']),
        'recenc_mean': ReceptorEncoderGVP(graph_cutoffs=CUT, **RECENC_CFGS['recenc_mean']),
        'recenc_norm10': ReceptorEncoderGVP(graph_cutoffs=CUT, **RECENC_CFGS['recenc_norm10']),
    }
    mods.update({tag: ReceptorEncoder(graph_cutoffs=CUT, **cfg) for tag, cfg in RECEGNN_CFGS.items()})
    for tag, m in mods.items():
        mine = {k: list(v.shape) for k, v in m.state_dict().items()}
        assert mine == layout[tag], tag
    # 348 tensors / 11 978 814 parameters for egnn_all_atom
    sd = mods['egnn_c2'].state_dict()
    assert len(sd) == 348 and sum(v.numel() for v in sd.values()) == 11978814


@pytest.mark.parametrize('tag,cfg,rec_nf', [('egnn_c2', util.EGNN_C2, 10), ('egnn_dev', util.EGNN_DEV, 20)])
def test_egnn_forward_and_blocks(tag, cfg, rec_nf):
    gd = load(f'{tag}.npz')
    model = synth.fill_state_dict_(LigRecDynamics(10, rec_nf, graph_cutoffs=CUT, **cfg), int(gd['seed']))
    sd = model.state_dict()
    ob = util.to_obatch(util.fixed_encode(util.make_batch(gd['n_rec'].tolist(), gd['n_lig'].tolist(), seed=5, n_rec_feat=rec_nf)))
    edges = {'ll': (gd['ll_src'], gd['ll_dst']), 'kl': (gd['kl_src'], gd['kl_dst']), 'lk': (gd['kl_dst'], gd['kl_src'])}
    # the oracle's own graph builders reproduce the stored lists
    mine = oegnn.lig_edges(ob, dict(cfg, graph_cutoffs=CUT))
    for et in ('ll', 'kl'):
        assert torch.equal(mine[et][0], edges[et][0]) and torch.equal(mine[et][1], edges[et][1])
    eps_h, eps_x = oegnn.egnn_dynamics_forward(sd, dict(cfg, graph_cutoffs=CUT), ob, gd['t'], edges=edges)
    close(eps_h, gd['eps_h'])
    close(eps_x, gd['eps_x'])
    # sub-module vectors
    lin = lambda p, x: F.linear(x, sd[p + '.weight'], sd.get(p + '.bias'))
    f, hc = gd['f'], gd['hc']
    etypes = ['ll', 'kl', 'lk', 'kk'] if cfg['update_kp_feat'] else ['ll', 'kl']
    for li in (0, cfg['n_layers'] - 1):
        pre = f'egnn.conv_layers.{li}'
        for et in etypes:
            m = F.silu(lin(f'{pre}.edge_mlp.{et}.2', F.silu(lin(f'{pre}.edge_mlp.{et}.0', f))))
            close(m, gd[f'L{li}_{et}_edge'])
            close(torch.sigmoid(lin(f'{pre}.soft_attention.{et}.0', m)), gd[f'L{li}_{et}_att'])
            c = F.silu(lin(f'{pre}.coord_mlp.{et}.2', F.silu(lin(f'{pre}.coord_mlp.{et}.0', f))))
            close(F.linear(c, sd[f'{pre}.coord_mlp.{et}.4.weight']), gd[f'L{li}_{et}_coord'])
        for nt in (['lig', 'kp'] if cfg['update_kp_feat'] else ['lig']):
            close(lin(f'{pre}.node_mlp.{nt}.2', F.silu(lin(f'{pre}.node_mlp.{nt}.0', hc))), gd[f'L{li}_{nt}_node'])
            close(F.layer_norm(hc[:, :257], (257,), sd[f'{pre}.layer_norm.{nt}.weight'], sd[f'{pre}.layer_norm.{nt}.bias'], 1e-5),
                  gd[f'L{li}_{nt}_ln'])


def test_gvp_primitives():
    from keypoint_diffusion_amd.dynamics_gvp import NoisePredictionBlock
    from keypoint_diffusion_amd.gvp import GVP, GVPLayerNorm
    gd = load('gvp_blocks.npz')
    for dmax in (3.5, 15.0, 100.0):
        close(ogvp.rbf(gd['d'], dmax), gd[f'rbf_{dmax}'], 1e-6)
    close(ogvp.norm_no_nan(gd['v']), gd['nnn'], 1e-6)
    close(ogvp.norm_no_nan(gd['v'], axis=-1, keepdims=True, sqrt=False), gd['nnn_sq_keep'], 1e-6)
    g0 = synth.fill_state_dict_(GVP(17, 16, 272, 256), 30)
    g1 = synth.fill_state_dict_(GVP(16, 1, 256, 64, vectors_activation=torch.nn.Identity()), 31)
    ln = synth.fill_state_dict_(GVPLayerNorm(256), 32)
    sd0 = {'g.' + k: v for k, v in g0.state_dict().items()}
    s1, v1 = ogvp.gvp(sd0, 'g', gd['s0'], gd['v'])
    close(s1, gd['s1']), close(v1, gd['v1'])
    sd1 = {'g.' + k: v for k, v in g1.state_dict().items()}
    s2, v2 = ogvp.gvp(sd1, 'g', gd['s1'], gd['v1'], vec_act='identity')
    close(s2, gd['s2']), close(v2, gd['v2'])
    sdl = {'n.' + k: v for k, v in ln.state_dict().items()}
    sl, vl = ogvp.gvp_layernorm(sdl, 'n', gd['s1'], gd['v1'])
    close(sl, gd['sl']), close(vl, gd['vl'])
    nb = synth.fill_state_dict_(NoisePredictionBlock(256, 10, 16, n_gvps=4), 40)
    sdn = {'b.' + k: v for k, v in nb.state_dict().items()}
    s, v = ogvp.gvp_chain(sdn, 'b.gvps', 4, gd['s1'], gd['v1'], last_vec_identity=True)
    close(F.linear(s, sdn['b.to_scalar_output.weight'], sdn['b.to_scalar_output.bias']), gd['ns'])
    close(v.squeeze(1), gd['nv'])


@pytest.mark.parametrize('tag', list(GVP_CFGS))
def test_gvp_dynamics_forward(tag):
    gd = load(f'{tag}.npz')
    cfg = dict(GVP_CFGS[tag], graph_cutoffs=CUT)
    model = synth.fill_state_dict_(LigRecDynamicsGVP(10, int(gd['n_kp_scalars']), **cfg), int(gd['seed']))
    ob = util.to_obatch(util.fixed_encode(util.make_batch(gd['n_rec'].tolist(), gd['n_lig'].tolist(), seed=8), n_vec=16))
    ob.v['kp'], ob.h['kp'] = gd['kp_v'], gd['kp_h']
    eps_h, eps_x = ogvp.gvp_dynamics_forward(model.state_dict(), cfg, ob, gd['t'])
    close(eps_h, gd['eps_h'])
    close(eps_x, gd['eps_x'])


@pytest.mark.parametrize('tag', list(RECENC_CFGS))
def test_receptor_encoder_forward(tag):
    gd = load(f'{tag}.npz')
    cfg = dict(RECENC_CFGS[tag], graph_cutoffs=CUT)
    model = synth.fill_state_dict_(ReceptorEncoderGVP(**cfg), int(gd['seed']))
    ob = util.to_obatch(util.make_batch(gd['n_rec'].tolist(), [4, 4], seed=17, n_keypoints=int(gd['n_keypoints'])))
    out = orec.rec_encoder_gvp_forward(model.state_dict(), cfg, ob)
    close(out.x['kp'], gd['kp_x'])
    close(out.h['kp'], gd['kp_s'])
    close(out.v['kp'], gd['kp_v'])


@pytest.mark.parametrize('tag', list(RECEGNN_CFGS))
def test_egnn_receptor_encoder_forward(tag):
    """models/receptor_encoder.py: oracle vs the composition of the reference's sub-modules."""
    gd = load(f'{tag}.npz')
    cfg = dict(RECEGNN_CFGS[tag], graph_cutoffs=CUT)
    model = synth.fill_state_dict_(ReceptorEncoder(**cfg), int(gd['seed']))
    ob = util.to_obatch(util.make_batch(gd['n_rec'].tolist(), [4, 4], seed=19, n_keypoints=int(gd['n_keypoints'])))
    a = same_res_feature(*ob.edges['rr']) if cfg['use_sameres_feat'] else None
    out, rec_h, rec_x = orecegnn.rec_encoder_egnn_forward(model.state_dict(), cfg, ob, a, return_rec=True)
    close(rec_h, gd['rec_h'])
    close(rec_x, gd['rec_x'])
    close(out.x['kp'], gd['kp_x'])
    close(out.h['kp'], gd['kp_h'])
    # the shipped egnn_20kp encoder: 68 tensors (fc_dst included although never applied upstream)
    if tag == 'recegnn_20kp':
        assert len(model.state_dict()) == 68


def test_noise_schedule_and_step_terms():
    gd = load('schedule.npz')
    for T in (100, 500, 1000):
        close(odiff.gamma_table(T, 1e-5), gd[f'gamma_{T}'], 1e-6)
    g1000 = odiff.gamma_table(1000, 1e-5)
    # SURVEY.md 8(c) probe values
    assert abs(float(g1000[0]) + 11.5129) < 1e-3 and abs(float(g1000[500]) + 0.25131) < 1e-4 and abs(float(g1000[1000]) - 11.5125) < 1e-3
    g = gd['g']
    s2, s, a = odiff.sigma_and_alpha_t_given_s(g[1:], g[:-1])
    close(s2, gd['sigma2_ts'], 1e-6), close(s, gd['sigma_ts'], 1e-6), close(a, gd['alpha_ts'], 1e-6)
    close(odiff.sigma(g), gd['sigma'], 1e-6), close(odiff.alpha(g), gd['alpha'], 1e-6)
    # the product's host-side schedule is the same table
    from keypoint_diffusion_amd.ligand_diffuser import PredefinedNoiseSchedule
    for T in (100, 500, 1000):
        close(PredefinedNoiseSchedule('polynomial_2', T, 1e-5).gamma.data, gd[f'gamma_{T}'], 1e-6)
