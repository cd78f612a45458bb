#!/usr/bin/env python
"""Generate the golden fixtures that pin the CPU oracle to the reference implementation.

Run in the build container only (needs /root/reference, which never travels to the GPU box):
    python tests/golden/make_golden.py
The reference is pure Python but depends on third-party packages that are not installed here
(dgl, torch_cluster, torch_scatter, ot, openbabel).  They are replaced by EMPTY placeholder modules
(names only, no behaviour), which is enough to import the reference's model files and to construct
and run every sub-module that is plain PyTorch (nn.Sequential MLPs, GVP, GVPLayerNorm, _rbf,
NoisePredictionBlock, PredefinedNoiseSchedule, sigma/alpha helpers).  `forward`s that go through
DGL cannot run; for those the fixture is *composed*: the reference's own sub-modules are applied,
in the order of the reference's forward text, to tensors gathered with explicit index operations
(index_select / index_add_ standing in for DGL's u_sub_v, copy_e+sum, readout_nodes as documented).

Weights are never stored: reference modules and this package's modules are filled by the same
name-keyed rule (keypoint_diffusion_amd.synth.fill_state_dict_), so the fixtures also pin the
state-dict key/shape layout.  Output: tests/golden/*.npz (+ state_dict_layout.json).
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)


def _placeholders():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    dgl = mod('dgl', DGLHeteroGraph=object, DGLGraph=object, heterograph=object)
    dgl.function = mod('dgl.function', sum=None, mean=None)
    mod('dgl.nn')
    mod('dgl.nn.functional', edge_softmax=None)
    mod('torch_cluster', radius=None, radius_graph=None, knn=None, knn_graph=None)
    mod('torch_scatter', segment_csr=None, segment_coo=None)
    mod('ot')
    mod('openbabel')


_placeholders()
sys.path.insert(0, REF)
from models.dynamics import LigRecDynamics as RefEGNN                      # noqa: E402
from models.dynamics_gvp import LigRecDynamicsGVP as RefGVPDyn             # noqa: E402
from models.dynamics_gvp import NoisePredictionBlock as RefNoiseBlock      # noqa: E402
from models.gvp import GVP as RefGVP, GVPLayerNorm as RefGVPLN, _rbf as ref_rbf, _norm_no_nan as ref_nnn   # noqa: E402
from models.receptor_encoder_gvp import ReceptorEncoderGVP as RefRecEnc    # noqa: E402
from models.receptor_encoder import ReceptorEncoder as RefRecEgnn          # noqa: E402
from models.ligand_diffuser import PredefinedNoiseSchedule as RefSchedule, KeypointDiffusion as RefKD    # noqa: E402

from keypoint_diffusion_amd import synth                                    # noqa: E402
from keypoint_diffusion_amd.dynamics import LigRecDynamics                  # noqa: E402
from keypoint_diffusion_amd.dynamics_gvp import LigRecDynamicsGVP           # noqa: E402
from keypoint_diffusion_amd.receptor_encoder_gvp import ReceptorEncoderGVP  # noqa: E402
from keypoint_diffusion_amd.receptor_encoder import ReceptorEncoder         # noqa: E402
from oracle import graph_ops as G                                           # noqa: E402
from oracle import egnn as oegnn                                            # noqa: E402
from tests import util                                                      # noqa: E402
from tests.golden.make_golden_cfgs import GVP_CFGS, RECENC_CFGS, RECEGNN_CFGS, same_res_feature   # noqa: E402

CUT = util.CUTOFFS_ALL_ATOM
layout = {}


def check_layout(tag, ref, mine):
    a = {k: list(v.shape) for k, v in ref.state_dict().items()}
    b = {k: list(v.shape) for k, v in mine.state_dict().items()}
    assert a == b, f'{tag}: state-dict layout differs: {set(a) ^ set(b)}'
    layout[tag] = a


def npz(name, **arrs):
    out = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(f'wrote {name}: {sum(v.nbytes for v in out.values()) / 1024:.0f} KiB uncompressed')


def small_batch(n_rec, n_lig, seed, rec_nf=10, v=None):
    g = util.fixed_encode(util.make_batch(n_rec, n_lig, seed=seed, n_rec_feat=rec_nf), n_vec=v)
    return g, util.to_obatch(g)


SRC_DST = oegnn.SRC_DST


# --------------------------------------------------------------------------------------------
# EGNN: composed forward of the reference's own sub-modules (models/dynamics.py:342-385)
# --------------------------------------------------------------------------------------------
def egnn_composed(ref, cfg, ob, t, edges):
    upd = cfg['update_kp_feat']
    etypes = ['ll', 'kl', 'lk', 'kk'] if upd else ['ll', 'kl']
    updated = ['lig', 'kp'] if upd else ['lig']
    lig_b, kp_b = G.counts_to_batch_idx(ob.n['lig']), G.counts_to_batch_idx(ob.n['kp'])
    lig_feat = ref.lig_encoder(ob.h['lig'])                                  # :355-356
    kp_feat = ref.rec_encoder(ob.h['kp'])
    lig_feat = torch.cat([lig_feat, t[lig_b].view(-1, 1)], dim=1)            # :359-363
    kp_feat = torch.cat([kp_feat, t[kp_b].view(-1, 1)], dim=1)
    h = {'lig': lig_feat, 'kp': kp_feat}
    x = {'lig': ob.x['lig'], 'kp': ob.x['kp']}
    z = {}
    for nt in updated:                                                       # :277-285
        if cfg['message_norm'] == 0:
            tot = sum(G.edges_per_graph(edges[et][1], ob.n[nt]) for et in etypes if SRC_DST[et][1] == nt)
            z[nt] = (tot / ob.n[nt])[{'lig': lig_b, 'kp': kp_b}[nt]].view(-1, 1) + 1
        else:
            z[nt] = cfg['message_norm']
    for layer in ref.egnn.conv_layers:
        h_neigh = {nt: torch.zeros_like(h[nt]) for nt in updated}
        x_neigh = {nt: torch.zeros_like(x[nt]) for nt in updated}
        for et in etypes:
            s_nt, d_nt = SRC_DST[et]
            src, dst = edges[et]
            x_diff = x[s_nt][src] - x[d_nt][dst]                             # :160
            dij = torch.linalg.vector_norm(x_diff, dim=1).unsqueeze(-1)      # :211
            x_diff = x_diff / (dij + 1)                                      # :169
            f = torch.cat([h[s_nt][src], h[d_nt][dst], dij], dim=-1)         # :103-105
            msg_h = layer.edge_mlp[et](f)                                    # :111
            msg_h = msg_h * layer.soft_attention[et](msg_h)                  # :112
            msg_x = torch.tanh(layer.coord_mlp[et](f)) * x_diff * layer.coords_range if layer.use_tanh \
                else layer.coord_mlp[et](f) * x_diff                         # :118-120 (:115 is never true)
            h_neigh[d_nt].index_add_(0, dst, msg_h)                          # :177-185
            x_neigh[d_nt].index_add_(0, dst, msg_x)
        hn, xn = {}, {}
        for nt in updated:
            inp = torch.cat([h[nt], h_neigh[nt] / z[nt]], dim=1)             # :188-202
            hn[nt] = layer.layer_norm[nt](h[nt] + layer.node_mlp[nt](inp))   # :203-204
            xn[nt] = x[nt] + x_neigh[nt] / z[nt]                             # :206
        h = {'lig': hn['lig'], 'kp': hn.get('kp', kp_feat)}
        x = {'lig': xn['lig'], 'kp': xn.get('kp', ob.x['kp'])}
    eps_h = ref.lig_decoder(h['lig'][:, :-1])                                # :376-380
    return eps_h, x['lig'] - ob.x['lig']


def make_egnn():
    for tag, cfg, n_rec, n_lig, rec_nf in [('egnn_c2', util.EGNN_C2, [40, 23], [9, 5], 10),
                                           ('egnn_dev', util.EGNN_DEV, [31], [8], 20)]:
        kw = dict(cfg, graph_cutoffs=CUT)
        ref, mine = RefEGNN(10, rec_nf, **kw), LigRecDynamics(10, rec_nf, **kw)
        check_layout(tag, ref, mine)
        synth.fill_state_dict_(ref, 21)
        g, ob = small_batch(n_rec, n_lig, seed=5, rec_nf=rec_nf)
        t = torch.linspace(0.2, 0.9, len(n_rec))
        edges = oegnn.lig_edges(ob, kw)
        edges['kk'] = ob.edges['kk']
        with torch.no_grad():
            eps_h, eps_x = egnn_composed(ref.eval(), kw, ob, t, edges)
            # direct sub-module vectors (layer 0 / last layer)
            gen = torch.Generator().manual_seed(3)
            f = torch.randn(17, 515, generator=gen)
            hc = torch.randn(7, 514, generator=gen)
            blocks = {}
            for li in (0, cfg['n_layers'] - 1):
                L = ref.egnn.conv_layers[li]
                for et in L.edge_types:
                    m = L.edge_mlp[et](f)
                    blocks[f'L{li}_{et}_edge'] = m
                    blocks[f'L{li}_{et}_att'] = L.soft_attention[et](m)
                    blocks[f'L{li}_{et}_coord'] = L.coord_mlp[et](f)
                for nt in L.updated_node_types:
                    blocks[f'L{li}_{nt}_node'] = L.node_mlp[nt](hc)
                    blocks[f'L{li}_{nt}_ln'] = L.layer_norm[nt](hc[:, :257])
        npz(f'{tag}.npz', seed=21, n_rec=n_rec, n_lig=n_lig, rec_nf=rec_nf, t=t, eps_h=eps_h, eps_x=eps_x,
            ll_src=edges['ll'][0], ll_dst=edges['ll'][1], kl_src=edges['kl'][0], kl_dst=edges['kl'][1],
            f=f, hc=hc, **blocks)


# --------------------------------------------------------------------------------------------
# GVP primitives: the reference functions run natively
# --------------------------------------------------------------------------------------------
def make_gvp_blocks():
    gen = torch.Generator().manual_seed(9)
    out = {}
    d = torch.rand(23, generator=gen) * 12
    for dmax in (3.5, 15.0, 100.0):
        out[f'rbf_{dmax}'] = ref_rbf(d, D_max=dmax, D_count=16)
    v = torch.randn(11, 17, 3, generator=gen)
    v[3] = 0
    out['nnn'] = ref_nnn(v)
    out['nnn_sq_keep'] = ref_nnn(v, axis=-1, keepdims=True, sqrt=False)
    s0 = torch.randn(11, 272, generator=gen)
    g0 = RefGVP(dim_vectors_in=17, dim_vectors_out=16, dim_feats_in=272, dim_feats_out=256)
    g1 = RefGVP(dim_vectors_in=16, dim_vectors_out=1, dim_feats_in=256, dim_feats_out=64,
                vectors_activation=torch.nn.Identity())
    ln = RefGVPLN(256)
    for i, m in enumerate((g0, g1, ln)):
        synth.fill_state_dict_(m, 30 + i)
    with torch.no_grad():
        s1, v1 = g0((s0, v))
        s2, v2 = g1((s1, v1))
        sl, vl = ln(s1, v1)
        nb = RefNoiseBlock(in_scalar_dim=256, out_scalar_dim=10, vector_size=16, n_gvps=4)
        synth.fill_state_dict_(nb, 40)
        ns, nv = nb((s1, None, v1))
    npz('gvp_blocks.npz', d=d, v=v, s0=s0, s1=s1, v1=v1, s2=s2, v2=v2, sl=sl, vl=vl, ns=ns, nv=nv, **out)


# --------------------------------------------------------------------------------------------
# GVP dynamics: composed forward (models/dynamics_gvp.py:149-199, models/gvp.py:459-551)
# --------------------------------------------------------------------------------------------
def ref_rbf_geom(xs, xd, src, dst, dmax):
    x_diff = xs[src] - xd[dst]
    dij = ref_nnn(x_diff, keepdims=True) + 1e-8
    return x_diff / dij, ref_rbf(dij.squeeze(1), D_max=dmax, D_count=16)


def gvp_composed(ref, cfg, ob, t, edges):
    lig_b, kp_b = G.counts_to_batch_idx(ob.n['lig']), G.counts_to_batch_idx(ob.n['kp'])
    ls = ref.lig_encoder(torch.cat([ob.h['lig'], t[lig_b].view(-1, 1)], dim=1))      # :161-169
    ks = ref.kp_encoder(torch.cat([ob.h['kp'], t[kp_b].view(-1, 1)], dim=1))
    node = {'lig': (ls, ob.x['lig'], torch.zeros(ls.shape[0], cfg['vector_size'], 3)),
            'kp': (ks, ob.x['kp'], ob.v['kp'])}
    bidx = {'lig': lig_b, 'kp': kp_b}
    for conv in ref.noise_predictor.conv_layers:
        ets = [e[1] for e in conv.etypes]
        mean = cfg['message_norm'] == 'mean'
        agg_s, agg_v = {}, {}
        for et in ets:
            s_nt, d_nt = SRC_DST[et]
            src, dst = edges[et]
            x_diff, d = ref_rbf_geom(node[s_nt][1], node[d_nt][1], src, dst, conv.rbf_dmax)      # gvp.py:472-480
            vec = torch.cat([x_diff.unsqueeze(1), node[s_nt][2][src]], dim=1)                    # :545
            sc = torch.cat([node[s_nt][0][src], d], dim=1)                                       # :547
            ms, mv = conv.edge_message_fns['_'.join((s_nt, et, d_nt))]((sc, vec))                # :549
            n_dst = node[d_nt][0].shape[0]
            ss = torch.zeros(n_dst, ms.shape[1]).index_add_(0, dst, ms)
            vv = torch.zeros(n_dst, *mv.shape[1:]).index_add_(0, dst, mv)
            if mean:
                deg = torch.zeros(n_dst).index_add_(0, dst, torch.ones(dst.shape[0])).clamp(min=1)
                ss, vv = ss / deg.view(-1, 1), vv / deg.view(-1, 1, 1)
            agg_s[d_nt] = agg_s.get(d_nt, 0) + ss                                                # cross_reducer sum
            agg_v[d_nt] = agg_v.get(d_nt, 0) + vv
        new = {}
        for nt in sorted({SRC_DST[e][1] for e in ets}):
            nv = conv.norm_values[nt]
            if nv == 0:
                tot = sum(G.edges_per_graph(edges[et][1], ob.n[nt]) for et in ets if SRC_DST[et][1] == nt)
                nv = (tot / ob.n[nt] + 1)[bidx[nt]].unsqueeze(1)                                 # :504-507
            s, x, v = node[nt]
            sm = agg_s[nt] / nv
            vm = agg_v[nt] / (nv.unsqueeze(-1) if isinstance(nv, torch.Tensor) else nv)
            s, v = conv.message_layer_norms[nt](s + sm, v + vm)                                  # :519-521
            rs, rv = conv.node_update_fns[nt]((s, v))                                            # :524
            s, v = conv.update_layer_norms[nt](s + rs, v + rv)                                   # :530-532
            new[nt] = (s, x, v)
        node = {**node, **new}
    return ref.noise_predictor.noise_predictor(node['lig'])                                      # dynamics_gvp.py:99




def make_gvp_dyn():
    for tag, cfg in GVP_CFGS.items():
        kw = dict(cfg, graph_cutoffs=CUT)
        n_kp_scalars = 128 if tag == 'gvp_kp' else 10
        ref, mine = RefGVPDyn(10, n_kp_scalars, **kw), LigRecDynamicsGVP(10, n_kp_scalars, **kw)
        check_layout(tag, ref, mine)
        synth.fill_state_dict_(ref, 51)
        g, ob = small_batch([26, 19], [7, 10], seed=8, v=16)
        gen = torch.Generator().manual_seed(2)
        ob.v['kp'] = 0.5 * torch.randn(ob.x['kp'].shape[0], 16, 3, generator=gen)
        if n_kp_scalars != 10:
            ob.h['kp'] = torch.randn(ob.x['kp'].shape[0], n_kp_scalars, generator=gen)
        t = torch.tensor([0.35, 0.8])
        edges = oegnn.lig_edges(ob, kw)
        edges['kk'] = ob.edges['kk']
        with torch.no_grad():
            eps_h, eps_x = gvp_composed(ref.eval(), kw, ob, t, edges)
        npz(f'{tag}.npz', seed=51, n_rec=[26, 19], n_lig=[7, 10], n_kp_scalars=n_kp_scalars, t=t, kp_v=ob.v['kp'],
            kp_h=ob.h['kp'], eps_h=eps_h, eps_x=eps_x)


# --------------------------------------------------------------------------------------------
# GVP receptor encoder: composed forward (models/receptor_encoder_gvp.py:212-294)
# --------------------------------------------------------------------------------------------
def edge_conv_composed(conv, src, dst, src_feats, dst_feats, z, mean):
    s_s, x_s, v_s = src_feats
    s_d, x_d, v_d = dst_feats
    x_diff, d = ref_rbf_geom(x_s, x_d, src, dst, conv.rbf_dmax)                      # gvp.py:275-281
    vec = [x_diff.unsqueeze(1), v_s[src]]
    sc = [s_s[src], d]
    if conv.use_dst_feats:                                                           # :323-337
        vec.append(v_d[dst])
        sc.append(s_d[dst])
    ms, mv = conv.edge_message((torch.cat(sc, 1), torch.cat(vec, 1)))
    n_dst = s_d.shape[0]
    ss = torch.zeros(n_dst, ms.shape[1]).index_add_(0, dst, ms)
    vv = torch.zeros(n_dst, *mv.shape[1:]).index_add_(0, dst, mv)
    if mean:
        deg = torch.zeros(n_dst).index_add_(0, dst, torch.ones(dst.shape[0])).clamp(min=1)
        ss, vv = ss / deg.view(-1, 1), vv / deg.view(-1, 1, 1)
    ss = ss / z                                                                      # :303-306
    vv = vv / (z.unsqueeze(-1) if isinstance(z, torch.Tensor) else z)
    s, v = conv.message_layer_norm(s_d + ss, v_d + vv)
    rs, rv = conv.node_update((s, v))
    return conv.update_layer_norm(s + rs, v + rv)


def make_rec_encoder():
    cfgs = RECENC_CFGS
    for tag, cfg in cfgs.items():
        kw = dict(cfg, graph_cutoffs=CUT)
        ref, mine = RefRecEnc(**kw), ReceptorEncoderGVP(**kw)
        check_layout(tag, ref, mine)
        synth.fill_state_dict_(ref, 61)
        ref.eval()
        n_rec = [33, 21]
        g = util.make_batch(n_rec, [4, 4], seed=17, n_keypoints=cfg['n_keypoints'])
        ob = util.to_obatch(g)
        K, S, V = cfg['n_keypoints'], 128, 16
        mean = cfg['message_norm'] == 'mean'
        with torch.no_grad():
            s = ref.scalar_norm(ref.scalar_embed(ob.h['rec']))                       # :221-222
            v = torch.zeros(s.shape[0], V, 3)
            x = ob.x['rec']
            rr_src, rr_dst = ob.edges['rr']
            z = 1 if mean else cfg['message_norm']
            for conv in ref.rr_conv_layers:
                s, v = edge_conv_composed(conv, rr_src, rr_dst, (s, x, v), (s, x, v), z, mean)
            ki = ref.keypoint_initializer                                            # :40-93
            meanf = G.segment_mean_nodes(s, ob.n['rec'])
            kp_emb = ki.keypoint_embedding(meanf).reshape(-1, S)
            ft_src, ft_dst = ki.src_net(s), ki.dst_net(kp_emb)
            rp = G.counts_to_ptr(ob.n['rec'])
            pos = []
            for b in range(len(n_rec)):
                a = torch.exp((ft_dst[b * K:(b + 1) * K] @ ft_src[rp[b]:rp[b + 1]].T) / S ** 0.5)
                pos.append((a / a.sum(1, keepdim=True)) @ x[rp[b]:rp[b + 1]])
            kp_x = torch.cat(pos)
            n_kp = torch.full((len(n_rec),), K)
            kp_idx, rec_idx = G.knn(x, kp_x, cfg['k_closest'], ob.n['rec'], n_kp)    # :302-316
            kp_s, kp_v = torch.zeros(len(n_rec) * K, S), torch.zeros(len(n_rec) * K, V, 3)
            for conv in ref.rk_conv_layers:
                kp_s, kp_v = edge_conv_composed(conv, rec_idx, kp_idx, (s, x, v), (kp_s, kp_x, kp_v), z, mean)
        npz(f'{tag}.npz', seed=61, n_rec=n_rec, n_keypoints=K, kp_x=kp_x, kp_s=kp_s, kp_v=kp_v, rec_s=s, rec_v=v)


# --------------------------------------------------------------------------------------------
# EGNN receptor encoder: composed forward of the reference's own sub-modules
# (models/receptor_encoder.py:98-154 ReceptorConv, :182-297 RecKeyConv, :483-555 ReceptorEncoder)
# --------------------------------------------------------------------------------------------
def make_rec_encoder_egnn():
    for tag, cfg in RECEGNN_CFGS.items():
        kw = dict(cfg, graph_cutoffs=CUT)
        ref, mine = RefRecEgnn(**kw), ReceptorEncoder(**kw)
        check_layout(tag, ref, mine)
        synth.fill_state_dict_(ref, 71)
        ref.eval()
        n_rec = [33, 21]
        K, D, k = cfg['n_keypoints'], cfg['out_n_node_feat'], cfg['k_closest']
        g = util.make_batch(n_rec, [4, 4], seed=19, n_keypoints=K)
        ob = util.to_obatch(g)
        src, dst = ob.edges['rr']
        a = same_res_feature(src, dst) if cfg['use_sameres_feat'] else None
        rec_b = G.counts_to_batch_idx(ob.n['rec'])
        with torch.no_grad():
            h, x = ob.h['rec'], ob.x['rec']
            if cfg['message_norm'] == 0:                                             # :505-509
                z = (G.edges_per_graph(dst, ob.n['rec']).float() / ob.n['rec'].float())[rec_b].view(-1, 1)
            else:
                z = cfg['message_norm']
            for conv in ref.rec_convs:                                               # ReceptorConv.forward :98-154
                x_diff = x[src] - x[dst]
                radial = torch.norm(x_diff, dim=1).unsqueeze(-1)
                x_diff = x_diff / (radial + 1)
                f = torch.cat([h[src], h[dst], radial] + ([a] if a is not None else []), dim=-1)
                msg_h = conv.edge_mlp(f)
                msg_h = msg_h * conv.soft_attention(msg_h)
                h_neigh = torch.zeros(h.shape[0], msg_h.shape[1]).index_add_(0, dst, msg_h) / z
                if conv.fix_pos:
                    x_new = x
                else:
                    msg_x = torch.tanh(conv.coord_mlp(f)) * x_diff * conv.coords_range if conv.use_tanh else conv.coord_mlp(f) * x_diff
                    x_new = x + torch.zeros_like(x).index_add_(0, dst, msg_x) / z
                h = conv.layer_norm(conv.node_mlp(torch.cat([h, h_neigh], dim=-1)))
                x = x_new
            meanf = G.segment_mean_nodes(h, ob.n['rec'])                             # :526
            kp_h0 = ref.keypoint_embedding(meanf).reshape(-1, D)                     # :529-530
            rk = ref.rec_kp_conv                                                     # RecKeyConv.forward :182-236
            ft_src, ft_dst = rk.fc_src(h), rk.fc_src(kp_h0)
            x_val = ob.x['rec'] if cfg['fix_pos'] else x
            rp = G.counts_to_ptr(ob.n['rec'])
            pos = []
            for b in range(len(n_rec)):
                att = torch.exp((ft_dst[b * K:(b + 1) * K] @ ft_src[rp[b]:rp[b + 1]].T) / rk.out_feats ** 0.5)
                pos.append((att / att.sum(1, keepdim=True)) @ x_val[rp[b]:rp[b + 1]])
            kp_x = torch.cat(pos)
            n_kp = torch.full((len(n_rec),), K)
            kp_idx, rec_idx = G.knn(ob.x['rec'], kp_x, k, ob.n['rec'], n_kp)         # k_closest_feats :257-291
            h_m = torch.zeros(kp_x.shape[0], D).index_add_(0, kp_idx, h[rec_idx]) / k
            d_k = torch.norm(ob.x['rec'][rec_idx] - kp_x[kp_idx] + 1e-30, dim=1).view(-1, k)
            kp_h = rk.layer_norm(rk.kp_feature_mlp(torch.cat([h_m, d_k], dim=1)))
        npz(f'{tag}.npz', seed=71, n_rec=n_rec, n_keypoints=K, kp_x=kp_x, kp_h=kp_h, rec_h=h, rec_x=x)


def make_schedule():
    out = {}
    for T in (100, 500, 1000):
        out[f'gamma_{T}'] = RefSchedule('polynomial_2', timesteps=T, precision=1e-5).gamma.data
    g = torch.linspace(-10, 10, 41)
    s2, s, a = RefKD.sigma_and_alpha_t_given_s(None, g[1:], g[:-1])
    out.update(g=g, sigma2_ts=s2, sigma_ts=s, alpha_ts=a, sigma=RefKD.sigma(None, g), alpha=RefKD.alpha(None, g))
    npz('schedule.npz', **out)


def xyz_cases():
    """Coordinates that exercise the text formatting: exact ties of the third decimal (round-half-even), values that
    round to zero with a sign, denormals, large magnitudes, non-finite values, plus ordinary samples."""
    g = torch.Generator().manual_seed(77)
    special = torch.tensor([0.0, -0.0, 0.0005, -0.0005, 0.0625, 0.1875, -0.0625, 2.5, 1e-4, -1e-4, 4.9e-4, 5.1e-4,
                            0.9995, 0.99951, 9.9995, 99.9995, 999.9995, 1e-45, -1e-45, 1.17549435e-38, 123456.789, -98765.4375,
                            1.0e7, 16777216.0, 3.0e9, 8.5e15, -8.9e15, float('inf'), -float('inf'), float('nan'),
                            0.0015, 0.0025, 0.0035, 0.0045, 1.0005, 1.0015, 2.0005, 0.4375, 0.3125, -7.0625], dtype=torch.float32)
    ties = (torch.randint(-80000, 80000, (120,), generator=g).float() * 2 + 1) / 16      # odd/16: many exact .xxx5 ties
    rnd = torch.randn(300, generator=g) * torch.tensor([0.01, 1.0, 30.0]).repeat(100)
    vals = torch.cat([special, ties, rnd])
    vals = vals[: (vals.numel() // 3) * 3]
    pos = vals.view(-1, 3)
    n = pos.shape[0]
    feat = torch.randn(n, 10, generator=g)
    feat[3] = 0.0                       # all equal: first index
    feat[4, 2] = feat[4, 7] = 5.0       # tie of the maximum: first of them
    feat[5, 6] = float('nan')           # NaN is the maximum for torch.argmax
    feat[6, 9] = float('inf')
    sizes = [1, 2, 9, 10, 11, 25, 0, 30]
    sizes.append(n - sum(sizes))
    return pos, feat, sizes


def make_xyz():
    from utils import write_xyz_file as ref_write_xyz                       # /root/reference/utils.py:11-21
    elements = ['C', 'N', 'O', 'S', 'P', 'F', 'Cl', 'Br', 'I', 'B']         # configs/dev_config.yml:19
    pos, feat, sizes = xyz_cases()
    text, ptr, elem = b'', [0], []
    a = 0
    for n in sizes:
        p, f = pos[a:a + n], feat[a:a + n]
        idxs = torch.argmax(f, dim=1).tolist() if n else []                # sample.py:77
        els = [elements[i] for i in idxs]                                   # crossdocked/dataset.py:147-149
        text += ref_write_xyz(p, els).encode()
        ptr.append(len(text))
        elem += idxs
        a += n
    npz('xyz.npz', pos=pos, feat=feat, sizes=np.asarray(sizes, np.int64), elem=np.asarray(elem, np.int64),
        text=np.frombuffer(text, np.uint8), text_ptr=np.asarray(ptr, np.int64))


def make_full_models():
    """The YAML boundary (model_setup.py:4-63): the REFERENCE's `model_from_config` run on its nine shipped configurations
    (trained_models/*/config.yml, configs/dev_config.yml).  Stored per configuration: the parsed configuration values (the input)
    and key -> shape of the full `KeypointDiffusion` state dict the reference builds from them (the expected output).
    dev_config names a dataset directory that is not shipped (data/bindingmoad_dev/); only for the ligand-size histogram file the
    constructor opens, it is pointed at the shipped data/bindingmoad_processed/ -- no tensor shape depends on that file."""
    import copy
    import glob
    import yaml
    from model_setup import model_from_config as ref_model_from_config
    out = {}
    files = sorted(glob.glob(os.path.join(REF, 'trained_models', '*', 'config.yml'))) + [os.path.join(REF, 'configs', 'dev_config.yml')]
    cwd = os.getcwd()
    os.chdir(REF)                                   # dataset locations in the configs are relative to the repository root
    try:
        for f in files:
            name = os.path.basename(os.path.dirname(f)) if f.endswith('config.yml') and 'trained_models' in f else 'dev_config'
            cfg = yaml.safe_load(open(f))
            given = copy.deepcopy(cfg)
            if name == 'dev_config':
                cfg['dataset']['location'] = 'data/bindingmoad_processed/'
            model = ref_model_from_config(cfg)          # (mutates cfg: in_scalar_size / in_n_node_feat are written into it)
            sd = model.state_dict()
            out[name] = {'config': given, 'layout': {k: list(v.shape) for k, v in sd.items()},
                         'n_tensors': len(sd), 'n_params': int(sum(p.numel() for p in model.parameters()))}
            print(f'{name}: {len(sd)} tensors, {out[name]["n_params"]} parameters')
    finally:
        os.chdir(cwd)
    with open(os.path.join(HERE, 'full_model_layouts.json'), 'w') as fh:
        json.dump(out, fh, indent=0, sort_keys=True)


if __name__ == '__main__':
    torch.manual_seed(0)
    makers = dict(full_models=make_full_models, xyz=make_xyz, egnn=make_egnn, gvp_blocks=make_gvp_blocks, gvp_dyn=make_gvp_dyn, rec_encoder=make_rec_encoder,
                  rec_encoder_egnn=make_rec_encoder_egnn, schedule=make_schedule)
    only = sys.argv[1:]                 # e.g. `make_golden.py rec_encoder_egnn`: regenerate one family, keep the others
    lp = os.path.join(HERE, 'state_dict_layout.json')
    if only and os.path.exists(lp):
        layout.update(json.load(open(lp)))
    for name, fn in makers.items():
        if not only or name in only:
            fn()
    with open(lp, 'w') as f:
        json.dump(layout, f, indent=0, sort_keys=True)
    print('layouts:', {k: len(v) for k, v in layout.items()})
