"""CPU restatement of the GVP layers and the GVP denoiser (test infrastructure only).

Follows models/gvp.py (_norm_no_nan :12-19, _rbf :26-41, GVP.forward :89-116,
GVPLayerNorm.forward :159-166, GVPEdgeConv :249-341, GVPMultiEdgeConv :459-551) and
models/dynamics_gvp.py (NoisePredictionBlock :38-44, LigRecGVP :46-101,
LigRecDynamicsGVP.forward :149-199).  Dropout is the identity (eval mode, gvp.py:133-134).
"""
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from . import graph_ops as G
from .batch import OBatch
from .egnn import SRC_DST, lig_edges

CANON = {'ll': 'lig_ll_lig', 'kl': 'kp_kl_lig', 'lk': 'lig_lk_kp', 'kk': 'kp_kk_kp'}


def norm_no_nan(x, axis=-1, keepdims=False, eps=1e-8, sqrt=True):
    out = torch.clamp(torch.sum(torch.square(x), axis, keepdims), min=eps)   # gvp.py:18
    return torch.sqrt(out) if sqrt else out


def rbf(d, d_max, d_count=16, d_min=0.0):
    mu = torch.linspace(d_min, d_max, d_count).view(1, -1)                   # gvp.py:35-41
    sigma = (d_max - d_min) / d_count
    return torch.exp(-((d.unsqueeze(-1) - mu) / sigma) ** 2)


def gvp(sd, p, s, v, vec_act='sigmoid', feat_act='silu'):
    """GVP.forward (gvp.py:89-116) with vector gating.  s [M,n], v [M,vin,3]."""
    Wh, Wu = sd[p + '.Wh'], sd[p + '.Wu']
    Vh = torch.einsum('bvc,vh->bhc', v, Wh)
    Vu = torch.einsum('bhc,hu->buc', Vh, Wu)
    sh = norm_no_nan(Vh)
    s_out = F.linear(torch.cat([s, sh], dim=1), sd[p + '.to_feats_out.0.weight'], sd[p + '.to_feats_out.0.bias'])
    if feat_act == 'silu':
        s_out = F.silu(s_out)
    gate = F.linear(s_out, sd[p + '.scalar_to_vector_gates.weight'], sd[p + '.scalar_to_vector_gates.bias'])
    gate = gate.unsqueeze(-1)
    if vec_act == 'sigmoid':
        gate = torch.sigmoid(gate)
    return s_out, gate * Vu


def gvp_chain(sd, p, n, s, v, last_vec_identity=False):
    for j in range(n):
        act = 'identity' if (last_vec_identity and j == n - 1) else 'sigmoid'
        s, v = gvp(sd, f'{p}.{j}', s, v, vec_act=act)
    return s, v


def gvp_layernorm(sd, p, s, v, eps=1e-5):
    """GVPLayerNorm.forward (gvp.py:159-166)."""
    s = F.layer_norm(s, (s.shape[1],), sd[p + '.feat_norm.weight'], sd[p + '.feat_norm.bias'], eps)
    vn = norm_no_nan(v, axis=-1, keepdims=True, sqrt=False)
    vn = torch.sqrt(torch.mean(vn, dim=-2, keepdim=True) + eps) + eps
    return s, v / vn


def edge_geometry(x_src, x_dst, src, dst, rbf_dmax, rbf_dim=16):
    x_diff = x_src[src] - x_dst[dst]                                          # gvp.py:474
    dij = norm_no_nan(x_diff, keepdims=True) + 1e-8                           # :478
    x_diff = x_diff / dij
    d = rbf(dij.squeeze(1), d_max=rbf_dmax, d_count=rbf_dim)                  # :480
    return x_diff, d


def multi_edge_conv(sd, p, etypes, edges, node, z_per_graph, cfg, rbf_dmax=15.0, masks=None):
    """GVPMultiEdgeConv.forward (gvp.py:459-538).  node: nt -> (s, x, v).  `masks`: training-mode GVPDropout
    (gvp.py:119-149, applied at :516 and :527) with given keep masks, (nt, position) -> (scalar mask [N,S], channel mask
    [N,V]) already scaled by 1 / (1 - rate); None = eval mode."""
    mn = cfg.get('message_norm', 10)
    use_mean = (mn == 'mean')                                                 # gvp.py:386-389
    dst_ntypes = sorted({SRC_DST[et][1] for et in etypes})
    agg_s = {nt: 0 for nt in dst_ntypes}
    agg_v = {nt: 0 for nt in dst_ntypes}
    for et in etypes:
        s_nt, d_nt = SRC_DST[et]
        src, dst = edges[et]
        x_diff, d = edge_geometry(node[s_nt][1], node[d_nt][1], src, dst, rbf_dmax)
        vec = torch.cat([x_diff.unsqueeze(1), node[s_nt][2][src]], dim=1)     # :545
        sc = torch.cat([node[s_nt][0][src], d], dim=1)                        # :547
        ms, mv = gvp_chain(sd, f'{p}.edge_message_fns.{CANON[et]}', cfg.get('n_message_gvps', 3), sc, vec)
        n_dst = node[d_nt][0].shape[0]
        red = G.scatter_mean if use_mean else G.scatter_sum
        agg_s[d_nt] = agg_s[d_nt] + red(ms, dst, n_dst)                       # :488-497 cross 'sum'
        agg_v[d_nt] = agg_v[d_nt] + red(mv, dst, n_dst)
    out = {}
    for nt in dst_ntypes:
        if use_mean:
            nv = 1.0
        elif mn == 0:
            nv = z_per_graph[nt].view(-1, 1)                                  # :504-507
        else:
            nv = mn
        s, x, v = node[nt]
        ms = agg_s[nt] / nv
        mv = agg_v[nt] / (nv.unsqueeze(-1) if isinstance(nv, torch.Tensor) else nv)
        if masks is not None:
            ms, mv = ms * masks[(nt, 0)][0], mv * masks[(nt, 0)][1].unsqueeze(-1)   # :516 dropout(scalar_msg, vec_msg)
        s, v = gvp_layernorm(sd, f'{p}.message_layer_norms.{nt}', s + ms, v + mv)   # :519-521
        rs, rv = gvp_chain(sd, f'{p}.node_update_fns.{nt}', cfg.get('n_update_gvps', 2), s, v)
        if masks is not None:
            rs, rv = rs * masks[(nt, 1)][0], rv * masks[(nt, 1)][1].unsqueeze(-1)   # :527
        s, v = gvp_layernorm(sd, f'{p}.update_layer_norms.{nt}', s + rs, v + rv)    # :530-532
        out[nt] = (s, x, v)
    return out


def gvp_dynamics_forward(sd: Dict[str, torch.Tensor], cfg: dict, batch: OBatch, t: torch.Tensor,
                         edges: Dict[str, tuple] = None, dropout_masks: Dict = None):
    """LigRecDynamicsGVP.forward (models/dynamics_gvp.py:149-199)."""
    lig_b = G.counts_to_batch_idx(batch.n['lig'])
    kp_b = G.counts_to_batch_idx(batch.n['kp'])
    V = cfg.get('vector_size', 16)
    update_kp = cfg.get('update_kp', False)
    n_convs = cfg.get('n_convs', 4)

    def enc(p, hh):                                                           # :124-134
        y = F.silu(F.linear(hh, sd[p + '.0.weight'], sd[p + '.0.bias']))
        return F.layer_norm(y, (y.shape[1],), sd[p + '.2.weight'], sd[p + '.2.bias'], 1e-5)

    ls = enc('lig_encoder', torch.cat([batch.h['lig'], t[lig_b].view(-1, 1)], dim=1))   # :161-169
    ks = enc('kp_encoder', torch.cat([batch.h['kp'], t[kp_b].view(-1, 1)], dim=1))
    node = {
        'lig': (ls, batch.x['lig'], torch.zeros(ls.shape[0], V, 3)),          # :179-184
        'kp': (ks, batch.x['kp'], batch.v['kp']),
    }
    if edges is None:
        edges = lig_edges(batch, cfg)
    edges = dict(edges)
    edges['kk'] = batch.edges.get('kk', (torch.zeros(0, dtype=torch.long),) * 2)

    all_et = ['ll', 'kl', 'lk', 'kk']
    # per-graph average in-degree + 1 (only used when message_norm == 0, gvp.py:504-507)
    z = {}
    bidx = {'lig': lig_b, 'kp': kp_b}

    for i in range(n_convs):
        if not update_kp or i == n_convs - 1:                                 # dynamics_gvp.py:67-72
            etypes = ['ll', 'kl']
        else:
            etypes = all_et
        if cfg.get('message_norm', 10) == 0:
            for nt in {SRC_DST[e][1] for e in etypes}:
                tot = torch.zeros(batch.batch_size, dtype=torch.long)
                for et in etypes:
                    if SRC_DST[et][1] == nt:
                        tot = tot + G.edges_per_graph(edges[et][1], batch.n[nt])
                zz = tot.to(torch.float32) / batch.n[nt].to(torch.float32) + 1
                z[nt] = zz[bidx[nt]]
        masks = None if dropout_masks is None else {k[1:]: v for k, v in dropout_masks.items() if k[0] == i}
        new = multi_edge_conv(sd, f'noise_predictor.conv_layers.{i}', etypes, edges, node, z, cfg, masks=masks)
        node = {**node, **new}

    s, _, v = node['lig']
    p = 'noise_predictor.noise_predictor'
    s, v = gvp_chain(sd, p + '.gvps', cfg.get('n_noise_gvps', 3), s, v, last_vec_identity=True)   # :14-33
    eps_h = F.linear(s, sd[p + '.to_scalar_output.weight'], sd[p + '.to_scalar_output.bias'])
    eps_x = v.squeeze(1)                                                      # :43
    return eps_h, eps_x
