"""CPU restatement of the EGNN denoiser (test infrastructure only; see oracle/__init__.py).

Follows models/dynamics.py: LigRecDynamics.forward :342-385, add_lig_edges :387-420,
LigRecEGNN.forward :266-294, LigRecConv.forward/message :89-217.
`sd` is the state_dict of the `dynamics` sub-module (keys as in the reference).
"""
from typing import Dict

import torch
import torch.nn.functional as F

from . import graph_ops as G
from .batch import OBatch

SRC_DST = {'ll': ('lig', 'lig'), 'kl': ('kp', 'lig'), 'lk': ('lig', 'kp'), 'kk': ('kp', 'kp')}


def _lin(sd, key, x):
    b = sd.get(key + '.bias')
    return F.linear(x, sd[key + '.weight'], b)


def lig_edges(batch: OBatch, cfg: dict) -> Dict[str, tuple]:
    """models/dynamics.py:387-420 (identical in dynamics_gvp.py:201-234): ll by radius graph
    (max 200) or knn graph; kl by knn(x=lig, y=kp) or radius (max 100); lk = kl reversed."""
    lx, kx = batch.x['lig'], batch.x['kp']
    nl, nk = batch.n['lig'], batch.n['kp']
    cut = cfg.get('graph_cutoffs', {})
    if cfg.get('ll_k', 0) > 0:
        ll = G.knn_graph(lx, cfg['ll_k'], nl)
    else:
        ll = G.radius_graph(lx, cut['ll'], nl, max_num_neighbors=200)
    if cfg.get('kl_k', 0) > 0:
        kp_idx, lig_idx = G.knn(lx, kx, cfg['kl_k'], nl, nk)
    else:
        kp_idx, lig_idx = G.radius(lx, kx, cut['kl'], nl, nk, max_num_neighbors=100)
    out = {'ll': ll, 'kl': (kp_idx, lig_idx)}        # g.add_edges(kl_idxs[0], kl_idxs[1], 'kl')
    out['lk'] = (lig_idx, kp_idx)                    # :413
    return out


def egnn_conv(sd, prefix, etypes, updated, edges, h, x, z, use_tanh, norm, coords_range=10.0):
    """One LigRecConv layer (models/dynamics.py:124-207)."""
    n_nodes = {nt: h[nt].shape[0] for nt in h}
    h_neigh = {nt: torch.zeros_like(h[nt]) for nt in updated}
    x_neigh = {nt: torch.zeros_like(x[nt]) for nt in updated}
    for et in etypes:
        s_nt, d_nt = SRC_DST[et]
        src, dst = edges[et]
        x_diff = x[s_nt][src] - x[d_nt][dst]                           # :160 u_sub_v
        dij = torch.linalg.vector_norm(x_diff, dim=1).unsqueeze(-1)     # :211 (not squared)
        x_diff = x_diff / (dij + 1)                                     # :169
        f = torch.cat([h[s_nt][src], h[d_nt][dst], dij], dim=-1)        # :103-105
        p = f'{prefix}.edge_mlp.{et}'
        m = F.silu(_lin(sd, p + '.2', F.silu(_lin(sd, p + '.0', f))))   # :37-46, :111
        att = torch.sigmoid(_lin(sd, f'{prefix}.soft_attention.{et}.0', m))
        msg_h = m * att                                                 # :112
        p = f'{prefix}.coord_mlp.{et}'
        c = F.silu(_lin(sd, p + '.2', F.silu(_lin(sd, p + '.0', f))))
        c = F.linear(c, sd[p + '.4.weight'])                            # :66-79, no bias
        # :115 `edge_type[1] in ["kk","lk"]` compares a character with 2-char strings and is
        # always False, so coordinate messages are produced on every edge type.
        if use_tanh:
            msg_x = torch.tanh(c) * x_diff * coords_range               # :118
        else:
            msg_x = c * x_diff                                          # :120
        # multi_update_all(sum, cross_reducer='sum') :177-185
        h_neigh[d_nt] = h_neigh[d_nt] + G.scatter_sum(msg_h, dst, n_nodes[d_nt])
        x_neigh[d_nt] = x_neigh[d_nt] + G.scatter_sum(msg_x, dst, n_nodes[d_nt])
    h_out, x_out = {}, {}
    for nt in updated:
        hn = h_neigh[nt] / z[nt]                                        # :188-192
        xn = x_neigh[nt] / z[nt]
        p = f'{prefix}.node_mlp.{nt}'
        upd = _lin(sd, p + '.2', F.silu(_lin(sd, p + '.0', torch.cat([h[nt], hn], dim=1))))
        hh = h[nt] + upd                                                # :203 residual
        if norm:
            hh = F.layer_norm(hh, (hh.shape[1],), sd[f'{prefix}.layer_norm.{nt}.weight'],
                              sd[f'{prefix}.layer_norm.{nt}.bias'], 1e-5)
        h_out[nt] = hh
        x_out[nt] = x[nt] + xn                                          # :206
    return h_out, x_out


def egnn_dynamics_forward(sd: Dict[str, torch.Tensor], cfg: dict, batch: OBatch, t: torch.Tensor,
                          edges: Dict[str, tuple] = None, return_layers: bool = False):
    """LigRecDynamics.forward (models/dynamics.py:342-385).

    cfg keys = reference ctor kwargs (n_layers, hidden_nf, use_tanh, message_norm,
    update_kp_feat, norm, ll_k, kl_k, graph_cutoffs).  `edges` may inject a precomputed
    edge dict {'ll','kl','lk'} (for layer-math parity with a fixed edge list); kk always
    comes from batch.edges['kk'].  Returns (eps_h [N_lig, atom_nf], eps_x [N_lig, 3]).
    """
    update_kp = cfg.get('update_kp_feat', False)
    etypes = ['ll', 'kl', 'lk', 'kk'] if update_kp else ['ll', 'kl']     # :29-34
    updated = ['lig', 'kp'] if update_kp else ['lig']
    lig_b = G.counts_to_batch_idx(batch.n['lig'])
    kp_b = G.counts_to_batch_idx(batch.n['kp'])

    lig_feat = F.silu(_lin(sd, 'lig_encoder.2', F.silu(_lin(sd, 'lig_encoder.0', batch.h['lig']))))
    if 'rec_encoder.0.weight' in sd:                                     # :326-334
        kp_feat = F.silu(_lin(sd, 'rec_encoder.2', F.silu(_lin(sd, 'rec_encoder.0', batch.h['kp']))))
    else:
        kp_feat = batch.h['kp']
    lig_feat = torch.cat([lig_feat, t[lig_b].view(-1, 1)], dim=1)        # :359-363
    kp_feat = torch.cat([kp_feat, t[kp_b].view(-1, 1)], dim=1)

    if edges is None:
        edges = lig_edges(batch, cfg)
    edges = dict(edges)
    edges['kk'] = batch.edges.get('kk', (torch.zeros(0, dtype=torch.long),) * 2)

    h = {'lig': lig_feat, 'kp': kp_feat}
    x = {'lig': batch.x['lig'], 'kp': batch.x['kp']}

    # z (models/dynamics.py:277-285)
    message_norm = cfg.get('message_norm', 1)
    z = {}
    bidx = {'lig': lig_b, 'kp': kp_b}
    for nt in updated:
        if message_norm == 0:
            tot = torch.zeros(batch.batch_size, dtype=torch.long)
            for et in etypes:
                if SRC_DST[et][1] == nt:
                    tot = tot + G.edges_per_graph(edges[et][1], batch.n[nt])
            zz = tot.to(torch.float32) / batch.n[nt].to(torch.float32)
            z[nt] = zz[bidx[nt]].view(-1, 1) + 1
        else:
            z[nt] = message_norm

    layers = []
    for i in range(cfg['n_layers']):
        h_new, x_new = egnn_conv(sd, f'egnn.conv_layers.{i}', etypes, updated, edges, h, x, z,
                                 cfg.get('use_tanh', False), cfg.get('norm', False))
        # :288-292 -- when kp is not updated it is re-read from the graph each layer
        h = {'lig': h_new['lig'], 'kp': h_new.get('kp', kp_feat)}
        x = {'lig': x_new['lig'], 'kp': x_new.get('kp', batch.x['kp'])}
        if return_layers:
            layers.append(({k: v.clone() for k, v in h.items()}, {k: v.clone() for k, v in x.items()}))

    hl = h['lig'][:, :-1]                                                # :376
    eps_h = _lin(sd, 'lig_decoder.2', F.silu(_lin(sd, 'lig_decoder.0', hl)))
    eps_x = x['lig'] - batch.x['lig']                                    # :381
    if return_layers:
        return eps_h, eps_x, layers
    return eps_h, eps_x
