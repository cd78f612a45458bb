"""CPU restatement of the receptor encoders (test infrastructure only).

Follows models/receptor_encoder_gvp.py (KeypointInitializer.forward :40-93,
ReceptorEncoderGVP.forward :212-294, update_rk_edges :297-321), models/gvp.py GVPEdgeConv
(:249-341) and models/receptor_encoder_fixed.py:15-66.
`sd` is the state_dict of the `rec_encoder` sub-module.
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F

from . import graph_ops as G
from .batch import OBatch
from .gvp import edge_geometry, gvp_chain, gvp_layernorm


def fixed_encoder(batch: OBatch, n_vec_feats=None) -> OBatch:
    """FixedReceptorEncoder.forward (receptor_encoder_fixed.py:15-66): kp := rec nodes,
    kk := rr edges, rec emptied."""
    out = batch.clone()
    out.n['kp'] = batch.n['rec'].clone()
    out.x['kp'] = batch.x['rec'].clone()
    out.h['kp'] = batch.h['rec'].clone()
    if n_vec_feats is not None:
        out.v['kp'] = torch.zeros(batch.x['rec'].shape[0], n_vec_feats, 3)
    out.edges['kk'] = tuple(t.clone() for t in batch.edges['rr'])
    out.n['rec'] = torch.zeros_like(batch.n['rec'])
    out.x['rec'] = batch.x['rec'][:0]
    out.h['rec'] = batch.h['rec'][:0]
    for et in ('rr', 'rk'):
        out.edges[et] = (torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long))
    return out


def edge_conv(sd, p, src, dst, src_feats, dst_feats, z, cfg, rbf_dmax, use_dst_feats):
    """GVPEdgeConv.forward (gvp.py:249-341) without edge features."""
    s_s, x_s, v_s = src_feats
    s_d, x_d, v_d = dst_feats
    x_diff, d = edge_geometry(x_s, x_d, src, dst, rbf_dmax)
    vec = [x_diff.unsqueeze(1), v_s[src]]
    sc = [s_s[src], d]
    if use_dst_feats:                                                         # :323-337
        vec.append(v_d[dst])
        sc.append(s_d[dst])
    ms, mv = gvp_chain(sd, p + '.edge_message', cfg.get('n_message_gvps', 1), torch.cat(sc, 1), torch.cat(vec, 1))
    red = G.scatter_mean if cfg.get('message_norm', 10) == 'mean' else G.scatter_sum
    n_dst = s_d.shape[0]
    ms = red(ms, dst, n_dst) / z                                              # :303-310
    mv = red(mv, dst, n_dst) / (z.unsqueeze(-1) if isinstance(z, torch.Tensor) else z)
    s, v = gvp_layernorm(sd, p + '.message_layer_norm', s_d + ms, v_d + mv)
    rs, rv = gvp_chain(sd, p + '.node_update', cfg.get('n_update_gvps', 1), s, v)
    return gvp_layernorm(sd, p + '.update_layer_norm', s + rs, v + rv)


def keypoint_positions(sd, p, rec_s, rec_x, n_rec, n_kp_per_graph, scalar_size):
    """KeypointInitializer.forward (receptor_encoder_gvp.py:40-93).  The reference relies on
    the dataset's complete, dst-major rec->kp edge list (pdbbind_processing.py:253-255); the
    dense per-graph form below is the same sum."""
    mean = G.segment_mean_nodes(rec_s, n_rec)                                 # :51
    e = F.silu(F.linear(mean, sd[p + '.keypoint_embedding.0.weight'], sd[p + '.keypoint_embedding.0.bias']))
    e = F.layer_norm(e, (e.shape[1],), sd[p + '.keypoint_embedding.2.weight'], sd[p + '.keypoint_embedding.2.bias'], 1e-5)
    kp_s = e.reshape(-1, scalar_size)                                         # 'b (k d) -> (b k) d'
    ft_src = F.linear(rec_s, sd[p + '.src_net.weight'])
    ft_dst = F.linear(kp_s, sd[p + '.dst_net.weight'])
    rp = G.counts_to_ptr(n_rec)
    pos = []
    for b in range(n_rec.numel()):
        fs = ft_src[rp[b]:rp[b + 1]]
        fd = ft_dst[b * n_kp_per_graph:(b + 1) * n_kp_per_graph]
        a = torch.exp((fd @ fs.T) / math.sqrt(scalar_size))                   # :69-73, no max-subtraction
        a = a / a.sum(dim=1, keepdim=True)                                    # :75-81
        pos.append(a @ rec_x[rp[b]:rp[b + 1]])                                # :84-87
    return torch.cat(pos, 0)


def rec_encoder_gvp_forward(sd: Dict[str, torch.Tensor], cfg: dict, batch: OBatch) -> OBatch:
    """ReceptorEncoderGVP.forward.  cfg = reference ctor kwargs (+ graph_cutoffs, n_keypoints)."""
    S = cfg.get('out_scalar_size', 128)
    V = cfg.get('vector_size', 16)
    K = cfg.get('n_keypoints', 20)
    cut = cfg['graph_cutoffs']
    mn = cfg.get('message_norm', 10)
    n_rec = batch.n['rec']
    B = batch.batch_size
    rec_b = G.counts_to_batch_idx(n_rec)

    s = F.silu(F.linear(batch.h['rec'], sd['scalar_embed.0.weight'], sd['scalar_embed.0.bias']))
    s = F.silu(F.linear(s, sd['scalar_embed.2.weight'], sd['scalar_embed.2.bias']))
    s = F.layer_norm(s, (S,), sd['scalar_norm.weight'], sd['scalar_norm.bias'], 1e-5)   # :221-222
    v = torch.zeros(s.shape[0], V, 3)
    x = batch.x['rec']
    rr_src, rr_dst = batch.edges['rr']

    if mn == 'mean':                                                          # :240-249
        z = 1
    elif mn == 0:
        z = G.edges_per_graph(rr_dst, n_rec).float() / n_rec.float()
        z = z[rec_b].view(-1, 1)
    else:
        z = mn
    for i in range(cfg.get('n_rr_convs', 3)):
        s, v = edge_conv(sd, f'rr_conv_layers.{i}', rr_src, rr_dst, (s, x, v), (s, x, v), z, cfg,
                         rbf_dmax=cut['rr'], use_dst_feats=False)

    kp_x = keypoint_positions(sd, 'keypoint_initializer', s, x, n_rec, K, S)
    n_kp = torch.full((B,), K, dtype=torch.long)
    kp_s = torch.zeros(B * K, S)                                              # :90-91
    kp_v = torch.zeros(B * K, V, 3)

    if cfg.get('k_closest', 0) > 0:                                           # :302-306
        kp_idx, rec_idx = G.knn(x, kp_x, cfg['k_closest'], n_rec, n_kp)
    else:
        kp_idx, rec_idx = G.radius(x, kp_x, cfg['kp_rad'], n_rec, n_kp, max_num_neighbors=10)
    if mn == 0:                                                               # :266-269
        kp_b = G.counts_to_batch_idx(n_kp)
        z = G.edges_per_graph(kp_idx, n_kp).float() / n_kp.float()
        z = z[kp_b].view(-1, 1)
    for i in range(cfg.get('n_rk_convs', 2)):
        kp_s, kp_v = edge_conv(sd, f'rk_conv_layers.{i}', rec_idx, kp_idx, (s, x, v), (kp_s, kp_x, kp_v), z, cfg,
                               rbf_dmax=cut['rk'], use_dst_feats=(i != 0))

    out = batch.clone()
    out.n['kp'] = n_kp
    out.x['kp'], out.h['kp'], out.v['kp'] = kp_x, kp_s, kp_v
    out.edges['rk'] = (rec_idx, kp_idx)
    out.edges['kk'] = G.radius_graph(kp_x, cut['kk'], n_kp, max_num_neighbors=100)   # :285
    return out
