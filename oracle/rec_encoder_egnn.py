"""CPU restatement of the EGNN keypoint receptor encoder (test infrastructure only).

Follows models/receptor_encoder.py: ReceptorConv (:14-154), RecKeyConv (:156-297, k_closest path), ReceptorEncoder
(:381-555).  `sd` is the state_dict of the `rec_encoder` sub-module; `edge_feat` is rr `same_res` [E,1] float when
use_sameres_feat.  Like the GVP encoder's KeypointInitializer, RecKeyConv assumes the dataset's complete rec->kp edge list
(pdbbind_processing.py:253-255); the dense per-graph form below is the same sum.
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import graph_ops as G
from .batch import OBatch


def _lin(sd, p, x):
    return F.linear(x, sd[p + '.weight'], sd.get(p + '.bias'))


def receptor_conv(sd, p, h, x, src, dst, z, a: Optional[torch.Tensor], cfg):
    """ReceptorConv.forward (:98-154): no residual on h; msg_x is zero with fix_pos."""
    x_diff = x[src] - x[dst]                                                   # :137
    radial = torch.linalg.vector_norm(x_diff, dim=1).unsqueeze(-1)             # :138 (distance, not squared)
    x_diff = x_diff / (radial + 1)                                             # :140-142
    f = torch.cat([h[src], h[dst], radial] + ([a] if a is not None else []), dim=-1)   # :69-83
    m = F.silu(_lin(sd, p + '.edge_mlp.0', f))
    m = F.silu(_lin(sd, p + '.edge_mlp.2', m))
    m = m * torch.sigmoid(_lin(sd, p + '.soft_attention.0', m))                # :85-86
    n = h.shape[0]
    h_neigh = G.scatter_sum(m, dst, n) / z                                     # :144-147
    if cfg.get('fix_pos', False):
        x_new = x
    else:
        c = _lin(sd, p + '.coord_mlp.2', F.silu(_lin(sd, p + '.coord_mlp.0', f)))
        msg_x = torch.tanh(c) * x_diff * cfg.get('coords_range', 10) if cfg.get('use_tanh', True) else c * x_diff   # :89-92
        x_new = x + G.scatter_sum(msg_x, dst, n) / z                           # :150
    hn = _lin(sd, p + '.node_mlp.2', F.silu(_lin(sd, p + '.node_mlp.0', torch.cat([h, h_neigh], dim=-1))))   # :149
    if cfg.get('norm', False):
        hn = F.layer_norm(hn, (hn.shape[1],), sd[p + '.layer_norm.weight'], sd[p + '.layer_norm.bias'], 1e-5)
    return hn, x_new


def rec_key_conv(sd, p, h, x_val, x0, kp_h0, n_rec, K, cfg):
    """RecKeyConv.forward (:182-297), k_closest or kp_rad features.  fc_src is applied to both sides (:190-191; fc_dst is unused)."""
    D = kp_h0.shape[1]
    k = cfg.get('k_closest', 0)
    ft_src = F.linear(h, sd[p + '.fc_src.weight'])
    ft_dst = F.linear(kp_h0, sd[p + '.fc_src.weight'])
    rp = G.counts_to_ptr(n_rec)
    pos = []
    for b in range(n_rec.numel()):
        a = torch.exp((ft_dst[b * K:(b + 1) * K] @ ft_src[rp[b]:rp[b + 1]].T) / math.sqrt(D))   # :200-204, no max-subtraction
        pos.append((a / a.sum(dim=1, keepdim=True)) @ x_val[rp[b]:rp[b + 1]])                  # :206-222
    kp_pos = torch.cat(pos, 0)
    n_kp = torch.full((n_rec.numel(),), K, dtype=torch.long)
    if k == 0:                                                                 # kp_rad_feats (:238-262)
        kp_idx, rec_idx = G.radius(x0, kp_pos, cfg['kp_rad'], n_rec, n_kp, max_num_neighbors=100)
        h_m = torch.zeros(kp_pos.shape[0], D, dtype=h.dtype).index_add_(0, kp_idx, h[rec_idx])                # fn.sum, :258
        z = G.edges_per_graph(kp_idx, n_kp).float() / K + 1.0                                  # :259-260
        feat = F.silu(_lin(sd, p + '.kp_feature_mlp.0', h_m / z[G.counts_to_batch_idx(n_kp)].view(-1, 1)))
        if cfg.get('norm', False):
            feat = F.layer_norm(feat, (D,), sd[p + '.layer_norm.weight'], sd[p + '.layer_norm.bias'], 1e-5)
        return kp_pos, feat, (rec_idx, kp_idx)
    kp_idx, rec_idx = G.knn(x0, kp_pos, k, n_rec, n_kp)                        # :262-267 (original positions x_0)
    h_m = G.scatter_mean(h[rec_idx], kp_idx, kp_pos.shape[0])                  # :284
    d = torch.linalg.vector_norm(x0[rec_idx] - kp_pos[kp_idx] + 1e-30, dim=1)  # :285-286
    d_k = d.view(-1, k)                                                        # mailbox order = edge order (nearest first)
    feat = F.silu(_lin(sd, p + '.kp_feature_mlp.0', torch.cat([h_m, d_k], dim=1)))   # :289, :234
    if cfg.get('norm', False):
        feat = F.layer_norm(feat, (D,), sd[p + '.layer_norm.weight'], sd[p + '.layer_norm.bias'], 1e-5)
    return kp_pos, feat, (rec_idx, kp_idx)


def rec_encoder_egnn_forward(sd: Dict[str, torch.Tensor], cfg: dict, batch: OBatch,
                             edge_feat: Optional[torch.Tensor] = None, return_rec: bool = False):
    """ReceptorEncoder.forward (:483-555).  cfg = reference ctor kwargs (+ graph_cutoffs, n_keypoints)."""
    if (cfg.get('kp_rad', 0) != 0) == (cfg.get('k_closest', 0) != 0):
        raise ValueError('exactly one of kp_rad and k_closest must be non-zero')                # :399-402
    n_rec = batch.n['rec']
    B, K = batch.batch_size, cfg.get('n_keypoints', 10)
    D = cfg.get('out_n_node_feat', 256)
    rec_b = G.counts_to_batch_idx(n_rec)
    x, h = batch.x['rec'], batch.h['rec']
    src, dst = batch.edges['rr']
    a = edge_feat.float().view(-1, 1) if cfg.get('use_sameres_feat', False) else None
    if cfg.get('message_norm', 1) == 0:                                        # :505-509 (no +1)
        z = (G.edges_per_graph(dst, n_rec).float() / n_rec.float())[rec_b].view(-1, 1)
    else:
        z = cfg.get('message_norm', 1)
    for i in range(cfg.get('n_convs', 6)):
        h, x = receptor_conv(sd, f'rec_convs.{i}', h, x, src, dst, z, a, cfg)  # :512-513
    mean = G.segment_mean_nodes(h, n_rec)                                      # :526
    kp_h0 = F.silu(_lin(sd, 'keypoint_embedding.0', mean)).reshape(-1, D)      # :529-530 'b (k d) -> (b k) d'
    x_val = batch.x['rec'] if cfg.get('fix_pos', False) else x                 # :209-213
    kp_pos, kp_feat, rk = rec_key_conv(sd, 'rec_kp_conv', h, x_val, batch.x['rec'], kp_h0, n_rec, K, cfg)
    out = batch.clone()
    out.n['kp'] = torch.full((B,), K, dtype=torch.long)
    out.x['kp'], out.h['kp'] = kp_pos, kp_feat
    out.edges['rk'] = rk
    out.edges['kk'] = G.radius_graph(kp_pos, cfg['graph_cutoffs']['kk'], out.n['kp'], max_num_neighbors=100)   # :541
    return (out, h, x) if return_rec else out
