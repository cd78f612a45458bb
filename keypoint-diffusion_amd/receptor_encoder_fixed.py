"""Fixed ("all-atom" / "C-alpha") receptor representation: receptor atoms become the keypoints.

Mirrors models/receptor_encoder_fixed.py:9-66 on the graph container: kp := rec nodes (x_0, h_0,
zero v_0 for GVP), kk := rr edges, rec emptied.  Pure index plumbing, run once per pocket.
"""
from typing import Dict, Optional

import torch
import torch.nn as nn

from .graph import HeteroBatch, get_batch_info


class FixedReceptorEncoder(nn.Module):

    def __init__(self, n_vec_feats: Optional[int]):
        super().__init__()
        self.n_vec_feats = n_vec_feats

    def forward(self, g: HeteroBatch, batch_idxs: Dict[str, torch.Tensor] = None) -> HeteroBatch:
        batch_size = g.batch_size
        nodes, edges = get_batch_info(g)
        rec = g.nodes['rec'].data
        n_rec = g.num_nodes('rec')

        g.remove_nodes(g.nodes('kp'), ntype='kp')
        g.add_nodes(n_rec, {'x_0': rec['x_0'], 'h_0': rec['h_0']}, ntype='kp')
        if self.n_vec_feats is not None:
            g.nodes['kp'].data['v_0'] = torch.zeros((n_rec, self.n_vec_feats, 3), device=g.device)
        g.add_edges(*g.edges(etype='rr'), etype='kk')

        nodes['kp'] = nodes['rec']
        edges[('kp', 'kk', 'kp')] = edges[('rec', 'rr', 'rec')]
        nodes['rec'] = torch.zeros_like(nodes['rec'])
        for et in edges:
            if 'rec' in et:
                edges[et] = torch.zeros_like(edges[et])
        g.remove_nodes(g.nodes('rec'), ntype='rec')
        g.set_batch_num_nodes(nodes)
        g.set_batch_num_edges(edges)
        assert g.batch_size == batch_size
        return g
