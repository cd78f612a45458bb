"""GVP keypoint receptor encoder behind the reference's module interface
(models/receptor_encoder_gvp.py:15-321).  Parameter containers with the reference state-dict
layout; `forward` runs in libkpd_hip.so.
"""
from typing import Dict, Union

import torch
import torch.nn as nn

from .graph import HeteroBatch
from .gvp import GVPEdgeConv


class KeypointInitializer(nn.Module):
    """receptor_encoder_gvp.py:19-37."""

    def __init__(self, n_keypoints: int, scalar_size: int, vector_size: int):
        super().__init__()
        self.scalar_size, self.vector_size, self.n_keypoints, self.num_heads = scalar_size, vector_size, n_keypoints, 1
        self.src_net = nn.Linear(scalar_size, scalar_size, bias=False)
        self.dst_net = nn.Linear(scalar_size, scalar_size, bias=False)
        self.keypoint_embedding = nn.Sequential(nn.Linear(scalar_size, scalar_size * n_keypoints), nn.SiLU(),
                                                nn.LayerNorm(scalar_size * n_keypoints))
        self.norm = nn.LayerNorm(scalar_size)


class ReceptorEncoderGVP(nn.Module):

    def __init__(self, in_scalar_size: int, out_scalar_size: int = 128, n_message_gvps: int = 1, n_update_gvps: int = 1,
                 vector_size: int = 16, n_rr_convs: int = 3, n_rk_convs: int = 2, message_norm: Union[float, str] = 10,
                 use_sameres_feat: bool = False, kp_rad: float = 0, k_closest: int = 0, dropout: float = 0.0,
                 n_keypoints: int = 20, no_cg: bool = False, graph_cutoffs: dict = {}):
        super().__init__()
        if no_cg:
            raise NotImplementedError('no_cg is not implemented yet')
        if kp_rad != 0 and k_closest != 0:
            raise ValueError('one of kp_rad and kp_closest can be zero but not both')
        if kp_rad == 0 and k_closest == 0:
            raise ValueError('one of kp_rad and kp_closest must be non-zero')
        if isinstance(message_norm, str) and message_norm != 'mean':
            raise ValueError(f'message norm must be either a float, int, or "mean". Got {message_norm}')
        if not isinstance(message_norm, (str, float, int)):
            raise ValueError(f'message norm must be either a float, int, or "mean". Got {message_norm}')
        self.n_rr_convs, self.n_rk_convs = n_rr_convs, n_rk_convs
        self.in_scalar_size, self.out_scalar_size, self.vector_size = in_scalar_size, out_scalar_size, vector_size
        self.n_keypoints, self.use_sameres_feat, self.kp_rad, self.k_closest = n_keypoints, use_sameres_feat, kp_rad, k_closest
        self.message_norm, self.graph_cutoffs = message_norm, graph_cutoffs
        self.n_message_gvps, self.n_update_gvps = n_message_gvps, n_update_gvps
        self.rk_graph_type = 'knn' if k_closest > 0 else 'radius'
        self.scalar_embed = nn.Sequential(nn.Linear(in_scalar_size, out_scalar_size), nn.SiLU(),
                                          nn.Linear(out_scalar_size, out_scalar_size), nn.SiLU())
        self.scalar_norm = nn.LayerNorm(out_scalar_size)
        common = dict(scalar_size=out_scalar_size, vector_size=vector_size, n_message_gvps=n_message_gvps,
                      n_update_gvps=n_update_gvps, edge_feat_size=1 if use_sameres_feat else 0, dropout=dropout,
                      message_norm=message_norm)
        self.rr_conv_layers = nn.ModuleList(
            [GVPEdgeConv(edge_type=('rec', 'rr', 'rec'), rbf_dmax=graph_cutoffs['rr'], **common) for _ in range(n_rr_convs)])
        self.keypoint_initializer = KeypointInitializer(n_keypoints=n_keypoints, scalar_size=out_scalar_size,
                                                        vector_size=vector_size)
        self.rk_conv_layers = nn.ModuleList(
            [GVPEdgeConv(edge_type=('rec', 'rk', 'kp'), use_dst_feats=(i != 0), rbf_dmax=graph_cutoffs['rk'], **common)
             for i in range(n_rk_convs)])

    def forward(self, g: HeteroBatch, batch_idxs: Dict[str, torch.Tensor] = None) -> HeteroBatch:
        raise NotImplementedError('the GVP receptor encoder HIP path is not built yet in this revision')
