"""Geometric-vector-perceptron layers: parameter containers with the reference's
state-dict layout (models/gvp.py:43-116, 118-166, 170-248, 343-437).

These modules own the weights only.  The arithmetic runs in the HIP library
(csrc/gvp_kernels.hip) through the enclosing dynamics / encoder `forward`; calling a layer's
`forward` directly is not part of the hot path and is not provided.
"""
import math
from typing import Dict, List, Tuple, Union

import torch
from torch import nn


class GVP(nn.Module):
    """Weights of one GVP (models/gvp.py:43-87): Wh [v_in,h], Wu [h,v_out],
    to_feats_out = Linear(h + s_in, s_out) (+act), scalar_to_vector_gates = Linear(s_out, v_out)."""

    def __init__(self, dim_vectors_in, dim_vectors_out, dim_feats_in, dim_feats_out, hidden_vectors=None,
                 feats_activation=None, vectors_activation=None, vector_gating=True, xavier_init=False):
        super().__init__()
        if not vector_gating:
            raise NotImplementedError('only the vector-gated GVP is used by the reference configs')
        self.dim_vectors_in, self.dim_vectors_out = dim_vectors_in, dim_vectors_out
        self.dim_feats_in, self.dim_feats_out = dim_feats_in, dim_feats_out
        self.dim_h = max(dim_vectors_in, dim_vectors_out) if hidden_vectors is None else hidden_vectors
        self.vectors_activation = vectors_activation if vectors_activation is not None else nn.Sigmoid()
        kh, ku = 1 / math.sqrt(dim_vectors_in), 1 / math.sqrt(self.dim_h)
        self.Wh = nn.Parameter(torch.empty(dim_vectors_in, self.dim_h).uniform_(-kh, kh))
        self.Wu = nn.Parameter(torch.empty(self.dim_h, dim_vectors_out).uniform_(-ku, ku))
        self.to_feats_out = nn.Sequential(
            nn.Linear(self.dim_h + dim_feats_in, dim_feats_out),
            feats_activation if feats_activation is not None else nn.SiLU())
        self.scalar_to_vector_gates = nn.Linear(dim_feats_out, dim_vectors_out)
        if xavier_init:
            nn.init.xavier_uniform_(self.scalar_to_vector_gates.weight, gain=1)
            nn.init.constant_(self.scalar_to_vector_gates.bias, 0)

    @property
    def vector_act_is_identity(self) -> bool:
        return isinstance(self.vectors_activation, nn.Identity)


class _VDropout(nn.Module):
    def __init__(self, drop_rate):
        super().__init__()
        self.drop_rate = drop_rate
        self.dummy_param = nn.Parameter(torch.empty(0))      # state-dict key kept (gvp.py:126)


class GVPDropout(nn.Module):
    """Identity in eval mode (gvp.py:133-134); sampling always runs under eval()."""

    def __init__(self, rate):
        super().__init__()
        self.vector_dropout = _VDropout(rate)
        self.feat_dropout = nn.Dropout(rate)


class GVPLayerNorm(nn.Module):
    def __init__(self, feats_h_size, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.feat_norm = nn.LayerNorm(feats_h_size)


def _gvp_stack(n, first_v_in, first_s_in, v, s):
    return nn.Sequential(*[
        GVP(dim_vectors_in=first_v_in if i == 0 else v, dim_vectors_out=v,
            dim_feats_in=first_s_in if i == 0 else s, dim_feats_out=s)
        for i in range(n)])


class GVPEdgeConv(nn.Module):
    """Single-edge-type GVP convolution weights (gvp.py:170-248); used by the receptor encoder."""

    def __init__(self, edge_type: Tuple[str, str, str], scalar_size=128, vector_size=16,
                 scalar_activation=nn.SiLU, vector_activation=nn.Sigmoid, n_message_gvps=1, n_update_gvps=1,
                 use_dst_feats=False, rbf_dmax=15, rbf_dim=16, edge_feat_size=0, coords_range=10,
                 message_norm: Union[float, str] = 10, dropout=0.0):
        super().__init__()
        if edge_feat_size:
            raise NotImplementedError('edge features (use_sameres_feat) are not used by any shipped config')
        self.edge_type, self.src_ntype, self.dst_ntype = edge_type, edge_type[0], edge_type[2]
        self.scalar_size, self.vector_size = scalar_size, vector_size
        self.n_message_gvps, self.n_update_gvps = n_message_gvps, n_update_gvps
        self.use_dst_feats, self.rbf_dmax, self.rbf_dim = use_dst_feats, rbf_dmax, rbf_dim
        self.message_norm = message_norm
        v_in = vector_size + 1 + (vector_size if use_dst_feats else 0)
        s_in = scalar_size + rbf_dim + (scalar_size if use_dst_feats else 0)
        self.edge_message = _gvp_stack(n_message_gvps, v_in, s_in, vector_size, scalar_size)
        self.node_update = _gvp_stack(n_update_gvps, vector_size, scalar_size, vector_size, scalar_size)
        self.dropout = GVPDropout(dropout)
        self.message_layer_norm = GVPLayerNorm(scalar_size)
        self.update_layer_norm = GVPLayerNorm(scalar_size)


class GVPMultiEdgeConv(nn.Module):
    """Multi-edge-type GVP convolution weights (gvp.py:343-437); used by the GVP denoiser."""

    def __init__(self, etypes: List[Tuple[str, str, str]], scalar_size=128, vector_size=16,
                 scalar_activation=nn.SiLU, vector_activation=nn.Sigmoid, n_message_gvps=1, n_update_gvps=1,
                 rbf_dmax=15, rbf_dim=16, message_norm: Union[float, str, Dict] = 10, dropout=0.0):
        super().__init__()
        self.etypes = list(etypes)
        self.scalar_size, self.vector_size = scalar_size, vector_size
        self.n_message_gvps, self.n_update_gvps = n_message_gvps, n_update_gvps
        self.rbf_dmax, self.rbf_dim = rbf_dmax, rbf_dim
        self.dst_ntypes = sorted({et[2] for et in self.etypes})
        if isinstance(message_norm, dict):
            # the reference's dict branch calls set.keys() and cannot run (gvp.py:453)
            raise NotImplementedError('dict-valued message_norm is not supported')
        if (isinstance(message_norm, str) and message_norm != 'mean') or \
                (isinstance(message_norm, (int, float)) and message_norm < 0):
            raise ValueError(f"message_norm values must be 'mean' or a positive number, got {message_norm}")
        self.message_norm = message_norm
        self.edge_message_fns = nn.ModuleDict({
            '_'.join(et): _gvp_stack(n_message_gvps, vector_size + 1, scalar_size + rbf_dim, vector_size, scalar_size)
            for et in self.etypes})
        self.node_update_fns = nn.ModuleDict()
        self.update_layer_norms = nn.ModuleDict()
        self.message_layer_norms = nn.ModuleDict()
        for nt in self.dst_ntypes:
            self.node_update_fns[nt] = _gvp_stack(n_update_gvps, vector_size, scalar_size, vector_size, scalar_size)
            self.message_layer_norms[nt] = GVPLayerNorm(scalar_size)
            self.update_layer_norms[nt] = GVPLayerNorm(scalar_size)
        self.dropout = GVPDropout(dropout)
