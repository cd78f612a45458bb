"""GVP keypoint receptor encoder behind the reference's module interface
(models/receptor_encoder_gvp.py:15-321).  Parameter containers with the reference state-dict
layout; `forward` runs in libkpd_hip.so.
"""
from typing import Dict, Union

import torch
import torch.nn as nn

from .graph import HeteroBatch, get_batch_info
from .gvp import GVPEdgeConv


class _RecEncTrainFn(torch.autograd.Function):
    """ReceptorEncoderGVP.forward as one autograd node (kpd_recenc_trainer_*): the keypoint positions, scalars and vectors are
    differentiable functions of every parameter; the rk / kk edge lists it also produces are data (torch_cluster ops upstream)."""

    @staticmethod
    def forward(ctx, module, rec_counts, rec_x, rec_h, rr_src, rr_dst, holder, *params):
        trainer, names = module._trainer()
        ctx.trainer, ctx.names = trainer, names
        trainer.generation = getattr(trainer, 'generation', 0) + 1
        ctx.generation = trainer.generation
        ctx.save_for_backward(*params)
        trainer.bind(names, params, [None] * len(params))
        rate = module.dropout_rate if module.training else 0.0
        module.last_dropout_seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if rate > 0 else 0
        trainer.set_dropout(rate, module.last_dropout_seed)
        out = trainer.forward(rec_counts, rec_x, rec_h, rr_src, rr_dst)
        ctx.keep = out.pop('_keep')                                 # device copies the C side reads again in backward
        holder.update(out)                                          # edge lists and counts: non-differentiable side outputs
        return out['kp_x'], out['kp_h'], out['kp_v']

    @staticmethod
    def backward(ctx, d_x, d_h, d_v):
        from . import hip
        if ctx.generation != ctx.trainer.generation:
            raise hip.KpdError('backward of a ReceptorEncoderGVP forward whose saved states were overwritten by a later grad-enabled '
                               'forward of the same module (one forward/backward pair at a time per module)')
        params = ctx.saved_tensors
        grads = hip.zero_grads_like(params, [ctx.needs_input_grad[7 + i] for i in range(len(params))])
        ctx.trainer.bind(ctx.names, params, grads)
        c = lambda t: None if t is None else t.contiguous().float()
        ctx.trainer.backward(c(d_x), c(d_h), c(d_v))
        return (None,) * 7 + tuple(grads)


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

        self.dropout_rate = dropout
        self._engine = None
        self._engine_key = None
        self._train = None

    def _trainer(self):
        """The training engine and the parameter names in `self.parameters()` order (reference state-dict names)."""
        from . import hip
        if self._train is None:
            if self.use_sameres_feat:
                raise NotImplementedError('use_sameres_feat=True cannot run in the reference GVP encoder; every shipped config sets it to False')
            mode, val = hip._norm_mode(self.message_norm)
            cfg = hip.KpdRecencConfig(int(self.in_scalar_size), int(self.out_scalar_size), int(self.vector_size), int(self.n_rr_convs),
                                      int(self.n_rk_convs), int(self.n_message_gvps), int(self.n_update_gvps), mode, val,
                                      int(self.k_closest), int(self.n_keypoints), float(self.graph_cutoffs['rr']),
                                      float(self.graph_cutoffs['rk']), float(self.graph_cutoffs['kk']), float(self.kp_rad))
            self._train = (hip.RecEncTrainer(cfg), [n for n, _ in self.named_parameters()])
        return self._train

    def engine(self):
        from . import hip
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._engine is None or key != self._engine_key:
            if self.use_sameres_feat:
                # upstream this switch cannot run: forward reads g.edges['rr'].data['a'] (receptor_encoder_gvp.py:230), a key no
                # dataset writes (pdbbind_processing.py:272 stores 'same_res'), and the rk convolutions are built with
                # edge_feat_size = 1 but called without edge features (:176-208, :279-281)
                raise NotImplementedError('use_sameres_feat=True cannot run in the reference GVP encoder (it reads an edge feature "a" '
                                          'that no dataset provides); every shipped config sets it to False')
            eng = hip.RecEncEngine(self.in_scalar_size, self.out_scalar_size, self.vector_size, self.n_rr_convs,
                                   self.n_rk_convs, self.n_message_gvps, self.n_update_gvps, self.message_norm,
                                   self.k_closest, self.n_keypoints, self.graph_cutoffs['rr'], self.graph_cutoffs['rk'],
                                   self.graph_cutoffs['kk'], kp_rad=self.kp_rad)
            eng.load_state_dict(self.state_dict())
            self._engine, self._engine_key = eng, key
        return self._engine

    def forward(self, g: HeteroBatch, batch_idxs: Dict[str, torch.Tensor] = None) -> HeteroBatch:
        """Writes keypoint x_0 / h_0 / v_0, replaces the rk edges by the kNN edges and adds the kk radius
        graph (receptor_encoder_gvp.py:212-294).  Under autograd (parameters requiring gradients) the call runs on the
        training engine and x_0 / h_0 / v_0 carry the graph back into the parameters; otherwise (torch.no_grad, the sampling
        paths) it runs on the fused inference engine, which has no dropout: call model.eval() there, as every sampling path does."""
        B, K = g.batch_size, self.n_keypoints
        if g.num_nodes('kp') != B * K:
            raise ValueError(f'expected {K} keypoint nodes per complex, graph has {g.num_nodes("kp")} for {B} complexes')
        rec = g.nodes['rec'].data
        rr_src, rr_dst = g.edges(etype='rr')
        n_rec = g.batch_num_nodes('rec')
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            out = {}
            x, h, v = _RecEncTrainFn.apply(self, n_rec, rec['x_0'], rec['h_0'], rr_src, rr_dst, out, *self.parameters())
            out = dict(out, kp_x=x, kp_h=h, kp_v=v)
        else:
            if self.training and self.dropout_rate > 0:
                raise NotImplementedError('train-mode dropout without autograd is not a path of the reference: call model.eval() for '
                                          'inference (every sampling path does)')
            out = self.engine().forward(n_rec, rec['x_0'], rec['h_0'], rr_src, rr_dst)
        kp = g.nodes['kp'].data
        kp['x_0'], kp['h_0'], kp['v_0'] = out['kp_x'], out['kp_h'], out['kp_v']
        nodes, edges = get_batch_info(g)
        g.remove_edges(g.edges(form='eid', etype='rk'), etype='rk')
        g.add_edges(out['rk_src'].long(), out['rk_dst'].long(), etype='rk')
        g.add_edges(out['kk_src'].long(), out['kk_dst'].long(), etype='kk')
        # rk edges per complex: K * min(k, n_rec) for the kNN graph, counted for the radius graph (:308-313)
        edges[('rec', 'rk', 'kp')] = torch.bincount(out['rk_dst'].long() // K, minlength=B).to(n_rec.device)
        edges[('kp', 'kk', 'kp')] = out['kk_per_graph'].long()
        g.set_batch_num_nodes(nodes)
        g.set_batch_num_edges(edges)
        return g
