"""EGNN keypoint receptor encoder behind the reference's module interface (models/receptor_encoder.py:14-555).
Parameter containers with the reference state-dict layout; `forward` runs in libkpd_hip.so (kpd_recegnn_*).
"""
from typing import Dict

import torch
import torch.nn as nn

from .graph import HeteroBatch, get_batch_info


class _RecEgnnTrainFn(torch.autograd.Function):
    """ReceptorEncoder.forward as one autograd node (kpd_recegnn_trainer_*): keypoint positions and features are differentiable
    functions of every parameter; the rk / kk edge lists and the learned receptor h / x it also produces are side outputs."""

    @staticmethod
    def forward(ctx, module, rec_counts, rec_x, rec_h, rr_src, rr_dst, same_res, holder, *params):
        trainer, names = module._trainer()
        ctx.trainer, ctx.names = trainer, names
        trainer.generation = getattr(trainer, 'generation', 0) + 1
        ctx.generation = trainer.generation
        ctx.save_for_backward(*params)
        trainer.bind(names, params, [None] * len(params))
        out = trainer.forward(rec_counts, rec_x, rec_h, rr_src, rr_dst, same_res)
        ctx.keep = out.pop('_keep')                                 # device copies the C side reads again in backward
        holder.update(out)
        return out['kp_x'], out['kp_h']

    @staticmethod
    def backward(ctx, d_x, d_h):
        from . import hip
        if ctx.generation != ctx.trainer.generation:
            raise hip.KpdError('backward of a ReceptorEncoder forward whose saved states were overwritten by a later grad-enabled '
                               'forward of the same module (one forward/backward pair at a time per module)')
        params = ctx.saved_tensors
        grads = hip.zero_grads_like(params, [ctx.needs_input_grad[8 + i] for i in range(len(params))])
        ctx.trainer.bind(ctx.names, params, grads)
        c = lambda t: None if t is None else t.contiguous().float()
        ctx.trainer.backward(c(d_x), c(d_h))
        return (None,) * 8 + tuple(grads)


class ReceptorConv(nn.Module):
    """receptor_encoder.py:17-66 (parameters only)."""

    def __init__(self, in_size, hidden_size, out_size, edge_feat_size=0, use_tanh=True, coords_range=10, message_norm=1,
                 fix_pos: bool = False, norm: bool = False):
        super().__init__()
        self.in_size, self.hidden_size, self.out_size, self.edge_feat_size = in_size, hidden_size, out_size, edge_feat_size
        self.use_tanh, self.coords_range, self.message_norm, self.fix_pos, self.norm = use_tanh, coords_range, message_norm, fix_pos, norm
        act_fn = nn.SiLU()
        self.edge_mlp = nn.Sequential(nn.Linear(in_size * 2 + edge_feat_size + 1, hidden_size), act_fn,
                                      nn.Linear(hidden_size, hidden_size), act_fn)
        self.node_mlp = nn.Sequential(nn.Linear(in_size + hidden_size, hidden_size), act_fn, nn.Linear(hidden_size, out_size))
        self.soft_attention = nn.Sequential(nn.Linear(hidden_size, 1), nn.Sigmoid())
        self.layer_norm = nn.LayerNorm(out_size) if norm else nn.Identity()
        if fix_pos:
            return
        coord_output_layer = nn.Linear(hidden_size, 1, bias=False)
        nn.init.xavier_uniform_(coord_output_layer.weight, gain=0.001)
        self.coord_mlp = nn.Sequential(nn.Linear(in_size * 2 + edge_feat_size + 1, hidden_size), act_fn, coord_output_layer)


class RecKeyConv(nn.Module):
    """receptor_encoder.py:158-180 (parameters only; fc_dst exists upstream but is never applied)."""

    def __init__(self, in_feats: int, out_feats: int, n_keypoints: int, num_heads: int = 1, k_closest: int = 0, kp_rad: float = 0,
                 fix_pos: bool = False, norm: bool = False):
        super().__init__()
        self.num_heads, self.out_feats, self.n_keypoints = num_heads, out_feats, n_keypoints
        self.fix_pos, self.k_closest, self.kp_rad, self.norm = fix_pos, k_closest, kp_rad, norm
        self.fc_src = nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.fc_dst = nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.kp_feature_mlp = nn.Sequential(nn.Linear(out_feats + self.k_closest, out_feats), nn.SiLU())
        self.layer_norm = nn.LayerNorm(out_feats) if norm else nn.Identity()


class ReceptorEncoder(nn.Module):

    def __init__(self, n_convs: int = 6, n_keypoints: int = 10, graph_cutoffs: dict = {}, in_n_node_feat: int = 13,
                 use_sameres_feat: bool = False, hidden_n_node_feat: int = 256, out_n_node_feat: int = 256, use_tanh=True,
                 coords_range=10, kp_feat_scale=1, message_norm=1, kp_rad: float = 0, k_closest: int = 0, norm: bool = False,
                 no_cg=False, fix_pos=False, n_kk_convs: int = 0, n_kk_heads: int = 4):
        super().__init__()
        if kp_rad != 0 and k_closest != 0:
            raise ValueError('one of kp_rad and kp_closest can be zero but not both')
        elif kp_rad == 0 and k_closest == 0:
            raise ValueError('one of kp_rad and kp_closest must be non-zero')
        if no_cg:
            raise NotImplementedError
        if n_kk_convs > 0:
            raise NotImplementedError('KeyKeyConv.forward raises NotImplementedError upstream (receptor_encoder.py:337)')
        self.n_convs, self.n_keypoints, self.out_n_node_feat, self.kp_feat_scale = n_convs, n_keypoints, out_n_node_feat, kp_feat_scale
        self.in_n_node_feat, self.hidden_n_node_feat = in_n_node_feat, hidden_n_node_feat
        self.kp_pos_norm = out_n_node_feat ** 0.5
        self.k_closest, self.kp_rad, self.no_cg, self.fix_pos = k_closest, kp_rad, no_cg, fix_pos
        self.use_sameres_feat, self.message_norm, self.graph_cutoffs = use_sameres_feat, message_norm, graph_cutoffs
        self.use_tanh, self.coords_range, self.norm, self.n_kk_convs = use_tanh, coords_range, norm, n_kk_convs
        convs = []
        for i in range(n_convs):                                               # :431-459
            in_size = in_n_node_feat if i == 0 else hidden_n_node_feat
            out_size = out_n_node_feat if i == n_convs - 1 else hidden_n_node_feat
            convs.append(ReceptorConv(in_size=in_size, hidden_size=hidden_n_node_feat, out_size=out_size, use_tanh=use_tanh,
                                      coords_range=coords_range, message_norm=message_norm, norm=norm, fix_pos=fix_pos,
                                      edge_feat_size=1 if use_sameres_feat else 0))
        self.rec_convs = nn.ModuleList(convs)
        self.keypoint_embedding = nn.Sequential(nn.Linear(out_n_node_feat, out_n_node_feat * n_keypoints), nn.SiLU())
        self.rec_kp_conv = RecKeyConv(in_feats=out_n_node_feat, out_feats=out_n_node_feat, n_keypoints=n_keypoints, fix_pos=fix_pos,
                                      num_heads=1, k_closest=k_closest, kp_rad=kp_rad, norm=norm)
        self._engine = None
        self._engine_key = None
        self._train = None

    def _trainer(self):
        """The training engine and the parameter names in `self.parameters()` order (reference state-dict names)."""
        from . import hip
        if self._train is None:
            cfg = hip.KpdRecegnnConfig(int(self.n_convs), int(self.n_keypoints), int(self.in_n_node_feat), int(self.hidden_n_node_feat),
                                       int(self.out_n_node_feat), int(bool(self.use_sameres_feat)), int(bool(self.use_tanh)),
                                       int(bool(self.norm)), int(bool(self.fix_pos)), float(self.coords_range), float(self.message_norm),
                                       int(self.k_closest), float(self.graph_cutoffs['kk']), float(self.kp_rad))
            self._train = (hip.RecEgnnTrainer(cfg), [n for n, _ in self.named_parameters()])
        return self._train

    def engine(self):
        from . import hip
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._engine is None or key != self._engine_key:
            eng = hip.RecEgnnEngine(self.n_convs, self.n_keypoints, self.in_n_node_feat, self.hidden_n_node_feat, self.out_n_node_feat,
                                    self.use_sameres_feat, self.use_tanh, self.coords_range, self.message_norm, self.k_closest,
                                    self.norm, self.fix_pos, self.graph_cutoffs['kk'], kp_rad=self.kp_rad)
            eng.load_state_dict(self.state_dict())
            self._engine, self._engine_key = eng, key
        return self._engine

    def forward(self, g: HeteroBatch, batch_idxs: Dict[str, torch.Tensor] = None) -> HeteroBatch:
        """Writes keypoint x_0 / h_0, replaces the rk edges by the kNN edges and adds the kk radius graph
        (receptor_encoder.py:483-555).  Under autograd the call runs on the training engine; otherwise on the fused inference one."""
        B, K = g.batch_size, self.n_keypoints
        if g.num_nodes('kp') != B * K:
            raise ValueError(f'expected {K} keypoint nodes per complex, graph has {g.num_nodes("kp")} for {B} complexes')
        rec = g.nodes['rec'].data
        rr_src, rr_dst = g.edges(etype='rr')
        n_rec = g.batch_num_nodes('rec')
        same_res = g.edges['rr'].data['same_res'] if self.use_sameres_feat else None
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training: keypoint x_0 / h_0 carry the graph back into the parameters (kpd_recegnn_trainer_*)
            out = {}
            x, h = _RecEgnnTrainFn.apply(self, n_rec, rec['x_0'], rec['h_0'], rr_src, rr_dst, same_res, out, *self.parameters())
            out = dict(out, kp_x=x, kp_h=h)
        else:
            out = self.engine().forward(n_rec, rec['x_0'], rec['h_0'], rr_src, rr_dst, same_res)
        rec['x'], rec['h'] = out['rec_x'], out['rec_h']                        # :516-517
        kp = g.nodes['kp'].data
        kp['x_0'], kp['h_0'] = out['kp_x'], out['kp_h']
        nodes, edges = get_batch_info(g)
        g.remove_edges(g.edges(form='eid', etype='rk'), etype='rk')
        g.add_edges(out['rk_src'].long(), out['rk_dst'].long(), etype='rk')
        g.add_edges(out['kk_src'].long(), out['kk_dst'].long(), etype='kk')
        # K * k_closest per complex for the kNN features (:275), counted for the radius features (:249)
        edges[('rec', 'rk', 'kp')] = torch.bincount(out['rk_dst'].long() // K, minlength=B).to(n_rec.device)
        edges[('kp', 'kk', 'kp')] = out['kk_per_graph'].long()
        g.set_batch_num_nodes(nodes)
        g.set_batch_num_edges(edges)
        return g
