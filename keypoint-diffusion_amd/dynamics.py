"""EGNN denoiser behind the reference's module interface.

`LigRecDynamics` keeps the constructor signature, `forward(g, timestep, batch_idxs)` contract and
state-dict layout of models/dynamics.py:298-385 (LigRecConv :9-87, LigRecEGNN :221-264), so a
reference `model.pt` loads unchanged.  The modules below only own parameters; the arithmetic of
`forward` -- encoders, per-step radius/kNN graph build, the EGNN stack and the decoder -- runs in
libkpd_hip.so (csrc/egnn.hip).  With gradients enabled the call goes through `_EgnnTrainFn`: forward with saved
layer states and the hand-derived backward pass of csrc/egnn_train.hip (`kpd_egnn_trainer_*`), so train.py-style
loops differentiate the denoiser without a PyTorch implementation of it.
"""
from typing import Dict

import torch
import torch.nn as nn

from . import hip
from .graph import HeteroBatch


def _mlp2(n_in, n_hidden, n_out, final_act):
    layers = [nn.Linear(n_in, n_hidden), nn.SiLU(), nn.Linear(n_hidden, n_out)]
    if final_act:
        layers.append(nn.SiLU())
    return nn.Sequential(*layers)


class _EgnnTrainFn(torch.autograd.Function):
    """LigRecDynamics.forward as one autograd node: inputs (lig x, lig h, kp x, kp h) and every parameter; the backward
    pass is kpd_egnn_trainer_backward (gradients in the reference parameter layout)."""

    @staticmethod
    def forward(ctx, module, pb, timestep, lig_x, lig_h, kp_x, kp_h, *params):
        trainer, names = module._trainer()
        ctx.trainer, ctx.names = trainer, names
        # the C side keeps raw pointers into the batch structure (per-complex offsets, the kk edge list) and reads them again in the
        # backward pass: the prepared batch must outlive the graph object the caller may drop right after the forward call
        ctx.pb = pb
        ctx.inputs = (lig_x, lig_h, kp_x, kp_h, timestep)          # kept alive until backward (the C side holds pointers)
        # The trainer keeps the saved layer states of ONE forward.  Every forward takes a new generation number; backward
        # refuses to run on a workspace a later forward has overwritten.  The parameters go through save_for_backward, so
        # autograd's version check catches an in-place update between forward and backward (the C side reads them in place).
        trainer.generation = getattr(trainer, 'generation', 0) + 1
        ctx.generation = trainer.generation
        ctx.save_for_backward(*params)
        trainer.bind(names, params, [None] * len(params))
        eps_h, eps_x = trainer.forward(pb, lig_x, lig_h, kp_x, kp_h, timestep)
        return eps_h, eps_x

    @staticmethod
    def backward(ctx, d_eps_h, d_eps_x):
        if ctx.generation != ctx.trainer.generation:
            raise hip.KpdError('backward of a LigRecDynamics forward whose saved layer states were overwritten by a later grad-enabled '
                               'forward of the same module (one forward/backward pair at a time per module)')
        params = ctx.saved_tensors
        lig_x, lig_h, kp_x, kp_h, _ = ctx.inputs
        need = ctx.needs_input_grad[3:7]
        grads = hip.zero_grads_like(params, [ctx.needs_input_grad[7 + i] for i in range(len(params))])
        ctx.trainer.bind(ctx.names, params, grads)
        d_in = [torch.empty_like(t) if n else None for t, n in zip((lig_x, lig_h, kp_x, kp_h), need)]
        ctx.trainer.backward(d_eps_h.contiguous().float(), d_eps_x.contiguous().float(), d_in[1], d_in[0], d_in[3], d_in[2])
        return (None, None, None, *d_in, *grads)


class LigRecConv(nn.Module):
    """Weights of one heterogeneous EGNN layer (models/dynamics.py:15-87)."""

    def __init__(self, in_size, hidden_size, out_size, edge_feat_size=0, use_tanh=False, coords_range=10,
                 update_kp_feat: bool = False, norm: bool = False):
        super().__init__()
        if edge_feat_size:
            raise NotImplementedError('edge features are not implemented (as in the reference, dynamics.py:153)')
        self.in_size, self.hidden_size, self.out_size = in_size, hidden_size, out_size
        self.use_tanh, self.coords_range, self.update_kp_feat, self.norm = use_tanh, coords_range, update_kp_feat, norm
        self.edge_types = ['ll', 'kl', 'lk', 'kk'] if update_kp_feat else ['ll', 'kl']
        self.updated_node_types = ['lig', 'kp'] if update_kp_feat else ['lig']
        f_in = 2 * in_size + 1                                   # [h_src, h_dst, d_ij]
        self.edge_mlp = nn.ModuleDict({et: _mlp2(f_in, hidden_size, hidden_size, True) for et in self.edge_types})
        self.soft_attention = nn.ModuleDict(
            {et: nn.Sequential(nn.Linear(hidden_size, 1), nn.Sigmoid()) for et in self.edge_types})
        self.node_mlp = nn.ModuleDict(
            {nt: _mlp2(in_size + hidden_size, hidden_size, out_size, False) for nt in self.updated_node_types})
        self.coord_mlp = nn.ModuleDict()
        for et in self.edge_types:
            head = nn.Linear(hidden_size, 1, bias=False)
            nn.init.xavier_uniform_(head.weight, gain=0.001)
            self.coord_mlp[et] = nn.Sequential(*_mlp2(f_in, hidden_size, hidden_size, True), head)
        self.layer_norm = nn.ModuleDict(
            {nt: nn.LayerNorm(out_size) if norm else nn.Identity() for nt in self.updated_node_types})


class LigRecEGNN(nn.Module):
    def __init__(self, n_layers, in_size, hidden_size, out_size, use_tanh=False, message_norm=1,
                 update_kp_feat: bool = False, norm: bool = False):
        super().__init__()
        if not (in_size == hidden_size == out_size):
            raise NotImplementedError('the fused layers assume in = hidden = out width (true for LigRecDynamics)')
        self.n_layers, self.message_norm = n_layers, message_norm
        self.update_kp_feat, self.norm = update_kp_feat, norm
        self.conv_layers = nn.ModuleList([
            LigRecConv(in_size, hidden_size, out_size, use_tanh=use_tanh, update_kp_feat=update_kp_feat, norm=norm)
            for _ in range(n_layers)])


class LigRecDynamics(nn.Module):

    def __init__(self, atom_nf, rec_nf, n_layers=4, hidden_nf=255, act_fn=nn.SiLU, use_tanh=False, message_norm=1,
                 no_cg: bool = False, n_keypoints: int = 20, graph_cutoffs: dict = {}, update_kp_feat: bool = False,
                 norm: bool = False, ll_k: int = 0, kl_k: int = 0):
        super().__init__()
        if act_fn is not nn.SiLU:
            raise NotImplementedError('only SiLU activations are implemented (every shipped config)')
        self.atom_nf, self.rec_nf, self.n_layers, self.hidden_nf = atom_nf, rec_nf, n_layers, hidden_nf
        self.use_tanh, self.message_norm, self.norm = use_tanh, message_norm, norm
        self.no_cg, self.n_keypoints, self.graph_cutoffs = no_cg, n_keypoints, graph_cutoffs
        self.update_kp_feat, self.ll_k, self.kl_k = update_kp_feat, ll_k, kl_k

        self.lig_encoder = _mlp2(atom_nf, 64, hidden_nf, True)
        self.lig_decoder = _mlp2(hidden_nf, 2 * atom_nf, atom_nf, False)
        self.rec_encoder = _mlp2(rec_nf, 2 * rec_nf, hidden_nf, True) if rec_nf != hidden_nf else nn.Identity()
        # +1: the timestep is appended to the encoded features (dynamics.py:337, 359-363)
        self.egnn = LigRecEGNN(n_layers=n_layers, in_size=hidden_nf + 1, hidden_size=hidden_nf + 1,
                               out_size=hidden_nf + 1, use_tanh=use_tanh, message_norm=message_norm,
                               update_kp_feat=update_kp_feat, norm=norm)
        self._engine = None
        self._engine_key = None
        # GEMM mode of the inference engine: None = the library default (exact fp32 MFMA, or what KPD_GEMM names when the engine
        # is created); 'f32' | 'f16x2' = an explicit choice that is re-applied to EVERY engine this module builds (weights
        # replaced, optimizer step, .to()), so a model cannot silently fall back to another mode.
        self.gemm_mode = None
        self._train = None

    def _trainer(self):
        """The training engine and the parameter names in `self.parameters()` order (reference state-dict names)."""
        if self._train is None:
            if isinstance(self.message_norm, (dict, str)):
                raise ValueError(f'message_norm must be a number for the EGNN denoiser, got {self.message_norm!r}')
            cfg = hip.KpdEgnnConfig(int(self.atom_nf), int(self.rec_nf), int(self.n_layers), int(self.hidden_nf),
                                    int(bool(self.use_tanh)), int(bool(self.norm)), int(bool(self.update_kp_feat)),
                                    float(self.message_norm), int(self.ll_k), int(self.kl_k),
                                    float(self.graph_cutoffs.get('ll', 0.0)), float(self.graph_cutoffs.get('kl', 0.0)), 10.0)
            self._train = (hip.EgnnTrainer(cfg, self.atom_nf, self.rec_nf), [n for n, _ in self.named_parameters()])
        return self._train

    # ---- HIP engine management ---------------------------------------------------------
    def _weights_key(self):
        """(storage pointer, version counter) of every parameter: changes when weights are replaced or modified in place.
        Walking the module tree costs ~0.7 ms of host time (hundreds of tensors), more than a B = 1 reverse step takes on the GPU,
        so the list of Parameter objects is cached.  It is rebuilt after `.to()` / `load_state_dict` and whenever ANY module of the
        process registered a parameter since it was built (`hip.param_generation`: `module.weight = nn.Parameter(...)`, parametrize
        and pruning all go through `register_parameter`), so a Parameter object swapped in deep inside the module is seen by the
        next forward.  Unsupported as an immediate trigger: writes into `module._parameters` that bypass `register_parameter` (seen by
        the periodic re-walk below)."""
        gen = hip.param_generation()
        ps = self.__dict__.get('_param_list')
        # every 256th call walks the tree again whatever the hook said: mutations that bypass `register_parameter` (a direct
        # `module._parameters[name] = p`, `__setstate__` / deepcopy swaps) are then seen after at most 256 forwards instead of never
        n = self.__dict__['_param_calls'] = self.__dict__.get('_param_calls', 0) + 1
        if ps is None or self.__dict__.get('_param_gen') != gen or (n & 255) == 0:
            ps = self.__dict__['_param_list'] = list(self.parameters())
            self.__dict__['_param_gen'] = gen
        return tuple([(p.data_ptr(), p._version) for p in ps])

    def _apply(self, fn, *a, **kw):
        self.__dict__.pop('_param_list', None)
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self.__dict__.pop('_param_list', None)
        return super().load_state_dict(*a, **kw)

    def engine(self) -> 'hip.EgnnEngine':
        """(Re)build the device engine when weights were replaced or modified in place."""
        key = self._weights_key()
        if self._engine is None or key != self._engine_key:
            if isinstance(self.message_norm, (dict, str)):
                raise ValueError(f'message_norm must be a number for the EGNN denoiser, got {self.message_norm!r}')
            eng = hip.EgnnEngine(self.atom_nf, self.rec_nf, self.n_layers, self.hidden_nf, self.use_tanh, self.norm,
                                 self.update_kp_feat, self.message_norm, self.ll_k, self.kl_k,
                                 self.graph_cutoffs.get('ll', 0.0), self.graph_cutoffs.get('kl', 0.0))
            eng.load_state_dict(self.state_dict())
            self._engine, self._engine_key = eng, key
        if self.gemm_mode is not None and getattr(self._engine, '_mode_applied', None) != self.gemm_mode:
            if self.gemm_mode not in ('f32', 'f16x2'):
                raise ValueError(f"gemm_mode must be None, 'f32' or 'f16x2', got {self.gemm_mode!r}")
            self._engine.set_gemm_mode(self.gemm_mode)            # (one library call per engine and choice, not per forward)
            if self._engine.gemm_mode() != self.gemm_mode:        # an explicit choice is never dropped silently
                raise hip.KpdError(f'gemm_mode={self.gemm_mode!r} was requested but the engine runs {self._engine.gemm_mode()!r} '
                                   f'(the f16x2 mode of the GVP denoiser needs n_hidden_scalars = 256)')
            self._engine._mode_applied = self.gemm_mode
        return self._engine

    def forward(self, g: HeteroBatch, timestep: torch.Tensor, batch_idxs: Dict[str, torch.Tensor] = None):
        """Predicted noise (eps_h [N_lig, atom_nf], eps_x [N_lig, 3]); `batch_idxs` is accepted for
        signature compatibility, the per-complex offsets are taken from the graph's batch info."""
        pb = g.prepared()
        lig, kp = g.nodes['lig'].data, g.nodes['kp'].data
        if torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or
                                        any(t.requires_grad for t in (lig['x_0'], lig['h_0'], kp['x_0'], kp['h_0']))):
            ins = [hip._dev_f32(t, n) for t, n in ((lig['x_0'], 'lig x_0'), (lig['h_0'], 'lig h_0'), (kp['x_0'], 'kp x_0'),
                                                   (kp['h_0'], 'kp h_0'))]
            return _EgnnTrainFn.apply(self, pb, hip._dev_f32(timestep, 'timestep'), *ins, *self.parameters())
        return self.engine().forward(pb, lig['x_0'], lig['h_0'], kp['x_0'], kp['h_0'], timestep)
