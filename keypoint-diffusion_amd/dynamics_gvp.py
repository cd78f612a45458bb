"""GVP denoiser behind the reference's module interface (models/dynamics_gvp.py:10-199).

Parameter containers with the reference state-dict layout; `forward` runs in libkpd_hip.so
(csrc/gvp.hip).
"""
from typing import Dict, Union

import torch
import torch.nn as nn

from . import hip
from .graph import HeteroBatch
from .gvp import GVP, GVPMultiEdgeConv


class NoisePredictionBlock(nn.Module):
    """dynamics_gvp.py:12-36: n GVPs, the last one to (64 scalars, 1 vector, identity vector act)."""

    def __init__(self, in_scalar_dim: int, out_scalar_dim: int, vector_size: int, n_gvps: int = 3,
                 intermediate_scalar_dim: int = 64):
        super().__init__()
        gvps = []
        for i in range(n_gvps):
            last = i == n_gvps - 1
            gvps.append(GVP(dim_vectors_in=vector_size, dim_vectors_out=1 if last else vector_size,
                            dim_feats_in=in_scalar_dim, dim_feats_out=intermediate_scalar_dim if last else in_scalar_dim,
                            vectors_activation=nn.Identity() if last else nn.Sigmoid()))
        self.gvps = nn.Sequential(*gvps)
        self.to_scalar_output = nn.Linear(intermediate_scalar_dim, out_scalar_dim)


class LigRecGVP(nn.Module):
    no_kp_update_edges = [('lig', 'll', 'lig'), ('kp', 'kl', 'lig')]
    kp_update_edges = no_kp_update_edges + [('lig', 'lk', 'kp'), ('kp', 'kk', 'kp')]

    def __init__(self, in_scalar_dim: int, in_vector_dim: int, out_scalar_dim: int, update_kp: bool = False,
                 n_convs: int = 4, n_message_gvps: int = 3, n_update_gvps: int = 2,
                 message_norm: Union[float, str, Dict] = 10, n_noise_gvps: int = 3, dropout: float = 0.0):
        super().__init__()
        self.update_kp = update_kp
        self.conv_layers = nn.ModuleList()
        for i in range(n_convs):
            # keypoints are not updated by the last convolution (dynamics_gvp.py:67-72)
            ets = self.kp_update_edges if (update_kp and i != n_convs - 1) else self.no_kp_update_edges
            self.conv_layers.append(GVPMultiEdgeConv(etypes=ets, scalar_size=in_scalar_dim, vector_size=in_vector_dim,
                                                     n_message_gvps=n_message_gvps, n_update_gvps=n_update_gvps,
                                                     message_norm=message_norm, dropout=dropout))
        self.noise_predictor = NoisePredictionBlock(in_scalar_dim=in_scalar_dim, out_scalar_dim=out_scalar_dim,
                                                    vector_size=in_vector_dim, n_gvps=n_noise_gvps)


class _GvpTrainFn(torch.autograd.Function):
    """LigRecDynamicsGVP.forward as one autograd node (kpd_gvp_trainer_*): differentiable in every parameter, in the scalar /
    vector input features and in the ligand / keypoint positions (learned keypoints: models/receptor_encoder_gvp.py:84-87)."""

    @staticmethod
    def forward(ctx, module, pb, timestep, lig_x, kp_x, lig_h, kp_h, kp_v, *params):
        trainer, names = module._trainer()
        ctx.trainer, ctx.names = trainer, names
        # the C side keeps raw pointers into the batch structure (per-complex offsets, the kk edge list) and reads them again in the
        # backward pass: the prepared batch must outlive the graph object the caller may drop right after the forward call
        ctx.pb = pb
        ctx.inputs = (lig_x, kp_x, lig_h, kp_h, kp_v, timestep)    # kept alive until backward (the C side holds pointers)
        # one forward's saved conv states per trainer: generation number + autograd's version check on the parameters
        # (see _EgnnTrainFn in dynamics.py)
        trainer.generation = getattr(trainer, 'generation', 0) + 1
        ctx.generation = trainer.generation
        ctx.save_for_backward(*params)
        trainer.bind(names, params, [None] * len(params))
        # GVPDropout is active in training mode only (gvp.py:133-134); the seed comes from torch's CPU generator, so
        # torch.manual_seed reproduces a step
        rate = module.dropout if module.training else 0.0
        module.last_dropout_seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if rate > 0 else 0
        trainer.set_dropout(rate, module.last_dropout_seed)
        return trainer.forward(pb, lig_x, lig_h, kp_x, kp_h, kp_v, timestep)

    @staticmethod
    def backward(ctx, d_eps_h, d_eps_x):
        if ctx.generation != ctx.trainer.generation:
            raise hip.KpdError('backward of a LigRecDynamicsGVP forward whose saved conv states were overwritten by a later '
                               'grad-enabled forward of the same module (one forward/backward pair at a time per module)')
        params = ctx.saved_tensors
        lig_x, kp_x, lig_h, kp_h, kp_v, _ = ctx.inputs
        grads = hip.zero_grads_like(params, [ctx.needs_input_grad[8 + i] for i in range(len(params))])
        ctx.trainer.bind(ctx.names, params, grads)
        d_in = [torch.empty_like(t) if n else None for t, n in zip((lig_h, kp_h, kp_v), ctx.needs_input_grad[5:8])]
        # positions enter through the unit edge vector and the rbf code of every edge (gvp.py:472-480); the edge lists are data
        d_x = [torch.empty_like(t) if n else None for t, n in zip((lig_x, kp_x), ctx.needs_input_grad[3:5])]
        ctx.trainer.backward(d_eps_h.contiguous().float(), d_eps_x.contiguous().float(), *d_in, *d_x)
        return (None, None, None, *d_x, *d_in, *grads)


class LigRecDynamicsGVP(nn.Module):

    def __init__(self, n_lig_scalars, n_kp_scalars, vector_size: int = 16, n_convs=4, n_hidden_scalars=128,
                 act_fn=nn.SiLU, message_norm=1, no_cg: bool = False, n_keypoints: int = 20, graph_cutoffs: dict = {},
                 update_kp: bool = False, ll_k: int = 0, kl_k: int = 0, n_message_gvps: int = 3, n_update_gvps: int = 2,
                 n_noise_gvps: int = 3, dropout: float = 0.0):
        super().__init__()
        if no_cg:
            raise NotImplementedError('No CG is not implemented for GVP')
        if act_fn is not nn.SiLU:
            raise NotImplementedError('only SiLU activations are implemented (every shipped config)')
        if not update_kp and n_convs > 1:
            # the reference drops 'kp' from node_data after the first conv and fails (gvp.py:501, 536)
            raise NotImplementedError('update_kp=False with more than one convolution cannot run in the reference')
        self.n_keypoints, self.graph_cutoffs, self.update_kp = n_keypoints, graph_cutoffs, update_kp
        self.n_lig_scalars, self.n_kp_scalars, self.vector_size = n_lig_scalars, n_kp_scalars, vector_size
        self.n_convs, self.n_hidden_scalars, self.message_norm = n_convs, n_hidden_scalars, message_norm
        self.n_message_gvps, self.n_update_gvps, self.n_noise_gvps = n_message_gvps, n_update_gvps, n_noise_gvps
        self.ll_k, self.kl_k = ll_k, kl_k
        self.lig_encoder = nn.Sequential(nn.Linear(n_lig_scalars + 1, n_hidden_scalars), act_fn(), nn.LayerNorm(n_hidden_scalars))
        self.kp_encoder = nn.Sequential(nn.Linear(n_kp_scalars + 1, n_hidden_scalars), act_fn(), nn.LayerNorm(n_hidden_scalars))
        self.noise_predictor = LigRecGVP(in_scalar_dim=n_hidden_scalars, in_vector_dim=vector_size,
                                         out_scalar_dim=n_lig_scalars, update_kp=update_kp, n_convs=n_convs,
                                         n_message_gvps=n_message_gvps, n_update_gvps=n_update_gvps,
                                         n_noise_gvps=n_noise_gvps, message_norm=message_norm, dropout=dropout)
        self.dropout = dropout
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
            mode, val = hip._norm_mode(self.message_norm)
            cfg = hip.KpdGvpConfig(int(self.n_lig_scalars), int(self.n_kp_scalars), int(self.vector_size), int(self.n_convs),
                                   int(self.n_hidden_scalars), int(bool(self.update_kp)), mode, val, int(self.ll_k), int(self.kl_k),
                                   float(self.graph_cutoffs.get('ll', 0.0)), float(self.graph_cutoffs.get('kl', 0.0)),
                                   int(self.n_message_gvps), int(self.n_update_gvps), int(self.n_noise_gvps))
            self._train = (hip.GvpTrainer(cfg), [n for n, _ in self.named_parameters()])
        return self._train

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

    def engine(self) -> 'hip.GvpEngine':
        """(Re)build the device engine when weights were replaced or modified in place."""
        key = self._weights_key()
        if self._engine is None or key != self._engine_key:
            eng = hip.GvpEngine(self.n_lig_scalars, self.n_kp_scalars, self.vector_size, self.n_convs,
                                self.n_hidden_scalars, self.update_kp, self.message_norm, self.ll_k, self.kl_k,
                                self.graph_cutoffs.get('ll', 0.0), self.graph_cutoffs.get('kl', 0.0), self.n_message_gvps,
                                self.n_update_gvps, self.n_noise_gvps)
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
        """Predicted noise (eps_h [N_lig, n_lig_scalars], eps_x [N_lig, 3]) -- eval mode (dropout is the identity)."""
        pb = g.prepared()
        lig, kp = g.nodes['lig'].data, g.nodes['kp'].data
        if self.training and self.dropout > 0 and not torch.is_grad_enabled():
            raise NotImplementedError('train-mode dropout without autograd is not a path of the reference: call model.eval() for '
                                      'inference (every sampling path does)')
        if torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or
                                        any(kp[k].requires_grad or (k in lig and lig[k].requires_grad) for k in ('h_0', 'v_0', 'x_0'))):
            ins = [hip._dev_f32(t, n) for t, n in ((lig['x_0'], 'lig x_0'), (kp['x_0'], 'kp x_0'), (lig['h_0'], 'lig h_0'),
                                                   (kp['h_0'], 'kp h_0'), (kp['v_0'], 'kp v_0'))]
            return _GvpTrainFn.apply(self, pb, hip._dev_f32(timestep, 'timestep'), *ins, *self.parameters())
        return self.engine().forward(pb, lig['x_0'], lig['h_0'], kp['x_0'], kp['h_0'], kp['v_0'], timestep)
