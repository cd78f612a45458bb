"""The optimizer step of the training loop on the library's one-launch kernel (csrc/optim.hip, kpd_adam_step).

`Adam` keeps the interface of `torch.optim.Adam(params, lr, betas, eps, weight_decay)` -- train.py:430-433 builds exactly that, the reference's
`Scheduler` rewrites `param_groups[i]['lr']`, checkpoints go through `state_dict()` -- and `clip_grad_value_` that of
`torch.nn.utils.clip_grad_value_` (train.py:541-542).  torch runs the two as ~50 multi-tensor launches behind Python loops over the 360 - 440
parameter tensors of the shipped models; here each is one launch over a device table of tensor addresses (the addresses of the gradients are
refreshed every step: autograd may hand out new tensors).  Same arithmetic as torch/optim/adam.py with default flags (tests/test_optim_gpu.py)."""
import torch

from . import hip

__all__ = ['Adam', 'clip_grad_value_']


class _Table:
    """[n, 5] int64 rows (param, grad, exp_avg, exp_avg_sq, numel): two pinned host copies used in turn, one device copy.  The host may run
    several steps ahead of the GPU, so a host copy is rewritten only after the upload that last read it has completed (an event per copy)."""

    def __init__(self, n, device):
        self.host = [torch.zeros(n, 5, dtype=torch.int64).pin_memory() for _ in range(2)]
        self.sent = [None, None]
        self.dev = torch.zeros(n, 5, dtype=torch.int64, device=device)
        self.turn = 0

    def upload(self, cols):
        k = self.turn
        self.turn ^= 1
        h = self.host[k]
        if self.sent[k] is not None:
            self.sent[k].synchronize()
        for c, col in cols.items():
            h[:, c] = torch.tensor(col, dtype=torch.int64)
        with torch.cuda.device(self.dev.device):
            self.dev.copy_(h, non_blocking=True)
            if self.sent[k] is None:
                self.sent[k] = torch.cuda.Event()
            self.sent[k].record()
        return self.dev


def _dense(t, what):
    if not (t.is_cuda and t.dtype == torch.float32 and t.layout == torch.strided):
        raise hip.KpdError(f'{what}: fp32 CUDA tensors only (got {t.dtype} on {t.device})')
    return t


def clip_grad_value_(parameters, clip_value: float):
    """torch.nn.utils.clip_grad_value_ in one launch: every gradient clamped to [-clip_value, clip_value] in place."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = []
    for p in parameters:
        if p.grad is None:
            continue
        g = _dense(p.grad, 'clip_grad_value_')
        if not g.is_contiguous():
            p.grad = g = g.contiguous()
        if g.numel():
            grads.append(g)
    if not grads:
        return
    n = len(grads)
    tab = _Table(n, grads[0].device) if not hasattr(clip_grad_value_, '_tab') or clip_grad_value_._tab.dev.shape[0] != n else clip_grad_value_._tab
    clip_grad_value_._tab = tab
    numel = [g.numel() for g in grads]
    dev = tab.upload({1: [g.data_ptr() for g in grads], 4: numel})
    hip.adam_step(dev, max(numel), 1, clip_value=float(clip_value))


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam (no amsgrad, no maximize) with the whole step as one launch.  `clip_value` (an extension, default None): clamp the
    gradients inside the same launch instead of calling clip_grad_value_ first."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=None):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError(f'invalid Adam hyper-parameters: lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.clip_value = clip_value
        self._tabs = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            by_step = {}
            for p in group['params']:
                if p.grad is None or p.numel() == 0:
                    continue
                _dense(p, 'Adam')
                if p.grad.is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients')
                if not p.is_contiguous():
                    raise hip.KpdError('Adam: parameters must be contiguous')
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                by_step.setdefault(int(st['step'].item()), []).append(p)
            b1, b2 = group['betas']
            for t, ps in by_step.items():
                n = len(ps)
                key = (gi, n)
                tab = self._tabs.get(key)
                if tab is None or tab.dev.device != ps[0].device:
                    tab = self._tabs[key] = _Table(n, ps[0].device)
                numel = [p.numel() for p in ps]
                dev = tab.upload({0: [p.data_ptr() for p in ps], 1: [p.grad.data_ptr() for p in ps],
                                  2: [self.state[p]['exp_avg'].data_ptr() for p in ps], 3: [self.state[p]['exp_avg_sq'].data_ptr() for p in ps], 4: numel})
                hip.adam_step(dev, max(numel), 0, lr=group['lr'], beta1=b1, beta2=b2, eps=group['eps'], weight_decay=group['weight_decay'], step=t,
                              clip_value=float(self.clip_value) if self.clip_value else 0.0)
        return loss
