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
    """[n, 5] int64 rows (param, grad, exp_avg, exp_avg_sq, numel): the rows, two pinned staging copies used in turn, one device copy.  The host may run
    several steps ahead of the GPU, so a host copy is rewritten only after the upload that last read it has completed (an event per copy)."""

    def __init__(self, n, device):
        self.rows = torch.zeros(n, 5, dtype=torch.int64)                  # the current table; uploads may rewrite some columns only
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
            self.rows[:, c] = torch.tensor(col, dtype=torch.int64)
        h.copy_(self.rows)
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


class _ClipPlan:
    tab = None
    key = None


def clip_grad_value_(parameters, clip_value: float):
    """torch.nn.utils.clip_grad_value_ in one launch: every gradient clamped to [-clip_value, clip_value] in place.  The address table is uploaded
    again only when a gradient moved since the last call (the caching allocator usually hands the same blocks out step after step)."""
    parameters = [parameters] if isinstance(parameters, torch.Tensor) else list(parameters)
    owners = [p for p in parameters if p.grad is not None]
    grads = [p.grad for p in owners]
    key = [(g.data_ptr(), g.numel()) if g.is_contiguous() else None for g in grads]
    plan = _ClipPlan
    if key != plan.key or None in key:
        live = []
        for p, g in zip(owners, grads):
            _dense(g, 'clip_grad_value_')
            if not g.is_contiguous():                        # a strided gradient: replaced by a dense copy on its parameter
                p.grad = g = g.contiguous()
            if g.numel():
                live.append(g)
        if not live:
            return
        if plan.tab is None or plan.tab.dev.shape[0] != len(live) or plan.tab.dev.device != live[0].device:
            plan.tab = _Table(len(live), live[0].device)
        plan.max_numel = max(g.numel() for g in live)
        plan.tab.upload({1: [g.data_ptr() for g in live], 4: [g.numel() for g in live]})
        plan.key = key if None not in key else None
    elif not key:
        return
    hip.adam_step(plan.tab.dev, plan.max_numel, 1, clip_value=float(clip_value))


class _StepPlan:
    """One parameter group whose tensors all took the last step together: the static columns of the table, the step tensors, the count."""

    def __init__(self, params, states, tab, t):
        self.params, self.tab, self.t = params, tab, t
        self.steps = [st['step'] for st in states]
        self.max_numel = max(p.numel() for p in params)
        self.key = None


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam (no amsgrad, no maximize) with the whole step as one launch.  `clip_value` (an extension, default None): clamp the
    gradients inside the same launch instead of calling clip_grad_value_ first.

    Host side: the first step of a group walks its tensors like torch does (state creation, checks); once every tensor of the group has taken a
    step with the same count, later steps only compare the parameter / gradient addresses with the uploaded table (825 tensors of gvp_40kp:
    3.2 -> about 1 ms of host time per clip + step, torch 3.9) and fall back to the walk whenever anything differs (a missing or strided gradient,
    a new parameter, a loaded state)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=None):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError(f'invalid Adam hyper-parameters: lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.clip_value = clip_value
        self._tabs = {}
        self._plans = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans = {}

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._plans = {}

    def _launch(self, group, dev, max_numel, t):
        b1, b2 = group['betas']
        hip.adam_step(dev, max_numel, 0, lr=group['lr'], beta1=b1, beta2=b2, eps=group['eps'], weight_decay=group['weight_decay'], step=t,
                      clip_value=float(self.clip_value) if self.clip_value else 0.0)

    def _planned_step(self, group, plan):
        if len(group['params']) != plan.n_group:
            return False
        grads = [p.grad for p in plan.params]
        if any(g is None for g in grads):                   # (`None in grads` would compare tensors with None one by one)
            return False
        key = [(p.data_ptr(), g.data_ptr()) if g.is_contiguous() else None for p, g in zip(plan.params, grads)]
        if key != plan.key:
            if None in key:
                return False
            for p, g in zip(plan.params, grads):
                _dense(g, 'Adam')
                if not p.is_contiguous():
                    return False
            plan.tab.upload({0: [k[0] for k in key], 1: [k[1] for k in key]})
            plan.key = key
        torch._foreach_add_(plan.steps, 1.0)
        plan.t += 1
        self._launch(group, plan.tab.dev, plan.max_numel, plan.t)
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            plan = self._plans.get(gi)
            if plan is not None and self._planned_step(group, plan):
                continue
            self._plans.pop(gi, None)
            by_step = {}
            for p in group['params']:
                if p.grad is None or p.numel() == 0:
                    continue
                _dense(p, 'Adam')
                if p.grad.is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients')
                _dense(p.grad, 'Adam')
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
            for t, ps in by_step.items():
                n = len(ps)
                key = (gi, n)
                tab = self._tabs.get(key)
                if tab is None or tab.dev.device != ps[0].device:
                    tab = self._tabs[key] = _Table(n, ps[0].device)
                numel = [p.numel() for p in ps]
                states = [self.state[p] for p in ps]
                dev = tab.upload({0: [p.data_ptr() for p in ps], 1: [p.grad.data_ptr() for p in ps], 2: [st['exp_avg'].data_ptr() for st in states],
                                  3: [st['exp_avg_sq'].data_ptr() for st in states], 4: numel})
                self._launch(group, dev, max(numel), t)
                if len(by_step) == 1 and n == sum(1 for p in group['params'] if p.numel()):
                    # the whole group moves together: the next steps take the planned form (this table then belongs to the plan alone)
                    plan = self._plans[gi] = _StepPlan(ps, states, tab, t)
                    plan.n_group = len(group['params'])
                    plan.key = [(p.data_ptr(), p.grad.data_ptr()) for p in ps]
                    del self._tabs[key]
        return loss
