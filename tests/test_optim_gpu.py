"""keypoint_diffusion_amd.optim (the one-launch clip + Adam of csrc/optim.hip) against torch.nn.utils.clip_grad_value_ + torch.optim.Adam, the pair
train.py:430-433, 541-543 runs: several steps with weight decay, a learning-rate change through param_groups (what the reference's Scheduler does),
tensors below, at and above the kernel's 4 096-element chunk, a parameter without a gradient, and a checkpoint that moves between the two
implementations."""
import copy

import pytest
import torch

from keypoint_diffusion_amd import optim

pytestmark = pytest.mark.gpu
SHAPES = [(257, 515), (1,), (4096,), (4097,), (16, 17), (3, 5, 7), (0,), (70001,)]


def _params(seed):
    gen = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(*s, generator=gen).cuda()) for s in SHAPES]


def _grads(ps, seed, skip=None):
    gen = torch.Generator().manual_seed(100 + seed)
    for i, p in enumerate(ps):
        p.grad = None if i == skip else (3.0 * torch.randn(*p.shape, generator=gen)).cuda()


def _close(a, b, tol=2e-6):
    return all(float((x.double() - y.double()).abs().max() / y.double().abs().max().clamp(min=1e-30)) < tol for x, y in zip(a, b) if x.numel())


@pytest.mark.parametrize('fused_clip', [False, True])
def test_steps_match_torch(cuda, fused_clip):
    ref, mine = _params(1), _params(1)
    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    o_ref = torch.optim.Adam(ref, **kw)
    o_mine = optim.Adam(mine, clip_value=1.5 if fused_clip else None, **kw)
    for t in range(6):
        skip = 4 if t == 2 else None                      # one step in which a parameter has no gradient: its moments and step count stand still
        _grads(ref, t, skip)
        _grads(mine, t, skip)
        if t == 3:
            for o in (o_ref, o_mine):
                o.param_groups[0]['lr'] = 2.5e-4
        torch.nn.utils.clip_grad_value_(ref, 1.5)
        if not fused_clip:
            optim.clip_grad_value_(mine, 1.5)
        o_ref.step()
        o_mine.step()
        torch.cuda.synchronize()
        assert all(torch.equal(a.grad, b.grad) for a, b in zip(mine, ref) if a.grad is not None), 'clamped gradients are written back'
        assert _close([p.data for p in mine], [p.data for p in ref]), t
    for a, b in zip(mine, ref):
        if a.numel():
            sa, sb = o_mine.state[a], o_ref.state[b]
            assert float(sa['step']) == float(sb['step'])
            assert _close([sa['exp_avg'], sa['exp_avg_sq']], [sb['exp_avg'], sb['exp_avg_sq']])


def test_checkpoint_moves_between_the_implementations(cuda):
    a, b = _params(2), _params(2)
    o_a, o_b = optim.Adam(a, lr=1e-3, weight_decay=1e-12), torch.optim.Adam(b, lr=1e-3, weight_decay=1e-12)
    for t in range(2):
        _grads(a, t)
        _grads(b, t)
        o_a.step()
        o_b.step()
    # torch's state into the library optimizer and the other way round, then one more step each
    sd_a, sd_b = copy.deepcopy(o_a.state_dict()), copy.deepcopy(o_b.state_dict())
    o_a.load_state_dict(sd_b)
    o_b.load_state_dict(sd_a)
    _grads(a, 9)
    _grads(b, 9)
    o_a.step()
    o_b.step()
    torch.cuda.synchronize()
    assert _close([p.data for p in a], [p.data for p in b])


def test_refusals(cuda):
    p = torch.nn.Parameter(torch.randn(8))            # a CPU parameter: the product has no CPU path
    p.grad = torch.randn(8)
    with pytest.raises(Exception):
        optim.Adam([p]).step()
    with pytest.raises(ValueError):
        optim.Adam(_params(3), betas=(1.0, 0.999))


def test_training_loop_with_either_optimizer(cuda):
    """The train.py inner loop (loss, backward, clip, step: train.py:523-543) on a denoiser, once with torch's optimizer and clamp and once with the
    library's: the trainers are bitwise repeatable, so the two runs see the same gradients up to the optimizers' rounding and end at the same weights."""
    from keypoint_diffusion_amd import graph as G, synth
    from keypoint_diffusion_amd.dynamics import LigRecDynamics
    from tests import util
    cut = util.CUTOFFS_ALL_ATOM
    g = util.fixed_encode(G.batch(synth.synth_complexes([50, 70], [10, 13], 20, cut, seed=11))).to('cuda')
    t = torch.tensor([0.3, 0.8], device='cuda')
    gen = torch.Generator().manual_seed(0)
    tgt_h, tgt_x = torch.randn(23, 10, generator=gen).cuda(), torch.randn(23, 3, generator=gen).cuda()
    runs = []
    for native in (False, True):
        model = LigRecDynamics(10, 10, graph_cutoffs=cut)
        synth.fill_state_dict_(model, 3)
        model = model.cuda().train()
        opt = (optim.Adam if native else torch.optim.Adam)(model.parameters(), lr=1e-3)
        clip = optim.clip_grad_value_ if native else torch.nn.utils.clip_grad_value_
        losses = []
        for _ in range(5):
            eh, ex = model(g, t, None)
            loss = (eh - tgt_h).square().mean() + (ex - tgt_x).square().mean()
            opt.zero_grad()
            loss.backward()
            clip(model.parameters(), 0.05)
            opt.step()
            losses.append(float(loss.detach()))
        runs.append((losses, [p.detach().clone() for p in model.parameters()]))
    (la, pa), (lb, pb) = runs
    assert la[-1] < la[0] and all(abs(x - y) <= 1e-4 * abs(x) for x, y in zip(la, lb)), (la, lb)
    worst = max(float((a - b).abs().max() / b.abs().max().clamp(min=1e-6)) for a, b in zip(pa, pb) if a.numel())
    assert worst < 1e-3, worst


def test_planned_steps_follow_moving_gradients(cuda):
    """After a first step in which every tensor took part, steps reuse the uploaded address table (optim.py, _StepPlan / _ClipPlan) and must notice
    every way it can go stale: gradients written in place (addresses stand), new gradient tensors, a strided gradient, a parameter that moved."""
    ref, mine = _params(5), _params(5)
    o_ref, o_mine = torch.optim.Adam(ref, lr=2e-3), optim.Adam(mine, lr=2e-3)
    keep = []
    for t in range(9):
        gen = torch.Generator().manual_seed(200 + t)
        for a, b in zip(mine, ref):
            g = (2.0 * torch.randn(*a.shape, generator=gen)).cuda()
            if t in (1, 2, 6) and a.grad is not None:                       # in place: the table stays valid
                a.grad.copy_(g); b.grad.copy_(g)
            elif t == 4 and a.dim() == 2:                                   # strided gradients on both sides
                a.grad, b.grad = g.t().contiguous().t(), g.t().contiguous().t()
                assert not a.grad.is_contiguous()
            else:                                                           # new tensors (the old ones kept alive, so the addresses differ)
                keep.append(a.grad)
                a.grad, b.grad = g.clone(), g.clone()
        if t == 7:                                                          # a parameter re-seated on new memory
            mine[2].data = mine[2].data.clone()
        torch.nn.utils.clip_grad_value_(ref, 1.0)
        optim.clip_grad_value_(mine, 1.0)
        o_ref.step()
        o_mine.step()
        torch.cuda.synchronize()
        if t in (1, 2, 6):
            assert o_mine._plans, 'the planned form was expected to be in use'
        assert all(torch.equal(a.grad, b.grad) for a, b in zip(mine, ref)), t
        assert _close([p.data for p in mine], [p.data for p in ref]), t
    for a, b in zip(mine, ref):
        if a.numel():
            sa, sb = o_mine.state[a], o_ref.state[b]
            assert float(sa['step']) == float(sb['step']) == 9.0
            assert _close([sa['exp_avg'], sa['exp_avg_sq']], [sb['exp_avg'], sb['exp_avg_sq']])
