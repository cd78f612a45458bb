"""Backward pass of the EGNN denoiser: gradients of every parameter and of the inputs from
kpd_egnn_trainer_* against torch autograd through the CPU oracle (the restated LigRecDynamics.forward)."""
import pytest
import torch

from keypoint_diffusion_amd import graph as G, synth
from keypoint_diffusion_amd.dynamics import LigRecDynamics
from oracle import egnn as oegnn
from tests import util

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('gemm_mode')]   # both GEMM modes of the EGNN edge kernel (conftest.py)
CUT = util.CUTOFFS_ALL_ATOM
TOL = 2e-4          # relative to the largest entry of each gradient tensor (fp32 both sides, different summation order)


def _case(cfg, n_rec, n_lig, rec_nf=10, seed=5):
    gs = synth.synth_complexes(n_rec, n_lig, 20, CUT, seed=seed, n_rec_feat=rec_nf)
    g = util.fixed_encode(G.batch(gs))
    model = LigRecDynamics(10, rec_nf, graph_cutoffs=CUT, **cfg)
    synth.fill_state_dict_(model, 21)
    with torch.no_grad():               # the synthetic fill leaves the tiny xavier head: give the coordinate head some weight
        for n, p in model.named_parameters():
            if n.endswith('.4.weight'):
                p.mul_(20.0)
    t = torch.rand(len(n_rec), generator=torch.Generator().manual_seed(seed + 100)) * 0.9 + 0.05
    return g, model, t


def _oracle_grads(model, cfg, g, t, w_h, w_x):
    """Forward + torch autograd through the oracle in float64 on the fp32 graph (edges are built from the fp32 coordinates): the
    reference gradients carry no rounding noise of their own, so even heavily cancelling sums (a lone attention bias) are judged
    on their own scale."""
    ob = util.to_obatch(g)
    with torch.no_grad():
        edges = oegnn.lig_edges(ob, dict(cfg, graph_cutoffs=CUT))
    sd = {k: v.detach().double().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    ins = {k: v.detach().double().clone().requires_grad_(True) for k, v in (('lx', ob.x['lig']), ('lh', ob.h['lig']),
                                                                           ('kx', ob.x['kp']), ('kh', ob.h['kp']))}
    ob.x['lig'], ob.h['lig'], ob.x['kp'], ob.h['kp'] = ins['lx'], ins['lh'], ins['kx'], ins['kh']
    eh, ex = oegnn.egnn_dynamics_forward(sd, dict(cfg, graph_cutoffs=CUT), ob, t.double(), edges=edges)
    loss = (eh * w_h.double()).sum() + (ex * w_x.double()).sum()
    loss.backward()
    f = lambda v: None if v is None else v.float()
    pg64 = {k: v.grad for k, v in sd.items()}
    # the same in fp32: what the reference's own arithmetic loses on each gradient (the yardstick for heavily cancelling sums)
    sd32 = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    ob32 = util.to_obatch(g)
    e32, x32 = oegnn.egnn_dynamics_forward(sd32, dict(cfg, graph_cutoffs=CUT), ob32, t, edges=edges)
    ((e32 * w_h).sum() + (x32 * w_x).sum()).backward()
    noise = {k: None if pg64[k] is None else float((sd32[k].grad.double() - pg64[k]).abs().max()) for k in sd}
    return (eh.detach().float(), ex.detach().float(), {k: f(v) for k, v in pg64.items()}, {k: f(v.grad) for k, v in ins.items()},
            noise)


@pytest.mark.parametrize('name,cfg,rec_nf', [
    ('c2', util.EGNN_C2, 10),
    ('dev', dict(n_layers=2, hidden_nf=256, use_tanh=True, message_norm=0, update_kp_feat=False, norm=True, kl_k=5), 20),
    ('notanh', dict(n_layers=2, hidden_nf=256, use_tanh=False, message_norm=3.0, update_kp_feat=True, norm=False, kl_k=0,
                    ll_k=4), 10),
])
def test_gradients_match_oracle_autograd(name, cfg, rec_nf):
    cfg = dict(cfg)
    if name == 'c2':
        cfg['n_layers'] = 3
    g, model, t = _case(cfg, [60, 35, 48], [9, 14, 6], rec_nf=rec_nf)
    gen = torch.Generator().manual_seed(2)
    n_lig = g.num_nodes('lig')
    w_h, w_x = torch.randn(n_lig, 10, generator=gen), torch.randn(n_lig, 3, generator=gen)
    eh_ref, ex_ref, pg_ref, ig_ref, fp32_noise = _oracle_grads(model, cfg, g, t, w_h, w_x)

    model = model.cuda()
    gd = g.to('cuda')
    ins = {}
    for nt, kx, kh in (('lig', 'lx', 'lh'), ('kp', 'kx', 'kh')):
        for key, name_ in ((kx, 'x_0'), (kh, 'h_0')):
            ins[key] = gd.nodes[nt].data[name_].detach().clone().requires_grad_(True)
            gd.nodes[nt].data[name_] = ins[key]
    eh, ex = model(gd, t.cuda(), None)
    assert util.rel_err(eh.detach().cpu(), eh_ref) < 1e-4 and util.rel_err(ex.detach().cpu(), ex_ref) < 1e-4
    loss = (eh * w_h.cuda()).sum() + (ex * w_x.cuda()).sum()
    loss.backward()
    worst = []
    for n, p in model.named_parameters():
        ref = pg_ref[n]
        assert p.grad is not None, n
        if ref is None:                     # no path to the loss (keypoint-side weights of the last layer): exactly zero
            assert float(p.grad.abs().max()) == 0.0, n
            continue
        # against the float64 gradient, within TOL of the tensor's largest entry.  One kind of tensor cannot be held to that
        # by any fp32 implementation: a lone scalar (the attention bias) is the sum of one term per edge that cancels to three
        # digits (e.g. 3e-4 of its own size here, the reference's fp32 autograd 1e-4: `fp32_noise`), so it is judged on the
        # scale of its companion weight gradient, whose entries are the same per-edge terms weighted by activations
        scale = ref.abs().max().item()
        if ref.numel() == 1 and n.endswith('.bias'):
            scale = max(scale, pg_ref[n[:-4] + 'weight'].abs().max().item())
        err = (p.grad.cpu() - ref).abs().max().item() / max(scale, 1e-12)
        worst.append((err, n, fp32_noise[n] / max(scale, 1e-12)))
    worst.sort(reverse=True)
    assert worst[0][0] < TOL, worst[:8]
    for k, ref in ig_ref.items():
        got = ins[k].grad
        assert got is not None, k
        err = (got.cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
        assert err < TOL, (k, err)


def test_training_step_reduces_loss():
    """KeypointDiffusion.forward (ligand_diffuser.py:89-175) + an optimizer step: the train.py inner loop on the GPU."""
    from keypoint_diffusion_amd.ligand_diffuser import KeypointDiffusion
    cfg = dict(util.EGNN_C2, n_layers=2)
    model = KeypointDiffusion(10, 10, None, n_timesteps=100, architecture='egnn', rec_encoder_type='fixed',
                              graph_config=dict(n_keypoints=20, graph_cutoffs=CUT), dynamics_config=cfg,
                              rec_encoder_config={'vector_size': 16}, precision=1e-5)
    synth.fill_state_dict_(model, 3)
    model = model.cuda().train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    gs = synth.synth_complexes([50, 70], [10, 13], 20, CUT, seed=11)
    losses = []
    for it in range(6):
        torch.manual_seed(0)                 # same t and eps every iteration: the loss must go down
        g = G.batch([x.to('cuda') for x in synth.synth_complexes([50, 70], [10, 13], 20, CUT, seed=11)])
        out = model(g, None)
        assert set(out) == {'l2', 'pos', 'feat', 'rec_encoder'} and float(out['rec_encoder']) == 0.0
        opt.zero_grad()
        out['l2'].backward()
        torch.nn.utils.clip_grad_value_(model.parameters(), 1.0)       # train.py:539-543
        opt.step()
        losses.append(float(out['l2'].detach()))
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses


def test_trainer_degenerate_shapes_and_errors():
    """One-atom ligands (no ll edges), a batch that grows between calls, frozen parameters, call-order errors."""
    from keypoint_diffusion_amd import hip
    cfg = dict(util.EGNN_C2, n_layers=2)
    g, model, t = _case(cfg, [30, 22], [1, 1])
    model = model.cuda()
    for n, p in model.named_parameters():
        if 'lig_decoder' in n:
            p.requires_grad_(False)                      # frozen: no gradient buffer is bound for it
    eh, ex = model(g.to('cuda'), t.cuda(), None)
    (eh.sum() + ex.sum()).backward()
    assert all((p.grad is None) == ('lig_decoder' in n) for n, p in model.named_parameters())
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    # a larger batch afterwards re-reserves the workspace; gradients accumulate into .grad like autograd's
    g2, _, t2 = _case(cfg, [60, 35, 48, 20], [9, 14, 6, 3])
    before = model.egnn.conv_layers[0].edge_mlp['kl'][0].weight.grad.clone()
    eh, ex = model(g2.to('cuda'), t2.cuda(), None)
    (eh.sum() + ex.sum()).backward()
    assert not torch.equal(before, model.egnn.conv_layers[0].edge_mlp['kl'][0].weight.grad)
    # backward without a forward is a state error, not a crash
    tr, _ = model._trainer()
    with pytest.raises(hip.KpdError):
        tr.backward(torch.zeros(1, 10).cuda(), torch.zeros(1, 3).cuda(), None, None, None, None)
    # no_grad calls keep using the fused inference engine and agree with the training forward
    with torch.no_grad():
        eh_inf, ex_inf = model(g2.to('cuda'), t2.cuda(), None)
    assert util.rel_err(eh.detach().cpu(), eh_inf.cpu()) < 1e-4 and util.rel_err(ex.detach().cpu(), ex_inf.cpu()) < 1e-4


def test_stale_forward_cannot_be_differentiated():
    """The trainer holds the saved layer states of ONE forward.  A backward through an output whose states a later forward has
    overwritten must raise instead of returning the second forward's gradients; so must a backward after an in-place weight
    update (the C side reads the parameters in place)."""
    from keypoint_diffusion_amd import hip
    cfg = dict(util.EGNN_C2, n_layers=2)
    g, model, t = _case(cfg, [30, 22], [5, 7])
    model = model.cuda()
    eh1, ex1 = model(g.to('cuda'), t.cuda(), None)
    g2, _, t2 = _case(cfg, [28, 25], [6, 4], seed=9)
    eh2, ex2 = model(g2.to('cuda'), t2.cuda(), None)
    with pytest.raises(hip.KpdError, match='overwritten'):
        (eh1.sum() + ex1.sum()).backward()
    (eh2.sum() + ex2.sum()).backward()                     # the latest forward is still differentiable
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    eh3, ex3 = model(g.to('cuda'), t.cuda(), None)
    with torch.no_grad():
        model.lig_decoder[2].bias.add_(1.0)               # an optimizer step between forward and backward
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        (eh3.sum() + ex3.sum()).backward()


def test_training_step_is_bitwise_reproducible():
    """No float atomics on the training path: sums over the out-edges of a node are segmented sums over a source-grouped edge
    index, column sums are reduced in block order, split-K partial products are summed in slice order.  Two forward/backward passes of the same
    batch give bit-identical outputs and gradients (parameters and inputs)."""
    g, model, t = _case(dict(util.EGNN_C2, n_layers=3), [60, 35, 48], [9, 14, 6])
    model = model.cuda()
    runs = []
    for _ in range(2):
        gd = g.to('cuda')
        ins = []
        for nt in ('lig', 'kp'):
            for key in ('x_0', 'h_0'):
                v = gd.nodes[nt].data[key].detach().clone().requires_grad_(True)
                gd.nodes[nt].data[key] = v
                ins.append(v)
        model.zero_grad(set_to_none=True)
        eh, ex = model(gd, t.cuda(), None)
        (eh.square().sum() + ex.square().sum()).backward()
        runs.append(([eh.detach().clone(), ex.detach().clone()] + [v.grad.clone() for v in ins] +
                     [p.grad.clone() for p in model.parameters()]))
    names = ['eps_h', 'eps_x', 'd lig x', 'd lig h', 'd kp x', 'd kp h'] + [n for n, _ in model.named_parameters()]
    diff = [n for n, a, b in zip(names, *runs) if not torch.equal(a, b)]
    assert not diff, diff[:10]


def test_recompute_mode_gives_the_same_gradients():
    """KPD_TRAIN_STORE=0 (recompute the edge activations in the backward pass instead of keeping them: the low-memory mode) must
    be bit-identical to the default.  The switch is read once per process, so the other mode runs in a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import torch, sys; sys.path.insert(0, %r)\n'
            'from tests import test_egnn_train_gpu as T, util\n'
            'g, model, t = T._case(dict(util.EGNN_C2, n_layers=2), [40, 33], [7, 9])\n'
            'model = model.cuda(); eh, ex = model(g.to("cuda"), t.cuda(), None)\n'
            '(eh.square().sum() + ex.square().sum()).backward()\n'
            'torch.save([p.grad.cpu() for p in model.parameters()], sys.argv[1])\n' % root)
    outs = []
    for mode in ('1', '0'):
        path = os.path.join(root, 'gpurun_out', f'_grads_store{mode}.pt') if os.path.isdir(os.path.join(root, 'gpurun_out')) else f'/tmp/_grads_store{mode}.pt'
        subprocess.run([sys.executable, '-c', code, path], check=True, env=dict(os.environ, KPD_TRAIN_STORE=mode), timeout=600)
        outs.append(torch.load(path))
        os.remove(path)
    assert all(torch.equal(a, b) for a, b in zip(*outs))


def test_layer_edge_kernels_on_poisoned_workspaces():
    """The forward / backward edge passes of a layer run as one kernel each (k_egnn_edge_train, k_egnn_edge_bwd); their gradients are checked
    against the oracle's autograd above.  Here: a child process runs the same step on NaN-poisoned workspaces (KPD_POISON=1, read once per
    process) and must reproduce the default run's bits -- a read of memory this step has not written would surface as a NaN or as different
    bits -- on a batch whose edge counts leave ragged last tiles and a keypoint type that is updated."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import torch, sys; sys.path.insert(0, %r)\n'
            'from tests import test_egnn_train_gpu as T, util\n'
            'g, model, t = T._case(dict(util.EGNN_C2, n_layers=3), [40, 33, 57], [7, 9, 12])\n'
            'model = model.cuda(); eh, ex = model(g.to("cuda"), t.cuda(), None)\n'
            '(eh.square().sum() + ex.square().sum()).backward()\n'
            'torch.save([p.grad.cpu() for p in model.parameters()], sys.argv[1])\n' % root)
    outs = []
    for tag, env in (('default', {}), ('poison', {'KPD_POISON': '1'})):
        base = os.path.join(root, 'gpurun_out') if os.path.isdir(os.path.join(root, 'gpurun_out')) else '/tmp'
        path = os.path.join(base, f'_grads_{tag}.pt')
        subprocess.run([sys.executable, '-c', code, path], check=True, env=dict(os.environ, **env), timeout=600)
        outs.append(torch.load(path))
        os.remove(path)
    assert all(torch.isfinite(a).all() for a in outs[0])
    assert all(torch.equal(a, b) for a, b in zip(*outs))


def test_backward_after_the_caller_dropped_the_graph():
    """The training engines keep raw pointers into the prepared batch (per-complex offsets, the kk edge list) between forward
    and backward.  A caller that builds the graph inside a helper and keeps only the losses (train.py's step functions do) frees
    it before backward runs: the autograd node must keep the batch alive.  (Round 3 found the use-after-free with exactly this
    pattern: garbage gradients, once a GPU memory fault.)"""
    import gc
    g, model, t = _case(dict(util.EGNN_C2, n_layers=2), [40, 25, 31], [6, 9, 4])
    model = model.cuda()

    def forward_only():
        gd = g.to('cuda')
        eh, ex = model(gd, t.cuda(), None)
        return eh.square().sum() + ex.square().sum()          # the graph object goes out of scope here

    def grads(churn):
        model.zero_grad(set_to_none=True)
        loss = forward_only()
        if churn:                                              # recycle the freed blocks before backward reads them
            gc.collect()
            junk = [torch.full((n,), 12345, dtype=torch.int32, device='cuda') for n in (64, 257, 1024, 4096, 9000, 20000)]
            torch.cuda.synchronize()
            del junk
        loss.backward()
        return [p.grad.clone() for p in model.parameters()]

    a, b = grads(False), grads(True)
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_large_batches_take_the_weight_stationary_node_products():
    """From 4 096 receptor nodes on, the node MLPs' n x 257 x 257 products (forward and backward) run on the weight-stationary kernel with fused
    epilogues instead of the tiled GEMM (egnn_train.hip, node_ws).  The oracle is too slow at that size, so the check is additivity: a linear
    functional of the outputs over 16 complexes in ONE batch (4 960 receptor nodes) has the parameter gradients of the same complexes in two
    batches of 8 (2 480 nodes each: the tiled path) added up, and every complex keeps its outputs."""
    cfg = dict(util.EGNN_C2, n_layers=2)
    n_rec, n_lig = [310] * 16, [12 + (i % 5) for i in range(16)]
    gs = synth.synth_complexes(n_rec, n_lig, 20, CUT, seed=9, n_rec_feat=10)
    model = LigRecDynamics(10, 10, graph_cutoffs=CUT, **cfg)
    synth.fill_state_dict_(model, 21)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('.4.weight'):
                p.mul_(20.0)
    model = model.cuda()
    t = torch.rand(16, generator=torch.Generator().manual_seed(3)) * 0.9 + 0.05
    gen = torch.Generator().manual_seed(4)
    w_h = [torch.randn(n, 10, generator=gen) for n in n_lig]
    w_x = [torch.randn(n, 3, generator=gen) for n in n_lig]

    def run(idx):
        g = util.fixed_encode(G.batch([gs[i] for i in idx])).to('cuda')
        assert (g.num_nodes('kp') >= 4096) == (len(idx) == 16)
        model.zero_grad(set_to_none=True)
        eh, ex = model(g, t[idx].cuda(), None)
        wh, wx = torch.cat([w_h[i] for i in idx]).cuda(), torch.cat([w_x[i] for i in idx]).cuda()
        ((eh * wh).sum() + (ex * wx).sum()).backward()
        return eh.detach().cpu(), ex.detach().cpu(), {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}

    eh, ex, g_all = run(list(range(16)))
    eh_a, ex_a, g_a = run(list(range(8)))
    eh_b, ex_b, g_b = run(list(range(8, 16)))
    assert util.rel_err(eh, torch.cat([eh_a, eh_b])) < 2e-5 and util.rel_err(ex, torch.cat([ex_a, ex_b])) < 2e-5
    worst = {}
    for n in g_all:
        ref = g_a[n] + g_b[n]
        scale = float(ref.abs().max())
        if scale > 0:
            worst[n] = float((g_all[n] - ref).abs().max()) / scale
    bad = {n: e for n, e in worst.items() if e > TOL}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
