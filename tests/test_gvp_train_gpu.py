"""Backward pass of the GVP denoiser: gradients of every parameter and of the scalar /
vector input features from kpd_gvp_trainer_* against torch autograd through the CPU oracle."""
import pytest
import torch

from keypoint_diffusion_amd import graph as G, synth
from keypoint_diffusion_amd.dynamics_gvp import LigRecDynamicsGVP
from oracle import egnn as oegnn
from oracle import gvp as ogvp
from tests import util
from tests.golden.make_golden_cfgs import GVP_CFGS

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('gemm_mode')]   # both GEMM modes of the denoiser engines (conftest.py)
CUT = util.CUTOFFS_ALL_ATOM
TOL = 2e-4          # relative to the largest entry of each gradient tensor


def _case(cfg, n_rec, n_lig, n_kp_scalars):
    g = util.fixed_encode(util.make_batch(n_rec, n_lig, seed=31), n_vec=16)
    gen = torch.Generator().manual_seed(3)
    g.nodes['kp'].data['v_0'] = 0.5 * torch.randn(g.num_nodes('kp'), 16, 3, generator=gen)
    if n_kp_scalars != 10:
        g.nodes['kp'].data['h_0'] = torch.randn(g.num_nodes('kp'), n_kp_scalars, generator=gen)
    model = LigRecDynamicsGVP(10, n_kp_scalars, graph_cutoffs=CUT, **cfg)
    synth.fill_state_dict_(model, 7)
    model.eval()
    B = g.batch_size
    t = (torch.arange(B, dtype=torch.float32) + 1) / (B + 1)
    return g, model, t


def _oracle_grads(model, cfg, g, t, w_h, w_x):
    ob = util.to_obatch(g)
    sd = {k: v.detach().clone().requires_grad_(v.numel() > 0) for k, v in model.state_dict().items()}
    ins = {'lh': ob.h['lig'].clone().requires_grad_(True), 'kh': ob.h['kp'].clone().requires_grad_(True),
           'kv': ob.v['kp'].clone().requires_grad_(True), 'lx': ob.x['lig'].clone().requires_grad_(True),
           'kx': ob.x['kp'].clone().requires_grad_(True)}
    ob.h['lig'], ob.h['kp'], ob.v['kp'], ob.x['lig'], ob.x['kp'] = ins['lh'], ins['kh'], ins['kv'], ins['lx'], ins['kx']
    ocfg = dict(cfg, graph_cutoffs=CUT)
    with torch.no_grad():
        edges = oegnn.lig_edges(ob, ocfg)
    eh, ex = ogvp.gvp_dynamics_forward(sd, ocfg, ob, t, edges=edges)
    ((eh * w_h).sum() + (ex * w_x).sum()).backward()
    return eh.detach(), ex.detach(), {k: v.grad for k, v in sd.items()}, {k: v.grad for k, v in ins.items()}


@pytest.mark.parametrize('tag,over', [('gvp_kp', {}), ('gvp_mean', {}), ('gvp_norm0', {}), ('gvp_norm0', dict(ll_k=3, kl_k=0)),
                                      ('gvp_norm0', dict(n_hidden_scalars=100)), ('gvp_mean', dict(n_hidden_scalars=37)),
                                      # hidden width 256 = the register-chained message kernels, in every normalisation mode and chain length,
                                      # and once at a size with many tiles per edge type and several K slices per weight gradient
                                      ('gvp_mean', dict(n_hidden_scalars=256)), ('gvp_norm0', dict(n_hidden_scalars=256)),
                                      ('gvp_norm0', dict(n_hidden_scalars=256, ll_k=3, kl_k=0, n_message_gvps=1)),
                                      ('gvp_kp', dict(n_convs=2, sizes=([150, 97], [25, 18]))),
                                      # single-atom ligands: no ligand-ligand edges at all; a model that does not update the keypoints (one conv)
                                      ('gvp_kp', dict(n_convs=2, sizes=([30, 22], [1, 1]))),
                                      ('gvp_kp', dict(n_convs=1, update_kp=False))])
def test_gradients_match_oracle_autograd(tag, over):
    over = dict(over)
    n_rec, n_lig = over.pop('sizes', ([26, 19, 33], [7, 10, 5]))
    cfg = dict(GVP_CFGS[tag], dropout=0.0, **over)
    g, model, t = _case(cfg, n_rec, n_lig, 128 if tag == 'gvp_kp' else 10)
    gen = torch.Generator().manual_seed(2)
    n_lig = g.num_nodes('lig')
    w_h, w_x = torch.randn(n_lig, 10, generator=gen), torch.randn(n_lig, 3, generator=gen)
    eh_ref, ex_ref, pg_ref, ig_ref = _oracle_grads(model, cfg, g, t, w_h, w_x)

    model = model.cuda()
    gd = g.to('cuda')
    ins = {}
    for key, nt, name in (('lh', 'lig', 'h_0'), ('kh', 'kp', 'h_0'), ('kv', 'kp', 'v_0'), ('lx', 'lig', 'x_0'), ('kx', 'kp', 'x_0')):
        ins[key] = gd.nodes[nt].data[name].detach().clone().requires_grad_(True)
        gd.nodes[nt].data[name] = ins[key]
    eh, ex = model(gd, t.cuda(), None)
    # hidden width 256: the register-chained message kernels (forward, backward, batched weight gradients); else one GVP at a time
    assert model._trainer()[0].message_path() == (1 if cfg['n_hidden_scalars'] == 256 else 0)
    assert util.rel_err(eh.detach().cpu(), eh_ref) < 1e-4 and util.rel_err(ex.detach().cpu(), ex_ref) < 1e-4
    ((eh * w_h.cuda()).sum() + (ex * w_x.cuda()).sum()).backward()
    worst = []
    for n, p in model.named_parameters():
        if p.numel() == 0:
            continue
        ref = pg_ref[n]
        assert p.grad is not None, n
        if ref is None:
            assert float(p.grad.abs().max()) == 0.0, n
            continue
        scale = ref.abs().max().item()
        if ref.numel() == 1 and n.endswith('.bias'):          # lone scalar = cancelling sum over all rows: companion weight's scale
            scale = max(scale, pg_ref[n[:-4] + 'weight'].abs().max().item())
        err = (p.grad.cpu() - ref).abs().max().item() / max(scale, 1e-12)
        worst.append((err, n))
    worst.sort(reverse=True)
    assert worst[0][0] < TOL, worst[:8]
    for k, ref in ig_ref.items():
        err = (ins[k].grad.cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)
        assert err < TOL, (k, err)


def test_gvp_training_contract():
    """Train-mode dropout without autograd is refused, no_grad calls keep using the fused engine."""
    cfg = dict(GVP_CFGS['gvp_norm0'])
    g, model, t = _case(cfg, [20, 15], [6, 4], 10)
    model = model.cuda()
    model2 = LigRecDynamicsGVP(10, 10, graph_cutoffs=CUT, **dict(cfg, dropout=0.1)).cuda().train()
    with pytest.raises(NotImplementedError), torch.no_grad():
        model2(g.to('cuda'), t.cuda(), None)
    gd = g.to('cuda')
    eh, ex = model(gd, t.cuda(), None)
    with torch.no_grad():
        eh2, ex2 = model(g.to('cuda'), t.cuda(), None)
    assert util.rel_err(eh.detach().cpu(), eh2.cpu()) < 1e-4 and util.rel_err(ex.detach().cpu(), ex2.cpu()) < 1e-4


def test_dropout_training_step_matches_oracle_with_the_same_masks():
    """Training mode with GVPDropout 0.1 (every shipped GVP config): replay the step on the oracle with the masks the
    library drew (kpd_dropout_mask) and compare outputs and gradients; check the masks' statistics."""
    from keypoint_diffusion_amd import hip
    rate = 0.1
    cfg = dict(GVP_CFGS['gvp_kp'], dropout=rate)
    g, model, t = _case(cfg, [26, 19, 33], [7, 10, 5], 128)
    gen = torch.Generator().manual_seed(4)
    n_lig, n_kp, S = g.num_nodes('lig'), g.num_nodes('kp'), cfg['n_hidden_scalars']
    w_h, w_x = torch.randn(n_lig, 10, generator=gen), torch.randn(n_lig, 3, generator=gen)
    model_gpu = LigRecDynamicsGVP(10, 128, graph_cutoffs=CUT, **cfg)
    model_gpu.load_state_dict(model.state_dict())
    model_gpu = model_gpu.cuda().train()
    torch.manual_seed(123)
    eh, ex = model_gpu(g.to('cuda'), t.cuda(), None)
    ((eh * w_h.cuda()).sum() + (ex * w_x.cuda()).sum()).backward()
    seed = model_gpu.last_dropout_seed
    assert seed != 0
    masks, kept = {}, []
    for conv in range(cfg['n_convs']):
        for nti, (nt, n) in enumerate((('lig', n_lig), ('kp', n_kp))):
            for pos in (0, 1):
                ms = hip.dropout_mask(seed, conv, nti, pos, 0, n * S, rate).cpu().view(n, S)
                mv = hip.dropout_mask(seed, conv, nti, pos, 1, n * 16, rate).cpu().view(n, 16)
                masks[(conv, nt, pos)] = (ms, mv)
                kept.append(float((ms > 0).float().mean()))
                assert all(u == 0.0 or abs(u - 1 / (1 - rate)) < 1e-6 for u in ms.unique().tolist())
    assert abs(sum(kept) / len(kept) - (1 - rate)) < 0.01
    # oracle replay
    ob = util.to_obatch(g)
    sd = {k: v.detach().clone().requires_grad_(v.numel() > 0) for k, v in model.state_dict().items()}
    ocfg = dict(cfg, graph_cutoffs=CUT)
    with torch.no_grad():
        edges = oegnn.lig_edges(ob, ocfg)
    eh_ref, ex_ref = ogvp.gvp_dynamics_forward(sd, ocfg, ob, t, edges=edges, dropout_masks=masks)
    ((eh_ref * w_h).sum() + (ex_ref * w_x).sum()).backward()
    assert util.rel_err(eh.detach().cpu(), eh_ref.detach()) < 1e-4 and util.rel_err(ex.detach().cpu(), ex_ref.detach()) < 1e-4
    worst = []
    for n, p in model_gpu.named_parameters():
        ref = sd[n].grad if p.numel() else None
        if ref is None:
            continue
        scale = ref.abs().max().item()
        if ref.numel() == 1 and n.endswith('.bias'):
            scale = max(scale, sd[n[:-4] + 'weight'].grad.abs().max().item())
        worst.append(((p.grad.cpu() - ref).abs().max().item() / max(scale, 1e-12), n))
    worst.sort(reverse=True)
    assert worst[0][0] < TOL, worst[:6]
    # another step draws other masks; eval mode ignores dropout
    eh2, _ = model_gpu(g.to('cuda'), t.cuda(), None)
    assert model_gpu.last_dropout_seed != seed and not torch.allclose(eh2, eh)


def test_stale_forward_cannot_be_differentiated():
    """As for the EGNN trainer: one forward/backward pair at a time per module, parameters unchanged in between."""
    from keypoint_diffusion_amd import hip
    cfg = dict(GVP_CFGS['gvp_norm0'])
    g, model, t = _case(cfg, [20, 15], [6, 4], 10)
    model = model.cuda()
    eh1, ex1 = model(g.to('cuda'), t.cuda(), None)
    eh2, ex2 = model(g.to('cuda'), t.cuda(), None)
    with pytest.raises(hip.KpdError, match='overwritten'):
        (eh1.sum() + ex1.sum()).backward()
    (eh2.sum() + ex2.sum()).backward()
    eh3, ex3 = model(g.to('cuda'), t.cuda(), None)
    with torch.no_grad():
        next(p for p in model.parameters() if p.numel()).mul_(1.01)
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        (eh3.sum() + ex3.sum()).backward()


@pytest.mark.parametrize('tag,over,sizes', [('gvp_norm0', {}, ([40, 33], [9, 12])),
                                            # the chained kernels and the K-split batched weight gradients (several slices per product)
                                            ('gvp_kp', dict(dropout=0.0, n_convs=2), ([150, 97], [25, 18])),
                                            ('gvp_kp', dict(dropout=0.2, n_convs=2), ([60, 45], [12, 9]))])
def test_training_step_is_bitwise_reproducible(tag, over, sizes):
    """As for the EGNN trainer: two forward/backward passes of the same batch are bit-identical -- with dropout on too (the masks are a
    function of the seed, which the module draws from torch's generator: re-seeded before each pass)."""
    cfg = dict(GVP_CFGS[tag], **over)
    g, model, t = _case(cfg, sizes[0], sizes[1], 128 if tag == 'gvp_kp' else 10)
    model = model.cuda()
    if cfg['dropout'] > 0:
        model.train()
    assert model._trainer()[0].message_path() in (0, 1)
    runs = []
    for _ in range(2):
        gd = g.to('cuda')
        ins = []
        for nt, key in (('lig', 'h_0'), ('kp', 'h_0'), ('kp', 'v_0')):
            v = gd.nodes[nt].data[key].detach().clone().requires_grad_(True)
            gd.nodes[nt].data[key] = v
            ins.append(v)
        model.zero_grad(set_to_none=True)
        torch.manual_seed(123)
        eh, ex = model(gd, t.cuda(), None)
        (eh.square().sum() + ex.square().sum()).backward()
        runs.append([eh.detach().clone(), ex.detach().clone()] + [v.grad.clone() for v in ins] +
                    [p.grad.clone() for p in model.parameters() if p.grad is not None])
    assert all(torch.equal(a, b) for a, b in zip(*runs))


def test_chained_kernels_on_poisoned_workspaces():
    """The chained message / node kernels and the batched weight gradients keep their activations, message pieces and gradient staging in
    buffers of the trainer's own.  A child process runs a training step with all of them (and the LDS of every CU) NaN-filled first
    (KPD_POISON=1) and must reproduce the default run's bits: a read of memory this step has not written would surface as a NaN or as
    different bits.  Ragged last tiles, nodes without in-edges of a type (kl_k = 3: most ligand atoms of the larger complexes), dropout on."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import torch, sys; sys.path.insert(0, %r)\n'
            'from tests import test_gvp_train_gpu as T\n'
            'cfg = dict(T.GVP_CFGS["gvp_kp"], n_convs=2, dropout=0.1, kl_k=3)\n'
            'g, model, t = T._case(cfg, [70, 33, 57], [14, 9, 12], 128)\n'
            'model = model.cuda().train(); torch.manual_seed(5)\n'
            'eh, ex = model(g.to("cuda"), t.cuda(), None)\n'
            'assert model._trainer()[0].message_path() == 1\n'
            '(eh.square().sum() + ex.square().sum()).backward()\n'
            'torch.save([eh.detach().cpu(), ex.detach().cpu()] + [p.grad.cpu() for p in model.parameters() if p.grad is not None], sys.argv[1])\n' % root)
    outs = []
    for tag, env in (('default', {}), ('poison', {'KPD_POISON': '1'})):
        base = os.path.join(root, 'gpurun_out') if os.path.isdir(os.path.join(root, 'gpurun_out')) else '/tmp'
        path = os.path.join(base, f'_gvp_grads_{tag}.pt')
        subprocess.run([sys.executable, '-c', code, path], check=True, env=dict(os.environ, **env), timeout=600)
        outs.append(torch.load(path))
        os.remove(path)
    assert all(torch.isfinite(a).all() for a in outs[0])
    assert all(torch.equal(a, b) for a, b in zip(*outs))


def test_chained_path_survives_a_growing_reservation():
    """A larger batch after a smaller one makes the trainer reserve again (activation slots, packs, gradient staging are re-carved and the weight
    pack's descriptor table rebuilt): the step on the larger batch must equal the same step on a fresh model, bit for bit."""
    cfg = dict(GVP_CFGS['gvp_kp'], n_convs=2, dropout=0.0)
    g_small, model, t_small = _case(cfg, [20, 15], [5, 4], 128)
    g_big, fresh, t_big = _case(cfg, [90, 70, 55], [14, 9, 11], 128)
    model, fresh = model.cuda(), fresh.cuda()

    def step(m, g, t):
        m.zero_grad(set_to_none=True)
        eh, ex = m(g.to('cuda'), t.cuda(), None)
        (eh.square().sum() + ex.square().sum()).backward()
        return [eh.detach().clone(), ex.detach().clone()] + [p.grad.clone() for p in m.parameters() if p.grad is not None]

    step(model, g_small, t_small)
    grown = step(model, g_big, t_big)
    assert model._trainer()[0].message_path() == 1
    direct = step(fresh, g_big, t_big)
    assert all(torch.equal(a, b) for a, b in zip(grown, direct))
