"""GPU tests of the diffusion wrapper: one reverse step against the oracle (noise injected), and the
public sampling entry points end to end."""
import pytest
import torch

from keypoint_diffusion_amd import graph as G
from keypoint_diffusion_amd import synth
from keypoint_diffusion_amd.ligand_diffuser import KeypointDiffusion, LigandDiffuser
from oracle import diffusion as odiff
from oracle import egnn as oegnn
from oracle import gvp as ogvp

from . import util
from .test_gvp_gpu import GVP_ALL_ATOM

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('gemm_mode')]   # both GEMM modes of the EGNN edge kernel (conftest.py)
CUT = util.CUTOFFS_ALL_ATOM


def _model(arch, T=50):
    dyn = util.EGNN_C2 if arch == 'egnn' else dict(GVP_ALL_ATOM, n_convs=3)
    m = KeypointDiffusion(10, 10, None, n_timesteps=T, architecture=arch, rec_encoder_type='fixed',
                          graph_config=dict(n_keypoints=20, graph_cutoffs=CUT), dynamics_config=dyn,
                          rec_encoder_config={'vector_size': 16}, precision=1e-5)
    synth.fill_state_dict_(m, 13)
    return m.eval()


@pytest.mark.parametrize('arch', ['egnn', 'gvp'])
def test_reverse_step_matches_oracle(cuda, arch):
    assert LigandDiffuser is KeypointDiffusion
    T = 50
    model = _model(arch, T)
    gs = synth.synth_complexes([90, 140], [11, 17], 20, CUT, seed=3)
    g = model.encode_receptors(G.batch(gs))
    ob = util.to_obatch(g)
    s = torch.tensor([20.0, 31.0]) / T
    t = s + 1.0 / T
    gen = torch.Generator().manual_seed(1)
    nx, nh = torch.randn(ob.x['lig'].shape, generator=gen), torch.randn(ob.h['lig'].shape, generator=gen)
    sd = {k[len('dynamics.'):]: v.clone() for k, v in model.state_dict().items() if k.startswith('dynamics.')}
    if arch == 'egnn':
        eh, ex = oegnn.egnn_dynamics_forward(sd, dict(util.EGNN_C2, graph_cutoffs=CUT), ob, t)
    else:
        eh, ex = ogvp.gvp_dynamics_forward(sd, dict(GVP_ALL_ATOM, n_convs=3, graph_cutoffs=CUT), ob, t)
    ref = odiff.sample_step(ob.clone(), eh, ex, s, t, odiff.gamma_table(T, 1e-5), T, nx, nh)
    model = model.to(cuda)
    gd = g.to(cuda)
    with torch.no_grad():
        model.sample_p_zs_given_zt(s.to(cuda), t.to(cuda), gd, G.get_batch_idxs(gd), noise=(nx.to(cuda), nh.to(cuda)))
    torch.cuda.synchronize()
    assert util.rel_err(gd.nodes['lig'].data['x_0'], ref.x['lig']) < 1e-4
    assert util.rel_err(gd.nodes['lig'].data['h_0'], ref.h['lig']) < 1e-4
    assert util.rel_err(gd.nodes['kp'].data['x_0'], ref.x['kp']) < 1e-5
    # ligand COM is zero after the step (remove_com, ligand_diffuser.py:536)
    com = G.readout_nodes(gd, 'x_0', op='mean', ntype='lig')
    assert float(com.abs().max()) < 1e-5


@pytest.mark.parametrize('T,precision', [(500, 1e-5), (50, 1e-4), (1000, 1e-5)])
def test_step_coefficients_all_timesteps(cuda, T, precision):
    """kpd_step_coefficients against the oracle's schedule arithmetic for every (s, t) = (i / T, (i + 1) / T)."""
    from keypoint_diffusion_amd import hip
    table = odiff.gamma_table(T, precision)
    s = torch.arange(T, dtype=torch.float32) / T
    t = (torch.arange(T, dtype=torch.float32) + 1) / T
    g_s, g_t = odiff.gamma_at(table, s, T), odiff.gamma_at(table, t, T)
    s2, s_ts, a_ts = odiff.sigma_and_alpha_t_given_s(g_t, g_s)
    ref = torch.stack([a_ts, s2 / a_ts / odiff.sigma(g_t), s_ts * odiff.sigma(g_s) / odiff.sigma(g_t)], dim=1)
    got = hip.step_coefficients(table.float().to(cuda), s.to(cuda), t.to(cuda)).cpu()
    assert got.shape == (T, 3)
    assert float(((got - ref.float()).abs() / ref.float().abs().clamp_min(1e-6)).max()) < 1e-4
    # the module-level mirror takes the same path on the GPU
    m = _model('egnn', T).to(cuda)
    assert torch.equal(m.step_coefficients(s.to(cuda), t.to(cuda)).cpu(), hip.step_coefficients(m.gamma.gamma, s.to(cuda), t.to(cuda)).cpu())


def test_sample_given_pocket_end_to_end(cuda):
    model = _model('egnn', T=8).to(cuda)
    pocket = synth.synth_complexes([70], [1], 20, CUT, seed=9)[0].to(cuda)
    pocket.remove_nodes(pocket.nodes('lig'), ntype='lig')
    pos, feat = model.sample_given_pocket(pocket, torch.tensor([6, 9, 9]), diff_batch_size=2)
    assert [p.shape for p in pos] == [(6, 3), (9, 3), (9, 3)]
    assert [f.shape for f in feat] == [(6, 10), (9, 10), (9, 10)]
    assert all(torch.isfinite(p).all() and p.device.type == 'cpu' for p in pos)
    # visualize=True returns one trajectory per ligand with T + 1 frames
    fx, fh = model.sample_given_pocket(pocket, torch.tensor([5]), visualize=True)
    assert len(fx) == 1 and len(fx[0]) == 9 and fx[0][0].shape == (5, 3)


def test_training_scope_contract(cuda):
    """Fixed- and learned-encoder models train; under no_grad the same forward evaluates the four losses."""
    model = _model('gvp').to(cuda).train()
    g = G.batch(synth.synth_complexes([30, 22], [5, 7], 20, CUT)).to(cuda)
    torch.manual_seed(0)
    out = model(g, None)
    assert set(out) == {'l2', 'pos', 'feat', 'rec_encoder'}
    out['l2'].backward()
    bad = [n for n, p in model.dynamics.named_parameters() if p.numel() and (p.grad is None or not torch.isfinite(p.grad).all())]
    assert not bad, bad[:8]
    assert any(float(p.grad.abs().max()) > 0 for p in model.dynamics.parameters() if p.numel())
    learned = KeypointDiffusion(10, 128, None, n_timesteps=50, architecture='egnn', rec_encoder_type='learned',
                                graph_config=dict(n_keypoints=8, graph_cutoffs=CUT), dynamics_config=dict(util.EGNN_C2, n_layers=1),
                                rec_encoder_config=dict(n_convs=1, in_n_node_feat=10, hidden_n_node_feat=32, out_n_node_feat=128,
                                                        use_tanh=True, coords_range=10, kp_feat_scale=1.0, message_norm=0.0,
                                                        use_sameres_feat=False, k_closest=3, kp_rad=0.0, norm=True, fix_pos=False,
                                                        n_kk_convs=0), precision=1e-5)
    # evaluation (train.py's test_model runs the model under no_grad): all four losses, the encoder loss being the optimal-
    # transport distance between the learned keypoints and the receptor atoms (losses/rec_encoder_loss.py:49-69)
    from oracle import rec_encoder_loss as oloss
    synth.fill_state_dict_(learned, 3)
    learned = learned.to(cuda).eval()
    g2 = G.batch(synth.synth_complexes([30, 22], [5, 7], 8, CUT, seed=12)).to(cuda)
    # gradients enabled: since round 3 the learned encoders have backward passes too (tests/test_recegnn_train_gpu.py,
    # test_recenc_train_gpu.py hold the parity checks); here only that the training entry point reaches them
    tr = learned(G.batch(synth.synth_complexes([30, 22], [5, 7], 8, CUT, seed=12)).to(cuda), None)
    (tr['l2'] + tr['rec_encoder']).backward()
    enc_grads = [p.grad for n, p in learned.rec_encoder.named_parameters() if p.grad is not None]
    assert len(enc_grads) > 10 and all(torch.isfinite(gr).all() for gr in enc_grads) and any(float(gr.abs().max()) > 0 for gr in enc_grads)
    learned.zero_grad(set_to_none=True)
    with torch.no_grad():
        enc = learned.encode_receptors(G.batch(synth.synth_complexes([30, 22], [5, 7], 8, CUT, seed=12)).to(cuda))
        out = learned(g2, None)
    assert set(out) == {'l2', 'pos', 'feat', 'rec_encoder'} and all(torch.isfinite(v).all() for v in out.values())
    units = G.unbatch(enc)
    want = oloss.ot_loss([u.nodes['kp'].data['x_0'].cpu() for u in units], [u.nodes['rec'].data['x_0'].cpu() for u in units])
    assert want > 0 and abs(float(out['rec_encoder']) - want) < 1e-4 * want


def test_complex_noise_is_sharding_invariant(cuda):
    """kpd_complex_noise: a complex draws the same values whatever batch it sits in; N(0,1) statistics."""
    from keypoint_diffusion_amd import hip
    gs = synth.synth_complexes([60, 90, 40, 75], [11, 17, 5, 23], 20, CUT, seed=3)
    ids = torch.tensor([100, 101, 102, 103], device=cuda)
    full = G.batch(gs).to(cuda)
    part = G.batch(gs[2:]).to(cuda)
    for width in (3, 10):
        a = hip.complex_noise(full.prepared(), width, ids, 1234, 17, 1)
        b = hip.complex_noise(part.prepared(), width, ids[2:], 1234, 17, 1)
        assert torch.equal(a[11 + 17:], b)                                       # bitwise, independent of batch composition
        assert not torch.equal(a, hip.complex_noise(full.prepared(), width, ids, 1234, 18, 1))       # step, tag, seed matter
        assert not torch.equal(a, hip.complex_noise(full.prepared(), width, ids, 1234, 17, 0))
        assert not torch.equal(a, hip.complex_noise(full.prepared(), width, ids, 1235, 17, 1))
    big = G.batch(synth.synth_complexes([30] * 64, [60] * 64, 20, CUT, seed=5)).to(cuda)
    z = hip.complex_noise(big.prepared(), 10, torch.arange(64, device=cuda), 7, 0, 0)
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02
    assert abs(float((z ** 4).mean()) - 3.0) < 0.2                               # kurtosis of a normal


def test_sharded_sampling_reproduces_single_process(cuda):
    """Reverse loop on [c0..c3] together vs on two shards with the same complex ids: same ligands up to fp32 summation order."""
    model = _model('egnn', T=8).to(cuda).use_complex_noise(99)
    gs = synth.synth_complexes([90, 140, 60, 110], [11, 17, 9, 14], 20, CUT, seed=3)
    ids = torch.tensor([40, 41, 42, 43])
    with torch.no_grad():
        full = model.encode_receptors(G.batch(gs).to(cuda))
        px, ph = model.sample_from_encoded_receptors(full, complex_ids=ids)
        out = []
        for lo, hi in ((0, 2), (2, 4)):
            gsh = model.encode_receptors(G.batch(synth.synth_complexes([90, 140, 60, 110], [11, 17, 9, 14], 20, CUT, seed=3)[lo:hi]).to(cuda))
            x, h = model.sample_from_encoded_receptors(gsh, complex_ids=ids[lo:hi])
            out += list(zip(x, h))
    for (x, h), fx, fh in zip(out, px, ph):
        assert util.rel_err(x, fx) < 1e-3 and util.rel_err(h, fh) < 1e-3


@pytest.mark.parametrize('arch', ['egnn', 'gvp'])
def test_step_graph_replays_the_eager_step(cuda, arch):
    """A captured reverse step (HIP graph) reproduces the eager step bit for bit for the same injected noise, for
    several timesteps from one capture; the sampling loop can run on it."""
    T = 20
    model = _model(arch, T).to(cuda)
    gs = synth.synth_complexes([60, 45], [9, 13], 20, CUT, seed=5)
    g1 = model.encode_receptors(G.batch(gs)).to(cuda)
    g2 = model.encode_receptors(G.batch(synth.synth_complexes([60, 45], [9, 13], 20, CUT, seed=5))).to(cuda)
    gen = torch.Generator().manual_seed(1)
    nx = torch.randn(g1.num_nodes('lig'), 3, generator=gen).to(cuda)
    nh = torch.randn(g1.num_nodes('lig'), 10, generator=gen).to(cuda)
    with torch.no_grad():
        sg = model.capture_step(g1, noise=(nx, nh))
        ones = torch.ones(2, device=cuda)
        for s in (19, 18, 7):
            sg.step(s / T, (s + 1) / T)
            model.sample_p_zs_given_zt(ones * (s / T), ones * ((s + 1) / T), g2, noise=(nx, nh))
            for nt, k in (('lig', 'x_0'), ('lig', 'h_0'), ('kp', 'x_0')):
                assert torch.equal(g1.nodes[nt].data[k], g2.nodes[nt].data[k]), (s, nt, k)
    pocket = synth.synth_complexes([70], [1], 20, CUT, seed=9)[0].to(cuda)
    pocket.remove_nodes(pocket.nodes('lig'), ntype='lig')
    enc = G.unbatch(model.encode_receptors(G.batch([pocket])))[0]
    bg = G.batch(G.copy_graph(enc, n_copies=2, lig_atoms_per_copy=torch.tensor([6, 9])))
    pos, feat = model.sample_from_encoded_receptors(bg, use_graph=True)
    assert [p.shape for p in pos] == [(6, 3), (9, 3)] and all(torch.isfinite(p).all() for p in pos)


def _assert_samples_equal(got, ref, tol=1e-3):
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        assert [tuple(p.shape) for p in a['positions']] == [tuple(p.shape) for p in b['positions']]
        for x, fx in zip(a['positions'], b['positions']):
            assert util.rel_err(x, fx) < tol
        for h, fh in zip(a['features'], b['features']):
            assert util.rel_err(h, fh) < tol


@pytest.mark.parametrize('world', [2, 3])
def test_sample_sharded_process_ranks_equal_single_process(cuda, tmp_path, world, gemm_mode):
    """`KeypointDiffusion._sample` under a process group of `world` rank PROCESSES (all on cuda:0, gloo): every rank returns ALL
    ligands in input order, equal to the single-process run of the same noise seed up to fp32 summation order (SURVEY.md 8(e),
    models/ligand_diffuser.py:292-324).  Five is the most a GPU box admits next to the test runner (six GPU-holding processes);
    the eight-rank job of configs[3] / configs[4] is rehearsed with thread ranks below.  (Sharding does not depend on the GEMM mode
    and every fresh rank process costs ~15 s of start-up on a GPU box: three process ranks in the f32 pass, two in the f16x2 pass; more ranks run as threads below.)"""
    import os
    import socket
    import subprocess
    import sys
    from . import sharded_worker as W
    if gemm_mode == 'f16x2' and world > 2:
        pytest.skip('process-rank sharding at world 3 is covered in the f32 pass')
    model = W.build_model(cuda).use_complex_noise(W.SEED)
    ref = model._sample(W.pockets(cuda), W.N_LIG, rec_enc_batch_size=2, diff_batch_size=2)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'shard')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, '-m', 'tests.sharded_worker', out], cwd=root, env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    for r in range(world):
        _assert_samples_equal(torch.load(f'{out}.{r}')['samples'], ref)


def test_sample_sharded_world8_thread_ranks_equal_single_process(cuda):
    """The eight-rank job (configs[3] / configs[4] run 8 ranks) on ONE card: eight thread ranks, each with its own model and
    engines on cuda:0, run `_sample` under torch's in-process process group; every rank returns every ligand in input order,
    equal to the single-process run.  (8 complexes of the 11 go one per rank, three ranks take two.)"""
    from . import sharded_worker as W
    model = W.build_model(cuda).use_complex_noise(W.SEED)
    ref = model._sample(W.pockets(cuda), W.N_LIG, rec_enc_batch_size=2, diff_batch_size=2)
    for got in util.run_threaded_world(8, lambda rank: W.run_rank(cuda)):
        _assert_samples_equal(got, ref)


def test_step_graph_refuses_to_replay_after_the_engine_changed(cuda):
    """A captured step holds raw pointers into the engine's arena and packed weights: a re-reserved workspace (larger batch) or
    changed weights must turn `step()` into an error, never into a replay on freed or stale memory."""
    from keypoint_diffusion_amd import hip
    T = 20
    model = _model('egnn', T).to(cuda)
    small = model.encode_receptors(G.batch(synth.synth_complexes([40], [6], 20, CUT, seed=5))).to(cuda)
    with torch.no_grad():
        sg = model.capture_step(small)
        sg.step(18 / T, 19 / T)                                         # fine
        big = model.encode_receptors(G.batch(synth.synth_complexes([90, 120, 70], [14, 9, 20], 20, CUT, seed=6))).to(cuda)
        model.dynamics(big, torch.tensor([0.5, 0.5, 0.5], device=cuda), None)      # grows the workspace: arena reallocated
        with pytest.raises(hip.KpdError, match='stale step graph'):
            sg.step(17 / T, 18 / T)
        sg2 = model.capture_step(small)
        sg2.step(17 / T, 18 / T)
        model.dynamics.lig_decoder[2].bias.add_(0.5)                    # weights changed in place
        with pytest.raises(hip.KpdError, match='weights changed'):
            for _ in range(16):                                         # one tensor somewhere: re-validated every 16th replay at the latest
                sg2.step(16 / T, 17 / T)
        sg3 = model.capture_step(small)
        sg3.step(17 / T, 18 / T)
        for p in model.parameters():                                    # what an optimizer step / load_state_dict / an EMA swap does: every tensor
            p.mul_(1.0)
        with pytest.raises(hip.KpdError, match='weights changed'):     # ... raises on the very next replay (sentinel parameters)
            sg3.step(16 / T, 17 / T)
    torch.cuda.synchronize()
