#!/usr/bin/env python
"""Headline benchmark: denoising steps/sec of the reverse-diffusion hot path on MI355X.

A *step* is one reverse-diffusion step (`KeypointDiffusion.sample_p_zs_given_zt`:
denoiser forward + z_s update + COM removal) over one batch of B synthetic complexes that
is already resident in HBM.  Workload = BASELINE.json configs[1]:
egnn_all_atom dynamics, B = 64 synthetic 300-atom pockets / 25-atom ligands, T = 500.

    python bench.py --gpus N --steps K --warmup W
N > 1: launched by torch.distributed.run, one rank per GPU; every rank owns its own batch of B
complexes (weak scaling, no data-path collective), one RCCL all-gather of the ligand tensors
at the end of the timed region (the exchange the sampler performs after the last step).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from keypoint_diffusion_amd import graph as G            # noqa: E402
from keypoint_diffusion_amd import synth                 # noqa: E402
from keypoint_diffusion_amd.ligand_diffuser import KeypointDiffusion   # noqa: E402

CUTOFFS = {'kk': 8, 'kl': 6, 'll': 6, 'rk': 100, 'rr': 3.5}      # trained_models/egnn_all_atom/config.yml
DYNAMICS = dict(hidden_nf=256, kl_k=5, ll_k=0, message_norm=0, n_layers=6, no_cg=False, norm=True,
                update_kp_feat=True, use_tanh=True)
N_TIMESTEPS = 500
PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0
# FLOPs the fused edge kernel is responsible for, per edge per layer: the two 257x257 second
# Linears of edge_mlp / coord_mlp plus the attention and coordinate heads
EDGE_KERNEL_FLOP_PER_EDGE = 2 * (2 * 257 * 257) + 2 * (2 * 257)
# reference formulation of the same edge work: both Linears of both MLPs
EDGE_ALGO_FLOP_PER_EDGE = 2 * (515 * 257 + 257 * 257) * 2 + 4 * 257
EDGE_ALGO_BYTES_PER_EDGE = 4 * 257 + 16          # gathered row + coords + index


# secondary workloads (BASELINE.json configs[2] and configs[4]); the JSON contract line is always configs[1]
GVP_DYN = dict(vector_size=16, n_convs=6, n_hidden_scalars=256, message_norm=10.0, update_kp=True, ll_k=0, kl_k=7,
               n_message_gvps=3, n_update_gvps=2, n_noise_gvps=4, dropout=0.1)
GVP_ENC = dict(out_scalar_size=128, n_message_gvps=3, n_update_gvps=2, vector_size=16, n_rr_convs=4, n_rk_convs=2,
               message_norm=10.0, k_closest=5, kp_rad=0, dropout=0.1)
EGNN_ENC = dict(coords_range=10, fix_pos=False, hidden_n_node_feat=128, k_closest=5, kp_feat_scale=1.0, kp_rad=0.0,
                message_norm=0.0, n_convs=4, n_kk_convs=0, n_kk_heads=4, no_cg=False, norm=True, out_n_node_feat=128,
                use_sameres_feat=True, use_tanh=True)                      # trained_models/egnn_40kp/config.yml:59-75
WORKLOADS = {
    'egnn_all_atom': dict(arch='egnn', enc='fixed', dyn=DYNAMICS, n_kp=20, cutoffs=CUTOFFS),
    'egnn_40kp': dict(arch='egnn', enc='learned', dyn=dict(DYNAMICS, message_norm=0.0), n_kp=40,
                      cutoffs=dict(CUTOFFS, kl=8, ll=5)),
    'gvp_40kp': dict(arch='gvp', enc='learned', dyn=GVP_DYN, n_kp=40, cutoffs=dict(CUTOFFS, kl=8, ll=6.0)),
    'gvp_all_atom': dict(arch='gvp', enc='fixed', dyn=dict(GVP_DYN, message_norm='mean'), n_kp=20, cutoffs=CUTOFFS),
    # one optimisation step of train.py's inner loop (loss of KeypointDiffusion.forward, backward, clip, Adam) on the
    # egnn_all_atom model: the backward pass of csrc/egnn_train.hip
    'egnn_train': dict(arch='egnn', enc='fixed', dyn=DYNAMICS, n_kp=20, cutoffs=CUTOFFS),
    # same for the gvp_all_atom model in training mode (GVPDropout 0.1): the backward pass of csrc/gvp_train.hip
    'gvp_train': dict(arch='gvp', enc='fixed', dyn=dict(GVP_DYN, message_norm='mean'), n_kp=20, cutoffs=CUTOFFS),
}


def build_model(device, workload='egnn_all_atom'):
    w = WORKLOADS[workload]
    rec_cfg = dict(GVP_ENC) if w['arch'] == 'gvp' else (dict(EGNN_ENC) if w['enc'] == 'learned' else {})
    if w['enc'] == 'learned':
        rec_cfg['in_scalar_size' if w['arch'] == 'gvp' else 'in_n_node_feat'] = 10
    model = KeypointDiffusion(10, 128 if w['enc'] == 'learned' else 10, None, n_timesteps=N_TIMESTEPS,
                              architecture=w['arch'], rec_encoder_type=w['enc'],
                              graph_config=dict(n_keypoints=w['n_kp'], graph_cutoffs=w['cutoffs']),
                              dynamics_config=w['dyn'], rec_encoder_config=rec_cfg, precision=1e-5)
    synth.fill_state_dict_(model, seed=0)
    return model.eval().to(device)


def build_batch(model, B, n_rec, n_lig, seed, device, workload='egnn_all_atom'):
    w = WORKLOADS[workload]
    if isinstance(n_rec, int):
        n_rec, n_lig = [n_rec] * B, [n_lig] * B
    gs = synth.synth_complexes(n_rec, n_lig, w['n_kp'], w['cutoffs'], seed=seed)
    g = G.batch(gs)
    if w['enc'] == 'learned':
        if w['arch'] == 'egnn':                      # synthetic rr `same_res` column (the dataset's bool edge feature)
            s_, d_ = g.edges(etype='rr')
            g.edges['rr'].data['same_res'] = ((s_ // 8) == (d_ // 8)).view(-1, 1)
        g = g.to(device)                             # the learned encoders run on the GPU (once per pocket)
    with torch.no_grad():
        g = model.encode_receptors(g)                # fixed encoder: kp := rec, kk := rr
    return g.to(device)


def cpu_baseline(B_sample=4, steps=2):
    """Oracle (CPU restatement of the reference path) on this box's host cores, same shape."""
    from oracle import diffusion as odiff
    from oracle import egnn as oegnn
    from tests.util import to_obatch
    model = build_model('cpu')
    g = build_batch(model, B_sample, 300, 25, seed=99, device='cpu')
    ob = to_obatch(g)
    sd = {k[len('dynamics.'):]: v for k, v in model.state_dict().items() if k.startswith('dynamics.')}
    cfg = dict(DYNAMICS, graph_cutoffs=CUTOFFS)
    table = odiff.gamma_table(N_TIMESTEPS, 1e-5)
    gen = torch.Generator().manual_seed(5)

    def one(si):
        s = torch.full((B_sample,), si / N_TIMESTEPS)
        t = torch.full((B_sample,), (si + 1) / N_TIMESTEPS)
        eh, ex = oegnn.egnn_dynamics_forward(sd, cfg, ob, t)
        odiff.sample_step(ob, eh, ex, s, t, table, N_TIMESTEPS, torch.randn(ob.x['lig'].shape, generator=gen),
                          torch.randn(ob.h['lig'].shape, generator=gen))

    with torch.no_grad():
        one(N_TIMESTEPS - 1)
        t0 = time.perf_counter()
        for i in range(steps):
            one(N_TIMESTEPS - 2 - i)
        dt = time.perf_counter() - t0
    complex_steps_per_s = B_sample * steps / dt
    return {'value': complex_steps_per_s / 64.0, 'unit': 'steps/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'complex_steps_per_s': complex_steps_per_s,
            'sample': f'oracle (plain PyTorch fp32 CPU restatement) on {B_sample} complexes of the same 300/25 shape x '
                      f'{steps} reverse steps after 1 warm-up, scaled to the B=64 batch'}


def train_cpu_baseline(workload, B_sample=2):
    """Loss + torch autograd through the CPU oracle for B_sample complexes of the same shape (one step after a warm-up)."""
    from oracle import egnn as oegnn
    from oracle import gvp as ogvp
    from tests.util import to_obatch
    w = WORKLOADS[workload]
    model = build_model('cpu', workload)
    g = build_batch(model, B_sample, 300, 25, seed=99, device='cpu', workload=workload)
    ob = to_obatch(g)
    sd = {k[len('dynamics.'):]: v.clone().requires_grad_(v.numel() > 0) for k, v in model.state_dict().items()
          if k.startswith('dynamics.')}
    cfg = dict(w['dyn'], graph_cutoffs=w['cutoffs'])
    fwd = oegnn.egnn_dynamics_forward if w['arch'] == 'egnn' else ogvp.gvp_dynamics_forward

    def one():
        eh, ex = fwd(sd, cfg, ob, torch.full((B_sample,), 0.5))
        (eh.square().sum() + ex.square().sum()).backward()

    one()
    t0 = time.perf_counter()
    one()
    dt = time.perf_counter() - t0
    return {'value': B_sample / dt / 64.0, 'unit': 'steps/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'complex_steps_per_s': B_sample / dt,
            'sample': f'oracle forward + torch autograd backward on {B_sample} complexes of the same 300/25 shape, 1 step after '
                      f'1 warm-up, scaled to the B=64 batch (no optimizer step on the CPU side)'}


def run_train(args, device, rank, world, dist):
    """Secondary workloads: training steps/sec of the EGNN / GVP denoiser (fixed receptor encoder) on synthetic complexes."""
    from keypoint_diffusion_amd.dist import allreduce_gradients
    w = WORKLOADS[args.workload]
    model = build_model(device, args.workload).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    B = args.batch
    gs = synth.synth_complexes([args.n_rec] * B, [args.n_lig] * B, w['n_kp'], w['cutoffs'], seed=1234 + rank * B)
    template = G.batch(gs).to(device)

    def step():
        g = template.to(device)             # fresh container over the same device tensors (forward re-binds node data)
        losses = model(g, None)
        opt.zero_grad(set_to_none=True)
        losses['l2'].backward()
        if dist is not None:
            allreduce_gradients(list(model.parameters()))
        torch.nn.utils.clip_grad_value_(model.parameters(), 1.0)
        opt.step()
        return losses['l2']

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if rank == 0:
        out = {'metric': 'training steps/sec', 'value': world * args.steps / elapsed, 'unit': 'steps/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': f'{args.workload}: loss + backward + clip + Adam on {w["arch"]}_all_atom (6 layers, hidden 256, '
                                      f'training mode), batch of {B} '
                                      f'synthetic {args.n_rec}-atom pockets / {args.n_lig}-atom ligands per GPU, one bucketed gradient all-reduce per step when N > 1',
                          'batch_per_gpu': B},
               'complex_steps_per_s': world * args.steps / elapsed * B, 'final_l2': float(last.detach())}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = train_cpu_baseline(args.workload)
        print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--n-rec', type=int, default=300)
    ap.add_argument('--n-lig', type=int, default=25)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', default='egnn_all_atom', choices=list(WORKLOADS),
                    help='egnn_all_atom = BASELINE.json configs[1] (the contract line); the others are secondary')
    ap.add_argument('--graph', action='store_true', help='replay the reverse step as a captured HIP graph (StepGraph)')
    ap.add_argument('--ragged', action='store_true', help='pockets 150-600 atoms, ligands 15-35 atoms (configs[4] shape)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the hot path has no CPU implementation')
    # KPD_BENCH_SHARE_GPU=1 (functional rehearsal on a one-GPU box only): every rank uses cuda:0 and the
    # process group runs over gloo, because RCCL refuses two ranks on one device
    share = os.environ.get('KPD_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)
    if args.gpus != world:
        print(f'[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}', file=sys.stderr)

    torch.manual_seed(1000 + rank)
    if args.workload in ('egnn_train', 'gvp_train'):
        run_train(args, device, rank, world, dist)
        if dist is not None:
            dist.destroy_process_group()
        return
    model = build_model(device, args.workload)
    B = args.batch
    n_rec, n_lig = args.n_rec, args.n_lig
    if args.ragged:
        gen = torch.Generator().manual_seed(77 + rank)
        n_rec = torch.randint(150, 601, (B,), generator=gen).tolist()
        n_lig = torch.randint(15, 36, (B,), generator=gen).tolist()
    g = build_batch(model, B, n_rec, n_lig, seed=1234 + rank * B, device=device, workload=args.workload)
    bidx = G.get_batch_idxs(g)
    eng = model.dynamics.engine()
    is_egnn = args.workload == 'egnn_all_atom'
    ones = torch.ones(B, device=device)
    # Random-init weights do not denoise: left to itself the chain drives the ligand atoms apart and
    # the lig-lig radius graph empties within ~20 steps, which would shrink the measured work.  Every
    # step therefore starts from the t = T state (x_0, h_0 ~ N(0, I), ligand COM removed: the complete
    # 600-edge lig-lig graph of SURVEY.md 8(d)); restoring it is three small device copies.
    lig, kp = g.nodes['lig'].data, g.nodes['kp'].data
    init = (lig['x_0'].clone(), lig['h_0'].clone(), kp['x_0'].clone())

    step_graph = None
    if args.graph:
        with torch.no_grad():
            step_graph = model.capture_step(g, bidx)

    def step(i):
        si = N_TIMESTEPS - 1 - (i % N_TIMESTEPS)
        if step_graph is not None:
            step_graph.step(si / N_TIMESTEPS, (si + 1) / N_TIMESTEPS)
        else:
            model.sample_p_zs_given_zt(ones * (si / N_TIMESTEPS), ones * ((si + 1) / N_TIMESTEPS), g, bidx)
        lig['x_0'].copy_(init[0]), lig['h_0'].copy_(init[1]), kp['x_0'].copy_(init[2])

    with torch.no_grad():
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        if is_egnn:
            eng.profile(True)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        if dist is not None:
            from keypoint_diffusion_amd.dist import all_gather_ligands
            all_gather_ligands(g)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    if not is_egnn:
        if rank == 0:
            print(json.dumps({'metric': 'denoising steps/sec', 'value': world * args.steps / elapsed, 'unit': 'steps/s',
                              'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                              'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
                              'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                              'config': {'workload': args.workload + (' ragged 150-600/15-35' if args.ragged else ''),
                                         'batch_per_gpu': B, 'n_kp_total': g.num_nodes('kp'), 'n_lig_total': g.num_nodes('lig'),
                                         'n_kk': g.num_edges('kk')},
                              'complex_steps_per_s': world * args.steps / elapsed * B}))
        if dist is not None:
            dist.destroy_process_group()
        return
    edge_ms, edge_launches = eng.profile_read()
    eng.profile(False)
    counts = eng.last_counts()

    if dist is not None:
        tmax = torch.tensor([elapsed], device='cpu' if share else device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        # HBM traffic of the dominant kernel comes from PMC counters, which need their own rocprofv3 passes
        # (profiles/tools/collect_round.sh); the committed per-launch figure for this workload is attached
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
        if os.path.exists(tfile) and B == 64 and args.n_rec == 300 and args.n_lig == 25:
            traffic = json.load(open(tfile))['hbm_bytes_per_launch']
        steps_per_s = world * args.steps / elapsed
        n_edges = counts['E_ll'] + counts['E_kl'] + counts['E_lk'] + counts['E_kk']
        edge_avg_s = edge_ms / max(edge_launches, 1) * 1e-3
        achieved = n_edges * EDGE_KERNEL_FLOP_PER_EDGE / edge_avg_s / 1e12 if edge_avg_s > 0 else 0.0
        hbm_gbs = n_edges * EDGE_ALGO_BYTES_PER_EDGE / edge_avg_s / 1e9 if edge_avg_s > 0 else 0.0
        out = {
            'metric': 'denoising steps/sec', 'value': steps_per_s, 'unit': 'steps/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'egnn_all_atom dynamics (6 EGNN layers, hidden 256, update_kp_feat), batch of {B} '
                                   f'synthetic {"150-600" if args.ragged else args.n_rec}-atom pockets / '
                                   f'{"15-35" if args.ragged else args.n_lig}-atom ligands per GPU, '
                                   f'T={N_TIMESTEPS}, seeded random-init weights, every step taken from the t=T ligand state',
                       'batch_per_gpu': B, 'n_rec': 'U{150..600}' if args.ragged else args.n_rec,
                       'n_lig': 'U{15..35}' if args.ragged else args.n_lig, 'parallelism': f'dp{world}'},
            'complex_steps_per_s': steps_per_s * B,
            'ligands_per_min': steps_per_s * B * 60.0 / N_TIMESTEPS,
            'edges_per_step_layer': counts,
            'roofline': {'kernel': 'k_egnn_edge', 'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F32_MATRIX_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': achieved / PEAK_F32_MATRIX_TFLOPS, 'traffic': traffic,
                         'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC, 2 x FETCH_SIZE + WRITE_SIZE, separate passes)',
                         'avg_launch_ms': edge_avg_s * 1e3, 'launches': edge_launches,
                         'flop_per_launch': n_edges * EDGE_KERNEL_FLOP_PER_EDGE,
                         'reference_formulation_tflops': n_edges * EDGE_ALGO_FLOP_PER_EDGE / edge_avg_s / 1e12
                         if edge_avg_s > 0 else 0.0,
                         'hbm': {'achieved': hbm_gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': hbm_gbs / PEAK_HBM_GBS,
                                 'bytes_per_launch': n_edges * EDGE_ALGO_BYTES_PER_EDGE}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
            out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
