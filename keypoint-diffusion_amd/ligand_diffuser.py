"""`KeypointDiffusion`: the diffusion wrapper around the denoiser, with the reference's public
surface (models/ligand_diffuser.py:24-538) on the HIP hot path.

Host code (this file) is tensor plumbing and the noise schedule; every per-timestep tensor
operation -- denoiser forward, z_s update, COM removal -- is a call into libkpd_hip.so.
`LigandDiffuser` is kept as an alias (BASELINE.json names the class that way).
"""
import pickle
from math import ceil
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import graph as G
from . import hip
from .dynamics import LigRecDynamics
from .dynamics_gvp import LigRecDynamicsGVP
from .receptor_encoder_fixed import FixedReceptorEncoder
from .receptor_encoder import ReceptorEncoder
from .receptor_encoder_gvp import ReceptorEncoderGVP
from .rec_encoder_loss import ReceptorEncoderLoss


class LigandSizeDistribution:
    """P(n_lig | n_rec) from the shipped joint histogram (models/n_nodes_dist.py:7-59)."""

    def __init__(self, processed_dataset_dir: Path):
        f = Path(processed_dataset_dir) / 'train_n_node_joint_dist.pkl'
        if not f.exists():
            raise ValueError(f'Joint distribution file {f} does not exist')
        with open(f, 'rb') as fh:
            hist, self.rec_bounds, self.lig_bounds = pickle.load(fh)
        self.joint_histogram = torch.from_numpy(hist)
        self.rec_idx_to_size = torch.arange(self.rec_bounds[0], self.rec_bounds[1] + 1)
        self.lig_idx_to_size = torch.arange(self.lig_bounds[0], self.lig_bounds[1] + 1)

    def sample(self, n_nodes_rec: torch.Tensor, n_replicates: int) -> torch.Tensor:
        lo, hi = self.rec_bounds
        clipped = n_nodes_rec.clamp(min=lo, max=hi)
        for a, b in zip(n_nodes_rec.tolist(), clipped.tolist()):
            if a != b:
                print(f'WARNING: Number of receptor nodes {a} is not in the range {self.rec_bounds} from the '
                      f'training set; conditioning on {b} nodes')
        rows = self.joint_histogram[(clipped - lo).long()]
        idx = torch.multinomial(rows, n_replicates, replacement=True)
        return self.lig_idx_to_size[idx]


def polynomial_schedule(timesteps: int, s: float = 1e-4, power: float = 3.0) -> np.ndarray:
    """alpha^2 of the clipped polynomial schedule (ligand_diffuser.py:620-650)."""
    steps = timesteps + 1
    t = np.linspace(0, steps, steps)
    a2 = np.concatenate([np.ones(1), (1 - np.power(t / steps, power)) ** 2])
    a2 = np.cumprod(np.clip(a2[1:] / a2[:-1], a_min=0.001, a_max=1.0))
    return (1 - 2 * s) * a2 + s


class PredefinedNoiseSchedule(nn.Module):
    """Lookup table gamma[0..T] = -(log alpha^2 - log sigma^2) (ligand_diffuser.py:654-690)."""

    def __init__(self, noise_schedule: str, timesteps: int, precision: float):
        super().__init__()
        self.timesteps = timesteps
        if not noise_schedule.startswith('polynomial_'):
            raise ValueError(noise_schedule)
        a2 = polynomial_schedule(timesteps, s=precision, power=float(noise_schedule.split('_')[1]))
        gamma = -(np.log(a2) - np.log(1 - a2))
        self.gamma = nn.Parameter(torch.from_numpy(gamma).float(), requires_grad=False)

    def forward(self, t: torch.Tensor) -> torch.Tensor:
        return self.gamma[torch.round(t * self.timesteps).long()]


class StepGraph:
    """A captured reverse step.  `step(s, t)` writes the two scalars into static device buffers and replays the graph;
    the graph holds the denoiser forward (graph build included), the noise draw and the in-place z_s update."""

    def __init__(self, model: 'KeypointDiffusion', g, bidx=None, noise=None):
        dev, B = g.device, g.batch_size
        if dev.type != 'cuda':
            raise hip.KpdError('a step graph needs the batch on the GPU')
        self.s, self.t = torch.zeros(B, device=dev), torch.ones(B, device=dev)
        lig, kp = g.nodes['lig'].data, g.nodes['kp'].data
        state = [lig['x_0'], lig['h_0'], kp['x_0']]
        for i, t in enumerate(state):
            if not (t.is_contiguous() and t.dtype == torch.float32):
                raise hip.KpdError('graph capture needs contiguous fp32 state tensors (they are updated in place)')
        saved = [t.clone() for t in state]
        T = model.n_timesteps
        self.s.fill_((T - 1) / T)
        # warm-up outside capture (workspace reservation, first-use initialisation), on a side stream as capture requires
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                model.sample_p_zs_given_zt(self.s, self.t, g, bidx, noise=noise)
        torch.cuda.current_stream(dev).wait_stream(side)
        for t, c in zip(state, saved):
            t.copy_(c)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            model.sample_p_zs_given_zt(self.s, self.t, g, bidx, noise=noise)
        self._keep = (g, noise)
        # The captured kernels hold raw pointers into the engine's workspace arena and packed weights.  Holding the engine
        # object keeps both allocations alive even if the module builds a new engine; the pin records what must not have
        # changed for a replay to mean "one reverse step of this model": the arena (a larger batch re-reserves it, which
        # frees the captured one) and the weights the engine was packed from.
        self._model = model
        self._engine = model.dynamics.engine()
        self._pin = (self._engine._reserved, model.dynamics._weights_key())
        ps = list(model.dynamics.parameters())
        self._sentinels = [ps[i] for i in sorted({0, len(ps) // 2, len(ps) - 1})] if ps else []
        self._sentinel_key = [(p.data_ptr(), p._version) for p in self._sentinels]

    def _check_pin(self):
        """Every replay: the engine object and its arena are the captured ones (two attribute reads) and three sentinel parameters
        (first, middle, last) still have the captured storage and version counter -- whatever rewrites the weights as a whole
        (an optimizer step, load_state_dict, an EMA swap, .to()) moves every one of them, so such a change raises on the NEXT
        replay, not up to 15 replays later.  Every 16th replay, starting with the first: all weights are the captured ones (a walk
        over all parameters -- host time the graph exists to avoid), which also catches a change to a single tensor."""
        dyn = self._model.dynamics
        self._replays = getattr(self, '_replays', -1) + 1
        if dyn._engine is not self._engine or self._engine._reserved != self._pin[0]:
            raise hip.KpdError('stale step graph: the denoiser engine was rebuilt or its workspace re-reserved (a larger batch ran) '
                               'after capture; the captured kernels point into freed memory -- capture the step again')
        if ([(p.data_ptr(), p._version) for p in self._sentinels] != self._sentinel_key or
                (self._replays % 16 == 0 and dyn._weights_key() != self._pin[1])):
            raise hip.KpdError('stale step graph: the model weights changed after capture (the graph replays the weights packed at '
                               'capture time) -- capture the step again')

    def step(self, s: float, t: float):
        self._check_pin()
        self.s.fill_(s)
        self.t.fill_(t)
        self.graph.replay()


class KeypointDiffusion(nn.Module):

    def __init__(self, atom_nf, rec_nf, processed_dataset_dir: Optional[Path], n_timesteps: int = 1000,
                 keypoint_centered=False, architecture: str = 'egnn', rec_encoder_type: str = 'learned',
                 graph_config={}, dynamics_config={}, rec_encoder_config={}, rec_encoder_loss_config={},
                 precision=1e-4, lig_feat_norm_constant=1, rl_dist_threshold=0, use_fake_atoms=False):
        super().__init__()
        if architecture not in ('egnn', 'gvp'):
            raise ValueError(f'Unsupported architecture: {architecture}')
        if rec_encoder_type not in ('learned', 'fixed'):
            raise ValueError(f'Receptor encoder type must be either "learned" or "fixed". Got {rec_encoder_type=} instead.')
        self.n_lig_features, self.n_kp_feat, self.n_timesteps = atom_nf, rec_nf, n_timesteps
        self.lig_feat_norm_constant = lig_feat_norm_constant
        self.use_fake_atoms, self.rec_encoder_type, self.architecture = use_fake_atoms, rec_encoder_type, architecture
        self.rl_dist_threshold = rl_dist_threshold
        if use_fake_atoms:
            raise NotImplementedError('fake atoms are unused by every shipped config (max_fake_atom_frac: 0.0) and '
                                      'the reference implementation of their removal cannot run (ligand_diffuser.py:559)')
        # the ligand-size prior is host-side and optional here: synthetic benches have no dataset directory
        self.lig_size_dist = LigandSizeDistribution(processed_dataset_dir) if processed_dataset_dir is not None else None
        self.gamma = PredefinedNoiseSchedule('polynomial_2', timesteps=n_timesteps, precision=precision)

        dynamics_config = dict(dynamics_config)
        if 'no_cg' in rec_encoder_config:
            dynamics_config['no_cg'] = rec_encoder_config['no_cg']
        dyn_cls = LigRecDynamics if architecture == 'egnn' else LigRecDynamicsGVP
        self.dynamics = dyn_cls(atom_nf, rec_nf, **graph_config, **dynamics_config)

        if rec_encoder_type == 'learned':
            enc_cls = ReceptorEncoder if architecture == 'egnn' else ReceptorEncoderGVP      # ligand_diffuser.py:62-67
            self.rec_encoder = enc_cls(**graph_config, **rec_encoder_config)
        else:
            self.rec_encoder = FixedReceptorEncoder(
                n_vec_feats=rec_encoder_config['vector_size'] if architecture == 'gvp' else None)
        rec_encoder_loss_config = dict(rec_encoder_loss_config)
        if rec_encoder_type == 'fixed':
            rec_encoder_loss_config['loss_type'] = 'none'                                      # ligand_diffuser.py:85-87
        self.rec_encoder_loss_fn = ReceptorEncoderLoss(**rec_encoder_loss_config)

    # ---- training entry point ----------------------------------------------------------
    def forward(self, complex_graphs, interface_points):
        """Losses of one batch (ligand_diffuser.py:89-175): {'l2', 'pos', 'feat', 'rec_encoder'}.

        Trainable end to end in all eight shipped configurations and dev_config: the noise prediction is differentiated by the HIP
        backward passes of the denoisers (kpd_egnn_trainer_* / kpd_gvp_trainer_*, with respect to the keypoint positions, features
        and vectors too), the learned keypoints by the backward passes of their encoder (kpd_recenc_trainer_* for gvp_20kp /
        gvp_40kp, kpd_recegnn_trainer_* for egnn_20kp / egnn_40kp), and the optimal-transport encoder loss (rec_encoder_loss.py)
        adds its gradient at the keypoint positions.  With a fixed encoder there are no encoder parameters and that loss is the
        constant 0 (:85-87)."""
        if self.rl_dist_threshold > 0:
            raise NotImplementedError('the receptor-ligand hinge loss (rl_dist_threshold > 0) is unused by every shipped config')
        losses = {}
        g = self.normalize(complex_graphs)
        batch_size, device = g.batch_size, g.device
        batch_idxs = G.get_batch_idxs(g)
        g = self.rec_encoder(g, batch_idxs)
        if self.rec_encoder_type == 'fixed':
            batch_idxs = G.get_batch_idxs(g)                  # :106-107: keypoints = receptor atoms now
        # :115; the exact transport plans are solved on host threads while the denoiser's forward is launched below
        losses['rec_encoder'] = None
        pending = self.rec_encoder_loss_fn.begin(g, interface_points=interface_points)
        try:
            g = self.remove_com(g, batch_idxs['lig'], batch_idxs['kp'], com='ligand')
            t = torch.randint(0, self.n_timesteps, size=(batch_size,), device=device).float() / self.n_timesteps
            eps = {'h': torch.randn(g.nodes['lig'].data['h_0'].shape, device=device),
                   'x': torch.randn(g.nodes['lig'].data['x_0'].shape, device=device)}
            gamma_t = self.gamma(t).to(device=device)
            g = self.noised_representation(g, batch_idxs['lig'], batch_idxs['kp'], eps, gamma_t)
            eps_h_pred, eps_x_pred = self.dynamics(g, t, batch_idxs)
        except BaseException:
            pending.abandon()                # the denoiser raised: do not leave the solver thread running behind the exception
            raise
        losses['rec_encoder'] = pending.finish()
        x_loss = (eps['x'] - eps_x_pred).square().sum()
        n_x_loss_terms = eps['x'].numel()
        h_loss = (eps['h'] - eps_h_pred).square().sum()
        losses['l2'] = (x_loss + h_loss) / (n_x_loss_terms + eps['h'].numel())
        losses['pos'] = x_loss / n_x_loss_terms
        losses['feat'] = h_loss / eps['h'].numel()
        return losses

    def noised_representation(self, g, lig_batch_idx, kp_batch_idx, eps, gamma_t):
        """z_t = alpha_t z_0 + sigma_t eps, then ligand-COM removal (ligand_diffuser.py:205-219)."""
        alpha_t = self.alpha(gamma_t)[lig_batch_idx][:, None]
        sigma_t = self.sigma(gamma_t)[lig_batch_idx][:, None]
        g.nodes['lig'].data['x_0'] = alpha_t * g.nodes['lig'].data['x_0'] + sigma_t * eps['x']
        g.nodes['lig'].data['h_0'] = alpha_t * g.nodes['lig'].data['h_0'] + sigma_t * eps['h']
        return self.remove_com(g, lig_batch_idx, kp_batch_idx, com='ligand')

    def denoised_representation(self, g, lig_batch_idx, kp_batch_idx, eps_x_pred, eps_h_pred, gamma_t):
        """ligand_diffuser.py:221-230."""
        alpha_t = self.alpha(gamma_t)[lig_batch_idx][:, None]
        sigma_t = self.sigma(gamma_t)[lig_batch_idx][:, None]
        g.nodes['lig'].data['x_0'] = (g.nodes['lig'].data['x_0'] - sigma_t * eps_x_pred) / alpha_t
        g.nodes['lig'].data['h_0'] = (g.nodes['lig'].data['h_0'] - sigma_t * eps_h_pred) / alpha_t
        return g

    # ---- small host helpers ------------------------------------------------------------
    def normalize(self, g):
        g.nodes['lig'].data['h_0'] = g.nodes['lig'].data['h_0'] / self.lig_feat_norm_constant
        return g

    def unnormalize(self, g):
        g.nodes['lig'].data['h_0'] = g.nodes['lig'].data['h_0'] * self.lig_feat_norm_constant
        return g

    def remove_com(self, g, lig_batch_idx, kp_batch_idx, com: str = None):
        if com is None:
            raise NotImplementedError('removing COM of receptor/ligand complex not implemented')
        if com not in ('ligand', 'receptor'):
            raise ValueError(f'invalid value for com: {com=}')
        c = G.readout_nodes(g, feat='x_0', ntype='lig' if com == 'ligand' else 'kp', op='mean')
        g.nodes['lig'].data['x_0'] = g.nodes['lig'].data['x_0'] - c[lig_batch_idx]
        g.nodes['kp'].data['x_0'] = g.nodes['kp'].data['x_0'] - c[kp_batch_idx]
        return g

    def sigma(self, gamma):
        return torch.sqrt(torch.sigmoid(gamma))

    def alpha(self, gamma):
        return torch.sqrt(torch.sigmoid(-gamma))

    def sigma_and_alpha_t_given_s(self, gamma_t, gamma_s):
        sigma2 = -torch.expm1(F.softplus(gamma_s) - F.softplus(gamma_t))
        log_alpha2 = F.logsigmoid(-gamma_t) - F.logsigmoid(-gamma_s)
        return sigma2, torch.sqrt(sigma2), torch.exp(0.5 * log_alpha2)

    # ---- sampling ----------------------------------------------------------------------
    def encode_receptors(self, g):
        return self.rec_encoder(g, G.get_batch_idxs(g))

    def step_coefficients(self, s: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """[B,3] = (alpha_t|s, sigma^2_t|s / alpha_t|s / sigma_t, sigma_t|s sigma_s / sigma_t)
        (ligand_diffuser.py:505-526).  On the GPU this is one kernel (kpd_step_coefficients); host tensors take
        the same arithmetic through the torch mirror below (schedule inspection, tests)."""
        if s.is_cuda:
            return hip.step_coefficients(self.gamma.gamma, s, t)
        g_s, g_t = self.gamma(s), self.gamma(t)
        sigma2_ts, sigma_ts, alpha_ts = self.sigma_and_alpha_t_given_s(g_t, g_s)
        sig_s, sig_t = self.sigma(g_s), self.sigma(g_t)
        return torch.stack([alpha_ts, sigma2_ts / alpha_ts / sig_t, sigma_ts * sig_s / sig_t], dim=1).contiguous()

    def use_complex_noise(self, seed):
        """Opt in to sharding-invariant noise: every draw of the sampler becomes a function of (seed, complex id, timestep,
        position in the complex) (kpd_complex_noise), so a run split over ranks reproduces the single-process run.
        `seed=None` returns to the reference behaviour (one global torch.randn per draw)."""
        self._noise_seed = None if seed is None else int(seed)
        return self

    def _draw(self, g, width, complex_ids, step, tag):
        if getattr(self, '_noise_seed', None) is None or complex_ids is None:
            return torch.randn(g.num_nodes('lig'), width, device=g.device)
        return hip.complex_noise(g.prepared(), width, complex_ids, self._noise_seed, step, tag)

    def sample_p_zs_given_zt(self, s, t, g, batch_idxs=None, noise=None, complex_ids=None, step=0):
        """One reverse step (ligand_diffuser.py:497-538).  `noise` = (pos_noise, feat_noise) may be
        injected for reproducible parity tests; by default it is drawn with torch.randn as upstream
        (or per complex, `use_complex_noise`, when `complex_ids` [B] int64 and the integer `step` are given)."""
        lig, kp = g.nodes['lig'].data, g.nodes['kp'].data
        for d, k in ((lig, 'x_0'), (lig, 'h_0'), (kp, 'x_0')):
            if not (d[k].is_contiguous() and d[k].dtype == torch.float32):
                d[k] = d[k].contiguous().float()
        coef = self.step_coefficients(s, t)
        eps_h, eps_x = self.dynamics(g, t, batch_idxs)
        if noise is None:
            noise = (self._draw(g, 3, complex_ids, step, 0), self._draw(g, lig['h_0'].shape[1], complex_ids, step, 1))
        hip.sample_update(g.prepared(), self.n_lig_features, lig['x_0'], lig['h_0'], kp['x_0'], eps_x, eps_h,
                          noise[0], noise[1], coef)
        return g

    @torch.no_grad()
    def capture_step(self, g, bidx=None, noise=None) -> 'StepGraph':
        """One reverse step (`sample_p_zs_given_zt`) captured as a HIP graph for this batch: replaying it costs one
        launch instead of ~30.  The step's kernels take shapes from host-known capacities and counts from device memory,
        so the same graph serves every timestep.  Measured gain is small (B = 1: 0.99 -> 0.96 ms/step, B = 64: 8.32 ->
        8.29): the step is bound by its chain of dependent kernels, not by launch overhead."""
        return StepGraph(self, g, bidx, noise)

    @torch.no_grad()
    def sample_from_encoded_receptors(self, g, visualize=False, init_lig_pos: torch.Tensor = None, complex_ids=None,
                                      use_graph: Optional[bool] = None):
        """Full reverse loop for a batch of encoded pockets (ligand_diffuser.py:342-469).  `complex_ids` [B] int64
        (global index of every complex in the job) selects the per-complex noise streams of `use_complex_noise`.
        `use_graph=True`: replay the reverse step as a captured HIP graph (not with the per-complex noise streams, which
        take the timestep as a launch argument); the default is the eager step, the measured difference is ≤ 3 %."""
        device, B = g.device, g.batch_size
        init_kp_com = G.readout_nodes(g, feat='x_0', op='mean', ntype='kp')
        bidx = G.get_batch_idxs(g)
        lig_b, kp_b = bidx['lig'], bidx['kp']
        if init_lig_pos is not None:
            assert init_lig_pos.shape == (B, 3)
            frame = init_lig_pos
        else:
            frame = G.readout_nodes(g, feat='x_0', op='mean', ntype='rec')
        g.nodes['kp'].data['x_0'] = g.nodes['kp'].data['x_0'] - frame[kp_b]
        if complex_ids is not None:
            complex_ids = complex_ids.to(device).long()
        for tag, feat in enumerate(('x_0', 'h_0')):
            g.nodes['lig'].data[feat] = self._draw(g, g.nodes['lig'].data[feat].shape[1], complex_ids, self.n_timesteps, tag)
        g = self.remove_com(g, lig_b, kp_b, com='ligand')

        def snapshot():
            f = G.copy_graph(g, n_copies=1, batched_graph=True)[0]
            f = self.unnormalize(f)
            delta = init_kp_com - G.readout_nodes(f, feat='x_0', ntype='kp', op='mean')
            f.nodes['lig'].data['x_0'] = f.nodes['lig'].data['x_0'] + delta[lig_b]
            parts = G.unbatch(f.to('cpu'))
            return [p.nodes['lig'].data['x_0'] for p in parts], [p.nodes['lig'].data['h_0'] for p in parts]

        frames_x, frames_h = [], []
        if visualize:
            fx, fh = snapshot()
            frames_x.append(fx), frames_h.append(fh)
        ones = torch.ones(B, device=device)
        per_complex = getattr(self, '_noise_seed', None) is not None and complex_ids is not None
        if use_graph and per_complex:
            raise ValueError('use_graph=True cannot be combined with per-complex noise streams (the timestep is a launch argument)')
        step_graph = self.capture_step(g, bidx) if use_graph else None
        for s in reversed(range(self.n_timesteps)):
            if step_graph is not None:
                step_graph.step(s / self.n_timesteps, (s + 1) / self.n_timesteps)
            else:
                g = self.sample_p_zs_given_zt(ones * (s / self.n_timesteps), ones * ((s + 1) / self.n_timesteps), g, bidx,
                                              complex_ids=complex_ids, step=s)
            if visualize:
                fx, fh = snapshot()
                frames_x.append(fx), frames_h.append(fh)

        g = self.remove_com(g, lig_b, kp_b, com='receptor')
        for nt in ('lig', 'kp'):
            g.nodes[nt].data['x_0'] = g.nodes[nt].data['x_0'] + init_kp_com[bidx[nt]]
        g = self.unnormalize(g)
        if visualize:
            return list(zip(*frames_x)), list(zip(*frames_h))
        parts = G.unbatch(g.to('cpu'))
        return [p.nodes['lig'].data['x_0'] for p in parts], [p.nodes['lig'].data['h_0'] for p in parts]

    @torch.no_grad()
    def _sample(self, ref_graphs: List[G.HeteroBatch], n_lig_atoms: List[List[int]], rec_enc_batch_size: int = 32,
                diff_batch_size: int = 32, visualize=False, use_ref_lig_com: bool = False, group=None):
        """Several pockets x several ligands per pocket (ligand_diffuser.py:271-340).

        The flat list of (pocket, replicate) complexes it builds (:292-313) is the unit of multi-GPU work:
        when a `torch.distributed` process group with more than one rank is initialised, every rank calls this with the SAME
        arguments, takes a contiguous, edge-count-balanced shard of that list (`dist.shard_complexes`), encodes only the pockets
        its shard touches, runs the reverse loop on its shard in `diff_batch_size` batches, and one all-gather of the ligand
        tensors (`dist.gather_ligand_lists`, RCCL over xGMI) returns ALL ligands, in input order, on every rank.  Noise then comes
        from the per-complex Philox streams keyed by the global complex index (`use_complex_noise`; switched on with a seed
        all ranks agree on if the caller did not choose one), so the sharded run reproduces the single-process run of the same
        seed up to fp32 summation order.  The pockets are encoded `rec_enc_batch_size` at a time."""
        from . import dist as D
        sharded = D.sharding_active(group)
        if sharded and visualize:
            raise ValueError('visualize=True returns whole trajectories and is a single-process feature')
        # the flat complex list: (pocket index, number of ligand atoms), in input order
        flat = [(i, int(n)) for i, sizes in enumerate(n_lig_atoms) for n in sizes]
        device = ref_graphs[0].device

        def run(mine):
            """Ligands of the complexes `mine` (a range into `flat`)."""
            pockets = sorted({flat[c][0] for c in mine})
            encoded = {}
            for lo in range(0, len(pockets), max(1, rec_enc_batch_size)):
                chunk = pockets[lo:lo + max(1, rec_enc_batch_size)]
                enc = self.encode_receptors(G.batch([ref_graphs[i] for i in chunk]))
                encoded.update(zip(chunk, G.unbatch(enc)))
            graphs = []
            for c in mine:           # one copy_graph call per (pocket, run of replicates) keeps the reference's copy semantics
                i, n = flat[c]
                graphs.extend(G.copy_graph(encoded[i], n_copies=1, lig_atoms_per_copy=torch.tensor([n])))
            pos, feat = [], []
            for lo in range(0, len(graphs), diff_batch_size):
                bg = G.batch(graphs[lo:lo + diff_batch_size])
                init = G.readout_nodes(bg, feat='x_0', op='mean', ntype='lig') if use_ref_lig_com else None
                ids = torch.arange(mine[lo], mine[lo] + bg.batch_size, dtype=torch.long)
                p, f = self.sample_from_encoded_receptors(bg, visualize=visualize, init_lig_pos=init, complex_ids=ids)
                pos.extend(p), feat.extend(f)
            return pos, feat

        if sharded:
            restore = getattr(self, '_noise_seed', None)
            if restore is None:
                self.use_complex_noise(D.common_seed(group))
            try:
                # cost of a complex ~ its edges per layer: kk + kl/lk grow with the pocket, ll with the ligand
                costs = [600.0 + 22.7 * ref_graphs[i].num_nodes('rec') + n * n for i, n in flat]
                lig_pos, lig_feat = D.sharded_map(costs, run, group=group, device=device)
            finally:
                self._noise_seed = restore
        else:
            lig_pos, lig_feat = run(range(len(flat)))
        samples, end = [], 0
        for i in range(len(ref_graphs)):
            start, end = end, end + len(n_lig_atoms[i])
            samples.append({'positions': lig_pos[start:end], 'features': lig_feat[start:end]})
        return samples

    @torch.no_grad()
    def sample_given_pocket(self, rec_graph, n_lig_atoms: torch.Tensor, rec_enc_batch_size: int = 32,
                            diff_batch_size: int = 32, visualize=False):
        s = self._sample([rec_graph], n_lig_atoms=[n_lig_atoms.tolist()], rec_enc_batch_size=rec_enc_batch_size,
                         diff_batch_size=diff_batch_size, visualize=visualize)
        return s[0]['positions'], s[0]['features']

    @torch.no_grad()
    def sample_random_sizes(self, ref_graphs, n_replicates: int = 10, rec_enc_batch_size: int = 32,
                            diff_batch_size: int = 32):
        if self.lig_size_dist is None:
            raise ValueError('no processed_dataset_dir was given: the ligand-size prior is unavailable')
        n_rec = torch.tensor([g.num_nodes('rec') for g in ref_graphs])
        n_lig = self.lig_size_dist.sample(n_rec, n_replicates)
        return self._sample(ref_graphs, n_lig_atoms=n_lig.tolist(), rec_enc_batch_size=rec_enc_batch_size,
                            diff_batch_size=diff_batch_size)


LigandDiffuser = KeypointDiffusion
