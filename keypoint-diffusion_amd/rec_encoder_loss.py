"""Receptor-encoder loss (losses/rec_encoder_loss.py:10-120): the optimal-transport distance between the keypoints and
the receptor atoms (or the interface points) of every complex, uniform masses on both sides, squared Euclidean cost.

The reference solves the transport problem on the host with `ot.emd` (POT, exact network simplex, :11-18); POT is not in
this image, so the same linear program -- min <P, C> s.t. P 1 = 1/n, P^T 1 = 1/m, P >= 0 -- is solved exactly by the library's own
host-side min-cost-flow solver (`kpd_ot_emd_uniform`, csrc/ot.hip: successive shortest paths with potentials, double precision, a few
ms per 40 x 300 problem, the complexes of a batch on parallel host threads; round 2 used HiGHS through scipy, 0.45 s per complex, which
no longer fits once these models train).  The optimal VALUE of the program is unique, so the loss equals the reference's; as there,
the plan is a constant and the gradient flows through the cost matrix only.  Host-side code in both implementations: not a kernel.
The module's `begin` / `finish` halves let KeypointDiffusion.forward launch the denoiser while the host threads solve."""
from typing import List, Optional

import torch
import torch.nn as nn

from . import graph as G
from . import hip


def transport_plans(costs: List[torch.Tensor]) -> List[torch.Tensor]:
    """Exact optimal plans (uniform masses) of a batch of [n_i, m_i] cost matrices, as constants on the costs' device."""
    if any(c.shape[0] == 0 or c.shape[1] == 0 for c in costs):
        raise ValueError('optimal transport needs at least one point on either side')
    plans = hip.ot_emd_uniform([c.detach().double().cpu().numpy() for c in costs])
    return [torch.from_numpy(p).to(device=c.device, dtype=c.dtype) for p, c in zip(plans, costs)]


def compute_ot_emd(cost_mat: torch.Tensor, device=None):
    """(sum(P * cost), P) with P the optimal plan, detached (rec_encoder_loss.py:11-18)."""
    plan_t = transport_plans([cost_mat])[0]
    if device is not None:
        plan_t = plan_t.to(device)
    return torch.sum(plan_t * cost_mat), plan_t


def _padded(x: torch.Tensor, counts: List[int]) -> torch.Tensor:
    """[B, max(counts), 3] view / gather of flat graph-major points; rows past a complex's count are other complexes' points (finite,
    multiplied by a zero plan entry afterwards)."""
    B, width = len(counts), max(counts)
    if all(c == width for c in counts):
        return x.reshape(B, width, x.shape[1])
    off = torch.tensor([0] + counts[:-1], device=x.device).cumsum(0)
    idx = (off[:, None] + torch.arange(width, device=x.device)[None, :]).clamp_(max=x.shape[0] - 1)
    return x[idx]


class PendingLoss:
    """The encoder loss of one batch between `begin` (costs on the device, copied to the host, the exact plans being solved on host
    threads) and `finish` (plans back on the device, sum(P * C) / B).  Whatever the caller launches in between -- the denoiser's forward,
    in KeypointDiffusion.forward -- overlaps the host solve."""

    def __init__(self, value=None):
        self.value, self.cost, self.counts, self._thread, self._plans, self._err = value, None, None, None, None, None

    def _solve(self, host_costs):
        try:
            self._plans = hip.ot_emd_uniform(host_costs)
        except BaseException as e:          # re-raised by finish() on the caller's thread
            self._err = e

    def abandon(self):
        """Join the solver thread without using its result (the caller is unwinding an exception of its own)."""
        if self._thread is not None:
            self._thread.join()

    def finish(self) -> torch.Tensor:
        if self.value is not None:
            return self.value
        self._thread.join()
        if self._err is not None:
            raise self._err
        import numpy as np
        plan = np.zeros(tuple(self.cost.shape), dtype=np.float32)
        for b, (p, (n, m)) in enumerate(zip(self._plans, self.counts)):
            plan[b, :n, :m] = p
        plan_t = torch.from_numpy(plan).to(self.cost.device)
        self.value = torch.sum(plan_t * self.cost) / len(self.counts)
        return self.value


class ReceptorEncoderLoss(nn.Module):
    """Same constructor, loss types and errors as the reference (rec_encoder_loss.py:20-47): 'optimal_transport',
    'none', and the two types the reference itself refuses to evaluate ('gaussian_repulsion', 'hinge' raise
    NotImplementedError there, :84-86, :105-107)."""

    def __init__(self, loss_type='optimal_transport', use_interface_points: bool = False, hinge_threshold: float = 4):
        super().__init__()
        if loss_type not in ('optimal_transport', 'gaussian_repulsion', 'hinge', 'none'):
            raise ValueError
        self.loss_type, self.use_interface_points, self.hinge_threshold = loss_type, use_interface_points, hinge_threshold

    def begin(self, batched_complex_graphs=None, interface_points: Optional[List[torch.Tensor]] = None) -> PendingLoss:
        """First half of `forward`: every complex's squared-distance matrix in ONE padded [B, K, M] tensor (a handful of launches and
        one device-to-host copy for the whole batch), then the exact plans on a background host thread."""
        g = batched_complex_graphs
        if self.loss_type == 'none':
            return PendingLoss(torch.tensor(0.0, device=g.device, dtype=g.nodes['rec'].data['x_0'].dtype))
        if self.loss_type in ('gaussian_repulsion', 'hinge'):
            raise NotImplementedError
        kp_x = g.nodes['kp'].data['x_0']
        n_kp = g.batch_num_nodes('kp').tolist()
        if self.use_interface_points:                                             # :71-82
            targets = [t.to(kp_x.device) for t in interface_points]
            n_tgt = [int(t.shape[0]) for t in targets]
            tgt_x = torch.cat(targets) if targets else kp_x.new_zeros(0, 3)
        else:                                                                     # :49-69
            tgt_x, n_tgt = g.nodes['rec'].data['x_0'], g.batch_num_nodes('rec').tolist()
        if len(n_tgt) != len(n_kp):
            raise ValueError(f'{len(n_tgt)} target point sets for {len(n_kp)} complexes')
        if min(n_kp) == 0 or min(n_tgt) == 0:
            raise ValueError('optimal transport needs at least one point on either side')
        kp_p, tgt_p = _padded(kp_x, n_kp), _padded(tgt_x.to(kp_x.dtype), n_tgt)
        pend = PendingLoss()
        pend.cost = (kp_p[:, :, None, :] - tgt_p[:, None, :, :]).square().sum(-1)            # squared Euclidean, [B, K, M]
        pend.counts = list(zip(n_kp, n_tgt))
        host = pend.cost.detach().double().cpu().numpy()
        import threading
        pend._thread = threading.Thread(target=pend._solve, args=([host[b, :n, :m] for b, (n, m) in enumerate(pend.counts)],))
        pend._thread.start()
        return pend

    def forward(self, batched_complex_graphs=None, interface_points: Optional[List[torch.Tensor]] = None):
        return self.begin(batched_complex_graphs, interface_points).finish()
