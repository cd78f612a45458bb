"""Receptor-encoder loss (losses/rec_encoder_loss.py:10-120): the optimal-transport distance between the keypoints and
the receptor atoms (or the interface points) of every complex, uniform masses on both sides, squared Euclidean cost.

The reference solves the transport problem on the host with `ot.emd` (POT, exact network simplex, :11-18); POT is not in
this image, so the same linear program -- min <P, C> s.t. P 1 = 1/n, P^T 1 = 1/m, P >= 0 -- is solved exactly by the library's own
host-side min-cost-flow solver (`kpd_ot_emd_uniform`, csrc/ot.hip: successive shortest paths with potentials, double precision, a few
ms per 40 x 300 problem, the complexes of a batch on parallel host threads; round 2 used HiGHS through scipy, 0.45 s per complex, which
no longer fits once these models train).  The optimal VALUE of the program is unique, so the loss equals the reference's; as there,
the plan is a constant and the gradient flows through the cost matrix only.  Host-side code in both implementations: not a kernel."""
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


class ReceptorEncoderLoss(nn.Module):
    """Same constructor, loss types and errors as the reference (rec_encoder_loss.py:20-47): 'optimal_transport',
    'none', and the two types the reference itself refuses to evaluate ('gaussian_repulsion', 'hinge' raise
    NotImplementedError there, :84-86, :105-107)."""

    def __init__(self, loss_type='optimal_transport', use_interface_points: bool = False, hinge_threshold: float = 4):
        super().__init__()
        if loss_type not in ('optimal_transport', 'gaussian_repulsion', 'hinge', 'none'):
            raise ValueError
        self.loss_type, self.use_interface_points, self.hinge_threshold = loss_type, use_interface_points, hinge_threshold

    def forward(self, batched_complex_graphs=None, interface_points: Optional[List[torch.Tensor]] = None):
        g = batched_complex_graphs
        if self.loss_type == 'none':
            return torch.tensor(0.0, device=g.device, dtype=g.nodes['rec'].data['x_0'].dtype)
        if self.loss_type in ('gaussian_repulsion', 'hinge'):
            raise NotImplementedError
        kp = [u.nodes['kp'].data['x_0'] for u in G.unbatch(g)]
        if self.use_interface_points:
            targets = list(interface_points)                                      # :71-82
        else:
            targets = [u.nodes['rec'].data['x_0'] for u in G.unbatch(g)]         # :49-69
        if len(targets) != len(kp):
            raise ValueError(f'{len(targets)} target point sets for {len(kp)} complexes')
        costs = [torch.square(torch.cdist(kp_pos, tgt.to(kp_pos.device))) for kp_pos, tgt in zip(kp, targets)]
        plans = transport_plans(costs)                         # one library call: the complexes are solved on parallel host threads
        total = 0
        for cost, plan in zip(costs, plans):
            total = total + torch.sum(plan * cost)
        return total / len(kp)
