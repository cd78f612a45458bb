"""Receptor-encoder loss (losses/rec_encoder_loss.py:10-120): the optimal-transport distance between the keypoints and
the receptor atoms (or the interface points) of every complex, uniform masses on both sides, squared Euclidean cost.

The reference solves the transport problem on the host with `ot.emd` (POT, exact network simplex, :11-18); POT is not in
this image, so the same linear program -- min <P, C> s.t. P 1 = 1/n, P^T 1 = 1/m, P >= 0 -- is solved exactly with HiGHS
through scipy.  The optimal VALUE of the program is unique, so the loss equals the reference's; as there, the plan is a
constant and the gradient flows through the cost matrix only.  It is host-side code in both implementations (a 40 x 300
program per complex, once per training batch): not a kernel."""
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import graph as G


def _transport_plan(cost: np.ndarray) -> np.ndarray:
    """Exact optimal plan of the uniform-mass transport problem for an [n, m] cost matrix."""
    from scipy.optimize import linprog
    from scipy.sparse import coo_matrix
    n, m = cost.shape
    if n == 0 or m == 0:
        raise ValueError('optimal transport needs at least one point on either side')
    rows = np.concatenate([np.repeat(np.arange(n), m), n + np.tile(np.arange(m), n)])
    cols = np.concatenate([np.arange(n * m), np.arange(n * m)])
    A = coo_matrix((np.ones(2 * n * m), (rows, cols)), shape=(n + m, n * m)).tocsr()
    b = np.concatenate([np.full(n, 1.0 / n), np.full(m, 1.0 / m)])
    # one of the n + m mass constraints is implied by the others; HiGHS handles the redundancy
    res = linprog(cost.reshape(-1).astype(np.float64), A_eq=A, b_eq=b, bounds=(0, None), method='highs')
    if res.status != 0:
        raise RuntimeError(f'optimal-transport program did not solve: {res.message}')
    return res.x.reshape(n, m)


def compute_ot_emd(cost_mat: torch.Tensor, device=None):
    """(sum(P * cost), P) with P the optimal plan, detached (rec_encoder_loss.py:11-18)."""
    plan = _transport_plan(cost_mat.detach().cpu().numpy())
    plan_t = torch.tensor(plan, device=device if device is not None else cost_mat.device).float()
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
        total = 0
        for kp_pos, tgt in zip(kp, targets):
            cost = torch.square(torch.cdist(kp_pos, tgt.to(kp_pos.device)))
            total = total + compute_ot_emd(cost, device=cost.device)[0]
        return total / len(kp)
