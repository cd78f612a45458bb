"""TEST INFRASTRUCTURE ONLY (tests/, smoke() and bench.py's cpu_baseline may import this; the product never does).

Independent restatement of the receptor-encoder loss of losses/rec_encoder_loss.py:11-18, 49-82 -- the optimal-transport
distance between n keypoints and m target points with uniform masses and squared Euclidean cost -- by a different algorithm
than the product's linear program: with L = lcm(n, m) every keypoint is replicated L / n times and every target L / m times, and
the transport problem becomes an L x L assignment problem (unit masses 1 / L; the transport polytope's vertices are integral in
that unit), solved by the Hungarian method.  Parity pinning: the reference solves the same program with POT's network simplex (`ot.emd`,
absent here); the optimum VALUE of a linear program is unique, and the closed-form cases in tests/test_rec_encoder_loss.py
(identical sets, points on a line) pin both implementations."""
import numpy as np
import torch


def ot_value_by_assignment(kp: torch.Tensor, tgt: torch.Tensor) -> float:
    from scipy.optimize import linear_sum_assignment
    n, m = kp.shape[0], tgt.shape[0]
    L = int(np.lcm(n, m))
    assert L <= 4096, 'keep the assignment form to test sizes'
    cost = torch.cdist(kp.double(), tgt.double()).square().numpy()
    big = np.repeat(np.repeat(cost, L // n, axis=0), L // m, axis=1)               # [L, L]
    r, c = linear_sum_assignment(big)
    return float(big[r, c].sum() / L)


def ot_loss(kp_sets, tgt_sets) -> float:
    """Mean over the complexes (rec_encoder_loss.py:68, :81)."""
    return float(np.mean([ot_value_by_assignment(k, t) for k, t in zip(kp_sets, tgt_sets)]))
