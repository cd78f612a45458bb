"""Shared helpers for the test-suite: batch construction and product <-> oracle conversion."""
import torch

from keypoint_diffusion_amd import graph as G
from keypoint_diffusion_amd import synth
from oracle.batch import OBatch

CUTOFFS_ALL_ATOM = {'kk': 8, 'kl': 6, 'll': 6, 'rk': 100, 'rr': 3.5}

EGNN_C2 = dict(n_layers=6, hidden_nf=256, use_tanh=True, message_norm=0, update_kp_feat=True, norm=True,
               ll_k=0, kl_k=5, no_cg=False)
EGNN_DEV = dict(n_layers=6, hidden_nf=256, use_tanh=True, message_norm=0, update_kp_feat=False, norm=True,
                ll_k=0, kl_k=5)


def to_obatch(g: G.HeteroBatch) -> OBatch:
    """Product graph container -> oracle batch (CPU copies)."""
    cpu = lambda t: t.detach().cpu()
    ob = OBatch(n={nt: cpu(g.batch_num_nodes(nt)) for nt in g.ntypes}, x={}, h={}, v={}, edges={})
    for nt in g.ntypes:
        d = g.nodes[nt].data
        if 'x_0' in d:
            ob.x[nt] = cpu(d['x_0']).float()
        if 'h_0' in d:
            ob.h[nt] = cpu(d['h_0']).float()
        if 'v_0' in d:
            ob.v[nt] = cpu(d['v_0']).float()
    for et in g.etypes:
        s, d = g.edges(etype=et)
        ob.edges[et] = (cpu(s), cpu(d))
    return ob


def fixed_encode(g: G.HeteroBatch, n_vec=None) -> G.HeteroBatch:
    """What FixedReceptorEncoder does (kp := rec, kk := rr), on the container, without the module."""
    from keypoint_diffusion_amd.receptor_encoder_fixed import FixedReceptorEncoder
    return FixedReceptorEncoder(n_vec)(g, G.get_batch_idxs(g))


def make_batch(n_rec, n_lig, cutoffs=CUTOFFS_ALL_ATOM, seed=1234, n_rec_feat=10, n_keypoints=20,
               density=synth.ATOM_DENSITY) -> G.HeteroBatch:
    gs = synth.synth_complexes(n_rec, n_lig, n_keypoints, cutoffs, seed=seed, n_rec_feat=n_rec_feat, density=density)
    return G.batch(gs)


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b| -- the 'within 1e-4 rel fp32' measure of BASELINE.json."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
