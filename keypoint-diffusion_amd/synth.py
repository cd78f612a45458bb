"""Deterministic synthetic inputs and weights.

There are no datasets or checkpoints in this environment (trained_models/*/model.pt are
missing upstream), so parity tests and bench.py run on seeded synthetic pockets / ligands
and a seeded weight fill.  Nothing here is on the timed path.
"""
import math
import zlib
from typing import Dict, List, Optional, Sequence

import torch

from . import graph as G

ATOM_DENSITY = 0.059      # heavy atoms per cubic Angstrom inside a pocket (all-atom)
CA_DENSITY = 0.0074       # C-alpha pockets


def fill_state_dict_(module: torch.nn.Module, seed: int = 0) -> torch.nn.Module:
    """Overwrite every parameter with a seeded draw keyed by its `state_dict()` name.

    The same rule applied to a reference module and to this package's module of the same
    constructor arguments yields identical weights (this is how the golden fixtures pin the
    state-dict layout without storing weights).  Rule: 2-D+ tensors ~ N(0, 1/fan_in)
    (fan_in = shape[0] for GVP `Wh`/`Wu`, shape[1] otherwise; the EGNN coordinate head
    `coord_mlp.*.4.weight` is scaled by a further 0.1), LayerNorm-like 1-D `weight`
    ~ 1 + 0.1 N(0,1), every other 1-D tensor ~ 0.1 N(0,1); `gamma` buffers and empty
    tensors are left untouched.
    """
    sd = module.state_dict()
    with torch.no_grad():
        for key in sd:
            t = sd[key]
            if t.numel() == 0 or key.endswith('gamma') or not t.is_floating_point():
                continue
            # one generator per tensor, seeded from the key: independent of registration order
            gen = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 1000003 * seed) % (2 ** 62))
            if t.dim() >= 2:
                fan_in = t.shape[0] if key.endswith(('.Wh', '.Wu')) else t.shape[1]
                w = torch.randn(t.shape, generator=gen) / math.sqrt(fan_in)
                if '.coord_mlp.' in key and key.endswith('.4.weight'):
                    w = w * 0.1
            elif key.endswith('weight'):
                w = 1.0 + 0.1 * torch.randn(t.shape, generator=gen)
            else:
                w = 0.1 * torch.randn(t.shape, generator=gen)
            t.copy_(w.to(t.device))
    return module


def radius_graph_dense(pos: torch.Tensor, r: float, max_num_neighbors: int = 100):
    """(src = neighbour, dst = centre) pairs of one point cloud with ||.|| < r, dst-major.
    Input-pipeline helper for synthetic pockets (pdbbind_processing.py:245 builds the rr
    graph the same way, on the CPU, once per pocket)."""
    d2 = torch.cdist(pos.double(), pos.double()) ** 2
    mask = d2 < r * r
    mask.fill_diagonal_(False)
    rank = torch.cumsum(mask.long(), dim=1)
    mask &= rank <= max_num_neighbors
    dst, src = torch.nonzero(mask, as_tuple=True)
    return src, dst


def synth_pocket(n_rec: int, seed: int, n_feat: int = 10, density: float = ATOM_DENSITY):
    gen = torch.Generator().manual_seed(seed)
    R = (3.0 * n_rec / (4.0 * math.pi * density)) ** (1.0 / 3.0)
    d = torch.randn(n_rec, 3, generator=gen)
    d = d / d.norm(dim=1, keepdim=True)
    rad = R * torch.rand(n_rec, 1, generator=gen) ** (1.0 / 3.0)
    pos = d * rad
    if n_feat == 10:
        p = torch.tensor([.62, .17, .19, .02] + [0.0] * 6)
    else:
        p = torch.full((n_feat,), 1.0 / n_feat)
    el = torch.multinomial(p, n_rec, replacement=True, generator=gen)
    feat = torch.nn.functional.one_hot(el, n_feat).float()
    return pos.float(), feat


def build_complex_graph(rec_pos, rec_feat, n_keypoints: int, cutoffs: dict,
                        lig_pos: Optional[torch.Tensor] = None, lig_feat: Optional[torch.Tensor] = None,
                        n_lig: int = 0, n_lig_feat: int = 10) -> G.HeteroBatch:
    """Layout of data_processing/pdbbind_processing.py:221-274: rr radius graph, complete
    rec->kp bipartite edges (dst-major), empty kk / kl / ll / lk."""
    n_rec = rec_pos.shape[0]
    src, dst = radius_graph_dense(rec_pos, cutoffs['rr'], max_num_neighbors=100)
    rk_src = torch.arange(n_rec).repeat(n_keypoints)
    rk_dst = torch.arange(n_keypoints).repeat_interleave(n_rec)
    if lig_pos is not None:
        n_lig = lig_pos.shape[0]
    g = G.heterograph({('rec', 'rr', 'rec'): (src, dst), ('rec', 'rk', 'kp'): (rk_src, rk_dst)},
                      num_nodes_dict={'rec': n_rec, 'kp': n_keypoints, 'lig': n_lig})
    g.nodes['rec'].data['x_0'] = rec_pos
    g.nodes['rec'].data['h_0'] = rec_feat
    g.nodes['lig'].data['x_0'] = lig_pos if lig_pos is not None else torch.zeros(n_lig, 3)
    g.nodes['lig'].data['h_0'] = lig_feat if lig_feat is not None else torch.zeros(n_lig, n_lig_feat)
    return g


def synth_complexes(n_rec: Sequence[int], n_lig: Sequence[int], n_keypoints: int, cutoffs: dict,
                    seed: int = 1234, n_rec_feat: int = 10, n_lig_feat: int = 10,
                    density: float = ATOM_DENSITY) -> List[G.HeteroBatch]:
    """One graph per complex: synthetic pocket + a ligand at the t = T state
    (x_0, h_0 ~ N(0, I), ligand COM removed; ligand_diffuser.py:366-370)."""
    out = []
    for i, (nr, nl) in enumerate(zip(n_rec, n_lig)):
        pos, feat = synth_pocket(int(nr), seed + i, n_rec_feat, density)
        gen = torch.Generator().manual_seed(10_000_019 * (seed + i) + 7)
        lx = torch.randn(int(nl), 3, generator=gen)
        lx = lx - lx.mean(0, keepdim=True)
        lh = torch.randn(int(nl), n_lig_feat, generator=gen)
        out.append(build_complex_graph(pos, feat, n_keypoints, cutoffs, lx, lh))
    return out
