"""CPU restatement of the input pipeline (test infrastructure only — never imported by the product path).

Follows data_processing/pdbbind_processing.py:221-274 (`build_initial_complex_graph`) and
data_processing/crossdocked/dataset.py:62-76, 128-131, 187-194 (`__getitem__`, `collate_fn`).  The radius graph is the
definitional O(N^2) one of oracle/graph_ops.py (torch_cluster is absent from the reference tree and from this image:
parity at that boundary is by definition); everything else is index arithmetic.
"""
from typing import Dict, List, Sequence

import torch

from . import graph_ops as G


def build_initial_complex_graph(rec_pos, rec_feat, res_idx, n_keypoints: int, cutoffs: dict, lig_pos=None, lig_feat=None) -> Dict:
    n_rec = rec_pos.shape[0]
    n = torch.tensor([n_rec])
    src, dst = G.radius_graph(rec_pos, cutoffs['rr'], n, max_num_neighbors=100)                        # :245
    same_res = (res_idx[src] == res_idx[dst]).view(-1, 1)                                             # :248, :272
    rk_src = torch.arange(n_rec).repeat(n_keypoints)                                                  # :251-252
    rk_dst = torch.arange(n_keypoints).repeat_interleave(n_rec)
    return dict(n_rec=n_rec, n_kp=n_keypoints, n_lig=0 if lig_pos is None else lig_pos.shape[0], rr=(src, dst), same_res=same_res,
                rk=(rk_src, rk_dst), rec_x=rec_pos, rec_h=rec_feat, lig_x=lig_pos, lig_h=lig_feat)


def get_item(data: dict, i: int, n_keypoints: int, cutoffs: dict):
    """dataset.py:62-76, 128-131."""
    ls, le = data['lig_segments'][i:i + 2]
    rs, re = data['rec_segments'][i:i + 2]
    ps, pe = data['ip_segments'][i:i + 2]
    g = build_initial_complex_graph(data['rec_pos'][rs:re], data['rec_feat'][rs:re].float(), data['rec_res_idx'][rs:re], n_keypoints,
                                    cutoffs, data['lig_pos'][ls:le], data['lig_feat'][ls:le].float())
    return g, data['interface_points'][ps:pe]


def collate(items: Sequence) -> Dict:
    """dgl.batch of the per-complex graphs: concatenation with node offsets (dataset.py:187-194)."""
    gs = [g for g, _ in items]
    off_r, off_k = 0, 0
    out = dict(rr_src=[], rr_dst=[], rk_src=[], rk_dst=[], same_res=[], rr_counts=[], rk_counts=[])
    for g in gs:
        out['rr_src'].append(g['rr'][0] + off_r)
        out['rr_dst'].append(g['rr'][1] + off_r)
        out['rk_src'].append(g['rk'][0] + off_r)
        out['rk_dst'].append(g['rk'][1] + off_k)
        out['same_res'].append(g['same_res'])
        out['rr_counts'].append(g['rr'][0].numel())
        out['rk_counts'].append(g['rk'][0].numel())
        off_r += g['n_rec']
        off_k += g['n_kp']
    res = {k: (torch.cat(v) if isinstance(v[0], torch.Tensor) else v) for k, v in out.items()}
    for k, key in (('rec_x', 'rec_x'), ('rec_h', 'rec_h'), ('lig_x', 'lig_x'), ('lig_h', 'lig_h')):
        res[k] = torch.cat([g[key] for g in gs])
    res['n_rec'] = [g['n_rec'] for g in gs]
    res['n_lig'] = [g['n_lig'] for g in gs]
    return res
