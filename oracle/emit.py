"""CPU restatement of the output side of sampling (test infrastructure only — never imported by the product path).

Follows sample.py:66-90 (`write_sampled_ligands`: `torch.argmax(lig_feat, dim=1)`, index -> element through the
dataset's reverse map = position in `lig_elements`) and utils.py:11-21 (`write_xyz_file`) as
analysis/molecule_builder.py:47-48 calls it for every ligand.  Pinned by tests/golden/xyz.npz, which holds the text
the reference's own `write_xyz_file` printed for the same coordinates.
"""
from typing import List, Sequence, Tuple

import torch


def write_xyz_file(coords: torch.Tensor, atom_types: Sequence[str]) -> str:
    """utils.py:11-21 with filename=None: the file contents."""
    out = f"{len(coords)}\n\n"
    assert len(coords) == len(atom_types)
    for i in range(len(coords)):
        out += f"{atom_types[i]} {coords[i, 0]:.3f} {coords[i, 1]:.3f} {coords[i, 2]:.3f}\n"
    return out


def decode_elements(lig_feat: torch.Tensor, lig_elements: Sequence[str]) -> Tuple[List[int], List[str]]:
    """sample.py:77-79: argmax over the feature columns, then dataset.lig_atom_idx_to_element
    (data_processing/crossdocked/dataset.py:147-149)."""
    idxs = torch.argmax(lig_feat, dim=1).tolist()
    return idxs, [lig_elements[i] for i in idxs]


def sampled_ligands_xyz(lig_pos: List[torch.Tensor], lig_feat: List[torch.Tensor], lig_elements: Sequence[str]):
    """Per ligand: (element indices, XYZ block) in the order write_sampled_ligands walks them."""
    out = []
    for pos, feat in zip(lig_pos, lig_feat):
        idxs, els = decode_elements(feat.detach().cpu(), lig_elements)
        out.append((idxs, write_xyz_file(pos.detach().cpu(), els)))
    return out
