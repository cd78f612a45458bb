"""Output side of sampling: sampled ligands -> element symbols and XYZ text, computed on the GPU.

Mirrors the tensor -> text part of the reference's `write_sampled_ligands` (sample.py:66-90) and `write_xyz_file`
(utils.py:11-21; called per ligand from analysis/molecule_builder.py:47-48).  Bond perception and SDF writing
(openbabel, rdkit) are the caller's: they take the XYZ blocks returned here.  There is no CPU implementation —
tensors must live on the GPU and the HIP library must be present.
"""
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import torch

from . import hip


def sampled_ligands_xyz(lig_pos: List[torch.Tensor], lig_feat: List[torch.Tensor],
                        lig_elements: Sequence[str]) -> List[Tuple[List[str], str]]:
    """For every sampled ligand (the lists `sample_given_pocket` returns, still on the GPU): its element symbols
    (argmax over the feature columns -> `lig_elements`, sample.py:77-79) and its XYZ file contents (utils.py:11-21).
    One batched launch sequence for all ligands."""
    if len(lig_pos) != len(lig_feat):
        raise ValueError('lig_pos and lig_feat must have one entry per ligand')
    if not lig_pos:
        return []
    sizes = [int(p.shape[0]) for p in lig_pos]
    pos = torch.cat([p.reshape(-1, 3) for p in lig_pos])
    feat = torch.cat(list(lig_feat))
    ptr = torch.tensor([0] + sizes, dtype=torch.int64).cumsum(0).to(torch.int32).to(pos.device)
    elem, text, tptr = hip.xyz_emit(pos, feat, ptr, list(lig_elements))
    elem = elem.cpu().tolist()
    out, a = [], 0
    for b, n in enumerate(sizes):
        out.append(([lig_elements[i] for i in elem[a:a + n]], text[tptr[b]:tptr[b + 1]].decode('ascii')))
        a += n
    return out


def write_xyz_file(coords: torch.Tensor, atom_types: Sequence[str], filename: Optional[Path] = None):
    """utils.py:11-21 for one ligand: `coords` [n,3] on the GPU, `atom_types` the element symbol of every atom.
    Returns the file contents when `filename` is None, else writes them."""
    symbols = sorted(set(atom_types))
    if len(coords) != len(atom_types):
        raise AssertionError('len(coords) != len(atom_types)')
    if not symbols:
        out = '0\n\n'
    else:
        idx = torch.tensor([symbols.index(a) for a in atom_types], device=coords.device)
        feat = torch.nn.functional.one_hot(idx, len(symbols)).float()
        out = sampled_ligands_xyz([coords], [feat], symbols)[0][1]
    if filename is None:
        return out
    with open(filename, 'w') as f:
        f.write(out)
