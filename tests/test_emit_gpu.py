"""Output side: element decode + XYZ text on the GPU, byte-exact against the reference's text."""
import os

import numpy as np
import pytest
import torch

from oracle import emit as oemit

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'xyz.npz')
ELEMENTS = ['C', 'N', 'O', 'S', 'P', 'F', 'Cl', 'Br', 'I', 'B']


def _split(d):
    pos, feat = torch.from_numpy(d['pos']), torch.from_numpy(d['feat'])
    out, a = [], 0
    for n in d['sizes'].tolist():
        out.append((pos[a:a + n], feat[a:a + n]))
        a += n
    return out


def test_oracle_matches_reference_text():
    """The restatement prints what the reference's write_xyz_file printed (fixture generated from it)."""
    d = np.load(GOLD)
    ligs = _split(d)
    res = oemit.sampled_ligands_xyz([p for p, _ in ligs], [f for _, f in ligs], ELEMENTS)
    text = ''.join(t for _, t in res).encode()
    assert text == bytes(d['text'])
    assert [i for idxs, _ in res for i in idxs] == d['elem'].tolist()


@pytest.mark.gpu
def test_xyz_emit_matches_golden_bytes():
    from keypoint_diffusion_amd import utils as kutils
    d = np.load(GOLD)
    ligs = _split(d)
    res = kutils.sampled_ligands_xyz([p.cuda() for p, _ in ligs], [f.cuda() for _, f in ligs], ELEMENTS)
    tptr = d['text_ptr'].tolist()
    gold = bytes(d['text'])
    elem = d['elem'].tolist()
    a = 0
    for b, (els, text) in enumerate(res):
        assert text.encode() == gold[tptr[b]:tptr[b + 1]], f'ligand {b}'
        n = len(els)
        assert els == [ELEMENTS[i] for i in elem[a:a + n]]
        a += n


@pytest.mark.gpu
def test_xyz_emit_random_against_oracle():
    from keypoint_diffusion_amd import utils as kutils
    g = torch.Generator().manual_seed(5)
    sizes = torch.randint(1, 61, (300,), generator=g).tolist() + [700]
    scale = torch.tensor([1e-3, 1.0, 25.0, 4000.0])
    lig_pos = [torch.randn(n, 3, generator=g) * scale[torch.randint(0, 4, (n, 1), generator=g)] for n in sizes]
    # thousandth-grid points +- half a step: every value sits on or next to a rounding boundary
    lig_pos[0] = (torch.randint(-5000, 5000, lig_pos[0].shape, generator=g).float() + 0.5) / 1000
    lig_feat = [torch.randn(n, 10, generator=g) for n in sizes]
    res = kutils.sampled_ligands_xyz([p.cuda() for p in lig_pos], [f.cuda() for f in lig_feat], ELEMENTS)
    ref = oemit.sampled_ligands_xyz(lig_pos, lig_feat, ELEMENTS)
    for b, ((els, text), (idxs, rtext)) in enumerate(zip(res, ref)):
        assert text == rtext, f'ligand {b}'
        assert els == [ELEMENTS[i] for i in idxs]


@pytest.mark.gpu
def test_write_xyz_file_mirror_and_errors(tmp_path):
    from keypoint_diffusion_amd import hip, utils as kutils
    coords = torch.tensor([[0.0625, -1.5, 3.14159], [10.0, 0.0005, -0.0]])
    types = ['Cl', 'C']
    assert kutils.write_xyz_file(coords.cuda(), types) == oemit.write_xyz_file(coords, types)
    kutils.write_xyz_file(coords.cuda(), types, tmp_path / 'a.xyz')
    assert (tmp_path / 'a.xyz').read_text() == oemit.write_xyz_file(coords, types)
    assert kutils.write_xyz_file(torch.zeros(0, 3).cuda(), []) == '0\n\n'
    with pytest.raises(hip.KpdError):          # CPU tensors: no CPU implementation
        kutils.sampled_ligands_xyz([coords], [torch.zeros(2, 10)], ELEMENTS)
    with pytest.raises(hip.KpdError):          # beyond the printable range
        kutils.sampled_ligands_xyz([torch.full((1, 3), 1e17).cuda()], [torch.zeros(1, 10).cuda()], ELEMENTS)
