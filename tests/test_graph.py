"""Graph container (the DGL stand-in) and the fixed receptor encoder: host plumbing semantics."""
import torch

from keypoint_diffusion_amd import graph as G
from keypoint_diffusion_amd import synth

from . import util


def test_batch_unbatch_roundtrip_and_offsets():
    gs = synth.synth_complexes([12, 7, 20], [4, 6, 3], 5, util.CUTOFFS_ALL_ATOM, seed=3)
    b = G.batch(gs)
    assert b.batch_size == 3 and b.num_nodes('rec') == 39 and b.num_nodes('lig') == 13 and b.num_nodes('kp') == 15
    assert b.batch_num_nodes('rec').tolist() == [12, 7, 20]
    assert b.batch_num_edges('rk').tolist() == [60, 35, 100]
    s, d = b.edges(etype='rr')
    bi = G.get_batch_idxs(b)['rec']
    assert torch.equal(bi[s], bi[d])                       # no edge crosses a complex
    parts = G.unbatch(b)
    for p, g in zip(parts, gs):
        assert torch.equal(p.nodes['rec'].data['x_0'], g.nodes['rec'].data['x_0'])
        ps, pd = p.edges(etype='rr')
        gs_, gd_ = g.edges(etype='rr')
        assert torch.equal(ps, gs_) and torch.equal(pd, gd_)


def test_readout_and_batch_idxs():
    b = G.batch(synth.synth_complexes([5, 9], [3, 4], 2, util.CUTOFFS_ALL_ATOM))
    com = G.readout_nodes(b, 'x_0', op='mean', ntype='lig')
    x = b.nodes['lig'].data['x_0']
    assert torch.allclose(com[0], x[:3].mean(0), atol=1e-6) and torch.allclose(com[1], x[3:].mean(0), atol=1e-6)
    assert G.get_batch_idxs(b)['lig'].tolist() == [0, 0, 0, 1, 1, 1, 1]


def test_fixed_encoder_moves_receptor_to_keypoints():
    b = G.batch(synth.synth_complexes([30, 18], [6, 5], 4, util.CUTOFFS_ALL_ATOM))
    rr = b.edges(etype='rr')
    x = b.nodes['rec'].data['x_0'].clone()
    out = util.fixed_encode(b, n_vec=16)
    assert out.num_nodes('rec') == 0 and out.num_nodes('kp') == 48
    assert out.batch_num_nodes('kp').tolist() == [30, 18] and out.batch_num_nodes('rec').tolist() == [0, 0]
    assert torch.equal(out.nodes['kp'].data['x_0'], x)
    assert out.nodes['kp'].data['v_0'].shape == (48, 16, 3)
    ks, kd = out.edges(etype='kk')
    assert torch.equal(ks, rr[0]) and torch.equal(kd, rr[1])
    assert out.num_edges('rr') == 0 and out.num_edges('rk') == 0
    assert out.batch_num_edges('kk').sum() == ks.numel()


def test_copy_graph_resizes_ligand():
    g = synth.synth_complexes([10], [4], 3, util.CUTOFFS_ALL_ATOM)[0]
    cps = G.copy_graph(g, 3, lig_atoms_per_copy=torch.tensor([2, 7, 5]))
    assert [c.num_nodes('lig') for c in cps] == [2, 7, 5]
    assert all(c.num_nodes('rec') == 10 for c in cps)
    assert cps[1].nodes['lig'].data['h_0'].shape == (7, 10)
    b = G.batch(cps)
    assert b.batch_num_nodes('lig').tolist() == [2, 7, 5]


def test_add_remove_edges_and_local_scope():
    g = synth.synth_complexes([6], [3], 2, util.CUTOFFS_ALL_ATOM)[0]
    g.add_edges(torch.tensor([0, 1]), torch.tensor([2, 2]), etype='ll')
    assert g.num_edges('ll') == 2
    g.remove_edges(g.edges(form='eid', etype='ll'), etype='ll')
    assert g.num_edges('ll') == 0
    with g.local_scope():
        g.nodes['lig'].data['tmp'] = torch.zeros(3)
    assert 'tmp' not in g.nodes['lig'].data
