"""The library's own fp32 GEMM (csrc/sgemm.hip: every dense product of the training engines) against a float64 product of the same
operands: all four operand forms, sizes that are no multiple of any tile, odd leading dimensions and misaligned views (the 513-float rows
of the EGNN first Linears), alpha / beta, split-K through the weight-gradient shape, bitwise repeatability."""
import pytest
import torch

from keypoint_diffusion_amd import hip

pytestmark = pytest.mark.gpu

SHAPES = [  # M, N, K
    (1, 1, 1), (5, 3, 2), (64, 257, 10), (300, 16, 16), (1000, 33, 47), (129, 130, 131), (4099, 256, 256), (20800, 257, 514),
    (37, 600, 5), (256, 256, 4096), (2, 700, 1), (70000, 16, 16), (257, 100, 50001), (16, 16, 30000), (16, 16, 1200001), (32, 5, 9000), (1, 32, 8192),
    (257, 257, 1600), (257, 257, 40000), (129, 65, 77), (385, 129, 300), (1000, 257, 257),        # one row / column past the tiles: the fringe riders
]


def _operand(rows, cols, ld_extra, offset, gen, dev):
    """[rows, cols] view with row stride cols + ld_extra starting `offset` floats into its storage."""
    buf = torch.randn(offset + rows * (cols + ld_extra) + 8, generator=gen).to(dev)
    return buf[offset:offset + rows * (cols + ld_extra)].view(rows, cols + ld_extra)[:, :cols]


@pytest.mark.parametrize('trans_a,trans_b', [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize('layout', ['aligned', 'odd_ld', 'offset'])
def test_sgemm_matches_float64_product(trans_a, trans_b, layout):
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(11)
    extra, off = {'aligned': (0, 0), 'odd_ld': (1, 0), 'offset': (4, 3)}[layout]
    for M, N, K in SHAPES:
        a = _operand(K if trans_a else M, M if trans_a else K, extra, off, gen, dev)
        b = _operand(N if trans_b else K, K if trans_b else N, extra, off, gen, dev)
        c0 = _operand(M, N, extra, off, gen, dev)
        ref = (a.double().T if trans_a else a.double()) @ (b.double().T if trans_b else b.double())
        ws = torch.full((1 << 23,), float('nan'), device=dev) if K >= 4096 else None     # split-K scratch: contents irrelevant
        out = hip.sgemm(a, b, trans_a, trans_b, workspace=ws)
        scale = ref.abs().max().clamp_min(1.0)
        tol = 1e-6 + 4e-7 * K ** 0.5                      # fp32 accumulation over K terms
        assert ((out.double() - ref).abs().max() / scale) < tol, (M, N, K)
        c = c0.clone() if layout == 'aligned' else c0
        before = c.double().clone()
        hip.sgemm(a, b, trans_a, trans_b, alpha=0.5, beta=2.0, out=c, workspace=ws)
        ref2 = 0.5 * ref + 2.0 * before
        assert ((c.double() - ref2).abs().max() / ref2.abs().max().clamp_min(1.0)) < 2 * tol, (M, N, K)


def test_sgemm_leaves_the_padding_of_a_strided_output_alone():
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(3)
    a, b = torch.randn(300, 40, generator=gen).to(dev), torch.randn(40, 70, generator=gen).to(dev)
    wide = torch.full((300, 96), 7.0, device=dev)
    hip.sgemm(a, b, out=wide[:, 5:75])
    assert torch.all(wide[:, :5] == 7.0) and torch.all(wide[:, 75:] == 7.0)
    assert torch.allclose(wide[:, 5:75], a @ b, atol=1e-4)


def test_sgemm_is_bitwise_repeatable_and_independent_of_other_rows():
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(5)
    a, b = torch.randn(5000, 300, generator=gen).to(dev), torch.randn(200, 300, generator=gen).to(dev)
    o1, o2 = hip.sgemm(a, b, trans_b=True), hip.sgemm(a, b, trans_b=True)
    assert torch.equal(o1, o2)
    part = hip.sgemm(a[1000:1200], b, trans_b=True)          # other tile position, other tile shape: same bits per element
    assert torch.equal(part, o1[1000:1200])


@pytest.mark.parametrize('M,N,K', [(256, 256, 50000), (257, 300, 777), (16, 16, 200000), (16, 256, 90000), (100, 7, 33), (130, 64, 5000),
                                   (257, 257, 30000), (257, 257, 100), (129, 193, 7000)])
@pytest.mark.parametrize('layout', ['aligned', 'odd_ld'])
def test_column_sums_ride_along_with_a_weight_gradient(M, N, K, layout):
    """colsum += the column sums of A in the pass that computes A^T B (bias gradient with the weight gradient), with and without the
    split along K, through the narrow-gradient kernel too; bitwise repeatable."""
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(2)
    extra = 0 if layout == 'aligned' else 1
    a, b = _operand(K, M, extra, 0, gen, dev), _operand(K, N, extra, 0, gen, dev)
    ws = torch.full((1 << 23,), float('nan'), device=dev)
    for workspace in (None, ws):
        cs = torch.full((M,), 3.0, device=dev)
        out = hip.sgemm(a, b, True, False, workspace=workspace, colsum=cs)
        ref_cs = 3.0 + a.double().sum(0)
        tol = 1e-6 + 4e-7 * K ** 0.5
        assert ((cs.double() - ref_cs).abs().max() / ref_cs.abs().max().clamp_min(1.0)) < tol
        ref = a.double().T @ b.double()
        assert ((out.double() - ref).abs().max() / ref.abs().max().clamp_min(1.0)) < tol
        cs2 = torch.full((M,), 3.0, device=dev)
        out2 = hip.sgemm(a, b, True, False, workspace=workspace, colsum=cs2)
        assert torch.equal(cs, cs2) and torch.equal(out, out2)


@pytest.mark.parametrize('M,N,K', [(257, 257, 65536), (257, 257, 70001), (257, 256, 66017), (256, 257, 131075), (256, 256, 166003),
                                   (257, 257, 65553), (257, 257, 65535)])
def test_full_output_weight_gradient_kernel(M, N, K):
    """Edge-sized weight gradients (K >= 65 536 rows of 16-B aligned activations, 256 (+ 1) x 256 (+ 1) outputs: dW2 = dpre2^T a1 of the
    EGNN trainer with its 264-float rows, the 256 x 256 scalar blocks of the GVP trainers) take k_sgemm_tn256 -- the whole output in one
    workgroup, every operand row fetched once (dispatch: sgemm.hip, `K >= 65536`).  Every 257-wide case here sits above that threshold, so
    the kernel's fringe riders (row / column 256, the corner, the fringe row's column sum, the x_part layout) are what runs: 65 536 = whole
    16-row slabs, 70 001 / 66 017 / 131 075 = K tails that are no multiple of 16, 65 553 = a last split-K slice of ONE row (no full slab
    in it); 65 535 pins the other side of the dispatch (the tiled kernel, same checks).  Against a float64 product, with the column sums
    of A, accumulation into a strided C, and bit for bit twice."""
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(17)
    a = torch.randn(K, 264, generator=gen).to(dev)[:, :M]            # the trainers' row stride
    b = torch.randn(K, 264, generator=gen).to(dev)[:, :N]
    ws = torch.full((260 * (256 * 256 + 1024),), float('nan'), device=dev)
    ref = a.double().T @ b.double()
    tol = 1e-6 + 4e-7 * K ** 0.5
    outs = []
    for _ in range(2):
        cs = torch.full((M,), 3.0, device=dev)
        c = torch.full((M, N + 3), 2.0, device=dev)                  # strided output, accumulated into (beta = 1 as grad_gemm does)
        hip.sgemm(a, b, True, False, beta=1.0, out=c[:, :N], workspace=ws, colsum=cs)
        assert torch.all(c[:, N:] == 2.0)
        assert ((c[:, :N].double() - 2.0 - ref).abs().max() / ref.abs().max()) < tol
        ref_cs = a.double().sum(0)
        assert ((cs.double() - 3.0 - ref_cs).abs().max() / ref_cs.abs().max().clamp_min(1.0)) < tol
        outs.append((c.clone(), cs.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
