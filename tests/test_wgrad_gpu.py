"""The batched weight-gradient kernels of the training engines (csrc/sgemm.hip: k_wgrad_tnx in its three forms, k_sgemm_tn256_batch) against plain
fp32 products: several products of different depth per launch, narrow riders, accumulation into the outputs, K tails that are not multiples
of the 16-row slab, a K below one slab, and operand rows wider than the 256 columns used (the EGNN trainer's 264-float rows)."""
import pytest
import torch

from keypoint_diffusion_amd import hip

pytestmark = pytest.mark.gpu
TOL = 2e-5          # relative to the largest entry of each output (fp32 sums over K in another order)


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30))


def _rand(gen, *shape):
    return torch.randn(*shape, generator=gen).cuda()


@pytest.mark.parametrize('Ks', [[1000, 37, 5003, 16], [70001, 300], [9], [4096] * 8])
def test_gvp_pairs_with_riders(cuda, Ks):
    gen = torch.Generator().manual_seed(len(Ks))
    ws = torch.empty(12_000_000, device='cuda')
    items, want = [], []
    for K in Ks:
        A, B, B2, A2 = _rand(gen, K, 256), _rand(gen, K, 256), _rand(gen, K, 17)[:, :16], _rand(gen, K, 16)
        ws_g, wg_g, b_g, bg_g = _rand(gen, 256, 272), _rand(gen, 16, 256), _rand(gen, 256), _rand(gen, 16)
        want.append((ws_g[:, :256] + A.T @ B, ws_g[:, 256:] + A.T @ B2, b_g + A.sum(0), wg_g + A2.T @ B, bg_g + A2.sum(0)))
        items.append(dict(A=A, B=B, C=ws_g[:, :256], B2=B2, nb2=16, Cx1=ws_g[:, 256:], colsum=b_g, A2=A2, na2=16, Cx2=wg_g, colsum2=bg_g, _out=(ws_g, wg_g, b_g, bg_g)))
    hip.wgrad_batch(0, items, ws)
    torch.cuda.synchronize()
    for it, (c, x1, cs, x2, cs2) in zip(items, want):
        ws_g, wg_g, b_g, bg_g = it['_out']
        assert _rel(ws_g[:, :256], c) < TOL and _rel(ws_g[:, 256:], x1) < TOL and _rel(b_g, cs) < TOL and _rel(wg_g, x2) < TOL and _rel(bg_g, cs2) < TOL


@pytest.mark.parametrize('Ks', [[777, 20000], [33]])
def test_rider_only_products(cuda, Ks):
    gen = torch.Generator().manual_seed(7)
    ws = torch.empty(4_000_000, device='cuda')
    items, want = [], []
    for K in Ks:
        A, B, rbf, sh, dg = _rand(gen, K, 256), _rand(gen, K, 256), _rand(gen, K, 16), _rand(gen, K, 17), _rand(gen, K, 16)
        w0, b0, wg, bg = _rand(gen, 256, 289), _rand(gen, 256), _rand(gen, 16, 256), _rand(gen, 16)
        want.append((w0[:, 256:272] + A.T @ rbf, w0[:, 272:] + A.T @ sh, b0 + A.sum(0), wg + dg.T @ B, bg + dg.sum(0), w0[:, :256].clone()))
        items.append(dict(A=A, B=B, B2=rbf, nb2=16, Cx1=w0[:, 256:272], colsum=b0, B3=sh, nb3=17, Cx3=w0[:, 272:], A2=dg, na2=16, Cx2=wg, colsum2=bg,
                          _out=(w0, b0, wg, bg)))
    hip.wgrad_batch(0, items, ws)
    torch.cuda.synchronize()
    for it, (x1, x3, cs, x2, cs2, untouched) in zip(items, want):
        w0, b0, wg, bg = it['_out']
        assert _rel(w0[:, 256:272], x1) < TOL and _rel(w0[:, 272:], x3) < TOL and _rel(b0, cs) < TOL and _rel(wg, x2) < TOL and _rel(bg, cs2) < TOL
        assert torch.equal(w0[:, :256], untouched)


@pytest.mark.parametrize('Ks', [[5000, 123, 70000], [40] * 8])
def test_egnn_257_products(cuda, Ks):
    gen = torch.Generator().manual_seed(11)
    ws = torch.empty(17_500_000, device='cuda')
    items, want = [], []
    for i, K in enumerate(Ks):
        A, B = _rand(gen, K, 264), _rand(gen, K, 264)          # rows of 264 floats, 257 used (the EGNN trainer's activation layout)
        g, cs = _rand(gen, 257, 257), _rand(gen, 257)
        want.append((g + A[:, :257].T @ B[:, :257], cs + A[:, :257].sum(0)))
        items.append(dict(A=A, B=B, C=g, colsum=cs if i % 2 == 0 else None, _cs=cs))
    hip.wgrad_batch(1, items, ws)
    torch.cuda.synchronize()
    for i, (it, (c, cs)) in enumerate(zip(items, want)):
        assert _rel(it['C'], c) < TOL
        if i % 2 == 0:
            assert _rel(it['_cs'], cs) < TOL
