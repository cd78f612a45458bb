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


def per_complex_rel_err(a: torch.Tensor, b: torch.Tensor, counts) -> float:
    """max over complexes of (max |a - b| / max |b|) taken INSIDE each complex: a small complex (a single-atom ligand, a
    quiet pocket) is judged on its own scale instead of hiding under the largest entry of the batch."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    worst, off = 0.0, 0
    for n in [int(c) for c in counts]:
        if n:
            ref = b[off:off + n].abs().max().clamp(min=1e-30)
            worst = max(worst, float((a[off:off + n] - b[off:off + n]).abs().max() / ref))
        off += n
    assert off == a.shape[0], (off, a.shape)
    return worst


def elementwise_excess(a: torch.Tensor, b: torch.Tensor, rtol: float = 1e-4, atol_rel: float = 1e-6) -> float:
    """allclose(a, b, rtol, atol = atol_rel * max |b|) as a number: max of |a - b| / (atol + rtol |b|); <= 1 passes."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    atol = atol_rel * float(b.abs().max().clamp(min=1e-30))
    return float(((a - b).abs() / (atol + rtol * b.abs())).max())


def assert_parity(got: torch.Tensor, ref: torch.Tensor, counts=None, tol: float = 1e-4, what: str = '', atol_rel: float = 1e-6):
    """The three parity measures of the denoiser tests: whole-tensor relative error (BASELINE.json's '1e-4 rel fp32'),
    the same per complex, and an elementwise allclose(rtol = tol, atol = atol_rel * max |ref|)."""
    e = rel_err(got, ref)
    assert e < tol, f'{what}: rel err {e:.3e} >= {tol}'
    if counts is not None:
        pc = per_complex_rel_err(got, ref, counts)
        assert pc < tol, f'{what}: per-complex rel err {pc:.3e} >= {tol}'
    ex = elementwise_excess(got, ref, rtol=tol, atol_rel=atol_rel)
    assert ex <= 1.0, f'{what}: allclose(rtol={tol}, atol={atol_rel} max|ref|) violated by a factor {ex:.2f}'


def run_threaded_world(world: int, fn, timeout: float = 600.0):
    """`fn(rank)` on `world` ranks that are THREADS of this process, joined by torch's in-process 'threaded' process group
    (torch.testing._internal.distributed.multi_threaded_pg: every collective of torch.distributed works, each thread sees its
    own rank / world size).  This is how more ranks than the GPU box admits GPU-holding processes (6, the test runner
    included) are rehearsed on one card.  Returns [fn(0), ..., fn(world - 1)]; re-raises the first rank failure."""
    import threading
    import torch.distributed as dist
    from torch.testing._internal.distributed import multi_threaded_pg as mt
    mt._install_threaded_pg()
    torch._C._distributed_c10d._set_thread_isolation_mode(True)
    store = dist.HashStore()
    results, errors = [None] * world, []

    def worker(rank):
        try:
            dist.init_process_group(backend='threaded', rank=rank, world_size=world, store=store)
            try:
                results[rank] = fn(rank)
                dist.barrier()
            finally:
                dist.destroy_process_group()
        except BaseException as ex:                                   # noqa: B036 -- wake the other ranks, report below
            errors.append((rank, ex))
            mt.ProcessLocalGroup.exception_handle(ex)

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(world)]
    try:
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout)
            assert not th.is_alive(), 'a thread rank did not finish'
    finally:
        torch._C._distributed_c10d._set_thread_isolation_mode(False)
        mt.ProcessLocalGroup.reset()
        mt._uninstall_threaded_pg()
    if errors:
        raise errors[0][1]
    return results
