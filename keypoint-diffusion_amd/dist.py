"""Multi-GPU layout of sampling: complexes are independent (no edge crosses a complex), so the
flat list of (pocket, replicate) complexes is sharded across ranks with no data-path collective;
the only exchange is one all-gather of the sampled ligand tensors after the last reverse step
(RCCL over xGMI on the GPU box, gloo in the CPU tests).  The payload is tiny (B=64 x 25 atoms x
13 floats = 83 KB per rank) and latency bound, so it is gathered as one padded block.
"""
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import graph as G


def shard_complexes(costs: Sequence[float], world: int) -> List[range]:
    """Contiguous, cost-balanced partition of complexes over ranks.  `costs[i]` ~ edges of complex i
    (use n_rec for ragged pockets: E ~ 600 + 22.7 n_rec).  Contiguity keeps the
    gathered order equal to the input order."""
    n = len(costs)
    total = float(sum(costs))
    bounds, acc, r = [0], 0.0, 1
    for i, c in enumerate(costs):
        acc += c
        while r < world and acc >= total * r / world - 1e-9 and len(bounds) < world:
            # close rank r-1 after complex i, but leave at least one complex per remaining rank
            cut = min(i + 1, n - (world - r))
            cut = max(cut, bounds[-1] + (1 if n >= world else 0))
            bounds.append(cut)
            r += 1
    while len(bounds) < world:
        bounds.append(n)
    bounds.append(n)
    return [range(bounds[i], max(bounds[i], bounds[i + 1])) for i in range(world)]


def _group_device(like: torch.device, group=None) -> torch.device:
    """RCCL moves device tensors; gloo (CPU tests, one-GPU rehearsals) wants host tensors."""
    return like if dist.get_backend(group) == 'nccl' else torch.device('cpu')


def gather_ligand_lists(pos: Sequence[torch.Tensor], feat: Sequence[torch.Tensor], group=None,
                        device: Optional[torch.device] = None) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """Every rank passes the ligands of ITS complexes (`pos[i]` [n_i, 3], `feat[i]` [n_i, F], any device) and receives those
    of ALL ranks, rank-major, as host tensors -- with contiguous shards (`shard_complexes`) that is the input order.
    Two collectives (sizes, then one padded block per rank); packing and unpacking cost no per-complex device work: the block
    is filled by one indexed assignment, copied to the host once, and the results are views into that copy.  A rank may hold
    no complexes at all (more ranks than complexes)."""
    world = dist.get_world_size(group)
    like = device if device is not None else (pos[0].device if len(pos) else torch.device('cpu'))
    dev = _group_device(like, group)
    B = len(pos)
    counts = torch.tensor([p.shape[0] for p in pos], dtype=torch.long, device=dev)
    F = feat[0].shape[1] if B else 0
    # 1) how many complexes / atoms / feature columns everybody has
    meta = torch.tensor([B, int(counts.max()) if B else 0, F], device=dev, dtype=torch.int32)
    metas = torch.empty(world * 3, device=dev, dtype=torch.int32)
    dist.all_gather_into_tensor(metas, meta, group=group)
    metas = metas.view(world, 3).cpu()
    max_B, max_n, F = int(metas[:, 0].max()), int(metas[:, 1].max()), int(metas[:, 2].max())
    if max_B == 0:
        return [], []
    # 2) one padded block per rank: [max_B, 1 + max_n * (3 + F)], column 0 = atom count, then atoms x (x, h)
    W = 3 + F
    block = torch.zeros(max_B, 1 + max_n * W, device=dev, dtype=torch.float32)
    if B:
        x = torch.cat([p.to(dev).float() for p in pos], dim=0)
        h = torch.cat([f.to(dev).float() for f in feat], dim=0)
        block[:B, 0] = counts.float()
        cid = torch.repeat_interleave(torch.arange(B, device=dev), counts)
        start = torch.cumsum(counts, 0) - counts
        local = torch.arange(x.shape[0], device=dev) - start[cid]
        body = block[:, 1:].view(max_B, max_n, W)
        body[cid, local] = torch.cat([x, h], dim=1)
    blocks = torch.empty(world, max_B, 1 + max_n * W, device=dev, dtype=torch.float32)
    dist.all_gather_into_tensor(blocks.view(world * max_B, -1), block, group=group)
    blocks = blocks.cpu()
    out_pos, out_feat = [], []
    for r in range(world):
        body = blocks[r, :, 1:].view(max_B, max_n, W)
        ns = blocks[r, :int(metas[r, 0]), 0].long().tolist()
        for b, n in enumerate(ns):
            out_pos.append(body[b, :n, :3])
            out_feat.append(body[b, :n, 3:])
    return out_pos, out_feat


def all_gather_ligands(g: G.HeteroBatch, group=None) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """Every rank returns the ligand positions / features of ALL ranks' complexes, rank-major (one batch per rank)."""
    counts = g.batch_num_nodes('lig').long().tolist()
    x, h = g.nodes['lig'].data['x_0'], g.nodes['lig'].data['h_0']
    return gather_ligand_lists(list(torch.split(x, counts)), list(torch.split(h, counts)), group=group, device=g.device)


def sharding_active(group=None) -> bool:
    """True under ANY initialised process group, a single-rank one included: a one-rank job then takes exactly the code path
    (shard -> per-complex Philox noise -> gather) and draws exactly the noise an eight-rank job does, so its ligands are
    the eight-rank job's ligands, and the RCCL branch can be tested on a one-GPU box (tests/test_nccl_gpu.py)."""
    return dist.is_available() and dist.is_initialized()


def sharded_map(costs: Sequence[float], fn, group=None, device: Optional[torch.device] = None):
    """The multi-GPU form of "for every complex: sample": `fn(range)` produces (positions, features) lists for the complexes
    of this rank's contiguous, cost-balanced shard; one gather at the end hands every rank all results in input order.
    No other collective touches the data path (complexes are independent)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = shard_complexes(costs, world)[rank]
    pos, feat = fn(mine) if len(mine) else ([], [])
    assert len(pos) == len(mine) and len(feat) == len(mine)
    return gather_ligand_lists(pos, feat, group=group, device=device)


def common_seed(group=None) -> int:
    """One 63-bit seed all ranks agree on (rank 0 draws it from torch's CPU generator, so torch.manual_seed reproduces a run)."""
    box = [int(torch.randint(0, 2 ** 62, (1,)).item()) if dist.get_rank(group) == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return int(box[0])


def allreduce_gradients(params, group=None, bucket_bytes: int = 64 << 20, average: bool = True):
    """Data-parallel training (train.py run one process per GPU): sum (or average) the `.grad` of `params` over ranks.

    The gradients of the denoiser arrive all at once from one backward call (kpd_egnn_trainer_backward), so there is no
    per-layer overlap to exploit; what matters on xGMI is few, large ring all-reduces: gradients are packed into
    flat buckets of `bucket_bytes` (the whole 48 MB of egnn_all_atom fits one), every bucket's all-reduce is launched
    asynchronously back to back, then results are scattered back into the `.grad` tensors.  RCCL on the GPU box, gloo in
    the CPU tests; frozen parameters (grad None) are skipped on every rank alike.  A one-rank group still runs the collective
    (a 48 MB pack + copy, ~0.1 ms): one code path for every world size.  Returns the number of buckets reduced."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or not dist.is_initialized():
        return 0
    world = dist.get_world_size(group)
    buckets, cur, size = [], [], 0
    for gr in grads:
        nbytes = gr.numel() * gr.element_size()
        if cur and size + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(gr)
        size += nbytes
    if cur:
        buckets.append(cur)
    work = []
    for b in buckets:
        flat = torch.cat([gr.reshape(-1) for gr in b])
        work.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True), flat, b))
    for handle, flat, b in work:
        handle.wait()
        if average:
            flat.div_(world)
        off = 0
        for gr in b:
            gr.copy_(flat[off:off + gr.numel()].view_as(gr))
            off += gr.numel()
    return len(buckets)
