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


def all_gather_ligands(g: G.HeteroBatch, group=None) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """Every rank returns the ligand positions / features of ALL ranks' complexes, rank-major.
    Two collectives (sizes, then one padded block per rank), packing and unpacking without per-complex device work: the
    block is filled by one indexed assignment, copied to the host once, and the results are views into that copy."""
    world = dist.get_world_size(group)
    # RCCL moves device tensors; gloo (CPU tests, one-GPU rehearsals) wants host tensors
    dev = g.device if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    counts = g.batch_num_nodes('lig').to(dev).long()
    x, h = g.nodes['lig'].data['x_0'].to(dev).float(), g.nodes['lig'].data['h_0'].to(dev).float()
    F, B = h.shape[1], counts.numel()
    # 1) how many complexes / atoms everybody has
    meta = torch.stack([torch.tensor(B, device=dev), counts.max() if B else torch.tensor(0, device=dev)]).to(torch.int32)
    metas = torch.empty(world * 2, device=dev, dtype=torch.int32)
    dist.all_gather_into_tensor(metas, meta, group=group)
    metas = metas.view(world, 2).cpu()
    max_B, max_n = int(metas[:, 0].max()), int(metas[:, 1].max())
    # 2) one padded block per rank: [max_B, 1 + max_n * (3 + F)], column 0 = atom count, then atoms x (x, h)
    W = 3 + F
    block = torch.zeros(max_B, 1 + max_n * W, device=dev, dtype=torch.float32)
    if B:
        block[:B, 0] = counts.float()
        cid = torch.repeat_interleave(torch.arange(B, device=dev), counts)
        start = torch.cumsum(counts, 0) - counts
        local = torch.arange(x.shape[0], device=dev) - start[cid]
        body = block[:, 1:].view(max_B, max_n, W)
        body[cid, local] = torch.cat([x, h], dim=1)
    blocks = torch.empty(world, max_B, 1 + max_n * W, device=dev, dtype=torch.float32)
    dist.all_gather_into_tensor(blocks.view(world * max_B, -1), block, group=group)
    blocks = blocks.cpu()
    pos, feat = [], []
    for r in range(world):
        body = blocks[r, :, 1:].view(max_B, max_n, W)
        ns = blocks[r, :int(metas[r, 0]), 0].long().tolist()
        for b, n in enumerate(ns):
            pos.append(body[b, :n, :3])
            feat.append(body[b, :n, 3:])
    return pos, feat


def allreduce_gradients(params, group=None, bucket_bytes: int = 64 << 20, average: bool = True):
    """Data-parallel training (train.py run one process per GPU): sum (or average) the `.grad` of `params` over ranks.

    The gradients of the denoiser arrive all at once from one backward call (kpd_egnn_trainer_backward), so there is no
    per-layer overlap to exploit; what matters on xGMI is few, large ring all-reduces: gradients are packed into
    flat buckets of `bucket_bytes` (the whole 48 MB of egnn_all_atom fits one), every bucket's all-reduce is launched
    asynchronously back to back, then results are scattered back into the `.grad` tensors.  RCCL on the GPU box, gloo in
    the CPU tests; frozen parameters (grad None) are skipped on every rank alike."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or not dist.is_initialized() or dist.get_world_size(group) == 1:
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
