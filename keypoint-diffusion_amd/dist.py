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
    """Every rank returns the ligand positions / features of ALL ranks' complexes, rank-major."""
    world = dist.get_world_size(group)
    # RCCL moves device tensors; gloo (CPU tests, one-GPU rehearsals) wants host tensors
    dev = g.device if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    counts = g.batch_num_nodes('lig').to(dev).int()
    x, h = g.nodes['lig'].data['x_0'].to(dev), g.nodes['lig'].data['h_0'].to(dev)
    F = h.shape[1]
    # 1) how many complexes / atoms everybody has
    meta = torch.tensor([counts.numel(), int(counts.max()) if counts.numel() else 0], device=dev, dtype=torch.int32)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    max_B = max(int(m[0]) for m in metas)
    max_n = max(int(m[1]) for m in metas)
    # 2) one padded block per rank: [max_B, 1 + max_n * (3 + F)]  (first column = atom count)
    block = torch.zeros(max_B, 1 + max_n * (3 + F), device=dev, dtype=torch.float32)
    ptr = g.node_ptr('lig').tolist()
    for b in range(counts.numel()):
        n = ptr[b + 1] - ptr[b]
        block[b, 0] = n
        block[b, 1:1 + n * 3] = x[ptr[b]:ptr[b + 1]].reshape(-1)
        block[b, 1 + max_n * 3:1 + max_n * 3 + n * F] = h[ptr[b]:ptr[b + 1]].reshape(-1)
    blocks = [torch.zeros_like(block) for _ in range(world)]
    dist.all_gather(blocks, block, group=group)
    pos, feat = [], []
    for r in range(world):
        blk = blocks[r].cpu()
        for b in range(int(metas[r][0])):
            n = int(blk[b, 0])
            pos.append(blk[b, 1:1 + n * 3].reshape(n, 3).clone())
            feat.append(blk[b, 1 + max_n * 3:1 + max_n * 3 + n * F].reshape(n, F).clone())
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
