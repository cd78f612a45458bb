"""Definitional (brute-force O(N^2)) restatements of the third-party graph operations on
the hot path.  Oracle code: test infrastructure only (see oracle/__init__.py).

Third-party libraries restated here (absent from /root/reference, versions unpinned,
readme.md:14-16): pytorch-cluster (radius, radius_graph, knn, knn_graph), pytorch-scatter
(segment_csr), DGL (u_sub_v, copy_e+sum/mean, readout_nodes).  Call sites in the reference:
models/dynamics.py:392-403, models/dynamics_gvp.py:206-217,
models/receptor_encoder_gvp.py:75-87, 285, 302-306, models/ligand_diffuser.py:199.

Conventions fixed by this oracle where upstream leaves them implementation-defined:
  * radius:  strict `dist < r`; neighbours of one centre enumerated in increasing index
    order and truncated to `max_num_neighbors` in that order.
  * knn:     neighbours of one query sorted by (distance, index); fewer than k when the
    graph has fewer than k candidates.
All functions work on a flat node array plus per-graph node counts.
"""
from typing import Tuple

import torch


def counts_to_ptr(counts: torch.Tensor) -> torch.Tensor:
    ptr = torch.zeros(counts.numel() + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(counts.long(), 0)
    return ptr


def counts_to_batch_idx(counts: torch.Tensor) -> torch.Tensor:
    """utils.py:159-169 get_batch_idxs: arange(B).repeat_interleave(counts)."""
    return torch.arange(counts.numel()).repeat_interleave(counts.long())


def radius(x, y, r, n_x, n_y, max_num_neighbors=32) -> Tuple[torch.Tensor, torch.Tensor]:
    """torch_cluster.radius(x, y, r, batch_x, batch_y): for each y all x of the same graph
    with ||x - y|| < r.  Returns (y_idx, x_idx), y-major."""
    xp, yp = counts_to_ptr(n_x), counts_to_ptr(n_y)
    ys, xs = [], []
    for b in range(n_x.numel()):
        xb, yb = x[xp[b]:xp[b + 1]], y[yp[b]:yp[b + 1]]
        if xb.shape[0] == 0 or yb.shape[0] == 0:
            continue
        d2 = ((yb[:, None, :] - xb[None, :, :]) ** 2).sum(-1)
        mask = d2 < r * r
        for i in range(yb.shape[0]):
            nb = torch.nonzero(mask[i]).flatten()[:max_num_neighbors]
            ys.append(torch.full_like(nb, i + int(yp[b])))
            xs.append(nb + int(xp[b]))
    if not ys:
        return torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long)
    return torch.cat(ys), torch.cat(xs)


def radius_graph(x, r, n, max_num_neighbors=32) -> Tuple[torch.Tensor, torch.Tensor]:
    """torch_cluster.radius_graph(x, r, batch, loop=False, flow='source_to_target').
    Returns (src = neighbour, dst = centre), dst-major, self pairs removed."""
    c, nb = radius(x, x, r, n, n, max_num_neighbors + 1)
    keep = c != nb
    c, nb = c[keep], nb[keep]
    # a centre that hit the +1 allowance without containing itself keeps max_num_neighbors
    if c.numel():
        order = torch.arange(c.numel())
        first = torch.ones_like(c, dtype=torch.bool)
        first[1:] = c[1:] != c[:-1]
        start = torch.cummax(torch.where(first, order, torch.zeros_like(order)), 0).values
        keep = (order - start) < max_num_neighbors
        c, nb = c[keep], nb[keep]
    return nb, c


def knn(x, y, k, n_x, n_y) -> Tuple[torch.Tensor, torch.Tensor]:
    """torch_cluster.knn(x, y, k, batch_x, batch_y): for each y its k nearest x of the same
    graph.  Returns (y_idx, x_idx), y-major, nearest first."""
    xp, yp = counts_to_ptr(n_x), counts_to_ptr(n_y)
    ys, xs = [], []
    for b in range(n_x.numel()):
        xb, yb = x[xp[b]:xp[b + 1]], y[yp[b]:yp[b + 1]]
        if xb.shape[0] == 0 or yb.shape[0] == 0:
            continue
        d2 = ((yb[:, None, :] - xb[None, :, :]) ** 2).sum(-1)
        kk = min(k, xb.shape[0])
        # stable sort => ties broken by lower index
        idx = torch.sort(d2, dim=1, stable=True).indices[:, :kk]
        ys.append((torch.arange(yb.shape[0]) + int(yp[b])).repeat_interleave(kk))
        xs.append(idx.reshape(-1) + int(xp[b]))
    if not ys:
        return torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long)
    return torch.cat(ys), torch.cat(xs)


def knn_graph(x, k, n) -> Tuple[torch.Tensor, torch.Tensor]:
    """torch_cluster.knn_graph(x, k, batch, loop=False): (src = neighbour, dst = centre)."""
    xp = counts_to_ptr(n)
    srcs, dsts = [], []
    for b in range(n.numel()):
        xb = x[xp[b]:xp[b + 1]]
        m = xb.shape[0]
        if m < 2:
            continue
        d2 = ((xb[:, None, :] - xb[None, :, :]) ** 2).sum(-1)
        d2.fill_diagonal_(float('inf'))
        kk = min(k, m - 1)
        idx = torch.sort(d2, dim=1, stable=True).indices[:, :kk]
        dsts.append((torch.arange(m) + int(xp[b])).repeat_interleave(kk))
        srcs.append(idx.reshape(-1) + int(xp[b]))
    if not srcs:
        return torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long)
    return torch.cat(srcs), torch.cat(dsts)


def scatter_sum(msg: torch.Tensor, dst: torch.Tensor, n_dst: int) -> torch.Tensor:
    """DGL update_all(copy_e, sum): out[i] = sum of msg over edges with dst == i."""
    out = torch.zeros((n_dst,) + tuple(msg.shape[1:]), dtype=msg.dtype)
    if msg.shape[0]:
        out.index_add_(0, dst, msg)
    return out


def scatter_mean(msg: torch.Tensor, dst: torch.Tensor, n_dst: int) -> torch.Tensor:
    """DGL update_all(copy_e, mean): sum / in-degree, 0 for isolated nodes."""
    s = scatter_sum(msg, dst, n_dst)
    deg = torch.zeros(n_dst, dtype=msg.dtype)
    if msg.shape[0]:
        deg.index_add_(0, dst, torch.ones(dst.shape[0], dtype=msg.dtype))
    deg = deg.clamp(min=1).view((-1,) + (1,) * (msg.dim() - 1))
    return s / deg


def segment_mean_nodes(feat: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    """dgl.readout_nodes(g, feat, op='mean', ntype): per-graph mean over nodes."""
    bidx = counts_to_batch_idx(counts)
    s = scatter_sum(feat, bidx, counts.numel())
    return s / counts.to(feat.dtype).clamp(min=1).view((-1,) + (1,) * (feat.dim() - 1))


def edges_per_graph(dst_or_src_nodes: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    """utils.py:92-98 get_edges_per_batch: number of edges per graph, keyed by the graph of
    the given endpoint."""
    bidx = counts_to_batch_idx(counts)
    out = torch.zeros(counts.numel(), dtype=torch.long)
    if dst_or_src_nodes.numel():
        out.index_add_(0, bidx[dst_or_src_nodes], torch.ones_like(dst_or_src_nodes))
    return out
