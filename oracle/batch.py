"""Flat batch-of-complexes container used by the oracle (test infrastructure only).

Mirrors the information the reference carries in a batched DGL heterograph
(data_processing/pdbbind_processing.py:236-274): node types rec / kp / lig with `x_0`, `h_0`
(+ `v_0` on kp for GVP), edge types rr, rk, kk, kl, ll, lk, per-graph node counts.
"""
from dataclasses import dataclass, field
from typing import Dict, Tuple

import torch


@dataclass
class OBatch:
    n: Dict[str, torch.Tensor]                       # ntype -> [B] node counts (long)
    x: Dict[str, torch.Tensor]                       # ntype -> [N,3]
    h: Dict[str, torch.Tensor]                       # ntype -> [N,F]
    v: Dict[str, torch.Tensor] = field(default_factory=dict)          # ntype -> [N,V,3]
    edges: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = field(default_factory=dict)

    @property
    def batch_size(self) -> int:
        return int(next(iter(self.n.values())).numel())

    def clone(self) -> "OBatch":
        return OBatch(
            n={k: t.clone() for k, t in self.n.items()},
            x={k: t.clone() for k, t in self.x.items()},
            h={k: t.clone() for k, t in self.h.items()},
            v={k: t.clone() for k, t in self.v.items()},
            edges={k: (s.clone(), d.clone()) for k, (s, d) in self.edges.items()},
        )
