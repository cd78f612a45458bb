"""Repeatability probe of the EGNN training path: the same case built several times in one process (a new trainer
instance each time) must give the same forward outputs and gradient norms up to the order of float atomics."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_egnn_train_gpu as T, util
cfg = dict(util.EGNN_C2, n_layers=3)
ref = None
for rep in range(4):
    g, model, t = T._case(cfg, [60, 35, 48], [9, 14, 6], rec_nf=10)
    model = model.cuda(); gd = g.to('cuda')
    eh, ex = model(gd, t.cuda(), None)
    with torch.no_grad():
        eh_i, ex_i = model(g.to('cuda'), t.cuda(), None)
    (eh.sum() + ex.sum()).backward()
    gsum = sum(float(p.grad.double().abs().sum()) for p in model.parameters())
    cur = (eh.detach().cpu(), ex.detach().cpu())
    if ref is None:
        ref = cur
    print(rep, 'train-vs-first', util.rel_err(cur[0], ref[0]), util.rel_err(cur[1], ref[1]), 'train-vs-inference',
          util.rel_err(cur[0], eh_i.cpu()), util.rel_err(cur[1], ex_i.cpu()), 'grad abs sum', gsum)
