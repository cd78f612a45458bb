import sys, os, torch
sys.path.insert(0, '/root/repo')
from tests import util
from keypoint_diffusion_amd import graph as G, synth
from keypoint_diffusion_amd.dynamics import LigRecDynamics
cuda = torch.device('cuda:0')
g = util.fixed_encode(util.make_batch([300, 150, 40], [25, 9, 3]))
model = LigRecDynamics(10, 10, graph_cutoffs=util.CUTOFFS_ALL_ATOM, **util.EGNN_C2)
synth.fill_state_dict_(model, 3)
model = model.eval().to(cuda); gd = g.to(cuda)
t = torch.tensor([0.3, 0.6, 0.9], device=cuda)
eng = model.engine()
n_kp, n_lig = gd.num_nodes('kp'), gd.num_nodes('lig')
with torch.no_grad():
    eng.debug('layers=1'); eng.debug('prune=0')
    eng.debug('gemm=f32'); model(gd, t, None)
    hk0 = eng.debug('h_kp', n_kp * 264).view(n_kp, 264).clone(); xk0 = eng.debug('x_kp', n_kp * 3).view(n_kp, 3).clone()
    eng.debug('gemm=f16x2'); model(gd, t, None)
    hk1 = eng.debug('h_kp', n_kp * 264).view(n_kp, 264).clone(); xk1 = eng.debug('x_kp', n_kp * 3).view(n_kp, 3).clone()
d = (hk1 - hk0).abs()
print('max err', float(d.max()), 'ref max', float(hk0.abs().max()))
pc = d.max(0).values
print('per-column err: top cols', torch.topk(pc, 10))
print('col blocks of 32:', [round(float(pc[i:i+32].max()), 5) for i in range(0, 264, 32)])
pn = d.max(1).values
print('frac nodes with err > 1e-4:', float((pn > 1e-4).float().mean()), ' top nodes', torch.topk(pn, 8).indices.tolist())
print('x_kp err', float((xk1 - xk0).abs().max()), float(xk0.abs().max()))
