"""f16x2 edge-kernel mode against the exact fp32 mode and the CPU oracle, layer by layer (diagnostic)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util
from keypoint_diffusion_amd import graph as G, synth
from keypoint_diffusion_amd.dynamics import LigRecDynamics
from oracle import egnn as oegnn
cuda = torch.device('cuda:0')
g = util.fixed_encode(util.make_batch([300, 150, 40], [25, 9, 3]))
model = LigRecDynamics(10, 10, graph_cutoffs=util.CUTOFFS_ALL_ATOM, **util.EGNN_C2)
synth.fill_state_dict_(model, 3)
model.eval()
t = torch.tensor([0.3, 0.6, 0.9])
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
rh, rx = oegnn.egnn_dynamics_forward(sd, dict(util.EGNN_C2, graph_cutoffs=util.CUTOFFS_ALL_ATOM), util.to_obatch(g), t)
model = model.to(cuda); gd = g.to(cuda)
eng = model.engine()
n_kp, n_lig = gd.num_nodes('kp'), gd.num_nodes('lig')
with torch.no_grad():
    for layers in (1, 2, 6):
        eng.debug(f'layers={layers}'); eng.debug('prune=0')
        eng.debug('gemm=f32'); h0, x0 = model(gd, t.to(cuda), None)
        hk0 = eng.debug('h_kp', n_kp * 264).view(n_kp, 264).clone(); hl0 = eng.debug('h_lig', n_lig * 264).view(n_lig, 264).clone()
        eng.debug('gemm=f16x2'); h1, x1 = model(gd, t.to(cuda), None)
        hk1 = eng.debug('h_kp', n_kp * 264).view(n_kp, 264).clone(); hl1 = eng.debug('h_lig', n_lig * 264).view(n_lig, 264).clone()
        torch.cuda.synchronize()
        print(f'layers={layers}: eps_h {util.rel_err(h1, h0):.3e} eps_x {util.rel_err(x1, x0):.3e}  h_kp {util.rel_err(hk1, hk0):.3e} '
              f'h_lig {util.rel_err(hl1, hl0):.3e}  h_lig[:, 256] {util.rel_err(hl1[:, 256], hl0[:, 256]):.3e} h_lig[:, :256] {util.rel_err(hl1[:, :256], hl0[:, :256]):.3e}')
    eng.debug('layers=6'); eng.debug('prune=1')
    eng.debug('gemm=f32'); h0, x0 = model(gd, t.to(cuda), None)
    eng.debug('gemm=f16x2'); h1, x1 = model(gd, t.to(cuda), None)
print('fp32  vs oracle', util.rel_err(h0, rh), util.rel_err(x0, rx))
print('f16x2 vs oracle', util.rel_err(h1, rh), util.rel_err(x1, rx))
