"""eps of the GVP denoiser (gvp_all_atom shape, 6 convs) against the float64-free CPU oracle in both GEMM modes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoint_diffusion_amd import graph as G, synth
from keypoint_diffusion_amd.dynamics_gvp import LigRecDynamicsGVP
from oracle import gvp as ogvp
from tests import util
from tests.test_gvp_gpu import GVP_ALL_ATOM

dev = torch.device('cuda:0'); cut = util.CUTOFFS_ALL_ATOM
gs = synth.synth_complexes([150, 220, 90], [18, 25, 12], 20, cut, seed=21)
g = util.fixed_encode(G.batch(gs), n_vec=16)
t = torch.tensor([0.3, 0.7, 0.95])
base = synth.fill_state_dict_(LigRecDynamicsGVP(10, 10, graph_cutoffs=cut, **GVP_ALL_ATOM), 1).eval()
sd = {k: v.clone() for k, v in base.state_dict().items()}
rh, rx = ogvp.gvp_dynamics_forward(sd, dict(GVP_ALL_ATOM, graph_cutoffs=cut), util.to_obatch(g), t)
for mode in ('f32', 'f16x2'):
    os.environ['KPD_GEMM'] = mode
    m = synth.fill_state_dict_(LigRecDynamicsGVP(10, 10, graph_cutoffs=cut, **GVP_ALL_ATOM), 1).eval().to(dev)
    with torch.no_grad():
        h, x = m(g.to(dev), t.to(dev), None)
    print(f'{mode}: eps_h rel err {util.rel_err(h.cpu(), rh):.3e}   eps_x rel err {util.rel_err(x.cpu(), rx):.3e}')
