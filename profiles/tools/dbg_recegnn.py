import sys, torch
sys.path.insert(0, '/root/repo')
from tests import util
from tests.golden.make_golden_cfgs import RECEGNN_CFGS, same_res_feature
from tests.test_recegnn_train_gpu import _batch, CUT
from keypoint_diffusion_amd import graph as G, synth
from keypoint_diffusion_amd.receptor_encoder import ReceptorEncoder
from oracle import rec_encoder_egnn as orec
cuda = torch.device('cuda:0')
cfg = RECEGNN_CFGS['recegnn_20kp']; n_rec = [33, 21]
kw = dict(cfg, graph_cutoffs=CUT)
model = synth.fill_state_dict_(ReceptorEncoder(**kw), 71).eval()
with torch.no_grad():
    for n, p in model.named_parameters():
        if 'coord_mlp.2.weight' in n: p.mul_(20.0)
g, a = _batch(cfg, n_rec)
n_kp, D = len(n_rec) * cfg['n_keypoints'], cfg['out_n_node_feat']
gen = torch.Generator().manual_seed(5)
w_x, w_h = torch.randn(n_kp, 3, generator=gen), torch.randn(n_kp, D, generator=gen) / D ** 0.5
for which in ('x', 'h'):
    sd = {k: v.detach().double().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    ob = util.to_obatch(g); ob.x['rec'], ob.h['rec'] = ob.x['rec'].double(), ob.h['rec'].double()
    ref = orec.rec_encoder_egnn_forward(sd, kw, ob, a.double())
    ((ref.x['kp'] * w_x.double()).sum() if which == 'x' else (ref.h['kp'] * w_h.double()).sum()).backward()
    m = ReceptorEncoder(**kw); m.load_state_dict(model.state_dict()); m = m.eval().to(cuda)
    gd = g.to(cuda)
    kp = m(gd, G.get_batch_idxs(gd)).nodes['kp'].data
    ((kp['x_0'] * w_x.to(cuda)).sum() if which == 'x' else (kp['h_0'] * w_h.to(cuda)).sum()).backward()
    for n, p in m.named_parameters():
        if 'rec_convs.3.coord' in n or 'rec_convs.2.coord_mlp.2' in n:
            r = sd[n].grad
            print(which, n, 'ours', None if p.grad is None else float(p.grad.abs().max()), 'ref', None if r is None else float(r.abs().max()))
