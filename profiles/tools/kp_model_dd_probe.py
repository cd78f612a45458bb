import sys, torch
sys.path.insert(0, '/root/repo')
from tests import test_recenc_train_gpu as T
from keypoint_diffusion_amd import graph as G, synth
cuda = torch.device('cuda:0')
K = 8
model, cut = T._kd_model(cuda, K)
model.eval()
mk = lambda: G.batch(synth.synth_complexes([60, 45, 52], [9, 13, 7], K, cut, seed=11)).to(cuda)
def loss(with_grad, w_l2=1.0, w_enc=0.1):
    torch.manual_seed(77)
    with torch.enable_grad() if with_grad else torch.no_grad():
        out = model(mk(), None)
    return w_l2 * out['l2'] + w_enc * out['rec_encoder']
params = dict(model.named_parameters())
pick = ['rec_encoder.rr_conv_layers.1.edge_message.0.to_feats_out.0.weight', 'rec_encoder.keypoint_initializer.dst_net.weight',
        'rec_encoder.rk_conv_layers.1.edge_message.0.Wh', 'rec_encoder.scalar_embed.2.weight',
        'rec_encoder.keypoint_initializer.keypoint_embedding.0.weight', 'rec_encoder.rk_conv_layers.0.node_update.0.to_feats_out.0.weight',
        'dynamics.noise_predictor.conv_layers.0.edge_message_fns.kp_kl_lig.0.to_feats_out.0.weight']
gen = torch.Generator().manual_seed(3)
dirs = {n: torch.randn(params[n].shape, generator=gen).to(cuda) for n in pick}
for (wl, we) in ((1.0, 0.0), (0.0, 1.0)):
    model.zero_grad(set_to_none=True)
    loss(True, wl, we).backward()
    for n in pick:
        if params[n].grad is None:
            print(wl, we, n, 'no grad'); continue
        a = float((params[n].grad.double() * dirs[n].double()).sum())
        res = []
        for eps in (1e-3, 3e-4):
            v = []
            with torch.no_grad():
                for sign in (1.0, -1.0):
                    params[n].add_(sign * eps * dirs[n])
                    v.append(float(loss(False, wl, we).double()))
                    params[n].sub_(sign * eps * dirs[n])
            res.append((v[0] - v[1]) / (2 * eps))
        print(f'l2={wl} enc={we} {n[-60:]:60s} analytic {a:+.5e} numeric {res[0]:+.5e} {res[1]:+.5e}')
