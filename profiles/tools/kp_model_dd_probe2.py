import sys, torch
sys.path.insert(0, '/root/repo')
from tests import test_recenc_train_gpu as T
from keypoint_diffusion_amd import graph as G, synth
cuda = torch.device('cuda:0')
K = 8
def build():
    model, cut = T._kd_model(cuda, K)
    return model.eval(), cut
def run(model, cut, w_l2, w_enc):
    torch.manual_seed(77)
    out = model(G.batch(synth.synth_complexes([60, 45, 52], [9, 13, 7], K, cut, seed=11)).to(cuda), None)
    return w_l2 * out['l2'] + w_enc * out['rec_encoder'], out
n = 'rec_encoder.scalar_embed.2.weight'
m1, cut = build()
l, parts = run(m1, cut, 1.0, 1.0)
(l - parts['l2']).backward()
m1.zero_grad(set_to_none=True)
run(m1, cut, 1.0, 0.0)[0].backward()
gA = dict(m1.named_parameters())[n].grad.clone()
m2, cut = build()
run(m2, cut, 1.0, 0.0)[0].backward()
gB = dict(m2.named_parameters())[n].grad.clone()
m3, cut = build()
run(m3, cut, 0.0, 1.0)[0].backward()
m3.zero_grad(set_to_none=True)
run(m3, cut, 1.0, 0.0)[0].backward()
gC = dict(m3.named_parameters())[n].grad.clone()
print('A vs B', float((gA - gB).abs().max()), float(gB.abs().max()), 'C vs B', float((gC - gB).abs().max()))
gen = torch.Generator().manual_seed(3)
for shp in ((128, 161), (33, 33), (128, 128)):
    d = torch.randn(shp, generator=gen)
print('dir check <gB, d>', float((gB.cpu().double() * d.double()).sum()), '<gA, d>', float((gA.cpu().double() * d.double()).sum()))
