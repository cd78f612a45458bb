import sys, torch
sys.path.insert(0,'.')
import bench
dev=torch.device('cuda:0')
for wl in ('egnn_40kp_train','gvp_40kp_train','egnn_train','gvp_train'):
    torch.manual_seed(0)
    model=bench.build_model(dev, wl).train()
    opt=torch.optim.Adam(model.parameters(), lr=3e-4)
    tmpl=bench.raw_batch(16,300,25,1234,dev,wl).to(dev)
    hist=[]
    for i in range(40):
        g=tmpl.to(dev)
        torch.manual_seed(100+i%4)           # four fixed (t, eps) draws: the loss on them must fall
        losses=model(g, None)
        loss=losses['l2'] + (0.1*losses['rec_encoder'] if wl.endswith('40kp_train') else 0)
        opt.zero_grad(set_to_none=True); loss.backward()
        torch.nn.utils.clip_grad_value_(model.parameters(), 1.0); opt.step()
        hist.append(float(losses['l2'].detach()))
    ok=all(map(lambda v: v==v and abs(v)<1e6, hist))
    print(wl, 'finite', ok, 'l2 first 4 avg %.4f last 4 avg %.4f'%(sum(hist[:4])/4, sum(hist[-4:])/4), flush=True)
