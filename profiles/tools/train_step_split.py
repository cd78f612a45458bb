"""Where a training step's wall time goes outside the engine kernels (synchronised between phases, so each phase is max(host, GPU) of that
phase; the async step beside it):  python profiles/tools/train_step_split.py [workload] [--fused-optimizer]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

wl = next((a for a in sys.argv[1:] if not a.startswith('-')), 'egnn_train')
fused = '--fused-optimizer' in sys.argv
dev = torch.device('cuda:0')
model = bench.build_model(dev, wl).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True) if fused else torch.optim.Adam(model.parameters(), lr=1e-4)
template = bench.raw_batch(64, 300, 25, 1234, dev, wl).to(dev)


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


def step(split, acc):
    t0 = sync() if split else 0
    g = template.to(dev)
    losses = model(g, None)
    t1 = sync() if split else 0
    opt.zero_grad(set_to_none=True)
    losses['l2'].backward()
    t2 = sync() if split else 0
    torch.nn.utils.clip_grad_value_(model.parameters(), 1.0)
    opt.step()
    t3 = sync() if split else 0
    if split:
        for k, v in (('forward (noise, graph, engine forward, loss)', t1 - t0), ('backward', t2 - t1), ('clip + Adam', t3 - t2)):
            acc.setdefault(k, []).append(1e3 * v)


acc = {}
for _ in range(3):
    step(False, acc)
for _ in range(8):
    step(True, acc)
for k, v in acc.items():
    print('%-48s %7.2f ms' % (k, statistics.median(v)))
print('%-48s %7.2f ms' % ('sum of the synchronised phases', sum(statistics.median(v) for v in acc.values())))
t0 = sync()
for _ in range(8):
    step(False, acc)
print('%-48s %7.2f ms' % ('asynchronous step (as bench.py times it)', 1e3 * (sync() - t0) / 8))
# host time of a step when nothing waits for the GPU: enqueue only
t0 = time.perf_counter()
for _ in range(4):
    step(False, acc)
host = 1e3 * (time.perf_counter() - t0) / 4
sync()
print('%-48s %7.2f ms' % ('host time per step (enqueue, incl. the forward sync)', host))
