"""Node-sized products of the EGNN training step as the engine calls them (row strides 264 / 513), one at a time and the four of a branch
back to back.  python profiles/tools/sgemm_node_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from keypoint_diffusion_amd import hip

dev = torch.device('cuda:0')
ws = torch.zeros(128 * 256 * 256, device=dev)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n in (19200, 1600):
    dU = torch.randn(n, 264, device=dev)[:, :256]
    h = torch.randn(n, 264, device=dev)[:, :256]
    W = torch.randn(256, 513, device=dev)
    Wg = torch.zeros(256, 513, device=dev)
    dh = torch.zeros(n, 264, device=dev)[:, :256]
    t_tn = timeit(lambda: hip.sgemm(dU, h, True, False, beta=1.0, out=Wg[:, :256], workspace=ws))
    t_nn = timeit(lambda: hip.sgemm(dU, W[:, :256], False, False, beta=1.0, out=dh))
    t_nt = timeit(lambda: hip.sgemm(h, W[:, :256], False, True, beta=0.0, out=dh))
    fl = 2.0 * n * 256 * 256
    print(f'n={n:6d}  TN grad {t_tn:7.1f} us ({fl / t_tn / 1e6:5.1f} TF)   NN dh {t_nn:7.1f} us ({fl / t_nn / 1e6:5.1f} TF)   NT fwd {t_nt:7.1f} us ({fl / t_nt / 1e6:5.1f} TF)', flush=True)

# the four node-sized products of one edge branch (two gradients, two input gradients): back to back on one stream / spread over streams
n = 19200
dU, dV = torch.randn(n, 264, device=dev)[:, :256], torch.randn(n, 264, device=dev)[:, :256]
h = torch.randn(n, 264, device=dev)[:, :256]
W = torch.randn(256, 513, device=dev)
Wg = torch.zeros(256, 513, device=dev)
dh1, dh2 = torch.zeros(n, 264, device=dev)[:, :256], torch.zeros(n, 264, device=dev)[:, :256]
ws2 = torch.zeros(128 * 256 * 256, device=dev)
streams = [torch.cuda.Stream() for _ in range(4)]


def serial():
    hip.sgemm(dU, h, True, False, beta=1.0, out=Wg[:, :256], workspace=ws)
    hip.sgemm(dV, h, True, False, beta=1.0, out=Wg[:, 256:512], workspace=ws2)
    hip.sgemm(dU, W[:, :256], False, False, beta=1.0, out=dh1)
    hip.sgemm(dV, W[:, 256:512], False, False, beta=1.0, out=dh2)


def spread():
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    calls = [lambda: hip.sgemm(dU, h, True, False, beta=1.0, out=Wg[:, :256], workspace=ws),
             lambda: hip.sgemm(dV, h, True, False, beta=1.0, out=Wg[:, 256:512], workspace=ws2),
             lambda: hip.sgemm(dU, W[:, :256], False, False, beta=1.0, out=dh1),
             lambda: hip.sgemm(dV, W[:, 256:512], False, False, beta=1.0, out=dh2)]
    for st, c in zip(streams, calls):
        st.wait_event(ev)
        with torch.cuda.stream(st):
            c()
        e = torch.cuda.Event()
        e.record(st)
        cur.wait_event(e)


print(f'four products of a branch, n = {n}: one stream {timeit(serial):7.1f} us, four streams {timeit(spread):7.1f} us')

# grouping the products of the (edge type, branch) pairs that share a node feature matrix: one K = 1024 / M = 1024 product instead of four
dUc = torch.randn(n, 4 * 264, device=dev)
Wst = torch.randn(1024, 256, device=dev)
out1 = torch.zeros(n, 264, device=dev)[:, :256]
G4 = torch.zeros(1024, 256, device=dev)
cat = dUc.view(n, 4, 264)[:, :, :256]
t4 = timeit(lambda: [hip.sgemm(dUc[:, 264 * b:264 * b + 256], W[:, :256], False, False, beta=1.0, out=out1) for b in range(4)])
dUflat = torch.randn(n, 1024, device=dev)
t1 = timeit(lambda: hip.sgemm(dUflat, Wst, False, False, beta=1.0, out=out1))
print(f'input gradients: four K = 256 products {t4:7.1f} us, one K = 1024 product {t1:7.1f} us')
t4 = timeit(lambda: [hip.sgemm(dUc[:, 264 * b:264 * b + 256], h, True, False, beta=1.0, out=Wg[:, :256], workspace=ws) for b in range(4)])
t1 = timeit(lambda: hip.sgemm(dUflat, h, True, False, beta=1.0, out=G4, workspace=ws))
print(f'weight gradients: four M = 256 products {t4:7.1f} us, one M = 1024 product {t1:7.1f} us')

# does the odd leading dimension of the weight operand (513-float rows) cost the node-sized products their direct path?
Wa = W[:, :256].contiguous()
for nn in (19200, 1600):
    x = torch.randn(nn, 264, device=dev)[:, :256]
    o = torch.zeros(nn, 264, device=dev)[:, :256]
    t_odd = timeit(lambda: hip.sgemm(x, W[:, :256], False, False, beta=1.0, out=o))
    t_al = timeit(lambda: hip.sgemm(x, Wa, False, False, beta=1.0, out=o))
    t_odd_nt = timeit(lambda: hip.sgemm(x, W[:, :256], False, True, beta=0.0, out=o))
    t_al_nt = timeit(lambda: hip.sgemm(x, Wa, False, True, beta=0.0, out=o))
    print(f'n={nn}: NN weight ld 513 {t_odd:6.1f} us, ld 256 {t_al:6.1f} us;  NT ld 513 {t_odd_nt:6.1f} us, ld 256 {t_al_nt:6.1f} us')

# all eight first-Linear blocks that share a node type's features as ONE product (padded 264-wide slots)
for nn in (19200, 1600):
    hh = torch.randn(nn, 264, device=dev)
    Wst = torch.randn(2112, 264, device=dev)
    Uc = torch.zeros(nn, 2112, device=dev)
    dh = torch.zeros(nn, 264, device=dev)
    G = torch.zeros(2112, 264, device=dev)
    t_nt = timeit(lambda: hip.sgemm(hh, Wst, False, True, out=Uc))
    t_nn = timeit(lambda: hip.sgemm(Uc, Wst, False, False, beta=1.0, out=dh))
    t_tn = timeit(lambda: hip.sgemm(Uc, hh, True, False, out=G, workspace=ws))
    print(f'n={nn}: grouped (8 slots)  NT {t_nt:6.1f} us  NN {t_nn:6.1f} us  TN {t_tn:6.1f} us')
