import sys, time, numpy as np, os, threading
sys.path.insert(0,'.')
import torch
from keypoint_diffusion_amd import hip
print('affinity', len(os.sched_getaffinity(0)))
rng=np.random.default_rng(0)
P=[]
for b in range(64):
    rec=rng.normal(size=(300,3))*6; kp=rng.normal(size=(40,3))*6
    P.append(((kp[:,None,:]-rec[None,:,:])**2).sum(-1))
for rep in range(3):
    for nt in (1,8,16,32):
        t0=time.perf_counter(); hip.ot_emd_uniform(P, n_threads=nt); t1=time.perf_counter()
        print('rep',rep,'threads', nt, '%.1f ms'%((t1-t0)*1e3), flush=True)
# with the main thread busy launching GPU work
x=torch.randn(4096,4096,device='cuda')
def solve(out):
    t0=time.perf_counter(); hip.ot_emd_uniform(P, n_threads=16); out.append((time.perf_counter()-t0)*1e3)
for rep in range(3):
    out=[]; th=threading.Thread(target=solve,args=(out,)); th.start()
    for _ in range(400): y=x*1.0001
    torch.cuda.synchronize(); th.join(); print('concurrent with launches: %.1f ms'%out[0])
