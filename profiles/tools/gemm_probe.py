import torch, time
torch.backends.cuda.preferred_blas_library('cublas')   # rocBLAS
dev='cuda'
E=167000
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
A264=torch.randn(E,264,device=dev); W257=torch.randn(257,257,device=dev); W264=torch.randn(264,264,device=dev)
W515=torch.randn(257,515,device=dev)
A=A264[:, :257]
out=torch.empty(E,264,device=dev)
print('fwd  A[E,257 ld264] @ W[257,257]^T      ', bench(lambda: torch.mm(A, W257.t(), out=out[:, :257])) )
print('fwd  A[E,264] @ W[264,264]^T            ', bench(lambda: torch.mm(A264, W264.t(), out=out)))
print('bwd  dA = G[E,257]@W[257,257]           ', bench(lambda: torch.mm(A, W257, out=out[:, :257])))
print('bwd  dA = G[E,264]@W[264,264]           ', bench(lambda: torch.mm(A264, W264, out=out)))
G=torch.randn(E,264,device=dev)
print('dW   G[E,257]^T @ A[E,257]              ', bench(lambda: torch.mm(G[:, :257].t(), A)))
print('dW   G[E,264]^T @ A[E,264]              ', bench(lambda: torch.mm(G.t(), A264)))
N=19200
H=torch.randn(N,264,device=dev)
print('node H[N,257] @ W515[:, :257]^T         ', bench(lambda: torch.mm(H[:, :257], W515[:, :257].t())))
print('node H[N,264] @ W264^T                  ', bench(lambda: torch.mm(H, W264.t())))
flop=2*E*257*257/1e9
print('GFLOP per big gemm', flop)
