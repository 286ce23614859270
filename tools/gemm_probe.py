import sys, torch
sys.path.insert(0, '/root/repo')
from gripnet_amd import _hip
dev = torch.device('cuda:0')
def timed(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters
for m, k, n in [(50000,128,64),(50000,64,64),(50000,256,64),(20000,128,128),(20000,128,32),(19081,32,16),(19081,64,16)]:
    x = torch.randn(m, k, device=dev); w = torch.randn(k, n, device=dev); out = torch.empty(m, n, device=dev)
    t_mine = timed(lambda: _hip.gemm(x, w, out))
    t_blas = timed(lambda: torch.matmul(x, w, out=out))
    err = (out - (x.double() @ w.double()).float()).abs().max().item()
    print("m=%d k=%d n=%d  gn_gemm_f32 %.1f us   torch.matmul %.1f us   (%.1f GFLOP)  blas err %.2e" % (m, k, n, t_mine, t_blas, 2e-9*m*k*n, err))
