"""Development probe: time of the tall-skinny products of the node-classification layers' backward (dx = g W^T of a 50,000-row
layer) on gn_gemm_f32 - B as stored / given transposed / with an addend / accumulating - next to torch.matmul.
    python tools/probes/gemm_time_probe.py
"""
import sys, torch
sys.path.insert(0, "/root/repo")
from gripnet_amd import _hip
dev = torch.device("cuda:0")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
for (m, n, k) in [(50000, 256, 128), (50000, 128, 128), (50000, 128, 64), (50000, 64, 128), (20000, 128, 128), (50000, 64, 256), (50000, 128, 256),
                  (50000, 128, 512), (50000, 32, 64)]:
    A = torch.randn(m, k, device=dev); Bt = torch.randn(n, k, device=dev); B = Bt.t().contiguous()
    out = torch.empty(m, n, device=dev)
    wide = torch.randn(m, n + 128, device=dev); add = wide[:, :n]
    print(m, n, k, "b as stored %.1f us | b transposed %.1f | + addend %.1f | + accumulate %.1f | torch %.1f" % (
        t(lambda: _hip.gemm(A, B, out)), t(lambda: _hip.gemm(A, Bt, out, b_transposed=True)),
        t(lambda: _hip.gemm(A, Bt, out, b_transposed=True, addend=add)), t(lambda: _hip.gemm(A, Bt, out, b_transposed=True, accumulate=True)),
        t(lambda: torch.matmul(A, B, out=out))))
