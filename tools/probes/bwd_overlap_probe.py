#!/usr/bin/env python3
"""Development probe: do independent parts of the PoSE backward pass overlap when they are launched on two streams?
(decoder backward of the positive list (planned) and of a fresh negative list; the relational layer's dx and dW.)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gripnet_amd import _hip                      # noqa: E402
from gripnet_amd.synth import make_pose           # noqa: E402

dev = torch.device("cuda:0")
d = make_pose("pose0-syn").to(dev)
n, R = d.n_d_node, d.n_dd_edge_type
E = d.train_idx.shape[1]
torch.manual_seed(1)
z, w = torch.randn(n, 80, device=dev) * 0.3, torch.randn(R, 80, device=dev) * 0.3
g1, g2 = torch.randn(E, device=dev), torch.randn(E, device=dev)
neg = _hip.NegativeSampler(d.train_idx, n, d.train_range).sample(seed=3)
plan = _hip.DistMultBwdPlan(d.train_idx, d.train_et, n, R)
dz1, dd1, dz2, dd2 = torch.empty_like(z), torch.empty_like(w), torch.empty_like(z), torch.empty_like(w)
side = torch.cuda.Stream()


def pos():
    plan.backward(z, w, g1, dz1, dd1)


def negs():
    _hip.distmult_backward(z, neg, d.train_et, w, g2, dz2, dd2)


def clock(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


def serial():
    pos(); negs()


def forked():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        negs()
    pos()
    torch.cuda.current_stream().wait_stream(side)


print("decoder backward  positives {:.0f} us, negatives {:.0f} us, one after the other {:.0f} us, on two streams {:.0f} us".format(
    clock(pos), clock(negs), clock(serial), clock(forked)))

rp = _hip.RgcnPlan(d.train_idx, d.train_range, n)
rev, pairs, deg = rp.grad_plans()
wg = rp.weight_grad_plan()
x, gm = torch.randn(n, 48, device=dev), torch.randn(n, 32, device=dev)
bt = (torch.randn(32, 32, 48, device=dev) * 0.1).contiguous()
att = torch.randn(R, 32, device=dev) * 0.2
dxe = torch.empty(n, 48, device=dev)
dw = torch.empty(R, 48 * 32, device=dev)


def dx():
    rev.forward(gm, bt, att, None, None, False, dxe, partial=True)


def dwf():
    wg.weight_grad(x, gm, out=dw)


def rserial():
    dx(); dwf()


def rforked():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dwf()
    dx()
    torch.cuda.current_stream().wait_stream(side)


print("relational backward  dx {:.0f} us, dW {:.0f} us, one after the other {:.0f} us, on two streams {:.0f} us".format(
    clock(dx), clock(dwf), clock(rserial), clock(rforked)))

# ---- the small dense products behind the big kernels: independent of each other, a few dozen workgroups each ----
gxw = torch.randn(19081, 16, device=dev)
xg = torch.randn(19081, 64, device=dev)
wext = torch.randn(64, 16, device=dev)
dxg = torch.empty(19081, 64, device=dev)
streams = [torch.cuda.Stream() for _ in range(3)]


def gene_serial():
    _hip.gemm(gxw, wext, dxg, b_transposed=True)
    _hip.xtg(xg, gxw)


def gene_forked():
    cur = torch.cuda.current_stream()
    streams[0].wait_stream(cur)
    with torch.cuda.stream(streams[0]):
        _hip.xtg(xg, gxw)
    _hip.gemm(gxw, wext, dxg, b_transposed=True)
    cur.wait_stream(streams[0])


print("external layer backward  dx = gxw W^T and dW = x^T gxw: one after the other {:.1f} us, on two streams {:.1f} us".format(
    clock(gene_serial), clock(gene_forked)))

g32 = torch.randn(n, 32, device=dev)
root = torch.randn(48, 32, device=dev)
basis = torch.randn(32, 48 * 32, device=dev)
dbasis, datt = torch.empty(32, 48 * 32, device=dev), torch.empty(R, 32, device=dev)


def small_serial():
    _hip.gemm(att, dw, dbasis, a_transposed=True)
    _hip.gemm(dw, basis, datt, b_transposed=True)
    _hip.gemm(g32, root, dxe, b_transposed=True, accumulate=True)
    _hip.xtg(x, g32)


def small_forked():
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    with torch.cuda.stream(streams[0]):
        _hip.gemm(dw, basis, datt, b_transposed=True)
    with torch.cuda.stream(streams[1]):
        _hip.gemm(g32, root, dxe, b_transposed=True, accumulate=True)
    with torch.cuda.stream(streams[2]):
        _hip.xtg(x, g32)
    _hip.gemm(att, dw, dbasis, a_transposed=True)
    for s in streams:
        cur.wait_stream(s)


print("relational layer backward, the four small products: one after the other {:.1f} us, on four streams {:.1f} us".format(
    clock(small_serial), clock(small_forked)))


def captured(fn):
    fn(); fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(10):
            fn()
    return lambda: graph.replay()


for name, serial, forked in (("external layer", gene_serial, gene_forked), ("relational layer, small products", small_serial, small_forked)):
    a, b = clock(captured(serial)) / 10, clock(captured(forked)) / 10
    print("{} inside a captured graph (ten copies per replay): one after the other {:.1f} us, forked {:.1f} us".format(name, a, b))
