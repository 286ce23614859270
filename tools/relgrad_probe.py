#!/usr/bin/env python3
"""Launch time of the fused relational weight gradient (gn_rel_weight_grad_f32) on a PoSE workload, next to the unfused
path it replaces (development tool).  python tools/relgrad_probe.py [workload]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gripnet_amd import _hip                      # noqa: E402
from gripnet_amd.synth import make_pose           # noqa: E402

dev = torch.device("cuda:0")
d = make_pose(sys.argv[1] if len(sys.argv) > 1 else "pose0-syn").to(dev)
n, R = d.n_d_node, d.n_dd_edge_type
torch.manual_seed(3)
x, gm = torch.randn(n, 48, device=dev), torch.randn(n, 32, device=dev)
plan = _hip.RgcnPlan(d.train_idx, d.train_range, n)
wg = plan.weight_grad_plan()
pairs = plan.grad_plans()[1]


def clock(fn, reps=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


q = torch.empty((R * n, 32), device=dev)


def unfused():
    pairs.aggregate(gm, None, False, q)
    return torch.matmul(x.t(), q.view(R, n, 32)).reshape(R, 48 * 32)


ref = unfused()
dw = wg.weight_grad(x, gm)
print("max |fused - unfused| / max |.|: {:.2e}".format(float((dw - ref).abs().max() / ref.abs().max())))
print("fused   {:7.1f} us".format(clock(lambda: wg.weight_grad(x, gm, out=dw))))
print("unfused {:7.1f} us".format(clock(unfused)))
