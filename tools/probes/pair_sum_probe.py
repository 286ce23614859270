#!/usr/bin/env python3
"""Times the (relation, source) sums of the relational layer's weight gradient on pose0-syn (development probe)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gripnet_amd import _hip
from gripnet_amd.synth import make_pose

dev = torch.device("cuda:0")
d = make_pose(sys.argv[1] if len(sys.argv) > 1 else "pose0-syn").to(dev)
plan = _hip.RgcnPlan(d.train_idx, d.train_range, d.n_d_node)
rev, pairs, deg = plan.grad_plans()
torch.manual_seed(3)
gm = torch.randn(d.n_d_node, 32, device=dev)
q = torch.empty(d.n_dd_edge_type * d.n_d_node, 32, device=dev)
for _ in range(5):
    pairs.aggregate(gm, None, False, q)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50):
    pairs.aggregate(gm, None, False, q)
e.record(); torch.cuda.synchronize()
n, R = d.n_d_node, d.n_dd_edge_type
ref = torch.zeros(R * n, 32, device=dev, dtype=torch.float64)
ref.index_add_(0, d.train_et * n + d.train_idx[0], gm.double().index_select(0, d.train_idx[1]))
print("max abs error against index_add in float64: {:.3e}".format(float((q.double() - ref).abs().max())))
print("pairs.aggregate: {:.1f} us per call, checksum {:.4f}".format(s.elapsed_time(e) / 50 * 1e3, float(q.double().sum())))
