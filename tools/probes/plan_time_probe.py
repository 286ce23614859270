#!/usr/bin/env python3
"""Where the one-time plan building of the pose forward goes (development probe): wall time of every plan constructor."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gripnet_amd import _hip
from gripnet_amd.synth import make_pose

dev = torch.device("cuda:0")
d = make_pose(sys.argv[1] if len(sys.argv) > 1 else "pose0-syn").to(dev)
_hip.load()
torch.cuda.synchronize()


def timed(label, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print("{:40s} {:8.1f} ms".format(label, 1e3 * (time.perf_counter() - t)))
    return r


g = timed("gcn plan (gg)", lambda: _hip.GraphPlan.gcn(d.gg_edge_index, d.n_g_node, d.edge_weight, False))
timed("  + blocked encoding (16 cols)", lambda: g.build_blocked(16))
timed("bipartite plan (gd)", lambda: _hip.GraphPlan.bipartite(d.gd_edge_index, d.n_g_node, d.n_d_node, None))
r = timed("relational plan (dd)", lambda: _hip.RgcnPlan(d.train_idx, d.train_range, d.n_d_node))
timed("decoder plan (positives)", lambda: _hip.DistMultPlan(d.train_idx, d.train_et, d.n_d_node, d.n_dd_edge_type))
timed("relational grad plans", lambda: r.grad_plans())
