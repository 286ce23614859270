#!/usr/bin/env python3
"""Times the destination-major relational layer alone (development tool; no result check - for timing experiments on
variant libraries, GN_HIP_LIBRARY=...): python tools/time_pair_layer.py [--workload pose0-syn] [--launches 300]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gripnet_amd
from gripnet_amd import _hip
from gripnet_amd.synth import make_pose

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="pose0-syn")
ap.add_argument("--launches", type=int, default=300)
args = ap.parse_args()
dev = torch.device("cuda:0")
data = make_pose(args.workload).to(dev)
torch.manual_seed(1111)
conv = gripnet_amd.myRGCN(48, 32, data.n_dd_edge_type, 32, False).to(dev)
x = torch.randn(data.n_d_node, 48, device=dev)
_hip.SplitPlanes(data.n_d_node, 3, dev).fill_from(x).tag(x)
with torch.no_grad():
    for _ in range(20):
        conv(x, data.train_idx, data.train_et, data.train_range, _relu=True)
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.launches):
            conv(x, data.train_idx, data.train_et, data.train_range, _relu=True)
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) * 1e3 / args.launches)
print("us per launch (back to back, three repeats):", " ".join("{:.2f}".format(v) for v in best))
