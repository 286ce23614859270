#!/usr/bin/env python3
"""The all-nodes relational baseline (pipeline.RgcnPoseModel on synth.make_rgcn_pose: baselines/LP_baselines/rgcn_pose.py)
on its own: HIP-event time of every entry point of one forward (development tool)."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gripnet_amd import _hip                       # noqa: E402
from gripnet_amd.pipeline import RgcnPoseModel     # noqa: E402
from gripnet_amd.synth import make_rgcn_pose       # noqa: E402

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
data = make_rgcn_pose(sys.argv[2] if len(sys.argv) > 2 else "pose0-syn").to(dev)
torch.manual_seed(1111)
model = RgcnPoseModel(data.n_node, data.n_edge_type).to(dev)
with torch.no_grad():
    for _ in range(3):
        model(data)
    torch.cuda.synchronize()
    with _hip.KernelTimer(pool=200) as t:
        for _ in range(iters):
            model(data)
    for k, (calls, ms) in t.summary().items():
        print("{:40s} {:3d} calls  {:8.1f} us per call".format(k, calls, 1e3 * ms / calls))
print("path", model.rgcn1._plan.path(64, 32, 16), "nodes", data.n_node, "edges", data.train_idx.shape[1])
