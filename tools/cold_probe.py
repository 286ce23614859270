"""bench.cold_start_entry under cProfile (why does its un-cached forward read 2 ms when tools/uncached_probe.py reads 0.11?)"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from gripnet_amd.pipeline import PoseModel, PoseStages
from gripnet_amd.synth import make_pose

dev = torch.device("cuda:0")
data = make_pose("pose0-syn").to(dev)
torch.manual_seed(1111)
model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
with torch.no_grad():
    eager = PoseStages(model, data, graphs=False)
    for _ in range(6):
        eager.step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    out = bench.cold_start_entry(model, data, torch.cuda.synchronize)
    pr.disable()
    print(out["forward_ms_decoder_uncached"], out["plan_build_ms_by_plan"])
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
