"""Where does the host time of the forward with a plan-less decoder go?  (cProfile of 20 steps; GPU box.)"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gripnet_amd.pipeline import PoseModel, PoseStages
from gripnet_amd.synth import make_pose

dev = torch.device("cuda:0")
data = make_pose("pose0-syn").to(dev)
torch.manual_seed(1111)
model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
with torch.no_grad():
    model.dmt.auto_static = False
    st = PoseStages(model, data, graphs=False)
    for _ in range(5):
        st.step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        st.step()
    torch.cuda.synchronize()
    print("ms per step", 1e3 * (time.perf_counter() - t) / 20)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        st.step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
