#!/usr/bin/env python3
"""Host-side cost of one forward step (development tool): enqueue time of 20 steps without a device sync,
against the device time of the same steps, for the eager and the hipGraph launch modes."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gripnet_amd.pipeline import PoseModel, PoseStages, Graphed    # noqa: E402
from gripnet_amd.synth import make_pose                             # noqa: E402

dev = torch.device("cuda:0")
data = make_pose("pose0-syn").to(dev)
torch.manual_seed(1111)
model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
with torch.no_grad():
    modes = {"eager": PoseStages(model, data, graphs=False)}
    for _ in range(3):
        modes["eager"].step()
    modes["graph+eager decode"] = PoseStages(model, data, graphs=True, timed_entry="gn_distmult_forward_f32")
    modes["graph x2"] = PoseStages(model, data, graphs=True, timed_entry="gn_rgcn_forward_f32")
    whole = PoseStages(model, data, graphs=False)
    modes["one graph"] = type("G", (), {"step": Graphed(whole.step).capture()})()
    conv = model.dd.conv_list[0]
    real_prefetch = conv.prefetch_weights
    conv.prefetch_weights = lambda: False                     # W_r in line, on the main stream
    modes["graph+eager decode, W in line"] = PoseStages(model, data, graphs=True, timed_entry="gn_distmult_forward_f32")
    whole2 = PoseStages(model, data, graphs=False)
    modes["one graph, W in line"] = type("G", (), {"step": Graphed(whole2.step).capture()})()
    conv.prefetch_weights = real_prefetch
    for name, st in modes.items():
        for _ in range(5):
            st.step()
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            st.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("{:32s} host enqueue {:7.1f} us/step   until drained {:7.1f} us/step".format(name, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
