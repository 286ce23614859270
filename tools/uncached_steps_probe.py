"""Which of the first un-cached steps stalls?  bench.py's cold_start_entry reads ~2 ms per step for its first 20 steps and 0.11 ms
for the next 40 (GPU box): every step timed on its own, with the caching allocator's counters before and after."""
import gc
import os
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
    eager = PoseStages(model, data, graphs=False)
    for _ in range(6):
        eager.step()
    torch.cuda.synchronize()
    dmt = model.dmt
    dmt.auto_static = False
    dmt.forget_static()
    stages = PoseStages(model, data, graphs=False)
    for _ in range(3):
        stages.step()
    torch.cuda.synchronize()
    stats0 = torch.cuda.memory_stats()
    times = []
    gc_before = gc.get_count()
    for k in range(24):
        t = time.perf_counter()
        stages.step()
        torch.cuda.synchronize()
        times.append(round(1e3 * (time.perf_counter() - t), 3))
    stats1 = torch.cuda.memory_stats()
    print("per-step ms (synchronised each):", times)
    for key in ("num_alloc_retries", "num_device_alloc", "num_device_free", "allocation.all.allocated", "segment.all.allocated"):
        print(key, stats0.get(key), "->", stats1.get(key))
    print("gc counts", gc_before, gc.get_count())
    t = time.perf_counter()
    for _ in range(20):
        stages.step()
    torch.cuda.synchronize()
    print("20 steps, one synchronisation: ms per step", round(1e3 * (time.perf_counter() - t) / 20, 4))
