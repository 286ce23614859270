#!/usr/bin/env python3
"""dev tool: the pose0-syn forward step under the launch modes bench.py can choose from, wall clock per step, no event timing:
eager / genes graph + two Python launches (bench's graphs mode) / two graphs (encode, decode) / ONE graph / the recorded
entry-point calls replayed from one loop."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gripnet_amd.pipeline import Graphed, PoseModel, PoseStages
from gripnet_amd.synth import make_pose

dev = torch.device("cuda:0")
data = make_pose("pose0-syn").to(dev)
torch.manual_seed(1111)
model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)


def clock(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / n)
    return 1e6 * best


with torch.no_grad():
    eager = PoseStages(model, data, graphs=False)
    for _ in range(3):
        eager.step()
    print("eager                                  {:7.1f} us".format(clock(eager.step)))
    mixed = PoseStages(model, data, graphs=True, timed_entry="gn_rgcn_forward_f32")
    print("genes graph + rgcn + decoder launches  {:7.1f} us".format(clock(mixed.step)))
    two = PoseStages(model, data, graphs=True, timed_entry=None)
    print("encode graph + decode graph            {:7.1f} us".format(clock(two.step)))
    one = Graphed(eager.step).capture()
    print("ONE graph                              {:7.1f} us".format(clock(one)))
    rec = PoseStages(model, data, recorded=True)
    print("recorded entry-point calls, one loop   {:7.1f} us".format(clock(rec.step)))
    z0, s0 = eager.step()
    z0, s0 = z0.clone(), s0.clone()
    z1, s1 = rec.step()
    torch.cuda.synchronize()
    print("recorded == eager:", bool(torch.equal(z0, z1) and torch.equal(s0, s1)), "calls per step:", len(rec._whole.calls))
