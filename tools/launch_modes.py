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
    # throughput of INDEPENDENT steps on two streams (two recordings, each with its own buffers): not a step's latency
    # (a second model with the same parameters: the plans' scratch - the gene layers' tables, the split planes of x - belongs to a
    # model's modules, and two steps in flight must not share it)
    import copy
    twin = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
    twin.load_state_dict(copy.deepcopy(model.state_dict()))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    recs = []
    for st, m in zip(streams, (model, twin)):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            recs.append(PoseStages(m, data, recorded=True))
    torch.cuda.synchronize()

    def both():
        recs[0].step()
        recs[1].step()
    for _ in range(5):
        both()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(100):
        both()
    torch.cuda.synchronize()
    print("two streams, independent steps         {:7.1f} us per step (throughput)".format(1e6 * (time.perf_counter() - t0) / 200))
    za, sa = recs[0].step()
    zb, sb = recs[1].step()
    torch.cuda.synchronize()
    print("both lanes == eager:", bool(torch.equal(za, z0) and torch.equal(sa, s0) and torch.equal(zb, z0) and torch.equal(sb, s0)))
