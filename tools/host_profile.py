#!/usr/bin/env python3
"""Host-side cost of the reference-shaped API: `model(data)` under no_grad, as a caller of GripNet-pose.py:117-138 makes it.
Prints the wall time per forward (eager; host-bound when the Python above the launches is slower than the kernels) and the
cProfile table of 300 forwards.  `--train` profiles the eager training step instead."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gripnet_amd.pipeline import PoseModel
from gripnet_amd.synth import make_pose

dev = torch.device("cuda:0")
data = make_pose("pose0-syn").to(dev)
torch.manual_seed(1111)
model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
n = 300
if "--train" in sys.argv:
    from gripnet_amd import _hip
    from gripnet_amd.optim import Adam
    from gripnet_amd.utils import link_loss, link_prediction_loss
    fused = "--fused" in sys.argv
    opt = Adam(model.parameters(), lr=0.01)
    sampler = _hip.NegativeSampler(data.train_idx, data.n_d_node, data.train_range)
    neg = sampler.sample(seed=0)
    drawn = torch.ones((1,), dtype=torch.int64, device=dev)
    one = torch.ones((), dtype=torch.float32, device=dev)

    def step():
        sampler.sample(seed=0, out=neg, step=drawn)
        opt.zero_grad()
        z = model.encode(data)
        if fused:
            loss, _, _ = link_prediction_loss(model.dmt, z, data.train_idx, neg, data.train_et)
        else:
            loss = link_loss(model.dmt(z, data.train_idx, data.train_et), model.dmt(z, neg, data.train_et))
        loss.backward(one)
        opt.step()

    n = 100
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("eager training step: {:.1f} us (host loop alone {:.1f} us)".format(1e6 * (time.perf_counter() - t0) / n, 1e6 * t_host / n))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(35)
    sys.exit(0)
with torch.no_grad():
    for _ in range(10):
        model(data)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            model(data)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print("eager model(data): {:.1f} us per forward (host loop alone {:.1f} us)".format(1e6 * t_all / n, 1e6 * t_host / n))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        model(data)
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
