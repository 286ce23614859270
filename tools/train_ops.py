#!/usr/bin/env python3
"""Which operator every launch of the PoSE training step comes from (development tool): one eager step under
torch.profiler, kernels listed in launch order with the innermost CPU operator that was active when each was launched."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gripnet_amd import _hip                      # noqa: E402
from gripnet_amd.optim import Adam                # noqa: E402
from gripnet_amd.pipeline import PoseModel        # noqa: E402
from gripnet_amd.synth import make_pose           # noqa: E402
from gripnet_amd.utils import link_loss           # noqa: E402

dev = torch.device("cuda:0")
data = make_pose("pose0-syn").to(dev)
torch.manual_seed(1111)
model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
opt = Adam(model.parameters(), lr=0.01)
sampler = _hip.NegativeSampler(data.train_idx, data.n_d_node, data.train_range)
neg = sampler.sample(seed=0)


def step():
    opt.zero_grad()
    z = model.encode(data)
    pos = model.dmt(z, data.train_idx, data.train_et)
    negs = model.dmt(z, neg, data.train_et)
    loss = link_loss(pos, negs)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
events = prof.events()
cpu = [e for e in events if e.device_type == torch.autograd.DeviceType.CPU]
kernels = sorted((e for e in events if e.device_type == torch.autograd.DeviceType.CUDA), key=lambda e: e.time_range.start)
launches = {}
for e in cpu:                                     # correlation: the launch call's id -> the chain of CPU operators above it
    launches.setdefault(e.id, e)
for k in kernels:
    chain, p = [], getattr(k, "linked_correlation_events", None)
    owner = None
    for e in cpu:
        if any(getattr(c, "id", None) == k.id for c in getattr(e, "kernels", [])):
            owner = e
    name = k.name[:48]
    names = []
    e = owner
    while e is not None and len(names) < 4:
        names.append(e.name[:40])
        e = e.cpu_parent
    print("{:7.1f} us  {:48s} <- {}".format(k.device_time if hasattr(k, "device_time") else k.cuda_time, name, " <- ".join(names)))
