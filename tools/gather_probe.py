#!/usr/bin/env python3
"""dev probe: how the destination-major gather's time depends on where its table rows live (L2 / Infinity Cache): the
same degree structure with the sources confined to the first `span` rows of the table."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gripnet_amd import _hip

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
n, e = 50_000, 500_000
for width in (32, 64, 128):
    for span in (1024, 4096, 16384, n):
        src = torch.randint(0, span, (e,), generator=g)
        dst = torch.randint(0, n, (e,), generator=g)
        plan = _hip.GraphPlan.plain_sum(torch.stack([src, dst]).to(dev), n, n)
        x = torch.randn(n, width, device=dev)
        out = torch.empty(n, width, device=dev)
        for _ in range(3):
            plan.aggregate(x, None, False, out)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(20):
            plan.aggregate(x, None, False, out)
        b.record(); torch.cuda.synchronize()
        us = 1e3 * a.elapsed_time(b) / 20
        print("width {:3d}, sources in first {:6d} rows ({:5.1f} MB): {:6.1f} us  {:5.2f} TB/s gathered".format(
            width, span, span * width * 4 / 1e6, us, e * width * 4 / us / 1e6))
