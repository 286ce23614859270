#!/usr/bin/env python3
"""Development probe: what does a trivial kernel cost inside a hipGraph replay, and as one of a list of recorded entry-point
calls made again from one loop?  (50 launches of gn_merge_f32 on a 16 x 16 matrix; 1 MB matrices for comparison.)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gripnet_amd import _hip                      # noqa: E402
from gripnet_amd.pipeline import Graphed, Recorded  # noqa: E402

dev = torch.device("cuda:0")
for rows in (16, 16384):
    a, b = torch.randn(rows, 16, device=dev), torch.empty(rows, 16, device=dev)

    def chain():
        for _ in range(50):
            _hip.merge(b, a, 1)
        return b

    for name, wrap in (("hipGraph", Graphed), ("recorded", Recorded)):
        fn = wrap(chain).capture()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            fn()
        torch.cuda.synchronize()
        print("{:5d} rows  {:9s} {:6.2f} us per kernel".format(rows, name, 1e6 * (time.perf_counter() - t0) / 40 / 50))
