#!/usr/bin/env python3
"""Development probe: the typed negative sampler on pose0-syn - the task kernel (tests staged in LDS) against the bitmap kernel
(GN_SAMPLER_TASKS=0): same draws, time per call."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gripnet_amd import _hip                       # noqa: E402
from gripnet_amd.synth import make_pose            # noqa: E402

dev = torch.device("cuda:0")
data = make_pose("pose0-syn").to(dev)


def clock(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


out = {}
for tasks in ("1", "0"):
    os.environ["GN_SAMPLER_TASKS"] = tasks
    s = _hip.NegativeSampler(data.train_idx, data.n_d_node, data.train_range)
    buf = s.sample(seed=3)
    out[tasks] = buf.clone()
    print("GN_SAMPLER_TASKS={}: {:.1f} us per draw of {} pairs".format(tasks, clock(lambda: s.sample(seed=4, out=buf)), buf.shape[1]))
print("same draws:", torch.equal(out["1"], out["0"]))
_hip.raise_if_index_errors(dev)
pk = torch.empty((buf.shape[1],), dtype=torch.int32, device=dev)
print("writing the outputs alone (fill of [2, E] int64 + [E] int32): {:.1f} us".format(clock(lambda: (buf.fill_(3), pk.fill_(1)))))
print("                                  [2, E] int64 only: {:.1f} us".format(clock(lambda: buf.fill_(3))))
