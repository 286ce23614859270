#!/usr/bin/env python3
"""Secondary measurement: one full training step (forward + loss + backward + Adam) of the PoSE model,
the loop body of GripNet-pose.py:112-146, on the synthetic ladder (development tool).

    python tools/bench_train.py --workload pose0-syn --steps 10
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gripnet_amd.pipeline import PoseModel        # noqa: E402
from gripnet_amd.synth import make_pose           # noqa: E402
from gripnet_amd.utils import EPS                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="pose0-syn")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    data = make_pose(args.workload).to(dev)
    torch.manual_seed(1111)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    neg = torch.randint(0, data.n_d_node, tuple(data.train_idx.shape), device=dev)

    def step():
        opt.zero_grad()
        z = model.encode(data)
        pos = model.dmt(z, data.train_idx, data.train_et)
        negs = model.dmt(z, neg, data.train_et)
        loss = -torch.log(pos + EPS).mean() - torch.log(1 - negs + EPS).mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print("{}: training step {:.3f} ms (forward + 2 decoder calls + backward + Adam), loss {:.4f}".format(
        args.workload, 1e3 * dt, float(loss)))


if __name__ == "__main__":
    main()
