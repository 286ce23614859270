#!/usr/bin/env python3
"""The headline's timed region and nothing else (for tools/prof.sh step): build the pose0-syn model and its plans, record the
step's entry-point calls (pipeline.Recorded - the launch mode bench.py's `value` is measured in), W warm-up steps, fence,
K steps, fence.  Prints one line `STEP_ONLY {json}` with the wall time per step of those K steps; the profiler's trace of the
last K steps' kernels is what tools/summarize_step.py sums against it."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gripnet_amd import _hip                       # noqa: E402
from gripnet_amd.pipeline import PoseModel, PoseStages   # noqa: E402
from gripnet_amd.synth import make_pose            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="pose0-syn")
    ap.add_argument("--launch", default="recorded", choices=["recorded", "eager"])
    ap.add_argument("--spin-up", type=int, default=0, help="un-timed steps in front of the warm-up that bring the device's clocks up "
                    "(an A / B of kernels in the steady state: 4000; under the profiler: 0 - they would all be traced)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    data = make_pose(args.workload).to(dev)
    torch.manual_seed(1111)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
    with torch.no_grad():
        stages = PoseStages(model, data, graphs=False, recorded=args.launch == "recorded")
        for _ in range(args.spin_up + args.warmup):
            stages.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            stages.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    _hip.raise_if_index_errors(dev)
    print("STEP_ONLY " + json.dumps({"workload": args.workload, "launch": args.launch, "steps": args.steps, "spin_up": args.spin_up,
                                     "us_per_step_wall": round(1e6 * dt / args.steps, 2)}))


if __name__ == "__main__":
    main()
