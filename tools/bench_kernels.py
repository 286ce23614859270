#!/usr/bin/env python3
"""Per-entry-point device timing on the synthetic PoSE workloads (development tool).

    python tools/bench_kernels.py --workload pose0-syn --what distmult,rgcn,gcn,full --iters 50

One HIP-event pair brackets `iters` back-to-back launches of the same entry point, so the
number is the steady-state device time per launch (no per-launch event overhead).
Prints one line per entry point: avg us, algorithmic GB/s (SURVEY.md 8d byte model), fraction of 8 TB/s.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gripnet_amd import _hip                      # noqa: E402
from gripnet_amd.pipeline import PoseModel        # noqa: E402
from gripnet_amd.synth import make_pose           # noqa: E402


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="pose0-syn")
    ap.add_argument("--what", default="distmult,rgcn,gcn,full")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--shuffle-types", action="store_true", help="DistMult on a shuffled edge list (unsorted relation ids)")
    ap.add_argument("--arith", default="fp32", choices=["fp32", "fast"], help="arithmetic of the relational layer")
    ap.add_argument("--rgcn-kernel", default="auto", help="relational kernel (auto, pair, lds, general)")
    ap.add_argument("--flush", type=int, default=0, help="MB of unrelated data streamed between launches (cold L2, as inside the forward)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    data = make_pose(args.workload).to(dev)
    torch.manual_seed(1111)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
    what = set(args.what.split(","))
    if args.flush:
        junk = torch.empty(args.flush << 18, dtype=torch.float32, device=dev)
        global timed
        plain_timed = timed

        def timed(fn, iters, warm=3):                      # time fn alone, with the flush between launches
            def both():
                junk.add_(1.0)
                fn()
            return plain_timed(both, iters, warm) - plain_timed(lambda: junk.add_(1.0), iters, warm)
    E = data.train_idx.shape[1]
    n_d, R = data.n_d_node, data.n_dd_edge_type
    with torch.no_grad():
        z, score = model(data)
        torch.cuda.synchronize()

        def report(name, us, nbytes):
            gbs = nbytes / us / 1e3
            print("{:28s} {:9.2f} us   {:8.1f} GB/s algorithmic   {:.3f} of 8 TB/s".format(name, us, gbs, gbs / 8000))

        if "distmult" in what:
            idx, et = data.train_idx, data.train_et
            if args.shuffle_types:
                perm = torch.randperm(E, device=dev)
                idx, et = idx[:, perm].contiguous(), et[perm].contiguous()
            us = timed(lambda: model.dmt(z, idx, et), args.iters)
            report("distmult E={}".format(E), us, E * 28 + n_d * 80 * 4 + R * 80 * 4)
        if "rgcn" in what:
            conv = model.dd.conv_list[0]
            conv.arithmetic, conv.kernel = args.arith, args.rgcn_kernel
            x = z[:, :48].contiguous()
            out = torch.empty(n_d, 32, device=dev)
            us = timed(lambda: conv(x, data.train_idx, data.train_et, data.train_range, _out=out, _relu=True), args.iters)
            report("rgcn E={}".format(E), us, E * 16 + n_d * 4 * 80 + 4 * (32 * 48 * 32 + R * 32 + 48 * 32))
        if "gcn" in what:
            conv = model.gg.conv_list[1]
            h = torch.randn(data.n_g_node, 16, device=dev)
            out = torch.empty(data.n_g_node, 16, device=dev)
            us = timed(lambda: conv(h, data.gg_edge_index, data.edge_weight, _out=out, _relu=True), args.iters)
            e1 = data.gg_edge_index.shape[1] + data.n_g_node
            report("gcn layer (gemm+aggregate)", us, e1 * 20 + data.n_g_node * 4 * 32)
            xw = torch.randn(data.n_g_node, 16, device=dev)
            plan = conv.cached_result
            us = timed(lambda: plan.aggregate(xw, conv.bias, True, out), args.iters)
            report("gcn aggregate only", us, e1 * 20 + data.n_g_node * 4 * 32)
        if "full" in what:
            us = timed(lambda: model(data), args.iters)
            from gripnet_amd.synth import pose_edges_aggregated
            print("{:28s} {:9.2f} us   {:.3e} edges aggregated/s".format("full forward (eager)", us,
                                                                      pose_edges_aggregated(data) / us * 1e6))
    _hip.raise_if_index_errors(dev)


if __name__ == "__main__":
    main()
