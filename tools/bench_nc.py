#!/usr/bin/env python3
"""Secondary measurement: forward of the aminer-style NC model (BASELINE.json configs[2]) on `aminer-syn`.

    python tools/bench_nc.py [--model aminer|freebase-c]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gripnet_amd.pipeline import AminerModel, FreebaseCModel      # noqa: E402
from gripnet_amd.synth import make_nc                             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="aminer")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--storage", default="fp32", choices=("fp32", "bf16"), help="storage of the gathered tables")
    ap.add_argument("--graph", action="store_true", help="replay the forward as one hipGraph instead of launching it from Python")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    data = make_nc("aminer-syn").to(dev)
    torch.manual_seed(1111)
    if args.model == "aminer":
        model = AminerModel(data.n_p_node, data.n_a_node, data.n_a_type).to(dev)
    else:
        model = FreebaseCModel(data.n_p_node, data.n_q_node, data.n_a_node, data.n_a_type).to(dev)
    from gripnet_amd.utils import set_table_storage
    set_table_storage(model, args.storage)
    nodes = torch.arange(0, data.n_a_node, 2, device=dev)
    with torch.no_grad():
        step = lambda: model(data, nodes)
        for _ in range(3):
            step()
        if args.graph:
            from gripnet_amd.pipeline import Graphed
            step = Graphed(step).capture()
            step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            step()
        b.record()
        torch.cuda.synchronize()
    edges = 2 * (data.pp_edge_idx.shape[1] + data.n_p_node) + data.pa_edge_idx.shape[1] + 2 * (data.aa_edge_idx.shape[1] + data.n_a_node)
    us = 1e3 * a.elapsed_time(b) / args.iters
    print("{} forward on aminer-syn, {} tables, {}: {:.1f} us  ({:.3e} edges aggregated/s)".format(args.model, args.storage, "one hipGraph" if args.graph else "eager", us, edges / us * 1e6))


if __name__ == "__main__":
    main()
