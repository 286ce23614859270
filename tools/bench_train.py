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
    ap.add_argument("--fused-adam", action="store_true", help="torch.optim.Adam(fused=True): the same update in one launch instead of ~6")
    ap.add_argument("--host-negatives", dest="sampler", action="store_false",
                    help="one fixed negative list instead of the default: new negatives drawn on the device every step "
                         "(gn_negative_sampler_sample_packed: the decoder then scores them from 32-bit pairs; eager steps only, the seed is a launch argument)")
    ap.add_argument("--graph", action="store_true", help="capture the whole step (forward, loss, backward, Adam) in one hipGraph and replay it")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    data = make_pose(args.workload).to(dev)
    torch.manual_seed(1111)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=args.graph, fused=True if args.fused_adam else None)
    neg = torch.randint(0, data.n_d_node, tuple(data.train_idx.shape), device=dev)

    # new negative pairs every step, as the reference draws them every epoch (GripNet-pose.py:131): only the positive
    # edge list is static, and only it gets the decoder's forward / backward plans.  (Copied into the buffer the
    # step reads, outside a captured step.)
    pool = [torch.randint(0, data.n_d_node, tuple(data.train_idx.shape), device=dev) for _ in range(4)]
    drawn = [0]

    sampler = None
    if args.sampler and not args.graph:
        from gripnet_amd._hip import NegativeSampler
        # typed (per relation block): on the synthetic ladder the 2 M positives of all relations together cover nearly every
        # one of the 645^2 pairs, so the untyped draw of GripNet-pose.py:131 would reject almost for ever
        sampler = NegativeSampler(data.train_idx, data.n_d_node, data.train_range)

    def resample():
        nonlocal neg
        if sampler is not None:
            neg = sampler.sample(seed=drawn[0])
        else:
            neg.copy_(pool[drawn[0] % len(pool)])
        drawn[0] += 1

    def step():
        opt.zero_grad()
        z = model.encode(data)
        pos = model.dmt(z, data.train_idx, data.train_et)
        negs = model.dmt(z, neg, data.train_et)
        loss = -torch.log(pos + EPS).mean() - torch.log(1 - negs + EPS).mean()
        loss.backward()
        opt.step()
        return loss

    if args.graph:
        # every plan, the relation-order check and the lazily created optimizer state exist after the warm-up steps;
        # from then on a step is a fixed sequence of launches on fixed buffers
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                resample()            # (a buffer that stayed the same would be taken for a static list and get plans,
                step()                #  and a captured step would replay them on new contents)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        resample()
        with torch.cuda.graph(graph):
            static_loss = step()
        eager_step = step

        def step():
            graph.replay()
            return static_loss
    train_step = step

    def step():
        resample()
        return train_step()
    for _ in range(3):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print("{}: training step {:.3f} ms (forward + 2 decoder calls + backward + Adam{}), loss {:.4f}".format(
        args.workload, 1e3 * dt, "; one hipGraph replay per step" if args.graph else "", float(loss)))


if __name__ == "__main__":
    main()
