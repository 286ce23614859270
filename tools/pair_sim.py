#!/usr/bin/env python3
"""Plan-level simulation of the destination-major relational kernel (rgcn_pair.hip).

Formulation: out_i * deg_i = sum_b (sum_s P_i[s, b] x[s, :]) basis[b],  P_i[s, :] = sum over the edges s -> i of att[r(e), :].
A wave gathers att rows (LDS) for four (dst, src) slots in lock step (one per 16-lane group of the MFMA operand layout),
`blk` edges per block.  Prints padded gather steps against the ideal, for several block sizes."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from gripnet_amd.synth import make_pose

def sim(name):
    d = make_pose(name)
    n = d.n_d_node
    src, dst = d.train_idx[0].numpy(), d.train_idx[1].numpy()
    E = src.size
    cnt = np.zeros((n, n), dtype=np.int64)          # [dst, src]
    np.add.at(cnt, (dst, src), 1)
    # global source order: by out-degree, descending (lock-step partners get similar expectations)
    order = np.argsort(-cnt.sum(0), kind="stable")
    kpad = (n + 31) // 32 * 32
    c = np.zeros((n, kpad), dtype=np.int64)
    c[:, :n] = cnt[:, order]
    c = c.reshape(n, kpad // 32, 4, 8)               # dst, chunk, kg, t
    mx = c.max(axis=2)                               # lock step over the four groups
    print("{}: n={} E={} pairs non-empty {:.1%} mean {:.2f} edges/pair; chunks/dst {}".format(
        name, n, E, (cnt > 0).mean(), E / n / n, kpad // 32))
    for blk in (1, 2, 4, 8):
        steps = (np.ceil(mx / blk) * blk).sum()
        solo = (np.ceil(c / blk) * blk).sum() / 4
        print("  block {}: lock-step steps {:.0f} = {:.2f}x ideal (E/4 = {:.0f}); without lock step {:.2f}x".format(
            blk, steps, steps / (E / 4), E / 4, solo / (E / 4)))
    per_dst = (np.ceil(mx / 4) * 4).sum(axis=(1, 2))
    print("  per-dst steps (block 4): min {:.0f} mean {:.0f} max {:.0f}".format(per_dst.min(), per_dst.mean(), per_dst.max()))

for w in sys.argv[1:] or ["pose0-syn", "pose2-syn", "small"]:
    sim(w)
