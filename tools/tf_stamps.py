#!/usr/bin/env python3
"""Per-wave anatomy of the transform-then-gather relational kernel (development tool).

    make -C gripnet_amd/csrc STAMPS=1 && GN_HIP_LIBRARY=$PWD/gripnet_amd/lib/libgripnet_hip_stamps.so \
        python tools/tf_stamps.py --workload pose0-syn

Stamps of the diagnostic build: 100 MHz real-time clock at kernel entry, after the prologue, after the slice loop and
at exit; shader cycles spent in the MFMA phases, in the gather phases and waiting at the two barriers of a slice.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gripnet_amd import _hip                      # noqa: E402
from gripnet_amd.pipeline import PoseModel        # noqa: E402
from gripnet_amd.synth import make_pose           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="pose0-syn")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    data = make_pose(args.workload).to(dev)
    torch.manual_seed(1111)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
    conv = model.dd.conv_list[0]
    x = torch.randn(data.n_d_node, 48, device=dev)
    out = torch.empty(data.n_d_node, 32, device=dev)
    with torch.no_grad():
        for _ in range(5):
            conv(x, data.train_idx, data.train_et, data.train_range, _out=out, _relu=True)
    torch.cuda.synchronize()
    lib = _hip.load()
    buf = np.zeros((4096, 12), dtype=np.uint64)
    lib.gn_debug_read_tf_stamps.argtypes = [C.c_void_p]
    assert lib.gn_debug_read_tf_stamps(buf.ctypes.data) == 0
    b = buf[buf[:, 0] > 0].astype(np.float64)
    t0 = b[:, 0].min()
    us = lambda v: v / 100.0
    print("waves {}".format(len(b)))
    print("entry spread {:.1f} us; prologue done at {:.1f} (mean) {:.1f} (max); loop done at {:.1f} (mean) {:.1f} (max); exit at {:.1f} (mean) {:.1f} (max)".format(
        us(b[:, 0].max() - t0), us(b[:, 7].mean() - t0), us(b[:, 7].max() - t0), us(b[:, 1].mean() - t0), us(b[:, 1].max() - t0),
        us(b[:, 2].mean() - t0), us(b[:, 2].max() - t0)))
    loop = us(b[:, 1] - b[:, 7])
    print("slice loop per wave: min {:.1f} mean {:.1f} p90 {:.1f} max {:.1f} us".format(loop.min(), loop.mean(), np.percentile(loop, 90), loop.max()))
    cyc = b[:, 3] + b[:, 4] + b[:, 5]
    print("slices per workgroup mean {:.2f} max {:.0f}".format(b[:, 6].mean(), b[:, 6].max()))
    print("cycles per wave: mfma {:.0f}  gather {:.0f}  barriers {:.0f}   (per slice: {:.0f} / {:.0f} / {:.0f})".format(
        b[:, 3].mean(), b[:, 4].mean(), b[:, 5].mean(), b[:, 3].sum() / b[:, 6].sum(), b[:, 4].sum() / b[:, 6].sum(), b[:, 5].sum() / b[:, 6].sum()))
    n = b[:, 6].sum()
    print("per slice and wave: mfma {:.0f}  gather {:.0f}  wait for DMA {:.0f}  barrier A {:.0f}  store H {:.0f}  barrier B {:.0f}".format(
        b[:, 3].sum() / n, b[:, 4].sum() / n, b[:, 8].sum() / n, b[:, 9].sum() / n, b[:, 10].sum() / n, b[:, 5].sum() / n))
    cyc = b[:, 3] + b[:, 4] + b[:, 5] + b[:, 8] + b[:, 9] + b[:, 10]
    print("stamped cycles / loop time = {:.2f} GHz-equivalent".format(cyc.sum() / (loop.sum() * 1e3)))
    W = int(os.environ.get("GN_TF_WAVES", "16"))
    if len(b) % W == 0:
        wg = loop.reshape(-1, W)[:, 0]
        order = np.argsort(-wg)[:6]
        g = b.reshape(-1, W, 12)
        for o in order:
            print("  wg {:3d}: loop {:.1f} us, slices {:.0f}, gather cycles per wave {:.0f}..{:.0f}, mfma {:.0f}..{:.0f}".format(
                o, wg[o], g[o, 0, 6], g[o, :, 4].min(), g[o, :, 4].max(), g[o, :, 3].min(), g[o, :, 3].max()))


if __name__ == "__main__":
    main()
