#!/usr/bin/env python3
"""Per-wave anatomy of the register-accumulated relational kernel (development tool).

    make -C gripnet_amd/csrc STAMPS=1 && GN_HIP_LIBRARY=$PWD/gripnet_amd/lib/libgripnet_hip_stamps.so \
        python tools/acc_stamps.py --workload pose0-syn

Reads the stamps the diagnostic build leaves behind: 100 MHz real-time clock at kernel entry, after the
table fill, after the unit loop and at exit; shader cycles spent in the gather loops and in the MFMA chains.
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
    buf = np.zeros((4096, 8), dtype=np.uint64)
    lib.gn_debug_read_acc_stamps.argtypes = [C.c_void_p]
    assert lib.gn_debug_read_acc_stamps(buf.ctypes.data) == 0
    b = buf[buf[:, 0] > 0].astype(np.float64)
    t0 = b[:, 0].min()
    us = lambda v: v / 100.0
    print("waves {}".format(len(b)))
    print("entry spread {:.1f} us; fill done at {:.1f} (mean) {:.1f} (max); loop done at {:.1f} (mean) {:.1f} (max); exit at {:.1f} (max)".format(
        us(b[:, 0].max() - t0), us(b[:, 1].mean() - t0), us(b[:, 1].max() - t0), us(b[:, 2].mean() - t0), us(b[:, 2].max() - t0),
        us(b[:, 3].max() - t0)))
    loop = us(b[:, 2] - b[:, 1])
    print("unit loop per wave: min {:.1f} mean {:.1f} p90 {:.1f} max {:.1f} us".format(loop.min(), loop.mean(), np.percentile(loop, 90), loop.max()))
    print("units per wave mean {:.2f} max {:.0f}; blocks per wave mean {:.1f} max {:.0f}".format(b[:, 6].mean(), b[:, 6].max(), b[:, 7].mean(), b[:, 7].max()))
    print("gather cycles per wave mean {:.0f} (per block {:.0f}); mfma cycles per wave mean {:.0f} (per non-empty tile ~{:.0f})".format(
        b[:, 4].mean(), b[:, 4].sum() / max(b[:, 7].sum(), 1), b[:, 5].mean(), b[:, 5].sum() / max(b[:, 6].sum() * 3, 1)))
    full = buf[: (len(buf[buf[:, 0] > 0]) // 12) * 12].astype(np.float64)
    if len(full) == len(b):                                      # 12 waves per workgroup, in launch order
        wg = us(full[:, 2] - full[:, 1]).reshape(-1, 12)
        fill = us(full[:, 1] - t0).reshape(-1, 12)[:, 0]
        blocks = full[:, 7].reshape(-1, 12)
        tiles = full[:, 6].reshape(-1, 12) * 3
        print("per workgroup: loop mean over waves min {:.1f} mean {:.1f} max {:.1f}; spread inside a workgroup (max - mean) mean {:.1f} max {:.1f}".format(
            wg.mean(1).min(), wg.mean(1).mean(), wg.mean(1).max(), (wg.max(1) - wg.mean(1)).mean(), (wg.max(1) - wg.mean(1)).max()))
        print("fill done per workgroup: min {:.1f} mean {:.1f} max {:.1f}".format(fill.min(), fill.mean(), fill.max()))
        A = np.stack([blocks.ravel(), tiles.ravel(), np.ones(blocks.size)], 1)
        coef, *_ = np.linalg.lstsq(A, wg.ravel(), rcond=None)
        res = wg.ravel() - A @ coef
        print("fit loop_us = {:.4f} * blocks + {:.4f} * tiles + {:.2f}; residual rms {:.2f} us, max {:.2f}".format(coef[0], coef[1], coef[2], np.sqrt((res ** 2).mean()), np.abs(res).max()))
        resw = res.reshape(-1, 12).mean(1)
        print("residual by workgroup: rms {:.2f}, worst {:.2f} us; by eight-way interleave (XCD) mean {}".format(
            np.sqrt((resw ** 2).mean()), resw.max(), np.round([resw[i::8].mean() for i in range(8)], 2)))
        order = np.argsort(-wg.max(1))[:8]
        for o in order:
            print("  wg {:3d}: loop max {:.1f} mean {:.1f} blocks/wave {:.0f}..{:.0f} tiles {:.0f}..{:.0f} fill {:.1f}".format(
                o, wg[o].max(), wg[o].mean(), blocks[o].min(), blocks[o].max(), tiles[o].min(), tiles[o].max(), fill[o]))
    cyc = b[:, 4] + b[:, 5]
    print("stamped cycles / loop time: implied clock x share = {:.2f} GHz-equivalent".format((cyc.sum() / (loop.sum() * 1e3))))


if __name__ == "__main__":
    main()
