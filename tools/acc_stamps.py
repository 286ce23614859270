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
    cyc = b[:, 4] + b[:, 5]
    print("stamped cycles / loop time: implied clock x share = {:.2f} GHz-equivalent".format((cyc.sum() / (loop.sum() * 1e3))))


if __name__ == "__main__":
    main()
