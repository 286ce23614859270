#!/usr/bin/env python3
"""Phase anatomy of the LDS-resident relational kernel (development tool).

    make -C gripnet_amd/csrc STAMPS=1 && GN_HIP_LIBRARY=gripnet_amd/lib/libgripnet_hip_stamps.so \
        python tools/rgcn_stamps.py --workload pose0-syn

Reads the per-workgroup s_memtime sums the diagnostic build leaves behind (wave 0 of every
workgroup): shader cycles per phase, items per workgroup, start/end on the 100 MHz real-time clock.
Shares only; the stamped build is slower than the product build.
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
        for _ in range(3):
            conv(x, data.train_idx, data.train_et, data.train_range, _out=out, _relu=True)
    torch.cuda.synchronize()
    lib = _hip.load()
    buf = np.zeros((512, 8), dtype=np.uint64)
    lib.gn_debug_read_stamps.argtypes = [C.c_void_p]
    rc = lib.gn_debug_read_stamps(buf.ctypes.data)
    assert rc == 0, rc
    b = buf.astype(np.float64)
    names = ["top(loads issue)", "barrier A", "mfma+ebuf", "barrier B", "gather"]
    tot = b[:, :5].sum(axis=1)
    print("items per workgroup: min {:.0f} mean {:.1f} max {:.0f}".format(b[:, 5].min(), b[:, 5].mean(), b[:, 5].max()))
    for k, n in enumerate(names):
        print("{:18s} mean {:9.0f} cyc/wg  ({:5.1f} %)   per item {:7.0f}   min {:8.0f} max {:8.0f}".format(
            n, b[:, k].mean(), 100 * b[:, k].sum() / tot.sum(), b[:, k].sum() / b[:, 5].sum(), b[:, k].min(), b[:, k].max()))
    print("stamped loop cycles per wg: min {:.0f} mean {:.0f} max {:.0f}".format(tot.min(), tot.mean(), tot.max()))
    wb = np.zeros((512, 16, 4), dtype=np.uint64)
    lib.gn_debug_read_wave_stamps.argtypes = [C.c_void_p]
    assert lib.gn_debug_read_wave_stamps(wb.ctypes.data) == 0
    w = wb.astype(np.float64)
    for k, n in enumerate(["gather", "top", "trips", "mfma+ebuf"]):
        per_wave = w[:, :, k].mean(axis=0)
        print("per-wave {:10s} (mean over workgroups, per item): ".format(n) + " ".join("{:6.0f}".format(v / b[:, 5].mean()) for v in per_wave))
    print("per-workgroup max-over-waves gather per item: mean {:.0f}".format((w[:, :, 0].max(axis=1) / b[:, 5]).mean()))
    t0 = b[:, 6].min()
    dur = (b[:, 7] - b[:, 6]) / 100.0
    print("wg duration us (100 MHz clock): min {:.1f} mean {:.1f} max {:.1f}; first start -> last end {:.1f} us; start spread {:.1f} us".format(
        dur.min(), dur.mean(), dur.max(), (b[:, 7].max() - t0) / 100.0, (b[:, 6].max() - t0) / 100.0))


if __name__ == "__main__":
    main()
