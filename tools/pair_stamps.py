#!/usr/bin/env python3
"""Per-wave anatomy of the destination-major relational kernel (development tool).

    make -C gripnet_amd/csrc STAMPS=1 && GN_HIP_LIBRARY=$PWD/gripnet_amd/lib/libgripnet_hip_stamps.so \
        python tools/pair_stamps.py --workload pose0-syn
"""
import argparse, ctypes as C, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gripnet_amd import _hip
import gripnet_amd
from gripnet_amd.synth import make_pose

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="pose0-syn")
    ap.add_argument("--planes", action="store_true", help="x arrives as bf16 split planes (gn_split_planes)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    data = make_pose(args.workload).to(dev)
    torch.manual_seed(1111)
    conv = gripnet_amd.myRGCN(48, 32, data.n_dd_edge_type, 32, False).to(dev)
    x = torch.randn(data.n_d_node, 48, device=dev)
    if args.planes:
        _hip.SplitPlanes(data.n_d_node, 3, dev).fill_from(x).tag(x)
    with torch.no_grad():
        for _ in range(5):
            conv(x, data.train_idx, data.train_et, data.train_range, _relu=True)
    torch.cuda.synchronize()
    lib = _hip.load()
    buf = np.zeros((4096, 12), dtype=np.uint64)
    lib.gn_debug_read_pair_stamps.argtypes = [C.c_void_p]
    assert lib.gn_debug_read_pair_stamps(buf.ctypes.data) == 0
    b = buf[buf[:, 0] > 0].astype(np.float64)
    t0 = b[:, 0].min()
    us = lambda v: v / 100.0
    print("waves", len(b))
    print("entry spread {:.1f} us; prologue done at {:.1f} mean {:.1f} max; loop done at {:.1f} mean {:.1f} max; exit {:.1f} mean {:.1f} max".format(
        us(b[:, 0].max() - t0), us(b[:, 1].mean() - t0), us(b[:, 1].max() - t0), us(b[:, 2].mean() - t0), us(b[:, 2].max() - t0),
        us(b[:, 3].mean() - t0), us(b[:, 3].max() - t0)))
    loop = us(b[:, 2] - b[:, 1])
    units = (buf[buf[:, 0] > 0][:, 7] >> np.uint64(32)).astype(np.float64)
    blocks = (buf[buf[:, 0] > 0][:, 7] & np.uint64(0xffffffff)).astype(np.float64)
    print("loop per wave: min {:.1f} mean {:.1f} p90 {:.1f} max {:.1f} us; epilogue (exit - loop end of the slowest wave of the kernel) {:.1f} us".format(
        loop.min(), loop.mean(), np.percentile(loop, 90), loop.max(), us(b[:, 3].max() - b[:, 2].max())))
    print("units per wave mean {:.2f} max {:.0f}; blocks per wave mean {:.0f} max {:.0f}".format(units.mean(), units.max(), blocks.mean(), blocks.max()))
    print("cycles per wave: gather {:.0f} ({:.1f} per block), contract {:.0f} ({:.0f} per unit), x chunks {:.0f}".format(
        b[:, 4].mean(), b[:, 4].sum() / blocks.sum(), b[:, 5].mean(), b[:, 5].sum() / max(units.sum(), 1), b[:, 6].mean()))
    e = lambda k: us(b[:, k].mean() - t0)
    print("epilogue (mean over waves): barrier passed + shares written {:.1f}; shares summed {:.1f}; contraction done {:.1f}; exit {:.1f}".format(e(8), e(9), e(10), e(3)))
    # per workgroup (16 consecutive waves): when its slowest wave left the loop, and the epilogue's stages behind that
    nw = len(b) // 16 * 16
    g = b[:nw].reshape(-1, 16, b.shape[1])
    loop_end = g[:, :, 2].max(1)
    st = [us(g[:, :, k].max(1) - loop_end) for k in (8, 9, 10, 3)]
    late = loop_end >= np.percentile(loop_end, 60)                 # the workgroups on the critical path (three rows)
    for name, sel in (("all workgroups", np.ones(len(loop_end), bool)), ("the 40 % that leave the loop last", late)):
        print("epilogue behind the workgroup's slowest wave, {}: loop end {:.1f} us; barrier + basis rows here {:.1f}; shares summed {:.1f}; "
              "contraction done {:.1f}; exit {:.1f}".format(name, us(loop_end[sel].mean() - t0), *[x[sel].mean() for x in st]))
    cyc = b[:, 4] + b[:, 5] + b[:, 6]
    print("stamped cycles / loop time = {:.2f} GHz".format(cyc.sum() / (loop.sum() * 1e3)))
    A = np.stack([blocks, units, np.ones(len(b))], 1)
    coef, *_ = np.linalg.lstsq(A, loop, rcond=None)
    print("fit loop_us = {:.4f} * blocks + {:.3f} * units + {:.2f}".format(*coef))

if __name__ == "__main__":
    main()
