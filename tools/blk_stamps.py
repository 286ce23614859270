#!/usr/bin/env python3
"""Phase times of the source-blocked GCN kernel k_blk_partial (development tool).

    make -C gripnet_amd/csrc STAMPS=1 && GN_HIP_LIBRARY=$PWD/gripnet_amd/lib/libgripnet_hip_stamps.so \
        python tools/blk_stamps.py

Reads the stamps the diagnostic build leaves behind per workgroup: 100 MHz real-time clock at kernel entry,
after the table fill (barrier) and at exit.  Set 0 = the 32 -> 16 layer, set 1 = the 16 -> 16 layer.
"""
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
    dev = torch.device("cuda:0")
    data = make_pose("pose0-syn").to(dev)
    torch.manual_seed(1111)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
    with torch.no_grad():
        for _ in range(5):
            model.gg(None, data.gg_edge_index, edge_weight=data.edge_weight, if_catout=True)
    torch.cuda.synchronize()
    lib = _hip.load()
    buf = np.zeros((2, 512, 4), dtype=np.uint64)
    lib.gn_debug_read_blk_stamps.argtypes = [C.c_void_p]
    assert lib.gn_debug_read_blk_stamps(buf.ctypes.data) == 0
    for s in range(2):
        b = buf[s][buf[s][:, 0] > 0].astype(np.float64)
        if not len(b):
            continue
        t0 = b[:, 0].min()
        us = lambda v: v / 100.0
        print("layer set {}: {} workgroups; entry spread {:.2f} us; fill done at mean {:.2f} max {:.2f}; exit at mean {:.2f} max {:.2f}; "
              "fill per wg mean {:.2f} max {:.2f}; gather per wg mean {:.2f} max {:.2f}; ".format(
                  s, len(b), us(b[:, 0].max() - t0), us(b[:, 1].mean() - t0), us(b[:, 1].max() - t0), us(b[:, 2].mean() - t0),
                  us(b[:, 2].max() - t0), us((b[:, 1] - b[:, 0]).mean()), us((b[:, 1] - b[:, 0]).max()),
                  us((b[:, 2] - b[:, 1]).mean()), us((b[:, 2] - b[:, 1]).max())))


if __name__ == "__main__":
    main()
