#!/usr/bin/env python3
"""Anatomy of the fused relational weight gradient (development tool; make -C gripnet_amd/csrc STAMPS=1, then
GN_HIP_LIBRARY=$PWD/gripnet_amd/lib/libgripnet_hip_stamps.so python tools/rel_stamps.py)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gripnet_amd import _hip                      # noqa: E402
from gripnet_amd.synth import make_pose           # noqa: E402

dev = torch.device("cuda:0")
d = make_pose(sys.argv[1] if len(sys.argv) > 1 else "pose0-syn").to(dev)
n, R = d.n_d_node, d.n_dd_edge_type
torch.manual_seed(3)
x, gm = torch.randn(n, 48, device=dev), torch.randn(n, 32, device=dev)
wg = _hip.RgcnPlan(d.train_idx, d.train_range, n).weight_grad_plan()
for _ in range(5):
    wg.weight_grad(x, gm)
torch.cuda.synchronize()
lib = _hip.load()
st, wv = np.zeros((256, 8), dtype=np.uint64), np.zeros((256, 16), dtype=np.uint64)
lib.gn_debug_read_rel_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert lib.gn_debug_read_rel_stamps(st.ctypes.data, wv.ctypes.data) == 0
b = st.astype(np.float64) / 100.0
t0 = b[:, 0].min()
for k, name in [(0, "entry"), (1, "table filled"), (2, "first wave out of the last entry"), (3, "last wave out of the last entry"),
                (6, "last entry: first tile's sums in LDS"), (7, "last entry: both tiles summed"), (4, "exit")]:
    v = b[:, k] - t0
    print("{:34s} mean {:6.1f} us  min {:6.1f}  max {:6.1f}".format(name, v.mean(), v.min(), v.max()))
w = wv.astype(np.float64) / 100.0
print("loop time per wave (all entries): mean {:.1f} us, per-workgroup max: mean {:.1f} min {:.1f} max {:.1f}".format(
    w.mean(), w.max(axis=1).mean(), w.max(axis=1).min(), w.max(axis=1).max()))
print("entries per workgroup: min {} mean {:.1f} max {}".format(int(st[:, 5].min()), st[:, 5].mean(), int(st[:, 5].max())))
worst = np.argsort(b[:, 4])[-5:]
for g in worst:
    print("wg {:3d}: exit {:6.1f} entries {:2d} waves' loop {}".format(int(g), b[g, 4] - t0, int(st[g, 5]), " ".join("{:.0f}".format(v) for v in w[g])))
