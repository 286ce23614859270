#!/usr/bin/env python3
"""Phase anatomy of the planned DistMult kernel (development tool; make -C gripnet_amd/csrc STAMPS=1, then
GN_HIP_LIBRARY=$PWD/gripnet_amd/lib/libgripnet_hip_stamps.so python tools/dm_stamps.py [--flush MB])."""
import argparse, ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gripnet_amd import _hip                      # noqa: E402
from gripnet_amd.pipeline import PoseModel        # noqa: E402
from gripnet_amd.synth import make_pose           # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--flush", type=int, default=64); args = ap.parse_args()
dev = torch.device("cuda:0")
data = make_pose("pose0-syn").to(dev)
torch.manual_seed(1111)
model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
junk = torch.empty(max(args.flush, 1) << 18, device=dev)
with torch.no_grad():
    z, _ = model(data)
    for _ in range(4):
        if args.flush: junk.add_(1.0)
        model.dmt(z, data.train_idx, data.train_et)
torch.cuda.synchronize()
lib = _hip.load()
buf = np.zeros((256, 12), dtype=np.uint64)
lib.gn_debug_read_dm_stamps.argtypes = [C.c_void_p]
assert lib.gn_debug_read_dm_stamps(buf.ctypes.data) == 0
b = buf.astype(np.float64) / 100.0
t0 = b[:, 0].min()
names = ["entry", "ph0 barrier", "ph0 filled", "ph0 done", "ph1 barrier", "ph1 filled", "ph1 done"]
if b[:, 4].max() == 0:                            # the row-class kernel stamps entry / table filled / wave 0 done / last wave done
    names = ["entry", "filled", "wave 0 done", "last wave done"]
for k, n in enumerate(names):
    print("{:12s} mean {:6.1f} us  min {:6.1f}  max {:6.1f}".format(n, (b[:, k] - t0).mean(), (b[:, k] - t0).min(), (b[:, k] - t0).max()))
