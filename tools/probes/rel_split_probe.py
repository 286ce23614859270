#!/usr/bin/env python3
"""How the relational layer's time splits between the hub relations and the many small ones (development probe):
the layer alone on pose0-syn restricted to the relations above / below a size rank, default kernel and GN_RGCN_TF=1."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gripnet_amd
from gripnet_amd.synth import make_pose
from gripnet_amd.utils import get_range_list

dev = torch.device("cuda:0")
d = make_pose("pose0-syn")
cut = int(sys.argv[1]) if len(sys.argv) > 1 else 134
rl = d.train_range
blocks = [d.train_idx[:, int(rl[r, 0]):int(rl[r, 1])] for r in range(rl.shape[0])]


def run(label, sel):
    ei = torch.cat([blocks[r] for r in sel], dim=1).to(dev)
    ranges = get_range_list([blocks[r] for r in sel])
    torch.manual_seed(1)
    conv = gripnet_amd.myRGCN(48, 32, len(sel), 32, False).to(dev)
    x = torch.randn(d.n_d_node, 48, device=dev)
    out = torch.empty(d.n_d_node, 32, device=dev)
    with torch.no_grad():
        for _ in range(5):
            conv(x, ei, None, ranges, _out=out, _relu=True)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            conv(x, ei, None, ranges, _out=out, _relu=True)
        e.record(); torch.cuda.synchronize()
    print("{:28s} relations {:4d} edges {:8d}: {:6.1f} us per call (weights + kernel + finalisation)".format(
        label, len(sel), ei.shape[1], s.elapsed_time(e) / 50 * 1e3))


R = rl.shape[0]
run("all", list(range(R)))
run("hubs (rank < {})".format(cut), list(range(cut)))
run("small (rank >= {})".format(cut), list(range(cut, R)))
