"""The general relational kernel on the two parts of the all-nodes baseline's graph separately (layer-1 shapes 64 -> 32, 16 bases):
the 645 hub rows (drug-drug edges only) and the gene rows (gene-gene + gene-drug edges only) - where does the launch's time go?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gripnet_amd
from gripnet_amd import _hip
from gripnet_amd.synth import make_rgcn_pose

dev = torch.device("cuda:0")
data = make_rgcn_pose("pose0-syn")
R, N = int(data.n_edge_type), int(data.n_node)
rl = data.train_range
E = int(rl[-1, 1])
cut = int(rl[R - 2, 0])                                 # the last two relations are gene-gene and gene-drug


def layer(fin, fout, idx, ranges, label):
    torch.manual_seed(3)
    conv = gripnet_amd.myRGCN(fin, fout, R, 16, False).to(dev)
    for p in conv.parameters():
        p.requires_grad_(False)
    conv.kernel = "general"
    x = torch.randn(N, fin, device=dev)
    with torch.no_grad():
        for _ in range(3):
            conv(x, idx, None, ranges)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            conv(x, idx, None, ranges)
        b.record()
        torch.cuda.synchronize()
    print("{:28s} {:3d} -> {:2d}: {:7.1f} us per layer ({} edges)".format(label, fin, fout, 1e2 * a.elapsed_time(b), idx.shape[1]))


idx = data.train_idx.to(dev)
for fin in (64, 32):
    layer(fin, 32, idx, rl, "all rows")
    hub_ranges = rl.clone()
    hub_ranges[R - 2:] = torch.tensor([[cut, cut], [cut, cut]])
    layer(fin, 32, idx[:, :cut].contiguous(), hub_ranges, "hub rows only (drug-drug)")
    gene_ranges = torch.zeros_like(rl)
    gene_ranges[R - 2:] = rl[R - 2:] - cut
    layer(fin, 32, idx[:, cut:].contiguous(), gene_ranges, "gene rows only")
