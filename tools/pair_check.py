#!/usr/bin/env python3
"""Parity and timing of the destination-major relational kernel (rgcn_pair.hip) against the fp64 oracle and the
previous kernels (GN_DISABLE_PAIR=1 in a child process)."""
import os, sys, time, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gripnet_amd
from gripnet_amd.synth import make_pose
from oracle import gripnet_oracle as orc

dev = torch.device("cuda:0")

def shapes():
    out = []
    for n, fin, bases in [(40, 16, 3), (200, 32, 5), (560, 48, 32), (645, 64, 8), (645, 48, 32), (700, 32, 17), (1, 16, 1), (33, 48, 32)]:
        gen = torch.Generator().manual_seed(n * 131 + fin)
        torch.manual_seed(n * 17 + fin)
        sizes = [0, 9000, 3, 0, 700, 1, 2500, 0]
        blocks = [torch.randint(0, max(1, n - n // 7), (2, s), generator=gen) for s in sizes]
        blocks[4] = torch.cat([blocks[4], blocks[4][:, :50]], dim=1)
        rei = torch.cat(blocks, dim=1)
        rl = gripnet_amd.utils.get_range_list(blocks)
        x = torch.randn(n, fin, generator=gen)
        rg = gripnet_amd.myRGCN(fin, 32, len(sizes), bases, False, bias=True).to(dev)
        rg.bias.data.normal_()
        y = rg(x.to(dev), rei.to(dev), None, rl, _relu=True)
        y2 = rg(x.to(dev), rei.to(dev), None, rl, _relu=True)
        sd = {k: v.detach().cpu().double() for k, v in rg.state_dict().items()}
        ref = torch.relu(orc.rgcn_forward(x.double(), rei, rl, sd["basis"], sd["att"], sd["root"], sd.get("bias")))
        err = (y.cpu().double() - ref).abs().max().item()
        out.append((n, fin, bases, err, bool(torch.equal(y, y2))))
        print("shape n={} fin={} bases={}: max err {:.3e} reproducible {}".format(n, fin, bases, err, torch.equal(y, y2)), flush=True)
    return out

def pose(name="pose0-syn", iters=50):
    data = make_pose(name).to(dev)
    n, R = data.n_d_node, data.n_dd_edge_type
    torch.manual_seed(1111)
    conv = gripnet_amd.myRGCN(48, 32, R, 32, False).to(dev)
    x = torch.randn(n, 48, device=dev).abs()
    with torch.no_grad():
        t0 = time.time()
        y = conv(x, data.train_idx, data.train_et, data.train_range, _relu=True)
        torch.cuda.synchronize()
        plan_s = time.time() - t0
        sd = {k: v.detach().cpu().double() for k, v in conv.state_dict().items()}
        ref = torch.relu(orc.rgcn_forward(x.cpu().double(), data.train_idx.cpu(), data.train_range, sd["basis"], sd["att"], sd["root"], None))
        err = (y.cpu().double() - ref).abs().max().item()
        ref32 = torch.relu(orc.rgcn_forward(x.cpu(), data.train_idx.cpu(), data.train_range, sd["basis"].float(), sd["att"].float(), sd["root"].float(), None))
        err32 = (ref32.double() - ref).abs().max().item()
        for _ in range(5):
            conv(x, data.train_idx, data.train_et, data.train_range, _relu=True)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
        ev[0].record()
        for i in range(iters):
            conv(x, data.train_idx, data.train_et, data.train_range, _relu=True)
            ev[i + 1].record()
        torch.cuda.synchronize()
        us = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(iters))
    print("{}: |y - fp64 oracle| {:.3e} (the fp32 oracle itself: {:.3e}); first call {:.2f} s; layer call {:.1f} us median, {:.1f} min".format(
        name, err, err32, plan_s, us[len(us) // 2], us[0]), flush=True)

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    print("GN_DISABLE_PAIR =", os.environ.get("GN_DISABLE_PAIR"), "GN_ACC_EXACT =", os.environ.get("GN_ACC_EXACT"), flush=True)
    if what in ("all", "shapes"):
        shapes()
    if what in ("all", "pose"):
        for name in (sys.argv[2:] or ["pose0-syn", "pose1-syn", "pose2-syn"]):
            pose(name)
