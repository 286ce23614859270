#!/usr/bin/env python3
"""dev tool: random graphs and widths through myGCN (default kernel choice) against the float64 oracle."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gripnet_amd
from oracle import gripnet_oracle as orc

dev = torch.device("cuda:0")
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    n = rnd.choice([1, 50, 700, 4095, 4096, 5000, 20000, 33000])
    fin = rnd.choice([16, 24, 32, 64, 128, 256])
    fout = rnd.choice([8, 16, 20, 32, 64, 80, 128])
    deg = rnd.choice([0, 1, 3, 8, 11, 40, 70])
    gen = torch.Generator().manual_seed(case * 31 + n)
    e = n * deg
    ei = torch.randint(0, max(1, n - n // 9), (2, e), generator=gen)
    if e > 200:
        ei[1, :min(e, 3000)] = 3 % n                       # a hub
    w = (torch.rand(e, generator=gen) + 0.1) if case % 3 else None
    x = torch.randn(n, fin, generator=gen)
    torch.manual_seed(case)
    conv = gripnet_amd.myGCN(fin, fout).to(dev)
    conv.bias.data.normal_()
    with torch.no_grad():
        y = conv(x.to(dev), ei.to(dev), None if w is None else w.to(dev), _relu=bool(case & 1)).cpu().double()
    ref = orc.gcn_forward(x.double(), conv.weight.detach().cpu().double(), conv.bias.detach().cpu().double(), ei, None if w is None else w.double())
    if case & 1:
        ref = torch.relu(ref)
    err = (y - ref).abs().max().item() if y.numel() else 0.0
    scale = max(1.0, ref.abs().max().item() if ref.numel() else 1.0)
    ok = err <= 2e-5 * scale
    print("case {:2d} n={:5d} fin={:3d} fout={:3d} deg={:2d} w={} err={:.2e} scale={:.1f} {}".format(case, n, fin, fout, deg, w is not None, err, scale, "ok" if ok else "FAIL"))
    assert ok
