#!/usr/bin/env python3
"""dev tool: random shapes through the relational layer (default kernel choice) against the float64 oracle."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gripnet_amd
from oracle import gripnet_oracle as orc

dev = torch.device("cuda:0")
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
worst = 0.0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    n = rnd.choice([1, 2, 17, 64, 255, 256, 257, 400, 645, 700, 768, 769, 1000])
    fin = rnd.choice([16, 32, 48, 64])
    fout = rnd.choice([4, 8, 16, 20, 32, 44, 48, 64])
    bases = rnd.choice([1, 2, 5, 8, 16, 17, 32])
    R = rnd.choice([1, 3, 8, 40])
    gen = torch.Generator().manual_seed(case * 7919 + n)
    sizes = [rnd.choice([0, 1, 5, 300, 2500, 9000]) for _ in range(R)]
    blocks = [torch.randint(0, n, (2, s), generator=gen) for s in sizes]
    rei = torch.cat(blocks, dim=1)
    rl = gripnet_amd.utils.get_range_list(blocks)
    x = torch.randn(n, fin, generator=gen)
    torch.manual_seed(case)
    rg = gripnet_amd.myRGCN(fin, fout, R, bases, False, bias=True).to(dev)
    rg.bias.data.normal_()
    with torch.no_grad():
        y = rg(x.to(dev), rei.to(dev), None, rl, _relu=bool(case & 1)).cpu().double()
        y2 = rg(x.to(dev), rei.to(dev), None, rl, _relu=bool(case & 1)).cpu().double()
    sd = {k: v.detach().cpu().double() for k, v in rg.state_dict().items()}
    ref = orc.rgcn_forward(x.double(), rei, rl, sd["basis"], sd["att"], sd["root"], sd.get("bias"))
    if case & 1:
        ref = torch.relu(ref)
    err = (y - ref).abs().max().item() if y.numel() else 0.0
    worst = max(worst, err)
    path = rg._plan.path(fin, fout, bases)
    ok = err <= 2e-5 and torch.equal(y, y2)
    print("case {:2d} n={:4d} fin={:2d} fout={:2d} bases={:2d} R={:2d} E={:6d} path={:7s} err={:.2e} {}".format(
        case, n, fin, fout, bases, R, rei.shape[1], path, err, "ok" if ok else "FAIL"))
    assert ok
print("worst", worst)
