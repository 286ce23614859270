#!/usr/bin/env python3
"""dev tool: gradients of the relational and GCN layers on random shapes against torch autograd through the float64 oracle."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gripnet_amd
from oracle import gripnet_oracle as orc

dev = torch.device("cuda:0")
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    n = rnd.choice([2, 17, 200, 256, 300, 645, 700, 900])
    fin = rnd.choice([16, 32, 48, 64, 24])
    fout = rnd.choice([8, 16, 20, 32, 48, 64])
    bases = rnd.choice([1, 4, 16, 32])
    R = rnd.choice([1, 3, 9])
    gen = torch.Generator().manual_seed(case * 101 + n)
    sizes = [rnd.choice([0, 2, 400, 3000]) for _ in range(R)]
    blocks = [torch.randint(0, n, (2, s), generator=gen) for s in sizes]
    rei = torch.cat(blocks, dim=1)
    rl = gripnet_amd.utils.get_range_list(blocks)
    x = torch.randn(n, fin, generator=gen)
    wgt = torch.randn(n, fout, generator=gen)
    torch.manual_seed(case)
    rg = gripnet_amd.myRGCN(fin, fout, R, bases, False, bias=True).to(dev)
    rg.bias.data.normal_()
    xg = x.to(dev).requires_grad_(True)
    y = torch.relu(rg(xg, rei.to(dev), None, rl))
    (y * wgt.to(dev)).sum().backward()
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in rg.state_dict().items()}
    xr = x.double().requires_grad_(True)
    yr = torch.relu(orc.rgcn_forward(xr, rei, rl, sd["basis"], sd["att"], sd["root"], sd["bias"]))
    (yr * wgt.double()).sum().backward()
    worst = 0.0
    for name, g, r in (("x", xg.grad, xr.grad), ("basis", rg.basis.grad, sd["basis"].grad), ("att", rg.att.grad, sd["att"].grad),
                       ("root", rg.root.grad, sd["root"].grad), ("bias", rg.bias.grad, sd["bias"].grad)):
        scale = max(1.0, r.abs().max().item())
        err = (g.cpu().double() - r).abs().max().item() / scale
        worst = max(worst, err)
        assert err <= 5e-5, (name, err, n, fin, fout, bases, R)
    print("rgcn grad case {:2d} n={:3d} fin={:2d} fout={:2d} bases={:2d} R={} E={:5d} worst rel err {:.2e} ok".format(case, n, fin, fout, bases, R, rei.shape[1], worst))
