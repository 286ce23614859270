#!/usr/bin/env python3
"""dev tool: random shapes through both decoders and the external layer against the float64 oracle; the planned decoder
(second sighting of a list) must give the bits of the first call."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gripnet_amd
from gripnet_amd import _hip
from oracle import gripnet_oracle as orc

dev = torch.device("cuda:0")
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    n = rnd.choice([1, 2, 12, 300, 645, 1000, 3000, 9000])
    f = rnd.choice([4, 8, 16, 32, 48, 80, 96, 128, 20])
    R = rnd.choice([1, 2, 7, 50])
    gen = torch.Generator().manual_seed(case * 13 + n)
    sizes = [rnd.choice([0, 1, 3, 64, 65, 700, 5000]) for _ in range(R)]
    bidir = bool(case & 1)
    blocks = []
    for s_ in sizes:
        b = torch.randint(0, n, (2, s_), generator=gen)
        blocks.append(torch.cat([b, b.flip(0)], dim=1) if bidir else b)
    rei = torch.cat(blocks, dim=1)
    et = torch.cat([torch.full((b.shape[1],), r, dtype=torch.long) for r, b in enumerate(blocks)])
    z = torch.randn(n, f, generator=gen)
    torch.manual_seed(case)
    dm = gripnet_amd.multiRelaInnerProductDecoder(f, R).to(dev)
    sig = bool(case & 2)
    zg, eg, tg = z.to(dev), rei.to(dev), et.to(dev)
    with torch.no_grad():
        a = dm(zg, eg, tg, sigmoid=sig)
        b = dm(zg, eg, tg, sigmoid=sig)          # second sighting: the planned kernel
        c = dm(zg, eg, tg, sigmoid=sig)
    ref = orc.distmult(z.double(), rei, et, dm.weight.detach().cpu().double(), sigmoid=sig)
    err = (a.cpu().double() - ref).abs().max().item() if a.numel() else 0.0
    same = torch.equal(a, b) and torch.equal(b, c)
    ok = err <= 2e-5 * max(1.0, ref.abs().max().item() if ref.numel() else 1.0) and same
    print("decoder case {:2d} n={:5d} f={:3d} R={:2d} E={:6d} bidir={} sigmoid={} err={:.2e} same_bits={} {}".format(
        case, n, f, R, rei.shape[1], bidir, sig, err, same, "ok" if ok else "FAIL"))
    assert ok
    # class decoder
    k = rnd.choice([4, 30, 32, 128, 288, 416]); ncls = rnd.choice([1, 2, 8, 16, 17, 40]); m = rnd.choice([1, 5, 1000, 4097])
    zz = torch.randn(max(n, 2), k, generator=gen); nodes = torch.randint(0, max(n, 2), (m,), generator=gen)
    mc = gripnet_amd.multiClassInnerProductDecoder(k, ncls).to(dev)
    with torch.no_grad():
        p = mc(zz.to(dev), nodes.to(dev), softmax=bool(case & 1))
    r2 = zz[nodes].double() @ mc.weight.detach().cpu().double()
    if case & 1:
        r2 = torch.softmax(r2, dim=1)
    e2 = (p.cpu().double() - r2).abs().max().item()
    print("   class k={:3d} classes={:2d} m={:4d} err={:.2e}".format(k, ncls, m, e2))
    assert e2 <= 2e-5 * max(1.0, r2.abs().max().item())
