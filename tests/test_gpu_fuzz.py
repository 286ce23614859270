"""Seeded property tests over random graphs and widths: every layer of the hot path on its
default kernel choice against the float64 oracle, run-to-run equality where the kernels promise it, and the shapes at
the kernel-selection thresholds (768 / 769 nodes of the destination-major relational kernel, 64 input features with more
than 16 bases, the 4,096-node threshold of the LDS-staged gene layers, 16 / 17 classes, 64 / 65-edge relation blocks).
Bounded to about a minute on one MI355X; the generators are the former tools/fuzz_*.py."""
import random

import pytest
import torch

import gripnet_amd
from gripnet_amd import _hip
from oracle import gripnet_oracle as orc

pytestmark = pytest.mark.gpu


def _rel_err(y, ref):
    if not ref.numel():
        return 0.0
    return (y - ref).abs().max().item() / max(1.0, ref.abs().max().item())


# ---- GCN-style layer (myGCN, gripnet/layers.py:52-100) -----------------------------------------------------------------
GCN_EDGE_CASES = [(4095, 16, 16, 16), (4096, 16, 16, 16), (4096, 32, 16, 17), (5000, 64, 32, 20), (1, 16, 8, 0),
                  (700, 24, 20, 3), (20000, 32, 16, 70), (33000, 16, 16, 8)]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_gcn_layer(gpu, seed):
    rnd = random.Random(seed)
    shapes = list(GCN_EDGE_CASES) if seed == 1 else []
    while len(shapes) < 30:
        shapes.append((rnd.choice([1, 50, 700, 4095, 4096, 5000, 20000]), rnd.choice([16, 24, 32, 64, 128, 256]),
                       rnd.choice([8, 16, 20, 32, 64, 80, 128]), rnd.choice([0, 1, 3, 8, 11, 40])))
    for case, (n, fin, fout, deg) in enumerate(shapes):
        gen = torch.Generator().manual_seed(seed * 1000 + case * 31 + n)
        e = n * deg
        ei = torch.randint(0, max(1, n - n // 9), (2, e), generator=gen)
        if e > 200:
            ei[1, :min(e, 3000)] = 3 % n                       # a hub
        w = (torch.rand(e, generator=gen) + 0.1) if case % 3 else None
        x = torch.randn(n, fin, generator=gen)
        torch.manual_seed(seed * 100 + case)
        conv = gripnet_amd.myGCN(fin, fout, cached=bool(case & 2)).to(gpu)      # cached: the LDS-staged plan where the graph qualifies
        conv.bias.data.normal_()
        with torch.no_grad():
            y = conv(x.to(gpu), ei.to(gpu), None if w is None else w.to(gpu), _relu=bool(case & 1)).cpu().double()
        ref = orc.gcn_forward(x.double(), conv.weight.detach().cpu().double(), conv.bias.detach().cpu().double(), ei,
                              None if w is None else w.double())
        if case & 1:
            ref = torch.relu(ref)
        assert _rel_err(y, ref) <= 2e-5, (seed, case, n, fin, fout, deg)


# ---- relational layer (myRGCN, gripnet/layers.py:165-197) --------------------------------------------------------------
RGCN_EDGE_CASES = [(768, 48, 32, 32, 8), (769, 48, 32, 32, 8), (645, 64, 32, 17, 3), (645, 64, 32, 16, 3), (256, 16, 4, 1, 1),
                   (257, 32, 44, 32, 40), (1, 16, 8, 1, 1), (1000, 48, 64, 5, 8)]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_relational_layer(gpu, seed):
    rnd = random.Random(seed)
    shapes = list(RGCN_EDGE_CASES) if seed == 1 else []
    while len(shapes) < 30:
        shapes.append((rnd.choice([1, 2, 17, 64, 255, 256, 257, 400, 645, 700, 768, 769, 1000]), rnd.choice([16, 32, 48, 64]),
                       rnd.choice([4, 8, 16, 20, 32, 44, 48, 64]), rnd.choice([1, 2, 5, 8, 16, 17, 32]), rnd.choice([1, 3, 8, 40])))
    for case, (n, fin, fout, bases, R) in enumerate(shapes):
        gen = torch.Generator().manual_seed(seed * 1000 + case * 7919 + n)
        sizes = [rnd.choice([0, 1, 5, 300, 2500, 9000]) for _ in range(R)]
        blocks = [torch.randint(0, n, (2, s), generator=gen) for s in sizes]
        rei = torch.cat(blocks, dim=1).to(gpu)
        rl = gripnet_amd.utils.get_range_list(blocks)
        x = torch.randn(n, fin, generator=gen).to(gpu)
        torch.manual_seed(seed * 100 + case)
        rg = gripnet_amd.myRGCN(fin, fout, R, bases, False, bias=True).to(gpu)
        rg.bias.data.normal_()
        if case % 3 == 0 and fin % 16 == 0:                     # x arrives with the bf16 split planes of its producer
            _hip.SplitPlanes(n, fin // 16, gpu).fill_from(x).tag(x)
        with torch.no_grad():
            y = rg(x, rei, None, rl, _relu=bool(case & 1))
            y2 = rg(x, rei, None, rl, _relu=bool(case & 1))
        assert torch.equal(y, y2), (seed, case, n, fin, fout, bases, R)
        sd = {k: v.detach().cpu().double() for k, v in rg.state_dict().items()}
        ref = orc.rgcn_forward(x.cpu().double(), rei.cpu(), rl, sd["basis"], sd["att"], sd["root"], sd.get("bias"))
        if case & 1:
            ref = torch.relu(ref)
        err = (y.cpu().double() - ref).abs().max().item() if y.numel() else 0.0
        assert err <= 2e-5, (seed, case, n, fin, fout, bases, R, rg._plan.path(fin, fout, bases), err)


# ---- decoders (gripnet/decoder.py:19-23,38-45) -------------------------------------------------------------------------
DEC_EDGE_CASES = [(645, 80, 50), (480, 80, 7), (481, 80, 7), (9000, 16, 2), (12, 128, 1), (2, 4, 1), (1000, 96, 50), (300, 20, 7)]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_decoders(gpu, seed):
    rnd = random.Random(seed)
    shapes = list(DEC_EDGE_CASES) if seed == 1 else []
    while len(shapes) < 30:
        shapes.append((rnd.choice([1, 2, 12, 300, 645, 1000, 3000, 9000]), rnd.choice([4, 8, 16, 32, 48, 80, 96, 128, 20]),
                       rnd.choice([1, 2, 7, 50])))
    for case, (n, f, R) in enumerate(shapes):
        gen = torch.Generator().manual_seed(seed * 1000 + case * 13 + n)
        sizes = [rnd.choice([0, 1, 3, 64, 65, 700, 5000]) for _ in range(R)]
        bidir = bool(case & 1)
        blocks = []
        for s_ in sizes:
            b = torch.randint(0, n, (2, s_), generator=gen)
            blocks.append(torch.cat([b, b.flip(0)], dim=1) if bidir else b)
        rei = torch.cat(blocks, dim=1)
        et = torch.cat([torch.full((b.shape[1],), r, dtype=torch.long) for r, b in enumerate(blocks)])
        z = torch.randn(n, f, generator=gen)
        torch.manual_seed(seed * 100 + case)
        dm = gripnet_amd.multiRelaInnerProductDecoder(f, R).to(gpu)
        sig = bool(case & 2)
        zg, eg, tg = z.to(gpu), rei.to(gpu), et.to(gpu)
        with torch.no_grad():
            a = dm(zg, eg, tg, sigmoid=sig)
            b = dm(zg, eg, tg, sigmoid=sig)          # second sighting: the planned kernels (row classes or column phases)
            c = dm(zg, eg, tg, sigmoid=sig)
        ref = orc.distmult(z.double(), rei, et, dm.weight.detach().cpu().double(), sigmoid=sig)
        assert _rel_err(a.cpu().double(), ref) <= 2e-5, (seed, case, n, f, R)
        assert torch.equal(a, b) and torch.equal(b, c), ("planned decoder differs from the plan-less bits", seed, case, n, f, R)
        # class decoder
        k, ncls, m = rnd.choice([4, 30, 32, 128, 288, 416]), rnd.choice([1, 2, 8, 16, 17, 40]), rnd.choice([1, 5, 1000, 4097])
        zz = torch.randn(max(n, 2), k, generator=gen)
        nodes = torch.randint(0, max(n, 2), (m,), generator=gen)
        mc = gripnet_amd.multiClassInnerProductDecoder(k, ncls).to(gpu)
        with torch.no_grad():
            p = mc(zz.to(gpu), nodes.to(gpu), softmax=bool(case & 1))
        r2 = zz[nodes].double() @ mc.weight.detach().cpu().double()
        if case & 1:
            r2 = torch.softmax(r2, dim=1)
        assert _rel_err(p.cpu().double(), r2) <= 2e-5, (seed, case, k, ncls, m)


# ---- gradients of the relational layer (autograd of GripNet-pose.py:140-146 over layers.py:172-186) -------------------
@pytest.mark.parametrize("seed", [1])
def test_fuzz_relational_gradients(gpu, seed):
    rnd = random.Random(seed)
    for case in range(20):
        n = rnd.choice([2, 17, 200, 256, 300, 645, 700, 900])
        fin = rnd.choice([16, 32, 48, 64, 24])
        fout = rnd.choice([8, 16, 20, 32, 48, 64])
        bases = rnd.choice([1, 4, 16, 32])
        R = rnd.choice([1, 3, 9])
        gen = torch.Generator().manual_seed(seed * 1000 + case * 101 + n)
        sizes = [rnd.choice([0, 2, 400, 3000]) for _ in range(R)]
        blocks = [torch.randint(0, n, (2, s), generator=gen) for s in sizes]
        rei = torch.cat(blocks, dim=1)
        rl = gripnet_amd.utils.get_range_list(blocks)
        x = torch.randn(n, fin, generator=gen)
        wgt = torch.randn(n, fout, generator=gen)
        torch.manual_seed(seed * 100 + case)
        rg = gripnet_amd.myRGCN(fin, fout, R, bases, False, bias=True).to(gpu)
        rg.bias.data.normal_()
        xg = x.to(gpu).requires_grad_(True)
        y = torch.relu(rg(xg, rei.to(gpu), None, rl))
        (y * wgt.to(gpu)).sum().backward()
        sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in rg.state_dict().items()}
        xr = x.double().requires_grad_(True)
        yr = torch.relu(orc.rgcn_forward(xr, rei, rl, sd["basis"], sd["att"], sd["root"], sd["bias"]))
        (yr * wgt.double()).sum().backward()
        for name, g, r in (("x", xg.grad, xr.grad), ("basis", rg.basis.grad, sd["basis"].grad), ("att", rg.att.grad, sd["att"].grad),
                           ("root", rg.root.grad, sd["root"].grad), ("bias", rg.bias.grad, sd["bias"].grad)):
            assert _rel_err(g.cpu().double(), r) <= 5e-5, (name, seed, case, n, fin, fout, bases, R)
