"""LDS-staged GCN-style layers (csrc/gcn_blocked.hip) against the CPU oracle.

The path serves graphs whose stored weights are all 1 (every GripNet caller,
GripNet-pose.py:52,117-120); GN_BLOCKED_ANY=1 lifts its size thresholds so that small graphs, ragged
tiles, hub rows, ranges with a single tile and both column-group widths are all exercised here.
"""
import os

import pytest
import torch

import gripnet_amd
from gripnet_amd import _hip
from oracle import gripnet_oracle as orc

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("GN_DISABLE_FAST") == "1" or os.environ.get("GN_DISABLE_BLOCKED") == "1",
                                 reason="the source-blocked path is switched off")]

TIGHT = 2e-5


def close(a, b, atol=TIGHT):
    a, b = torch.as_tensor(a).detach().cpu(), torch.as_tensor(b).detach().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all()
    err = (a.double() - b.double()).abs().max().item() if a.numel() else 0.0
    assert err <= atol, "max abs err {:.3e} > {:.1e}".format(err, atol)


@pytest.fixture
def any_size(monkeypatch):
    monkeypatch.setenv("GN_BLOCKED_ANY", "1")


def graph(n, e, seed, loops=0, hub=False):
    g = torch.Generator().manual_seed(seed)
    a = torch.randint(0, n, (2, e), generator=g)
    if hub:                                   # one node receives a quarter of the edges, a range of nodes none
        a[1, : e // 4] = n // 3
        a[1][a[1] >= n - 40] = 0
    a = a[:, a[0] != a[1]]
    ei = torch.cat([a, a.flip(0)], dim=1)
    if loops:
        l = torch.randint(0, n, (loops,), generator=g)
        ei = torch.cat([ei, torch.stack([l, l])], dim=1)
    return ei.long()


CASES = [
    # n,     edges, fin, fout, hub,   loops
    (50,     200,   32,  16,   False, 0),
    (333,    4000,  16,  16,   True,  5),      # ragged last tile, existing unit self loops
    (6000,   90000, 32,  16,   False, 0),
    (6000,   90000, 16,  16,   True,  0),
    (3001,   40000, 64,  32,   False, 0),      # 16 column groups
    (25000,  300000, 32, 16,   True,  0),      # one column per group: the table of 2-column groups would not fit the LDS
    (2000,   30000, 64,  16,   False, 3),
    (1500,   20000, 32,  32,   True,  0),
]


@pytest.mark.parametrize("n,e,fin,fout,hub,loops", CASES)
def test_blocked_layer_vs_oracle(gpu, any_size, n, e, fin, fout, hub, loops):
    torch.manual_seed(n + fin)
    ei = graph(n, e, seed=n + e, loops=loops, hub=hub)
    conv = gripnet_amd.myGCN(fin, fout, cached=True)
    conv.bias.data.normal_()
    x = torch.randn(n, fin)
    ref = torch.relu(orc.gcn_forward(x, conv.weight.data, conv.bias.data, ei, None))
    conv = conv.to(gpu)
    # the layer writes its slot of a wider concat buffer and copies the input into slot 0, as homoGraph does
    buf = torch.full((n, fin + fout), float("nan"), device=gpu)
    with torch.no_grad():
        y = conv(x.to(gpu), ei.to(gpu), None, _out=buf[:, fin:], _relu=True, _side=(x.to(gpu), buf[:, :fin], 0))
    assert conv.cached_result.blocked_cols >= fout, "the plan did not get its LDS-staged encoding"
    close(y, ref)
    close(buf[:, :fin], x, 0.0)
    with torch.no_grad():
        again = conv(x.to(gpu), ei.to(gpu), None, _relu=True)
    assert torch.equal(again, y), "not bitwise reproducible"


def test_blocked_equals_wave_per_row_kernels(gpu, any_size, monkeypatch):
    """Same layer on both gather paths."""
    ei = graph(5000, 120000, seed=3).to(gpu)
    conv = gripnet_amd.myGCN(32, 16, cached=False).to(gpu)
    x = torch.randn(5000, 32, device=gpu)
    with torch.no_grad():
        a = conv(x, ei, None, _relu=True)
        monkeypatch.setenv("GN_DISABLE_BLOCKED", "1")
        b = conv(x, ei, None, _relu=True)
    close(a, b)


def test_blocked_table_given(gpu, any_size):
    """gn_graph_aggregate_f32 without a weight: the table x W is given and only scaled on its way into LDS."""
    n, f = 4100, 16
    ei = graph(n, 60000, seed=9)
    plan = _hip.GraphPlan.gcn(ei.to(gpu), n)
    assert plan.build_blocked(f) == 16
    xw = torch.randn(n, f)
    bias = torch.randn(f)
    ref = orc.gcn_forward(xw, torch.eye(f), bias, ei, None)
    out = torch.empty(n, f, device=gpu)
    plan.aggregate(xw.to(gpu), bias.to(gpu), False, out)
    close(out, ref)


def test_weighted_and_improved_graphs_keep_the_general_kernels(gpu, any_size):
    n = 500
    ei = graph(n, 6000, seed=4).to(gpu)
    w = torch.rand(ei.shape[1], device=gpu) + 0.5
    assert _hip.GraphPlan.gcn(ei, n, w).build_blocked(16) == 0
    assert _hip.GraphPlan.gcn(ei, n, None, improved=True).build_blocked(16) == 0
    assert _hip.GraphPlan.gcn(ei, n, torch.ones(ei.shape[1], device=gpu)).build_blocked(16) == 16
    conv = gripnet_amd.myGCN(32, 16, cached=True)
    x = torch.randn(n, 32)
    ref = orc.gcn_forward(x, conv.weight.data, conv.bias.data, ei.cpu(), w.cpu())
    with torch.no_grad():
        close(conv.to(gpu)(x.to(gpu), ei, w), ref)


def test_small_graphs_are_not_blocked_by_default(gpu):
    ei = graph(300, 3000, seed=5).to(gpu)
    assert _hip.GraphPlan.gcn(ei, 300).build_blocked(16) == 0
