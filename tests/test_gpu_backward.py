"""Gradients of the HIP path (autograd Functions over the C ABI) against torch autograd through the CPU
oracle (the backward pass GripNet-pose.py:140-146 needs).

Same seeded inputs and weights on both sides; the loss is a fixed random projection of the output so
that every output element carries a different gradient.  Bar: 1e-4 abs on every gradient (the
reference's own tolerance for fp32); checks use 2e-5 unless stated.
"""
import os

import pytest
import torch

import gripnet_amd
from gripnet_amd import _hip
from gripnet_amd.pipeline import AminerModel, FreebaseCModel
from gripnet_amd.synth import make_nc
from oracle import gripnet_oracle as orc

pytestmark = pytest.mark.gpu
needs_fast_paths = pytest.mark.skipif(os.environ.get("GN_DISABLE_FAST") == "1",
                                      reason="exercises a fast path that GN_DISABLE_FAST=1 turns off")
TIGHT = 2e-5


def close(a, b, atol=TIGHT, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.isfinite(a).all(), what
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert err <= atol, "{}: max abs err {:.3e} > {:.1e}".format(what, err, atol)


def leaf(t):
    return t.detach().clone().requires_grad_(True)


def test_gcn_conv_gradients(gpu):
    gen = torch.Generator().manual_seed(5)
    n, fin, fout, e = 300, 24, 16, 4000
    ei = torch.randint(0, n, (2, e), generator=gen)
    ew = torch.rand(e, generator=gen) + 0.1
    x = torch.randn(n, fin, generator=gen)
    proj = torch.randn(n, fout, generator=gen)
    torch.manual_seed(9)
    conv = gripnet_amd.myGCN(fin, fout).to(gpu)
    conv.bias.data.normal_()
    for relu in (False, True):
        xg = leaf(x.to(gpu))
        conv.zero_grad()
        y = conv(xg, ei.to(gpu), ew.to(gpu), _relu=relu)
        (y * proj.to(gpu)).sum().backward()
        xr, wr, br = leaf(x), leaf(conv.weight.cpu()), leaf(conv.bias.cpu())
        yr = orc.gcn_forward(xr, wr, br, ei, ew)
        yr = torch.relu(yr) if relu else yr
        (yr * proj).sum().backward()
        close(y, yr, what="forward")
        close(xg.grad, xr.grad, what="dx relu={}".format(relu))
        close(conv.weight.grad, wr.grad, 1e-4, what="dW")          # K = 300 rows summed in a different order
        close(conv.bias.grad, br.grad, 1e-4, what="db")


@pytest.mark.parametrize("mod,tfd,one_ext", [("cat", 20, True), ("add", 12, True), ("add", 20, True), ("cat", 20, False)])
def test_inter_graph_gradients(gpu, mod, tfd, one_ext):
    gen = torch.Generator().manual_seed(11)
    n_src, n_tgt, fin, fout, e = 200, 60, 18, 12, 900
    ei = torch.stack([torch.randint(0, n_src, (e,), generator=gen), torch.randint(0, n_tgt - 5, (e,), generator=gen)])
    x = torch.randn(n_src, fin, generator=gen)
    torch.manual_seed(13)
    m = gripnet_amd.interGraph(fin, fout, n_tgt, target_feat_dim=tfd, if_one_external=one_ext).to(gpu)
    m.conv.bias.data.normal_()
    xg = leaf(x.to(gpu))
    y = m(xg, ei.to(gpu), if_relu=True, mod=mod)
    proj = torch.randn(y.shape, generator=gen)
    (y * proj.to(gpu)).sum().backward()
    sd = {"g." + k: leaf(v.cpu()) for k, v in m.state_dict().items()}
    xr = leaf(x)
    yr = orc.inter_forward(sd, "g.", xr, ei, None, if_relu=True, mod=mod, n_target=n_tgt)
    (yr * proj).sum().backward()
    close(y, yr, what="forward")
    close(xg.grad, xr.grad, what="dx")
    for k, p in m.named_parameters():
        ref = sd["g." + k].grad
        if ref is None:                                       # parameter not on this branch (e.g. target_feat_down)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0
        else:
            close(p.grad, ref, 1e-4, what=k)


def test_homo_graph_gcn_gradients(gpu):
    gen = torch.Generator().manual_seed(17)
    n, e = 150, 1200
    a = torch.randint(0, n, (2, e), generator=gen)
    ei = torch.cat([a, a.flip(0)], dim=1)
    torch.manual_seed(19)
    m = gripnet_amd.homoGraph([10, 8, 6], start_graph=True, in_dim=n).to(gpu)
    for c in m.conv_list:
        c.bias.data.normal_()
    y = m(None, ei.to(gpu), None, if_catout=True)
    proj = torch.randn(y.shape, generator=gen)
    (y * proj.to(gpu)).sum().backward()
    sd = {"h." + k: leaf(v.cpu()) for k, v in m.state_dict().items()}
    yr = orc.homo_forward(sd, "h.", None, ei, None, if_catout=True)
    (yr * proj).sum().backward()
    close(y, yr, what="forward")
    for k, p in m.named_parameters():
        close(p.grad, sd["h." + k].grad, 1e-4, what=k)


@pytest.mark.parametrize("tables", ["lds", "l2"])
@pytest.mark.parametrize("sigmoid", [True, False])
@pytest.mark.parametrize("n,f,shuffle", [(100, 80, False), (645, 80, False), (37, 20, True), (3000, 24, True)])
def test_distmult_gradients(gpu, sigmoid, n, f, shuffle, tables, monkeypatch):
    """Both reductions: factor tables in LDS (small supervertex; (3000, 24) does not fit and takes the other one
    anyway) and gathered from L2 behind three sorts; relation ids sorted (no sort for dD) and shuffled."""
    if tables == "l2":
        monkeypatch.setenv("GN_DISABLE_FAST", "1")
    gen = torch.Generator().manual_seed(23 + n)
    R, e = 7, 5000
    ei = torch.randint(0, n, (2, e), generator=gen)
    ei[:, :50] = ei[:, 50:100]                                # repeated edges
    ei[1, 100:120] = ei[0, 100:120]                           # u == v
    et = torch.sort(torch.randint(0, R, (e,), generator=gen)).values
    if shuffle:
        et = et[torch.randperm(e, generator=gen)]
    z = torch.randn(n, f, generator=gen) * 0.5
    proj = torch.randn(e, generator=gen)
    torch.manual_seed(29)
    dm = gripnet_amd.multiRelaInnerProductDecoder(f, R).to(gpu)
    zg = leaf(z.to(gpu))
    s = dm(zg, ei.to(gpu), et.to(gpu), sigmoid=sigmoid)
    (s * proj.to(gpu)).sum().backward()
    zr, wr = leaf(z), leaf(dm.weight.cpu())
    sr = orc.distmult(zr, ei, et, wr, sigmoid=sigmoid)
    (sr * proj).sum().backward()
    close(s, sr, what="forward")
    close(zg.grad, zr.grad, 1e-4, what="dz")
    close(dm.weight.grad, wr.grad, 1e-4, what="dD")
    first = (zg.grad.clone(), dm.weight.grad.clone())         # no atomics on floats: the same bits every time
    zg.grad = None
    dm.weight.grad = None
    s = dm(zg, ei.to(gpu), et.to(gpu), sigmoid=sigmoid)
    (s * proj.to(gpu)).sum().backward()
    assert torch.equal(zg.grad, first[0]) and torch.equal(dm.weight.grad, first[1])


def test_multiclass_decoder_gradients(gpu):
    gen = torch.Generator().manual_seed(31)
    n, f, c = 90, 40, 6
    z = torch.randn(n, f, generator=gen)
    nodes = torch.randint(0, n, (70,), generator=gen)
    torch.manual_seed(37)
    mc = gripnet_amd.multiClassInnerProductDecoder(f, c).to(gpu)
    for softmax in (True, False):
        mc.zero_grad()
        zg = leaf(z.to(gpu))
        p = mc(zg, nodes.to(gpu), softmax=softmax)
        proj = torch.randn(p.shape, generator=gen)
        (p * proj.to(gpu)).sum().backward()
        zr, wr = leaf(z), leaf(mc.weight.cpu())
        pr = orc.multiclass(zr, nodes, wr, softmax=softmax)
        (pr * proj).sum().backward()
        close(p, pr, what="forward")
        close(zg.grad, zr.grad, what="dz")
        close(mc.weight.grad, wr.grad, what="dW")


def test_multiclass_decoder_node_lists_that_change_every_step(gpu):
    """The backward's row-gather plan (autograd.node_gather_plan) is remembered under the CALLER's list: a strided list that is
    the same object every step is made contiguous on every forward and still builds ONE plan; lists that are new every step stop
    getting plans after two misses (a build synchronises the stream and walks the list on the host) and take the plan-less
    gather / scatter - the same gradients either way."""
    from gripnet_amd import autograd as ag
    gen = torch.Generator().manual_seed(33)
    n, f, c = 200, 24, 5
    z = torch.randn(n, f, generator=gen)
    torch.manual_seed(39)
    mc = gripnet_amd.multiClassInnerProductDecoder(f, c).to(gpu)
    built = []
    real = _hip.GraphPlan.plain_sum.__func__
    def counting(cls, *args, **kw):
        built.append(1)
        return real(cls, *args, **kw)
    _hip.GraphPlan.plain_sum = classmethod(counting)
    try:
        with ag._node_plans_lock:
            ag._node_plans.clear()
            ag._node_plan_misses = 0
        def grads(node_list, nodes_cpu):
            mc.zero_grad()
            zg = leaf(z.to(gpu))
            p = mc(zg, node_list, softmax=True)
            proj = torch.linspace(-1, 1, p.numel()).reshape(p.shape)
            (p * proj.to(gpu)).sum().backward()
            zr, wr = leaf(z), leaf(mc.weight.cpu())
            (orc.multiclass(zr, nodes_cpu, wr, softmax=True) * proj).sum().backward()
            close(zg.grad, zr.grad, what="dz")
            close(mc.weight.grad, wr.grad, what="dW")
        same = torch.randint(0, n, (64,), generator=gen)
        same32 = torch.stack([same, same], dim=1).to(gpu)[:, 0]   # a strided view: copied on every forward, the caller's object repeats
        assert not same32.is_contiguous()
        for _ in range(4):
            grads(same32, same)
        assert len(built) == 1
        for k in range(6):                                     # a new list (with repeated nodes) every step
            fresh = torch.randint(0, n, (64,), generator=gen)
            grads(fresh.to(gpu), fresh)
        assert len(built) <= 3, built                          # two more builds, then none
        grads(same32, same)                                    # (dropped from the four remembered lists meanwhile: a plan again or not - right either way)
    finally:
        _hip.GraphPlan.plain_sum = classmethod(real)
        with ag._node_plans_lock:
            ag._node_plans.clear()
            ag._node_plan_misses = 0


def _nll(pred, labels):
    return -torch.log(pred[torch.arange(labels.shape[0]), labels] + 1e-13).mean()      # GripNet-aminer.py:133


def test_aminer_training_step_gradients(gpu):
    """One full training-step gradient of the aminer-style model (GripNet-aminer.py:124-135): every
    parameter gradient against torch autograd through the oracle."""
    data = make_nc("tiny")
    torch.manual_seed(41)
    model = AminerModel(data.n_p_node, data.n_a_node, data.n_a_type, pp_nhids=(16, 8, 8), pa_out=(8, 8), aa_hidden=(16, 8))
    sd = {k: leaf(v) for k, v in model.state_dict().items()}
    nodes = torch.arange(0, data.n_a_node, 2)
    labels = data.a_label[nodes]
    model = model.to(gpu)
    data_gpu = make_nc("tiny").to(gpu)                       # Data.to moves in place: keep a CPU copy for the oracle
    z, pred = model(data_gpu, nodes.to(gpu))
    loss = _nll(pred, labels.to(gpu))
    loss.backward()
    ref = orc.aminer_forward(sd, data.pp_edge_idx, data.pp_edge_weight, data.pa_edge_idx, data.aa_edge_idx,
                             data.aa_edge_weight, nodes)
    loss_ref = _nll(ref["score"], labels)
    loss_ref.backward()
    close(loss, loss_ref, what="loss")
    for k, p in model.named_parameters():
        if sd[k].grad is None:
            continue
        close(p.grad, sd[k].grad, 1e-4, what=k)
    # and an optimiser step moves the loss down, as the reference's train() expects (GripNet-aminer.py:134-135)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    losses = []
    for _ in range(5):
        opt.zero_grad()
        _, pred = model(data_gpu, nodes.to(gpu))
        l2 = _nll(pred, labels.to(gpu))
        l2.backward()
        opt.step()
        losses.append(float(l2))
    assert losses[-1] < losses[0]


def test_freebase_c_training_step_gradients(gpu):
    data = make_nc("tiny")
    torch.manual_seed(43)
    model = FreebaseCModel(data.n_p_node, data.n_q_node, data.n_a_node, data.n_a_type, pp_nhids=(16, 8, 8),
                           qq_nhids=(16, 8, 8), pa_out=(8, 8), aa_hidden=(8,))
    sd = {k: leaf(v) for k, v in model.state_dict().items()}
    nodes = torch.arange(1, data.n_a_node, 2)
    labels = data.a_label[nodes]
    model = model.to(gpu)
    z, pred = model(make_nc("tiny").to(gpu), nodes.to(gpu))
    _nll(pred, labels.to(gpu)).backward()
    ref = orc.freebase_c_forward(sd, data.pp_edge_idx, data.pp_edge_weight, data.pa_edge_idx, data.qq_edge_idx,
                                 data.qq_edge_weight, data.qa_edge_idx, sd["aa_embeddings"], data.aa_edge_idx,
                                 data.aa_edge_weight, nodes, data.n_a_node)
    _nll(ref["score"], labels).backward()
    for k, p in model.named_parameters():
        if sd[k].grad is None:
            continue
        close(p.grad, sd[k].grad, 1e-4, what=k)


@pytest.mark.parametrize("scale", ["tiny", "aminer-syn"])
def test_freebase_c_training_step_with_bf16_tables(gpu, scale):
    """BASELINE config 5 under training (round 6): with `table_storage = "bf16"` the forward of a training step gathers from the
    bf16-rounded x W (the inference path's launches) and the backward is the fp32 layer's (the rounding's straight-through
    derivative).  No reference exists for reduced precision: the step is held to the SAME model's fp32 step - loss within 2e-2
    relative, every parameter gradient within 1e-1 of its largest entry (one bf16 rounding, 2^-8, of every gathered element
    through four stacked layers and the ReLU masks they flip), and the bf16 forward under autograd gives the bits of the bf16
    inference forward."""
    from gripnet_amd.utils import class_loss, set_table_storage
    data = make_nc(scale)
    torch.manual_seed(43)
    kw = dict(pp_nhids=(16, 8, 8), qq_nhids=(16, 8, 8), pa_out=(8, 8), aa_hidden=(8,)) if scale == "tiny" else {}
    model = FreebaseCModel(data.n_p_node, data.n_q_node, data.n_a_node, data.n_a_type, **kw).to(gpu)
    nodes = torch.arange(1, data.n_a_node, 2).to(gpu)
    dg = make_nc(scale).to(gpu)
    labels = dg.a_label[nodes].contiguous()
    grads, losses = {}, {}
    for storage in ("fp32", "bf16"):
        touched = set_table_storage(model, storage)
        assert len(touched) == 7
        model.zero_grad()
        z, pred = model(dg, nodes)
        loss = class_loss(pred, labels)
        loss.backward()
        losses[storage] = float(loss)
        grads[storage] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        if storage == "bf16":
            with torch.no_grad():
                z_inf, pred_inf = model(dg, nodes)
            assert torch.equal(z_inf, z.detach())          # (the class decoder's training path is other launches: its probabilities agree to rounding)
            assert float((pred_inf - pred.detach()).abs().max()) <= 1e-6
    set_table_storage(model, "fp32")
    assert abs(losses["bf16"] - losses["fp32"]) <= 2e-2 * abs(losses["fp32"]), losses
    assert set(grads["bf16"]) == set(grads["fp32"])
    for k, g32 in grads["fp32"].items():
        scale_k = float(g32.abs().max())
        if scale_k == 0.0:
            continue
        assert float((grads["bf16"][k] - g32).abs().max()) <= 1e-1 * scale_k, k     # (measured: 5.6e-2 on pp.embedding, four layers deep, at full size)
    _hip.raise_if_index_errors(gpu)


@pytest.mark.parametrize("n,fin,fout,bias", [(100, 48, 32, False), (40, 12, 20, True), (645, 48, 32, False)])
def test_rgcn_conv_gradients(gpu, n, fin, fout, bias):
    gen = torch.Generator().manual_seed(47 + n)
    sizes = [300, 0, 1200, 5, 40, 700]
    blocks = [torch.randint(0, max(1, n - 3), (2, s), generator=gen) for s in sizes]     # last nodes: no edges
    ei = torch.cat(blocks, dim=1)
    rl = gripnet_amd.utils.get_range_list(blocks)
    x = torch.randn(n, fin, generator=gen)
    proj = torch.randn(n, fout, generator=gen)
    torch.manual_seed(53)
    rg = gripnet_amd.myRGCN(fin, fout, len(sizes), 5, False, bias=bias).to(gpu)
    if bias:
        rg.bias.data.normal_()
    xg = leaf(x.to(gpu))
    y = rg(xg, ei.to(gpu), None, rl, _relu=True)
    (y * proj.to(gpu)).sum().backward()
    sd = {k: leaf(v.cpu()) for k, v in rg.state_dict().items()}
    xr = leaf(x)
    yr = torch.relu(orc.rgcn_forward(xr, ei, rl, sd["basis"], sd["att"], sd["root"], sd.get("bias")))
    (yr * proj).sum().backward()
    close(y, yr, what="forward")
    close(xg.grad, xr.grad, 1e-4, what="dx")
    for k, p in rg.named_parameters():
        close(p.grad, sd[k].grad, 2e-4 if k == "basis" else 1e-4, what=k)     # basis: sums over R x n terms


@pytest.mark.parametrize("scale", ["tiny", "small"])
def test_pose_training_step_gradients(gpu, scale):
    """The reference's training step (GripNet-pose.py:117-146): encoder, decoder on positive and negative
    edges, log loss, backward; every parameter gradient against torch autograd through the oracle."""
    from gripnet_amd.pipeline import PoseModel
    from gripnet_amd.synth import make_pose
    from gripnet_amd.utils import EPS
    data = make_pose(scale)
    torch.manual_seed(59)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type)
    sd = {k: leaf(v) for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(61)
    neg = torch.randint(0, data.n_d_node, data.train_idx.shape, generator=gen)

    def loss_fn(pos, negs):
        return -torch.log(pos + EPS).mean() - torch.log(1 - negs + EPS).mean()          # GripNet-pose.py:140-142

    model = model.to(gpu)
    dg = make_pose(scale).to(gpu)
    z = model.encode(dg)
    loss = loss_fn(model.dmt(z, dg.train_idx, dg.train_et), model.dmt(z, neg.to(gpu), dg.train_et))
    loss.backward()
    ref = orc.pose_forward(sd, data.gg_edge_index, data.edge_weight, data.gd_edge_index, data.train_idx,
                           data.train_et, data.train_range)
    neg_ref = orc.distmult(ref["z_dd"], neg, data.train_et, sd["dmt.weight"])
    loss_ref = loss_fn(ref["score"], neg_ref)
    loss_ref.backward()
    close(loss, loss_ref, what="loss")
    for k, p in model.named_parameters():
        if sd[k].grad is None:                                 # gd.target_feat_down is unused in cat mode
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        scale_k = max(1.0, float(sd[k].grad.abs().max()))
        close(p.grad / scale_k, sd[k].grad / scale_k, 1e-4, what=k)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)       # GripNet-pose.py:104
    losses = []
    for _ in range(5):
        opt.zero_grad()
        z = model.encode(dg)
        l2 = loss_fn(model.dmt(z, dg.train_idx, dg.train_et), model.dmt(z, neg.to(gpu), dg.train_et))
        l2.backward()
        opt.step()
        losses.append(float(l2))
    assert losses[-1] < losses[0], losses


@pytest.mark.parametrize("m,k1,k2", [(19081, 64, 16), (645, 48, 32), (100, 7, 5), (1, 16, 16), (3000, 64, 64)])
def test_weight_gradient_contraction(gpu, m, k1, k2):
    """x^T g of the tall-skinny weight gradients (gn_xtg_f32) against float64; column slices as operands."""
    from gripnet_amd import _hip
    gen = torch.Generator().manual_seed(41 + m)
    wide = torch.randn(m, k1 + 3, generator=gen).to(gpu)
    x, g = wide[:, 3:], torch.randn(m, k2, generator=gen).to(gpu)
    got = _hip.xtg(x, g)
    want = (x.double().t() @ g.double()).float()
    close(got, want, 1e-4 * max(1.0, want.abs().max().item()), what="x^T g")
    assert torch.equal(got, _hip.xtg(x, g))                  # fixed summation order


def test_training_step_replays_as_one_graph(gpu):
    """Once the plans exist, a whole training step (encoder, both decoder calls, loss, backward, Adam) is a fixed
    sequence of launches on fixed buffers: captured in one hipGraph and replayed, it follows the same loss curve as
    the eager loop from the same initial state."""
    import copy
    from gripnet_amd.pipeline import PoseModel
    from gripnet_amd.synth import make_pose
    from gripnet_amd.utils import EPS
    dg = make_pose("small").to(gpu)
    torch.manual_seed(67)
    base = PoseModel(dg.n_g_node, dg.n_d_node, dg.n_dd_edge_type).to(gpu)
    neg = torch.randint(0, dg.n_d_node, tuple(dg.train_idx.shape), device=gpu)

    def make(model, capturable):
        opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=capturable)

        def step():
            opt.zero_grad()
            z = model.encode(dg)
            loss = -torch.log(model.dmt(z, dg.train_idx, dg.train_et) + EPS).mean() \
                   - torch.log(1 - model.dmt(z, neg, dg.train_et) + EPS).mean()
            loss.backward()
            opt.step()
            return loss.detach()
        return step

    eager = make(copy.deepcopy(base), False)
    eager_losses = [float(eager()) for _ in range(6)]
    graphed_model = copy.deepcopy(base)
    step = make(graphed_model, True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        warm = [float(step()) for _ in range(3)]              # plans, relation-order check, optimizer state
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = step()
    replayed = []
    for _ in range(3):
        graph.replay()
        replayed.append(float(static_loss))
    got = warm + replayed                                     # the capture itself does not run the step
    for a, b in zip(got, eager_losses):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (got, eager_losses)
    assert got[-1] < got[0]


@needs_fast_paths
def test_decoder_backward_plan_matches_planless(gpu):
    """A static edge list gets its pairing (both directions of an edge reduced as one triple with g1 + g2), the offsets
    of its sort and its task lists once (gn_distmult_bwd_plan): the gradients equal the plan-less call's up to that
    rounding and are the same bits on every call; an edge list with an id outside its table, or with unsorted relation
    ids, is refused when the plan is built."""
    from gripnet_amd import _hip
    gen = torch.Generator().manual_seed(71)
    n, f, R, e = 300, 80, 11, 20000
    half = torch.randint(0, n, (2, e // 2), generator=gen)
    th = torch.sort(torch.randint(0, R, (e // 2,), generator=gen)).values
    # the reference's layout (utils.py:168-198): per relation the edges, then the same edges reversed
    parts_i, parts_t = [], []
    for r in range(R):
        blk = half[:, th == r]
        parts_i += [blk, blk.flip(0)]
        parts_t += [torch.full((2 * blk.shape[1],), r, dtype=torch.int64)]
    ei, et = torch.cat(parts_i, 1), torch.cat(parts_t)
    e = ei.shape[1]
    z = (torch.randn(n, f, generator=gen) * 0.5).to(gpu)
    w = (torch.randn(R, f, generator=gen) * 0.5).to(gpu)
    g = torch.randn(e, generator=gen).to(gpu)
    probs = torch.rand(e, generator=gen).to(gpu)
    ei, et = ei.to(gpu), et.to(gpu)
    plan = _hip.DistMultBwdPlan(ei, et, n, R)
    for p in (None, probs):
        dz0, dd0 = torch.empty_like(z), torch.empty_like(w)
        _hip.distmult_backward(z, ei, et, w, g, dz0, dd0, probs=p)
        dz1, dd1 = torch.full_like(z, 7.0), torch.full_like(w, 7.0)
        plan.backward(z, w, g, dz1, dd1, probs=p)
        close(dz1, dz0, 1e-5 * max(1.0, float(dz0.abs().max())), what="dz")
        close(dd1, dd0, 1e-5 * max(1.0, float(dd0.abs().max())), what="dD")
        dz2, dd2 = torch.empty_like(z), torch.empty_like(w)
        plan.backward(z, w, g, dz2, dd2, probs=p)
        assert torch.equal(dz1, dz2) and torch.equal(dd1, dd2)
    bad = ei.clone()
    bad[1, 5] = n
    with pytest.raises(IndexError):
        _hip.DistMultBwdPlan(bad, et, n, R)
    with pytest.raises(_hip.GripNetHipError):
        _hip.DistMultBwdPlan(ei, et[torch.randperm(e, generator=gen).to(gpu)], n, R)


@pytest.mark.parametrize("n,f,R", [(645, 80, 40), (100, 20, 6), (768, 48, 9), (300, 16, 70)])
@pytest.mark.parametrize("use_probs", [False, True])
def test_decoder_backward_large_sorted_lists(gpu, n, f, R, use_probs):
    """The decoder's backward on type-sorted lists of tens of thousands of triples: hub relations, empty relations, a
    relation of one triple, repeated triples, u == v - the plan-less and the planned call against float64, the same bits
    on every call."""
    gen = torch.Generator().manual_seed(n + f + R)
    sizes = [9000, 0, 1, 4096, 4097, 300] + [int(x) for x in torch.randint(0, 1500, (R - 6,), generator=gen)]
    half = [torch.randint(0, n, (2, s), generator=gen) for s in sizes]
    half[0][:, :200] = half[0][:, 200:400]                     # repeated triples
    half[0][1, 400:450] = half[0][0, 400:450]                  # u == v
    # the reference's layout (utils.py:168-198): per relation the edges, then the same edges reversed
    ei = torch.cat([torch.cat([b, b.flip(0)], 1) for b in half], 1).to(gpu)
    et = torch.cat([torch.full((2 * s,), r, dtype=torch.int64) for r, s in enumerate(sizes)]).to(gpu)
    e = ei.shape[1]
    z = (torch.randn(n, f, generator=gen) * 0.5).to(gpu)
    w = (torch.randn(R, f, generator=gen) * 0.5).to(gpu)
    g = torch.randn(e, generator=gen).to(gpu)
    probs = torch.rand(e, generator=gen).to(gpu) if use_probs else None
    gd = (g * probs * (1 - probs) if use_probs else g).double()
    zu, zv, wr = z.double()[ei[0]], z.double()[ei[1]], w.double()[et]
    ref_dz = torch.zeros(n, f, dtype=torch.float64, device=gpu)
    ref_dz.index_add_(0, ei[0], gd[:, None] * zv * wr)
    ref_dz.index_add_(0, ei[1], gd[:, None] * zu * wr)
    ref_dd = torch.zeros(R, f, dtype=torch.float64, device=gpu).index_add_(0, et, gd[:, None] * zu * zv)
    tol_z, tol_d = 2e-5 * max(1.0, float(ref_dz.abs().max())), 2e-5 * max(1.0, float(ref_dd.abs().max()))
    outs = []
    for _ in range(2):
        dz, dd = torch.full_like(z, 7.0), torch.full_like(w, 7.0)
        _hip.distmult_backward(z, ei, et, w, g, dz, dd, probs=probs)
        assert float((dz.double() - ref_dz).abs().max()) <= tol_z and float((dd.double() - ref_dd).abs().max()) <= tol_d
        outs.append((dz, dd))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    plan = _hip.DistMultBwdPlan(ei, et, n, R)
    dz, dd = torch.full_like(z, 7.0), torch.full_like(w, 7.0)
    plan.backward(z, w, g, dz, dd, probs=probs)
    assert float((dz.double() - ref_dz).abs().max()) <= tol_z and float((dd.double() - ref_dd).abs().max()) <= tol_d
    _hip.raise_if_index_errors(gpu)


def _sharded_hip_worker(rank, world, port, q):
    """One rank of a 2-way sharded training step on the HIP kernels; both ranks share cuda:0, the exchanges travel
    over gloo through host copies (the product uses RCCL on the ranks' own GPUs: gripnet_amd/sharded.py)."""
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gripnet_amd.pipeline import PoseModel
        from gripnet_amd.sharded import ShardedPoseTraining
        from gripnet_amd.synth import make_pose
        dev = torch.device("cuda:0")
        data = make_pose("small").to(dev)
        torch.manual_seed(59)
        model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
        neg = torch.randint(0, data.n_d_node, data.train_idx.shape, generator=torch.Generator().manual_seed(61)).to(dev)
        step = ShardedPoseTraining(model, data, rank, world)

        def all_reduce(t):
            host = t.detach().cpu()
            dist.all_reduce(host)
            t.copy_(host)
            return t
        step.all_reduce = all_reduce
        loss = step.step(neg)
        torch.cuda.synchronize()
        q.put((rank, float(loss), {k: (None if p.grad is None else p.grad.detach().cpu().numpy()) for k, p in model.named_parameters()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_training_step_on_hip_kernels(gpu):
    """gripnet_amd.sharded.ShardedPoseTraining on the HIP kernels, two ranks = two processes on this GPU.  Loss and
    every parameter gradient against torch autograd through the oracle, on both ranks; the ranks agree bit for bit."""
    import socket

    import torch.multiprocessing as mp

    from gripnet_amd.pipeline import PoseModel
    from gripnet_amd.synth import make_pose
    from gripnet_amd.utils import EPS
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_hip_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, loss, grads = q.get(timeout=400)
        got[rank] = (loss, grads)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    data = make_pose("small")
    torch.manual_seed(59)
    base = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type)
    sd = {k: leaf(v) for k, v in base.state_dict().items()}
    neg = torch.randint(0, data.n_d_node, data.train_idx.shape, generator=torch.Generator().manual_seed(61))
    ref = orc.pose_forward(sd, data.gg_edge_index, data.edge_weight, data.gd_edge_index, data.train_idx,
                           data.train_et, data.train_range)
    neg_ref = orc.distmult(ref["z_dd"], neg, data.train_et, sd["dmt.weight"])
    loss_ref = -torch.log(ref["score"] + EPS).mean() - torch.log(1 - neg_ref + EPS).mean()
    loss_ref.backward()
    for rank in range(2):
        loss, grads = got[rank]
        assert abs(loss - float(loss_ref)) <= 1e-4, (rank, loss, float(loss_ref))
        for k, g in grads.items():
            if sd[k].grad is None:
                continue
            scale_k = max(1.0, float(sd[k].grad.abs().max()))
            close(torch.from_numpy(g) / scale_k, sd[k].grad / scale_k, 1e-4, what="rank {} {}".format(rank, k))
    for k, g in got[0][1].items():
        if g is not None:
            assert (g == got[1][1][k]).all(), "ranks disagree on " + k


def test_rgcn_shard_gradient_shares_add_up(gpu):
    """The edge sums of the relational layer's gradient over two edge ranges add up to those of the whole graph."""
    from gripnet_amd import _hip
    from gripnet_amd.autograd import rgcn_edge_gradients
    from gripnet_amd.synth import make_pose
    d = make_pose("small").to(gpu)
    n, R = d.n_d_node, d.n_dd_edge_type
    torch.manual_seed(2)
    x, gm = torch.randn(n, 48, device=gpu), torch.randn(n, 32, device=gpu)
    basis, att = torch.randn(32, 48, 32, device=gpu) * 0.1, torch.randn(R, 32, device=gpu) * 0.2
    E = int(d.train_idx.shape[1])
    cut = E // 3                                               # falls inside a relation
    full = rgcn_edge_gradients(_hip.RgcnPlan(d.train_idx, d.train_range, n), x, basis, att, gm)
    a = rgcn_edge_gradients(_hip.RgcnPlan(d.train_idx, d.train_range, n, 0, cut), x, basis, att, gm)
    b = rgcn_edge_gradients(_hip.RgcnPlan(d.train_idx, d.train_range, n, cut, E), x, basis, att, gm)
    for name, f, p, q in zip(("dx", "dbasis", "datt"), full, a, b):
        scale = max(1.0, float(f.abs().max()))
        close((p + q) / scale, f / scale, 1e-4, what=name)


@pytest.mark.parametrize("n,fin,fout,hub", [(645, 48, 32, True), (100, 16, 16, False), (300, 64, 32, True), (37, 32, 16, False),
                                            (200, 48, 16, True), (1000, 16, 32, False)])
def test_relational_weight_gradient_in_one_launch(gpu, n, fin, fout, hub):
    """gn_rel_weight_grad_f32: dW_r = sum over the edges of relation r of x[src]^T gm[dst], against index_add + matmul in
    float64 - with empty relations, a relation of one edge, sources without edges, a hub relation large enough to be cut
    into parts that several workgroups add up (the last to arrive, in part order), strided x; the same bits on every
    launch; and through rgcn_edge_gradients against the unfused path (GN_DISABLE_FAST)."""
    gen = torch.Generator().manual_seed(n + fin + fout)
    sizes = [300, 0, 1, 1200, 5, 40, 700, 0] + ([60000 if n > 250 else 9000] if hub else []) + [17] * 40
    blocks = [torch.randint(0, max(1, n - 3), (2, s), generator=gen) for s in sizes]
    ei = torch.cat(blocks, dim=1).to(gpu)
    rl = gripnet_amd.utils.get_range_list(blocks)
    R = len(sizes)
    wide = torch.randn(n, fin + 8, generator=gen).to(gpu)
    x = wide[:, 4:4 + fin]                                             # rows 4 floats off, leading dimension fin + 8
    gm = torch.randn(n, fout, generator=gen).to(gpu)
    plan = _hip.RgcnPlan(ei, rl, n)
    wg = plan.weight_grad_plan()
    assert wg is not None and wg.supported(fin, fout)
    dw = wg.weight_grad(x, gm)
    rel = torch.repeat_interleave(torch.arange(R), torch.tensor(sizes)).to(gpu)
    q = torch.zeros(R * n, fout, dtype=torch.float64, device=gpu)
    q.index_add_(0, rel * n + ei[0], gm.double().index_select(0, ei[1]))
    ref = torch.matmul(x.double().t(), q.view(R, n, fout)).reshape(R, fin * fout)
    scale = max(1.0, float(ref.abs().max()))
    assert float((dw.double() - ref).abs().max()) / scale <= 2e-6
    for _ in range(3):
        assert torch.equal(wg.weight_grad(x, gm), dw)
    assert not wg.supported(fin + 1, fout) and not wg.supported(fin, 24)
    _hip.raise_if_index_errors(gpu)


@pytest.mark.parametrize("features,table_rows", [(32, 645), (16, 1500), (64, 300)])
def test_short_row_sums_over_a_small_table(gpu, monkeypatch, features, table_rows):
    """The (relation, source) sums of the relational layer's weight gradient at their real shape: ~10^5 rows of a few
    edges over a table that fits the LDS (k_aggregate_lds_table), with empty rows, rows of more than eight edges and the
    last rows of the grid; against index_add in float64, and equal to the wave-per-row kernels' result."""
    gen = torch.Generator().manual_seed(features + table_rows)
    rows, edges = 70001, 190000
    dst = torch.randint(0, rows, (edges,), generator=gen)
    dst[:4000] = torch.randint(0, 40, (4000,), generator=gen)                  # forty long rows
    dst[4000:4100] = rows - 1                                                   # the last row is long too
    src = torch.randint(0, table_rows, (edges,), generator=gen)
    ei = torch.stack([src, dst]).to(gpu)
    table = torch.randn(table_rows, features, generator=gen).to(gpu)
    ref = torch.zeros(rows, features, dtype=torch.float64, device=gpu).index_add_(0, ei[1], table.double().index_select(0, ei[0]))
    outs = []
    for disabled in ("0", "1"):
        monkeypatch.setenv("GN_DISABLE_LDS_TABLE", disabled)
        plan = _hip.GraphPlan.plain_sum(ei, table_rows, rows)
        out = torch.full((rows, features), float("nan"), device=gpu)
        plan.aggregate(table, None, False, out)
        assert (out.double() - ref).abs().max().item() <= 1e-4
        outs.append(out)
    assert (outs[0] - outs[1]).abs().max().item() <= 1e-4


@pytest.mark.parametrize("rows,cols", [(19081, 16), (645, 32), (645, 48), (7, 80), (1, 1), (3000, 200), (50000, 128), (50000, 64), (70000, 256)])
def test_backward_prologue_in_one_launch(gpu, rows, cols):
    """gn_grad_prologue_f32: ReLU mask by the saved output, division by the rows' divisor and the column sums of the masked
    gradient against the torch expressions, on a row-strided gradient (a slot of a concatenated output's gradient); the
    same bits on every launch; every output optional.  (50,000 x 128 and 70,000 x 256: the node-classification models'
    layers - 1,024 workgroups whose sums meet in 32 sets of 32, then once more.)"""
    gen = torch.Generator().manual_seed(rows + cols)
    wide = torch.randn(rows, cols + 24, generator=gen).to(gpu)
    g = wide[:, 8:8 + cols]
    out = torch.relu(torch.randn(rows, cols, generator=gen)).to(gpu)
    div = (1.0 + torch.randint(0, 9, (rows,), generator=gen).float()).to(gpu)
    gm, gd, cs = _hip.grad_prologue(g, out, div, True, True)
    ref = g * (out > 0)
    assert torch.equal(gm, ref)
    assert float((gd - ref / div.view(-1, 1)).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max()))
    want = ref.double().sum(dim=0)
    assert float((cs.double() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    for _ in range(3):
        assert torch.equal(_hip.grad_prologue(g, out, div, True, True)[2], cs)
    gm2, gd2, cs2 = _hip.grad_prologue(g, None, None, True, False)
    assert torch.equal(gm2, g) and gd2 is None and cs2 is None
    gm3, _, cs3 = _hip.grad_prologue(g, None, None, False, True)
    assert gm3 is None and float((cs3.double() - g.double().sum(dim=0)).abs().max()) <= 2e-5 * max(1.0, float(g.abs().sum(dim=0).max()))
    _hip.raise_if_index_errors(gpu)


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_adam_step_matches_torch(gpu, wd):
    """gripnet_amd.optim.Adam (gn_adam_step_f32: all parameters in one launch, step counter on the device) against
    torch.optim.Adam over several steps: tensors of 1 to ~600 K elements, sizes that are not multiples of four, a parameter
    without a gradient, new gradient tensors every step."""
    import gripnet_amd
    gen = torch.Generator().manual_seed(7)
    shapes = [(19081, 32), (32, 16), (16,), (1,), (645, 7), (963, 32), (5, 3, 11), (4097,), (64, 48, 32)]
    ours = [torch.nn.Parameter(torch.randn(s, generator=gen).to(gpu)) for s in shapes]
    theirs = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    idle_a, idle_b = torch.nn.Parameter(torch.ones(3, device=gpu)), torch.nn.Parameter(torch.ones(3, device=gpu))
    a = gripnet_amd.optim.Adam(ours + [idle_a], lr=0.01, weight_decay=wd)
    b = torch.optim.Adam(theirs + [idle_b], lr=0.01, weight_decay=wd)
    for step in range(6):
        a.zero_grad()
        b.zero_grad()
        for p, q in zip(ours, theirs):
            g = torch.randn(p.shape, generator=gen).to(gpu) * (10.0 ** (step - 3))
            p.grad, q.grad = g.clone(), g.clone()
        a.step()
        b.step()
        for k, (p, q) in enumerate(zip(ours, theirs)):
            err = float((p - q).abs().max())
            assert err <= 2e-6 * max(1.0, float(q.abs().max())), (step, k, err)
    assert float(a.param_groups[0]["step"]) == 6.0
    assert torch.equal(idle_a, idle_b)
    _hip.raise_if_index_errors(gpu)


def test_link_loss_matches_the_torch_expression(gpu):
    """utils.link_loss (gn_link_loss_forward_f32 / _backward_f32: one launch each) against the expression the reference's
    driver spells out with torch ops (GripNet-pose.py:140-142): value, both gradients, an upstream factor, run-to-run bits,
    empty lists."""
    from gripnet_amd.utils import EPS, link_loss
    gen = torch.Generator().manual_seed(3)
    for n_pos, n_neg in ((200003, 150001), (7, 1), (1, 70000)):
        p = (torch.rand(n_pos, generator=gen) * 0.98 + 0.01).to(gpu).requires_grad_(True)
        q = (torch.rand(n_neg, generator=gen) * 0.98 + 0.01).to(gpu).requires_grad_(True)
        p.data[0] = 0.0                                                # log(0 + EPS): the reason EPS exists
        ref = -torch.log(p.double() + EPS).mean() - torch.log(1 - q.double() + EPS).mean()
        (3.0 * ref).backward()
        gp, gq = p.grad.clone(), q.grad.clone()
        p.grad = q.grad = None
        loss = link_loss(p, q)
        (3.0 * loss).backward()
        assert abs(float(loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
        assert ((p.grad - gp).abs() <= 2e-6 * gp.abs() + 1e-12).all() and ((q.grad - gq).abs() <= 2e-6 * gq.abs() + 1e-12).all()
        assert torch.equal(link_loss(p, q), loss)
    empty = torch.empty(0, device=gpu)
    some = torch.full((5,), 0.5, device=gpu)
    assert abs(float(link_loss(some, empty)) + float(torch.log(some + EPS).mean())) < 1e-6


def test_wide_layer_backward(gpu):
    """A layer wider than the 256 columns one launch of gn_grad_prologue_f32 covers (the reference accepts any width,
    layers.py:16-24): the prologue runs per block of 256 columns, the gradients match torch autograd through the oracle."""
    gen = torch.Generator().manual_seed(83)
    n, fin, fout, e = 120, 12, 520, 900
    ei = torch.randint(0, n, (2, e), generator=gen)
    x = torch.randn(n, fin, generator=gen)
    proj = torch.randn(n, fout, generator=gen)
    torch.manual_seed(85)
    conv = gripnet_amd.myGCN(fin, fout).to(gpu)
    conv.bias.data.normal_()
    xg = leaf(x.to(gpu))
    y = conv(xg, ei.to(gpu), None, _relu=True)
    (y * proj.to(gpu)).sum().backward()
    xr, wr, br = leaf(x), leaf(conv.weight.cpu()), leaf(conv.bias.cpu())
    yr = torch.relu(orc.gcn_forward(xr, wr, br, ei, None))
    (yr * proj).sum().backward()
    close(y, yr, what="forward")
    close(xg.grad, xr.grad, 1e-4, what="dx")
    close(conv.weight.grad, wr.grad, 1e-4, what="dW")
    close(conv.bias.grad, br.grad, 1e-4, what="db")
    wide = torch.randn(300, 700, generator=gen).to(gpu)
    out = torch.relu(torch.randn(300, 700, generator=gen)).to(gpu)
    div = (1.0 + torch.randint(0, 9, (300,), generator=gen).float()).to(gpu)
    gm, gd, cs = _hip.grad_prologue(wide, out, div, True, True)
    ref = wide * (out > 0)
    assert torch.equal(gm, ref)
    assert float((gd - ref / div.view(-1, 1)).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max()))
    assert float((cs.double() - ref.double().sum(0)).abs().max()) <= 2e-5 * max(1.0, float(ref.double().sum(0).abs().max()))
    _hip.raise_if_index_errors(gpu)


def test_adam_resumes_from_a_cpu_checkpoint(gpu, tmp_path):
    """save -> torch.load(map_location="cpu") -> load_state_dict -> step: the step counter a loader leaves on the CPU is moved
    to the parameters' device, the address table is rebuilt for the loaded moments (an optimizer that had stepped before
    the load must not keep updating its old ones); the resumed run equals torch.optim.Adam resumed the same way."""
    gen = torch.Generator().manual_seed(89)
    shapes = [(300, 7), (16,), (33, 5)]
    ours = [torch.nn.Parameter(torch.randn(s, generator=gen).to(gpu)) for s in shapes]
    theirs = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    a, b = gripnet_amd.optim.Adam(ours, lr=0.01), torch.optim.Adam(theirs, lr=0.01)

    def one_step(oa, ob, pa, pb):
        for p, q in zip(pa, pb):
            g = torch.randn(p.shape, generator=gen).to(gpu)
            p.grad, q.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for _ in range(3):
        one_step(a, b, ours, theirs)
    torch.save(a.state_dict(), tmp_path / "a.pt")
    torch.save(b.state_dict(), tmp_path / "b.pt")
    # new optimizers over the same parameters; the first has already stepped once (its address table exists)
    a2, b2 = gripnet_amd.optim.Adam(ours, lr=0.01), torch.optim.Adam(theirs, lr=0.01)
    keep = [p.detach().clone() for p in ours]
    for p in ours:
        p.grad = torch.zeros_like(p)
    a2.step()
    for p, k in zip(ours, keep):
        p.data.copy_(k)
    a2.load_state_dict(torch.load(tmp_path / "a.pt", map_location="cpu"))
    b2.load_state_dict(torch.load(tmp_path / "b.pt", map_location="cpu"))
    for _ in range(3):
        one_step(a2, b2, ours, theirs)
    assert float(a2.param_groups[0]["step"]) == 6.0 and a2.param_groups[0]["step"].device.type == "cuda"
    for p, q in zip(ours, theirs):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max()))
    _hip.raise_if_index_errors(gpu)


def test_frozen_layers_inside_a_training_model(gpu):
    """A frozen first layer (its parameters do not require grad) under a trainable second one, and a frozen external conv
    under a trainable target_feat: the stacks hand their concat Slots to a layer that takes the inference launches."""
    gen = torch.Generator().manual_seed(97)
    n, e = 150, 1200
    a = torch.randint(0, n, (2, e), generator=gen)
    ei = torch.cat([a, a.flip(0)], dim=1)
    x = torch.randn(n, 10, generator=gen)
    torch.manual_seed(99)
    m = gripnet_amd.homoGraph([10, 8, 6]).to(gpu)
    for p in m.conv_list[0].parameters():
        p.requires_grad_(False)
    y = m(x.to(gpu), ei.to(gpu), None, if_catout=True)
    proj = torch.randn(y.shape, generator=gen)
    (y * proj.to(gpu)).sum().backward()
    sd = {"h." + k: leaf(v.cpu()) for k, v in m.state_dict().items()}
    yr = orc.homo_forward(sd, "h.", x, ei, None, if_catout=True)
    (yr * proj).sum().backward()
    close(y, yr, what="forward")
    assert m.conv_list[0].weight.grad is None
    close(m.conv_list[1].weight.grad, sd["h.conv_list.1.weight"].grad, 1e-4, what="dW of the trainable layer")
    n_src, n_tgt = 200, 60
    gd = torch.stack([torch.randint(0, n_src, (900,), generator=gen), torch.randint(0, n_tgt, (900,), generator=gen)])
    xs = torch.randn(n_src, 18, generator=gen)
    ig = gripnet_amd.interGraph(18, 12, n_tgt, target_feat_dim=20).to(gpu)
    for p in ig.conv.parameters():
        p.requires_grad_(False)
    y = ig(xs.to(gpu), gd.to(gpu))
    proj = torch.randn(y.shape, generator=gen)
    (y * proj.to(gpu)).sum().backward()
    sd = {"g." + k: leaf(v.cpu()) for k, v in ig.state_dict().items()}
    yr = orc.inter_forward(sd, "g.", xs, gd, None, if_relu=True, mod="cat", n_target=n_tgt)
    (yr * proj).sum().backward()
    close(y, yr, what="external forward")
    close(ig.target_feat.grad, sd["g.target_feat"].grad, 1e-4, what="d target_feat")
    assert ig.conv.weight.grad is None


def test_last_arriver_hand_overs_replayed(gpu):
    """The kernels whose last workgroup to arrive adds partial sums that other workgroups left in a REUSED scratch
    (gn_xtg_f32, gn_grad_prologue_f32, gn_rel_weight_grad_f32): thousands of back-to-back launches with different inputs,
    every result against float64 - a hand-over that lets the ticket overtake a store reads the previous launch's partials."""
    gen = torch.Generator().manual_seed(101)
    m, k1, k2 = 19081, 64, 16
    xs = [torch.randn(m, k1, generator=gen).to(gpu) for _ in range(4)]
    gs = [torch.randn(m, k2, generator=gen).to(gpu) for _ in range(4)]
    want = [[(x.double().t() @ g.double()) for g in gs] for x in xs]
    outs = []
    for it in range(1500):
        outs.append((it % 4, (it // 4) % 4, _hip.xtg(xs[it % 4], gs[(it // 4) % 4])))
    for i, j, got in outs:
        assert float((got.double() - want[i][j]).abs().max()) <= 1e-4 * float(want[i][j].abs().max())
    # a long product: more than 32 slices, handed over in two levels (sets of sixteen slices, then the sets)
    xl = [torch.randn(45000, 64, generator=gen).to(gpu) for _ in range(2)]
    gl = [torch.randn(45000, 32, generator=gen).to(gpu) for _ in range(2)]
    wantl = [[(x.double().t() @ g.double()) for g in gl] for x in xl]
    outs = [(it % 2, (it // 2) % 2, _hip.xtg(xl[it % 2], gl[(it // 2) % 2])) for it in range(600)]
    for i, j, got in outs:
        assert float((got.double() - wantl[i][j]).abs().max()) <= 1e-4 * float(wantl[i][j].abs().max())
    del xl, gl, outs
    outs = []
    sav = torch.relu(torch.randn(m, k2, generator=gen)).to(gpu)
    wantc = [(g * (sav > 0)).double().sum(0) for g in gs]
    for it in range(1500):
        outs.append((it % 4, _hip.grad_prologue(gs[it % 4], sav, None, False, True)[2]))
    for j, got in outs:
        assert float((got.double() - wantc[j]).abs().max()) <= 2e-5 * float(wantc[j].abs().max())
    n, fin, fout = 645, 48, 32
    sizes = [60000, 300, 0, 1, 1200] + [17] * 30                       # a hub relation cut into parts
    blocks = [torch.randint(0, n, (2, s), generator=gen) for s in sizes]
    ei = torch.cat(blocks, dim=1).to(gpu)
    rl = gripnet_amd.utils.get_range_list(blocks)
    R = len(sizes)
    wg = _hip.RgcnPlan(ei, rl, n).weight_grad_plan()
    x = torch.randn(n, fin, generator=gen).to(gpu)
    gms = [torch.randn(n, fout, generator=gen).to(gpu) for _ in range(3)]
    rel = torch.repeat_interleave(torch.arange(R), torch.tensor(sizes)).to(gpu)
    refs = []
    for gm in gms:
        q = torch.zeros(R * n, fout, dtype=torch.float64, device=gpu)
        q.index_add_(0, rel * n + ei[0], gm.double().index_select(0, ei[1]))
        refs.append(torch.matmul(x.double().t(), q.view(R, n, fout)).reshape(R, fin * fout))
    outs = [(it % 3, wg.weight_grad(x, gms[it % 3])) for it in range(600)]
    for j, got in outs:
        assert float((got.double() - refs[j]).abs().max()) <= 2e-6 * max(1.0, float(refs[j].abs().max()))
    _hip.raise_if_index_errors(gpu)


@pytest.mark.parametrize("workload", ["pose0-syn", "pose2-syn"])
def test_decoder_gradients_at_full_size(gpu, workload):
    """The decoder's backward on the whole type-sorted list of a BASELINE configuration (2.0 M / 8.4 M triples: pose2-syn
    takes the unstaged scatter) against index_add_ in float64 (decoder.py:19-23 under autograd)."""
    from gripnet_amd.synth import make_pose
    data = make_pose(workload).to(gpu)
    n, R, f = data.n_d_node, data.n_dd_edge_type, 80
    torch.manual_seed(3)
    z, D = torch.randn(n, f, device=gpu) * 0.3, torch.randn(R, f, device=gpu) * 0.3
    ei, et = data.train_idx, data.train_et
    g = torch.randn(ei.shape[1], device=gpu)
    dz, dd = torch.empty_like(z), torch.empty_like(D)
    _hip.distmult_backward(z, ei, et, D, g, dz, dd)
    u, v = ei[0], ei[1]
    rz = torch.zeros(n, f, device=gpu, dtype=torch.float64)
    rd = torch.zeros(R, f, device=gpu, dtype=torch.float64)
    zz, DD, gg = z.double(), D.double(), g.double()
    step = 1 << 20
    for a in range(0, ei.shape[1], step):
        s = slice(a, a + step)
        rz.index_add_(0, u[s], gg[s, None] * zz[v[s]] * DD[et[s]])
        rz.index_add_(0, v[s], gg[s, None] * zz[u[s]] * DD[et[s]])
        rd.index_add_(0, et[s], gg[s, None] * zz[u[s]] * zz[v[s]])
    assert float((dz.double() - rz).abs().max() / rz.abs().max()) <= 1e-5
    assert float((dd.double() - rd).abs().max() / rd.abs().max()) <= 1e-5
    dz2, dd2 = torch.empty_like(z), torch.empty_like(D)
    _hip.distmult_backward(z, ei, et, D, g, dz2, dd2)
    assert torch.equal(dz, dz2) and torch.equal(dd, dd2)
    _hip.raise_if_index_errors(gpu)


@pytest.mark.timeout(900)
def test_pose0_syn_training_step_gradients(gpu):
    """The training step of GripNet-pose.py:117-146 at the size the driver's training entry is timed on (pose0-syn): ONE
    oracle forward + backward under torch autograd on the host against the HIP step - loss <= 1e-6 relative, every parameter
    gradient <= 1e-4 of its largest entry."""
    from gripnet_amd.pipeline import PoseModel
    from gripnet_amd.synth import make_pose
    from gripnet_amd.utils import link_loss
    data = make_pose("pose0-syn")
    torch.manual_seed(59)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type)
    sd = {k: leaf(v) for k, v in model.state_dict().items()}
    neg = torch.randint(0, data.n_d_node, data.train_idx.shape, generator=torch.Generator().manual_seed(61))
    model = model.to(gpu)
    dg = make_pose("pose0-syn").to(gpu)
    for _ in range(2):                                          # the second pass runs on the static list's plans
        model.zero_grad()
        z = model.encode(dg)
        loss = link_loss(model.dmt(z, dg.train_idx, dg.train_et), model.dmt(z, neg.to(gpu), dg.train_et))
        loss.backward()
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = orc.pose_forward(sd, data.gg_edge_index, data.edge_weight, data.gd_edge_index, data.train_idx,
                           data.train_et, data.train_range)
    neg_ref = orc.distmult(ref["z_dd"], neg, data.train_et, sd["dmt.weight"])
    eps = gripnet_amd.utils.EPS
    loss_ref = -torch.log(ref["score"] + eps).mean() - torch.log(1 - neg_ref + eps).mean()      # GripNet-pose.py:140-142
    loss_ref.backward()
    # the expression summed in float64 over the oracle's fp32 scores (the fp32 `.mean()` of two million terms carries ~1e-6 itself)
    loss_64 = -torch.log(ref["score"].detach().double() + eps).mean() - torch.log(1 - neg_ref.detach().double() + eps).mean()
    assert abs(float(loss) - float(loss_64)) <= 1e-6 * abs(float(loss_64)), (float(loss), float(loss_64))
    assert abs(float(loss) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref)), (float(loss), float(loss_ref))
    for k, p in model.named_parameters():
        if sd[k].grad is None:                                 # gd.target_feat_down is unused in cat mode
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        scale_k = float(sd[k].grad.abs().max())
        assert scale_k > 0, k
        close(p.grad / scale_k, sd[k].grad / scale_k, 1e-4, what=k)
    _hip.raise_if_index_errors(gpu)


@pytest.mark.parametrize("n,fin,fout,bases,slabs", [(300, 160, 16, 4, False), (300, 160, 16, 4, True), (250, 48, 24, 80, False)])
def test_relational_gradients_outside_every_kernel(gpu, n, fin, fout, bases, slabs, monkeypatch):
    """The shapes NO relational gradient kernel covers (more than 128 input features: the weight gradient of neither
    rel_grad.hip nor rgcn_basis.hip; more than 64 bases: `att^T dW`) take torch expressions inside autograd.rgcn_edge_gradients
    (a dense (relation, source) sum + matmul, its slabbed form when that sum would be huge, `att.t() @ dw`).  Unreachable from
    any caller of the reference (its widest relational layer is 64 -> 32 on 32 bases) - so this test forces them, against torch
    autograd through the oracle."""
    from gripnet_amd import autograd as ag
    if slabs:
        monkeypatch.setattr(ag, "Q_BUDGET_FLOATS", 2 * n * fout)      # (two relations per slab)
    gen = torch.Generator().manual_seed(n + fin + bases)
    torch.manual_seed(fin + bases)
    R = 6
    sizes = [900, 0, 1, 300, 40, 120]
    blocks = [torch.randint(0, n - n // 11, (2, s), generator=gen) for s in sizes]
    ei = torch.cat(blocks, dim=1)
    rl = gripnet_amd.utils.get_range_list(blocks)
    x = torch.randn(n, fin, generator=gen)
    proj = torch.randn(n, fout, generator=gen)
    rg = gripnet_amd.myRGCN(fin, fout, R, bases, False, bias=True).to(gpu)
    rg.bias.data.normal_()
    xg, eig = leaf(x.to(gpu)), ei.to(gpu)
    y = rg(xg, eig, None, rl, _relu=True)
    plan = rg._plan
    wg = plan.weight_grad_plan()
    assert (wg is None or not wg.supported(fin, fout)) and (fin <= 128 or plan.general_weight_grad(x.to(gpu), proj.to(gpu)) is None)
    (y * proj.to(gpu)).sum().backward()
    sd = {k: leaf(v.cpu()) for k, v in rg.state_dict().items()}
    xr = leaf(x)
    yr = torch.relu(orc.rgcn_forward(xr, ei, rl, sd["basis"], sd["att"], sd["root"], sd["bias"]))
    (yr * proj).sum().backward()
    close(y, yr, 2e-5 * max(1.0, float(yr.abs().max())), what="forward")
    close(xg.grad / max(1.0, float(xr.grad.abs().max())), xr.grad / max(1.0, float(xr.grad.abs().max())), 1e-4, what="dx")
    for k, p in rg.named_parameters():
        scale_k = max(1.0, float(sd[k].grad.abs().max()))
        close(p.grad / scale_k, sd[k].grad / scale_k, 1e-4, what=k)
    _hip.raise_if_index_errors(gpu)


@pytest.mark.parametrize("workload", ["small", "pose0-syn"])
def test_link_prediction_loss_is_the_three_calls_bit_for_bit(gpu, workload):
    """utils.link_prediction_loss (round 6): the two decoder calls and the loss as one autograd node whose backward computes
    the loss's derivative inside the decoder's backward launches (gn_distmult_backward_loss_planned_f32 on the static
    positives, gn_distmult_backward_loss_packed_f32 on the sampler's negatives).  Loss, scores and the gradients of z and D are
    those of `link_loss(dmt(z, pos), dmt(z, neg))` BIT FOR BIT, with an upstream gradient other than one as well; on int64
    negatives without packed pairs (the two-step path for that list) too."""
    from gripnet_amd.pipeline import PoseModel
    from gripnet_amd.synth import make_pose
    from gripnet_amd.utils import link_loss, link_prediction_loss
    data = make_pose(workload).to(gpu)
    torch.manual_seed(71)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(gpu)
    sampler = _hip.NegativeSampler(data.train_idx, data.n_d_node, data.train_range)
    neg_packed = sampler.sample(seed=5)                              # carries the sampler's 32-bit pairs
    neg_plain = neg_packed.clone()                                   # the same pairs as a plain int64 tensor
    with torch.no_grad():
        z0 = model.encode(data)
    dmt = model.dmt
    # the positives are the static list (its plans are built now); the negatives of a training loop are new every epoch and never
    # get a plan - this test scores the SAME negatives several times, so the second-sighting rule is switched off
    dmt.auto_static = False
    dmt.register_static(data.train_idx, data.train_et, num_nodes=int(data.n_d_node))
    for neg in (neg_packed, neg_plain):
        for scale in (1.0, 0.37):
            outs = []
            for fused in (False, True):
                for _ in range(2):                                   # the second pass runs on the static list's plans
                    z = z0.clone().requires_grad_(True)
                    dmt.weight.grad = None
                    if fused:
                        loss, pos, negs = link_prediction_loss(dmt, z, data.train_idx, neg, data.train_et)
                    else:
                        pos, negs = dmt(z, data.train_idx, data.train_et), dmt(z, neg, data.train_et)
                        loss = link_loss(pos, negs)
                    (scale * loss).backward()
                outs.append((loss.detach().clone(), pos.detach().clone(), negs.detach().clone(), z.grad.clone(), dmt.weight.grad.clone()))
            for a, b, name in zip(outs[0], outs[1], ("loss", "pos", "neg", "dz", "dD")):
                assert torch.equal(a, b), (name, float((a - b).abs().max()))
    assert not pos.requires_grad and not negs.requires_grad          # (the scores of the fused call feed the metrics only)
    _hip.raise_if_index_errors(gpu)


def test_class_loss_and_softmax_gradients(gpu):
    """utils.class_loss (gn_class_loss_*) against the expression the NC drivers spell out (GripNet-aminer.py:133), and the
    row softmax's gradient (gn_softmax_rows_backward_f32) against torch's, through the class decoder under autograd."""
    from gripnet_amd.utils import EPS, class_loss
    gen = torch.Generator().manual_seed(211)
    n, C = 3001, 8
    score = torch.softmax(torch.randn(n, C, generator=gen), dim=1).to(gpu).requires_grad_(True)
    cls = torch.randint(0, C, (n,), generator=gen).to(gpu)
    ref = -torch.log(score.double()[torch.arange(n, device=gpu), cls] + EPS).mean()
    (2.0 * ref).backward()
    want = score.grad.clone()
    score.grad = None
    loss = class_loss(score, cls)
    (2.0 * loss).backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref))
    assert float((score.grad - want).abs().max()) <= 2e-6 * float(want.abs().max())
    z = torch.randn(500, 40, generator=gen)
    nodes = torch.randint(0, 500, (300,), generator=gen)
    nodes[:20] = nodes[20:40]                                          # nodes listed twice: their gradients add up
    proj = torch.randn(300, C, generator=gen)
    torch.manual_seed(5)
    dec = gripnet_amd.multiClassInnerProductDecoder(40, C).to(gpu)
    zg = leaf(z.to(gpu))
    p = dec(zg, nodes.to(gpu))
    (p * proj.to(gpu)).sum().backward()
    zr, wr = leaf(z), leaf(dec.weight.detach().cpu())
    pr = torch.softmax(zr[nodes] @ wr, dim=1)
    (pr * proj).sum().backward()
    close(p, pr, what="softmax")
    close(zg.grad, zr.grad, what="dz")
    close(dec.weight.grad, wr.grad, 1e-4, what="dW")
    _hip.raise_if_index_errors(gpu)


@pytest.mark.parametrize("m,k1,k2", [(5000, 128, 64), (3000, 256, 128), (700, 65, 33), (900, 200, 7), (50000, 128, 64), (70001, 64, 32), (21000, 48, 32),
                                     (50000, 256, 128), (20003, 512, 128), (9000, 64, 64), (9000, 192, 64), (4097, 128, 32), (30000, 256, 32), (8000, 64, 128)])
def test_wide_weight_gradient_in_tiles(gpu, m, k1, k2):
    """x^T g wider than one launch of gn_xtg_f32 covers (the 128 x 64 ... 256 x 128 layers of the NC models): as WIDE products
    (a workgroup per row slice computes all 2-16 tiles of 64 x 32 outputs, a fold adds the slices; every tile shape the kernel
    has, 512 columns as two blocks) where rows and widths allow, else tiles of 64 x 32 outputs over column slices; alone and
    inside a dense batch, against float64; the same bits on every call."""
    gen = torch.Generator().manual_seed(m + k1)
    x, g = torch.randn(m, k1, generator=gen).to(gpu), torch.randn(m, k2, generator=gen).to(gpu)
    want = x.double().t() @ g.double()
    got = _hip.xtg(x, g)
    assert float((got.double() - want).abs().max()) <= 1e-4 * float(want.abs().max())
    assert torch.equal(_hip.xtg(x, g), got)
    with _hip.dense_batch(gpu):
        batched = _hip.xtg(x, g, join_batch=True)
    assert torch.equal(batched, got)
    _hip.raise_if_index_errors(gpu)


@pytest.mark.parametrize("m,n,k", [(19081, 32, 16), (645, 48, 32), (50, 7, 1000), (3000, 96, 40), (9, 5, 3), (5000, 64, 64), (50000, 256, 128)])
def test_product_with_an_addend(gpu, m, n, k):
    """c = a b^T + addend (gn_gemm_addend_f32): the dx of a layer whose input also sits in a concat - the addend is a column
    slice of the concat's gradient; every kernel the product can take (deep and narrow, tall-skinny fp32 and split bf16, generic), alone,
    on top of GN_GEMM_ACCUMULATE, and queued in a dense batch, against float64; the addend is left as it was."""
    gen = torch.Generator().manual_seed(m + n + k)
    a, b = torch.randn(m, k, generator=gen).to(gpu), torch.randn(n, k, generator=gen).to(gpu)
    wide = torch.randn(m, n + 24, generator=gen).to(gpu)
    addend = wide[:, 8:8 + n]
    kept = wide.clone()
    want = a.double() @ b.double().t() + addend.double()
    tol = 2e-5 * float(want.abs().max())
    got = _hip.gemm(a, b, torch.empty(m, n, device=gpu), b_transposed=True, addend=addend)
    assert float((got.double() - want).abs().max()) <= tol
    base = torch.randn(m, n, generator=gen).to(gpu)
    acc = _hip.gemm(a, b, base.clone(), b_transposed=True, accumulate=True, addend=addend)
    assert float((acc.double() - want - base.double()).abs().max()) <= tol
    with _hip.dense_batch(gpu):
        queued = _hip.gemm(a, b, torch.empty(m, n, device=gpu), b_transposed=True, join_batch=True, addend=addend)
        other = _hip.xtg(a, base, join_batch=True)
    assert torch.equal(queued, got)
    assert float((other.double() - a.double().t() @ base.double()).abs().max()) <= 1e-4 * float(other.abs().max())
    assert torch.equal(wide, kept)
    assert _hip.addend_ok(addend, m, n) and not _hip.addend_ok(addend.t(), n, m) and not _hip.addend_ok(addend.double(), m, n)
    with pytest.raises(ValueError):
        _hip.gemm(a, b, torch.empty(m, n, device=gpu), b_transposed=True, addend=wide)
    _hip.raise_if_index_errors(gpu)


def test_an_input_that_also_sits_in_the_concat_gets_one_gradient(gpu):
    """homoGraph with if_catout (layers.py:280-281,307-309): a layer's input is used by the layer AND by the concat.  The
    training path hands the input through the layer's Function, whose backward adds the concat's gradient columns where it
    stores dx (no element-wise launch of the autograd engine in between: torch's profiler sees no aten::add in the backward);
    gradients against torch autograd through the oracle for a GCN stack whose first input is the embedding, at a size where
    the products take the tall-skinny kernel, with every layer trainable and with a frozen first layer; an input that is a
    plain tensor needing a gradient gets the sum as well."""
    gen = torch.Generator().manual_seed(77)
    n, e = 3000, 40000
    ei = torch.randint(0, n, (2, e), generator=gen)
    torch.manual_seed(5)
    hg = gripnet_amd.homoGraph([24, 16, 16], start_graph=True, in_dim=n).to(gpu)
    proj = torch.randn(n, 24 + 16 + 16, generator=gen)
    for frozen in (False, True):
        hg.zero_grad()
        for p in hg.conv_list[0].parameters():
            p.requires_grad_(not frozen)
        out = hg(None, ei.to(gpu), if_catout=True)
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
            (out * proj.to(gpu)).sum().backward()
        adds = [ev.name for ev in prof.events() if ev.name in ("aten::add", "aten::add_")]
        assert not adds, adds
        sd = {"h." + k: leaf(v.cpu()) for k, v in hg.state_dict().items()}
        ref = orc.homo_forward(sd, "h.", None, ei, None, if_catout=True)
        (ref * proj).sum().backward()
        close(out, ref, what="forward")
        for k, p in hg.named_parameters():
            if not p.requires_grad:
                assert p.grad is None
                continue
            close(p.grad, sd["h." + k].grad, 1e-4, what=k)
    x = torch.randn(n, 24, generator=gen)
    xg = x.to(gpu).requires_grad_(True)
    plain = gripnet_amd.homoGraph([24, 16]).to(gpu)
    out = plain(xg, ei.to(gpu), if_catout=True)
    (out * proj[:, :40].to(gpu)).sum().backward()
    sd = {"h." + k: leaf(v.cpu()) for k, v in plain.state_dict().items()}
    xr = leaf(x)
    ref = orc.homo_forward(sd, "h.", xr, ei, None, if_catout=True)
    (ref * proj[:, :40]).sum().backward()
    close(xg.grad, xr.grad, 1e-4, what="dx of an input that sits in the concat")
    _hip.raise_if_index_errors(gpu)


@pytest.mark.parametrize("n,fin,fout,bases,R", [(8000, 64, 32, 16, 520), (70000, 16, 32, 4, 7), (3000, 40, 24, 5, 12), (2500, 128, 16, 8, 5)])
def test_relational_layer_gradients_beyond_the_lds_kernels(gpu, n, fin, fout, bases, R):
    """myRGCN under autograd on graphs the LDS-resident kernels do not cover (the all-nodes baseline, rgcn_pose.py:53-106):
    forward on the O(E) basis-space path, dx through the same path on the reversed edges, the weight gradient per relation
    on gn_rgcn_weight_grad_f32 (relations of one item, hub relations cut into parts, empty relations; widths that are no
    multiple of 16) - every gradient against torch autograd through the oracle; the kernel alone against float64, over two
    shards' edge ranges, and the same bits on every call."""
    gen = torch.Generator().manual_seed(n + R)
    torch.manual_seed(n + fin)
    sizes = [20000, 0, 1, 700] + [int(s) for s in torch.randint(0, 300, (R - 4,), generator=gen)]
    blocks = [torch.randint(0, n - n // 11, (2, s), generator=gen) for s in sizes]
    ei = torch.cat(blocks, dim=1)
    rl = gripnet_amd.utils.get_range_list(blocks)
    x = torch.randn(n, fin, generator=gen)
    proj = torch.randn(n, fout, generator=gen)
    rg = gripnet_amd.myRGCN(fin, fout, R, bases, False, bias=True).to(gpu)
    rg.bias.data.normal_()
    xg, eig = leaf(x.to(gpu)), ei.to(gpu)
    y = rg(xg, eig, None, rl, _relu=True)
    wg = rg._plan.weight_grad_plan()
    assert rg._plan.path(fin, fout, bases) == "general" and (wg is None or not wg.supported(fin, fout))
    (y * proj.to(gpu)).sum().backward()
    sd = {k: leaf(v.cpu()) for k, v in rg.state_dict().items()}
    xr = leaf(x)
    yr = torch.relu(orc.rgcn_forward(xr, ei, rl, sd["basis"], sd["att"], sd["root"], sd["bias"]))
    (yr * proj).sum().backward()
    close(y, yr, 2e-5 * max(1.0, float(yr.abs().max())), what="forward")
    close(xg.grad / max(1.0, float(xr.grad.abs().max())), xr.grad / max(1.0, float(xr.grad.abs().max())), 1e-4, what="dx")
    for k, p in rg.named_parameters():
        scale_k = max(1.0, float(sd[k].grad.abs().max()))
        close(p.grad / scale_k, sd[k].grad / scale_k, 1e-4, what=k)
    # the kernel on its own
    gm = torch.randn(n, fout, generator=gen).to(gpu)
    xd = x.to(gpu)
    plan = rg._plan
    dw = plan.general_weight_grad(xd, gm)
    rel = torch.repeat_interleave(torch.arange(R), torch.tensor(sizes)).to(gpu)
    ref = torch.zeros(R, fin * fout, dtype=torch.float64, device=gpu)
    step = 1 << 16
    for a0 in range(0, ei.shape[1], step):
        s = slice(a0, a0 + step)
        outer = (xd.double()[eig[0, s]].unsqueeze(2) * gm.double()[eig[1, s]].unsqueeze(1)).reshape(-1, fin * fout)
        ref.index_add_(0, rel[s], outer)
    scale = max(1.0, float(ref.abs().max()))
    assert float((dw.double() - ref).abs().max()) / scale <= 2e-6
    assert torch.equal(plan.general_weight_grad(xd, gm), dw)
    E = ei.shape[1]
    parts = [_hip.RgcnPlan(eig, rl, n, lo, hi).general_weight_grad(xd, gm) for lo, hi in ((0, E // 3), (E // 3, E))]
    assert float(((parts[0] + parts[1]).double() - ref).abs().max()) / scale <= 2e-6
    _hip.raise_if_index_errors(gpu)
