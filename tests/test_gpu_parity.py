"""Parity of the HIP path (through the C ABI) with the golden vectors and the CPU oracle.

Bar (BASELINE.json north_star): index outputs bit-exact; fp32 outputs within 1e-4 abs of the
reference.  Most checks are held to a tighter 2e-5 so that regressions show early.
"""
import numpy as np
import os

import pytest
import torch

import gripnet_amd
from gripnet_amd import _hip
from gripnet_amd.pipeline import AminerModel, FreebaseAModel, FreebaseBModel, FreebaseCModel, PoseModel
from gripnet_amd.synth import Data, make_pose
from oracle import gripnet_oracle as orc

pytestmark = pytest.mark.gpu
needs_fast_paths = pytest.mark.skipif(os.environ.get("GN_DISABLE_FAST") == "1",
                                      reason="exercises a fast path that GN_DISABLE_FAST=1 turns off")

TOL = 1e-4      # the contract
TIGHT = 2e-5    # what we actually expect


def close(a, b, atol=TIGHT):
    a, b = torch.as_tensor(a).detach().cpu(), torch.as_tensor(b).detach().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all(), "non-finite values in the HIP output"
    err = (a.double() - b.double()).abs().max().item() if a.numel() else 0.0
    assert err <= atol, "max abs err {:.3e} > {:.1e}".format(err, atol)


def load_into(module, state, dev):
    module.load_state_dict(state)
    return module.to(dev)


# ---------------------------------------------------------------------------------------------
def test_library_is_the_hip_one(gpu):
    assert _hip.load().gn_version() == _hip.ABI_VERSION
    assert "libgripnet_hip.so" in open("/proc/self/maps").read()


def test_norm_known_answer(gpu, golden):
    g = golden("norm_known")
    ei2, norm = gripnet_amd.myGCN.norm(g.t("edge_index", gpu), 3, None, improved=True, dtype=torch.float32)
    assert torch.equal(ei2.cpu(), torch.tensor([[0, 1, 0, 1, 2], [1, 0, 0, 1, 2]]))
    close(norm, torch.tensor([1 / 3, 1 / 3, 2 / 3, 2 / 3, 1.0]), 1e-6)


def test_norm_cases_bit_exact_indices(gpu, golden):
    g = golden("norm_cases")
    for i, spec in enumerate(g.meta["cases"]):
        p = "c{}.".format(i)
        w = g.t(p + "edge_weight", gpu) if spec["weighted"] else None
        ei2, norm = gripnet_amd.myGCN.norm(g.t(p + "edge_index", gpu), spec["num_nodes"], w,
                                           improved=spec["improved"])
        assert ei2.dtype == torch.int64
        assert torch.equal(ei2.cpu(), g.t(p + "out.edge_index")), "case {}".format(i)
        close(norm, g.t(p + "out.norm"), 1e-6)


def test_gcn_forward_and_cache(gpu, golden):
    g = golden("gcn_forward")
    conv = load_into(gripnet_amd.myGCN(g.meta["fin"], g.meta["fout"], cached=True),
                     {"weight": g.t("sd.weight"), "bias": g.t("sd.bias")}, gpu)
    ei, w = g.t("edge_index", gpu), g.t("edge_weight", gpu)
    close(conv(g.t("x0", gpu), ei, w), g.t("out.y0"))
    plan = conv.cached_result
    close(conv(g.t("x1", gpu), ei, w), g.t("out.y1"))
    assert conv.cached_result is plan                       # second call served from the cache
    cached_ei, cached_norm = conv.cached_result             # unpacks like the reference's tuple
    ref_ei, ref_norm = orc.gcn_norm(g.t("edge_index"), g.meta["n"], g.t("edge_weight"))
    assert torch.equal(cached_ei.cpu(), ref_ei)
    close(cached_norm, ref_norm, 1e-6)
    with pytest.raises(RuntimeError, match="Cached 300 number of edges, but found 299"):
        conv(g.t("x0", gpu), ei[:, :299], w[:299])
    # same edge count, different edges: the stale graph is reused silently, like the reference
    close(conv(g.t("x0", gpu), ei.flip(0), w), g.t("out.y0"))
    nb = load_into(gripnet_amd.myGCN(g.meta["fin"], 20, cached=False, bias=False), g.state("nb."), gpu)
    close(nb(g.t("x0", gpu), ei), g.t("out.y_nobias_unweighted"))


def test_inter_cases(gpu, golden):
    g = golden("inter_cases")
    x, ei, w = g.t("x", gpu), g.t("edge_index", gpu), g.t("edge_weight", gpu)
    for v in g.meta["variants"]:
        m = gripnet_amd.interGraph(g.meta["source_dim"], v["target_dim"], g.meta["n_target"],
                                   target_feat_dim=v["target_feat_dim"], if_one_external=v["if_one_external"])
        m = load_into(m, g.state(v["tag"] + "."), gpu)
        before = ei.clone()
        y = m(x, ei, w if v["weighted"] else None, if_relu=v["if_relu"], mod=v["mod"])
        assert torch.equal(before, ei)                      # caller's tensor is not mutated
        close(y, g.t(v["tag"] + ".out"))


def test_rgcn_cases(gpu, golden):
    g = golden("rgcn_cases")
    x, ei, et, rl = g.t("x", gpu), g.t("edge_index", gpu), g.t("edge_type", gpu), g.t("range_list", gpu)
    for v in g.meta["variants"]:
        m = gripnet_amd.myRGCN(g.meta["fin"], g.meta["fout"], g.meta["R"], g.meta["B"], v["after_relu"], bias=v["bias"])
        m = load_into(m, g.state(v["tag"] + "."), gpu)
        close(m(x, ei, et, rl), g.t(v["tag"] + ".out"))
        close(m(x, ei, et, rl.cpu()), g.t(v["tag"] + ".out"))     # range_list may stay on the host


def test_homo_cases(gpu, golden):
    g = golden("homo_cases")
    x, ei, w = g.t("x", gpu), g.t("edge_index", gpu), g.t("edge_weight", gpu)
    m = load_into(gripnet_amd.homoGraph([12, 8, 8]), g.state("gcn2."), gpu)
    close(m(x, ei, w, if_catout=True), g.t("gcn2.out_cat"))
    close(m(x, ei, w, if_catout=False), g.t("gcn2.out_nocat"))
    m = load_into(gripnet_amd.homoGraph([10, 16], start_graph=True, in_dim=g.meta["n"]), g.state("start1."), gpu)
    close(m(torch.full((3, 3), 7.0, device=gpu), ei, None, if_catout=True), g.t("start1.out_cat"))
    m = load_into(gripnet_amd.homoGraph([12, 8, 6], multi_relational=True, n_rela=3, n_base=5), g.state("rgcn2."), gpu)
    close(m(x, g.t("rel.edge_index", gpu), edge_type=g.t("rel.edge_type", gpu), range_list=g.t("rel.range_list", gpu),
            if_catout=True), g.t("rgcn2.out_cat"))
    with pytest.raises(AssertionError):
        m(x, g.t("rel.edge_index", gpu), if_catout=True)    # missing edge_type / range_list


def test_decoder_cases(gpu, golden):
    g = golden("decoder_cases")
    z, ei, et = g.t("z", gpu), g.t("edge_index", gpu), g.t("edge_type", gpu)
    dm = load_into(gripnet_amd.multiRelaInnerProductDecoder(g.meta["F"], g.meta["R"]), g.state("dmt."), gpu)
    close(dm(z, ei, et), g.t("dmt.out_sigmoid"))
    close(dm(z, ei, et, sigmoid=False), g.t("dmt.out_logits"))
    mc = load_into(gripnet_amd.multiClassInnerProductDecoder(g.meta["F"], g.meta["n_class"]), g.state("mcip."), gpu)
    close(mc(z, g.t("node_list", gpu)), g.t("mcip.out_softmax"))
    close(mc(z, g.t("node_list", gpu), softmax=False), g.t("mcip.out_logits"))
    # strided z (a column slice of a wider matrix) takes the same path
    wide = torch.zeros(z.shape[0], z.shape[1] + 8, device=gpu)
    wide[:, 4:4 + z.shape[1]] = z
    close(dm(wide[:, 4:4 + z.shape[1]], ei, et), g.t("dmt.out_sigmoid"))


def pose_data_from_golden(g, dev):
    return Data(gg_edge_index=g.t("gg_edge_index", dev), edge_weight=g.t("edge_weight", dev),
                gd_edge_index=g.t("gd_edge_index", dev), train_idx=g.t("train_idx", dev),
                train_et=g.t("train_et", dev), train_range=g.t("train_range", dev))


@pytest.mark.parametrize("scale", ["tiny", "small"])
def test_pose_pipeline_vs_reference(gpu, golden, scale):
    g = golden("pose_" + scale)
    model = load_into(PoseModel(g.meta["n_g"], g.meta["n_d"], g.meta["R"]), g.state("", strip=False), gpu)
    data = pose_data_from_golden(g, gpu)
    with torch.no_grad():
        for _ in range(2):                                   # 2nd pass runs from the cached plans
            z_gg = model.gg(None, data.gg_edge_index, edge_weight=data.edge_weight, if_catout=True)
            z_gd = model.gd(z_gg, data.gd_edge_index, mod="cat", if_relu=True)
            z_dd, score = model(data)
            close(z_gg, g.t("out.z_gg"))
            close(z_gd, g.t("out.z_gd"))
            close(z_dd, g.t("out.z_dd"))
            close(score, g.t("out.score"))
        close(model(data, sigmoid=False)[1], g.t("out.logits"), TOL)


def test_nc_pipelines_vs_reference(gpu, golden):
    g = golden("aminer_tiny")
    m = AminerModel(g.meta["n_p"], g.meta["n_a"], g.meta["n_class"], pp_nhids=g.meta["pp_nhids"],
                    pa_out=g.meta["pa_out"], aa_hidden=g.meta["aa_nhids"][1:])
    m = load_into(m, g.state("", strip=False), gpu)
    data = Data(**{k: g.t(k, gpu) for k in ("pp_edge_idx", "pa_edge_idx", "aa_edge_idx", "pp_edge_weight", "aa_edge_weight")})
    with torch.no_grad():
        z, score = m(data, g.t("node_list", gpu))
    close(z, g.t("out.z"))
    close(score, g.t("out.score"))

    g = golden("freebase_c_tiny")
    m = FreebaseCModel(g.meta["n_p"], g.meta["n_q"], g.meta["n_a"], g.meta["n_class"], pp_nhids=g.meta["pp_nhids"],
                       qq_nhids=g.meta["qq_nhids"], pa_out=g.meta["pa_out"], aa_hidden=g.meta["aa_nhids"][1:])
    sd = g.state("", strip=False)
    sd["aa_embeddings"] = g.t("aa_embeddings")
    m = load_into(m, sd, gpu)
    data = Data(**{k: g.t(k, gpu) for k in ("pp_edge_idx", "pa_edge_idx", "qq_edge_idx", "qa_edge_idx", "aa_edge_idx",
                                            "pp_edge_weight", "qq_edge_weight", "aa_edge_weight")})
    with torch.no_grad():
        z, score = m(data, g.t("node_list", gpu))
    close(z, g.t("out.z"))
    close(score, g.t("out.score"))


def test_freebase_a_and_b_pipelines_vs_reference(gpu, golden):
    """BASELINE.json config 5 callers on the golden fixtures: freebase-a (one supervertex, no concat,
    GripNet-freebase-a.py:94-122) and freebase-b (aminer's call sequence, pa_out halves of equal width,
    GripNet-freebase-b.py:96-135)."""
    g = golden("freebase_a_tiny")
    m = load_into(FreebaseAModel(g.meta["n_a"], g.meta["n_class"], pp_nhids=g.meta["pp_nhids"]), g.state("", strip=False), gpu)
    data = Data(aa_edge_idx=g.t("aa_edge_idx", gpu), aa_edge_weight=g.t("aa_edge_weight", gpu))
    with torch.no_grad():
        z, score = m(data, g.t("node_list", gpu))
        _, logits = m(data, g.t("node_list", gpu), softmax=False)
    close(z, g.t("out.z"))
    close(score, g.t("out.score"))
    close(logits, g.t("out.logits"))
    g = golden("freebase_b_tiny")
    m = FreebaseBModel(g.meta["n_p"], g.meta["n_a"], g.meta["n_class"], pp_nhids=g.meta["pp_nhids"],
                       pa_out=g.meta["pa_out"], aa_hidden=g.meta["aa_nhids"][1:])
    m = load_into(m, g.state("", strip=False), gpu)
    data = Data(**{k: g.t(k, gpu) for k in ("pp_edge_idx", "pa_edge_idx", "aa_edge_idx", "pp_edge_weight", "aa_edge_weight")})
    with torch.no_grad():
        z, score = m(data, g.t("node_list", gpu))
    close(z, g.t("out.z"))
    close(score, g.t("out.score"))


def use_relational_kernel(module, kernel, arithmetic="fp32"):
    """Pins the relational kernel (a flag of gn_rgcn_forward_f32, set per layer) of every myRGCN under `module`."""
    layers = [m for m in module.modules() if isinstance(m, gripnet_amd.myRGCN)]
    for m in layers:
        m.kernel, m.arithmetic = kernel, arithmetic
    return layers


@pytest.fixture(params=["pair", "pair-fast", "lds", "general", "general-forced", "table"])
def kernel_path(request, monkeypatch):
    """Run a test once per relational kernel: destination-major (default; three- and two-term splits), LDS-resident
    accumulator, general = the O(E) basis-space path (with every fast path switched off, and FORCED by flag while the fast
    kernels apply: the workspace then has to be the general path's), table = the [R, N, out] table path forced by flag;
    the last four also take the shuffle-based form of the 16-wide GCN gather instead of the quad form."""
    monkeypatch.setenv("GN_DISABLE_FAST", "1" if request.param == "general" else "0")
    monkeypatch.setenv("GN_DISABLE_QUAD", "0" if request.param in ("pair", "pair-fast") else "1")
    return request.param


def test_both_kernel_paths_on_pose_small(gpu, golden, kernel_path):
    g = golden("pose_small")
    model = load_into(PoseModel(g.meta["n_g"], g.meta["n_d"], g.meta["R"]), g.state("", strip=False), gpu)
    layers = use_relational_kernel(model, kernel_path.split("-")[0], "fast" if kernel_path.endswith("fast") else "fp32")
    data = pose_data_from_golden(g, gpu)
    with torch.no_grad():
        z_dd, score = model(data)
        logits = model(data, sigmoid=False)[1]
    for m in layers:                                          # the kernel asked for is the kernel that ran
        assert m._plan.path(m.in_channels, m.out_channels, m.num_bases, m._fast(), m.kernel) == kernel_path.split("-")[0]
    close(z_dd, g.t("out.z_dd"))
    close(score, g.t("out.score"))
    close(logits, g.t("out.logits"), TOL)
    # unsorted relation ids and a shuffled edge list must give the same scores, permuted
    perm = torch.randperm(data.train_idx.shape[1], device=gpu)
    with torch.no_grad():
        shuffled = model.dmt(z_dd, data.train_idx[:, perm], data.train_et[perm])
    close(shuffled, g.t("out.score")[perm.cpu()])


# ---- error behaviour -------------------------------------------------------------------------
def test_errors(gpu):
    ei = torch.tensor([[0, 1, 5], [1, 0, 2]], device=gpu)
    with pytest.raises(IndexError):
        gripnet_amd.myGCN.norm(ei, 4, None)
    conv = gripnet_amd.myRGCN(4, 4, 2, 2, False).to(gpu)
    x = torch.randn(6, 4, device=gpu)
    with pytest.raises(ValueError, match="range_list"):
        conv(x, ei, None, torch.tensor([[0, 2], [1, 3]]))          # overlapping ranges
    with pytest.raises(ValueError, match="range_list"):
        conv(x, ei, None, torch.tensor([[0, 1], [1, 2]]))          # does not cover E
    dm = gripnet_amd.multiRelaInnerProductDecoder(4, 2).to(gpu)
    out = dm(x, ei, torch.tensor([0, 1, 9], device=gpu))          # relation 9 does not exist
    assert torch.isnan(out[2]) and torch.isfinite(out[:2]).all()
    with pytest.raises(IndexError):
        _hip.raise_if_index_errors(gpu)
    _hip.raise_if_index_errors(gpu)                                # flag was cleared
    # the metric helpers are the epoch loop's synchronisation points: the deferred IndexError surfaces there
    out = dm(x, torch.tensor([[0, 7], [1, 0]], device=gpu), torch.tensor([0, 1], device=gpu))   # node 7 of 6
    assert torch.isnan(out[1])
    with pytest.raises(IndexError):
        gripnet_amd.utils.relation_metrics(out[:1], out[:1], torch.tensor([[0, 1]]))
    out = dm(x, torch.tensor([[0, 7], [1, 0]], device=gpu), torch.tensor([0, 1], device=gpu))
    with pytest.raises(IndexError):
        gripnet_amd.utils.auprc_auroc_ap(torch.tensor([1.0, 0.0], device=gpu), out)


def test_empty_inputs(gpu):
    dm = gripnet_amd.multiRelaInnerProductDecoder(8, 2).to(gpu)
    out = dm(torch.randn(3, 8, device=gpu), torch.zeros(2, 0, dtype=torch.long, device=gpu),
             torch.zeros(0, dtype=torch.long, device=gpu))
    assert out.shape == (0,)
    conv = gripnet_amd.myGCN(5, 3).to(gpu)                          # no edges: self loops only
    x = torch.randn(4, 5, device=gpu)
    y = conv(x, torch.zeros(2, 0, dtype=torch.long, device=gpu))
    close(y, x.cpu() @ conv.weight.detach().cpu() + conv.bias.detach().cpu())
    rg = gripnet_amd.myRGCN(5, 3, 2, 2, False).to(gpu)              # no edges: root term only
    y = rg(x, torch.zeros(2, 0, dtype=torch.long, device=gpu), None, torch.zeros(2, 2, dtype=torch.long))
    close(y, x.cpu() @ rg.root.detach().cpu())


# ---- randomised cross-checks against the oracle -----------------------------------------------
@pytest.mark.parametrize("seed", range(6))
def test_random_graphs_vs_oracle(gpu, seed):
    gen = torch.Generator().manual_seed(1000 + seed)
    n = int(torch.randint(1, 400, (1,), generator=gen))
    e = int(torch.randint(0, 5000, (1,), generator=gen))
    fin = [3, 16, 33, 48, 64, 7][seed]
    fout = [5, 16, 32, 8, 128, 12][seed]
    ei = torch.randint(0, n, (2, e), generator=gen)
    w = torch.rand(e, generator=gen) + 0.1
    x = torch.randn(n, fin, generator=gen)
    conv = gripnet_amd.myGCN(fin, fout, improved=bool(seed % 2)).to(gpu)
    conv.bias.data.normal_()
    y = conv(x.to(gpu), ei.to(gpu), w.to(gpu))
    close(y, orc.gcn_forward(x, conv.weight.detach().cpu(), conv.bias.detach().cpu(), ei, w, improved=bool(seed % 2)))

    R, B = [1, 3, 7, 2, 5, 4][seed], [2, 4, 3, 8, 1, 6][seed]
    sizes = torch.randint(0, 900, (R,), generator=gen).tolist()
    blocks = [torch.randint(0, n, (2, s), generator=gen) for s in sizes]
    rei = torch.cat(blocks, dim=1)
    rl = gripnet_amd.utils.get_range_list(blocks)
    rg = gripnet_amd.myRGCN(fin, fout, R, B, False, bias=bool(seed % 2)).to(gpu)
    y = rg(x.to(gpu), rei.to(gpu), None, rl)
    sd = {k: v.detach().cpu() for k, v in rg.state_dict().items()}
    close(y, orc.rgcn_forward(x, rei, rl, sd["basis"], sd["att"], sd["root"], sd.get("bias")))

    dm = gripnet_amd.multiRelaInnerProductDecoder(fout, R).to(gpu)
    et = torch.randint(0, R, (rei.shape[1],), generator=gen)
    z = torch.randn(n, fout, generator=gen)
    close(dm(z.to(gpu), rei.to(gpu), et.to(gpu), sigmoid=False),
          orc.distmult(z, rei, et, dm.weight.detach().cpu(), sigmoid=False))


@pytest.mark.parametrize("n,fin,fout", [(5000, 64, 64), (4100, 128, 128), (6001, 64, 128), (4097, 64, 80)])
def test_gcn_wide_layer_contracts_in_the_gather_launch(gpu, n, fin, fout):
    """Layers at least as wide as their input (the second layer of the NC stacks): gn_graph_aggregate_f32(weight=W) gathers
    the input rows and contracts with W on the matrix cores in the same launch (k_aggregate_mfma) - weighted edges, a few
    long rows, isolated nodes, a ragged last row block, bias + ReLU, written into a column slice of a wider matrix."""
    gen = torch.Generator().manual_seed(n + fout)
    ei = torch.randint(0, n - 37, (2, 9 * n), generator=gen)                 # the last 37 nodes stay isolated
    hub = torch.stack([torch.randint(0, n - 37, (700,), generator=gen), torch.full((700,), 11)])
    ei = torch.cat([ei, hub], dim=1)
    w = torch.rand(ei.shape[1], generator=gen) + 0.1
    x = torch.randn(n, fin, generator=gen)
    conv = gripnet_amd.myGCN(fin, fout).to(gpu)
    conv.bias.data.normal_()
    wide = torch.full((n, fout + 8), float("nan"), device=gpu)
    with torch.no_grad():                                                    # (slot-fused outputs are the inference path)
        y = conv(x.to(gpu), ei.to(gpu), w.to(gpu), _relu=True, _out=wide[:, 4:4 + fout])
    plan = _hip.GraphPlan.gcn(ei.to(gpu), n, w.to(gpu))
    assert plan.transform_ok(fin, fout, x.to(gpu))
    ref = torch.relu(orc.gcn_forward(x, conv.weight.detach().cpu(), conv.bias.detach().cpu(), ei, w))
    close(y, ref)
    assert torch.isnan(wide[:, :4]).all() and torch.isnan(wide[:, 4 + fout:]).all()


@pytest.mark.parametrize("fout", [16, 32, 64, 128, 20])
def test_gcn_skewed_degrees_and_empty_rows(gpu, fout):
    """Destination-major aggregation on skewed graphs, one case per lanes-per-neighbour specialisation:
    a hub with thousands of neighbours, isolated nodes, trailing targets without edges, duplicate edges;
    square (self loops) and bipartite graphs."""
    gen = torch.Generator().manual_seed(77 + fout)
    n, fin = 700, 24
    ei = torch.randint(0, n - 50, (2, 6000), generator=gen)             # nodes >= n-50 stay isolated
    hub = torch.stack([torch.randint(0, n - 50, (3000,), generator=gen), torch.full((3000,), 5)])
    ei = torch.cat([ei, hub, ei[:, :200]], dim=1)
    w = torch.rand(ei.shape[1], generator=gen) + 0.1
    x = torch.randn(n, fin, generator=gen)
    conv = gripnet_amd.myGCN(fin, fout).to(gpu)
    conv.bias.data.normal_()
    y = conv(x.to(gpu), ei.to(gpu), w.to(gpu), _relu=True)
    ref = torch.relu(orc.gcn_forward(x, conv.weight.detach().cpu(), conv.bias.detach().cpu(), ei, w))
    close(y, ref)
    # bipartite: 300 sources -> 90 targets, the last 20 targets and target 3 have no edge at all
    src = torch.randint(0, 300, (2500,), generator=gen)
    tgt = torch.randint(0, 70, (2500,), generator=gen)
    tgt[tgt == 3] = 4
    tgt[:1200] = 9                                                       # a hub target
    bei = torch.stack([src, tgt])
    ig = gripnet_amd.interGraph(fin, fout, 90, target_feat_dim=fout, if_one_external=False).to(gpu)
    ig.conv.bias.data.normal_()
    xs = torch.randn(300, fin, generator=gen)
    yb = ig(xs.to(gpu), bei.to(gpu), if_relu=False)
    sd = {"g.conv.weight": ig.conv.weight.detach().cpu(), "g.conv.bias": ig.conv.bias.detach().cpu()}
    close(yb, orc.inter_forward_closed(sd, "g.", xs, bei, 90))


@pytest.mark.parametrize("n,fin,bases", [(40, 16, 3), (200, 32, 5), (560, 48, 32), (645, 64, 8), (900, 48, 4),
                                         (1000, 32, 2), (1, 16, 1)])
@pytest.mark.parametrize("path", ["pair", "pair-fast", "lds", "general", "table"])
def test_rgcn_lds_resident_shapes(gpu, n, fin, bases, path):
    """Every specialisation of the three LDS-resident relational kernels (destination-major: one or two bases per lane,
    one to four feature tiles, one to three rows per workgroup, a pair run longer than one unit; register-accumulated:
    K depth, row groups, chunked long runs; LDS accumulator: row tiles per wave, 1..16 source tiles): empty
    relations, a relation longer than one work item, duplicate edges, destinations with no in-edges;
    checked against the oracle and for run-to-run equality.  Shapes a kernel does not cover fall
    through to the next one."""
    gen = torch.Generator().manual_seed(n * 131 + fin)
    torch.manual_seed(n * 17 + fin)                                   # layer weights come from the global RNG
    sizes = [0, 9000, 3, 0, 700, 1, 2500, 0]
    blocks = [torch.randint(0, max(1, n - n // 7), (2, s), generator=gen) for s in sizes]   # top ids never a dst/src
    blocks[4] = torch.cat([blocks[4], blocks[4][:, :50]], dim=1)                          # duplicate edges
    rei = torch.cat(blocks, dim=1)
    rl = gripnet_amd.utils.get_range_list(blocks)
    x = torch.randn(n, fin, generator=gen)
    rg = gripnet_amd.myRGCN(fin, 32, len(sizes), bases, False, bias=True).to(gpu)
    rg.bias.data.normal_()
    use_relational_kernel(rg, path.split("-")[0], "fast" if path.endswith("fast") else "fp32")
    y = rg(x.to(gpu), rei.to(gpu), None, rl, _relu=True)
    y2 = rg(x.to(gpu), rei.to(gpu), None, rl, _relu=True)
    assert torch.equal(y, y2)
    if path == "pair" and n <= 768 and not (fin == 64 and bases > 16):
        assert rg._plan.path(fin, 32, bases) == "pair"        # the default arithmetic takes the destination-major kernel
    # float64 oracle: with thousands of edges into one destination (n == 1: all 12,000 of them) the
    # reference's own sequential fp32 sum is off by ~1e-4; the kernel folds its running sums every 64
    # addends, so it is held to the tight bar against the exact result
    sd = {k: v.detach().cpu().double() for k, v in rg.state_dict().items()}
    ref = torch.relu(orc.rgcn_forward(x.double(), rei, rl, sd["basis"], sd["att"], sd["root"], sd.get("bias")))
    close(y, ref.float())


@needs_fast_paths
@pytest.mark.parametrize("n,fin,bases", [(645, 48, 32), (300, 16, 5), (520, 32, 16), (700, 64, 8), (90, 48, 20)])
def test_relational_layer_as_two_launches_gives_the_same_bits(gpu, n, fin, bases):
    """Round 6: GN_RGCN_PAIR_SUMS_ONLY (the x-independent half - the att rows of every (destination, source) pair's edges summed -
    on the side stream) + GN_RGCN_PAIR_SUMS_READY (the contraction with x, ordered behind it by gn_stream_order) = the bits of
    the one-launch layer; the sums follow att (recomputed by every start_pair_sums, never kept); a note that was not consumed
    by the next forward is dropped by the next start; arithmetic "fast", another kernel and training refuse the split."""
    gen = torch.Generator().manual_seed(n * 5 + fin)
    torch.manual_seed(n + fin + bases)
    sizes = [3000, 0, 40, 900, 1, 600]
    blocks = [torch.randint(0, max(1, n - n // 9), (2, s), generator=gen) for s in sizes]
    rei = torch.cat(blocks, dim=1).to(gpu)
    rl = gripnet_amd.utils.get_range_list(blocks)
    x = torch.randn(n, fin, generator=gen).to(gpu)
    rg = gripnet_amd.myRGCN(fin, 32, len(sizes), bases, False, bias=True).to(gpu)
    rg.bias.data.normal_()
    for prm in rg.parameters():
        prm.requires_grad_(False)
    planes = _hip.SplitPlanes(n, fin // 16, gpu).fill_from(x)
    planes.tag(x)
    with torch.no_grad():
        y_one = rg(x, rei, None, rl, _relu=True)
        assert rg.start_pair_sums(rei, rl, n) is True
        y_two = rg(x, rei, None, rl, _relu=True)
        assert torch.equal(y_one, y_two)
        assert rg.__dict__.get("_sums_pending") is None                  # consumed
        # the sums are this step's: att changed in place -> both forms follow, and still agree bit for bit
        rg.att.data.mul_(1.5)
        y_one2 = rg(x, rei, None, rl, _relu=True)
        assert not torch.equal(y_one2, y_one)
        assert rg.start_pair_sums(rei, rl, n)
        assert torch.equal(rg(x, rei, None, rl, _relu=True), y_one2)
        # garbage in the buffer must show (the second launch really reads it) ...
        assert rg.start_pair_sums(rei, rl, n)
        torch.cuda.synchronize()
        rg._plan._sums.zero_()
        assert not torch.equal(rg(x, rei, None, rl, _relu=True), y_one2)
        # ... a forward without a start in front of it is the one-launch layer again
        assert torch.equal(rg(x, rei, None, rl, _relu=True), y_one2)
        # what the split does not cover refuses it (and the layer stays one launch)
        rg.arithmetic = "fast"
        assert rg.start_pair_sums(rei, rl, n) is False
        rg.arithmetic = "fp32"
        rg.kernel = "general"
        assert rg.start_pair_sums(rei, rl, n) is False
        rg.kernel = "auto"
    sd = {k: v.detach().cpu().double() for k, v in rg.state_dict().items()}
    ref = torch.relu(orc.rgcn_forward(x.cpu().double(), rei.cpu(), rl, sd["basis"], sd["att"], sd["root"], sd.get("bias")))
    close(y_one2, ref.float(), TIGHT)
    for prm in rg.parameters():
        prm.requires_grad_(True)
    assert rg.start_pair_sums(rei, rl, n) is False                       # training: the autograd path keeps the one-launch layer


def test_relational_layer_on_changing_edge_lists(gpu):
    """myRGCN.static_graph = False (round 6): every new edge list gets a LIGHT plan (GN_RGCN_PLAN_LIGHT: the device-sorted key
    list only, none of the host-built schedules) and runs on the general O(E) kernel - the reference's myRGCN has no set-up
    cost (layers.py:165-169).  Two different lists in turn, each against the oracle; the same layer with static_graph = True
    agrees to rounding."""
    gen = torch.Generator().manual_seed(77)
    n, fin, R, bases = 500, 48, 6, 32
    x = torch.randn(n, fin, generator=gen).to(gpu)
    rg = gripnet_amd.myRGCN(fin, 32, R, bases, False).to(gpu)
    for prm in rg.parameters():
        prm.requires_grad_(False)
    sd = {k: v.detach().cpu() for k, v in rg.state_dict().items()}
    lists = []
    for k in range(2):
        blocks = [torch.randint(0, n, (2, s), generator=gen) for s in (2000 + 100 * k, 0, 30, 700, 1, 400)]
        lists.append((torch.cat(blocks, dim=1).to(gpu), gripnet_amd.utils.get_range_list(blocks)))
    rg.static_graph = False
    for rep in range(2):
        for rei, rl in lists:
            y = rg(x, rei, None, rl, _relu=True)
            assert rg._plan.light and rg._plan.path(fin, 32, bases) == "general"
            ref = torch.relu(orc.rgcn_forward(x.cpu(), rei.cpu(), rl, sd["basis"], sd["att"], sd["root"], None))
            close(y, ref, TIGHT)
    rg.static_graph = True
    rei, rl = lists[0]
    y_full = rg(x, rei, None, rl, _relu=True)
    assert not rg._plan.light
    close(y_full, torch.relu(orc.rgcn_forward(x.cpu(), rei.cpu(), rl, sd["basis"], sd["att"], sd["root"], None)), TIGHT)


@needs_fast_paths
def test_pose_forward_with_the_split_relational_layer(gpu):
    """PoseModel.split_relational: the whole forward with the pair sums started in front of the gene layers - eager modules,
    their memoised replays and the recorded step - gives the bits of the default forward, step after step."""
    from gripnet_amd.pipeline import PoseStages
    data = make_pose("small", n_d=300, e_dd_dir=60000).to(gpu)
    torch.manual_seed(1111)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(gpu)
    with torch.no_grad():
        z0, s0 = model(data)
        z0, s0 = z0.clone(), s0.clone()
        model.split_relational = True
        for _ in range(4):                                               # first sighting, recording, replays
            z, s = model(data)
            assert torch.equal(z, z0) and torch.equal(s, s0)
        rec = PoseStages(model, data, recorded=True)
        names = [c[3] or c[2] for c in rec._whole.calls]
        assert "gn_rgcn_forward_f32[pair sums]" in names and names.count("gn_stream_order") == 2
        for _ in range(3):
            z, s = rec.step()
            assert torch.equal(z, z0) and torch.equal(s, s0)
        model.split_relational = False
        z, s = model(data)
        assert torch.equal(z, z0) and torch.equal(s, s0)


@needs_fast_paths
@pytest.mark.parametrize("n,fin,bases,arith", [(645, 48, 32, "fp32"), (300, 16, 5, "fp32"), (520, 32, 16, "fp32"),
                                               (700, 64, 8, "fp32"), (645, 48, 32, "fast"), (90, 48, 20, "fp32")])
def test_rgcn_takes_x_as_split_planes(gpu, n, fin, bases, arith):
    """gn_split_planes: the relational kernel fed with the bf16 terms of x (cut exactly as it cuts them itself) gives the
    same bits as the kernel that splits x in every unit - for planes written by gn_split_planes_f32 and for planes left
    by the external layer's own launch (its output and its |target_feat| side copy); planes that no longer describe the
    tensor (rewritten, or the tensor modified through torch) are not used."""
    gen = torch.Generator().manual_seed(n * 7 + fin)
    torch.manual_seed(n + fin + bases)
    sizes = [3000, 0, 40, 900, 1, 600]
    blocks = [torch.randint(0, max(1, n - n // 9), (2, s), generator=gen) for s in sizes]
    rei = torch.cat(blocks, dim=1).to(gpu)
    rl = gripnet_amd.utils.get_range_list(blocks)
    x = torch.randn(n, fin, generator=gen).to(gpu)
    x[3, 1] = 0.0
    x[5] = 1e-30                                                       # small values: the terms underflow to zero together
    rg = gripnet_amd.myRGCN(fin, 32, len(sizes), bases, False, bias=True).to(gpu)
    rg.bias.data.normal_()
    rg.arithmetic = arith
    rg.kernel = "pair"
    for prm in rg.parameters():
        prm.requires_grad_(False)                                      # the inference path (autograd builds its own)
    y_plain = rg(x, rei, None, rl, _relu=True)
    assert rg._plan.path(fin, 32, bases, fast=arith == "fast", path="pair") == "pair"
    planes = _hip.SplitPlanes(n, fin // 16, gpu).fill_from(x)
    planes.tag(x)
    assert _hip.SplitPlanes.of(x, fin // 16) is planes
    y_planes = rg(x, rei, None, rl, _relu=True)
    assert torch.equal(y_plain, y_planes)
    # garbage in the planes must show (the tagged planes are really what the kernel read) ...
    keep = planes.buf.clone()
    planes.buf[: planes.buf.numel() - 16 * 4 * ((3 * (fin // 16) + 1) // 2)].zero_()
    assert not torch.equal(rg(x, rei, None, rl, _relu=True), y_plain)
    planes.buf.copy_(keep)
    # ... and a tensor modified through torch drops them
    x.add_(0.0)
    assert _hip.SplitPlanes.of(x, fin // 16) is None
    assert torch.equal(rg(x, rei, None, rl, _relu=True), y_plain)
    sd = {k: v.detach().cpu().double() for k, v in rg.state_dict().items()}
    ref = torch.relu(orc.rgcn_forward(x.cpu().double(), rei.cpu(), rl, sd["basis"], sd["att"], sd["root"], sd.get("bias")))
    close(y_planes, ref.float(), TIGHT if arith == "fp32" else TOL)


@needs_fast_paths
def test_external_layer_leaves_split_planes(gpu):
    """interGraph(mod="cat") leaves [y | |target_feat|] as split planes in the launch that computes y; they are the bytes
    gn_split_planes_f32 makes of the finished matrix, and the relational layer behind it uses them (pose pipeline)."""
    gen = torch.Generator().manual_seed(5)
    torch.manual_seed(9)
    n_src, n_tgt = 700, 333
    ig = gripnet_amd.interGraph(64, 16, n_tgt, target_feat_dim=32).to(gpu)
    xs = torch.randn(n_src, 64, generator=gen).to(gpu)
    bei = torch.stack([torch.randint(0, n_src, (4000,), generator=gen), torch.randint(0, n_tgt, (4000,), generator=gen)]).to(gpu)
    with torch.no_grad():
        out = ig(xs, bei, mod="cat", if_relu=True)
    planes = _hip.SplitPlanes.of(out, 3)
    assert planes is not None and planes is ig._planes
    ref = _hip.SplitPlanes(n_tgt, 3, gpu).fill_from(out)
    assert torch.equal(planes.buf, ref.buf)
    assert int(planes.buf[-16 * 20:].abs().sum()) == 0                 # the zero row stays zero
    with torch.no_grad():
        out2 = ig(xs * 2.0, bei, mod="cat", if_relu=True)             # the module's planes now describe out2, not out
    assert _hip.SplitPlanes.of(out, 3) is None and _hip.SplitPlanes.of(out2, 3) is planes


@pytest.mark.parametrize("n,fin,fout,bases", [(700, 32, 48, 32), (645, 16, 64, 8), (300, 48, 4, 3), (768, 32, 44, 32)])
def test_rgcn_destination_major_output_widths(gpu, n, fin, fout, bases):
    """The destination-major relational kernel on output widths other than 32 (any multiple of 4 up to 64): above 256 nodes a
    workgroup owns up to three rows, i.e. more than 128 (row, output) pairs for outputs wider than 42 - two passes of its
    final stage.  32 -> 48 with 32 bases is the reversed layer of the PoSE backward (dx of the 48 -> 32 drug layer)."""
    gen = torch.Generator().manual_seed(n + fout)
    torch.manual_seed(n * 3 + fout)
    sizes = [5000, 0, 2, 1500, 800]
    blocks = [torch.randint(0, n, (2, s), generator=gen) for s in sizes]
    rei = torch.cat(blocks, dim=1)
    rl = gripnet_amd.utils.get_range_list(blocks)
    x = torch.randn(n, fin, generator=gen)
    rg = gripnet_amd.myRGCN(fin, fout, len(sizes), bases, False, bias=True).to(gpu)
    rg.bias.data.normal_()
    y = rg(x.to(gpu), rei.to(gpu), None, rl, _relu=True)
    assert rg._plan.path(fin, fout, bases) == "pair"
    assert torch.equal(y, rg(x.to(gpu), rei.to(gpu), None, rl, _relu=True))
    sd = {k: v.detach().cpu().double() for k, v in rg.state_dict().items()}
    ref = torch.relu(orc.rgcn_forward(x.double(), rei, rl, sd["basis"], sd["att"], sd["root"], sd.get("bias")))
    close(y, ref.float())


@pytest.mark.parametrize("n,fin,fout,bases", [(645, 32, 48, 32), (300, 16, 64, 8), (700, 48, 20, 5), (645, 64, 32, 8)])
def test_rgcn_reads_a_transposed_basis(gpu, n, fin, fout, bases):
    """GN_RGCN_BASIS_TRANSPOSED: the destination-major kernel on `basis` stored [bases, out, in] - how the reversed layer of the
    backward sees the forward's parameter (autograd.rgcn_edge_gradients: dx = the layer on the reversed graph with W_r^T) - gives
    the bits of the same call on a transposed copy; the other kernels refuse the flag."""
    gen = torch.Generator().manual_seed(n + fout)
    sizes = [4000, 0, 3, 1200, 900]
    blocks = [torch.randint(0, n, (2, s), generator=gen) for s in sizes]
    rei = torch.cat(blocks, dim=1).to(gpu)
    rl = gripnet_amd.utils.get_range_list(blocks)
    x = torch.randn(n, fin, generator=gen).to(gpu)
    stored = torch.randn(bases, fout, fin, generator=gen).to(gpu)          # [bases, out, in]
    att = torch.randn(len(sizes), bases, generator=gen).to(gpu)
    plan = _hip.RgcnPlan(rei, rl, n)
    assert plan.path(fin, fout, bases) == "pair"
    want, got = torch.empty(n, fout, device=gpu), torch.full((n, fout), float("nan"), device=gpu)
    plan.forward(x, stored.transpose(1, 2).contiguous(), att, None, None, False, want, partial=True)
    plan.forward(x, stored, att, None, None, False, got, partial=True, basis_transposed=True)
    assert torch.equal(got, want)
    with pytest.raises(RuntimeError):
        plan.forward(x, stored, att, None, None, False, got, partial=True, basis_transposed=True, path="general")


def test_plan_builders_host_scratch_is_kept_and_released(gpu):
    """The builders' large host arrays come out of one kept block (host_layout.hpp: HostArena): the first decoder plan of a process
    sizes it, the next ones use it, gn_host_scratch_release gives it back - and the plans are the same plans throughout."""
    gen = torch.Generator().manual_seed(5)
    n, R = 300, 12
    blocks = [gripnet_amd.utils.to_bidirection(torch.randint(0, n, (2, 30000 // (r + 1)), generator=gen)) for r in range(R)]
    ei = torch.cat(blocks, dim=1).to(gpu)
    et = torch.cat([torch.full((b.shape[1],), r, dtype=torch.int64) for r, b in enumerate(blocks)]).to(gpu)
    z, d = torch.randn(n, 32, generator=gen).to(gpu), torch.randn(R, 32, generator=gen).to(gpu)
    _hip.release_host_scratch()
    scores = []
    for k in range(3):
        plan = _hip.DistMultPlan(ei, et, n, R, 32)
        out = torch.empty(ei.shape[1], device=gpu)
        plan.forward(z, d, True, out)
        scores.append(out)
        if k == 1:
            assert _hip.release_host_scratch() > 0             # the block the second build used
            assert _hip.release_host_scratch() == 0
    assert torch.equal(scores[0], scores[1]) and torch.equal(scores[1], scores[2])


@pytest.mark.parametrize("formulation", ["pair", "lds"])
def test_rgcn_sharded_partials_sum_to_full(gpu, formulation):
    """G edge-range shards, un-normalised partials summed, then finalised == unsharded layer
    (the multi-GPU contract, here run sequentially on one device); with either LDS-resident
    relational kernel."""
    data = make_pose("small").to(gpu)
    n, fin, fout, R = data.n_d_node, 48, 32, data.n_dd_edge_type
    conv = gripnet_amd.myRGCN(fin, fout, R, 32, False, bias=True).to(gpu)
    conv.bias.data.normal_()
    conv.kernel = formulation
    x = torch.randn(n, fin, device=gpu)
    full = conv(x, data.train_idx, None, data.train_range, _relu=True)
    for world in (2, 3, 8):
        total = torch.zeros(n, fout, device=gpu)
        for lo, hi in gripnet_amd.utils.shard_edge_ranges(data.train_idx.shape[1], world):
            plan = _hip.RgcnPlan(data.train_idx, data.train_range, n, lo, hi)
            part = torch.empty(n, fout, device=gpu)
            plan.forward(x, conv.basis, conv.att, None, None, False, part, partial=True, path=formulation)
            total += part
        out = torch.empty(n, fout, device=gpu)
        plan.finalize(total, x, conv.root, conv.bias, True, out)
        close(out, full)


# ---- full-size checks (BASELINE.json configs) ---------------------------------------------------
def test_pose0_syn_vs_oracle(gpu):
    data = make_pose("pose0-syn")
    torch.manual_seed(1111)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ref = orc.pose_forward(sd, data.gg_edge_index, data.edge_weight, data.gd_edge_index, data.train_idx,
                           data.train_et, data.train_range, sigmoid=False)
    model = model.to(gpu)
    data = data.to(gpu)
    with torch.no_grad():
        z, logits = model(data, sigmoid=False)
        _, score = model(data)
    close(z, ref["z_dd"], TOL)
    close(logits, ref["score"], TOL)
    close(score, torch.sigmoid(ref["score"]), TOL)
    _hip.raise_if_index_errors(gpu)


def test_pose1_syn_vs_oracle(gpu):
    """BASELINE.json config 4, first rung: pose1-syn (E_dd = 4.0 M) against the CPU oracle."""
    data = make_pose("pose1-syn")
    torch.manual_seed(1111)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ref = orc.pose_forward(sd, data.gg_edge_index, data.edge_weight, data.gd_edge_index, data.train_idx,
                           data.train_et, data.train_range, sigmoid=False)
    model = model.to(gpu)
    data = data.to(gpu)
    with torch.no_grad():
        z, logits = model(data, sigmoid=False)
    close(z, ref["z_dd"], TOL)
    close(logits, ref["score"], TOL)
    _hip.raise_if_index_errors(gpu)


def test_pose2_syn_logits_vs_oracle(gpu):
    """BASELINE.json config 4, largest rung: one oracle forward of pose2-syn (E_dd = 8.4 M; a few seconds and
    ~8 GB of temporaries on the host) against the HIP logits and z."""
    data = make_pose("pose2-syn")
    torch.manual_seed(1111)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = orc.pose_forward(sd, data.gg_edge_index, data.edge_weight, data.gd_edge_index, data.train_idx,
                               data.train_et, data.train_range, sigmoid=False)
    model = model.to(gpu)
    data = data.to(gpu)
    with torch.no_grad():
        z, logits = model(data, sigmoid=False)
    close(z, ref["z_dd"], TOL)
    close(logits, ref["score"], TOL)
    _hip.raise_if_index_errors(gpu)


def test_pose2_syn_properties(gpu):
    """Largest configuration: size-independent properties instead of a CPU run."""
    data = make_pose("pose2-syn").to(gpu)
    torch.manual_seed(1111)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(gpu)
    with torch.no_grad():
        z, score = model(data)
        z2, score2 = model(data)
    assert torch.equal(z, z2) and torch.equal(score, score2)            # run-to-run reproducible
    assert torch.isfinite(z).all() and torch.isfinite(score).all()
    assert (z >= 0).all()                                              # cat of ReLU / abs blocks
    # DistMult is symmetric in (u, v) and every relation block is cat(fwd, reversed fwd)
    rl = data.train_range.cpu()
    for r in (0, 1, 17, rl.shape[0] - 1):
        s, e = int(rl[r, 0]), int(rl[r, 1])
        half = (e - s) // 2
        assert torch.equal(score[s:s + half], score[s + half:e])
    # decoder on a random subset == decoder on the full list, gathered
    pick = torch.randint(0, data.train_idx.shape[1], (100000,), device=gpu)
    with torch.no_grad():
        sub = model.dmt(z, data.train_idx[:, pick], data.train_et[pick])
    assert torch.equal(sub, score[pick])
    # RGCN is linear in the edge set: two half shards add up to the layer's un-normalised sum
    conv = model.dd.conv_list[0]
    x = z[:, :48].contiguous()
    n, E = data.n_d_node, data.train_idx.shape[1]
    parts = []
    for lo, hi in ((0, E // 3), (E // 3, E)):
        plan = _hip.RgcnPlan(data.train_idx, data.train_range, n, lo, hi)
        p = torch.empty(n, 32, device=gpu)
        plan.forward(x, conv.basis, conv.att, None, None, False, p, partial=True)
        parts.append(p)
    out = torch.empty(n, 32, device=gpu)
    plan.finalize(parts[0] + parts[1], x, conv.root, None, True, out)
    close(out, z[:, 48:], TOL)
    _hip.raise_if_index_errors(gpu)


# ---- device-side negative sampling ---------------------------------------------------
def test_device_negative_sampling(gpu):
    """Same contract as gripnet/utils.py:98-119: one pair per positive edge, drawn from the n^2 pairs of
    its relation block that are not positives of that block; deterministic per seed, uniform."""
    from gripnet_amd.utils import device_negative_sampler
    gen = torch.Generator().manual_seed(71)
    n, sizes = 40, [600, 0, 1500, 7, 300]                  # block 2 holds almost every pair: many redraws
    blocks = [torch.randint(0, n, (2, s), generator=gen) for s in sizes]
    pos = torch.cat(blocks, dim=1)
    rl = gripnet_amd.utils.get_range_list(blocks)
    sampler = device_negative_sampler(pos.to(gpu), n, rl)
    neg = sampler.sample(seed=5)
    assert neg.shape == pos.shape and neg.dtype == torch.int64 and neg.is_cuda
    assert int(neg.min()) >= 0 and int(neg.max()) < n
    negc = neg.cpu()
    for r, (s, e) in enumerate(rl.tolist()):
        pos_keys = set((pos[0, s:e] * n + pos[1, s:e]).tolist())
        neg_keys = (negc[0, s:e] * n + negc[1, s:e]).tolist()
        assert not pos_keys.intersection(neg_keys), "relation {} drew a positive pair".format(r)
    assert torch.equal(sampler.sample(seed=5), neg)                       # deterministic in the seed
    assert not torch.equal(sampler.sample(seed=6), neg)
    _hip.raise_if_index_errors(gpu)
    # untyped form and uniformity on a sparse positive set: every row / column about equally likely
    big = torch.randint(0, 500, (2, 200000), generator=gen)
    s2 = device_negative_sampler(big.to(gpu), 500)
    draw = s2.sample(seed=1).cpu()
    for axis in (0, 1):
        counts = torch.bincount(draw[axis], minlength=500).double()
        assert counts.min() > 250 and counts.max() < 560                  # mean 400, sigma 20
    with pytest.raises(IndexError):
        device_negative_sampler(torch.tensor([[0, 41], [1, 2]], device=gpu), n)


# ---- per-relation metrics on the device -------------------------------------------------
def test_relation_metrics_match_sklearn(gpu):
    """AUPRC / AUROC / AP of every relation block at once against scikit-learn called per block, the way
    GripNet-pose.py:148-160 does it; heavy ties, a tiny block, an empty block."""
    from gripnet_amd.utils import auprc_auroc_ap, relation_metrics
    gen = torch.Generator().manual_seed(83)
    sizes = [400, 3, 0, 2500, 1, 90]
    rl = gripnet_amd.utils.get_range_list([torch.zeros(2, s) for s in sizes])
    E = sum(sizes)
    pos = torch.sigmoid(torch.randn(E, generator=gen) + 0.7)
    neg = torch.sigmoid(torch.randn(E, generator=gen) - 0.3)
    pos[:300] = torch.round(pos[:300] * 8) / 8                       # many tied scores inside relation 0
    neg[:300] = torch.round(neg[:300] * 8) / 8
    auprc, auroc, ap = relation_metrics(pos.to(gpu), neg.to(gpu), rl)
    for r, (s, e) in enumerate(rl.tolist()):
        if e == s:
            assert torch.isnan(auprc[r]) and torch.isnan(auroc[r]) and torch.isnan(ap[r])
            continue
        score = torch.cat([pos[s:e], neg[s:e]])
        target = torch.cat([torch.ones(e - s), torch.zeros(e - s)])
        ref = auprc_auroc_ap(target, score)
        got = (float(auprc[r]), float(auroc[r]), float(ap[r]))
        for a, b, name in zip(got, ref, ("auprc", "auroc", "ap")):
            assert abs(a - b) <= 1e-9, "relation {} {}: {} vs sklearn {}".format(r, name, a, b)


@pytest.mark.parametrize("sizes", [[5000, 7, 4096, 4097, 0, 9000], [70000, 300, 8192, 20000], [33000, 1, 1024], [150000, 12]])
def test_relation_metrics_with_relations_beyond_one_chunk(gpu, sizes):
    """Round 6 (own segment sort): relations longer than the 4,096 scores a workgroup sorts in LDS go through merge rounds (1, 2, 4
    and 6 here: even and odd numbers of ping-pong rounds; lengths on and next to chunk and tile boundaries); all scores of a
    relation tied, ties that straddle chunk boundaries, ties BETWEEN the classes; against scikit-learn per block (<= 1e-9), the
    same bits twice in a row, and the plan-less entry point agrees bit for bit."""
    from gripnet_amd.utils import auprc_auroc_ap, relation_metrics
    gen = torch.Generator().manual_seed(sum(sizes))
    rl = gripnet_amd.utils.get_range_list([torch.zeros(2, s) for s in sizes])
    E = sum(sizes)
    pos = torch.sigmoid(torch.randn(E, generator=gen) + 0.4)
    neg = torch.sigmoid(torch.randn(E, generator=gen) - 0.1)
    a, b = int(rl[0, 0]), int(rl[0, 1])
    pos[a:b] = torch.round(pos[a:b] * 64) / 64                       # 65 distinct values over many chunks, shared with the negatives
    neg[a:b] = torch.round(neg[a:b] * 64) / 64
    a, b = int(rl[1, 0]), int(rl[1, 1])
    pos[a:b] = 0.5                                                    # a relation whose scores are all the same
    neg[a:b] = 0.5
    got = torch.stack(relation_metrics(pos.to(gpu), neg.to(gpu), rl)).cpu()
    again = torch.stack(relation_metrics(pos.to(gpu), neg.to(gpu), rl)).cpu()
    assert torch.equal(got.nan_to_num(-1.0), again.nan_to_num(-1.0))
    out = torch.empty((3, len(sizes)), dtype=torch.float64, device=gpu)
    need = int(_hip.load().gn_link_metrics_workspace_bytes(len(sizes), E))
    ws = torch.empty((need,), dtype=torch.uint8, device=gpu)
    pg, ng = pos.to(gpu), neg.to(gpu)
    _hip._call("gn_link_metrics_f32", pg.data_ptr(), ng.data_ptr(), rl.contiguous().data_ptr(), len(sizes), E, out.data_ptr(), ws.data_ptr(), need,
               _hip.stream_ptr(gpu))
    assert torch.equal(out.cpu().nan_to_num(-1.0), got.nan_to_num(-1.0))
    for r, (s, e) in enumerate(rl.tolist()):
        if e == s:
            assert torch.isnan(got[:, r]).all()
            continue
        ref = auprc_auroc_ap(torch.cat([torch.ones(e - s), torch.zeros(e - s)]), torch.cat([pos[s:e], neg[s:e]]))
        for k, name in enumerate(("auprc", "auroc", "ap")):
            assert abs(float(got[k, r]) - ref[k]) <= 1e-9, "relation {} ({} edges) {}: {} vs sklearn {}".format(r, e - s, name, float(got[k, r]), ref[k])


# ---- BASELINE.json configs 2 and 4 at scale: the NC suite, large supervertices, wide features --------------
@pytest.mark.parametrize("arithmetic", ["fp32", "fast"])
def test_aminer_syn_vs_oracle(gpu, arithmetic):
    """aminer-style model at the `aminer-syn` scale with the reference's own layer widths
    (GripNet-aminer.py:96-98: [128,64,64] / [64,64] / [128,128,32], 8 classes): 50,000 / 20,000 nodes per
    supervertex, so nothing is LDS-resident and the wide-row specialisations of the gather run.  The tall-skinny
    x W products run on three-term bf16 splits by default (fp32-faithful) and on two-term splits with arithmetic "fast"."""
    from gripnet_amd.synth import make_nc
    data = make_nc("aminer-syn")
    torch.manual_seed(1111)
    model = AminerModel(data.n_p_node, data.n_a_node, data.n_a_type)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    nodes = torch.arange(0, data.n_a_node, 3)
    ref = orc.aminer_forward(sd, data.pp_edge_idx, data.pp_edge_weight, data.pa_edge_idx, data.aa_edge_idx,
                             data.aa_edge_weight, nodes)
    model = model.to(gpu)
    gripnet_amd.utils.set_arithmetic(model, arithmetic)
    with torch.no_grad():
        z, pred = model(make_nc("aminer-syn").to(gpu), nodes.to(gpu))
    close(z, ref["z"], TOL)
    close(pred, ref["score"], TOL)


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
def test_freebase_c_syn_vs_oracle(gpu, storage):
    """freebase-c/d-style model (three supervertices, two external layers in `add` mode merged with the
    target embeddings, GripNet-freebase-c.py:102-105,150-165) at the same scale; fp32, and with bf16 storage of the gathered
    tables (BASELINE.json config 5 as written; the reference has no reduced precision: tolerance as for freebase-a/b - one bf16
    rounding of every gathered element, 2^-7 of the largest activation on z, 5e-3 on the class probabilities)."""
    from gripnet_amd.synth import make_nc
    data = make_nc("aminer-syn")
    torch.manual_seed(1111)
    model = FreebaseCModel(data.n_p_node, data.n_q_node, data.n_a_node, data.n_a_type)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    nodes = torch.arange(1, data.n_a_node, 3)
    ref = orc.freebase_c_forward(sd, data.pp_edge_idx, data.pp_edge_weight, data.pa_edge_idx, data.qq_edge_idx,
                                 data.qq_edge_weight, data.qa_edge_idx, sd["aa_embeddings"], data.aa_edge_idx,
                                 data.aa_edge_weight, nodes, data.n_a_node)
    model = model.to(gpu)
    if storage == "bf16":
        assert len(gripnet_amd.utils.set_table_storage(model, "bf16")) == 7
    with torch.no_grad():
        z, pred = model(make_nc("aminer-syn").to(gpu), nodes.to(gpu))
    if storage == "fp32":
        close(z, ref["z"], TOL)
        close(pred, ref["score"], TOL)
    else:
        close(z, ref["z"], 2.0 ** -7 * float(ref["z"].abs().max()))
        close(pred, ref["score"], 5e-3)


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
def test_freebase_a_and_b_syn_vs_oracle(gpu, storage):
    """freebase-a and freebase-b at the freebase-syn scale (the aminer-syn node / edge scale with the
    reference's own widths, GripNet-freebase-a.py:94 [256,128,128], GripNet-freebase-b.py:96-98 [128,64,64] /
    [128,128] / [256,128,32]), fp32 and with bf16 storage of the gathered tables (BASELINE.json config 5; the reference
    has no reduced precision: tolerance of the bf16 variant = one bf16 rounding of every gathered element, 2^-8 of the
    largest activation per layer, checked on the class probabilities at 5e-3)."""
    from gripnet_amd.synth import make_nc
    from gripnet_amd.utils import set_table_storage
    data = make_nc("aminer-syn")
    nodes = torch.arange(2, data.n_a_node, 3)
    torch.manual_seed(1111)
    a = FreebaseAModel(data.n_a_node, data.n_a_type)
    b = FreebaseBModel(data.n_p_node, data.n_a_node, data.n_a_type)
    sda = {k: v.detach().clone() for k, v in a.state_dict().items()}
    sdb = {k: v.detach().clone() for k, v in b.state_dict().items()}
    ref_a = orc.freebase_a_forward(sda, data.aa_edge_idx, data.aa_edge_weight, nodes)
    ref_b = orc.aminer_forward(sdb, data.pp_edge_idx, data.pp_edge_weight, data.pa_edge_idx, data.aa_edge_idx,
                               data.aa_edge_weight, nodes)
    dg = make_nc("aminer-syn").to(gpu)
    a, b = a.to(gpu), b.to(gpu)
    if storage == "bf16":
        assert len(set_table_storage(a, "bf16")) == 2 and len(set_table_storage(b, "bf16")) == 5
    with torch.no_grad():
        za, pa = a(dg, nodes.to(gpu))
        zb, pb = b(dg, nodes.to(gpu))
    if storage == "fp32":
        close(za, ref_a["z"], TOL)
        close(pa, ref_a["score"], TOL)
        close(zb, ref_b["z"], TOL)
        close(pb, ref_b["score"], TOL)
    else:
        close(za, ref_a["z"], 2.0 ** -7 * float(ref_a["z"].abs().max()))
        close(zb, ref_b["z"], 2.0 ** -7 * float(ref_b["z"].abs().max()))
        close(pa, ref_a["score"], 5e-3)
        close(pb, ref_b["score"], 5e-3)


def test_distmult_improved_baseline_caller(gpu):
    """The reference's third caller of the DistMult decoder (baselines/LP_baselines/dmt_pose.py:54,70,84-97: a learned
    embedding of ALL nodes -> decoder): n = 20,000 nodes x 64 features is far beyond the LDS-resident tables, so the
    general kernel (tables in L2) carries it; also through the module's static-list plan, which must refuse this size
    and fall back."""
    gen = torch.Generator().manual_seed(131)
    n, R, F, E = 20000, 37, 64, 300000
    emb = torch.randn(n, F, generator=gen) * 0.3
    et = torch.sort(torch.randint(0, R, (E,), generator=gen)).values
    ei = torch.randint(0, n, (2, E), generator=gen)
    dm = gripnet_amd.multiRelaInnerProductDecoder(F, R)
    ref = orc.distmult(emb, ei, et, dm.weight.detach())
    ref_logits = orc.distmult(emb, ei, et, dm.weight.detach(), sigmoid=False)
    dm = dm.to(gpu)
    z, eig, etg = emb.to(gpu), ei.to(gpu), et.to(gpu)
    with torch.no_grad():
        for _ in range(3):                                   # 2nd and 3rd call: the same tensors again (static list)
            close(dm(z, eig, etg), ref, TOL)
        close(dm(z, eig, etg, sigmoid=False), ref_logits, TOL)
    _hip.raise_if_index_errors(gpu)


def test_rgcn_improved_baseline_caller(gpu):
    """The reference's second caller of myRGCN + the DistMult decoder (baselines/LP_baselines/rgcn_pose.py:53-106:
    embedding -> myRGCN x2 -> decoder on a homogenised graph of ~10^4 nodes): too many nodes for the
    LDS-resident path, so the general relational kernel and the general decoder carry it."""
    gen = torch.Generator().manual_seed(97)
    n, R, B = 6000, 12, 16
    sizes = [int(s) for s in torch.randint(200, 4000, (R,), generator=gen)]
    blocks = [torch.randint(0, n, (2, s), generator=gen) for s in sizes]
    ei = torch.cat([torch.cat([b, b.flip(0)], dim=1) for b in blocks], dim=1)
    rl = gripnet_amd.utils.get_range_list([torch.zeros(2, 2 * s) for s in sizes])
    et = torch.cat([torch.full((2 * s,), r, dtype=torch.long) for r, s in enumerate(sizes)])
    torch.manual_seed(101)
    emb = torch.randn(n, 64)
    c1 = gripnet_amd.myRGCN(64, 32, R, B, after_relu=False)
    c2 = gripnet_amd.myRGCN(32, 32, R, B, after_relu=True)
    dm = gripnet_amd.multiRelaInnerProductDecoder(32, R)
    sd1 = {k: v.detach().clone() for k, v in c1.state_dict().items()}
    sd2 = {k: v.detach().clone() for k, v in c2.state_dict().items()}
    h = torch.relu(orc.rgcn_forward(emb, ei, rl, sd1["basis"], sd1["att"], sd1["root"]))
    zr = orc.rgcn_forward(h, ei, rl, sd2["basis"], sd2["att"], sd2["root"])
    ref = orc.distmult(zr, ei, et, dm.weight.detach())
    c1, c2, dm = c1.to(gpu), c2.to(gpu), dm.to(gpu)
    with torch.no_grad():
        hg = c1(emb.to(gpu), ei.to(gpu), et.to(gpu), rl, _relu=True)
        zg = c2(hg, ei.to(gpu), et.to(gpu), rl)
        score = dm(zg, ei.to(gpu), et.to(gpu))
    close(zg, zr, TOL)
    close(score, ref, TOL)


@pytest.mark.parametrize("n,fin,fout,bases,R", [(20000, 64, 32, 16, 600), (5000, 40, 24, 5, 520), (3000, 128, 16, 20, 7),
                                                 (2500, 20, 8, 64, 3), (70000, 16, 32, 2, 501)])
def test_rgcn_general_path_is_linear_in_the_edges(gpu, n, fin, fout, bases, R):
    """The O(E)-memory relational path (rgcn_basis.hip: basis-space sums per destination on the fp32 matrix instruction, one
    dense product per slab of rows) where the reference's per-relation loop is the only other O(E) formulation
    (layers.py:171-189; the all-nodes caller baselines/LP_baselines/rgcn_pose.py:53-106 has N = 2 x 10^4, R ~ 10^3): graphs
    with >= 500 relations and up to 7 x 10^4 nodes, skewed relation sizes, empty relations, hub destinations, nodes without
    edges, input widths that are no multiple of 16, 2..64 bases, an x that is a column slice (no vector loads); against the
    float64 oracle; workspace independent of R and N; bias, ReLU, concat slot and the un-normalised partial sums of a shard."""
    gen = torch.Generator().manual_seed(n + fin + R)
    torch.manual_seed(n * 3 + R)
    ranks = torch.arange(1, R + 1, dtype=torch.float64)
    sizes = [int(s) for s in torch.floor(300000 * ranks.pow(-0.9) / ranks.pow(-0.9).sum())]
    sizes[R // 2] = 0
    blocks = [torch.randint(0, n - n // 9, (2, s), generator=gen) for s in sizes]          # the top ids have no edges
    blocks[0][1, :4000] = 7                                                               # a hub destination
    ei = torch.cat(blocks, dim=1)
    rl = gripnet_amd.utils.get_range_list(blocks)
    wide = torch.randn(n, fin + 5, generator=gen)
    conv = gripnet_amd.myRGCN(fin, fout, R, bases, False, bias=True).to(gpu)
    conv.bias.data.normal_()
    sd = {k: v.detach().cpu().double() for k, v in conv.state_dict().items()}
    eig = ei.to(gpu)
    for x in (wide[:, :fin].contiguous(), wide[:, 3:3 + fin]):
        xg = x.to(gpu) if x.is_contiguous() else wide.to(gpu)[:, 3:3 + fin]
        ref = torch.relu(orc.rgcn_forward(x.double(), ei, rl, sd["basis"], sd["att"], sd["root"], sd["bias"])).float()
        with torch.no_grad():
            slot = torch.full((n, fin + fout), float("nan"), device=gpu)
            y = conv(xg, eig, None, rl, _relu=True, _out=slot[:, fin:], _side=(xg, slot[:, :fin], 0))
        plan = conv._plan
        assert plan.path(fin, fout, bases) == "general"
        close(y, ref, TOL * max(1.0, float(ref.abs().max())))
        assert torch.equal(slot[:, :fin], xg)
        with torch.no_grad():
            assert torch.equal(conv(xg, eig, None, rl, _relu=True), y)                    # the same bits on every call
    need = int(_hip.load().gn_rgcn_workspace_bytes(plan._h, fin, fout, bases, 0))
    assert need <= (258 << 20) and (R < 500 or need < R * n * fout * 4)
    # several slabs of rows (GN_RGCN_SLAB_MB: the default 256 MB holds these graphs in one): the same bits
    os.environ["GN_RGCN_SLAB_MB"] = "2"
    try:
        assert int(_hip.load().gn_rgcn_workspace_bytes(plan._h, fin, fout, bases, 0)) < need or need < (3 << 20)
        with torch.no_grad():
            assert torch.equal(conv(xg, eig, None, rl, _relu=True), y)
    finally:
        del os.environ["GN_RGCN_SLAB_MB"]
    # a shard's un-normalised partial sums (GN_RGCN_PARTIAL) over two edge ranges add up to deg * (out - root term)
    E = ei.shape[1]
    xg = wide[:, :fin].contiguous().to(gpu)
    parts = []
    for lo, hi in ((0, E // 3), (E // 3, E)):
        p = torch.empty(n, fout, device=gpu)
        _hip.RgcnPlan(eig, rl, n, lo, hi).forward(xg, conv.basis.detach(), conv.att.detach(), None, None, False, p, partial=True)
        parts.append(p)
    whole = torch.empty(n, fout, device=gpu)
    plan.forward(xg, conv.basis.detach(), conv.att.detach(), None, None, False, whole, partial=True)
    scale = max(1.0, float(whole.abs().max()))
    close((parts[0] + parts[1]) / scale, whole / scale, TOL)
    if n <= 5000:                                                                         # the table path, where its table is small
        conv.kernel = "table"
        with torch.no_grad():
            close(conv(xg, eig, None, rl, _relu=True), ref if x.is_contiguous() else
                  torch.relu(orc.rgcn_forward(wide[:, :fin].double(), ei, rl, sd["basis"], sd["att"], sd["root"], sd["bias"])).float(),
                  TOL * max(1.0, float(ref.abs().max())))
    _hip.raise_if_index_errors(gpu)


def test_sharded_forward_on_hip_kernels(gpu):
    """gripnet_amd.sharded with the product kernels: world_size 1 end to end, and the two ranks of a
    world_size-2 job run one after the other on this device with the all-reduce done by hand
    (the RCCL call itself is covered by bench.py --gpus N; its gloo twin by tests/test_sharded_gloo.py)."""
    from gripnet_amd.sharded import ShardedPoseForward
    data = make_pose("small").to(gpu)
    torch.manual_seed(1111)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(gpu)
    with torch.no_grad():
        z_ref, score_ref = model(data)
        z1, s1 = ShardedPoseForward(model, data, 0, 1)()
        close(z1, z_ref)
        close(s1, score_ref)
        ranks = [ShardedPoseForward(model, data, k, 2) for k in range(2)]
        x = ranks[0].kernels.encode_genes()
        parts = []
        for f in ranks:
            p = torch.empty(data.n_d_node, 32, device=gpu)
            f.kernels.partial(x, p)
            parts.append(p)
        total = parts[0] + parts[1]                               # what all_reduce(SUM) leaves on every rank
        scores = []
        for f in ranks:
            out = torch.empty(data.n_d_node, 80, device=gpu)
            f.kernels.finalize(total.clone(), x, out[:, 48:], out[:, :48])
            close(out, z_ref)
            scores.append(f.kernels.score(out))
        assert ranks[0].edge_hi == ranks[1].edge_lo
        close(torch.cat(scores), score_ref)
        # the decoder's input columns scored beside the exchange (what a world_size > 1 job does): same z, same scores
        if os.environ.get("GN_DISABLE_FAST") != "1":
            over = ShardedPoseForward(model, data, 0, 1)
            over.overlap_decoder = True
            z2, s2 = over()
            assert torch.equal(z2, z1)
            close(s2, s1)                                             # (same bits only when the split is a phase boundary of the single launch)
            z3, l3 = over(sigmoid=False)
            close(l3, model(data, sigmoid=False)[1])
        # the same forward as recorded entry-point calls around the exchange (what bench.py runs at N > 1): the same bits,
        # replay after replay, with and without the decoder's input columns beside the all-reduce
        for overlap in (False, True):
            f = ShardedPoseForward(model, data, 0, 1)
            f.overlap_decoder = overlap and os.environ.get("GN_DISABLE_FAST") != "1"
            want_z, want_s = f()
            want_z, want_s = want_z.clone(), want_s.clone()
            run = f.record()
            for _ in range(3):
                zr, sr = run()
                assert torch.equal(zr, want_z) and torch.equal(sr, want_s)
    _hip.raise_if_index_errors(gpu)


@needs_fast_paths
def test_decoder_plan_in_column_ranges(gpu):
    """gn_distmult_plan_forward_cols_f32: the planned decoder in two or three launches over column ranges.  Split
    where the single launch changes phase (48 | 32 at 645 nodes) the scores are the same bits; elsewhere they are the
    same sums in another association.  The first launch may be given a table that holds the first columns only."""
    gen = torch.Generator().manual_seed(5)
    n, f, R = 645, 80, 9
    sizes = [3000, 0, 41, 7000, 64, 1, 129, 900, 2]
    ei = torch.cat([gripnet_amd.utils.to_bidirection(torch.randint(0, n, (2, s), generator=gen)) for s in sizes], dim=1).to(gpu)
    et = torch.cat([torch.full((2 * s,), r, dtype=torch.int64) for r, s in enumerate(sizes)]).to(gpu)
    z = torch.randn(n, f, generator=gen).to(gpu)
    w = torch.randn(R, f, generator=gen).to(gpu)
    plan = _hip.DistMultPlan(ei, et, n, R)
    E = ei.shape[1]
    for sigmoid in (True, False):
        whole = plan.forward(z, w, sigmoid, torch.empty(E, device=gpu))
        out = torch.full((E,), float("nan"), device=gpu)
        plan.forward_cols(z[:, :48].contiguous(), f, 0, 48, w, sigmoid, out)      # a table of the first 48 columns only
        plan.forward_cols(z, f, 48, 80, w, sigmoid, out)
        assert torch.equal(out, whole)
        out3 = torch.full((E,), float("nan"), device=gpu)
        plan.forward_cols(z, f, 0, 16, w, sigmoid, out3)
        plan.forward_cols(z, f, 16, 36, w, sigmoid, out3)
        plan.forward_cols(z, f, 36, 80, w, sigmoid, out3)
        close(out3, whole, 1e-5)
    ref = orc.distmult(z.cpu(), ei.cpu(), et.cpu(), w.cpu(), sigmoid=True)
    close(whole if sigmoid else torch.sigmoid(whole), ref)
    with pytest.raises(ValueError):
        plan.forward_cols(z, f, 2, 80, w, True, out)                              # not a multiple of 4
    # a table that needs several column phases inside each launch (1,200 nodes: 16 columns per phase): partial sums
    # wait in LDS between the phases of a launch and travel through `out` between the launches
    n2 = 1200
    ei2 = torch.cat([gripnet_amd.utils.to_bidirection(torch.randint(0, n2, (2, s), generator=gen)) for s in sizes], dim=1).to(gpu)
    z2 = torch.randn(n2, 64, generator=gen).to(gpu)
    w2 = torch.randn(R, 64, generator=gen).to(gpu)
    plan2 = _hip.DistMultPlan(ei2, et, n2, R)
    whole2 = plan2.forward(z2, w2, True, torch.empty(E, device=gpu))
    ref2 = orc.distmult(z2.cpu(), ei2.cpu(), et.cpu(), w2.cpu(), sigmoid=True)
    close(whole2, ref2)
    for cut in (16, 32, 48):
        out2 = torch.full((E,), float("nan"), device=gpu)
        plan2.forward_cols(z2, 64, 0, cut, w2, True, out2)
        plan2.forward_cols(z2, 64, cut, 64, w2, True, out2)
        close(out2, ref2)
        if cut == 32:
            assert torch.equal(out2, whole2)                                      # the single launch's phases are 32 + 32 columns


# ---- the decoder's plan for static edge lists ---------------------------------------------------
@needs_fast_paths
def test_decoder_plan_matches_planless_scores_bitwise(gpu):
    """Second sighting of the same (edge_index, edge_type) tensors builds a plan (packed, batch-ordered edges);
    its scores are the plan-less kernel's, bit for bit, in the caller's edge order: ragged last batch, batches
    that straddle relations, repeated edges, u == v, sigmoid on and off.  New tensors never get a plan."""
    gen = torch.Generator().manual_seed(11)
    n, f, R = 300, 80, 7
    sizes = [1000, 0, 37, 5000, 64, 1, 129]
    ei = torch.cat([torch.randint(0, n, (2, s), generator=gen) for s in sizes], dim=1)
    ei[1, :50] = ei[0, :50]                                           # u == v
    ei[:, 100:164] = ei[:, 200:201]                                   # one edge 64 times
    et = torch.cat([torch.full((s,), r, dtype=torch.int64) for r, s in enumerate(sizes)])
    z = torch.randn(n, f, generator=gen).to(gpu)
    dec = gripnet_amd.multiRelaInnerProductDecoder(f, R).to(gpu)
    ei_g, et_g = ei.to(gpu), et.to(gpu)
    with torch.no_grad():
        for sig in (True, False):
            first = dec(z, ei_g, et_g, sigmoid=sig)                   # plan-less (first sighting) or planned
            again = dec(z, ei_g, et_g, sigmoid=sig)
            assert torch.equal(first, again)
            close(again, orc.distmult(z.cpu(), ei, et, dec.weight.detach().cpu(), sigmoid=sig))
        assert dec._seen[0].plan not in (None, False)                   # the plan exists and was used
        fresh = dec(z, ei_g.clone(), et_g.clone())                    # same values, new tensors: plan-less path
        assert torch.equal(fresh, dec(z, ei_g, et_g))
        ei_g[0, 0] = (ei_g[0, 0] + 1) % n                             # in-place change: the cached plan is stale
        changed = dec(z, ei_g, et_g)
        ei2 = ei.clone(); ei2[0, 0] = (ei2[0, 0] + 1) % n
        close(changed, orc.distmult(z.cpu(), ei2, et, dec.weight.detach().cpu()))


def test_decoder_plan_rejects_out_of_range_edges(gpu):
    dec = gripnet_amd.multiRelaInnerProductDecoder(16, 3).to(gpu)
    z = torch.randn(10, 16, device=gpu)
    ei = torch.tensor([[0, 1, 2], [3, 4, 10]], device=gpu)           # node 10 does not exist
    et = torch.tensor([0, 1, 2], device=gpu)
    with torch.no_grad():
        dec(z, ei, et)                                                # first sighting: NaN + lazy error flag, as before
        with pytest.raises(IndexError):
            _hip.raise_if_index_errors(gpu)
        with pytest.raises(IndexError):
            dec(z, ei, et)                                            # second sighting validates at plan time


def test_forced_general_path_gets_its_own_workspace(gpu):
    """A kernel forced by flag while the destination-major kernel applies: gn_rgcn_workspace_bytes answers for THAT
    call (the general path's slab of rows and stacked weights - independent of R; the table path's [R, N, out] table), and
    a forward handed a smaller workspace is refused instead of writing past it (round-3 advisor finding)."""
    data = make_pose("small").to(gpu)
    n, fin, fout, R = data.n_d_node, 48, 32, data.n_dd_edge_type
    conv = gripnet_amd.myRGCN(fin, fout, R, 32, False, bias=True).to(gpu)
    x = torch.randn(n, fin, device=gpu)
    with torch.no_grad():
        base = conv(x, data.train_idx, None, data.train_range, _relu=True)
        plan = conv._plan
        assert plan.path(fin, fout, 32) == "pair" and plan.path(fin, fout, 32, path="general") == "general"
        lib = _hip.load()
        need_default = int(lib.gn_rgcn_workspace_bytes(plan._h, fin, fout, 32, plan.mode_flags()))
        need_general = int(lib.gn_rgcn_workspace_bytes(plan._h, fin, fout, 32, plan.mode_flags(path="general")))
        need_table = int(lib.gn_rgcn_workspace_bytes(plan._h, fin, fout, 32, plan.mode_flags(path="table")))
        kp = (33 * fin + 31) // 32 * 32                                # [U_i | x_i] padded to the dense product's K step
        assert need_default == 0 and need_table >= R * n * fout * 4 and kp * (n + fout) * 4 <= need_general <= kp * (n + fout) * 4 + 512
        for forced in ("general", "table"):
            conv.kernel = forced
            close(conv(x, data.train_idx, None, data.train_range, _relu=True), base)
        conv.kernel = "general"
        out = torch.empty(n, fout, device=gpu)
        small = torch.empty(1024, dtype=torch.uint8, device=gpu)
        status = lib.gn_rgcn_forward_f32(plan._h, x.data_ptr(), fin, fin, conv.basis.data_ptr(), conv.att.data_ptr(), 32,
                                         conv.root.data_ptr(), None, fout, 1, plan.mode_flags(path="general"), out.data_ptr(), fout,
                                         None, None, small.data_ptr(), 1024, _hip.stream_ptr(gpu))
        assert status == _hip.GN_ERR_INVALID_ARG and b"workspace too small" in lib.gn_last_error()


# ---- bf16 storage of the gathered table -----------------------------------
def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("fout", [8, 24, 64, 200])
def test_gcn_bf16_table_is_exact_against_the_rounded_table(gpu, fout):
    """gn_cast_bf16 + gn_graph_aggregate_bf16: with x W rounded to bf16 the layer equals the fp32 oracle run on the
    same rounded table to the usual tolerance (fp32 sums), and stays within one bf16 rounding of the fp32 layer."""
    gen = torch.Generator().manual_seed(fout)
    n, fin, e = 700, 40, 9000
    ei = torch.randint(0, n, (2, e), generator=gen)
    ei[1, :400] = 5                                                   # one hub row, isolated nodes elsewhere
    w = torch.rand(e, generator=gen) + 0.5
    x = torch.randn(n, fin, generator=gen)
    torch.manual_seed(1000 + fout)
    conv = gripnet_amd.myGCN(fin, fout, cached=True).to(gpu)
    conv.bias.data.normal_()
    with torch.no_grad():
        # the table that gets rounded is the GPU's own fp32 product: a host product differs from it in the last bit
        # here and there, and where that flips a bf16 rounding the two tables differ by 2^-8 of the entry
        xw_gpu = torch.empty(n, fout, device=gpu)
        _hip.gemm(x.to(gpu), conv.weight.detach(), xw_gpu)
        y32 = conv(x.to(gpu), ei.to(gpu), w.to(gpu), _relu=True)
        gripnet_amd.utils.set_table_storage(conv, "bf16")
        y16 = conv(x.to(gpu), ei.to(gpu), w.to(gpu), _relu=True)
        gripnet_amd.utils.set_table_storage(conv, "fp32")
        assert torch.equal(y32, conv(x.to(gpu), ei.to(gpu), w.to(gpu), _relu=True))
    wt, b = conv.weight.detach().cpu(), conv.bias.detach().cpu()
    ei2, norm = orc.gcn_norm(ei, n, w)
    xw16 = _bf16_round(xw_gpu.cpu())
    ref16 = torch.relu(torch.zeros(n, fout).index_add_(0, ei2[1], norm.view(-1, 1) * xw16.index_select(0, ei2[0])) + b)
    close(y16, ref16)
    scale = float(y32.abs().max())
    assert float((y16 - y32).abs().max()) <= 2.0 ** -8 * max(scale, 1.0)          # stated tolerance of the bf16 variant


@needs_fast_paths
@pytest.mark.parametrize("n,fin,fout", [(4096, 64, 32), (9000, 128, 128), (5003, 256, 128), (20000, 32, 24)])
def test_product_stores_the_bf16_table_itself(gpu, n, fin, fout):
    """GN_GEMM_OUT_BF16 (round 6): the tall-skinny product rounds its output to bf16 where it stores it - bit for bit the
    table gn_cast_bf16 makes of the fp32 product (same arithmetic, nearest-even, once), without the extra pass; with bias and
    ReLU; shapes outside the tall-skinny kernel are refused with GN_ERR_UNSUPPORTED (the caller then casts)."""
    gen = torch.Generator().manual_seed(n + fout)
    x = torch.randn(n, fin, generator=gen).to(gpu)
    w = (torch.randn(fin, fout, generator=gen) / fin ** 0.5).to(gpu)
    b = torch.randn(fout, generator=gen).to(gpu)
    for bias, relu in ((None, False), (b, True)):
        c32 = torch.empty(n, fout, device=gpu)
        _hip.gemm(x, w, c32, bias=bias, relu=relu)
        want = c32.to(torch.bfloat16)                                  # nearest even, as gn_cast_bf16 (whose pass needs fout % 8 == 0)
        if fout % 8 == 0:
            cast = torch.empty(n, fout, dtype=torch.bfloat16, device=gpu)
            _hip._call("gn_cast_bf16", c32.data_ptr(), _hip.ld(c32), cast.data_ptr(), _hip.ld(cast), n, fout, _hip.stream_ptr(gpu))
            assert torch.equal(cast.view(torch.int16), want.view(torch.int16))
        got = torch.full((n, fout), float("nan"), dtype=torch.bfloat16, device=gpu)
        _hip.gemm(x, w, got, bias=bias, relu=relu, out_bf16=True)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    small = torch.empty(100, fout, dtype=torch.bfloat16, device=gpu)
    with pytest.raises(_hip.GripNetHipError) as err:
        _hip.gemm(x[:100], w, small, out_bf16=True)
    assert err.value.status == _hip.GN_ERR_UNSUPPORTED


@needs_fast_paths
@pytest.mark.parametrize("fout", [8, 32, 128, 256])
def test_bf16_group_gather_on_many_short_rows(gpu, fout):
    """k_aggregate_group_bf16 (round 6: lane groups own rows, as the fp32 layers of the node-classification graphs): the layer
    with bf16 storage on a graph of many short rows equals the fp32 oracle run on the rounded table; hubs, isolated rows
    and weights included; and it agrees with the wave-per-row kernel (GN_DISABLE_FAST=1) to fp32 summation order."""
    gen = torch.Generator().manual_seed(fout)
    n, fin, e = 6000, 64, 40000
    ei = torch.randint(0, n, (2, e), generator=gen)
    ei[1, :300] = 17                                                  # one hub row
    ei = ei[:, ei[1] % 11 != 3]                                       # isolated destination rows
    w = torch.rand(ei.shape[1], generator=gen) + 0.5
    x = torch.randn(n, fin, generator=gen)
    torch.manual_seed(2000 + fout)
    conv = gripnet_amd.myGCN(fin, fout, cached=True).to(gpu)
    conv.bias.data.normal_()
    gripnet_amd.utils.set_table_storage(conv, "bf16")
    with torch.no_grad():
        y16 = conv(x.to(gpu), ei.to(gpu), w.to(gpu), _relu=True)
        table = torch.empty(n, fout, dtype=torch.bfloat16, device=gpu)
        _hip.gemm(x.to(gpu), conv.weight.detach(), table, out_bf16=True)
    wt, b = conv.weight.detach().cpu(), conv.bias.detach().cpu()
    ei2, norm = orc.gcn_norm(ei, n, w)
    xw16 = table.float().cpu()
    ref16 = torch.relu(torch.zeros(n, fout).index_add_(0, ei2[1], norm.view(-1, 1) * xw16.index_select(0, ei2[0])) + b)
    close(y16, ref16)
    ref32 = torch.relu(torch.zeros(n, fout).index_add_(0, ei2[1], norm.view(-1, 1) * (x @ wt).index_select(0, ei2[0])) + b)
    assert float((y16.cpu() - ref32).abs().max()) <= 2.0 ** -7 * max(float(ref32.abs().max()), 1.0)


def test_nc_pipeline_with_bf16_tables(gpu, golden):
    """The aminer pipeline of the golden fixtures with bf16 tables in every GCN-style layer: class probabilities within
    5e-3 of the reference's fp32 result (three stacked layers of rounded tables), argmax unchanged on >= 99 % of nodes."""
    g = golden("aminer_tiny")
    m = load_into(AminerModel(g.meta["n_p"], g.meta["n_a"], g.meta["n_class"], pp_nhids=g.meta["pp_nhids"],
                              pa_out=g.meta["pa_out"], aa_hidden=g.meta["aa_nhids"][1:]), g.state("", strip=False), gpu)
    data = Data(**{k: g.t(k, gpu) for k in ("pp_edge_idx", "pa_edge_idx", "aa_edge_idx", "pp_edge_weight", "aa_edge_weight")})
    touched = gripnet_amd.utils.set_table_storage(m, "bf16")
    assert len(touched) >= 5
    with torch.no_grad():
        z, score = m(data, g.t("node_list", gpu))
    ref = g.t("out.score")
    assert float((score.cpu() - ref).abs().max()) <= 5e-3
    assert float((score.cpu().argmax(1) == ref.argmax(1)).float().mean()) >= 0.99


# ---- the default arithmetic is fp32-faithful (round 3: the mode is a flag of the C ABI, never the environment) ----------
@pytest.mark.parametrize("n,fin,bases", [(560, 48, 32), (645, 48, 32), (200, 32, 5), (645, 64, 8)])
def test_default_relational_arithmetic_is_as_exact_as_fp32(gpu, n, fin, bases):
    """Against the float64 oracle the default relational path (three-term bf16 splits on the matrix cores) is held to the
    error of the exact fp32 paths (the LDS-accumulator and the general kernel on v_mfma_f32_16x16x4_f32) and of the
    reference's own fp32 op sequence; the two-term "fast" mode is allowed to be worse and is reported, not held."""
    gen = torch.Generator().manual_seed(n * 7 + fin)
    torch.manual_seed(n + fin)
    sizes = [0, 9000, 3, 0, 700, 1, 2500, 0]
    blocks = [torch.randint(0, max(1, n - n // 7), (2, s), generator=gen) for s in sizes]
    rei = torch.cat(blocks, dim=1)
    rl = gripnet_amd.utils.get_range_list(blocks)
    x = torch.randn(n, fin, generator=gen)
    rg = gripnet_amd.myRGCN(fin, 32, len(sizes), bases, False, bias=True).to(gpu)
    sd = {k: v.detach().cpu() for k, v in rg.state_dict().items()}
    ref64 = orc.rgcn_forward(x.double(), rei, rl, sd["basis"].double(), sd["att"].double(), sd["root"].double(), sd["bias"].double())
    ref32 = orc.rgcn_forward(x, rei, rl, sd["basis"], sd["att"], sd["root"], sd["bias"])
    err = {}
    for name, kernel, arith in (("default", "auto", "fp32"), ("lds-exact", "lds", "fp32"), ("general-exact", "general", "fp32"), ("fast", "auto", "fast")):
        rg.kernel, rg.arithmetic = kernel, arith
        y = rg(x.to(gpu), rei.to(gpu), None, rl)
        err[name] = (y.cpu().double() - ref64).abs().max().item()
    err["reference-fp32"] = (ref32.double() - ref64).abs().max().item()
    exact = max(err["lds-exact"], err["general-exact"], err["reference-fp32"])
    assert err["default"] <= 1.5 * exact + 1e-7, err
    assert err["fast"] <= 2e-5, err


def test_default_gemm_arithmetic_is_as_exact_as_fp32(gpu, monkeypatch):
    """gn_gemm_f32 on a tall-skinny product: the default (three-term splits) against the exact fp32 matrix instruction
    (the general kernel, GN_DISABLE_FAST=1) and torch's fp32 matmul, all measured against float64; "fast" is two-term."""
    gen = torch.Generator().manual_seed(3)
    a = torch.randn(5000, 128, generator=gen)
    b = torch.randn(128, 64, generator=gen) * 0.2
    ref64 = a.double() @ b.double()
    ag, bg = a.to(gpu), b.to(gpu)
    out = torch.empty(5000, 64, device=gpu)
    err = {}
    err["default"] = (_hip.gemm(ag, bg, out).cpu().double() - ref64).abs().max().item()
    err["fast"] = (_hip.gemm(ag, bg, out, fast=True).cpu().double() - ref64).abs().max().item()
    monkeypatch.setenv("GN_DISABLE_FAST", "1")
    err["exact"] = (_hip.gemm(ag, bg, out).cpu().double() - ref64).abs().max().item()
    err["torch-fp32"] = ((a @ b).double() - ref64).abs().max().item()
    assert err["default"] <= 1.5 * max(err["exact"], err["torch-fp32"]) + 1e-7, err
    assert err["fast"] > err["default"] and err["fast"] <= 1e-3, err


@pytest.mark.parametrize("m,k,n", [(2048, 32, 8), (5003, 64, 64), (70001, 96, 48), (4100, 160, 128), (9000, 256, 72),
                                   (3000, 288, 8), (6000, 512, 128), (2500, 800, 20)])
@pytest.mark.parametrize("fast", [False, True])
def test_tall_skinny_gemm_shapes(gpu, m, k, n, fast):
    """gn_gemm_f32 on the tall-skinny path (B through LDS, K deeper than 256 in slabs, several rounds of row tiles, ragged
    last tile and column block) against float64, with bias and ReLU in the epilogue."""
    gen = torch.Generator().manual_seed(m + k + n)
    a = torch.randn(m, k, generator=gen)
    b = torch.randn(k, n, generator=gen) * 0.1
    bias = torch.randn(n, generator=gen)
    ref64 = torch.relu(a.double() @ b.double() + bias.double())
    out = torch.full((m, n), float("nan"), device=gpu)
    _hip.gemm(a.to(gpu), b.to(gpu), out, bias=bias.to(gpu), relu=True, fast=fast)
    err = (out.cpu().double() - ref64).abs().max().item()
    ref32 = (torch.relu(a @ b + bias).double() - ref64).abs().max().item()
    # (fp32 accumulators either way: the order of the K sum differs from the library's, not the precision of the products)
    assert err <= (2e-3 if fast else 3 * ref32 + 1e-6), (err, ref32)


@pytest.mark.parametrize("m,k,n,at,bt", [(32, 964, 1536, True, False), (964, 1536, 32, False, True), (19081, 16, 64, False, True),
                                         (645, 32, 48, False, True), (17, 300, 21, True, True), (16, 256, 70, False, False), (64, 6, 1536, True, False), (40, 5, 100, True, True),
                                         (700, 1000, 9, False, False), (5, 7, 3, False, True)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_gemm_with_operands_given_transposed(gpu, m, k, n, at, bt, accumulate):
    """gn_gemm_f32 with GN_GEMM_A_TRANSPOSED / GN_GEMM_B_TRANSPOSED / GN_GEMM_ACCUMULATE: the backward passes' dx = g W^T,
    dbasis = att^T dW and datt = dW basis^T with the operands as they are stored - the deep and narrow kernel (K over the
    waves of a workgroup) and the general kernels, against float64; the same bits on every launch."""
    gen = torch.Generator().manual_seed(m * 7 + k * 3 + n)
    a = torch.randn(m, k, generator=gen)
    b = torch.randn(k, n, generator=gen) * 0.1
    c0 = torch.randn(m, n, generator=gen)
    ref64 = a.double() @ b.double() + (c0.double() if accumulate else 0.0)
    ag = (a.t().contiguous() if at else a).to(gpu)
    bg = (b.t().contiguous() if bt else b).to(gpu)
    outs = []
    for _ in range(2):
        out = c0.clone().to(gpu) if accumulate else torch.full((m, n), float("nan"), device=gpu)
        _hip.gemm(ag, bg, out, a_transposed=at, b_transposed=bt, accumulate=accumulate)
        outs.append(out)
    err = (outs[0].cpu().double() - ref64).abs().max().item()
    ref32 = ((a @ b + (c0 if accumulate else 0.0)).double() - ref64).abs().max().item()
    assert err <= 3 * ref32 + 1e-6, (err, ref32)
    assert torch.equal(outs[0], outs[1])


def test_independent_products_in_one_launch(gpu):
    """gn_dense_batch_begin / _end: the deep-and-narrow products and the one-launch x^T g products called inside the bracket
    leave as one grid (four at most per launch) and give the bits of their own launches; a product of another kind inside the
    bracket launches at once; a batch of one is the product's own kernel; a second x^T g on a queued workspace is not queued;
    a bracket cannot be nested."""
    gen = torch.Generator().manual_seed(11)
    R, B, K = 964, 32, 48 * 32
    att = torch.randn(R, B, generator=gen).to(gpu)
    dw = torch.randn(R, K, generator=gen).to(gpu)
    basis = torch.randn(B, K, generator=gen).to(gpu)
    x = torch.randn(645, 48, generator=gen).to(gpu)
    g = torch.randn(645, 32, generator=gen).to(gpu)
    wide = torch.randn(K, 40, generator=gen).to(gpu)          # 964 x 40 output over K = 1536: the tall-skinny kernel
    dw2, dw3 = dw * 2, dw * 3

    def products(join=False):
        outs = [torch.full((B, K), float("nan"), device=gpu), torch.full((R, B), float("nan"), device=gpu), None,
                torch.full((R, 40), float("nan"), device=gpu), torch.full((B, K), float("nan"), device=gpu),
                torch.full((R, B), float("nan"), device=gpu), None]
        _hip.gemm(att, dw, outs[0], a_transposed=True, join_batch=join)
        _hip.gemm(dw, basis, outs[1], b_transposed=True, join_batch=join)
        outs[2] = _hip.xtg(x, g, join_batch=join)
        _hip.gemm(dw, wide, outs[3], join_batch=join)
        _hip.gemm(att, dw2, outs[4], a_transposed=True, join_batch=join)
        _hip.gemm(dw3, basis, outs[5], b_transposed=True)       # without the flag: at once
        outs[6] = _hip.xtg(x, g, join_batch=join)             # same shape, same workspace as outs[2]: at once
        return outs

    alone = products()
    with _hip.dense_batch(gpu):
        together = products(True)
    for a, b in zip(alone, together):
        assert torch.equal(a, b)
    assert torch.equal(alone[2], alone[6])
    close(alone[0], att.t().double().cpu() @ dw.double().cpu(), atol=2e-3)
    with _hip.dense_batch(gpu):                               # a batch of one
        single = torch.full((R, B), float("nan"), device=gpu)
        _hip.gemm(dw, basis, single, b_transposed=True, join_batch=True)
    assert torch.equal(single, alone[1])
    with pytest.raises(ValueError):
        _hip.check(_hip.load().gn_dense_batch_end(None))      # no open batch
    with _hip.dense_batch(gpu):
        with pytest.raises(ValueError):
            _hip.check(_hip.load().gn_dense_batch_begin())    # not nested


@pytest.mark.parametrize("k,n", [(288, 8), (64, 3), (128, 16), (32, 1), (96, 20), (30, 4)])
@pytest.mark.parametrize("softmax", [True, False])
def test_class_scores_vs_torch(gpu, k, n, softmax):
    """gn_class_scores_f32 = softmax?(z[node_list] @ W) (decoder.py:42-43): the one-pass kernel (<= 16 classes) and the
    GEMM + softmax route behind the same entry point, with repeated nodes in the list."""
    gen = torch.Generator().manual_seed(k * 31 + n)
    z = torch.randn(5000, k, generator=gen)
    w = torch.randn(k, n, generator=gen) * 0.2
    nodes = torch.randint(0, 5000, (3001,), generator=gen)
    ref = z[nodes].double() @ w.double()
    if softmax:
        ref = torch.softmax(ref, dim=1)
    out = torch.full((nodes.shape[0], n), float("nan"), device=gpu)
    _hip.class_scores(z.to(gpu), w.to(gpu), nodes.to(gpu), out, softmax)
    assert (out.cpu().double() - ref).abs().max().item() <= 2e-5


def test_rgcn_non_finite_input_rows_stay_local(gpu):
    """A non-finite x[s] reaches the destinations s has an edge to (and s itself through the root term) and no others, as in
    the reference's sum over edges: the destination-major kernel contracts over ALL sources of a chunk, so pairs without
    edges must not read their x row (0 . inf = NaN)."""
    gen = torch.Generator().manual_seed(5)
    n, fin, R = 300, 48, 6
    blocks = [torch.randint(0, n, (2, 400), generator=gen) for _ in range(R)]
    rei = torch.cat(blocks, dim=1)
    rl = gripnet_amd.utils.get_range_list(blocks)
    x = torch.randn(n, fin, generator=gen)
    bad = 17
    x[bad, 3] = float("inf")
    rg = gripnet_amd.myRGCN(fin, 32, R, 8, False).to(gpu)
    with torch.no_grad():
        y = rg(x.to(gpu), rei.to(gpu), None, rl).cpu()
    assert rg._plan.path(fin, 32, 8) == "pair"
    touched = set(rei[1][rei[0] == bad].tolist()) | {bad}
    finite = torch.isfinite(y).all(dim=1)
    assert all(bool(finite[i]) for i in range(n) if i not in touched)
    assert not bool(finite[bad])


@pytest.mark.parametrize("fin,fout", [(16, 32), (16, 16), (32, 32), (64, 32), (32, 16), (24, 32), (64, 48)])
def test_gcn_every_small_width_pair_has_a_path(gpu, fin, fout):
    """The widths the fused gather + transform kernels cover and the ones next to them (16 -> 32 was reported fusable without
    a specialisation behind it; found by tools/fuzz_gcn.py): gn_transform_fusable must agree with what the launch supports."""
    gen = torch.Generator().manual_seed(fin * 100 + fout)
    n = 900
    ei = torch.randint(0, n, (2, 7000), generator=gen)
    x = torch.randn(n, fin, generator=gen)
    conv = gripnet_amd.myGCN(fin, fout).to(gpu)
    conv.bias.data.normal_()
    with torch.no_grad():
        y = conv(x.to(gpu), ei.to(gpu), None, _relu=True)
    close(y, torch.relu(orc.gcn_forward(x, conv.weight.detach().cpu(), conv.bias.detach().cpu(), ei, None)))
