"""Pin the CPU oracle: every oracle function vs. golden vectors produced by the reference
itself (tests/golden/make_golden.py) and vs. independent dense closed forms.

Tolerances: index outputs bit-exact; floating point <= 1e-6 abs.
"""
import numpy as np
import pytest
import torch

from oracle import gripnet_oracle as orc

ATOL = 1e-6


def close(a, b, atol=ATOL):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a.double() - b.double()).abs().max().item() if a.numel() else 0.0
    assert err <= atol, "max abs err {:.3e} > {:.1e}".format(err, atol)


def test_norm_known_answer(golden):
    g = golden("norm_known")
    ei2, norm = orc.gcn_norm(g.t("edge_index"), 3, None, improved=True)
    assert torch.equal(ei2, torch.tensor([[0, 1, 0, 1, 2], [1, 0, 0, 1, 2]]))
    assert torch.equal(ei2, g.t("out.edge_index"))
    close(norm, torch.tensor([1 / 3, 1 / 3, 2 / 3, 2 / 3, 1.0]), 1e-6)
    close(norm, g.t("out.norm"), 1e-7)


def test_norm_cases(golden):
    g = golden("norm_cases")
    for i, spec in enumerate(g.meta["cases"]):
        p = "c{}.".format(i)
        w = g.t(p + "edge_weight") if spec["weighted"] else None
        ei2, norm = orc.gcn_norm(g.t(p + "edge_index"), spec["num_nodes"], w, improved=spec["improved"])
        assert torch.equal(ei2, g.t(p + "out.edge_index")), "case {}: edge_index' differs".format(i)
        close(norm, g.t(p + "out.norm"), 1e-7)


def test_gcn_forward(golden):
    g = golden("gcn_forward")
    sd = {"weight": g.t("sd.weight"), "bias": g.t("sd.bias")}
    ei, w = g.t("edge_index"), g.t("edge_weight")
    cache = orc.gcn_norm(ei, g.meta["n"], w)
    close(orc.gcn_forward(g.t("x0"), sd["weight"], sd["bias"], ei, w), g.t("out.y0"))
    close(orc.gcn_forward(g.t("x1"), sd["weight"], sd["bias"], ei, w, cached=cache), g.t("out.y1"))
    nb = g.state("nb.")
    close(orc.gcn_forward(g.t("x0"), nb["weight"], None, ei, None), g.t("out.y_nobias_unweighted"))
    # independent dense closed form
    close(orc.dense_gcn(g.t("x0").double(), sd["weight"].double(), sd["bias"].double(), ei, w.double()),
          g.t("out.y0"), 2e-6)


def test_inter_cases(golden):
    g = golden("inter_cases")
    x, ei, w = g.t("x"), g.t("edge_index"), g.t("edge_weight")
    for v in g.meta["variants"]:
        sd = g.state(v["tag"] + ".")
        y = orc.inter_forward(sd, "", x, ei, w if v["weighted"] else None, if_relu=v["if_relu"],
                              mod=v["mod"], n_target=g.meta["n_target"])
        close(y, g.t(v["tag"] + ".out"))
    # closed form: isolated targets -> relu(b); no padding, no index shift
    sd = g.state("cat.")
    y = torch.relu(orc.inter_forward_closed(sd, "", x, ei, g.meta["n_target"]))
    ref = g.t("cat.out")
    close(y, ref[:, :16], 2e-6)
    close(ref[-1, :16], torch.relu(sd["conv.bias"]), 1e-7)
    close(ref[:, 16:], sd["target_feat"].abs(), 0)
    sd = g.state("cat_w_norelu.")
    close(orc.inter_forward_closed(sd, "", x, ei, g.meta["n_target"], w), g.t("cat_w_norelu.out")[:, :16], 2e-6)


def test_rgcn_cases(golden):
    g = golden("rgcn_cases")
    x, ei, rl = g.t("x"), g.t("edge_index"), g.t("range_list")
    for v in g.meta["variants"]:
        sd = g.state(v["tag"] + ".")
        y = orc.rgcn_forward(x, ei, rl, sd["basis"], sd["att"], sd["root"], sd.get("bias"))
        close(y, g.t(v["tag"] + ".out"))
        yd = orc.dense_rgcn(x.double(), ei, rl, sd["basis"].double(), sd["att"].double(), sd["root"].double(),
                            None if "bias" not in sd else sd["bias"].double())
        close(yd, g.t(v["tag"] + ".out"), 2e-6)
    # zero-in-degree destinations get the root term only
    sd = g.state("plain.")
    n = g.meta["n"]
    close(g.t("plain.out")[n - 3:], (x @ sd["root"])[n - 3:], 1e-6)


def test_homo_cases(golden):
    g = golden("homo_cases")
    x, ei, w = g.t("x"), g.t("edge_index"), g.t("edge_weight")
    sd = g.state("gcn2.")
    close(orc.homo_forward(sd, "", x, ei, w, if_catout=True), g.t("gcn2.out_cat"))
    close(orc.homo_forward(sd, "", x, ei, w, if_catout=False), g.t("gcn2.out_nocat"))
    sd = g.state("start1.")
    close(orc.homo_forward(sd, "", None, ei, None, if_catout=True), g.t("start1.out_cat"))
    sd = g.state("rgcn2.")
    close(orc.homo_forward(sd, "", x, g.t("rel.edge_index"), range_list=g.t("rel.range_list"), if_catout=True),
          g.t("rgcn2.out_cat"))


def test_decoder_cases(golden):
    g = golden("decoder_cases")
    z, ei, et = g.t("z"), g.t("edge_index"), g.t("edge_type")
    D = g.t("sd.dmt.weight")
    close(orc.distmult(z, ei, et, D), g.t("dmt.out_sigmoid"))
    close(orc.distmult(z, ei, et, D, sigmoid=False), g.t("dmt.out_logits"), 2e-6)
    W = g.t("sd.mcip.weight")
    close(orc.multiclass(z, g.t("node_list"), W), g.t("mcip.out_softmax"))
    close(orc.multiclass(z, g.t("node_list"), W, softmax=False), g.t("mcip.out_logits"), 2e-6)


@pytest.mark.parametrize("scale", ["tiny", "small"])
def test_pose_pipeline(golden, scale):
    g = golden("pose_" + scale)
    sd = g.state("", strip=False)
    cache = {}
    for _ in range(2):   # second pass runs from the norm cache, like epochs 2.. of the reference
        out = orc.pose_forward(sd, g.t("gg_edge_index"), g.t("edge_weight"), g.t("gd_edge_index"),
                               g.t("train_idx"), g.t("train_et"), g.t("train_range"), gcn_cache=cache)
        close(out["z_gg"], g.t("out.z_gg"), 2e-6)
        close(out["z_gd"], g.t("out.z_gd"), 2e-6)
        close(out["z_dd"], g.t("out.z_dd"), 2e-6)
        close(out["score"], g.t("out.score"), 2e-6)
    logits = orc.pose_forward(sd, g.t("gg_edge_index"), g.t("edge_weight"), g.t("gd_edge_index"),
                              g.t("train_idx"), g.t("train_et"), g.t("train_range"), sigmoid=False)["score"]
    close(logits, g.t("out.logits"), 1e-5)


def test_nc_pipelines(golden):
    g = golden("aminer_tiny")
    sd = g.state("", strip=False)
    out = orc.aminer_forward(sd, g.t("pp_edge_idx"), g.t("pp_edge_weight"), g.t("pa_edge_idx"),
                             g.t("aa_edge_idx"), g.t("aa_edge_weight"), g.t("node_list"))
    close(out["z"], g.t("out.z"), 2e-6)
    close(out["score"], g.t("out.score"), 2e-6)
    g = golden("freebase_c_tiny")
    sd = g.state("", strip=False)
    out = orc.freebase_c_forward(sd, g.t("pp_edge_idx"), g.t("pp_edge_weight"), g.t("pa_edge_idx"),
                                 g.t("qq_edge_idx"), g.t("qq_edge_weight"), g.t("qa_edge_idx"),
                                 g.t("aa_embeddings"), g.t("aa_edge_idx"), g.t("aa_edge_weight"),
                                 g.t("node_list"), g.meta["n_a"])
    close(out["z"], g.t("out.z"), 2e-6)
    close(out["score"], g.t("out.score"), 2e-6)


def test_freebase_a_and_b_pipelines(golden):
    """BASELINE.json config 5: the one-supervertex freebase-a caller (no concat) and freebase-b (the aminer call
    sequence with equal-width pa_out halves), GripNet-freebase-a.py:94-122, GripNet-freebase-b.py:96-135."""
    g = golden("freebase_a_tiny")
    sd = g.state("", strip=False)
    out = orc.freebase_a_forward(sd, g.t("aa_edge_idx"), g.t("aa_edge_weight"), g.t("node_list"))
    close(out["z"], g.t("out.z"), 2e-6)
    close(out["score"], g.t("out.score"), 2e-6)
    logits = orc.freebase_a_forward(sd, g.t("aa_edge_idx"), g.t("aa_edge_weight"), g.t("node_list"), softmax=False)["score"]
    close(logits, g.t("out.logits"), 2e-6)
    g = golden("freebase_b_tiny")
    sd = g.state("", strip=False)
    out = orc.aminer_forward(sd, g.t("pp_edge_idx"), g.t("pp_edge_weight"), g.t("pa_edge_idx"),
                             g.t("aa_edge_idx"), g.t("aa_edge_weight"), g.t("node_list"))
    close(out["z"], g.t("out.z"), 2e-6)
    close(out["score"], g.t("out.score"), 2e-6)

