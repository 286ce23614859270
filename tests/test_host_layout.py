"""Host-side layout helpers and synthetic generators (no GPU)."""
import os

import numpy as np
import pytest
import torch

from gripnet_amd import utils
from gripnet_amd.synth import make_pose, make_nc, pose_edges_aggregated

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layout_helpers_match_reference(golden):
    g = golden("layout_helpers")
    raw = [g.t("raw{}".format(i)) for i in range(g.meta["n_raw"])]
    np.random.seed(g.meta["np_seed"])          # the reference draws from numpy's global stream
    outs = utils.process_edge_multirelational(raw, p=g.meta["p"])
    for k, v in zip(("train_idx", "train_et", "train_range", "test_idx", "test_et", "test_range"), outs):
        assert torch.equal(v, g.t("out." + k)), k
    assert torch.equal(utils.to_bidirection(raw[0]), g.t("out.bidir"))


def test_range_list_and_bidirection():
    blocks = [torch.zeros(2, 3, dtype=torch.long), torch.zeros(2, 0, dtype=torch.long), torch.zeros(2, 5, dtype=torch.long)]
    rl = utils.get_range_list(blocks)
    assert rl.tolist() == [[0, 3], [3, 3], [3, 8]]
    e = torch.tensor([[0, 1, 2], [3, 4, 5]])
    b, t = utils.to_bidirection(e, torch.tensor([7, 8, 9]))
    assert b.tolist() == [[0, 1, 2, 3, 4, 5], [3, 4, 5, 0, 1, 2]] and t.tolist() == [7, 8, 9, 7, 8, 9]
    assert utils.remove_bidirection(b).tolist() == [[3, 4, 5], [0, 1, 2]]


def test_negative_sampling_avoids_positives():
    rng = np.random.RandomState(3)
    pos = torch.stack([torch.arange(0, 40) % 9, (torch.arange(0, 40) * 7 + 1) % 9])
    neg = utils.negative_sampling(pos, 9, rng)
    assert neg.shape == pos.shape and neg.dtype == torch.int64
    assert not np.isin((neg[0] * 9 + neg[1]).numpy(), (pos[0] * 9 + pos[1]).numpy()).any()
    tneg = utils.typed_negative_sampling(pos, 9, [(0, 10), (10, 40)], rng)
    assert tneg.shape == pos.shape


def test_shard_edge_ranges():
    for e, g in [(10, 3), (7, 8), (0, 2), (1996020, 8)]:
        r = utils.shard_edge_ranges(e, g)
        assert r[0][0] == 0 and r[-1][1] == e and len(r) == g
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
        sizes = [hi - lo for lo, hi in r]
        assert max(sizes) - min(sizes) <= 1


def test_pose_generator_layout():
    d = make_pose("tiny")
    assert d.n_g_node == 50 and d.n_d_node == 12 and d.n_dd_edge_type == 5
    assert d.gg_edge_index.dtype == torch.int64 and d.train_idx.dtype == torch.int64
    rl = d.train_range
    assert rl[0, 0] == 0 and rl[-1, 1] == d.train_idx.shape[1]
    for r in range(rl.shape[0]):                      # per relation: cat(fwd, reversed fwd), typed r
        s, e = int(rl[r, 0]), int(rl[r, 1])
        half = (e - s) // 2
        assert torch.equal(d.train_idx[:, s:s + half], d.train_idx[:, s + half:e].flip(0))
        assert (d.train_et[s:e] == r).all()
    assert pose_edges_aggregated(d) == 2 * (d.gg_edge_index.shape[1] + 50) + 40 + d.train_idx.shape[1]
    d2 = make_pose("tiny")
    assert torch.equal(d.train_idx, d2.train_idx)    # seeded
    moved = d.to("cpu")
    assert moved is d


def test_nc_generator_fields():
    d = make_nc("tiny")
    for k in ("pp_edge_idx", "qq_edge_idx", "aa_edge_idx", "pa_edge_idx", "qa_edge_idx"):
        assert getattr(d, k).dtype == torch.int64 and getattr(d, k).shape[0] == 2
    assert int(d.pa_edge_idx[0].max()) < d.n_p_node and int(d.pa_edge_idx[1].max()) < d.n_a_node


# ---- the drivers' own import lines (boundary) --------------------------------------------------
def _driver_imports():
    import json, os
    with open(os.path.join(os.path.dirname(__file__), "golden", "driver_imports.json")) as f:
        return json.load(f)


def test_every_name_the_callers_import_exists():
    """`from gripnet.X import a, b, c` -> `from gripnet_amd.X import a, b, c` for every reference caller: the name lists
    were read out of the callers by tests/golden/make_golden.py (names only)."""
    import importlib
    table = _driver_imports()
    library = table.pop("__library_top_level__")
    assert len(table) >= 8
    for script, modules in table.items():
        for module, names in modules.items():
            mod = importlib.import_module(module.replace("gripnet", "gripnet_amd", 1))
            missing = [n for n in names if not hasattr(mod, n)]
            assert not missing, "{}: {} lacks {}".format(script, mod.__name__, missing)
    for module, names in library.items():        # `from gripnet.utils import *` keeps working too
        mod = importlib.import_module(module.replace("gripnet", "gripnet_amd", 1))
        missing = [n for n in names if not hasattr(mod, n)]
        assert not missing, "{} lacks {}".format(mod.__name__, missing)


def test_swapped_import_block_of_the_pose_driver_executes():
    """GripNet-pose.py:1-12 with `gripnet` -> `gripnet_amd`, rebuilt from the committed name lists and executed."""
    table = _driver_imports()
    for script in ("GripNet-pose.py", "GripNet-aminer.py", "baselines/LP_baselines/rgcn_pose.py"):
        src = "\n".join("from {} import {}".format(m.replace("gripnet", "gripnet_amd", 1), ", ".join(n))
                        for m, n in table[script].items())
        scope = {}
        exec(src, scope)
        for names in table[script].values():
            assert all(n in scope for n in names)


def test_remaining_helpers_match_reference(golden):
    g = golden("utils_helpers")
    sp = utils.sparse_id(g.meta["n"])
    assert str(sp.layout) == g.meta["sparse_layout"] and str(sp.dtype) == g.meta["sparse_dtype"]
    assert str(sp.device) == g.meta["sparse_device"] and list(sp.shape) == g.arrays["sparse_id.shape"].tolist()
    assert torch.equal(sp._indices(), g.t("sparse_id.indices")) and torch.equal(sp._values(), g.t("sparse_id.values"))
    assert torch.equal(sp.to_dense(), g.t("sparse_id.dense"))
    assert torch.equal(utils.normalize(g.t("normalize.in")), g.t("normalize.out"))
    np.random.seed(g.meta["seed_edge"])
    tr, te = utils.process_edge(g.t("process_edge.in"))
    assert torch.equal(tr, g.t("process_edge.train")) and torch.equal(te, g.t("process_edge.test"))
    np.random.seed(g.meta["seed_nodes"])
    outs = utils.process_node_multilabel([g.t("nodes{}".format(i)) for i in range(g.meta["n_lists"])])
    for k, v in zip(("train_idx", "train_class", "train_range", "test_idx", "test_class", "test_range"), outs):
        assert torch.equal(v, g.t("multilabel." + k)), k


@pytest.mark.parametrize("sanitizer,flags", [("asan", "-fsanitize=address,undefined"), ("tsan", "-fsanitize=thread")])
def test_plan_builders_host_layout_under_sanitizers(tmp_path, sanitizer, flags):
    """The pure host side of the plan builders (gripnet_amd/csrc/host_layout.hpp: pairing and row classes of the decoder,
    the destination-major streams of the relational layer, the LDS-staged schedule of the gene layers) built with g++
    under AddressSanitizer + UBSan and under ThreadSanitizer; tests/host_layout_san.cpp runs each builder on 1 and on 16
    threads, checks that every edge lands in exactly one slot and that a plan does not depend on the thread count
    (sanitizers; the same programs: make -C gripnet_amd/csrc SAN=asan san / SAN=tsan san)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / ("host_layout_" + sanitizer)
    cmd = ["g++", "-std=c++17", "-O1", "-g", flags, "-fno-omit-frame-pointer", "-pthread",
           "-I", os.path.join(REPO, "gripnet_amd", "csrc"), os.path.join(REPO, "tests", "host_layout_san.cpp"), "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    if run.returncode != 0 and "unexpected memory mapping" in run.stderr:       # a sandbox whose address-space layout TSan refuses
        pytest.skip("the sanitizer runtime does not start in this sandbox: " + run.stderr.strip()[:200])
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-3000:])
    assert "all builders ok" in run.stdout
    if sanitizer == "asan":            # (TSan does not follow new threads in the child of a multi-threaded fork)
        # the builder threads are parked between builds; a forked child (a DataLoader worker, multiprocessing) has none of them
        run = subprocess.run([str(exe), "fork"], capture_output=True, text=True, timeout=300)
        assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-3000:])
        assert "ok across fork" in run.stdout
