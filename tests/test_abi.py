"""The C-ABI library loads and exports every symbol include/gripnet_hip.h declares (no GPU)."""
import ctypes
import os
import re

import pytest
import torch

import gripnet_amd
from gripnet_amd import _hip

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "gripnet_hip.h")).read()
    return sorted(set(re.findall(r"GN_API[^;(]*?\b(gn_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    names = declared_symbols()
    assert len(names) >= 15
    assert sorted(_hip.SIGNATURES) == names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_hip.library_path()), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_hip.library_path())
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert _hip.load().gn_version() == _hip.ABI_VERSION
    assert _hip.load().gn_last_error() is not None


def test_no_cpu_fallback():
    m = gripnet_amd.homoGraph([4, 4], start_graph=True, in_dim=3)
    with pytest.raises(RuntimeError, match="MI355X only"):
        m(None, torch.zeros(2, 2, dtype=torch.long))
    d = gripnet_amd.multiRelaInnerProductDecoder(4, 2)
    with pytest.raises(RuntimeError, match="MI355X only"):
        d(torch.zeros(3, 4), torch.zeros(2, 2, dtype=torch.long), torch.zeros(2, dtype=torch.long))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "gripnet_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cuh")):
                text = open(os.path.join(root, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), os.path.join(root, f)


def test_state_dict_keys_match_reference(golden):
    g = golden("pose_tiny")
    from gripnet_amd.pipeline import PoseModel
    model = PoseModel(g.meta["n_g"], g.meta["n_d"], g.meta["R"])
    want = {k[3:]: v.shape for k, v in g.arrays.items() if k.startswith("sd.")}
    have = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert have == {k: tuple(s) for k, s in want.items()}
    assert model.gg.out_dim == 16 and model.gg.n_cov == 2 and model.dd.out_dim == 32


def test_cache_protocol_is_keyed_by_edge_count_only():
    conv = gripnet_amd.myGCN(4, 4, cached=True)
    conv.cached_result, conv.cached_num_edges = object(), 7
    with pytest.raises(RuntimeError, match="Cached 7 number of edges, but found 3"):
        conv._plan(torch.zeros(2, 3, dtype=torch.long), lambda: None)
    assert conv._plan(torch.zeros(2, 7, dtype=torch.long), lambda: None) is conv.cached_result
    conv.reset_parameters()
    assert conv.cached_result is None and conv.cached_num_edges is None


def test_kernel_register_budgets():
    """The occupancy the design counts on, read from the gfx950 code objects' metadata (no GPU): the destination-major
    relational kernel runs sixteen waves per compute unit (four per SIMD: at most 128 registers, and its gather - written
    in assembly on fixed physical registers v92-v119 - leaves no room for scratch traffic inside the loop), the row-class
    decoder and the LDS-staged gene gather likewise.  A toolchain change that makes one of them spill fails here, not as
    a silent slowdown."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from kernel_resources import kernel_resources
    res = kernel_resources(_hip.library_path())
    assert len(res) > 50, "no kernels found in the gfx950 code objects"
    # the kernels of the headline step: no scratch at all
    # (k_rgcn_pair's last argument: 0 = the whole layer, 1 = the pair sums only, 2 = their contraction - round 6's two-launch form)
    strict = ["k_rgcn_pair<3, 2, 3, true, 0>", "k_rgcn_pair<3, 2, 3, false, 0>", "k_rgcn_pair<3, 2, 2, false, 0>",
              "k_rgcn_pair<1, 2, 3, false, 1>", "k_rgcn_pair<3, 2, 3, true, 2>", "k_distmult_class<5, 3>",
              "k_distmult_class<3, 3>", "k_distmult_class<2, 2>", "k_col_gather<2>", "k_col_gather<1>", "k_col_transform<32, 1, 2>",
              "k_col_transform<16, 1, 2>"]
    for name in strict:
        assert name in res, (name, sorted(k for k in res if "rgcn_pair" in k or "distmult_class" in k or "col_" in k))
        r = res[name]
        assert r[".vgpr_count"] + r.get(".agpr_count", 0) <= 128, (name, r)
        assert r[".private_segment_fixed_size"] == 0, (name, r)
        assert r[".vgpr_spill_count"] == 0 and r[".sgpr_spill_count"] == 0, (name, r)
    # every instantiation of the destination-major kernel keeps four waves per SIMD (the narrow ones park up to sixteen
    # rows of basis per thread in their epilogue and spill at most eight registers THERE, behind the unit loop: round 5 - they
    # spilled 14-24, 19 MB of scratch traffic per launch of the reversed layer of a training step)
    for name, r in res.items():
        if name.startswith("k_rgcn_pair<"):
            assert r[".vgpr_count"] + r.get(".agpr_count", 0) <= 128, (name, r)
            narrow = name.startswith(("k_rgcn_pair<1,", "k_rgcn_pair<2,"))
            assert r[".private_segment_fixed_size"] <= (64 if narrow else 160), (name, r)


def test_host_scratch_release_without_a_build():
    """gn_host_scratch_release is host-only: in a process that has built no plan there is nothing to free."""
    assert _hip.release_host_scratch() == 0
    assert _hip.release_host_scratch() == 0


def test_env_hooks_cover_the_library():
    """A memoised forward (_hip.CallMemo) replays recorded kernel choices; its key carries the library's environment hooks.
    Every getenv("GN_...") of the C sources must be in _hip._ENV_HOOKS (part of the key) or in _hip._ENV_NEUTRAL (declared
    unable to change a forward's launches) - a new hook that is in neither fails here instead of replaying stale choices."""
    found = set()
    csrc = os.path.join(REPO, "gripnet_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h", ".hpp", ".cuh", ".inc")):
            found |= set(re.findall(r'getenv\(\s*"(GN_[A-Z0-9_]+)"', open(os.path.join(csrc, f)).read()))
    assert found, "no getenv calls found: the pattern is stale"
    assert found == set(_hip._ENV_HOOKS) | set(_hip._ENV_NEUTRAL), (sorted(found), _hip._ENV_HOOKS, _hip._ENV_NEUTRAL)
    assert len(_hip.env_stamp()) == len(_hip._ENV_HOOKS)


def test_generated_gather_is_current_and_the_generator_does_not_write_by_default():
    """tools/gen_pair_asm.py without --write only compares (its --help once rewrote the tracked .inc and made every object stale)."""
    import subprocess
    import sys
    inc = os.path.join(REPO, "gripnet_amd", "csrc", "rgcn_pair_asm.inc")
    before = os.stat(inc).st_mtime_ns
    for extra in ([], ["--help"]):
        r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_pair_asm.py")] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    assert os.stat(inc).st_mtime_ns == before
