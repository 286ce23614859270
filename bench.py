#!/usr/bin/env python3
"""Headline benchmark: edges aggregated/sec for one steady-state GripNet forward
(gg -> gd -> dd -> DistMult on the positive edges) on the synthetic PoSE-0 supergraph.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one pass of the hot path over the whole (synthetic) supergraph with every input
already resident in HBM and the graph plans cached, as the reference caches its normalised
edge list after the first epoch (gripnet/layers.py:83-90).  fp32 arithmetic, int64 indices.

N = 1 : workload `pose0-syn` (BASELINE.json configs[1]; SURVEY.md 8d ladder).
N > 1 : weak scaling (default).  The dd relation set grows with N (N x the pose0-syn dd edge
        budget, N=2 ~ pose1-syn, N=4 ~ pose2-syn); every rank owns one contiguous edge range of
        the type-sorted dd edge list (= relation-id sharding balanced by edge count), computes
        the un-normalised RGCN partial for its range, one RCCL all-reduce of the [n_d,32]
        partial, then finalises and scores its own edge range with the DistMult decoder.  The
        small gene layers (gg, gd) are replicated and counted once.
        `--scaling strong` keeps the graph fixed (default workload pose2-syn) and cuts its dd
        edges into N ranges: `value` is then directly comparable with the N = 1 run of the same
        workload.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant
kernel, HIP-event timed inside the timed region) and `cpu_baseline` (the oracle, i.e. a port
of the reference's op sequence, timed on this box's host cores; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
# What in-kernel stamps and the issue-rate probe say bounds each entry point: none of them is HBM-bound;
# "bound": "hbm" names the roofline the path is priced against, this names what actually limits it.
LIMITERS = {
    "gn_rgcn_forward_f32": "instruction issue (7.6 M vector + 2.7 M scalar + 1.3 M LDS instructions per launch; waves wait to be picked 34 % and at "
                           "s_waitcnt 41 % of their resident cycles) next to the 121 MB of x-plane pieces a launch pulls through the L2s; evening out "
                           "the waves or the workgroups (63 / 42 units) was built twice and bought nothing (profiles/r05_experiments.md 11-12); "
                           "6.3 us epilogue; MFMA pipe 9 % busy; HBM moves 18 MB, less than the algorithmic bytes (DESIGN.md 4.1)",
    "gn_distmult_plan_forward_f32": "VALU issue (4.1 M instructions = ~6.6 us of the ~10 us loop) next to the LDS array (busy 5.9 us, 21 % of it "
                                    "bank conflicts); 4.4 us table fill from the Infinity Cache; scores written through one L2 per list range "
                                    "(8.3 MB = the scores themselves) (DESIGN.md 4.2, profiles/r04_decoder_pmc.md)",
    "gn_distmult_forward_f32": "HBM index stream / LDS reads",
    "gn_graph_aggregate_f32[gcn]": "launch ramp / drain + LDS-DMA fill of the 153 KB table slice from the Infinity Cache (4.5 of 7.6 us in-kernel), "
                                   "two launches per layer (DESIGN.md 4.3)",
}



def spin_up(fn, seconds=0.12):
    """Un-timed calls of `fn` for about `seconds`, then a synchronisation: the device's clocks follow its load, and behind host-side
    work (building plans, capturing a graph, a CPU oracle run) the first milliseconds of a timed loop ran 6-8 % slower than the
    steady state every figure of this file is meant to be (tools/step_only.py: 40 steps 82-83 us each, 60,000 steps 76.6, same box).
    GN_BENCH_SPIN_UP=0 turns it off."""
    import torch
    if os.environ.get("GN_BENCH_SPIN_UP") == "0":
        return
    t = time.perf_counter()
    while time.perf_counter() - t < seconds:
        for _ in range(4):
            fn()
    torch.cuda.synchronize()

def algorithmic_bytes(data, n_dd_edges_local, n_dd_edges_global):
    """Compulsory HBM bytes with the reference's data types: int64 indices, fp32
    values, node / parameter tables read once, no temporaries."""
    n_g, n_d, R = data.n_g_node, data.n_d_node, data.n_dd_edge_type
    e_gg = int(data.gg_edge_index.shape[1]) + n_g          # E' (self loops appended; synthetic gg has none)
    e_gd = int(data.gd_edge_index.shape[1])
    gcn = lambda e, n, fi, fo: e * 20 + n * 4 * (fi + fo)
    return {
        "gg": gcn(e_gg, n_g, 32, 16) + gcn(e_gg, n_g, 16, 16),
        "gd": e_gd * 20 + n_g * 64 * 4 + n_d * (16 + 2 * 32) * 4,
        "dd": n_dd_edges_local * 16 + n_d * 4 * (48 + 32) + 4 * (32 * 48 * 32 + R * 32 + 48 * 32),
        "dmt": n_dd_edges_local * 28 + n_d * 80 * 4 + R * 80 * 4,
    }


def nc_algorithmic_bytes(model, n_nodes_scored):
    """Compulsory HBM bytes of one node-classification forward with the reference's data types (the same
    model as the headline): per GCN-style layer E' x 20 + N x 4 x (F_in + F_out) (the transform fused, tables once), the
    external layers E x 20 + source table + target rows, the three-way merge of freebase-c, the class decoder's gathered
    rows.  Read off the modules and their cached plans after a forward.  Returns (total, per entry-point tag)."""
    from gripnet_amd.layers import homoGraph, interGraph
    from gripnet_amd.decoder import multiClassInnerProductDecoder
    per = {}

    def add(tag, b):
        per[tag] = per.get(tag, 0) + int(b)
    for m in model.modules():
        if isinstance(m, homoGraph) and not m.multi_relational:
            for conv in m.conv_list:
                plan = conv.cached_result
                add("gn_graph_aggregate_f32[gcn]", plan.nnz * 20 + plan.n_rows * 4 * (conv.in_channels + conv.out_channels))
        elif isinstance(m, interGraph):
            plan = m.conv.cached_result
            tgt = m.target_dim + (2 * m.target_feat_dim if m.if_one_external else 0)
            add("gn_graph_aggregate_f32[bipartite]", plan.nnz * 20 + plan.n_table * m.source_dim * 4 + plan.n_rows * tgt * 4)
        elif isinstance(m, multiClassInnerProductDecoder):
            add("gn_class_scores_f32", n_nodes_scored * (8 + 4 * m.in_dim + 4 * m.num_class) + 4 * m.in_dim * m.num_class)
    if hasattr(model, "aa_embeddings"):                          # (z + z1 + aa_embeddings) / 3, GripNet-freebase-c.py:159
        add("gn_merge_f32", 4 * model.aa_embeddings.numel() * 4)
    return sum(per.values()), per


def train_step_entry(dev, steps=20):
    """One full training step of the PoSE model on pose0-syn (forward, both decoder calls, loss, backward, Adam: the loop
    body of GripNet-pose.py:112-146, negative sampling included) as ONE hipGraph replay per step: the draw
    (gn_negative_sampler_sample_stepped: its seed moves with a counter on the device) is the graph's first node."""
    from gripnet_amd import _hip
    from gripnet_amd.pipeline import PoseModel
    from gripnet_amd.synth import make_pose
    from gripnet_amd.utils import link_prediction_loss
    from gripnet_amd.synth import add_pose_test_split
    data = add_pose_test_split(make_pose("pose0-syn")).to(dev)
    torch.manual_seed(1111)
    model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
    from gripnet_amd.optim import Adam
    opt = Adam(model.parameters(), lr=0.01)                    # gn_adam_step_f32: all parameters in one launch
    sampler = _hip.NegativeSampler(data.train_idx, data.n_d_node, data.train_range)
    neg = sampler.sample(seed=0)
    drawn = torch.ones((1,), dtype=torch.int64, device=dev)    # draw counter on the device: a replay draws seed + counter

    def step():
        sampler.sample(seed=0, out=neg, step=drawn)            # gn_negative_sampler_sample_stepped
        opt.zero_grad()
        z = model.encode(data)
        # GripNet-pose.py:137-142 (both decoder calls + the loss) as one autograd node: its backward computes the loss's derivative
        # inside the two decoder-backward launches (utils.link_prediction_loss; the same bits as the three separate calls)
        loss, pos, negs = link_prediction_loss(model.dmt, z, data.train_idx, neg, data.train_et)
        loss.backward(one)                                     # (the seed gradient is kept: backward() would fill a new 1 per step)
        opt.step()
        kept[:] = [z, pos, negs]                               # what the epoch's metrics and test() read (GripNet-pose.py:148-160,213-215)
        return loss

    one = torch.ones((), dtype=torch.float32, device=dev)
    kept = []
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(5):                                     # plans (the decoder's on the second sighting of its list, its backward
            step()                                             # plan in that step's backward), relation-order check, optimizer state
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    # the eager step (the Python loop a caller of GripNet-pose.py:112-146 runs), and its entry points between HIP events
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    eager_ms = 1e3 * (time.perf_counter() - t0) / 10
    with _hip.KernelTimer(pool=400) as timer:
        for _ in range(3):
            step()
    per_entry = {k: 1e3 * tot / 3 for k, (calls, tot) in timer.summary().items()}      # us per step (all calls of the entry point)
    graph = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        loss = step()
    losses = []
    for _ in range(3):
        graph.replay()
    spin_up(graph.replay)
    before = neg.clone()
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    losses.append(float(loss.detach()))
    if steps > 0 and torch.equal(before, neg):
        raise RuntimeError("the replayed training step did not draw new negatives")
    _hip.raise_if_index_errors(dev)
    # ---- algorithmic bytes of the step (SURVEY.md 8d's model carried to the backward pass: the reference's data types,
    # every edge list read once per pass that needs it, node / parameter tables and their gradients once) ----
    E = int(data.train_idx.shape[1])
    fwd = algorithmic_bytes(data, E, E)
    n_g, n_d, R = data.n_g_node, data.n_d_node, data.n_dd_edge_type
    e_gg, e_gd = int(data.gg_edge_index.shape[1]) + n_g, int(data.gd_edge_index.shape[1])
    dm_bwd = E * 28 + 2 * (n_d + R) * 80 * 4                  # one list: (u, v, r, g) once, z and D read, dz and dD written
    alg = {
        "forward (gg, gd, dd, DistMult on the positives)": fwd["gg"] + fwd["gd"] + fwd["dd"] + fwd["dmt"],
        "negative sampling (E pairs written as int64 + the positives' pair set read)": E * 16 + E * 16,
        "DistMult on the negatives": fwd["dmt"],
        "loss forward + backward": 2 * 2 * E * 4 + 2 * E * 4,
        "DistMult backward, positives + negatives": 2 * dm_bwd,
        "relational layer backward (dx over the reversed edges, dW_r -> dbasis / datt, droot)": 2 * E * 16 + 2 * n_d * 4 * (48 + 32) + 2 * 4 * (32 * 48 * 32 + R * 32 + 48 * 32),
        "external layer backward": e_gd * 20 + 2 * (n_g * 64 * 4 + n_d * 48 * 4),
        "gene layers backward": 2 * e_gg * 20 + 2 * n_g * 4 * (32 + 16 + 16 + 16),
        "Adam (parameters, gradients, two moments read; parameters and moments written)": 7 * 4 * sum(p.numel() for p in model.parameters()),
    }
    alg_total = int(sum(alg.values()))
    bwd_entries = {k: v for k, v in per_entry.items() if "backward" in k or k in ("gn_rel_weight_grad_f32", "gn_graph_aggregate_t_f32", "gn_grad_prologue_f32",
                                                                                 "gn_dense_batch_end", "gn_xtg_f32", "gn_gemm_f32")}
    dom = max(bwd_entries, key=bwd_entries.get) if bwd_entries else None
    dom_bytes = {"gn_distmult_backward_packed_f32": dm_bwd, "gn_distmult_backward_planned_f32": dm_bwd, "gn_distmult_backward_f32": dm_bwd,
                 "gn_rel_weight_grad_f32": E * 16 + n_d * 4 * (48 + 32) + R * 48 * 32 * 4}.get(dom)
    epoch = epoch_entry(model, data, graph, kept, dt)
    return {"workload": "pose0-syn training step", "ms_per_step": round(1e3 * dt, 4), "steps": steps,
            "epoch": epoch,
            "ms_per_step_eager": round(eager_ms, 4),
            "what": "negative sampling + forward + DistMult on positives and on the fresh negatives + loss (utils.link_prediction_loss: the loss derivative computed inside the decoder backward launches) + backward + Adam (gripnet_amd.optim.Adam, one launch): "
                    "ms_per_step = one hipGraph replay per step and nothing else; the draw (typed sampler) is the graph's first node, its seed moves with a counter on the device; "
                    "ms_per_step_eager = the same step as a Python loop over the modules (host-bound)",
            "algorithmic_bytes": alg_total, "algorithmic_bytes_by_part": {k: int(v) for k, v in alg.items()},
            "frac": round(alg_total / (dt * 1e9) / HBM_PEAK_GBS, 4),
            "entry_point_us_per_step": {k: round(v, 1) for k, v in sorted(per_entry.items(), key=lambda kv: -kv[1])},
            "dominant_backward_entry_point": None if dom is None else {
                "name": dom, "us_per_step": round(bwd_entries[dom], 1), "algorithmic_bytes": dom_bytes,
                "frac": None if dom_bytes is None else round(dom_bytes / (bwd_entries[dom] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
            "loss_after": round(losses[-1], 5)}


def epoch_entry(model, data, graph, kept, train_dt, epochs=10):
    """One EPOCH of the reference's loop (GripNet-pose.py:210-225): train() = the replayed training step + per-relation
    AUPRC / AUROC / AP of the step's positive and negative scores (:148-160), then test(z) = DistMult on the held-out
    positives and on the static test negatives + the same metrics (:180-201), the six [R] vectors' means read back by the
    host as the reference prints them.  us per part (a device synchronisation between the parts, as the metrics' read-back
    is one anyway) and the whole epoch without the extra synchronisations."""
    from gripnet_amd.utils import relation_metrics, typed_negative_sampling
    z, pos, negs = [t.detach() for t in kept]
    dev = z.device
    import numpy as np
    test_neg = typed_negative_sampling(data.test_idx, data.n_d_node, data.test_range, rng=np.random.RandomState(1111))  # static (pose.py:175-177)
    E, Et, R = int(data.train_idx.shape[1]), int(data.test_idx.shape[1]), int(data.n_dd_edge_type)

    def test_scores():
        with torch.no_grad():
            return model.dmt(z, data.test_idx, data.test_et), model.dmt(z, test_neg, data.test_et)

    def epoch(parts=None):
        t = [time.perf_counter()]
        def mark():
            if parts is not None:
                torch.cuda.synchronize()
                t.append(time.perf_counter())
        graph.replay(); mark()                                       # train(): the replayed step
        tr = relation_metrics(pos, negs, data.train_range); mark()   # pose.py:148-160
        tp, tn = test_scores(); mark()                               # pose.py:185-186
        te = relation_metrics(tp, tn, data.test_range); mark()       # pose.py:188-199
        host = torch.stack(list(tr) + list(te)).nanmean(dim=1).cpu() # record.mean(axis=1), both records
        t.append(time.perf_counter())
        if parts is not None:
            for k, name in enumerate(("train_step", "train_metrics", "test_decoder_x2", "test_metrics", "host_readback")):
                parts[name] = parts.get(name, 0.0) + (t[k + 1] - t[k])
        return host

    for _ in range(2):
        last = epoch()
    spin_up(epoch)
    t0 = time.perf_counter()
    for _ in range(epochs):
        last = epoch()
    torch.cuda.synchronize()
    whole = (time.perf_counter() - t0) / epochs
    parts = {}
    for _ in range(epochs):
        epoch(parts)
    parts_us = {k: round(1e6 * v / epochs, 1) for k, v in parts.items()}
    # algorithmic bytes of the metrics: the 2 E scores read once, ONE sort of the 2 E (score key, label) pairs (read and
    # written once: 8 + 4 bytes each way), 3 R doubles written
    m_bytes = lambda e: 2 * e * 4 + 2 * e * 12 * 2 + 3 * R * 8
    dm_bytes = lambda e: e * 28 + int(data.n_d_node) * 80 * 4 + R * 80 * 4
    out = {"workload": "pose0-syn epoch (GripNet-pose.py:210-225: train() incl. its per-relation metrics, then test())",
           "ms_per_epoch": round(1e3 * whole, 4), "epochs": epochs, "E_train": E, "E_test": Et,
           "us_by_part": parts_us,
           "metrics": {"train": {"us": parts_us["train_metrics"], "algorithmic_bytes": m_bytes(E),
                                 "frac": round(m_bytes(E) / (parts_us["train_metrics"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                       "test": {"us": parts_us["test_metrics"], "algorithmic_bytes": m_bytes(Et),
                                "frac": round(m_bytes(Et) / (parts_us["test_metrics"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                       "share_of_epoch": round((parts_us["train_metrics"] + parts_us["test_metrics"]) / max(sum(parts_us.values()), 1e-9), 4)},
           "test_decoder": {"us": parts_us["test_decoder_x2"], "algorithmic_bytes": 2 * dm_bytes(Et),
                            "frac": round(2 * dm_bytes(Et) / (parts_us["test_decoder_x2"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
           "mean_metrics_last_epoch": {"train_auprc_auroc_ap": [round(float(v), 4) for v in last[:3]],
                                       "test_auprc_auroc_ap": [round(float(v), 4) for v in last[3:]]},
           "note": "us_by_part: with a device synchronisation between the parts; ms_per_epoch: the same epoch without them "
                   "(the metrics' own read-back is the only synchronisation, as in the reference's loop)"}
    return out


def nc_train_entry(model, data, nodes, labels, steps=10):
    """us per training step of a node-classification model: a Python loop over the modules (host-bound), and one hipGraph replay."""
    from gripnet_amd.optim import Adam
    from gripnet_amd.utils import class_loss
    opt = Adam(model.parameters(), lr=0.01)

    def step():
        opt.zero_grad()
        _, score = model(data, nodes)
        loss = class_loss(score, labels)
        loss.backward(one)
        opt.step()
        return loss

    one = torch.ones((), dtype=torch.float32, device=nodes.device)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):                                     # plans, optimizer state
            first = float(step().detach())
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    eager_us = 1e6 * (time.perf_counter() - t0) / steps
    out = {"us_per_step_eager": round(eager_us, 1)}
    try:
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            loss = step()
        for _ in range(2):
            graph.replay()
        spin_up(graph.replay)
        t0 = time.perf_counter()
        for _ in range(steps):
            graph.replay()
        torch.cuda.synchronize()
        out["us_per_step"] = round(1e6 * (time.perf_counter() - t0) / steps, 1)
        out["loss_first_last"] = [round(first, 4), round(float(loss.detach()), 4)]
        del graph
    except Exception as exc:                                   # (the eager figure stands; say why the replay does not)
        out["us_per_step"] = None
        out["graph_error"] = str(exc)[:200]
    return out


def extra_workloads(dev, budget_s, with_cpu):
    """The other BASELINE.json configs on one GPU, each with the same event timing as the headline and a CPU-oracle time:
    pose2-syn (config 4's graph, unsharded), aminer-syn (config 3), freebase-c-syn (config 5, fp32 storage)."""
    from gripnet_amd import _hip
    from gripnet_amd.pipeline import AminerModel, FreebaseCModel, PoseModel, PoseStages
    from gripnet_amd.synth import Data, make_nc, make_pose, pose_edges_aggregated
    out, t_begin = [], time.perf_counter()

    def left():
        return budget_s - (time.perf_counter() - t_begin)

    def timed(fn, n):
        for _ in range(3):
            fn()
        spin_up(fn)
        with _hip.KernelTimer(pool=600) as t:                 # events created (and first recorded) up front: a fresh event's first record costs ~100 us
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / n
        return wall, {k: 1e3 * tot / calls for k, (calls, tot) in t.summary().items()}, sum(1e3 * tot / n for _, tot in t.summary().values())

    with torch.no_grad():
        # ---- pose2-syn: the full step, its relational layer and its decoder against their algorithmic bytes ----
        if left() > 20:
            data_cpu = make_pose("pose2-syn")
            torch.manual_seed(1111)
            model = PoseModel(data_cpu.n_g_node, data_cpu.n_d_node, data_cpu.n_dd_edge_type).to(dev)
            data = Data(**data_cpu.__dict__).to(dev)
            eager = PoseStages(model, data, graphs=False)
            for _ in range(3):
                eager.step()
            _, calls, _ = timed(eager.step, 10)
            stages = PoseStages(model, data, recorded=True)   # the step's entry-point calls recorded once, made again from one loop
            for _ in range(5):
                stages.step()
            spin_up(stages.step)
            t0 = time.perf_counter()
            for _ in range(30):
                stages.step()
            torch.cuda.synchronize()
            step_us = 1e6 * (time.perf_counter() - t0) / 30
            E = int(data.train_idx.shape[1])
            alg = algorithmic_bytes(data, E, E)
            A = pose_edges_aggregated(data)
            rel_us = calls.get("gn_rgcn_forward_f32")
            dec_us = calls.get("gn_distmult_plan_forward_f32", calls.get("gn_distmult_forward_f32"))
            # what its two large plans cost a caller the first time (the gene and external plans are pose0-syn's: cold_start)
            def build_ms(fn, n=2):
                best = float("inf")
                for _ in range(n):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    keep = fn()
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t1)
                    del keep
                return round(1e3 * best, 2)
            from gripnet_amd import _hip as _h
            plan_ms = {"relational layer": build_ms(lambda: _h.RgcnPlan(data.train_idx, data.train_range, data.n_d_node)),
                       "relational layer, light plan": build_ms(lambda: _h.RgcnPlan(data.train_idx, data.train_range, data.n_d_node, light=True)),
                       "decoder": build_ms(lambda: _h.DistMultPlan(data.train_idx, data.train_et, data.n_d_node, data.n_dd_edge_type, model.dmt.in_dim))}
            out.append({"workload": "pose2-syn", "E_dd": E, "us_per_step": round(step_us, 1), "edges_per_s": A / (step_us * 1e-6),
                        "plan_build_ms": plan_ms,
                        "launch": "recorded entry-point calls made again from one loop (as the headline)",
                        "relational": {"us": round(rel_us, 1), "frac": round(alg["dd"] / (rel_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                        "decoder": {"us": round(dec_us, 1), "frac": round(alg["dmt"] / (dec_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}})
            del model, data, eager, stages
            torch.cuda.empty_cache()
        # ---- rgcn-pose-syn: the reference's all-nodes relational baseline (baselines/LP_baselines/rgcn_pose.py:53-106):
        #      N = 19,726, R = 964, 64 -> 32 -> 32, 16 bases - the O(E)-memory general relational path (rgcn_basis.hip) ----
        from oracle import gripnet_oracle as orc
        if left() > 25:
            from gripnet_amd.pipeline import RgcnPoseModel
            from gripnet_amd.synth import make_rgcn_pose
            data_cpu = make_rgcn_pose("pose0-syn")
            torch.manual_seed(1111)
            model = RgcnPoseModel(data_cpu.n_node, data_cpu.n_edge_type)
            sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
            model = model.to(dev)
            data = Data(**data_cpu.__dict__).to(dev)
            for _ in range(3):
                model(data)
            wall, calls, busy = timed(lambda: model(data), 10)
            E, N, R = int(data.train_idx.shape[1]), int(data.n_node), int(data.n_edge_type)
            rel_bytes = [E * 16 + N * 4 * (fi + fo) + 4 * (16 * fi * fo + R * 16 + fi * fo) for fi, fo in ((64, 32), (32, 32))]
            dec_bytes = E * 28 + N * 32 * 4 + R * 32 * 4
            rel_us = calls.get("gn_rgcn_forward_f32")
            plan = model.rgcn1._plan
            ws = int(_hip.load().gn_rgcn_workspace_bytes(plan._h, 64, 32, 16, 0))
            entry = {"workload": "rgcn-pose-syn", "nodes": N, "relations": R, "edges": E, "layers": "64 -> 32 -> 32, 16 bases",
                     "forward_us_entry_points": round(busy, 1), "forward_us_eager_wall": round(1e6 * wall, 1),
                     "algorithmic_bytes": sum(rel_bytes) + dec_bytes,
                     "frac": round((sum(rel_bytes) + dec_bytes) / (busy * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "relational": {"kernel": plan.path(64, 32, 16), "us_per_call": round(rel_us, 1),
                                    "algorithmic_bytes_per_call": sum(rel_bytes) // 2,
                                    "frac": round(sum(rel_bytes) / 2 / (rel_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                    "workspace_bytes": ws, "table_path_workspace_bytes": R * N * 32 * 4},
                     "note": "us_per_call: average over the two relational layers (HIP events around the entry point: the gather "
                             "launches and the dense products of its slabs of rows); workspace independent of R and N"}
            if with_cpu and left() > 20:
                d = data_cpu
                t1 = time.perf_counter()
                h = orc.rgcn_forward(sd["embedding"], d.train_idx, d.train_range, sd["rgcn1.basis"], sd["rgcn1.att"], sd["rgcn1.root"])
                zr = orc.rgcn_forward(h, d.train_idx, d.train_range, sd["rgcn2.basis"], sd["rgcn2.att"], sd["rgcn2.root"])
                entry["cpu_oracle_encoder_s"] = round(time.perf_counter() - t1, 3)
                entry["cpu_threads"] = torch.get_num_threads()
                entry["parity_max_abs_err_z"] = float((model.encode(data).cpu() - zr).abs().max())
            out.append(entry)
            del model, data
            torch.cuda.empty_cache()
        # ---- the node-classification models (configs 3 and 5) ----
        from gripnet_amd.utils import set_table_storage
        nc_refs = {}                                     # the oracle's forward per model (the bf16 run is checked against the same one)
        for name, cls, storage in (("aminer-syn", AminerModel, "fp32"), ("freebase-c-syn", FreebaseCModel, "fp32"),
                                   ("freebase-c-syn, bf16 table storage", FreebaseCModel, "bf16")):
            if left() < 15:
                break
            data_cpu = make_nc("aminer-syn")             # the NC ladder shares one synthetic scale
            torch.manual_seed(1111)
            model = (cls(data_cpu.n_p_node, data_cpu.n_a_node, data_cpu.n_a_type) if cls is AminerModel else
                     cls(data_cpu.n_p_node, data_cpu.n_q_node, data_cpu.n_a_node, data_cpu.n_a_type))
            sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
            nodes = torch.arange(0, data_cpu.n_a_node, 3)
            model = model.to(dev)
            if storage == "bf16":                       # BASELINE config 5 as written: gathered tables rounded to bf16, fp32 sums
                set_table_storage(model, "bf16")
            data = Data(**data_cpu.__dict__).to(dev)
            nodes_dev = nodes.to(dev)
            wall, calls, busy = timed(lambda: model(data, nodes_dev), 10)
            alg_total, alg_per = nc_algorithmic_bytes(model, int(nodes.numel()))
            dom = max(calls, key=calls.get)
            # the same forward as ONE hipGraph replay: what the device needs without the Python loop between the launches
            from gripnet_amd.pipeline import Graphed, Recorded
            replay = Graphed(lambda: model(data, nodes_dev)).capture()
            for _ in range(3):
                replay()
            spin_up(replay)
            t1 = time.perf_counter()
            for _ in range(20):
                replay()
            torch.cuda.synchronize()
            graph_us = 1e6 * (time.perf_counter() - t1) / 20
            del replay
            # ... and as its recorded entry-point calls (ordinary launches: no idle gap in front of a graph's first kernel) - only
            # if the forward is nothing but entry-point calls, i.e. the replay gives the eager forward's bits
            launch_mode = "one hipGraph replay per forward"
            flat = lambda o: [t for t in (o if isinstance(o, (tuple, list)) else (o,)) if torch.is_tensor(t)]
            want = [t.clone() for t in flat(model(data, nodes_dev))]
            rec = Recorded(lambda: model(data, nodes_dev)).capture()
            got = flat(rec())
            torch.cuda.synchronize()
            if len(got) == len(want) and all(torch.equal(x, y) for x, y in zip(got, want)):
                for _ in range(3):
                    rec()
                spin_up(rec)
                t1 = time.perf_counter()
                for _ in range(20):
                    rec()
                torch.cuda.synchronize()
                rec_us = 1e6 * (time.perf_counter() - t1) / 20
                if rec_us < graph_us:
                    graph_us, launch_mode = rec_us, "recorded entry-point calls made again from one loop"
            del rec
            dom_bytes = alg_per.get(dom)
            dom_calls = {"gn_graph_aggregate_f32[gcn]": sum(len(m.conv_list) for m in model.modules() if hasattr(m, "conv_list")),
                         "gn_graph_aggregate_f32[bipartite]": sum(1 for m in model.modules() if hasattr(m, "if_one_external"))}.get(dom, 1)
            entry = {"workload": name, "table_storage": storage, "forward_us": round(graph_us, 1), "forward_us_entry_points": round(busy, 1),
                     "forward_us_eager_wall": round(1e6 * wall, 1),
                     "algorithmic_bytes": alg_total, "frac": round(alg_total / (graph_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "launch": launch_mode,
                     "note": "forward_us: the faster of one hipGraph replay and the recorded entry-point calls, wall clock over 20 forwards; forward_us_entry_points: sum of the "
                             "HIP-event timed entry points of one eager forward (each pays its events); the wall time of the eager "
                             "Python loop is host-bound",
                     "dominant_entry_point": {"name": dom, "us_per_call": round(calls[dom], 1),
                                              "algorithmic_bytes_per_call": None if dom_bytes is None else dom_bytes // max(dom_calls, 1),
                                              "frac": None if dom_bytes is None else round(dom_bytes / max(dom_calls, 1) / (calls[dom] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}}
            if with_cpu and left() > 10:
                # parity of THIS forward against the CPU oracle, asserted in the run (fp32: the contract's 1e-4; bf16 storage has no
                # reference - one bf16 rounding of every gathered element: 2^-7 of the largest activation on z, 5e-3 on the class
                # probabilities, the tolerances of tests/test_gpu_parity.py)
                d = data_cpu
                key = "aminer" if cls is AminerModel else "freebase-c"
                if key not in nc_refs:
                    t1 = time.perf_counter()
                    if cls is AminerModel:
                        ref = orc.aminer_forward(sd, d.pp_edge_idx, d.pp_edge_weight, d.pa_edge_idx, d.aa_edge_idx, d.aa_edge_weight, nodes)
                    else:
                        ref = orc.freebase_c_forward(sd, d.pp_edge_idx, d.pp_edge_weight, d.pa_edge_idx, d.qq_edge_idx, d.qq_edge_weight,
                                                     d.qa_edge_idx, sd["aa_embeddings"], d.aa_edge_idx, d.aa_edge_weight, nodes, d.n_a_node)
                    nc_refs[key] = (ref, round(time.perf_counter() - t1, 3))
                ref, cpu_s = nc_refs[key]
                zg, pg = model(data, nodes_dev)
                err_z, err_p = float((zg.cpu() - ref["z"]).abs().max()), float((pg.cpu() - ref["score"]).abs().max())
                tol_z, tol_p = (1e-4, 1e-4) if storage == "fp32" else (2.0 ** -7 * float(ref["z"].abs().max()), 5e-3)
                entry["parity"] = {"max_abs_err_z": err_z, "max_abs_err_score": err_p, "tolerance_z": tol_z, "tolerance_score": tol_p,
                                   "ok": bool(err_z <= tol_z and err_p <= tol_p)}
                assert entry["parity"]["ok"], "{}: GPU result differs from the CPU oracle: {}".format(name, entry["parity"])
                entry["cpu_oracle_forward_s"] = cpu_s
                entry["cpu_threads"] = torch.get_num_threads()
            if left() > 12:
                # the model's training step (with bf16 table storage too, round 6: the forward gathers the rounded table, the backward is the fp32 layer's) (the loop body of GripNet-aminer.py:120-147 / GripNet-freebase-c.py:146-176: forward, class
                # loss, backward, Adam - every launch the library's own): eager, and as one hipGraph replay per step
                with torch.enable_grad():
                    tr = nc_train_entry(model, data, nodes_dev, data.a_label[nodes_dev].contiguous())
                # algorithmic bytes of the step (the forward's model carried to the backward pass, as for the PoSE step): every
                # layer's edge list once more (the transposed aggregation), its rows read (g, x) and written (dx) once, the class
                # decoder's rows both ways, Adam's seven passes over the parameters
                bwd = sum(v for k, v in alg_per.items() if k != "gn_merge_f32") + 2 * alg_per.get("gn_merge_f32", 0)
                adam = 7 * 4 * sum(p.numel() for p in model.parameters())
                tr["algorithmic_bytes"] = int(alg_total + bwd + adam)
                tr["algorithmic_bytes_by_part"] = {"forward": int(alg_total), "backward": int(bwd), "Adam": int(adam)}
                if tr.get("us_per_step"):
                    tr["frac"] = round(tr["algorithmic_bytes"] / (tr["us_per_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                tr["frac_eager"] = round(tr["algorithmic_bytes"] / (tr["us_per_step_eager"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                entry["training_step"] = tr
            out.append(entry)
            del model, data
            torch.cuda.empty_cache()
    if left() > 10:
        out.append(train_step_entry(dev))
        torch.cuda.empty_cache()
    return out


def cold_start_entry(model, data, fence):
    """What a caller pays the FIRST time (and a caller whose edge lists change, every time): the reference's myRGCN / decoder
    have no set-up cost (gripnet/layers.py:165-169, decoder.py:19-23), its myGCN normalises once (layers.py:83-90).  Every plan
    of the headline built again on its own, timed with the device idle on both sides, and the forward with NOTHING cached on
    the decoder's side (the raw int64 list streamed by the plan-less kernel)."""
    from gripnet_amd import _hip
    d = data
    def timed_ms(fn, n=3):
        best = float("inf")
        for _ in range(n):
            fence()
            t = time.perf_counter()
            keep = fn()
            fence()
            best = min(best, time.perf_counter() - t)
            del keep
        return round(1e3 * best, 2)
    out = {"plan_build_ms_by_plan": {}}
    per = out["plan_build_ms_by_plan"]
    # the forward with a decoder that keeps nothing: auto_static off -> the plan-less kernel on the raw int64 list
    from gripnet_amd.pipeline import PoseStages
    dmt = model.dmt
    was, seen = dmt.auto_static, list(dmt._seen)
    dmt.auto_static = False
    dmt.forget_static()
    try:
        stages = PoseStages(model, data, graphs=False)
        for _ in range(3):
            stages.step()
        reps = []
        import gc
        gc.collect()                                           # (a generation-2 collection of THIS process - 40-50 ms with sklearn, the oracle and
        spin_up(stages.step)
        for _ in range(3):                                     # three models alive - used to land in the first repeat: 2 ms per step instead of
            fence()                                            # 0.11, tools/uncached_steps_probe.py; collected here, and the fastest of three kept)
            t = time.perf_counter()
            for _ in range(20):
                z, score = stages.step()
            fence()
            reps.append((time.perf_counter() - t) / 20)
        out["forward_ms_decoder_uncached"] = round(1e3 * min(reps), 5)
        out["forward_ms_decoder_uncached_repeats"] = [round(1e3 * r, 5) for r in reps]
        with _hip.KernelTimer() as kt:
            for _ in range(5):
                stages.step()
        out["entry_point_us_decoder_uncached"] = {k: round(1e3 * tot / calls, 2) for k, (calls, tot) in kt.summary().items()}
    finally:
        dmt.auto_static = was
        dmt._seen = seen
        dmt.__dict__.pop("_memo", None)
    def gcn():
        p = _hip.GraphPlan.gcn(d.gg_edge_index, d.n_g_node, d.edge_weight, False)
        p.build_blocked(16)
        return p
    per["gene layers: self-loop rewrite + degree + norm + CSR (device), LDS-staged schedule (host)"] = timed_ms(gcn)
    per["external layer: bipartite CSR + padded rows"] = timed_ms(lambda: _hip.GraphPlan.bipartite(d.gd_edge_index, d.n_g_node, d.n_d_node, None))
    per["relational layer: destination-major key list (device sort), LDS-accumulator segments and destination-major units + LPT deal (host)"] = timed_ms(
        lambda: _hip.RgcnPlan(d.train_idx, d.train_range, d.n_d_node))
    per["decoder: row classes of the static list (host)"] = timed_ms(
        lambda: _hip.DistMultPlan(d.train_idx, d.train_et, d.n_d_node, d.n_dd_edge_type, model.dmt.in_dim))
    out["plan_build_ms_sum"] = round(sum(per.values()), 2)
    # ... and with NOTHING kept on the relational layer's side either: two copies of the dd edge list in turn, so that every
    # call sees a list it did not see the call before - a light plan (device sort only) + the general O(E) kernel per call
    conv = model.dd.conv_list[0]
    x48 = stages.x
    lists = [(d.train_idx.clone(), d.train_range.clone()) for _ in range(2)]
    per["relational layer, LIGHT plan (static_graph = False: the device-sorted key list only)"] = timed_ms(
        lambda: _hip.RgcnPlan(d.train_idx, d.train_range, d.n_d_node, light=True))
    conv.static_graph = False
    kept = (conv._plan, conv._plan_key)
    try:
        for k in range(4):
            model.dd(x48, lists[k & 1][0], edge_type=d.train_et, range_list=lists[k & 1][1], if_catout=True)
        fence()
        t = time.perf_counter()
        for k in range(10):
            model.dd(x48, lists[k & 1][0], edge_type=d.train_et, range_list=lists[k & 1][1], if_catout=True)
        fence()
        out["relational_layer_ms_uncached"] = round(1e3 * (time.perf_counter() - t) / 10, 4)
    finally:
        conv.static_graph = True
        conv._plan, conv._plan_key = kept
        model.dd.__dict__.pop("_memo", None)
    out["note"] = ("forward_ms_decoder_uncached: the eager step with the decoder scoring the raw int64 list (no plan, no remembered list); "
                   "relational_layer_ms_uncached: the dd layer alone on an edge list it did not see the call before (myRGCN.static_graph = False: "
                   "a light plan - the device-sorted key list - built per call, then the general O(E) kernel); every relational kernel reads a "
                   "destination-major encoding, so a sort per new list is the floor")
    return out


def spawn_ranks(n):
    """One process per GPU through torch.distributed.run on 127.0.0.1; returns the exit code for this process."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if proc.returncode != 0 or not lines:
        sys.stderr.write(proc.stdout)
        sys.stderr.write(proc.stderr)
        sys.stderr.write("bench.py: the {}-rank job failed (exit code {})\n".format(n, proc.returncode))
        return proc.returncode or 1
    print(lines[-1])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, help="pose0-syn (default), pose1-syn, pose2-syn; strong scaling defaults to pose2-syn")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch", choices=("auto", "eager", "graphs", "recorded"), default="auto",
                    help="eager: every entry point launched through the Python modules (~100-150 us of host work per step, "
                         "depending on the box, against ~77 us of kernels on pose0-syn); graphs: the stages replayed as "
                         "hipGraphs, except the dominant entry point, which stays a Python launch between HIP events; "
                         "recorded: the step's entry-point calls written down once and made again from one loop "
                         "(gripnet_amd.pipeline.Recorded: ordinary launches, no graph); auto: whichever of the three runs "
                         "the step fastest on this box, measured before the timed region")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra workloads (pose2-syn, aminer-syn, freebase-c-syn)")
    ap.add_argument("--extra-seconds", type=float, default=90.0, help="time budget of the extra workloads")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()
    if args.workload is None:
        args.workload = "pose2-syn" if args.scaling == "strong" else "pose0-syn"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks as children of this process (which has not touched the GPU
        # and will not), relay rank 0's JSON line, and fail loudly with the children's output if any of them fails
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)     # backend "nccl" is RCCL on ROCm

    from gripnet_amd import _hip
    from gripnet_amd.pipeline import PoseModel
    from gripnet_amd.synth import make_pose, pose_edges_aggregated
    from gripnet_amd.utils import shard_edge_ranges

    _hip.load()
    data_cpu = make_pose(args.workload, dd_scale=1 if args.scaling == "strong" else world)
    torch.manual_seed(1111)
    model_cpu = PoseModel(data_cpu.n_g_node, data_cpu.n_d_node, data_cpu.n_dd_edge_type)
    state_cpu = {k: v.detach().clone() for k, v in model_cpu.state_dict().items()}
    model = model_cpu.to(dev)
    from gripnet_amd.synth import Data
    data = Data(**data_cpu.__dict__).to(dev)

    E_dd = int(data.train_idx.shape[1])
    lo, hi = shard_edge_ranges(E_dd, world)[rank]
    n_d = data.n_d_node
    # entry points whose launch duration is a roofline candidate (the GCN-style one runs once per gene layer)
    CANDIDATES = {"gn_rgcn_forward_f32": "drugs", "gn_distmult_forward_f32": "decode",
                  "gn_distmult_plan_forward_f32": "decode", "gn_graph_aggregate_f32[gcn]": "genes"}

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def per_entry_us(fn, n):
        """us per call of every entry point, HIP events around each launch (eager)."""
        with _hip.KernelTimer() as t:
            for _ in range(n):
                fn()
        return {k: 1e3 * tot / calls for k, (calls, tot) in t.summary().items()}, \
               {k: 1e3 * tot / n for k, (calls, tot) in t.summary().items()}

    with torch.no_grad():
        sharded = None
        if world == 1 and os.environ.get("GN_BENCH_SHARDED_PATH") != "1":   # (the variable rehearses the N > 1 code path on one GPU)
            from gripnet_amd.pipeline import PoseStages
            fence()
            t_plan = time.perf_counter()
            eager = PoseStages(model, data, graphs=False)
            for _ in range(3):                        # builds the plans (the decoder's on the second sighting of its list)
                eager.step()
            fence()
            t_plan = time.perf_counter() - t_plan
            t_steady = time.perf_counter()
            for _ in range(3):
                eager.step()
            fence()
            plan_build_ms = 1e3 * max(0.0, t_plan - (time.perf_counter() - t_steady))
            per_call0, breakdown = per_entry_us(eager.step, 5)
            cold = cold_start_entry(model, data, fence)
            # the two lanes of the two-stream side figure (below the timed region) are made HERE, before any hipGraph is
            # captured: streams created behind captured graphs came to share a hardware queue and the lanes serialised.
            # (A second model with the same parameters: the plans' scratch - the gene layers' tables, the split planes of x -
            # belongs to a model's modules, and two steps in flight must not share it.)
            lanes = None
            if args.launch in ("auto", "recorded"):
                twin = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
                twin.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
                lanes = []
                for m in (model, twin):
                    st = torch.cuda.Stream()
                    st.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(st):
                        lanes.append(PoseStages(m, data, recorded=True))
                fence()
            # side figure: the same step with the dense products on two-term bf16 splits (GN_RGCN_ARITH_FAST /
            # GN_GEMM_ARITH_FAST, flags of the C ABI set per layer): narrower than the reference's fp32, never the headline
            from gripnet_amd.utils import set_arithmetic
            set_arithmetic(model, "fast")
            fast_stages = PoseStages(model, data, graphs=False)
            for _ in range(3):
                fast_stages.step()
            fast_call, fast_breakdown = per_entry_us(fast_stages.step, 5)
            set_arithmetic(model, "fp32")
            for _ in range(2):
                eager.step()
            # the entry point timed inside the timed region: the longest single launch.  (A gene layer is two launches
            # behind one entry point since the LDS-staged path; it is listed in roofline_all with the others.)
            dom = max((k for k in CANDIDATES if k != "gn_graph_aggregate_f32[gcn]"), key=lambda k: per_call0.get(k, 0.0))
            # graphs: every stage but the one that holds the dominant entry point replays as a hipGraph; that entry
            # point is launched from Python in both modes, with HIP events around it on its stream in EVERY timed step.
            # every fourth launch of the dominant entry point is timed: its kernel carries the HIP events as the start / stop
            # stamps of its own dispatch (gn_time_next_launch -> hipExtLaunchKernel: the duration is the kernel's own, 28.4 us
            # where event records around the launch read 30.5) - a stamped dispatch still costs the stream ~5 us (measured:
            # +5.0 us per step with every launch stamped), event records around a launch ~9 us
            timed_every = 4
            def quick(fn, n=20):                      # under the same event timing as the timed region
                with _hip.KernelTimer(only=(dom,), every=timed_every, pool=64):
                    for _ in range(5):
                        fn()
                    fence()
                    t = time.perf_counter()
                    for _ in range(n):
                        fn()
                    fence()
                    return (time.perf_counter() - t) / n
            launch = args.launch
            modes = {"eager": eager}
            if launch in ("auto", "graphs"):
                modes["graphs"] = PoseStages(model, data, graphs=True, timed_entry=dom)
            if launch in ("auto", "recorded"):
                modes["recorded"] = PoseStages(model, data, recorded=True)
            launch_ms = None
            if launch == "auto":
                # best of three short runs each, alternating: one descheduled launch thread must not decide
                best = {k: float("inf") for k in modes}
                for _ in range(3):
                    for k, m in modes.items():
                        best[k] = min(best[k], quick(m.step))
                launch = min(best, key=best.get)
                launch_ms = {k: round(1e3 * v, 5) for k, v in best.items()}
            step = modes[launch].step
        else:
            from gripnet_amd.pipeline import Graphed, Recorded
            from gripnet_amd.sharded import ShardedPoseForward
            fwd = sharded = ShardedPoseForward(model, data, rank, world)
            if os.environ.get("GN_SHARD_EXCHANGE") == "one_shot" and world > 1:
                # (opt-in: the forward's all-reduce as the one-shot direct exchange over peer-mapped buffers, csrc/exchange.hip, instead
                # of RCCL - what SURVEY 8e asks to measure against ncclAllReduce; never the default of a driver run: untested across
                # real GPUs on this pool)
                fwd.use_one_shot_exchange()
            fence()
            t_plan = time.perf_counter()
            for _ in range(3):
                fwd()
            fence()
            plan_build_ms = 1e3 * (time.perf_counter() - t_plan)
            fast_call, fast_breakdown = {}, {}
            cold = None
            per_call0, breakdown = per_entry_us(fwd, 5)
            dom = max((k for k in CANDIDATES if k != "gn_graph_aggregate_f32[gcn]"), key=lambda k: per_call0.get(k, 0.0))
            launch = "eager" if args.launch == "eager" else "graphs" if args.launch == "graphs" else "recorded"
            launch_ms = None
            timed_every = 4
            if launch == "graphs":                    # replicated gene layers as one graph; the collective is never captured
                fwd.kernels.encode_genes = Graphed(fwd.kernels.encode_genes).capture()
            step = fwd
            if launch == "recorded":                  # ... or everything but the collective as recorded entry-point calls
                step = fwd.record()

        # the event pool exists before the warm-up, and the W warm-up steps run under the same timer as the K timed ones
        # (their records are dropped): nothing but the fence sits between the last warm-up step and the first timed one
        # (every fourth launch of the dominant entry point is timed - its kernel carries the events as its dispatch's own stamps
        # where it can, else event records bracket the launch; `timed_launches` says how many went into the average)
        timer = _hip.KernelTimer(only=(dom,), pool=2 * (args.steps + max(args.warmup, 1)) + 8, every=timed_every)
        import gc
        gc.collect()                                  # (a full collection of this process is 40-50 ms: not inside the K steps by accident;
        fence()                                       # the collector stays ON through the timed region)
        # The device's clocks follow its load: behind the host-side work above (plans, recordings, a collection) the first few
        # thousand steps run 6-8 % slower than the steady state (K = 40 steps: 82-83 us each, 60,000 steps: 76.6, same box, same
        # minute - tools/step_only.py).  The steady state is what is quoted: `spin_up_steps` un-timed steps bring the clocks up, then
        # the W warm-up steps and the K timed ones follow without a pause.
        spin_up_steps = int(os.environ.get("GN_BENCH_SPIN_UP", "4000"))
        # (for the record: the same W + K steps WITHOUT the spin-up first - what the line would say with the clocks where the
        # host-side work left them; a side figure in `spread`, never `value`)
        for _ in range(max(args.warmup, 1)):
            step()
        fence()
        t_cold = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        cold_ms = 1e3 * (time.perf_counter() - t_cold) / max(args.steps, 1)
        for _ in range(spin_up_steps):
            step()
        fence()
        with timer:
            for _ in range(max(args.warmup, 1)):
                z, score = step()
            fence()
            timer.events.clear()
            timer._seen.clear()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                z, score = step()
        fence()
        elapsed = time.perf_counter() - t0
        # two more repeats of the same K steps, for the spread only (`value` and `ms_per_step` are the region above)
        repeats = [elapsed]
        for _ in range(2):
            fence()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            repeats.append(time.perf_counter() - t1)
    # side figure (never `value`): INDEPENDENT steps on two streams, each a recording with its own buffers - what the device
    # sustains when forwards do not wait for each other (serving); a step's own latency is `ms_per_step` above
    two_streams = None
    if sharded is None and launch == "recorded" and lanes is not None:
        with torch.no_grad():
            fence()
            for _ in range(3):
                lanes[0].step(); lanes[1].step()
            fence()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                lanes[0].step(); lanes[1].step()
            fence()
            per = (time.perf_counter() - t2) / (2 * args.steps)
            two_streams = {"ms_per_step": round(1e3 * per, 5), "edges_per_s": pose_edges_aggregated(data) / per,
                           "note": "throughput of independent forwards, two in flight (two recordings on two HIP streams); not a "
                                   "step's latency and not the headline"}
            za, zb = lanes[0].step()[0], lanes[1].step()[0]
            fence()
            assert torch.equal(za, zb), "the two lanes disagree"
            del lanes
    # side figure (never `value`): the N > 1 code path (ShardedPoseForward.record(): partial sums of my edge range, the exchange
    # hook, finalisation, my range's scores) rehearsed on this one GPU - what the step costs before any communication
    sharded_n1 = None
    if sharded is None:
        with torch.no_grad():
            from gripnet_amd.sharded import ShardedPoseForward
            fwd1 = ShardedPoseForward(model, data, 0, 1)
            run1 = fwd1.record()
            for _ in range(3):
                run1()
            fence()
            t3 = time.perf_counter()
            for _ in range(args.steps):
                z1, s1 = run1()
            fence()
            per1 = (time.perf_counter() - t3) / args.steps
            sharded_n1 = {"ms_per_step": round(1e3 * per1, 5), "max_abs_diff_z_vs_fused": float((z1 - z).abs().max()),
                          "max_abs_diff_score_vs_fused": float((s1 - score).abs().max()),
                          "note": "gripnet_amd.sharded.ShardedPoseForward.record() at world_size 1: the partial / all-reduce hook / finalise "
                                  "form of the relational layer and the shard's decoder, no communication; what GN_BENCH_SHARDED_PATH=1 runs as the headline"}
            del fwd1, run1
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    _hip.raise_if_index_errors(dev)

    A = pose_edges_aggregated(data)                    # replicated gg/gd counted once, dd over all ranks
    ms_per_step = 1e3 * elapsed / args.steps
    rep_ms = sorted(1e3 * r / args.steps for r in repeats)
    spread = {"ms_per_step_min": round(rep_ms[0], 5), "ms_per_step_median": round(rep_ms[len(rep_ms) // 2], 5),
              "ms_per_step_max": round(rep_ms[-1], 5), "repeats": len(rep_ms),
              "ms_per_step_before_spin_up": round(cold_ms, 5),
              "note": "K steps each, rank-local clocks; the first repeat is the timed region `value` is computed from; "
                      "ms_per_step_before_spin_up: W + K steps timed the same way BEFORE the spin_up_steps un-timed steps (the device's clocks "
                      "where the host-side set-up left them: not the steady state)"}
    value = A * args.steps / elapsed

    # ---- roofline of the dominant entry point (HIP events on its stream, inside the timed region) ----
    alg = algorithmic_bytes(data, hi - lo, E_dd)
    stage_bytes = {"gn_distmult_forward_f32": alg["dmt"], "gn_distmult_plan_forward_f32": alg["dmt"],
                   "gn_rgcn_forward_f32": alg["dd"], "gn_graph_aggregate_f32[gcn]": alg["gg"] // 2,      # per gene layer
                   "gn_graph_aggregate_f32[bipartite]": alg["gd"]}
    calls, total_ms = timer.summary()[dom]
    dom_us = 1e3 * total_ms / calls
    achieved = stage_bytes[dom] / (dom_us * 1e-6) / 1e9
    traffic_all, mfma = {}, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if world == 1 and os.path.exists(tpath):      # PMC-measured HBM bytes of the same launches (profiles/, N = 1)
        try:
            traffic_all = json.load(open(tpath)).get(args.workload, {})
        except Exception:
            traffic_all = {}
    mpath = os.path.join(ROOT, "profiles", "mfma_util.json")
    if world == 1 and os.path.exists(mpath):      # SQ_VALU_MFMA_BUSY_CYCLES pass of the same command (profiles/)
        try:
            mfma = json.load(open(mpath)).get(args.workload)
        except Exception:
            mfma = None
    traffic = traffic_all.get(dom)
    # the same average with event RECORDS around every fourth launch (the method of rounds 1-4: comparable across rounds; a record
    # in front of and behind the launch reads ~2 us more than the kernel's own dispatch stamps), outside the timed region
    records_us = None
    if dom in _hip._STAMPED:
        with torch.no_grad(), _hip.KernelTimer(only=(dom,), pool=2 * args.steps + 8, every=timed_every, records_only=True) as rt:
            for _ in range(args.steps):
                step()
        rc, rtot = rt.summary().get(dom, (0, 0.0))
        records_us = round(1e3 * rtot / rc, 2) if rc else None
    roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": "profiles/traffic.json (PMC passes of tools/prof.sh step r06: warm-up + recorded steps of tools/step_only.py and nothing else, same round; not counters of this run)" if traffic else None,
                "algorithmic_bytes_per_launch": stage_bytes[dom], "avg_launch_us": round(dom_us, 2),
                "timing_method": "dispatch stamps (hipExtLaunchKernel start / stop events of the kernel itself)" if dom in _hip._STAMPED else "event records around the launch",
                "avg_launch_us_event_records": records_us,
                "frac_event_records": None if not records_us else round(stage_bytes[dom] / (records_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "timed_launches": calls,
                "limiter": LIMITERS.get(dom)}

    def line(name, us):
        b = stage_bytes.get(name)
        return {"entry_point": name, "algorithmic_bytes": b, "us_per_call": round(us, 2),
                "achieved_GBs": None if b is None else round(b / (us * 1e-6) / 1e9, 1),
                "frac": None if b is None else round(b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "traffic": traffic_all.get(name)}
    # every entry point of the step, HIP-event timed around each launch in an eager pass before the timed region
    roofline_all = [line(k, v) for k, v in sorted(per_call0.items())]
    roofline_fast = None
    if fast_call.get("gn_rgcn_forward_f32"):
        roofline_fast = line("gn_rgcn_forward_f32", fast_call["gn_rgcn_forward_f32"])
        roofline_fast["note"] = ("arithmetic 'fast' (GN_RGCN_ARITH_FAST): the same destination-major kernel on two-term bf16 splits "
                                 "(three products instead of six, <= 2^-16 per product); the eager step with it takes {:.1f} us of entry "
                                 "points against {:.1f}".format(sum(fast_breakdown.values()), sum(breakdown.values())))

    timed_how = ("every fourth launch, the events carried by the kernel's own dispatch (hipExtLaunchKernel: the kernel's own start / stop stamps)"
                 if dom in _hip._STAMPED else "event records around every fourth launch")
    if sharded is not None:
        launch_note = ("the forward's entry-point calls recorded once and made again from one loop around the all-reduce (which "
                       "torch.distributed issues in every step); {} HIP-event timed, {}".format(dom, timed_how)
                       if launch == "recorded" else
                       "replicated gene layers replayed as {}, the rest eager; {} HIP-event timed, {}".format(
                           "one hipGraph" if launch == "graphs" else "eager launches", dom, timed_how))
    elif launch == "eager":
        launch_note = "eager; {} HIP-event timed, {}".format(dom, timed_how)
    elif launch == "recorded":
        launch_note = ("the step's {} entry-point calls recorded once and made again from one Python loop (ordinary launches on the "
                       "recording's buffers; no hipGraph: a graph launch leaves the GPU idle ~9 us in front of its first kernel on this "
                       "stack); {} HIP-event timed, {}".format(len(modes["recorded"]._whole.calls), dom, timed_how))
    elif dom == "gn_rgcn_forward_f32":
        launch_note = ("gene and external layers replayed as one hipGraph; {} launched and HIP-event timed from Python, the decoder "
                       "(one kernel) launched from Python as well".format(dom))
    else:
        launch_note = "hipGraph replay of every stage but {}, which is launched and HIP-event timed from Python".format(dom)

    result = {
        "metric": "edges aggregated/sec, GripNet forward on pose-0", "value": value, "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spin_up_steps": spin_up_steps, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload if world == 1 else (
                       "{} x{} dd relation shards".format(args.workload, world) if args.scaling == "weak" else
                       "{} (fixed), dd edges cut into {} ranges".format(args.workload, world)),
                   "edges_aggregated_per_step": A, "n_g": data.n_g_node, "n_d": n_d, "relations": data.n_dd_edge_type,
                   "E_gg": int(data.gg_edge_index.shape[1]), "E_gd": int(data.gd_edge_index.shape[1]), "E_dd": E_dd,
                   "decoder": "static positive list: every unordered (node pair, relation) scored once, the score written to both "
                              "directions' positions (same bits as scoring both; parity checked on all E scores)",
                   "parallelism": "single GPU" if world == 1 else "dd edge-range (relation) shards x{} + {}".format(
                       world, "one-shot direct exchange (peer-mapped buffers)" if getattr(sharded, "one_shot", None) is not None else "RCCL all-reduce"),
                   "launch": launch_note},
        "ms_per_step_min": spread["ms_per_step_min"],
        # the same step with nothing kept on the decoder's side (the reference's decoder has no set-up; cold_start has the plans' build times)
        "ms_per_step_decoder_uncached": None if not cold else cold.get("forward_ms_decoder_uncached"),
        "spread": spread,
        "launch_modes_ms_per_step": launch_ms if sharded is None else None,
        "two_streams": two_streams,
        "sharded_path_n1": sharded_n1,
        "cold_start": cold,
        "roofline": roofline,
        "roofline_fast": roofline_fast,
        "roofline_all": roofline_all,
        "mfma_util": mfma,
        "mfma_util_source": "profiles/mfma_util.json (SQ_VALU_MFMA_BUSY_CYCLES pass of tools/prof.sh step r06, same command as the traffic passes; not a counter of this run)" if mfma else None,
        "plan_build_ms": round(plan_build_ms, 1),
        "entry_point_us_per_step": {k: round(v, 2) for k, v in sorted(breakdown.items())},
        "edges_scored_per_sec": (hi - lo) / (per_call0.get("gn_distmult_plan_forward_f32",
                                                           per_call0.get("gn_distmult_forward_f32", float("nan"))) * 1e-6),
    }

    # ---- CPU baseline + parity in the same run (rank 0, N = 1) ---------------------------------
    if world == 1 and not args.no_cpu_baseline:
        from oracle import gripnet_oracle as orc        # checker / baseline only, never the product path
        # 16 threads is the fastest setting for this op mix on the 2x64-core host of the GPU box
        # (measured: 8 -> 1.05 s, 16 -> 0.92 s, 32 -> 0.94 s, 64 -> 1.2 s, 128 -> 3-6 s per forward)
        torch.set_num_threads(min(args.cpu_threads, os.cpu_count() or 1))
        cache = {}
        inputs = (data_cpu.gg_edge_index, data_cpu.edge_weight, data_cpu.gd_edge_index, data_cpu.train_idx,
                  data_cpu.train_et, data_cpu.train_range)
        with torch.no_grad():
            for _ in range(3):                                                   # warm-ups; the first fills the norm cache
                ref = orc.pose_forward(state_cpu, *inputs, gcn_cache=cache)
            times = []
            t_begin = time.perf_counter()
            while len(times) < 5 or (time.perf_counter() - t_begin < args.cpu_seconds and len(times) < 10):
                t1 = time.perf_counter()
                ref = orc.pose_forward(state_cpu, *inputs, gcn_cache=cache)
                times.append(time.perf_counter() - t1)
        times.sort()
        med = times[len(times) // 2]
        err_z = (z.cpu() - ref["z_dd"]).abs().max().item()
        err_s = (score.cpu() - ref["score"]).abs().max().item()
        result["cpu_baseline"] = {
            "value": A / med, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_cpu_count": os.cpu_count(),
            "sample": "{} full {} forwards (A={} edges each) after 3 warm-ups, median; min {:.3f}s median {:.3f}s; {} threads "
                      "(the fastest setting for this op mix on the 2 x 64-core host: 8 -> 1.05 s, 16 -> 0.92 s, 32 -> 0.94 s, "
                      "64 -> 1.2 s, 128 -> 3-6 s per forward)".format(
                len(times), args.workload, A, times[0], med, torch.get_num_threads()),
        }
        result["parity"] = {"max_abs_err_z": err_z, "max_abs_err_score": err_s, "tolerance": 1e-4,
                            "ok": bool(max(err_z, err_s) <= 1e-4)}
        assert max(err_z, err_s) <= 1e-4, "GPU result differs from the CPU oracle: {}".format(result["parity"])

    if world == 1 and not args.no_extra:
        del model, data
        torch.cuda.empty_cache()
        result["extra_workloads"] = extra_workloads(dev, args.extra_seconds, not args.no_cpu_baseline)

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
