"""Round 6, verdict item 1: the relational layer's x-independent half (the pair sums) on a side stream under the gene layers.
Same-box A/B of the recorded pose0-syn step, one-launch layer against the two-launch form; the three launches' own durations."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gripnet_amd import _hip
from gripnet_amd.pipeline import PoseModel, PoseStages
from gripnet_amd.synth import make_pose

workload = sys.argv[1] if len(sys.argv) > 1 else "pose0-syn"
dev = torch.device("cuda:0")
data = make_pose(workload).to(dev)
torch.manual_seed(1111)
model = PoseModel(data.n_g_node, data.n_d_node, data.n_dd_edge_type).to(dev)
out = {"workload": workload}
with torch.no_grad():
    fused = PoseStages(model, data, recorded=True)
    z0, s0 = [t.clone() for t in fused.step()]
    model.split_relational = True
    split = PoseStages(model, data, recorded=True)
    model.split_relational = False
    z1, s1 = split.step()
    torch.cuda.synchronize()
    out["same_bits"] = bool(torch.equal(z0, z1) and torch.equal(s0, s1))
    out["calls_split"] = [c[3] or c[2] for c in split._whole.calls]

    def run(stages, n=200):
        for _ in range(20):
            stages.step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            stages.step()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t) / n

    res = {"fused": [], "split": []}
    for _ in range(5):
        res["fused"].append(round(run(fused), 5))
        res["split"].append(round(run(split), 5))
    out["ms_per_step"] = res
    out["ms_per_step_min"] = {k: min(v) for k, v in res.items()}
    # the launches' own durations (event records around each, so +~2 us each; the step is serialised by the records)
    for name, st in (("fused", fused), ("split", split)):
        with _hip.KernelTimer(pool=400, records_only=True) as kt:
            for _ in range(10):
                st.step()
        out["us_per_call_" + name] = {k: round(1e3 * tot / calls, 2) for k, (calls, tot) in kt.summary().items()}
    # ---- the three launches on their own: back-to-back on ONE stream, events around 20 of them ----
    conv = model.dd.conv_list[0]
    plan = conv._plan
    x = fused.x
    x = x.clone()
    planes = _hip.SplitPlanes(x.shape[0], conv.in_channels // 16, dev).fill_from(x)
    planes.tag(x)
    zbuf = torch.empty((data.n_d_node, 80), device=dev)
    cur = torch.cuda.current_stream()

    def evtime(fn, n=20):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return round(1e3 * a.elapsed_time(b) / n, 2)

    args = (x, conv.basis, conv.att, conv.root, conv.bias, True, zbuf[:, 48:])
    out["us_alone"] = {
        "one launch": evtime(lambda: plan.forward(*args, x_planes=planes)),
        "pair sums": evtime(lambda: plan.start_pair_sums(conv.att.detach(), 48, 32, 32, cur)),
        "contraction": evtime(lambda: plan.forward(*args, x_planes=planes, pair_sums=True)),
    }
    out["pair_sums_bytes"] = int(plan._sums.numel())
    # ---- the gene + external chain alone, and with the pair sums running beside it on the side stream ----
    side = _hip.side_stream(dev)

    def genes():
        fused._genes_eager()

    def genes_with_sums():
        plan.start_pair_sums(conv.att.detach(), 48, 32, 32, side)
        fused._genes_eager()
        _hip._call("gn_stream_order", side.cuda_stream, _hip.stream_ptr(dev))

    def wall(fn, n=200):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return round(1e6 * (time.perf_counter() - t) / n, 2)

    out["us_gene_chain"] = {"alone": wall(genes), "with the pair sums beside it (joined at its end)": wall(genes_with_sums), "alone again": wall(genes)}
print(json.dumps(out, indent=1))
