#!/usr/bin/env python3
"""Assemble profiles/<tag>_pose0_rocprof.md, profiles/<tag>_train_rocprof.md, profiles/traffic.json and
profiles/mfma_util.json from what `tools/profile_round.sh <tag>`, `tools/mfma_util.sh <tag>`, a plain
`python bench.py > gpurun_out/bench_<tag>.json` and `tools/prof_stats.sh <tag>_train tools/bench_train.py ...` left under
gpurun_out/ (development tool, round 2)."""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02b"
rd = lambda p: open(p).read().strip()
stats = rd("gpurun_out/prof_{}_stats.md".format(tag)).split("\n")
traffic = rd("gpurun_out/prof_{}_traffic.md".format(tag))
mfma = rd("gpurun_out/mfma_{}.md".format(tag))
bench = rd("gpurun_out/prof_{}_bench.json".format(tag))
final = rd("gpurun_out/bench_{}.json".format(tag))
wanted = ("k_rgcn", "k_distmult", "k_aggregate", "k_col_", "fillBuffer", "copyBuffer", "radix_sort_onesweep_iteration", "k_indegree",
          "k_acc", "k_degree", "k_fill_csr", "k_gcn_norm")
keep = stats[:2] + [l for l in stats[2:] if any(k in l for k in wanted)][:24]
md = """# Round 2, state {t} - pose0-syn, 1x MI355X (gfx950)

`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --launch eager`
(plan-building + breakdown + exact-mode + mode-selection + warm-up + 20 timed forwards of each per-step kernel; plan kernels run once.
Kernel durations are those inside the forward, i.e. with nothing of a kernel's inputs left in L2 by its previous launch.)

{stats}

Per step: `k_col_transform<32,1,2>` + `k_col_gather<2>` (gene layer 1), `k_col_transform<16,1,2>` + `k_col_gather<2>` (gene layer 2),
`k_aggregate_transform_with_weights<16,16>` (external layer + W_r of the relational layer in one launch), `k_rgcn_acc<48,2,true>`,
`k_rgcn_slab_finalize`, `k_distmult_plan`: eight launches.  `k_rgcn_acc<48,2,false>` is the same kernel with the fp32 matrix
instruction (`GN_ACC_EXACT=1`, bench.py's `roofline_exact` pass); `k_distmult_lds<false>` is the first sighting of the positive list.

## HBM traffic per launch (separate PMC passes: FETCH_SIZE, then WRITE_SIZE; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md)

{traffic}

Algorithmic bytes of the same launches (SURVEY.md 8d): gn_rgcn_forward_f32 32.5 MB, gn_distmult[_plan]_forward_f32 56.4 MB, GCN layer 31.4 MB
(a gene layer = one `k_col_transform` + one `k_col_gather`: 21.2 MB moved, 0.67x algorithmic - 16-bit ids instead of int64 pairs + fp32
coefficients), external layer 5.5 MB (13.7 MB moved: 7.7 MB of it are the W_r fragments the launch also writes).

## MFMA utilisation of the dense steps (one PMC pass: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE; tools/mfma_util.sh)

{mfma}

mfma_util = MFMA busy cycles / (kernel cycles x 256 CUs x 4 SIMDs).  The relational transform is 9 % of the matrix pipe as three
bf16 products on split operands (34 % on the fp32 instruction, which runs at the vector rate); W_r = att . basis 3.7 %.  The path is
HBM / LDS / latency bound: the matrix cores only carry the small dense contractions.

## bench.py line of the profiled run (slower than an un-profiled run: host-side launch gaps under the profiler)

```
{bench}
```

## bench.py, un-profiled, same build (default arguments, CPU baseline and parity check included)

```
{final}
```
""".format(t=tag, stats="\n".join(keep), traffic=traffic, mfma=mfma, bench=bench, final=final)
open("profiles/{}_pose0_rocprof.md".format(tag), "w").write(md)
t = json.load(open("gpurun_out/traffic_{}.json".format(tag)))
json.dump(t, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
m = json.load(open("gpurun_out/mfma_{}.json".format(tag)))
json.dump(m, open("profiles/mfma_util.json", "w"), indent=1, sort_keys=True)
tp = "gpurun_out/prof_{}_train".format(tag)
if os.path.isdir(tp):
    import subprocess
    out = subprocess.run([sys.executable, "tools/summarize_prof.py", tp], capture_output=True, text=True).stdout.strip().split("\n")
    open("profiles/{}_train_rocprof.md".format(tag), "w").write("""# Round 2, state {t} - one training step of the PoSE model on pose0-syn, 1x MI355X

`rocprofv3 --kernel-trace --stats -- python3 tools/bench_train.py --steps 20 --fused-adam` (3 warm-up + 20 timed eager steps: forward,
positive and negative decoder calls, loss, backward, Adam; new negative pairs every step).  Un-profiled: 1.21 ms per eager step,
1.05 ms as one hipGraph replay (`--graph --fused-adam`).

{rows}
""".format(t=tag, rows="\n".join(out[:42])))
print(json.dumps(t), json.dumps(m)[:200])
