#!/usr/bin/env python3
"""Assemble profiles/<tag>_pose0_rocprof.md, profiles/traffic.json and profiles/mfma_util.json from what
`tools/profile_round.sh <tag>`, `tools/mfma_util.sh <tag>` and a plain `python bench.py > gpurun_out/bench_<tag>.json` left
under gpurun_out/ (development tool, round 3)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03a"
rd = lambda p: open(p).read().strip()
stats = rd("gpurun_out/prof_{}_stats.md".format(tag)).split("\n")
traffic = rd("gpurun_out/prof_{}_traffic.md".format(tag))
mfma = rd("gpurun_out/mfma_{}.md".format(tag))
bench = rd("gpurun_out/prof_{}_bench.json".format(tag))
final = rd("gpurun_out/bench_{}.json".format(tag))
wanted = ("k_rgcn", "k_distmult", "k_aggregate", "k_col_", "fillBuffer", "copyBuffer", "radix_sort_onesweep_iteration", "k_indegree",
          "k_acc", "k_pair", "k_degree", "k_fill_csr", "k_gcn_norm")
keep = stats[:2] + [l for l in stats[2:] if any(k in l for k in wanted)][:26]
md = """# Round 3, state {t} - pose0-syn, 1x MI355X (gfx950)

`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --launch eager`
(plan building + per-entry breakdown + the arithmetic-"fast" side pass + warm-up + 20 timed forwards of each per-step kernel; plan
kernels run once.  Kernel durations are those inside the forward, i.e. with nothing of a kernel's inputs left in L2 by its previous launch.)

{stats}

Per step (default, fp32-faithful arithmetic): `k_col_transform<32,1,2>` + `k_col_gather<2>` (gene layer 1), `k_col_transform<16,1,2>` +
`k_col_gather<2>` (gene layer 2), `gn::k_aggregate_transform<16,16>` (external layer), `k_rgcn_pair<3,2,3>` (the relational layer:
ONE launch, no W_r, no slabs, no workspace), `k_distmult_plan`: seven launches.  `k_aggregate_transform_with_weights<16,16>`,
`k_rgcn_acc<48,2,true>` and `k_rgcn_slab_finalize` are the arithmetic-"fast" pass of bench.py (`roofline_fast`: two-term splits, round
2's path); `k_distmult_lds<false>` is the first sighting of the positive list; `k_pair_*` and the radix sorts build plans once.

## HBM traffic per launch (separate PMC passes: FETCH_SIZE, then WRITE_SIZE; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md)

{traffic}

Algorithmic bytes of the same launches (SURVEY.md 8d): gn_rgcn_forward_f32 32.5 MB (the kernel streams ~10.5 MB of 32-bit att-row
offsets since the per-destination K order (16.4 MB before), reads the 123 KB att table and 196 KB of basis per workgroup from L2 and writes 82 KB), gn_distmult[_plan]_forward_f32 56.4 MB,
GCN layer 31.4 MB (a gene layer = one `k_col_transform` + one `k_col_gather`), external layer 5.5 MB.

## MFMA utilisation of the dense steps (one PMC pass: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE; tools/mfma_util.sh)

{mfma}

mfma_util = MFMA busy cycles / (kernel cycles x 256 CUs x 4 SIMDs).  The relational contraction (six bf16 products on three-term
splits: 36 MFMAs per (destination, 32 sources)) keeps the matrix pipe ~5-8 % busy; the kernel is bound by instruction issue of its
gather (DESIGN.md section 4.1), not by the matrix cores.

## bench.py line of the profiled run (slower than an un-profiled run: host-side launch gaps under the profiler)

```
{bench}
```

## bench.py, un-profiled, same build (default arguments, CPU baseline, parity check and extra workloads included)

```
{final}
```
""".format(t=tag, stats="\n".join(keep), traffic=traffic, mfma=mfma, bench=bench, final=final)
open("profiles/{}_pose0_rocprof.md".format(tag), "w").write(md)
t = json.load(open("gpurun_out/traffic_{}.json".format(tag)))
json.dump(t, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
m = json.load(open("gpurun_out/mfma_{}.json".format(tag)))
json.dump(m, open("profiles/mfma_util.json", "w"), indent=1, sort_keys=True)
print(json.dumps(t), json.dumps(m)[:300])
