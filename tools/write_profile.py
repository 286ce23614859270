#!/usr/bin/env python3
"""Assemble profiles/<tag>_pose0_rocprof.md and profiles/traffic.json from what tools/profile_round.sh <tag> and a plain
`python bench.py > gpurun_out/bench_<tag>.json` left under gpurun_out/ (development tool)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01d"
stats = open("gpurun_out/prof_{}_stats.md".format(tag)).read().strip().split("\n")
traffic = open("gpurun_out/prof_{}_traffic.md".format(tag)).read().strip()
bench = open("gpurun_out/prof_{}_bench.json".format(tag)).read().strip()
final = open("gpurun_out/bench_{}.json".format(tag)).read().strip()
wanted = ("k_rgcn", "k_distmult", "k_aggregate", "fillBuffer", "copyBuffer", "radix_sort_onesweep_iteration", "k_indegree",
          "k_acc", "k_degree", "k_fill_csr", "k_gcn_norm")
keep = stats[:2] + [l for l in stats[2:] if any(k in l for k in wanted)][:22]
notes = open("profiles/{}_notes.md".format(tag)).read() if len(sys.argv) > 2 else ""
md = """# Round 1, state {tag_short} — pose0-syn, 1x MI355X (gfx950)

`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --launch eager`
(3 plan-building + 5 breakdown + 25 mode-selection + 3 warm-up + 20 timed forwards of each per-step kernel; plan kernels run once.
Kernel durations are those inside the forward, i.e. with nothing of a kernel's inputs left in L2 by its previous launch;
`tools/bench_kernels.py --flush 64` reproduces that for a single entry point, without `--flush` the decoder reads 29 us.)

{stats}

Per step: `k_aggregate_transform<8,16>` (gene layer 1), `k_aggregate_transform_q<1>` (gene layer 2),
`k_aggregate_transform_with_weights<16,16>` (external layer + W_r of the relational layer in one launch),
`k_rgcn_acc`, `k_rgcn_slab_finalize`, `k_distmult_plan`: six launches.

## HBM traffic per launch (separate PMC passes: FETCH_SIZE, then WRITE_SIZE; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md)

{traffic}

Algorithmic bytes of the same launches (SURVEY.md 8d): gn_rgcn_forward_f32 32.5 MB, gn_distmult[_plan]_forward_f32 56.4 MB, GCN layer 31.4 MB.
The relational kernel moves 31 MB: 9.4 MB of edge stream, the 7.7 MB of split W_r fragments (written by the external layer's
launch) once or twice - the plan keeps a relation's units on the slabs of one XCD, so its fragments go into one L2 instead of
eight (72 MB before that) -, the node table of 256 workgroups where it misses the L2s, 1.5 MB of slabs.
The planned decoder moves {dm_plan:.0f} MB (three 32-bit words per scored edge - half of the bidirectional list -, the scores
once per position; the partial sums wait in LDS between the phases); the plan-less kernel, which negative samples and the first sighting of a list still take, 127-155 MB.

## bench.py line of the profiled run (slower than an un-profiled run: host-side launch gaps under the profiler)

```
{bench}
```

## bench.py, un-profiled, same build (default arguments, CPU baseline and parity check included)

```
{final}
```
""".format(tag_short=tag[-1], stats="\n".join(keep), traffic=traffic, bench=bench, final=final,
           dm_plan=json.load(open("gpurun_out/traffic_{}.json".format(tag)))["pose0-syn"].get("gn_distmult_plan_forward_f32", 0) / 1e6)
open("profiles/{}_pose0_rocprof.md".format(tag), "w").write(md)
t = json.load(open("gpurun_out/traffic_{}.json".format(tag)))
json.dump(t, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
print(json.dumps(t))
