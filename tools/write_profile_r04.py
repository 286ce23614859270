#!/usr/bin/env python3
"""Assemble profiles/r04_pose0_rocprof.md, profiles/r04_counters.md, profiles/traffic.json and profiles/mfma_util.json from
what `tools/profile_r04.sh r04` left under gpurun_out/ (development tool, round 4)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
rd = lambda p: open(p).read().strip()
stats = rd("gpurun_out/prof_{}_stats.md".format(tag)).split("\n")
traffic = rd("gpurun_out/prof_{}_traffic.md".format(tag))
mfma = rd("gpurun_out/mfma_{}.md".format(tag))
bench = rd("gpurun_out/prof_{}_bench.json".format(tag))
final = rd("gpurun_out/bench_{}.json".format(tag))
wanted = ("k_rgcn", "k_distmult", "k_aggregate", "k_col_", "fillBuffer", "copyBuffer", "radix_sort_onesweep_iteration", "k_indegree",
          "k_pair", "k_degree", "k_fill_csr", "k_gcn_norm", "k_split")
keep = stats[:2] + [l for l in stats[2:] if any(k in l for k in wanted)][:24]
md = """# Round 4 - pose0-syn, 1x MI355X (gfx950)

`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --launch recorded`
(plan building + per-entry breakdown + the arithmetic-"fast" side pass + warm-up + 3 x 20 forwards of each per-step kernel; plan kernels
run once.  Kernel durations are those inside the forward, i.e. with nothing of a kernel's inputs left in L2 by its previous launch.)

{stats}

Per step (default, fp32-faithful arithmetic), seven launches: `k_col_transform<32,1,2>` + `k_col_gather<2>` (gene layer 1),
`k_col_transform<16,1,2>` + `k_col_gather<2>` (gene layer 2), `gn::k_aggregate_transform<16,16>` (external layer; also leaves x as bf16
split planes), `k_rgcn_pair<3,2,3,true>` (the relational layer on those planes: one launch, no W_r, no workspace), `k_distmult_class<5,3>`
(decoder on the static positive list: row classes, one table fill).  `k_rgcn_pair<3,2,2,true>` is the arithmetic-"fast" pass of bench.py
(`roofline_fast`: the same kernel on two-term splits); `k_distmult_lds<false>` is the first sighting of the positive list; `k_pair_*`
and the radix sorts build plans once.

## HBM traffic per launch (separate PMC passes: FETCH_SIZE, then WRITE_SIZE; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md)

{traffic}

Algorithmic bytes of the same launches (SURVEY.md 8d): gn_rgcn_forward_f32 32.5 MB, gn_distmult_plan_forward_f32 56.4 MB, GCN layer 31.4 MB
(a gene layer = one `k_col_transform` + one `k_col_gather`), external layer 5.5 MB.

## MFMA utilisation of the dense steps (one PMC pass: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE; tools/mfma_util.sh)

{mfma}

mfma_util = MFMA busy cycles / (kernel cycles x 256 CUs x 4 SIMDs).

## bench.py line of the profiled run (slower than an un-profiled run: host-side launch gaps under the profiler)

```
{bench}
```

## bench.py, un-profiled, same build, the driver's arguments (`--steps 20 --warmup 5`; CPU baseline, parity check and extra workloads included)

```
{final}
```
""".format(stats="\n".join(keep), traffic=traffic, mfma=mfma, bench=bench, final=final)
open("profiles/{}_pose0_rocprof.md".format(tag), "w").write(md)

pick = lambda path, words: "\n".join(l for l in open(path).read().split("\n") if any(w in l for w in words))
kern = ("k_rgcn_pair<3, 2, 3, true>", "k_distmult_class<5, 3>", "k_col_gather<2>", "k_aggregate_transform<16, 16>", "k_col_transform")
counters = """# Round 4 - counters, in-kernel stamps and launch modes (pose0-syn, 1x MI355X)

Counter passes: `tools/pmc.sh <tag> "<counters>" tools/bench_kernels.py --what full --iters 10` (rocprofv3 --kernel-trace --pmc, one pass per
counter set, averages per launch over the chip).  SQ_* cycle counters are in units of four clocks, summed over the SIMDs.

## Wave occupancy of the issue slots (SQ_WAVE_CYCLES = waves resident; WAIT_ANY = at an s_waitcnt; WAIT_INST_ANY = waiting to be picked)

```
{sq}
```

## LDS and instruction mix (SQ_LDS_IDX_ACTIVE = LDS-array busy cycles, SQ_LDS_BANK_CONFLICT = the conflict part of them)

```
{lds}
```

## In-kernel stamps (make STAMPS=1; 100 MHz wall clock inside the kernel)

Relational kernel (`tools/pair_stamps.py --planes`):
```
{pair}
```
Decoder (`tools/dm_stamps.py`):
```
{dm}
```
Gene gather (`tools/blk_stamps.py`):
```
{blk}
```

## Launch modes of the step, wall clock without event timing (`tools/launch_modes.py`)

```
{modes}
```

## Arithmetic "fast" on the destination-major kernel (`tools/bench_kernels.py --what rgcn --arith fast --rgcn-kernel pair`)

```
{fast}
```
""".format(sq=pick("gpurun_out/pmc_{}_sq.txt".format(tag), kern), lds=pick("gpurun_out/pmc_{}_lds.txt".format(tag), kern),
           pair=rd("gpurun_out/stamps_{}_pair.txt".format(tag)), dm=rd("gpurun_out/stamps_{}_dm.txt".format(tag)),
           blk=rd("gpurun_out/stamps_{}_blk.txt".format(tag)), modes=pick("gpurun_out/launch_modes_{}.txt".format(tag), ("us",)),
           fast=pick("gpurun_out/pair_fast_{}.txt".format(tag), ("rgcn",)))
open("profiles/{}_counters.md".format(tag), "w").write(counters)

# ---- the training step ----
import os
def opt(path):
    return rd(path) if os.path.exists(path) else "(not collected)"
train = """# Round 4 - the PoSE training step (pose0-syn, 1x MI355X): forward, both decoder calls, loss, backward, Adam

`tools/train_step.py` = the `train_step_entry` of bench.py: ONE hipGraph replay per step, the negatives drawn on the device in front
of every replay.  Un-profiled:

```
{step}
```

## Kernel stats (`rocprofv3 --kernel-trace --stats -- python3 tools/train_step.py 20`: 3 eager warm-up steps, the capture, 23 replays)

{stats}

## Timeline of the last replayed steps (`tools/timeline.py`; inside a replay a kernel node costs >= ~4.7 us, whatever it does)

```
{timeline}
```

## The fused relational weight gradient `k_rel_weight_grad<3,2>` (gn_rel_weight_grad_f32)

Against the path it replaces (`tools/relgrad_probe.py`):
```
{probe}
```
In-kernel stamps (`tools/rel_stamps.py`, make STAMPS=1):
```
{stamps}
```
Counters (`tools/pmc_relgrad.sh`, one pass per set; SQ_* cycle counters in units of four clocks, summed over the SIMDs):
```
{pmc}
```

## Do VALU instructions issue in the shadow of fp32 MFMAs?  (`tools/probes/mfma_valu_probe.hip`)

```
{coissue}
```
They do not: with four waves per SIMD the cycles per trip are the SUM of the matrix pipe's (M x 32) and the vector instructions'
(V x ~5), not their maximum.  The weight-gradient kernel is therefore bound by MFMA + VALU cycles together (DESIGN.md section 7).

## The rate of LDS atomics (`tools/probes/lds_atomic_probe.hip`): why the decoder's backward sorts instead of scattering

```
{atomics}
```
`ds_add_f32` retires about one lane per three clocks per CU whatever the addresses (193 clocks per wave instruction); `ds_add_u32`
and `ds_add_u64` run at the rate of plain LDS writes.

## The negative sampler's two kernels (`tools/probes/sampler_probe.py`)

```
{sampler}
```
""".format(step=opt("gpurun_out/train_step_{}.txt".format(tag)).split("\n")[-1],
           stats=opt("gpurun_out/train_stats_{}.md".format(tag)),
           timeline=opt("gpurun_out/train_timeline_{}.txt".format(tag)),
           probe="\n".join(l for l in opt("gpurun_out/relgrad_{}.txt".format(tag)).split("\n") if "us" in l or "max" in l),
           stamps=opt("gpurun_out/stamps_{}_rel.txt".format(tag)),
           pmc="\n".join(l for l in opt("gpurun_out/pmc_{}_relgrad.txt".format(tag)).split("\n") if "rel_weight" in l),
           coissue=opt("gpurun_out/mfma_valu_probe_{}.txt".format(tag)),
           atomics=opt("gpurun_out/lds_atomic_probe_{}.txt".format(tag)),
           sampler="\n".join(l for l in opt("gpurun_out/sampler_probe_{}.txt".format(tag)).split("\n") if "amdgpu.ids" not in l))
open("profiles/{}_train_rocprof.md".format(tag), "w").write(train)
t = json.load(open("gpurun_out/traffic_{}.json".format(tag)))
json.dump(t, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
m = json.load(open("gpurun_out/mfma_{}.json".format(tag)))
json.dump(m, open("profiles/mfma_util.json", "w"), indent=1, sort_keys=True)
print(json.dumps(t), json.dumps(m)[:300])
