#!/usr/bin/env python3
"""profiles/r03b_nc_rocprof.md from the rocprofv3 summaries tools/profile_secondary.sh leaves under gpurun_out/."""
import os, re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")


def head(path, n):
    with open(path) as f:
        return "".join(f.readlines()[:n])


def last_line(path, key):
    with open(path, errors="replace") as f:
        hits = [l.strip() for l in f if key in l and "rocprofv3" not in l]
    return hits[-1] if hits else "(missing)"


out = ["# Round 3, state r03b - the secondary workloads on 1x MI355X (gfx950)\n",
       "`rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/bench_nc.py [--model freebase-c] --iters 20` and\n"
       "`... -- python3 tools/bench_train.py --steps 20` (tools/profile_secondary.sh; 3 warm-up + 20 timed iterations each, plan kernels\n"
       "run once; times under the profiler are longer than un-profiled ones: host-side launch gaps).\n",
       "\n## aminer-style model on aminer-syn (BASELINE config 3): forward\n\n", last_line(os.path.join(G, "prof_nc.log"), "forward on") + "\n\n",
       head(os.path.join(G, "prof_nc_stats.md"), 16),
       "\n## freebase-c-style model on aminer-syn scale (BASELINE config 5, fp32 storage): forward\n\n", last_line(os.path.join(G, "prof_fb.log"), "forward on") + "\n\n",
       head(os.path.join(G, "prof_fb_stats.md"), 16),
       "\n## PoSE training step on pose0-syn (forward, two decoder calls, backward, Adam; new negatives drawn on the device every step)\n\n",
       last_line(os.path.join(G, "prof_tr.log"), "training step") + "\n\n",
       head(os.path.join(G, "prof_tr_stats.md"), 40)]
with open(os.path.join(ROOT, "profiles", "r03b_nc_rocprof.md"), "w") as f:
    f.write("".join(out))
print("written")
