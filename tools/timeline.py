#!/usr/bin/env python3
"""Device timeline of the last steps of a rocprofv3 kernel trace (development tool).

    tools/prof.sh stats tl bench.py --steps 20 --warmup 3 --no-cpu-baseline [--graphs]
    python tools/timeline.py gpurun_out/prof_tl [n_kernels]

Prints start offset, duration and the idle gap before every kernel, so that launch gaps and overlap
between streams are visible."""
import csv
import glob
import os
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]          # (the newest trace under the directory)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    print("{:9.1f} us  dur {:7.2f}  gap {:7.2f}  q{} {}".format((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3,
                                                              r.get("Queue_Id", "?"), name))
    prev_end = max(prev_end, e)
