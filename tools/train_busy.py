#!/usr/bin/env python3
"""dev helper: device-busy share of the training steps in a rocprofv3 kernel trace of tools/bench_train.py.
    tools/prof.sh stats train tools/bench_train.py --steps 10 && python tools/train_busy.py gpurun_out/prof_train"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
# steady state = the last 60 % of the launches
sel = rows[int(len(rows) * 0.4):]
busy = sum(e - s for s, e, _ in sel)
span = sel[-1][1] - sel[0][0]
print("steady-state window {:.2f} ms, kernels {:d}, device busy {:.2f} ms ({:.0f} %)".format(span / 1e6, len(sel), busy / 1e6, 100.0 * busy / span))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in sel:
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    agg[k][0] += e - s; agg[k][1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("{:72s} {:6d} {:9.1f} us  {:5.1f} %".format(k, c, t / 1e3, 100.0 * t / busy))
