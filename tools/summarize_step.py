#!/usr/bin/env python3
"""The per-step kernels of a TIMED REGION from a rocprofv3 kernel trace (tools/prof.sh step | train).

    python tools/summarize_step.py gpurun_out/step_<tag> --steps K [--log <stdout of the run>]

The traced program ends with K back-to-back steps (tools/step_only.py, tools/train_step.py): the trace's last dispatches are
those steps.  The step's launch sequence is found as the shortest period of kernel names at the end of the trace; the last K
periods are the timed region.  Printed: every launch of the step with its average / min / max duration over the K steps, the
SUM of the averages, the GPU-side span per step (first start to last end of the region / K) and the wall time the program
itself measured - the sum must not exceed the span by more than the overlap of consecutive launches' ramp and drain.
"""
import argparse
import csv
import glob
import json
import os
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:64]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--log")
    args = ap.parse_args()
    rows = []
    for f in glob.glob(os.path.join(args.root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    names = [r[2] for r in rows]
    K = args.steps
    period, drop = None, 0
    for drop in range(0, 9):                                # (what follows the last step: the error word's read-back copy)
        tail_names = names[:len(names) - drop] if drop else names
        for p in range(1, len(tail_names) // max(K, 1) + 1):
            tail = tail_names[-p * K:]
            if len(tail) == p * K and all(tail[i] == tail[i % p] for i in range(p * K)):
                period = p
                break
        if period is not None:
            break
    if period is None:
        raise SystemExit("no periodic tail of {} steps in {} dispatches".format(K, len(names)))
    if drop:
        rows = rows[:len(rows) - drop]
    region = rows[-period * K:]
    print("## Timed region only: the last {} steps x {} launches of `{}`\n".format(K, period, os.path.basename(args.root.rstrip("/"))))
    print("| # | kernel | avg us | min us | max us | gap in front, avg us |")
    print("|---:|---|---:|---:|---:|---:|")
    total = 0.0
    for i in range(period):
        d = [(region[s * period + i][1] - region[s * period + i][0]) / 1e3 for s in range(K)]
        gaps = []
        for s in range(K):
            j = s * period + i
            if j > 0:
                gaps.append((region[j][0] - region[j - 1][1]) / 1e3)
        total += sum(d) / K
        print("| {} | `{}` | {:.2f} | {:.2f} | {:.2f} | {:.2f} |".format(i, region[i][2], sum(d) / K, min(d), max(d),
                                                                         sum(gaps) / max(1, len(gaps))))
    span = (region[-1][1] - region[0][0]) / 1e3 / K
    wall = None
    if args.log and os.path.exists(args.log):
        for line in open(args.log, errors="replace"):
            if line.startswith("STEP_ONLY "):
                wall = json.loads(line[len("STEP_ONLY "):])["us_per_step_wall"]
    print("\nsum of the per-step kernels' average durations: **{:.2f} us**; GPU-side span of the region per step (first start to "
          "last end / {}): **{:.2f} us**{}".format(total, K, span, "" if wall is None else
                                                   "; wall time per step measured by the program (under the profiler): **{:.2f} us**".format(wall)))
    print("sum check: sum {} span ({:+.2f} us: negative = launches overlap their neighbours' ramp / drain, positive = idle "
          "gaps between launches)\n".format("<=" if total <= span else ">", span - total))


if __name__ == "__main__":
    main()
