#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` into a short table (kernel names shortened).

    python tools/summarize_prof.py gpurun_out/prof_dir > profiles/<name>.md
"""
import csv
import glob
import os
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    m = re.match(r"(void )?([A-Za-z0-9_:<>, ]+?)\(", name)
    base = (m.group(2) if m else name).strip()
    if "rocprim::" in name:
        kinds = re.findall(r"rocprim::(radix_sort_[a-z_]+|scan_impl|transform_impl|merge_sort_[a-z_]+|init_lookback[a-z_]+)", name)
        base = "rocprim::" + (kinds[-1] if kinds else "kernel")
    return base[:70]


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_stats.csv under " + root)
    rows = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            c, t = int(r["Calls"]), float(r["TotalDurationNs"])
            mn, mx = float(r["MinNs"]), float(r["MaxNs"])
            if k in rows:
                rows[k] = (rows[k][0] + c, rows[k][1] + t, min(rows[k][2], mn), max(rows[k][3], mx))
            else:
                rows[k] = (c, t, mn, mx)
    total = sum(v[1] for v in rows.values())
    print("| kernel | calls | total us | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for k, (c, t, mn, mx) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("| `{}` | {} | {:.1f} | {:.2f} | {:.2f} | {:.2f} | {:.1f} |".format(k, c, t / 1e3, t / c / 1e3, mn / 1e3, mx / 1e3, 100 * t / total))


if __name__ == "__main__":
    main()
