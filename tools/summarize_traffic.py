#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    python tools/summarize_traffic.py <fetch_dir> <write_dir> [--json profiles/traffic.json --workload pose0-syn]

Corrections of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on
gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read, so it is doubled; WRITE_SIZE is
exact for 16-byte-per-lane stores.  Prints bytes per launch for every kernel of the hot path and the sum
per C-ABI entry point.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re

ENTRY = {
    "gn_rgcn_forward_f32": ("k_rgcn_pair",),
    "gn_distmult_forward_f32": ("k_distmult_lds", "k_distmult<"),
    "gn_distmult_plan_forward_f32": ("k_distmult_class", "k_distmult_plan"),
    "gn_graph_aggregate_f32": ("k_aggregate", "k_col_"),
    "gn_gemm_f32": ("k_gemm_f32",),
}
# entry points that are several kernels with their own weights per call: (kernel name prefix, launches per call)
WEIGHTED = {
    # one call per gene layer: its transform (two instantiations, one per layer) and its gather (same kernel, both layers)
    "gn_graph_aggregate_f32[gcn]": (("k_col_transform<32", 0.5), ("k_col_transform<16", 0.5), ("k_col_gather", 1.0)),
    "gn_graph_aggregate_f32[bipartite]": (("gn::k_aggregate_transform<16, 16>", 1.0), ("k_aggregate_transform<16, 16>", 1.0)),
}
# (the default step runs k_rgcn_pair<3, 2, 3, true>; the two-term pass of bench.py - roofline_fast - is the <3, 2, 2, true> instantiation)
FAST_ONLY = ("k_rgcn_pair<3, 2, 2",)


def per_kernel(root, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0]
            out[name].append(float(r["Counter_Value"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--json")
    ap.add_argument("--workload", default="pose0-syn")
    args = ap.parse_args()
    fetch, write = per_kernel(args.fetch_dir, "FETCH_SIZE"), per_kernel(args.write_dir, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(fetch) | set(write)):
        if not any(k.startswith(p) or p in k for ps in ENTRY.values() for p in ps):
            continue
        f = fetch.get(k, [0.0])
        w = write.get(k, [0.0])
        rows[k] = (len(f), 2.0 * 1024 * sum(f) / len(f), 1024 * sum(w) / len(w))
    print("| kernel | launches | HBM read B/launch (2 x FETCH_SIZE) | HBM write B/launch | total MB |")
    print("|---|---:|---:|---:|---:|")
    for k, (n, rd, wr) in rows.items():
        print("| `{}` | {} | {:.0f} | {:.0f} | {:.2f} |".format(k[:60], n, rd, wr, (rd + wr) / 1e6))
    sums = {}
    for entry, pats in ENTRY.items():
        tot = sum(rd + wr for k, (n, rd, wr) in rows.items() if any(p in k for p in pats) and not any(f in k for f in FAST_ONLY))
        if entry == "gn_graph_aggregate_f32" or entry == "gn_gemm_f32":
            continue            # several launches of different sizes per step: per-kernel rows above are averages
        sums[entry] = tot
        print("entry point {}: {:.2f} MB per launch (sum of its kernels)".format(entry, tot / 1e6))
    for entry, parts in WEIGHTED.items():
        tot, seen = 0.0, False
        for prefix, weight in parts:
            for k, (n, rd, wr) in rows.items():
                if k.startswith(prefix):
                    tot += weight * (rd + wr)
                    seen = True
        if seen:
            sums[entry] = tot
            print("entry point {}: {:.2f} MB per call (weighted sum of its kernels)".format(entry, tot / 1e6))
    if args.json:
        data = {}
        if os.path.exists(args.json):
            data = json.load(open(args.json))
        data[args.workload] = {k: round(v) for k, v in sums.items()}
        json.dump(data, open(args.json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
