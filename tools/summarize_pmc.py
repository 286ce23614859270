#!/usr/bin/env python3
"""Per-kernel averages of one rocprofv3 --pmc pass (the kernels of the hot path only).

    python tools/summarize_pmc.py gpurun_out/pmc_<tag> [--all]
"""
import collections
import csv
import glob
import os
import re
import sys

HOT = ("distmult", "rgcn", "aggregate", "gemm", "merge", "rel_weight", "seg_lds", "dense_batch", "he_s", "place_g", "sample_neg",
       "col_gather", "col_transform", "grad_prologue", "link_loss", "adam", "xtg", "pair_grad", "rel_basis", "node_sums", "class_scores")


def main():
    root = sys.argv[1]
    everything = "--all" in sys.argv
    print("\n### counters of `{}`\n".format(os.path.basename(root.rstrip("/"))))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"^void ", "", k).split("(")[0][-56:]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            if everything or any(s in k for s in HOT):
                print("- `{}` n={}: {}".format(k, len(next(iter(v.values()))),
                                               ", ".join("{} {:.0f}".format(c, sum(x) / len(x)) for c, x in sorted(v.items()))))


if __name__ == "__main__":
    main()
