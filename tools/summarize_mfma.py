#!/usr/bin/env python3
"""MFMA utilisation per kernel from one rocprofv3 PMC pass (tools/prof.sh step).

    python tools/summarize_mfma.py <pmc_dir> [--json profiles/mfma_util.json --workload pose0-syn]

SQ_VALU_MFMA_BUSY_CYCLES counts, summed over the chip, the cycles in which a SIMD's matrix pipe is busy (MI355X_MICROARCH.md:
32 per v_mfma_f32_32x32x16_bf16).  GRBM_GUI_ACTIVE is reported as the sum over the 8 XCDs, so the kernel's
duration in shader cycles is GRBM_GUI_ACTIVE / 8 and

    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)

(the gfx94x MfmaUtil formula; ROCm 7.2 ships no gfx950 derived-counter section).
"""
import argparse
import collections
import csv
import glob
import json
import os
import re

WANT = ("k_rgcn_pair", "k_gemm", "k_col_transform", "k_rgcn_lds")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pmc_dir")
    ap.add_argument("--json")
    ap.add_argument("--workload", default="pose0-syn")
    args = ap.parse_args()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(args.pmc_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    print("| kernel | launches | MFMA busy cycles | MFMA instructions | GRBM_GUI_ACTIVE / 8 | mfma_util |")
    print("|---|---:|---:|---:|---:|---:|")
    for k, v in sorted(agg.items()):
        if not any(w in k for w in WANT):
            continue
        mean = {c: sum(x) / len(x) for c, x in v.items()}
        busy, gui = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), mean.get("GRBM_GUI_ACTIVE", 0.0) / 8
        util = busy / (gui * 256 * 4) if gui else 0.0
        out[k] = {"mfma_busy_cycles": round(busy), "mfma_instructions": round(mean.get("SQ_INSTS_MFMA", 0.0)),
                  "kernel_cycles": round(gui), "mfma_util": round(util, 4)}
        print("| `{}` | {} | {:.0f} | {:.0f} | {:.0f} | {:.4f} |".format(k[:60], len(next(iter(v.values()))), busy,
                                                                        mean.get("SQ_INSTS_MFMA", 0.0), gui, util))
    if args.json:
        data = json.load(open(args.json)) if os.path.exists(args.json) else {}
        data[args.workload] = out
        json.dump(data, open(args.json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
