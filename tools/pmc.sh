#!/bin/bash
# dev helper: PMC counters for a python command, per kernel averages.  usage: tools/pmc.sh <tag> "<counters>" <python args...>
tag=$1; ctrs=$2; shift; shift
mkdir -p gpurun_out; root=$PWD
cd /tmp && export TMPDIR=/tmp && cd $root
rm -rf gpurun_out/pmc_$tag
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag -- python3 "$@" > gpurun_out/pmc_$tag.log 2>&1
echo "rocprof rc=$?"
python3 - <<PY
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_$tag/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][-44:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if any(s in k for s in ("distmult","rgcn","aggregate","gemm","merge","rel_weight","seg_lds","dense_batch","he_s","place_g","sample_neg")):
            print(k, {c: round(sum(x)/len(x)) for c,x in sorted(v.items())}, "n=",len(next(iter(v.values()))))
PY
