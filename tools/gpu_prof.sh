#!/bin/bash
# dev helper: micro-bench + PMC counters for one entry point.  usage: WHAT=distmult WL=pose0-syn ./tools/gpu_prof.sh
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WHAT=${WHAT:-distmult}; WL=${WL:-pose0-syn}
python tools/bench_kernels.py --workload $WL --what $WHAT,full --iters 50 2>&1 | grep -v amdgpu.ids
python tools/bench_kernels.py --workload pose2-syn --what $WHAT --iters 20 2>&1 | grep -v amdgpu.ids
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc1 -- python3 tools/bench_kernels.py --workload $WL --what $WHAT --iters 3 > gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc2 -- python3 tools/bench_kernels.py --workload $WL --what $WHAT --iters 3 > gpurun_out/pmc2.log 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc1","gpurun_out/pmc2"):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][-40:]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            if "distmult" in k or "rgcn" in k or "aggregate" in k or "gemm" in k:
                print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, "n=",len(next(iter(v.values()))))
PY
