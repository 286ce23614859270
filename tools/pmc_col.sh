cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_col
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_col -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --launch eager > gpurun_out/pmc_col.log 2>&1
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmc_col/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][-40:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if "k_col" in k or "rgcn_acc" in k or "distmult_plan" in k:
            print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, "n=",len(next(iter(v.values()))))
PY
