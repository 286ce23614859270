#!/bin/bash
# Round profile on the GPU box: kernel stats of the default bench command + HBM traffic counters.
#   tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/, gpurun_out/pmc_<tag>_{fetch,write}/
# Counters are collected in their own passes (kernel-trace only), as the MI355X guide prescribes.
tag=${1:-r01}
root=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $root
CMD="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --launch ${LAUNCH:-eager}"
rm -rf gpurun_out/prof_$tag gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 $CMD > gpurun_out/prof_$tag.log 2>&1; echo "stats rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_fetch -- python3 $CMD > gpurun_out/pmc_${tag}_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_write -- python3 $CMD > gpurun_out/pmc_${tag}_write.log 2>&1; echo "write rc=$?"
grep '^{' gpurun_out/prof_$tag.log | tail -1 > gpurun_out/prof_${tag}_bench.json
python3 tools/summarize_prof.py gpurun_out/prof_$tag > gpurun_out/prof_${tag}_stats.md; head -18 gpurun_out/prof_${tag}_stats.md
python3 tools/summarize_traffic.py gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write --json gpurun_out/traffic_${tag}.json | tee gpurun_out/prof_${tag}_traffic.md
