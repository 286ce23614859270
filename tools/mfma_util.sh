#!/bin/bash
# MFMA utilisation of the dense steps on the GPU box: one PMC pass of the default bench command (kernel-trace only).
#   tools/mfma_util.sh <tag>   -> gpurun_out/pmc_<tag>_mfma/, gpurun_out/mfma_<tag>.json + .md
tag=${1:-r02}
root=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $root
CMD="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --launch eager"
rm -rf gpurun_out/pmc_${tag}_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_mfma -- python3 $CMD > gpurun_out/pmc_${tag}_mfma.log 2>&1; echo "mfma rc=$?"
python3 tools/summarize_mfma.py gpurun_out/pmc_${tag}_mfma --json gpurun_out/mfma_${tag}.json | tee gpurun_out/mfma_${tag}.md
