#!/bin/bash
# The one profiling driver (run on the GPU box, from the repository root; everything lands under gpurun_out/).
# Counters are collected in their own rocprofv3 passes (--kernel-trace + --pmc only), the program directly behind `--`.
#
#   tools/prof.sh stats <tag> <python args...>            kernel stats of a command      -> gpurun_out/prof_<tag>/, condensed table
#   tools/prof.sh pmc   <tag> "<counters>" <python args...>   one counter pass, per-kernel averages -> gpurun_out/pmc_<tag>/
#   tools/prof.sh step  <tag> [K]                         the TIMED REGION ONLY of the headline: warm-up + K recorded steps of
#                                                         tools/step_only.py and nothing else, under the kernel trace; then
#                                                         FETCH_SIZE, WRITE_SIZE, TCC_EA0_RDREQ and MFMA passes of the SAME command.
#                                                         -> gpurun_out/step_<tag>.md (per-step kernel durations, their sum against
#                                                         the step's wall time), traffic_<tag>.json, mfma_<tag>.json
#   tools/prof.sh quick <tag> [K]                         only the kernel trace + per-step table of the same command (A / B of a kernel in its step)
#   tools/prof.sh train <tag> [K]                         the same for the replayed training step (tools/train_step.py)
#   tools/prof.sh counters <tag> <python args...>         the SQ / LDS passes used for DESIGN.md's limiter statements
set -u
mode=${1:?mode}; tag=${2:?tag}; shift; shift
root=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $root

trace() {   # trace <dir> <python args...>
    local d=$1; shift
    rm -rf $d
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 "$@" > $d.log 2>&1
    echo "trace $d rc=$?"
}
pmc() {     # pmc <dir> "<counters>" <python args...>
    local d=$1 ctrs=$2; shift; shift
    rm -rf $d
    rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $d -- python3 "$@" > $d.log 2>&1
    echo "pmc $d rc=$?"
}

case $mode in
stats)
    trace gpurun_out/prof_$tag "$@"
    python3 tools/summarize_prof.py gpurun_out/prof_$tag | head -${TOPN:-24}
    ;;
pmc)
    ctrs=$1; shift
    pmc gpurun_out/pmc_$tag "$ctrs" "$@"
    python3 tools/summarize_pmc.py gpurun_out/pmc_$tag
    ;;
counters)
    pmc gpurun_out/pmc_${tag}_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES" "$@"
    python3 tools/summarize_pmc.py gpurun_out/pmc_${tag}_sq
    pmc gpurun_out/pmc_${tag}_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE" "$@"
    python3 tools/summarize_pmc.py gpurun_out/pmc_${tag}_lds
    ;;
quick)      # tools/prof.sh quick <tag> [K]: the kernel trace of the timed region only (no counter passes): an A / B of kernel variants in their step
    K=${1:-40}
    trace gpurun_out/step_$tag tools/step_only.py --steps $K
    python3 tools/summarize_step.py gpurun_out/step_$tag --steps $K --log gpurun_out/step_$tag.log
    ;;
step|train)
    K=${1:-40}
    if [ $mode = step ]; then CMD="tools/step_only.py --steps $K"; else CMD="tools/train_step.py $K"; fi
    trace gpurun_out/${mode}_$tag $CMD
    pmc gpurun_out/${mode}_${tag}_fetch FETCH_SIZE $CMD
    pmc gpurun_out/${mode}_${tag}_write WRITE_SIZE $CMD
    pmc gpurun_out/${mode}_${tag}_rdreq "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" $CMD
    pmc gpurun_out/${mode}_${tag}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE" $CMD
    python3 tools/summarize_step.py gpurun_out/${mode}_$tag --steps $K --log gpurun_out/${mode}_$tag.log > gpurun_out/${mode}_$tag.md
    python3 tools/summarize_traffic.py gpurun_out/${mode}_${tag}_fetch gpurun_out/${mode}_${tag}_write --json gpurun_out/traffic_$tag.json >> gpurun_out/${mode}_$tag.md
    python3 tools/summarize_pmc.py gpurun_out/${mode}_${tag}_rdreq >> gpurun_out/${mode}_$tag.md
    python3 tools/summarize_mfma.py gpurun_out/${mode}_${tag}_mfma --json gpurun_out/mfma_$tag.json >> gpurun_out/${mode}_$tag.md
    cat gpurun_out/${mode}_$tag.md
    ;;
*)
    echo "unknown mode $mode"; exit 2;;
esac
