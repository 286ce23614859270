#!/bin/bash
# Round-4 profile on the GPU box (everything lands under gpurun_out/, tools/write_profile_r04.py turns it into profiles/r04*):
#   kernel stats of the default bench command, HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes), MFMA utilisation,
#   SQ / LDS counter passes for the relational kernel, the decoder and the gene gather, in-kernel stamps, launch modes.
tag=${1:-r04}
LAUNCH=recorded tools/profile_round.sh $tag > gpurun_out/profile_round_$tag.txt 2>&1
tools/mfma_util.sh $tag > gpurun_out/mfma_util_$tag.txt 2>&1
tools/pmc.sh ${tag}_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES" tools/bench_kernels.py --what full --iters 10 > gpurun_out/pmc_${tag}_sq.txt 2>&1
tools/pmc.sh ${tag}_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE" tools/bench_kernels.py --what full --iters 10 > gpurun_out/pmc_${tag}_lds.txt 2>&1
export GN_HIP_LIBRARY=$PWD/gripnet_amd/lib/libgripnet_hip_stamps.so
python3 tools/pair_stamps.py --planes > gpurun_out/stamps_${tag}_pair.txt 2>&1
python3 tools/dm_stamps.py > gpurun_out/stamps_${tag}_dm.txt 2>&1
python3 tools/blk_stamps.py > gpurun_out/stamps_${tag}_blk.txt 2>&1
python3 tools/rel_stamps.py > gpurun_out/stamps_${tag}_rel.txt 2>&1
unset GN_HIP_LIBRARY
# the training step: kernel stats and the timeline of one replayed step, the fused weight gradient's counters, the co-issue probe
TOPN=90 tools/prof_stats.sh ${tag}_train tools/train_step.py 20 > gpurun_out/train_stats_$tag.md 2>&1
python3 tools/timeline.py gpurun_out/prof_${tag}_train 75 > gpurun_out/train_timeline_$tag.txt 2>&1
tools/pmc_relgrad.sh > gpurun_out/pmc_${tag}_relgrad.txt 2>&1
python3 tools/relgrad_probe.py > gpurun_out/relgrad_$tag.txt 2>&1
hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_valu_probe.hip -o /tmp/mfma_valu_probe > /dev/null 2>&1 && /tmp/mfma_valu_probe > gpurun_out/mfma_valu_probe_$tag.txt 2>&1
python3 tools/train_step.py 40 > gpurun_out/train_step_$tag.txt 2>&1
# the sampler's two kernels, the rate of LDS atomics (why the decoder's backward sorts instead of scattering)
python3 tools/probes/sampler_probe.py > gpurun_out/sampler_probe_$tag.txt 2>&1
mkdir -p tools/probes/bin && hipcc -O3 --offload-arch=gfx950 tools/probes/lds_atomic_probe.hip -o tools/probes/bin/lds_atomic_probe > /dev/null 2>&1 && tools/probes/bin/lds_atomic_probe > gpurun_out/lds_atomic_probe_$tag.txt 2>&1
python3 tools/launch_modes.py > gpurun_out/launch_modes_$tag.txt 2>&1
python3 tools/bench_kernels.py --what rgcn --arith fast --rgcn-kernel pair --iters 40 > gpurun_out/pair_fast_$tag.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_$tag.log 2>&1
grep '^{' gpurun_out/bench_$tag.log | tail -1 > gpurun_out/bench_$tag.json
echo done
