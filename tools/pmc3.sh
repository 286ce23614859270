#!/bin/bash
# dev helper: three counter passes over the full forward (bench_kernels --what full), per-kernel averages
tools/pmc.sh sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES" tools/bench_kernels.py --what ${1:-full} --iters 10
tools/pmc.sh tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" tools/bench_kernels.py --what ${1:-full} --iters 10
tools/pmc.sh ta "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" tools/bench_kernels.py --what ${1:-full} --iters 10
tools/pmc.sh lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE" tools/bench_kernels.py --what ${1:-full} --iters 10
