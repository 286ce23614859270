#!/bin/bash
# dev helper: the counter passes the round-3 verdict asked for on the planned decoder (k_distmult_plan), each in its own run
what=${1:-distmult}
tools/pmc.sh dm_sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES" tools/bench_kernels.py --what $what --iters 10
tools/pmc.sh dm_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE" tools/bench_kernels.py --what $what --iters 10
tools/pmc.sh dm_lds2 "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" tools/bench_kernels.py --what $what --iters 10
