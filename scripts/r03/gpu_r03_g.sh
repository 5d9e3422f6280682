#!/bin/bash
# counters of the tile-sorted walk kernel on ml.tree (small sets: the TA / TCP blocks have few counters each)
set -u
export PMC_TIMEOUT=150
export COUNTERS="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY;TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum;TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum;TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum;TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum;TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum;TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum;GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"
bash scripts/profile_pmc.sh r03_walk_sorted_ml --tree ml --pairs 10000000 --strategy walk --rounds 3 > gpurun_out/r03g_walk_sorted_ml.txt 2>&1
cat gpurun_out/r03g_walk_sorted_ml.txt | tail -45
