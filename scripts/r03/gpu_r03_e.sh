#!/bin/bash
# counters of k_walk on ml.tree: stride-3 climb (lineage_lens=0) vs lineage-length stream (1)
set -u
export COUNTERS="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU;TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum;TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE;TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
for L in 0 1; do
  bash scripts/profile_pmc.sh r03_walk_ml_lens$L --tree ml --pairs 10000000 --strategy walk --rounds 3 --opt lineage_lens=$L > gpurun_out/r03e_lens$L.txt 2>&1
done
tail -50 gpurun_out/r03e_lens0.txt gpurun_out/r03e_lens1.txt
