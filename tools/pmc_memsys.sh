for set in "TA_BUSY_avr TCC_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_LATENCY_FIFO_FULL_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" "GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  mkdir -p gpurun_out/r02k_$tag
  # a pass killed by its timeout (a hang) ends the whole sweep: no further GPU work behind it, and its log is kept
  bash tools/pmc_variants.sh r02k_$tag "$set" base5 d6 d8 > gpurun_out/r02k_$tag/pass.log 2>&1 || { echo "counter set '$set' failed: see gpurun_out/r02k_$tag/pass.log"; exit 1; }
  grep "core_kernel<0>" gpurun_out/r02k_$tag/summary.txt
done
