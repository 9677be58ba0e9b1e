#!/bin/bash
# Memory-system counters of the step kernel in both SE modes (one launch per TTI): tools/pmc_memsys_r3.sh <tag>
tag=${1:-r03mem}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for mode in stream gather; do
  if [ $mode = gather ]; then export RANENV_SE_MODE=gather; else unset RANENV_SE_MODE; fi
  i=0
  for set in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" "TA_BUSY_avr TCC_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set -d $out/${mode}_$i -o p --output-format csv -- python3 tools/profile_step.py 30 > $out/${mode}_$i.log 2>&1; rc=$?
    echo "[$mode: $set] rc=$rc"
    if [ $rc -ne 0 ]; then echo "counter pass failed: see $out/${mode}_$i.log"; exit 1; fi
  done
  echo "== SE mode $mode (per launch of the STEP kernel = per TTI of 4096 envs, one launch per TTI on one stream)" >> $out/summary.txt
  python3 tools/pmc_summary.py $out/${mode}_1 $out/${mode}_2 $out/${mode}_3 | grep "<0" >> $out/summary.txt
done
cat $out/summary.txt
