#!/bin/bash
set -o pipefail
tag=${1:-r04pack3}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 $out/$name.log; echo "step $name failed: stopping"; exit 1; fi
}
K=50; CALLS=1; TT=$((10 + K * CALLS))
for pk in 0 1; do
  export RANENV_PACK=$pk
  for c in FETCH_SIZE WRITE_SIZE; do
    step pmc_p${pk}_$c 300 rocprofv3 --pmc $c -d $out/pmc_p${pk}_$c -o p --output-format csv -- python3 tools/profile_rollout.py $K $CALLS stream 5
  done
  step pmc_p${pk}_sq 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $out/pmc_p${pk}_sq -o p --output-format csv -- python3 tools/profile_rollout.py $K $CALLS stream 5
  step pmc_p${pk}_sq2 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/pmc_p${pk}_sq2 -o p --output-format csv -- python3 tools/profile_rollout.py $K $CALLS stream 5
done
python3 tools/pmc_collect_r4.py $out/native_pmc.json 16384 5 \
  pack0_rollout:$TT:$out/pmc_p0_FETCH_SIZE,$out/pmc_p0_WRITE_SIZE,$out/pmc_p0_sq,$out/pmc_p0_sq2 \
  pack1_rollout:$TT:$out/pmc_p1_FETCH_SIZE,$out/pmc_p1_WRITE_SIZE,$out/pmc_p1_sq,$out/pmc_p1_sq2 > $out/pmc_collect.log
echo "pass complete"
