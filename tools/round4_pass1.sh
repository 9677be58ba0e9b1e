#!/bin/bash
# Round 4, first GPU pass: the parity suite with the new tests, the XCC-id probe, the bench lines on this box, and the counter
# passes of the schedule the headline runs (compact, fused rollouts).  bash tools/round4_pass1.sh <tag> -> gpurun_out/<tag>/
set -o pipefail
tag=${1:-r04a}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; tail -n ${TAILN:-3} $out/$name.log | cut -c1-700
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
step xcc 60 ./tools/xcc_probe
step pytest_gpu 900 python3 -m pytest tests -q -m gpu -x
step kprobe_default 300 python3 tools/kprobe.py
RANENV_FUSE_FIRST=1,3,5 step kprobe_first_135 300 python3 tools/kprobe.py
RANENV_FUSE_FIRST=5,3,1 step kprobe_first_531 300 python3 tools/kprobe.py
RANENV_SE_MODE=gather step kprobe_gather 300 python3 tools/kprobe.py
step bench 900 python3 bench.py
step bench_driver 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
K=100; CALLS=2; TT=$((10 + K * CALLS))
for mode in stream gather; do
  for c in FETCH_SIZE WRITE_SIZE; do
    step pmc_${mode}_$c 300 rocprofv3 --pmc $c -d $out/pmc_${mode}_$c -o p --output-format csv -- python3 tools/profile_rollout.py $K $CALLS $mode
  done
  step pmc_${mode}_sq 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $out/pmc_${mode}_sq -o p --output-format csv -- python3 tools/profile_rollout.py $K $CALLS $mode
  step pmc_${mode}_sq2 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/pmc_${mode}_sq2 -o p --output-format csv -- python3 tools/profile_rollout.py $K $CALLS $mode
done
python3 tools/pmc_collect_r4.py $out/r04_pmc.json 4096 2 \
  stream_rollout:$TT:$out/pmc_stream_FETCH_SIZE,$out/pmc_stream_WRITE_SIZE,$out/pmc_stream_sq,$out/pmc_stream_sq2 \
  gather_rollout:$TT:$out/pmc_gather_FETCH_SIZE,$out/pmc_gather_WRITE_SIZE,$out/pmc_gather_sq,$out/pmc_gather_sq2 > $out/pmc_collect.log
tail -c 3000 $out/pmc_collect.log
echo "pass complete"
