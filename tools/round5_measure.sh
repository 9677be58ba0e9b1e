#!/bin/bash
# One measurement pass on the GPU box (round 5): parity tests, bench lines, PMC passes of the schedules bench.py times, rocprofv3
# kernel stats.  Usage (through gpurun): bash tools/round5_measure.sh <tag> [skip_pytest]   -> everything lands in gpurun_out/<tag>/
set -o pipefail
tag=${1:-r05}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | tail -n ${TAILN:-3} | cut -c1-500
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
if [ -z "$2" ]; then step pytest_gpu 900 python3 -m pytest tests -q -m gpu; fi
step bench 900 python3 bench.py
step bench_driver 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
RANENV_PERSIST=1 step bench_persist 600 python3 bench.py --no-cpu-baseline
RANENV_PERSIST=1 step bench_driver_persist 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
RANENV_SE_LAYOUT=rb step bench_rbmajor 600 python3 bench.py --no-cpu-baseline
RANENV_SE_LAYOUT=rb step bench_driver_rbmajor 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
K=100; CALLS=2; TT=$((10 + K * CALLS))
for mode in stream gather; do
  for c in FETCH_SIZE WRITE_SIZE; do
    step pmc_${mode}_$c 300 rocprofv3 --pmc $c -d $out/pmc_${mode}_$c -o p --output-format csv -- python3 tools/profile_rollout.py $K $CALLS $mode
  done
  step pmc_${mode}_sq 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $out/pmc_${mode}_sq -o p --output-format csv -- python3 tools/profile_rollout.py $K $CALLS $mode
  step pmc_${mode}_sq2 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/pmc_${mode}_sq2 -o p --output-format csv -- python3 tools/profile_rollout.py $K $CALLS $mode
done
python3 tools/pmc_collect_r4.py $out/r05_pmc.json 4096 2 \
  stream_rollout:$TT:$out/pmc_stream_FETCH_SIZE,$out/pmc_stream_WRITE_SIZE,$out/pmc_stream_sq,$out/pmc_stream_sq2 \
  gather_rollout:$TT:$out/pmc_gather_FETCH_SIZE,$out/pmc_gather_WRITE_SIZE,$out/pmc_gather_sq,$out/pmc_gather_sq2 > $out/pmc_collect.log
python3 tools/pmc_summary.py $out/pmc_stream_sq $out/pmc_stream_sq2 > $out/pmc_stream_summary.txt
python3 tools/pmc_summary.py $out/pmc_gather_sq $out/pmc_gather_sq2 > $out/pmc_gather_summary.txt
step prof_stream 400 rocprofv3 --kernel-trace --stats -d $out/prof_stream -o p --output-format csv -- python3 bench.py --steps 200 --no-cpu-baseline --no-single-stream --no-gather --no-other-configs
step prof_stream_k20 400 rocprofv3 --kernel-trace --stats -d $out/prof_stream_k20 -o p --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-stream --no-gather --no-other-configs
step prof_gather 400 rocprofv3 --kernel-trace --stats -d $out/prof_gather -o p --output-format csv -- python3 bench.py --steps 200 --no-cpu-baseline --only-gather --no-other-configs
step bench_cfg1 300 python3 bench.py --config 1 --no-cpu-baseline
step bench_cfg4 300 python3 bench.py --config 4 --no-cpu-baseline
step bench_native 400 python3 bench.py --config native --no-cpu-baseline
RANENV_PACK=0 step bench_native_nopack 400 python3 bench.py --config native --no-cpu-baseline
step bench_philox 300 python3 bench.py --traffic philox --no-cpu-baseline --no-gather
echo "pass complete"
