#!/bin/bash
# One measurement pass on the GPU box (round 3): parity tests, bench lines, PMC passes of both SE modes, rocprofv3 kernel stats.
# Usage (through gpurun): bash tools/round3_measure.sh <tag> [skip_pytest]     -> everything lands in gpurun_out/<tag>/
set -o pipefail
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; tail -n ${TAILN:-3} $out/$name.log | cut -c1-600
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
if [ -z "$2" ]; then step pytest_gpu 900 python3 -m pytest tests -q -m gpu; fi
step bench 900 python3 bench.py
step bench_driver 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
for mode in stream gather; do
  if [ $mode = gather ]; then export RANENV_SE_MODE=gather; else unset RANENV_SE_MODE; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    step pmc_${mode}_$c 300 rocprofv3 --pmc $c -d $out/pmc_${mode}_$c -o p --output-format csv -- python3 tools/profile_step.py 30
  done
  step pmc_${mode}_sq 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $out/pmc_${mode}_sq -o p --output-format csv -- python3 tools/profile_step.py 30
  step pmc_${mode}_sq2 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/pmc_${mode}_sq2 -o p --output-format csv -- python3 tools/profile_step.py 30
done
unset RANENV_SE_MODE
python3 tools/pmc_collect.py $out/r03_pmc.json 4096 2 stream:$out/pmc_stream_FETCH_SIZE,$out/pmc_stream_WRITE_SIZE,$out/pmc_stream_sq,$out/pmc_stream_sq2 gather:$out/pmc_gather_FETCH_SIZE,$out/pmc_gather_WRITE_SIZE,$out/pmc_gather_sq,$out/pmc_gather_sq2 > $out/pmc_collect.log
python3 tools/pmc_summary.py $out/pmc_stream_sq $out/pmc_stream_sq2 > $out/pmc_stream_summary.txt
python3 tools/pmc_summary.py $out/pmc_gather_sq $out/pmc_gather_sq2 > $out/pmc_gather_summary.txt
step prof_stream 400 rocprofv3 --kernel-trace --stats -d $out/prof_stream -o p --output-format csv -- python3 bench.py --steps 200 --no-cpu-baseline --no-single-stream --no-gather
step prof_gather 400 rocprofv3 --kernel-trace --stats -d $out/prof_gather -o p --output-format csv -- python3 bench.py --steps 200 --no-cpu-baseline --only-gather
step bench_cfg1 300 python3 bench.py --config 1 --no-cpu-baseline
step bench_cfg4 300 python3 bench.py --config 4 --no-cpu-baseline
step bench_philox 300 python3 bench.py --traffic philox --no-cpu-baseline --no-gather
if [ -f tools/variants/stamps.so ]; then
  export RANENV_LIB=$PWD/tools/variants/stamps.so RANENV_LATE=0
  for m in stream gather; do for c in 0 1; do
    RANENV_SE_MODE=$m RANENV_COMPACT=$c step residency_${m}_compact$c 200 python3 tools/residency.py
  done; done
  unset RANENV_LIB RANENV_LATE
fi
echo "pass complete"
