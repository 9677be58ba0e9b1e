#!/bin/bash
# One measurement pass on the GPU box: parity tests, smoke, PMC traffic + SQ counters, rocprofv3 kernel stats, bench lines.
# Usage (through gpurun): bash tools/round_measure.sh <tag>     -> everything lands in gpurun_out/<tag>/
set -o pipefail
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; tail -n ${TAILN:-3} $out/$name.log
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
step pytest_gpu 900 python3 -m pytest tests -q -m gpu
step smoke 300 python3 -c "import __graft_entry__ as g; g.smoke()"
for c in FETCH_SIZE WRITE_SIZE; do
  step pmc_$c 300 rocprofv3 --pmc $c -d $out/pmc_$c -o p --output-format csv -- python3 tools/profile_step.py 30
done
python3 tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE 4096 2 $out/pmc_traffic.json && cp $out/pmc_traffic.json profiles/pmc_traffic.json
step pmc_sq 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d $out/pmc_sq -o p --output-format csv -- python3 tools/profile_step.py 30
step pmc_sq2 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT -d $out/pmc_sq2 -o p --output-format csv -- python3 tools/profile_step.py 30
python3 tools/pmc_summary.py $out/pmc_sq $out/pmc_sq2 > $out/pmc_sq_summary.txt; cat $out/pmc_sq_summary.txt
step bench_prof 400 rocprofv3 --kernel-trace --stats -d $out/prof -o p --output-format csv -- python3 bench.py --steps 200 --no-cpu-baseline --no-single-stream
step bench 600 python3 bench.py
step bench_driver 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
step bench_single 300 python3 bench.py --partitions 1 --no-cpu-baseline
step bench_philox 300 python3 bench.py --traffic philox --no-cpu-baseline
step bench_cfg1 300 python3 bench.py --config 1 --no-cpu-baseline
step bench_cfg4 300 python3 bench.py --config 4 --no-cpu-baseline
if [ -f tools/variants/stamps.so ]; then
  RANENV_LIB=$PWD/tools/variants/stamps.so RANENV_LATE=0 step stamps_late0 300 python3 tools/stamps.py
  RANENV_LIB=$PWD/tools/variants/stamps.so step stamps_late1 300 python3 tools/stamps.py
fi
