#!/bin/bash
set -o pipefail
tag=${1:-r04mix3}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | grep -v "persist stats" | tail -n ${TAILN:-3} | cut -c1-2000
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
step pytest_gpu 1000 python3 -m pytest tests -q -m gpu -x
RANENV_MIX=2 step pytest_gpu_mix2 1000 python3 -m pytest tests -q -m gpu -x --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_fuzz.py
step bench 600 python3 bench.py --no-cpu-baseline
RANENV_MIX=0 step bench_nomix 600 python3 bench.py --no-cpu-baseline
echo "pass complete"
