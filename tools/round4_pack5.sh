#!/bin/bash
set -o pipefail
tag=${1:-r04pack5}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | grep -v "persist stats" | tail -n ${TAILN:-3} | cut -c1-400
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
step pytest_gpu 1000 python3 -m pytest tests -q -m gpu -x
step bench_native 400 python3 bench.py --config native --no-cpu-baseline
RANENV_PACK=0 step bench_native_nopack 400 python3 bench.py --config native --no-cpu-baseline
echo "pass complete"
