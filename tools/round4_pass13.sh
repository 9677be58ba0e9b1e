#!/bin/bash
set -o pipefail
tag=${1:-r04p}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | tail -n ${TAILN:-3} | cut -c1-600
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
step pytest_gpu 900 python3 -m pytest tests -q -m gpu -x
RANENV_SE_MODE=gather step pytest_gpu_gather 900 python3 -m pytest tests -q -m gpu -x
RANENV_ROW_WIDTH=16 RANENV_PERSIST=1 step pytest_gpu_np16_persist 900 python3 -m pytest tests -q -m gpu -x
echo "pass complete"
