#!/bin/bash
set -o pipefail
tag=${1:-r04u}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | tail -n ${TAILN:-3} | cut -c1-300
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
RANENV_PERSIST=1 RANENV_PERSIST_GRID=64 RANENV_PERSIST_CHUNK=1 step pytest_handover_everywhere 900 python3 -m pytest tests -q -m gpu -x
RANENV_PERSIST=1 RANENV_PERSIST_GRID=600 RANENV_PERSIST_CHUNK=3 RANENV_SE_MODE=gather step pytest_handover_gather 900 python3 -m pytest tests -q -m gpu -x
RANENV_PERSIST=1 RANENV_FUZZ_CASES=150 step fuzz_soak_persist 900 python3 -m pytest tests/test_gpu_fuzz.py -q -x
RANENV_PERSIST=0 step pytest_nopersist 900 python3 -m pytest tests -q -m gpu -x
echo "pass complete"
