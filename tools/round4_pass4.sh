#!/bin/bash
set -o pipefail
tag=${1:-r04e}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | tail -n ${TAILN:-16} | cut -c1-300
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
export RANENV_SE_MODE=gather
step probe_4096 300 python3 tools/persist_probe.py 20 10
RANENV_PERSIST_GRID=5000 step probe_4096_g5000 300 python3 tools/persist_probe.py 20 6
RANENV_PERSIST_GRID=4800 step probe_4096_g4800 300 python3 tools/persist_probe.py 20 6
KPROBE_BATCH=3900 step probe_3900 300 python3 tools/persist_probe.py 20 6
echo "pass complete"
