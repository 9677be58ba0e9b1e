#!/bin/bash
set -o pipefail
tag=${1:-r04o}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | tail -n ${TAILN:-3} | cut -c1-2500
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
step bench_native 400 python3 bench.py --config native --no-cpu-baseline
KPROBE_CONFIG=5 step s_native 300 python3 tools/kprobe.py
KPROBE_CONFIG=5 RANENV_SE_MODE=gather step g_native 300 python3 tools/kprobe.py
KPROBE_CONFIG=5 RANENV_SE_MODE=gather RANENV_PERSIST=0 step g_native_nopersist 300 python3 tools/kprobe.py
KPROBE_CONFIG=5 RANENV_PERSIST=1 step s_native_persist 300 python3 tools/kprobe.py
echo "pass complete"
