#!/bin/bash
set -o pipefail
tag=${1:-r04r}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | grep -v "persist stats" | tail -n ${TAILN:-2} | cut -c1-300
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
for pr in 0 1 2 3; do
  RANENV_PERSIST_PRIO=$pr step s_prio$pr 300 python3 tools/kprobe.py
done
for pr in 0 1 3; do
  RANENV_SE_MODE=gather RANENV_PERSIST_PRIO=$pr step g_prio$pr 300 python3 tools/kprobe.py
done
echo "pass complete"
