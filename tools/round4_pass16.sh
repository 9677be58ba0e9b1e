#!/bin/bash
set -o pipefail
tag=${1:-r04t}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | grep -v "persist stats" | tail -n ${TAILN:-2} | cut -c1-300
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
for o in 0 1 0 1; do
  RANENV_PERSIST_ORDER=$o step s_order${o}_$RANDOM 300 python3 tools/kprobe.py
done
for o in 0 1; do
  RANENV_SE_MODE=gather RANENV_PERSIST_ORDER=$o step g_order$o 300 python3 tools/kprobe.py
done
echo "pass complete"
