#!/bin/bash
set -o pipefail
tag=${1:-r04l}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | tail -n ${TAILN:-3} | cut -c1-400
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
export KPROBE_CONFIG=1
for st in 0 6 12 16 24; do
  RANENV_PERSIST_STAGGER=$st step s_cfg1_stagger$st 300 python3 tools/kprobe.py
done
RANENV_SE_MODE=gather RANENV_PERSIST_STAGGER=0 step g_cfg1_stagger0 300 python3 tools/kprobe.py
RANENV_SE_MODE=gather RANENV_PERSIST_STAGGER=8 step g_cfg1_stagger8 300 python3 tools/kprobe.py
echo "pass complete"
