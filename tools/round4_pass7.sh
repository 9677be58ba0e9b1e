#!/bin/bash
set -o pipefail
tag=${1:-r04i}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | tail -n ${TAILN:-3} | cut -c1-400
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
export KPROBE_CONFIG=1 RANENV_PERSIST=1
step s_persist_cfg1 300 python3 tools/kprobe.py
for v in pd16w2 pd8w2 pd4w4; do
  RANENV_LIB=$PWD/tools/variants/$v.so step s_${v}_cfg1 300 python3 tools/kprobe.py
done
echo "pass complete"
