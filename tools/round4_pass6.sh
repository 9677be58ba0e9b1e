#!/bin/bash
set -o pipefail
tag=${1:-r04h}
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
RANENV_LIB=$PWD/tools/variants/stamps.so step stamps_cfg1 200 python3 tools/stamps_cfg.py 1 40
RANENV_LIB=$PWD/tools/variants/stamps.so step stamps_cfg1_gather 200 python3 tools/stamps_cfg.py 1 40 gather
step s_default_cfg1 300 python3 tools/kprobe.py
for v in d8w2 d16w2 d8w3; do
  RANENV_LIB=$PWD/tools/variants/$v.so step s_${v}_cfg1 300 python3 tools/kprobe.py
done
RANENV_SMALL_BATCH=0 step s_lean_cfg1 300 python3 tools/kprobe.py
echo "pass complete"
