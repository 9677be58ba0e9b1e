#!/bin/bash
# Packed waves (two envs per wave at the reference's own size): parity suite, then A/B against one env per wave (option pack).
set -o pipefail
tag=${1:-r04pack}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | grep -v "persist stats" | tail -n ${TAILN:-3} | cut -c1-400
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
step pytest_gpu 900 python3 -m pytest tests -q -m gpu -x
RANENV_PACK=0 KPROBE_CONFIG=5 step s_native_pack0 300 python3 tools/kprobe.py
RANENV_PACK=1 KPROBE_CONFIG=5 step s_native_pack1 300 python3 tools/kprobe.py
RANENV_PACK=0 KPROBE_CONFIG=5 RANENV_SE_MODE=gather step g_native_pack0 300 python3 tools/kprobe.py
RANENV_PACK=1 KPROBE_CONFIG=5 RANENV_SE_MODE=gather step g_native_pack1 300 python3 tools/kprobe.py
echo "pass complete"
