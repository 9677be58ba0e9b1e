#!/bin/bash
set -o pipefail
tag=${1:-r04pack2}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | grep -v "persist stats" | tail -n ${TAILN:-2} | cut -c1-400
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
for xm in 0 1; do for pk in 0 1; do
  RANENV_XCD_MAP=$xm RANENV_PACK=$pk KPROBE_CONFIG=5 step s_native_x${xm}_p$pk 300 python3 tools/kprobe.py
  RANENV_XCD_MAP=$xm RANENV_PACK=$pk KPROBE_CONFIG=5 RANENV_SE_MODE=gather step g_native_x${xm}_p$pk 300 python3 tools/kprobe.py
done; done
RANENV_PERSIST=0 RANENV_XCD_MAP=0 step s_4096_x0 300 python3 tools/kprobe.py
RANENV_PERSIST=0 RANENV_XCD_MAP=1 step s_4096_x1 300 python3 tools/kprobe.py
RANENV_PERSIST=0 RANENV_XCD_MAP=0 RANENV_SE_MODE=gather step g_4096_x0 300 python3 tools/kprobe.py
RANENV_PERSIST=0 RANENV_XCD_MAP=1 RANENV_SE_MODE=gather step g_4096_x1 300 python3 tools/kprobe.py
echo "pass complete"
