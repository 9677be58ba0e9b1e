#!/bin/bash
set -o pipefail
tag=${1:-r04f}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | tail -n ${TAILN:-5} | cut -c1-300
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
step pytest_persist 400 python3 -m pytest tests/test_gpu_fused_rollout.py -q -x -k persistent
export RANENV_SE_MODE=gather
step g_default_4096 300 python3 tools/kprobe.py
RANENV_PERSIST=1 step g_persist_4096 300 python3 tools/kprobe.py
RANENV_PERSIST=1 RANENV_PERSIST_CHUNK=5 step g_persist_4096_c5 300 python3 tools/kprobe.py
RANENV_PERSIST=1 RANENV_PERSIST_CHUNK=3 step g_persist_4096_c3 300 python3 tools/kprobe.py
KPROBE_CONFIG=1 step g_default_cfg1 300 python3 tools/kprobe.py
KPROBE_CONFIG=1 RANENV_PERSIST=1 step g_persist_cfg1 300 python3 tools/kprobe.py
KPROBE_CONFIG=4 step g_default_cfg4 300 python3 tools/kprobe.py
KPROBE_CONFIG=4 RANENV_PERSIST=1 step g_persist_cfg4 300 python3 tools/kprobe.py
unset RANENV_SE_MODE
step s_default_4096 300 python3 tools/kprobe.py
RANENV_PERSIST=1 step s_persist_4096 300 python3 tools/kprobe.py
KPROBE_CONFIG=1 step s_default_cfg1 300 python3 tools/kprobe.py
KPROBE_CONFIG=1 RANENV_PERSIST=1 step s_persist_cfg1 300 python3 tools/kprobe.py
echo "pass complete"
