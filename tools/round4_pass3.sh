#!/bin/bash
set -o pipefail
tag=${1:-r04c}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | tail -n ${TAILN:-4} | cut -c1-900
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
step pytest_persist 400 python3 -m pytest tests/test_gpu_fused_rollout.py -q -x -k persistent
export RANENV_SE_MODE=gather
KPROBE_BATCH=3600 step g_default_3600 300 python3 tools/kprobe.py
KPROBE_BATCH=3600 RANENV_PERSIST=1 step g_persist_3600 300 python3 tools/kprobe.py
step g_default_4096 300 python3 tools/kprobe.py
RANENV_PERSIST=1 step g_persist_4096 300 python3 tools/kprobe.py
RANENV_PERSIST=1 RANENV_LIB=$PWD/tools/variants/noacq.so step g_persist_4096_noacq 300 python3 tools/kprobe.py
RANENV_PERSIST=1 RANENV_PERSIST_GRID=4000 step g_persist_4096_g4000 300 python3 tools/kprobe.py
RANENV_PERSIST=1 RANENV_PERSIST_CHUNK=5 step g_persist_4096_c5 300 python3 tools/kprobe.py
RANENV_PERSIST=1 RANENV_PERSIST_CHUNK=20 step g_persist_4096_c20 300 python3 tools/kprobe.py
unset RANENV_SE_MODE
step s_default_4096 300 python3 tools/kprobe.py
RANENV_PERSIST=1 step s_persist_4096 300 python3 tools/kprobe.py
echo "pass complete"
