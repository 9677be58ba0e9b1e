#!/bin/bash
set -o pipefail
tag=${1:-r04k}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | tail -n ${TAILN:-3} | cut -c1-400
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
step pytest_persist 400 python3 -m pytest tests/test_gpu_fused_rollout.py -q -x
KPROBE_CONFIG=1 step s_auto_cfg1 300 python3 tools/kprobe.py
KPROBE_CONFIG=1 RANENV_SE_MODE=gather step g_auto_cfg1 300 python3 tools/kprobe.py
step bench_cfg1 300 python3 bench.py --config 1 --no-cpu-baseline
KPROBE_CONFIG=1 KPROBE_BATCH=2048 RANENV_PERSIST=0 step s_default_2048 300 python3 tools/kprobe.py
KPROBE_CONFIG=1 KPROBE_BATCH=2048 RANENV_PERSIST=1 step s_persist_2048 300 python3 tools/kprobe.py
KPROBE_CONFIG=1 KPROBE_BATCH=2048 RANENV_PERSIST=0 RANENV_SE_MODE=gather step g_default_2048 300 python3 tools/kprobe.py
KPROBE_CONFIG=1 KPROBE_BATCH=2048 RANENV_PERSIST=1 RANENV_SE_MODE=gather step g_persist_2048 300 python3 tools/kprobe.py
echo "pass complete"
