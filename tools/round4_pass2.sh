#!/bin/bash
# Round 4, pass 2: the persistent rollout.  Parity first (bounded), then A/B against the launch-per-chunk schedule.
set -o pipefail
tag=${1:-r04b}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; tail -n ${TAILN:-4} $out/$name.log | cut -c1-900
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
step pytest_persist 400 python3 -m pytest tests/test_gpu_fused_rollout.py -q -x -k persistent
step kprobe_default 300 python3 tools/kprobe.py
RANENV_PERSIST=1 step kprobe_persist 300 python3 tools/kprobe.py
RANENV_SE_MODE=gather step kprobe_gather 300 python3 tools/kprobe.py
RANENV_SE_MODE=gather RANENV_PERSIST=1 step kprobe_gather_persist 300 python3 tools/kprobe.py
RANENV_SE_MODE=gather RANENV_PERSIST=1 RANENV_PERSIST_CHUNK=5 step kprobe_gather_persist_c5 300 python3 tools/kprobe.py
RANENV_SE_MODE=gather RANENV_PERSIST=1 RANENV_PERSIST_CHUNK=20 step kprobe_gather_persist_c20 300 python3 tools/kprobe.py
RANENV_SE_MODE=gather RANENV_PERSIST=1 RANENV_PERSIST_GRID=4600 step kprobe_gather_persist_g4600 300 python3 tools/kprobe.py
RANENV_PERSIST=1 RANENV_PERSIST_GRID=4600 step kprobe_persist_g4600 300 python3 tools/kprobe.py
echo "pass complete"
