#!/bin/bash
set -o pipefail
tag=${1:-r04m}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | tail -n ${TAILN:-3} | cut -c1-400
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
step pytest_cfg 400 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_fused_rollout.py -q -x
KPROBE_CONFIG=1 step s_cfg1 300 python3 tools/kprobe.py
step bench_cfg1 300 python3 bench.py --config 1 --no-cpu-baseline
echo "pass complete"
