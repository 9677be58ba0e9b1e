#!/bin/bash
# Last sanity pass: what the driver runs (smoke), the examples, and the multi-rank plumbing rehearsal on one GPU.
set -o pipefail
tag=${1:-r04sanity}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to "$@" > $out/$name.log 2>&1; local rc=$?
  echo "[$name] rc=$rc"; grep -v amdgpu.ids $out/$name.log | tail -n ${TAILN:-3} | cut -c1-300
  if [ $rc -ne 0 ]; then echo "step $name failed: stopping"; exit 1; fi
}
step smoke 300 python3 -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')"
step example_eval 300 python3 examples/evaluate_baselines.py
step example_learner 300 python3 examples/learner_in_the_loop.py --ttis 200
step rehearsal 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --rehearse-on-one-gpu --traces 40 --trace-len 100 --no-cpu-baseline
step bench_driver 300 python3 bench.py --gpus 1 --steps 20 --warmup 5
echo "pass complete"
