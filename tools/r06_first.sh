#!/bin/bash
# Round 6, first GPU pass: the GPU suite on the split library, the plain multi-rank command as a one-GPU rehearsal, the driver's command.
set -o pipefail
out=gpurun_out/${1:-r06a}; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; tail -3 $out/tests.log; [ $rc -ne 0 ] && exit 1
timeout -k 10 400 python3 bench.py --gpus 2 --rehearse-on-one-gpu --traces 40 --trace-len 100 --no-cpu-baseline --steps 20 --warmup 5 > $out/rehearse.json 2> $out/rehearse.err; rc=$?
echo "rehearsal rc=$rc"; tail -2 $out/rehearse.err | cut -c1-300; cut -c1-400 $out/rehearse.json; [ $rc -ne 0 ] && exit 1
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_k20.json 2> $out/bench_k20.err; rc=$?
echo "bench rc=$rc"; python3 - $out/bench_k20.json <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.2f M  frac %.3f  ms/step %.4f" % (l["value"] / 1e6, l["roofline"]["frac"], l["ms_per_step"]))
for k in ("single_stream", "pipelined_step"):
    print(k, "%.3f" % l[k]["roofline_frac"])
print("se_gather %.2f M issue %.3f" % (l["se_gather"]["value"] / 1e6, l["se_gather"].get("issue_frac", 0)))
print({k: round(v.get("roofline_frac", 0), 3) for k, v in l["other_configs"].items()})
PY
