#!/bin/bash
# Round 6: GPU suite on the shipped library, then a same-box alternating A/B of library builds (headline + gather, K = 20 and 200).
# tools/r06_ab.sh <tag> [notests] <spec> <spec> ...      spec = lib.so[@NAME=label[,ENV=value...]]  (environment presets of the run, e.g. RANENV_PERSIST_STATIC=0; entries starting with -- are bench.py arguments, ':' for a space: --partitions:1)
set -o pipefail
out=gpurun_out/$1; shift; mkdir -p $out; export TMPDIR=/tmp
if [ "$1" = notests ]; then shift; else
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; tail -3 $out/tests.log; [ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error" $out/tests.log | head; exit 1; }
fi
REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do for spec in "$@"; do v=${spec%%@*}; n=$(basename $v .so); envs=""; xargs=""
  if [ "$spec" != "$v" ]; then IFS=, read -ra kv <<< "${spec#*@}"; for e in "${kv[@]}"; do if [ "${e%%=*}" = NAME ]; then n=${e#*=}; elif [ "${e:0:2}" = "--" ]; then xargs="$xargs ${e//:/ }"; else envs="$envs $e"; fi; done; fi
  for K in ${KS:-20 200}; do
  env $envs RANENV_LIB=$PWD/$v timeout -k 10 300 python3 bench.py --steps $K --warmup $((K/4)) --no-cpu-baseline ${BENCH_FLAGS:---no-single-stream --no-other-configs} $xargs > $out/${n}_K${K}_r${rep}.json 2> $out/${n}_K${K}_r${rep}.err || { echo FAILED $spec; tail -3 $out/${n}_K${K}_r${rep}.err; exit 1; }
done; done; done
python3 - "$out" "$@" <<'PY'
import glob, json, sys, os
out = sys.argv[1]
for spec in sys.argv[2:]:
    v, _, rest = spec.partition("@")
    n = os.path.basename(v)[:-3]
    for e in rest.split(","):
        if e.startswith("NAME="): n = e[5:]
    row = []
    for f0 in sorted(set(f.split("_K")[1].split("_r")[0] for f in glob.glob(f"{out}/{n}_K*_r*.json")), key=int):
        K = int(f0)
        hs, gs, ss, ps = [], [], [], []
        for f in sorted(glob.glob(f"{out}/{n}_K{K}_r*.json")):
            try:
                d = json.loads(open(f).read().strip().splitlines()[-1]); hs.append(d["ms_per_step"] * 1e3)
                if "ms_per_step" in d.get("se_gather", {}): gs.append(d["se_gather"]["ms_per_step"] * 1e3)
                if "single_stream" in d: ss.append(d["single_stream"]["ms_per_step"] * 1e3); ps.append(d["pipelined_step"]["ms_per_step"] * 1e3)
            except Exception as e: print("bad", f, e)
        row.append(f"K={K}: stream " + " ".join(f"{x:.2f}" for x in hs) + "  gather " + " ".join(f"{x:.2f}" for x in gs)
                   + ("  step-loop " + " ".join(f"{x:.2f}" for x in ss) + "  pipelined " + " ".join(f"{x:.2f}" for x in ps) if ss else ""))
    print(f"[{n}] " + "   ".join(row), flush=True)
PY
