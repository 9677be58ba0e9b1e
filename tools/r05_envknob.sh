#!/bin/bash
# runtime environment knobs, same box, alternating: tools/r05_envknob.sh <outdir> <reps> "ENV=.." ...   (headline + single_stream at K = 20)
out=$1; reps=$2; shift 2; mkdir -p $out
for rep in $(seq 1 $reps); do i=0
for knobs in "$@"; do i=$((i+1))
  env $knobs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-other-configs > $out/v${i}_r${rep}.json 2> $out/v${i}_r${rep}.err || echo "FAILED $knobs"
done; done
i=0
for knobs in "$@"; do i=$((i+1))
python - "$out" "$i" "$knobs" <<'PY'
import glob, json, sys
out, i, knobs = sys.argv[1:4]
hs, ss, ps = [], [], []
for f in sorted(glob.glob(f"{out}/v{i}_r*.json")):
    try:
        d = json.load(open(f)); hs.append(d["ms_per_step"] * 1e3)
        ss.append(d["single_stream"]["ms_per_step"] * 1e3); ps.append(d["pipelined_step"]["ms_per_step"] * 1e3)
    except Exception as e: print("?", e)
f = lambda v: " ".join(f"{x:.2f}" for x in v)
print(f"[{knobs}] rollout {f(hs)}   single_stream {f(ss)}   pipelined {f(ps)}", flush=True)
PY
done
