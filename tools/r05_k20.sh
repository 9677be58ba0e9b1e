#!/bin/bash
# K = 20 only, headline only, alternating: tools/r05_k20.sh <outdir> <reps> "ENV=.." ...
out=$1; reps=$2; shift 2; mkdir -p $out
for rep in $(seq 1 $reps); do i=0
for knobs in "$@"; do i=$((i+1))
  env $knobs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather --no-single-stream --no-other-configs > $out/v${i}_r${rep}.json 2> $out/v${i}_r${rep}.err || echo "FAILED $knobs"
done; done
i=0
for knobs in "$@"; do i=$((i+1))
python - "$out" "$i" "$knobs" <<'PY'
import glob, json, sys
out, i, knobs = sys.argv[1:4]
vs = []
for f in sorted(glob.glob(f"{out}/v{i}_r*.json")):
    try: vs.append(json.load(open(f))["ms_per_step"] * 1e3)
    except Exception: pass
print(f"[{knobs}] " + " ".join(f"{v:.2f}" for v in vs) + f"   mean {sum(vs)/max(1,len(vs)):.2f}", flush=True)
PY
done
