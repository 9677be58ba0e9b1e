#!/bin/bash
# headline-only bench under option knobs, K = 20 and K = 200, two alternating repeats: tools/r05_matrix.sh <outdir> "ENV1=a ENV2=b" "..." ...
out=$1; shift
mkdir -p $out
i=0
for rep in 1 2; do
i=0
for knobs in "$@"; do
  i=$((i+1))
  for K in 20 200; do
    env $knobs python bench.py --steps $K --warmup $((K/4)) --no-cpu-baseline --no-gather --no-single-stream --no-other-configs > $out/v${i}_K${K}_r${rep}.json 2> $out/v${i}_K${K}_r${rep}.err || { echo "FAILED: $knobs"; tail -3 $out/v${i}_K${K}_r${rep}.err; }
  done
done
done
i=0
for knobs in "$@"; do
  i=$((i+1))
  python - "$out" "$i" "$knobs" <<'PY'
import glob, json, sys
out, i, knobs = sys.argv[1:4]
r = {}
for K in (20, 200):
    vs = []
    for f in sorted(glob.glob(f"{out}/v{i}_K{K}_r*.json")):
        try:
            d = json.load(open(f)); vs.append(d["ms_per_step"] * 1e3)
        except Exception:
            pass
    r[K] = vs
print(f"[{knobs or 'default'}]  K=20: " + " ".join(f"{v:.2f}" for v in r[20]) + "   K=200: " + " ".join(f"{v:.2f}" for v in r[200]) + "  us per TTI", flush=True)
PY
done
