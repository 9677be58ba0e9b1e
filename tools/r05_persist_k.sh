#!/bin/bash
# launch-per-chunk against persistent launches for the STREAMING headline over rollout lengths, same box, alternating:
#   tools/r05_persist_k.sh <outdir> <reps> <K> <K> ...
out=$1; reps=$2; shift 2; mkdir -p $out
for rep in $(seq 1 $reps); do for K in "$@"; do for P in 0 1; do
  RANENV_PERSIST=$([ $P = 1 ] && echo 1 || echo -1) python bench.py --steps $K --warmup $((K/4)) --no-cpu-baseline --no-gather --no-single-stream --no-other-configs > $out/K${K}_p${P}_r${rep}.json 2>/dev/null || echo FAILED
done; done; done
python - "$out" "$@" <<'PY'
import glob, json, sys
out = sys.argv[1]
for K in sys.argv[2:]:
    row = []
    for P in (0, 1):
        v = [json.load(open(f))["ms_per_step"] * 1e3 for f in sorted(glob.glob(f"{out}/K{K}_p{P}_r*.json"))]
        row.append(("persistent " if P else "per chunk  ") + " ".join(f"{x:.2f}" for x in v) + f"  mean {sum(v) / max(1, len(v)):.2f}")
    print(f"K = {K}: " + "   |   ".join(row), flush=True)
PY
