#!/bin/bash
# configs[1] alone under library builds, alternating: tools/r05_cfg1.sh <outdir> <lib.so> ...
out=$1; shift; mkdir -p $out
for rep in 1 2 3; do for v in "$@"; do n=$(basename $v .so); for K in 20 200; do
  RANENV_LIB=$v python bench.py --config 1 --steps $K --warmup $((K/4)) --no-cpu-baseline --no-gather > $out/${n}_K${K}_r${rep}.json 2>/dev/null || echo FAILED
done; done; done
python - "$out" "$@" <<'PY'
import glob, json, sys, os
out = sys.argv[1]
for v in sys.argv[2:]:
    n = os.path.basename(v)[:-3]
    print(f"[{n}] " + "   ".join(f"K={K}: " + " ".join("%.2f" % (json.load(open(f))["ms_per_step"] * 1e3) for f in sorted(glob.glob(f"{out}/{n}_K{K}_r*.json"))) for K in (20, 200)), flush=True)
PY
