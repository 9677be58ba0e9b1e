#!/bin/bash
# A/B of library variants on one box: tools/r05_variants.sh <outdir> <variant .so ...>; gather-only bench at K = 20 and K = 200, alternating
out=$1; shift
mkdir -p $out
for rep in 1 2; do
for v in "$@"; do
  n=$(basename $v .so)
  for K in 20 200; do
    RANENV_LIB=$v python bench.py --only-gather --steps $K --warmup $((K/4)) --no-cpu-baseline > $out/${n}_K${K}_r${rep}.json 2> $out/${n}_K${K}_r${rep}.err || exit 1
  done
done
done
python - "$out" <<'PY'
import glob, json, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.load(open(f)); g = d["se_gather"]
    print(os.path.basename(f), "%.1f M  %.2f us  hbm %.3f" % (g["value"] / 1e6, g["ms_per_step"] * 1e3, g["hbm_frac"]), flush=True)
PY
