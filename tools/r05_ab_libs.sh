#!/bin/bash
# same-box alternating A/B of library builds: tools/r05_ab_libs.sh <outdir> <lib.so> <lib.so> ...  (headline + gather, K = 20 and 200, two rounds)
out=$1; shift; mkdir -p $out
for rep in 1 2; do for v in "$@"; do n=$(basename $v .so); for K in 20 200; do
  RANENV_LIB=$v python bench.py --steps $K --warmup $((K/4)) --no-cpu-baseline --no-single-stream --no-other-configs > $out/${n}_K${K}_r${rep}.json 2> $out/${n}_K${K}_r${rep}.err || { echo FAILED $v; tail -3 $out/${n}_K${K}_r${rep}.err; }
done; done; done
python - "$out" "$@" <<'PY'
import glob, json, sys, os
out = sys.argv[1]
for v in sys.argv[2:]:
    n = os.path.basename(v)[:-3]
    row = []
    for K in (20, 200):
        hs, gs = [], []
        for f in sorted(glob.glob(f"{out}/{n}_K{K}_r*.json")):
            try:
                d = json.load(open(f)); hs.append(d["ms_per_step"] * 1e3); gs.append(d["se_gather"]["ms_per_step"] * 1e3)
            except Exception: pass
        row.append(f"K={K}: stream " + " ".join(f"{x:.2f}" for x in hs) + "  gather " + " ".join(f"{x:.2f}" for x in gs))
    print(f"[{n}] " + "   ".join(row), flush=True)
PY
