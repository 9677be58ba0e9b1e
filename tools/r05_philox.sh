#!/bin/bash
# the device traffic generator under library builds, same box, alternating: tools/r05_philox.sh <outdir> <lib.so> ...
out=$1; shift; mkdir -p $out
for rep in 1 2; do for v in "$@"; do n=$(basename $v .so); for K in 20 200; do
  RANENV_LIB=$v python bench.py --traffic philox --steps $K --warmup $((K/4)) --no-cpu-baseline --no-other-configs > $out/${n}_K${K}_r${rep}.json 2>/dev/null || echo FAILED
done; done; done
python - "$out" "$@" <<'PY'
import glob, json, sys, os
out = sys.argv[1]
for v in sys.argv[2:]:
    n = os.path.basename(v)[:-3]
    def f(key): return "   ".join(f"K={K}: " + " ".join("%.2f" % (key(json.load(open(f))) * 1e3) for f in sorted(glob.glob(f"{out}/{n}_K{K}_r*.json"))) for K in (20, 200))
    print(f"[{n}] rollout " + f(lambda d: d["ms_per_step"]) + "  | step loop " + f(lambda d: d["single_stream"]["ms_per_step"]) + "  | pipelined " + f(lambda d: d["pipelined_step"]["ms_per_step"]) + "  | gather " + f(lambda d: d["se_gather"]["ms_per_step"]), flush=True)
PY
