#!/bin/bash
# env.step() loop and learner-in-the-loop schedules under library builds, same box, alternating: tools/r05_steploop.sh <outdir> <lib.so> ...
out=$1; shift; mkdir -p $out
for rep in 1 2 3; do for v in "$@"; do n=$(basename $v .so)
  RANENV_LIB=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $out/${n}_r${rep}.json 2>/dev/null || echo FAILED
done; done
python - "$out" "$@" <<'PY'
import glob, json, sys, os
out = sys.argv[1]
for v in sys.argv[2:]:
    n = os.path.basename(v)[:-3]
    f = lambda key: " ".join("%.2f" % (key(json.load(open(p))) * 1e3) for p in sorted(glob.glob(f"{out}/{n}_r*.json")))
    print(f"[{n}] rollout " + f(lambda d: d["ms_per_step"]) + "  | step loop " + f(lambda d: d["single_stream"]["ms_per_step"]) + "  | pipelined " + f(lambda d: d["pipelined_step"]["ms_per_step"])
          + "  | gather " + f(lambda d: d["se_gather"]["ms_per_step"]) + "  | gather step loop " + f(lambda d: d["se_gather"].get("single_stream", {}).get("ms_per_step", 0)), flush=True)
PY
