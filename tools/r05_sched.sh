#!/bin/bash
# single_stream / pipelined_step under knobs: tools/r05_sched.sh <outdir> "ENV=.." ...
out=$1; shift; mkdir -p $out
for rep in 1 2; do i=0
for knobs in "$@"; do i=$((i+1))
  env $knobs python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gather --no-other-configs > $out/s${i}_r${rep}.json 2> $out/s${i}_r${rep}.err || { echo "FAILED $knobs"; tail -3 $out/s${i}_r${rep}.err; }
done; done
i=0
for knobs in "$@"; do i=$((i+1))
python - "$out" "$i" "$knobs" <<'PY'
import glob, json, sys
out, i, knobs = sys.argv[1:4]
row = []
for f in sorted(glob.glob(f"{out}/s{i}_r*.json")):
    try:
        d = json.load(open(f)); row.append("head %.1f  single %.1f (%.3f)  pipe %.1f (%.3f)" % (d["ms_per_step"] * 1e3, d["single_stream"]["ms_per_step"] * 1e3, d["single_stream"]["roofline_frac"], d["pipelined_step"]["ms_per_step"] * 1e3, d["pipelined_step"]["roofline_frac"]))
    except Exception as e:
        row.append("?")
print(f"[{knobs}] " + " | ".join(row), flush=True)
PY
done
