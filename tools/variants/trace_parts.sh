cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for P in 3 4; do
  export RANENV_PARTS=$P
  rm -rf gpurun_out/tr$P
  timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/tr$P -o t --output-format csv -- python3 tools/benchprobe.py 2 2>&1 | grep "cfg 2"
  f=$(find gpurun_out/tr$P -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_overlap.py $f
  python3 - $f <<'PY'
import csv, sys, collections
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
tail=rows[-60:]
t0=int(tail[0]["Start_Timestamp"])
for r in tail[:40]:
    print(r["Queue_Id"], r["Kernel_Name"][:50].replace("(anonymous namespace)::",""), r["Grid_Size"] if "Grid_Size" in r else "", (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3)
PY
  rm -rf gpurun_out/tr$P
done
