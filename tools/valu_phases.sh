#!/bin/bash
# VALU wave-instructions per TTI by role: SQ_INSTS_VALU of the full build and of the ablation builds (-DRANENV_DIAG=3 no UE step, 4 no observation
# tail, 7 no allocation, 11 no masked half of the stream), launch-per-chunk rollouts (RANENV_PERSIST=0), both SE modes
# The ablation builds are made in the build container first: tools/build_variants.sh diag3:-DRANENV_DIAG=3 diag4:-DRANENV_DIAG=4 diag7:-DRANENV_DIAG=7 diag11:-DRANENV_DIAG=11
out=$1; V=tools/variants; mkdir -p $out; export TMPDIR=/tmp
for mode in stream gather; do
for v in full diag3 diag4 diag7 diag11; do
  lib=$V/$v.so; [ $v = full ] && lib=intent_radio_sched_multi_slice_amd/csrc/libranenv_hip.so
  RANENV_LIB=$lib RANENV_PERSIST=0 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $out/${mode}_$v -o p --output-format csv -- python3 tools/profile_rollout.py 50 2 $mode > $out/${mode}_$v.log 2>&1 || { echo "failed $mode $v"; tail -3 $out/${mode}_$v.log; }
done; done
python3 - $out <<'PY'
import csv, glob, re, sys
out = sys.argv[1]
for mode in ("stream", "gather"):
    res = {}
    for v in ("full", "diag3", "diag4", "diag7", "diag11"):
        tot = {"SQ_INSTS_VALU": 0.0, "SQ_INSTS_SALU": 0.0, "SQ_INSTS_LDS": 0.0}
        for f in glob.glob(f"{out}/{mode}_{v}/*counter_collection.csv") + glob.glob(f"{out}/{mode}_{v}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] in tot and re.search(r"ranenv_core_kernel\w*<0[,>]", r["Kernel_Name"]):
                    tot[r["Counter_Name"]] += float(r["Counter_Value"])
        res[v] = {k: x / 110.0 for k, x in tot.items()}          # 10 + 2 x 50 TTIs of the whole batch
    f = res["full"]
    print(f"{mode}: per TTI of 4096 envs: VALU {f['SQ_INSTS_VALU']/1e6:.2f} M  SALU {f['SQ_INSTS_SALU']/1e6:.2f} M  LDS {f['SQ_INSTS_LDS']/1e6:.2f} M")
    for v, name in (("diag7", "allocation"), ("diag3", "UE step"), ("diag4", "observation tail"), ("diag11", "masked half of the stream")):
        if res[v]["SQ_INSTS_VALU"] > 0:
            print(f"   {name:28s} VALU {(f['SQ_INSTS_VALU'] - res[v]['SQ_INSTS_VALU'])/1e6:6.2f} M   SALU {(f['SQ_INSTS_SALU'] - res[v]['SQ_INSTS_SALU'])/1e6:6.2f} M   LDS {(f['SQ_INSTS_LDS'] - res[v]['SQ_INSTS_LDS'])/1e6:6.2f} M", flush=True)
PY
