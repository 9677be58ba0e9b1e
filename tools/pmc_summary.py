"""Summarise rocprofv3 counter_collection CSVs per kernel: python tools/pmc_summary.py DIR...  (average over dispatches,
the first two of each kernel dropped as warm-up)."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "ranenv_" in kn:
                short = kn.split("ranenv_")[1].split("(")[0]
                agg[short + " " + r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            v = v[2:] if len(v) > 4 else v
            print(f"{k:44s} avg={sum(v)/len(v):.6g} n={len(v)}")
