"""Summarise rocprofv3 counter_collection CSVs for the step kernel: python tools/pmc_summary.py DIR..."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "ranenv_" in kn:
                short = kn.split("ranenv_")[1].split("(")[0]
                agg[short + " " + r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            v = v[2:] if len(v) > 4 else v
            print(f"{d:28s} {k:24s} avg={sum(v)/len(v):.5g} n={len(v)}")
