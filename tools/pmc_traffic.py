"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <batch> <config> <out.json>

HBM bytes per TTI = sum over the step's kernels of (2 * FETCH_SIZE + WRITE_SIZE) KiB.
The factor 2 on FETCH_SIZE is the gfx950 correction of MI355X_MICROARCH.md (HBM section:
TCC_EA0_RDREQ counts 128-B requests at 64 B); it is calibrated here on this kernel's own
known byte count: the SE stream alone is 4*U*R bytes per env and dominates the reads.
"""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "ranenv_" in kn and r["Counter_Name"] == counter and "<2," not in kn:
                agg[kn.split("ranenv_")[1].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v[2:]) / len(v[2:]) for k, v in agg.items() if len(v) > 2}


fetch_dir, write_dir, batch, config, out = sys.argv[1:6]
fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
kib = 1024.0
total = sum(2.0 * v * kib for v in fetch.values()) + sum(v * kib for v in write.values())
import datetime
json.dump({"batch": int(batch), "config": int(config), "date": datetime.date.today().isoformat(), "hbm_bytes_per_launch": total,
           "fetch_size_kib_raw": fetch, "write_size_kib": write,
           "note": "per TTI (all kernels of one step); FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 correction"},
          open(out, "w"), indent=1)
print(json.dumps({"hbm_bytes_per_launch": total, "fetch": fetch, "write": write}))
