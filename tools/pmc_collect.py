"""Turn rocprofv3 --pmc passes of tools/profile_step.py into profiles/r03_pmc.json (what bench.py reads for `traffic` and
for the gather mode's issue-rate bound).

    python tools/pmc_collect.py <out.json> <batch> <config> stream:<dir>[,<dir>...] gather:<dir>[,<dir>...]

Per mode, per launch of the STEP kernel (ranenv_core_kernel<0> / ranenv_core_kernel_gather<0>; one launch per TTI on one
stream, the first two launches dropped as warm-up):
  hbm_bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes).  The factor 2 on FETCH_SIZE is the gfx950 correction
  of MI355X_MICROARCH.md (HBM section: TCC_EA0_RDREQ counts 128-B requests at 64 B), calibrated on the streaming kernel's
  own known byte count (the SE stream alone is 4*U*R bytes per env and dominates its reads).
  valu_insts_per_launch = SQ_INSTS_VALU (wave-instructions), salu likewise.
"""
import collections, csv, datetime, glob, json, re, sys


def per_launch(dirs, counter):
    vals = []
    for d in dirs:
        for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                kn = r["Kernel_Name"]
                if r["Counter_Name"] == counter and re.search(r"ranenv_core_kernel\w*<0[,>]", kn):
                    vals.append(float(r["Counter_Value"]))
    vals = vals[2:] if len(vals) > 4 else vals
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


out, batch, config = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
res = {"batch": batch, "config": config, "date": datetime.date.today().isoformat(),
       "note": "per launch of the STEP kernel = per TTI of the whole batch (one stream); FETCH_SIZE doubled per the gfx950 correction"}
for spec in sys.argv[4:]:
    mode, _, ds = spec.partition(":")
    dirs = ds.split(",")
    m = {}
    fetch, nf = per_launch(dirs, "FETCH_SIZE")
    write, nw = per_launch(dirs, "WRITE_SIZE")
    if fetch is not None and write is not None:
        m.update({"fetch_size_kib_raw": fetch, "write_size_kib": write, "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
                  "n_launches": min(nf, nw)})
    for ctr, key in (("SQ_INSTS_VALU", "valu_insts_per_launch"), ("SQ_INSTS_SALU", "salu_insts_per_launch"),
                     ("SQ_WAVES", "waves_per_launch"), ("SQ_WAVE_CYCLES", "wave_cycles_per_launch"),
                     ("SQ_WAIT_ANY", "wait_any_per_launch"), ("SQ_ACTIVE_INST_VALU", "active_inst_valu_per_launch")):
        v, _ = per_launch(dirs, ctr)
        if v is not None:
            m[key] = v
    res[mode] = m
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
