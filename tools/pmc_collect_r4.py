"""rocprofv3 --pmc passes -> profiles/r04_pmc.json.

    python tools/pmc_collect_r4.py <out.json> <batch> <config> <entry>:<ttis>:<dir>[,<dir>...][:<batch>:<config>] ...

An entry is one (schedule, SE mode): `stream_rollout`, `gather_rollout` (tools/profile_rollout.py: the headline's compact, fused
rollouts over 3 partitions; <ttis> = the TTIs of the whole batch that process stepped, printed by the driver) or `stream`, `gather`
(tools/profile_step.py: one full-width launch per TTI; <ttis> = its launches).  Per entry, over ALL launches of the STEP kernels
(ranenv_core_kernel*<0, ...>, any build) of the process:
  *_per_tti        = counter summed over the launches / ttis  (one TTI of the whole batch)
  ttis_per_launch  = ttis * partitions_seen / launches  is NOT derivable from counters; what is: launches and ttis, both listed
  hbm_bytes_per_tti = (2 * FETCH_SIZE + WRITE_SIZE) KiB -> bytes; the factor 2 on FETCH_SIZE is the gfx950 correction of
  MI355X_MICROARCH.md (HBM section: TCC_EA0_RDREQ counts 128-B requests at 64 B), calibrated in round 2 on the streaming kernel's
  own known byte count."""
import csv, datetime, glob, json, re, sys


def summed(dirs, counter):
    tot, n = 0.0, 0
    for d in dirs:
        for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == counter and re.search(r"ranenv_(core|persist)_kernel\w*<0[,>]|ranenv_persist_kernel|ranenv_core_kernel_packed", r["Kernel_Name"]):
                    tot += float(r["Counter_Value"]); n += 1
    return (tot, n) if n else (None, 0)


out, batch, config = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
res = {"batch": batch, "config": config, "date": datetime.date.today().isoformat(),
       "note": "counters summed over every STEP-kernel launch of the profiled process, divided by the TTIs of the whole batch it "
               "stepped; FETCH_SIZE doubled per the gfx950 correction; counter passes serialise launches (times do not stand, "
               "bytes and instructions do)"}
for spec in sys.argv[4:]:
    fields = spec.split(":")
    entry, ttis, ds = fields[0], fields[1], fields[2]
    ttis, dirs = float(ttis), ds.split(",")
    m = {"ttis": ttis}
    if len(fields) > 4:            # a block of another workload: entry:ttis:dirs:batch:config
        m.update({"batch": int(fields[3]), "config": int(fields[4])})
    fetch, nf = summed(dirs, "FETCH_SIZE")
    write, nw = summed(dirs, "WRITE_SIZE")
    if fetch is not None and write is not None:
        m.update({"fetch_size_kib_raw_per_tti": fetch / ttis, "write_size_kib_per_tti": write / ttis,
                  "hbm_bytes_per_tti": (2.0 * fetch + write) * 1024.0 / ttis, "launches": min(nf, nw),
                  "ttis_per_launch_and_partition": ttis * 3 / max(1, min(nf, nw)) if "rollout" in entry and m.get("batch", batch) >= 2048 else ttis / max(1, min(nf, nw))})
        m["hbm_bytes_per_launch"] = m["hbm_bytes_per_tti"]          # (the key bench.py's round-3 reader used: per TTI of the batch)
    for ctr, key in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_INSTS_SALU", "salu_insts"), ("SQ_WAVES", "waves"),
                     ("SQ_WAVE_CYCLES", "wave_cycles"), ("SQ_WAIT_ANY", "wait_any"), ("SQ_ACTIVE_INST_VALU", "active_inst_valu"),
                     ("SQ_INSTS_VMEM_RD", "vmem_rd_insts"), ("SQ_INSTS_VMEM_WR", "vmem_wr_insts"), ("SQ_INSTS_LDS", "lds_insts"),
                     ("SQ_BUSY_CYCLES", "sq_busy_cycles")):
        v, _ = summed(dirs, ctr)
        if v is not None:
            m[key + "_per_tti"] = v / ttis
    if "valu_insts_per_tti" in m:
        m["valu_insts_per_launch"] = m["valu_insts_per_tti"]
    res[entry] = m
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
