"""Copy the judged summaries of one tools/round_measure.sh pass into profiles/:  python tools/collect_profiles.py gpurun_out/<tag> r02"""
import json, os, shutil, sys
src, rnd = sys.argv[1], sys.argv[2]
names = {"bench.log": "bench.py (default: --steps 200 --warmup 20; rollout over 3 partitions)",
         "bench_driver.log": "bench.py --gpus 1 --steps 20 --warmup 5 (the driver's invocation)",
         "bench_single.log": "bench.py --partitions 1 (one launch per TTI on one stream)",
         "bench_philox.log": "bench.py --traffic philox", "bench_cfg1.log": "bench.py --config 1", "bench_cfg4.log": "bench.py --config 4"}
lines = {}
for f, label in names.items():
    path = os.path.join(src, f)
    if os.path.exists(path):
        for ln in open(path):
            if ln.startswith("{"):
                lines[label] = json.loads(ln)
json.dump(lines, open(f"profiles/{rnd}_bench_lines.json", "w"), indent=1)
ks = os.path.join(src, "prof/p_kernel_stats.csv")
if os.path.exists(ks):        # this library's kernels only (torch's pool-generation kernels have kilobyte-long names)
    rows = open(ks).read().splitlines()
    open(f"profiles/{rnd}_bench_kernel_stats.csv", "w").write("\n".join([rows[0]] + [r for r in rows[1:] if "ranenv_" in r]) + "\n")
for a, b in (("pmc_sq_summary.txt", f"{rnd}_pmc_sq_summary.txt"), ("pmc_traffic.json", f"{rnd}_pmc_traffic.json")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join("profiles", b))
print("profiles updated from", src, "->", sorted(lines))
