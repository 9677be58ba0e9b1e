"""Copy the judged summaries of one tools/round6_measure.sh pass into profiles/r06_*:  python tools/collect_profiles_r6.py gpurun_out/<tag>"""
import csv, json, os, shutil, sys
src = sys.argv[1]
names = {"bench.log": "bench.py (default: --steps 200 --warmup 20)",
         "bench_driver.log": "bench.py --gpus 1 --steps 20 --warmup 5 (the driver's invocation)",
         "bench_persist.log": "RANENV_PERSIST=1 bench.py (the streaming headline as persistent launches too)",
         "bench_k10.log": "bench.py --steps 10 --warmup 3", "bench_k100.log": "bench.py --steps 100 --warmup 26",
         "rehearse2.log": "bench.py --gpus 2 --rehearse-on-one-gpu --traces 40 --trace-len 100 --steps 20 --warmup 5 (the plain multi-rank command: two ranks sharing one GPU over gloo, a plumbing rehearsal)",
         "bench_cfg1.log": "bench.py --config 1", "bench_cfg4.log": "bench.py --config 4", "bench_native.log": "bench.py --config native",
         "bench_native_nopack.log": "RANENV_PACK=0 bench.py --config native (one env per wave)",
         "bench_philox.log": "bench.py --traffic philox"}
lines = {}
for f, label in names.items():
    path = os.path.join(src, f)
    if os.path.exists(path):
        for ln in open(path):
            if ln.startswith("{"):
                lines[label] = json.loads(ln)
json.dump(lines, open("profiles/r06_bench_lines.json", "w"), indent=1)
for mode in ("stream", "gather"):
    ks = os.path.join(src, f"prof_{mode}/p_kernel_stats.csv")
    if os.path.exists(ks):        # this library's kernels only (torch's pool-generation kernels have kilobyte-long names)
        rows = open(ks).read().splitlines()
        open(f"profiles/r06_{mode}_kernel_stats.csv", "w").write("\n".join([rows[0]] + [r for r in rows[1:] if "ranenv_" in r]) + "\n")
    kt = os.path.join(src, f"prof_{mode}/p_kernel_trace.csv")
    if os.path.exists(kt):
        with open(f"profiles/r06_{mode}_kernel_launches.txt", "w") as out:
            out.write(f"# every launch of this library's kernels in `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 "
                      f"{'--only-gather' if mode == 'gather' else '--no-single-stream --no-gather'} --no-other-configs --no-cpu-baseline`, in start order: "
                      "duration ms, block threads, workgroups, kernel\n")
            for r in csv.DictReader(open(kt)):
                if "ranenv_" in r["Kernel_Name"]:
                    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                    out.write(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:9.3f}  {r['Workgroup_Size_X']:>4}  "
                              f"{int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):>6}  {name}\n")
    log = os.path.join(src, f"prof_{mode}.log")
    if os.path.exists(log):
        for ln in open(log):
            if ln.startswith("{"):
                lines_ = json.loads(ln)
                json.dump(lines_, open(f"profiles/r06_{mode}_profiled_bench_line.json", "w"), indent=1)
for a, b in (("r06_pmc.json", "r06_pmc.json"), ("pmc_stream_summary.txt", "r06_pmc_sq_stream.txt"), ("pmc_gather_summary.txt", "r06_pmc_sq_gather.txt"),
             ("pmc_native_summary.txt", "r06_pmc_sq_native.txt"), ("pmc_config1_summary.txt", "r06_pmc_sq_config1.txt"), ("steploop_timeline.txt", "r06_steploop_timeline.txt")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join("profiles", b))
kt20 = os.path.join(src, "prof_stream_k20/p_kernel_trace.csv")
if os.path.exists(kt20):          # what a 20-TTI block (the driver's --steps 20) spends outside the steady state
    import subprocess
    steady = None
    for ln in open(os.path.join(src, "bench.log")):
        if ln.startswith("{"):
            steady = json.loads(ln)["ms_per_step"] * 1e3
    out = subprocess.run([sys.executable, "tools/block_timeline.py", kt20, "20"] + ([f"{steady:.3f}"] if steady else []), capture_output=True, text=True).stdout
    open("profiles/r06_block_timeline_k20.txt", "w").write("# tools/block_timeline.py on `rocprofv3 --kernel-trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "
                                                           "--no-single-stream --no-gather --no-other-configs`\n" + out)
# the driver's command under rocprofv3 --kernel-trace --stats: this library's kernels + the bench line of that very run
ks20 = os.path.join(src, "prof_stream_k20/p_kernel_stats.csv")
if os.path.exists(ks20):
    rows = open(ks20).read().splitlines()
    open("profiles/r06_stream_k20_kernel_stats.csv", "w").write("\n".join([rows[0]] + [r for r in rows[1:] if "ranenv_" in r]) + "\n")
log20 = os.path.join(src, "prof_stream_k20.log")
if os.path.exists(log20):
    for ln in open(log20):
        if ln.startswith("{"):
            json.dump(json.loads(ln), open("profiles/r06_stream_k20_profiled_bench_line.json", "w"), indent=1)
print("profiles updated from", src, "->", sorted(lines))
