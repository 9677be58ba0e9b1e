"""Per-call times and queue statistics of the persistent rollout: python tools/persist_probe.py [K] [calls]  (RANENV_* knobs apply)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B = int(os.environ.get("KPROBE_BATCH", "4096"))
wl, _ = make_bench_workload(2, torch.device("cuda", 0), batch=B)
env = wl.env
env.set_option("persist", 1)
env.set_partitions(3); env.reset(); env.rollout(30); torch.cuda.synchronize()
names = ("keep", "push", "pop", "fresh", "idle_polls")
prev = {k: env.get_option("persist_stat_" + k) for k in names}
for i in range(calls):
    torch.cuda.synchronize(); t0 = time.perf_counter(); env.rollout(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    cur = {k: env.get_option("persist_stat_" + k) for k in names}
    print(f"call {i}: {dt / K * 1e6:7.1f} us per TTI  " + "  ".join(f"{k} {cur[k] - prev[k]}" for k in names), flush=True)
    prev = cur
env.profile_begin(); env.rollout(K); k = env.profile_end()
print("launches", k, flush=True)
