"""One TTI per call, 200 calls without a host sync in between: env.step() against rollout(1) as persistent class launches.
python tools/step_probe.py   (RANENV_SE_MODE selects the SE mode)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
wl, _ = make_bench_workload(2, torch.device("cuda", 0))
env = wl.env
env.reset(); env.rollout(20); torch.cuda.synchronize()
def timed(fn, n=200, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e6)
    return np.median(ts)
env.set_partitions(1)
print("env.step() loop, one stream:        %.1f us per TTI" % timed(lambda: env.step()), flush=True)
env.set_option("persist", 1)
print("rollout(1) loop, persistent classes: %.1f us per TTI" % timed(lambda: env.rollout(1)), flush=True)
env.set_option("persist", 0)
print("rollout(1) loop, launch per TTI:     %.1f us per TTI" % timed(lambda: env.rollout(1)), flush=True)
env.set_partitions(3)
print("rollout(1) loop, 3 partitions:       %.1f us per TTI" % timed(lambda: env.rollout(1)), flush=True)
