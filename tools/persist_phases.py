"""Where a TTI of a PERSISTENT launch goes: per-phase time of thread 0 of every workgroup, accumulated over the TTIs of a rollout
(build with -DRANENV_DIAG=12, run with RANENV_LIB=<that build>):
    python tools/persist_phases.py <config: 1 | 2 | 4 | native> <K> [gather]
Phases: gap = between two TTIs of a chunk (barrier + loop); 1 entry (loads requested, tables parked, first barrier); 2 allocation;
3 SE stream / gather (+ the rest of the state in the builds that request it there); 4 barrier; 5 UE step; 6 barrier;
7 observation tail; 8 tail / hand-over."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
cfg = sys.argv[1] if len(sys.argv) > 1 else "1"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wl, _ = make_bench_workload(5 if cfg == "native" else int(cfg), torch.device("cuda", 0), n_traces=100, trace_len=100)
env = wl.env
if len(sys.argv) > 3 and sys.argv[3] == "gather":
    env.set_se_mode("gather")
env.set_partitions(3 if env.B >= 2048 else 1)
env.reset()
env.rollout(20)
torch.cuda.synchronize()
env.views()["policy_scores"].zero_()
torch.cuda.synchronize()
import time
t0 = time.perf_counter(); env.rollout(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
assert env.get_option("last_rollout_persistent") == 1, "not a persistent rollout"
st = env.views()["policy_scores"].cpu().numpy()
n = st[:, 9]
us = st[:, :9] * 0.01 / np.maximum(n, 1)[:, None]
members = (wl.tables.ue_slice[wl.scenario] >= 0).sum(axis=1)
print(f"config {cfg}, B {env.B}, rollout({K}): {dt / K * 1e6:.2f} us per TTI of the batch (wall, incl. sync); TTIs stamped per env: {n.min():.0f}..{n.max():.0f}")
names = ["gap between TTIs", "entry", "allocation", "SE stream / gather", "barrier", "UE step", "barrier", "obs tail", "tail / hand-over"]
for label, sel in (("all envs", members >= 0), ("one-wave envs (<= 64 members)", members <= 64), ("two-wave envs", members > 64)):
    if sel.sum() == 0:
        continue
    print(f"-- {label}: {int(sel.sum())}")
    tot = 0.0
    for k, nm in enumerate(names):
        d = us[sel, k]
        tot += d.mean()
        print(f"   {nm:22s} mean {d.mean():6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f} us")
    print(f"   {'sum = a TTI of an env':22s} mean {tot:6.2f} us")
