"""Diagnostic: s_memtime phase stamps of the step kernel (build with -DRANENV_DIAG=9, run with
RANENV_LIB=tools/variants/stamps.so).  Thread 0 of every workgroup stamps (100 MHz counter):
0 entry / 1 tables parked + first barrier / 2 allocation done / 3 stream done / 4 after the barrier /
5 UE step done / 6 after the barrier / 7 end of the obs tail."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
wl, _ = make_bench_workload(2, torch.device("cuda", 0), n_traces=100, trace_len=100)
env = wl.env
env.reset()
parts = int(os.environ.get("RANENV_PARTS", "1"))
if parts > 1:
    env.set_partitions(parts)
    env.rollout(60)
    print(f"(rollout over {parts} partitions)")
else:
    for _ in range(30):
        env.step()
torch.cuda.synchronize()
st = env.views()["policy_scores"].cpu().numpy()[:, :8] * 0.01      # us
t0 = st[:, 0].min()
print("kernel span (first entry -> last end) %.1f us" % (st[:, 7].max() - t0))
print("block entry times: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(st[:, 0] - t0, [10, 50, 90, 100])))
names = ["entry: scalars, loads issued, tables parked", "allocation (inter + intra)", "SE stream", "barrier wait",
         "UE step", "barrier wait", "obs tail"]
for k, n in enumerate(names):
    d = st[:, k + 1] - st[:, k]
    print(f"{n:44s} median {np.median(d):7.2f}  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f}")
d = st[:, 7] - st[:, 0]
print(f"{'block lifetime':44s} median {np.median(d):7.2f}  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f}")
