"""Workgroup timeline of one launch of the step kernel from its s_memrealtime stamps (build -DRANENV_DIAG=9, run with
RANENV_LIB=tools/variants/stamps.so; RANENV_SE_MODE / RANENV_COMPACT / RANENV_LATE select the variant): when workgroups
enter and end (us from the first entry), how many are resident per CU over the launch, the median workgroup lifetime and
the per-phase medians (100 MHz ticks -> us)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
wl, _ = make_bench_workload(2, torch.device("cuda", 0), n_traces=100, trace_len=100)
env = wl.env
env.reset()
for _ in range(30):
    env.step()
torch.cuda.synchronize()
st = env.views()["policy_scores"].cpu().numpy()[:, :8] * 0.01      # us
st -= st[:, 0].min()
ent, end = st[:, 0], st[:, 7]
pc = [1, 10, 25, 50, 75, 90, 99, 100]
print("percentiles          ", pc)
print("workgroup entries us:", np.round(np.percentile(ent, pc), 1).tolist())
print("workgroup ends us:   ", np.round(np.percentile(end, pc), 1).tolist())
ts = np.linspace(0, end.max(), 25)
print("resident per CU over the launch:", [float(round((((ent <= t).sum() - (end <= t).sum()) / 256), 1)) for t in ts])
names = ["entry", "allocation", "SE stream / gather", "barrier", "UE step", "barrier", "obs tail"]
print("phase medians us:", {n: round(float(np.median(st[:, k + 1] - st[:, k])), 2) for k, n in enumerate(names)})
print("workgroup lifetime median us %.1f, launch span us %.1f" % (float(np.median(end - ent)), float(end.max())))
