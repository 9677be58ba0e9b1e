"""Diagnostic: per-phase s_memtime shares of wave 0 (needs RANENV_LIB=tools/diag9.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intent_radio_sched_multi_slice_amd import _lib
from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
wl = make_mult_slice_workload(4096, torch.device("cuda", 0), policy=_lib.POLICY_MAPF, intra=_lib.INTRA_PF,
                              n_traces=100, trace_len=100)
env = wl.env
env.reset()
for _ in range(30):
    env.step()
torch.cuda.synchronize()
st = env.views()["policy_scores"].cpu().numpy()[:, :9]
names = ["P0 load+barrier", "P1 policy", "P2 inter", "P3 intra", "P4 row loop", "P5 ue step", "P6 obs", "P7 reward"]
d = np.diff(st, axis=1)
print("median total cycles (s_memtime ticks @100MHz?)", np.median(st[:, 8]))
for n, col in zip(names, d.T):
    print(f"{n:18s} median {np.median(col):9.0f}  p90 {np.percentile(col, 90):9.0f}  share {np.median(col)/np.median(st[:,8])*100:5.1f}%")
