"""Per-kernel time on the bench workload (big SE pool, no cache reuse): python tools/benchprobe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intent_radio_sched_multi_slice_amd import _lib
from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
wl = make_mult_slice_workload(4096, torch.device("cuda", 0), policy=_lib.POLICY_MAPF, intra=_lib.INTRA_PF, n_traces=200, trace_len=200)
env = wl.env
env.reset()
for _ in range(30):
    env.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 300
e0.record()
for _ in range(K):
    env.step()
e1.record(); torch.cuda.synchronize()
r = [env.step_profiled() for _ in range(40)]
a = np.array([x["alloc"] for x in r]) * 1e3; c = np.array([x["core"] for x in r]) * 1e3
print(f"PF_KB={os.environ.get('RANENV_PF_KB', '0'):3s} step {e0.elapsed_time(e1) / K * 1e3:6.1f} us   alloc {np.median(a):5.1f}   core {np.median(c):5.1f}", flush=True)
