"""Per-kernel time of one TTI from hipEvents: python tools/kprobe.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intent_radio_sched_multi_slice_amd import _lib
from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wl = make_mult_slice_workload(batch, torch.device("cuda", 0), policy=_lib.POLICY_MAPF, intra=_lib.INTRA_PF)
env = wl.env
env.reset()
for _ in range(30):
    env.step()
torch.cuda.synchronize()
r = [env.step_profiled() for _ in range(60)]
a = np.array([x["alloc"] for x in r]) * 1e3
c = np.array([x["core"] for x in r]) * 1e3
print(f"{os.environ.get('RANENV_LIB', 'default'):18s} alloc {np.median(a):6.1f} us   core {np.median(c):6.1f} us (min {c.min():.1f})", flush=True)
