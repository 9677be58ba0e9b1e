"""Small driver for rocprofv3: reset + N steps of the headline workload (no CPU baseline)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from intent_radio_sched_multi_slice_amd import _lib
from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
wl = make_mult_slice_workload(batch, torch.device("cuda", 0), policy=_lib.POLICY_MAPF, intra=_lib.INTRA_PF,
                              n_traces=100, trace_len=100)
wl.env.reset()
for _ in range(n):
    wl.env.step()
torch.cuda.synchronize()
print("done", n)
