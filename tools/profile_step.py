"""Small driver for rocprofv3: reset + N steps of a bench workload (no CPU baseline).
    python tools/profile_step.py [steps] [config] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
config = int(sys.argv[2]) if len(sys.argv) > 2 else 2
batch = int(sys.argv[3]) if len(sys.argv) > 3 else None
wl, _ = make_bench_workload(config, torch.device("cuda", 0), batch=batch, n_traces=100, trace_len=100)
wl.env.reset()
for _ in range(n):
    wl.env.step()
torch.cuda.synchronize()
print("done", n)
