"""step() (joined with the caller's stream every TTI) over 1..4 batch partitions:  python tools/steppart_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
wl, _ = make_bench_workload(2, torch.device("cuda", 0))
env = wl.env
env.reset()
K = 300
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for parts in (1, 2, 3, 4, 1):
    env.set_partitions(parts)
    for _ in range(30):
        env.step()
    torch.cuda.synchronize(); e0.record()
    for _ in range(K):
        env.step()
    e1.record(); torch.cuda.synchronize()
    print(f"step() over {parts} partition(s): {e0.elapsed_time(e1) / K * 1e3:6.1f} us/TTI", flush=True)
