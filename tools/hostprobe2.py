import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from intent_radio_sched_multi_slice_amd import _lib
from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
dev = torch.device("cuda", 0)
wl = make_mult_slice_workload(4096, dev, policy=_lib.POLICY_MAPF, intra=_lib.INTRA_PF)
env = wl.env
side = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
with torch.cuda.stream(side):
    env.reset()
    for _ in range(20): env.step()
    side.synchronize()
    for K in (200, 1000):
        t0 = time.perf_counter()
        for _ in range(K): env.step()
        t1 = time.perf_counter()
        side.synchronize()
        t2 = time.perf_counter()
        print(f"side stream K={K}: enqueue {1e6*(t1-t0)/K:.1f} us/step, total {1e6*(t2-t0)/K:.1f} us/step", flush=True)
