"""Rollout time with the traffic drawn on the device: python tools/philoxprobe.py  (RANENV_LIB / RANENV_SE_MODE select the variant)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
wl, _ = make_bench_workload(2, torch.device("cuda", 0), traffic="philox")
env = wl.env
env.set_partitions(3); env.reset(); env.rollout(30); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); env.rollout(300); e1.record(); torch.cuda.synchronize()
print(f"philox rollout {e0.elapsed_time(e1) / 300 * 1e3:.1f} us per TTI", flush=True)
