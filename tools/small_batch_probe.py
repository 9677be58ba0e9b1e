"""Step latency of small batches with the caller's scores every TTI (an SB3 / RLlib trainer with a few hundred envs):
    python tools/small_batch_probe.py   -> us per env.step(scores, intra) for B = 64, 256, 1024, tiny_step 0 / 1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from intent_radio_sched_multi_slice_amd import _lib
from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
dev = torch.device("cuda", 0)
for B in (64, 256, 1024):
    wl = make_mult_slice_workload(B, dev, n_traces=50, trace_len=200)
    env = wl.env
    env.set_policy(_lib.POLICY_EXTERNAL, _lib.INTRA_PF)
    g = torch.Generator(device=dev); g.manual_seed(1)
    sc = torch.rand((B, env.S), generator=g, device=dev, dtype=torch.float64) * 2 - 1
    row = []
    for rep in range(2):
        for opt in (0, 1):
            env.set_option("tiny_step", opt)
            env.reset()
            for _ in range(50): env.step(sc)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 400
            for _ in range(n): env.step(sc)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            row.append(f"tiny_step={opt}: {dt / n * 1e6:6.2f} us")
    # and the GPU time alone: the same steps captured in a graph
    print(f"B {B:5d}  " + "  ".join(row), flush=True)
    env.close()
