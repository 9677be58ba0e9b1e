"""env.step() in a loop with and without device auto-reset (what an RL trainer runs): us per TTI at the headline size.
    python tools/autoreset_probe.py [n_steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
wl, _ = make_bench_workload(2, torch.device("cuda", 0), n_traces=100, trace_len=100)
env = wl.env
def loop(label):
    env.reset()
    for _ in range(30): env.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): env.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label:60s} {dt / n * 1e6:7.2f} us per TTI", flush=True)
for rep in range(2):
    env.disable_autoreset() if rep else None
    loop("env.step() loop, no auto-reset")
    L = wl.trace_len
    ep = np.arange(64)
    env.set_episode_table(scenario=ep % wl.tables.n_scenarios, se_base=(ep % 100) * L, se_len=L, se_offset=0, trf_base=(ep % wl.tables.n_scenarios) * L, trf_len=L, trf_offset=0)
    env.set_max_steps(np.full(env.B, 100000, dtype=np.int32))
    env.enable_autoreset(0, 64, episode_numbers=np.arange(env.B) % 64)
    loop("env.step() loop + ranenv_autoreset every TTI (no episode ends)")
    env.set_max_steps(137 + (np.arange(env.B) % 64))
    loop("... with episodes of 137-200 TTIs ending all the time")
