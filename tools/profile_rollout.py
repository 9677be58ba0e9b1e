"""Driver for rocprofv3 counter passes of the schedule bench.py's headline actually runs: compact, fused rollouts over batch
partitions (ranenv_rollout), in either SE mode.
    python tools/profile_rollout.py <K> <calls> [stream|gather] [config] [persist]
Prints `ttis <total TTIs of the whole batch, warm-up rollout included>`: tools/pmc_collect_r4.py divides the counters summed over
all STEP-kernel launches of the process by it (counters serialise the launches; bytes and instructions per launch stand)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = sys.argv[3] if len(sys.argv) > 3 else "stream"
config = int(sys.argv[4]) if len(sys.argv) > 4 else 2
persist = int(sys.argv[5]) if len(sys.argv) > 5 else 0
wl, _ = make_bench_workload(config, torch.device("cuda", 0), n_traces=100, trace_len=100)
env = wl.env
env.set_se_mode(mode)
if persist:
    env.set_option("persist", 1)
env.set_partitions(3 if env.B >= 2048 else 1)
env.reset()
warm = 10
env.rollout(warm)
for _ in range(calls):
    env.rollout(K)
torch.cuda.synchronize()
print("ttis", warm + K * calls, "batch", env.B, "mode", mode, "persist", persist, flush=True)
