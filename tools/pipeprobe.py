"""Learner-in-the-loop schedule alone (for rocprofv3 --kernel-trace): python tools/pipeprobe.py [ranges] [steps] [se_mode]
Prints us per TTI (host clock) and the host time spent enqueueing one TTI."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from intent_radio_sched_multi_slice_amd import _lib
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mode = sys.argv[3] if len(sys.argv) > 3 else "stream"
wl, _ = make_bench_workload(2, torch.device("cuda", 0), n_traces=100, trace_len=200)
env = wl.env
if mode == "gather":
    env.set_se_mode("gather")
env.set_policy(_lib.POLICY_EXTERNAL, wl.intra)
ranges = env.set_ranges(nr)
S = env.S
scores = torch.zeros((env.B, S), dtype=torch.float64, device=env.device)
p_in = [env.obs_inter[lo:hi].view(hi - lo, S, 10)[:, :, 0] for lo, hi in ranges]
p_out = [scores[lo:hi] for lo, hi in ranges]
own = os.environ.get("PIPE_OWN", "1") == "1"      # policy on the range's own stream (no cross-queue events)
rs = [env.range_stream(k) for k in range(nr)]
main = torch.cuda.current_stream()
env.reset(); torch.cuda.synchronize()
def block(n):
    for _ in range(n):
        for k in range(nr):
            if own: torch.cuda.set_stream(rs[k])
            env.step_wait(k); torch.tanh(p_in[k], out=p_out[k]); env.step_async(k, scores)
    torch.cuda.set_stream(main)
for k in range(nr):
    if own: torch.cuda.set_stream(rs[k])
    torch.tanh(p_in[k], out=p_out[k]); env.step_async(k, scores)
torch.cuda.set_stream(main)
block(20); torch.cuda.synchronize()
t0 = time.perf_counter(); block(K); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"ranges {nr} mode {mode} own_stream {own}: {(t2 - t0) / K * 1e6:.1f} us per TTI; host enqueue {(t1 - t0) / K * 1e6:.1f} us per TTI", flush=True)
# the same scores, one launch for the whole batch on the caller's stream (external scores, no ranges)
env.set_partitions(1)
def block1(n):
    for _ in range(n):
        torch.tanh(env.obs_inter.view(env.B, S, 10)[:, :, 0], out=scores); env.step(scores)
block1(20); torch.cuda.synchronize()
t0 = time.perf_counter(); block1(K); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"one launch per TTI, external scores: {(t2 - t0) / K * 1e6:.1f} us per TTI", flush=True)
