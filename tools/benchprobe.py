"""Step time on the bench workload (big SE pool, no cache reuse): python tools/benchprobe.py [config] [batch]
RANENV_LIB selects the build.  Prints the device time per TTI over 300 steps and the per-kernel event average."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
batch = int(sys.argv[2]) if len(sys.argv) > 2 else None
kw = {}
if os.environ.get("RANENV_TRACES"):        # pool size: traces x trace_len tiles of 54 KB (small pools stay in L2 / MALL)
    kw = dict(n_traces=int(os.environ["RANENV_TRACES"]), trace_len=int(os.environ.get("RANENV_TRACE_LEN", "200")))
wl, _ = make_bench_workload(config, torch.device("cuda", 0), batch=batch, **kw)
env = wl.env
if os.environ.get("RANENV_METRICS") == "1":
    env.enable_metrics(0)
env.reset()
for _ in range(30):
    env.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 300
e0.record()
for _ in range(K):
    env.step()
e1.record(); torch.cuda.synchronize()
step_us = e0.elapsed_time(e1) / K * 1e3
env.profile_begin()
for _ in range(100):
    env.step()
k = env.profile_end()
parts = int(os.environ.get("RANENV_PARTS", "3"))
extra = ""
if parts > 1:
    env.set_partitions(parts)
    env.rollout(30); torch.cuda.synchronize()
    e0.record(); env.rollout(K); e1.record(); torch.cuda.synchronize()
    r = e0.elapsed_time(e1) / K * 1e3
    env.profile_begin(); env.rollout(100); kk = env.profile_end()
    extra = f"   rollout x{parts} {r:6.1f} us/TTI (launch avg {kk['step'] * 1e3:6.1f})"
print(f"{os.path.basename(os.environ.get('RANENV_LIB', 'default')):14s} cfg {config} step {step_us:6.1f} us   kernel {k['step'] * 1e3:6.1f}{extra}", flush=True)
