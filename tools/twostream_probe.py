"""Two half-batches stepped on two streams against one full batch on one stream: do the ramp and tail of one
half's launches hide under the other half's steady state?   python tools/twostream_probe.py [n_streams]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
dev = torch.device("cuda", 0)
K = 300
def run(n_parts):
    B = 4096 // n_parts
    wls = [make_bench_workload(2, dev, batch=B, n_traces=100, trace_len=100, rank=i)[0] for i in range(n_parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_parts)]
    for wl, st in zip(wls, streams):
        with torch.cuda.stream(st):
            wl.env.reset()
            for _ in range(20):
                wl.env.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(K):
        for wl, st in zip(wls, streams):
            with torch.cuda.stream(st):
                wl.env.step()
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / K * 1e3
    print(f"{n_parts} part(s) of {B} envs on {n_parts} stream(s): {t:6.1f} us per TTI of all 4096 envs  (small_batch env: {os.environ.get('RANENV_SMALL_BATCH', 'auto')})", flush=True)
    for wl in wls:
        wl.env.close()
for n in (1, 2, 4):
    run(n)
