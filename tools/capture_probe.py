"""Does a partitioned env.step() survive stream capture (hipStreamQuery on a capturing stream)?  python tools/capture_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
dev = torch.device("cuda", 0)
outs = []
for mode in ("eager", "graph"):
    wl = make_mult_slice_workload(256, dev, n_scenarios=16, n_traces=8, trace_len=16, max_steps=1000)
    env = wl.env
    env.set_partitions(3)
    env.reset(); env.step(); torch.cuda.synchronize()
    if mode == "eager":
        for _ in range(6):
            env.step()
    else:
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
                env.step(); env.step()
        for _ in range(3):
            g.replay()
    torch.cuda.synchronize()
    outs.append((env.obs_inter.clone(), env.reward.clone(), env.views()["step_number"].clone()))
    env.close()
print("steps", outs[0][2][:4].tolist(), outs[1][2][:4].tolist())
print("capture ok, identical:", all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])))
