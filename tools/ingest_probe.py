"""Bandwidth of the channel-ingest kernel (ranenv_se_from_power): python tools/ingest_probe.py [n_tiles]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from intent_radio_sched_multi_slice_amd.workloads import quadriga_pool_from_power
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
U, R = 100, 135
g = torch.rand((n, R, U), dtype=torch.float64, device="cuda") * 1e-9
quadriga_pool_from_power(g, R)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(5):
    e0.record(); quadriga_pool_from_power(g, R); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
bytes_moved = g.numel() * 12
print(f"{n} tiles: {best:.3f} ms, {bytes_moved / best / 1e6:.0f} GB/s (8 B read + 4 B written per element; includes the output allocation)")
