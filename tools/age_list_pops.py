"""How many age-list entries a UE consumes per TTI in the bench workload (each one is a dependent load in the UE step): the CPU oracle's
age histograms of a sample of envs, bins emptied per TTI = non-empty bins before + (a bin admitted) - non-empty bins after.
    python tools/age_list_pops.py [config] [n_envs] [ttis]        (GPU box: the workload is built on the device)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
from oracle import pyoracle

config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ttis = int(sys.argv[3]) if len(sys.argv) > 3 else 120
wl, label = make_bench_workload(config, torch.device("cuda", 0), batch=256, n_traces=32, trace_len=ttis + 1)
env = wl.env
S, U, R = env.S, env.U, env.R
cfg = pyoracle.make_cfg(S, U, R, env.G, env.Us, bandwidth_hz=env.bandwidth_hz, max_age_cap=env.max_age_cap, max_steps=env.max_steps)
eps, L = env.episodes, wl.trace_len
se_all = wl.se_pool.transpose(1, 2).contiguous().cpu().numpy()
trf = wl.traffic_pool.cpu().numpy().astype(np.float64)
intra = np.full(S, wl.intra, dtype=np.int32)
score = (lambda o: o.policy_mapf()) if wl.policy == 2 else (lambda o: o.policy_marr())
hist_all, wave_max = np.zeros(16, dtype=np.int64), np.zeros(16, dtype=np.int64)
for b in range(n_envs):
    o = pyoracle.OracleEnv(cfg)
    o.set_scenario(wl.tables, int(wl.scenario[b]))
    tile = lambda t: se_all[int(eps["se_base"][b] + (eps["se_offset"][b] + t) % L)]
    o.reset(tile(0))
    nz_prev = np.zeros(U, dtype=np.int64)
    for t in range(ttis):
        bits = trf[int(eps["trf_base"][b] + (eps["trf_offset"][b] + t) % L)]
        o.step(score(o), intra, tile(t), bits)
        raw = o.raw()
        nz = np.array([(o.buffer(u) > 0).sum() for u in range(U)])
        admitted = (raw["pkt_incoming"] > 0).astype(np.int64)          # (a full buffer admits nothing: rare, counts one pop too many)
        pops = np.maximum(nz_prev + admitted - nz, 0)
        nz_prev = nz
        if t < 20:
            continue                                                     # warm-up: queues fill
        for p in pops:
            hist_all[min(int(p), 15)] += 1
        for w0 in range(0, U, 64):
            wave_max[min(int(pops[w0:w0 + 64].max()), 15)] += 1
tot = hist_all.sum()
print(label)
print("entries consumed per UE and TTI:  " + "  ".join(f"{k}: {100.0 * hist_all[k] / tot:.1f} %" for k in range(8)))
print("the most of any lane of a wave:   " + "  ".join(f"{k}: {100.0 * wave_max[k] / wave_max.sum():.1f} %" for k in range(12)))
print("mean of that maximum: %.2f dependent loads per wave and TTI" % (sum(k * wave_max[k] for k in range(16)) / wave_max.sum()))
