"""How many 8-RB groups of its SE row a UE's allocated range touches, per numpy leaf (the gather mode's walk: gather_part): from the rb_start /
rb_count views of the bench workload after some TTIs.  python tools/rb_range_stats.py [config] [ttis]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ttis = int(sys.argv[2]) if len(sys.argv) > 2 else 60
wl, _ = make_bench_workload(cfg, torch.device("cuda", 0), n_traces=40, trace_len=100)
env = wl.env
env.set_se_mode("gather")
env.reset()
R = env.R
n2 = R // 2; n2 -= n2 % 8
leaves = [(0, R)] if R <= 128 else [(0, n2), (n2, R)]
members = (wl.tables.ue_slice[wl.scenario] >= 0).sum(axis=1)
hist_old, hist_new, tail_frac = [], [], []
for t in range(ttis):
    env.rollout(1)
    if t < 20:
        continue
    torch.cuda.synchronize()
    s = env.views()["rb_start"].cpu().numpy(); c = env.views()["rb_count"].cpu().numpy()
    for narrow in (True, False):
        sel = (members <= 64) if narrow else (members > 64)
        ss, cc = s[sel], c[sel]
        per_leaf = []
        for (b, e) in leaves:
            end = b + ((e - b) & ~7)
            lo = np.maximum(ss, b); hi = np.minimum(ss + cc, end)
            ng = np.where(lo < hi, ((hi - 1 - b) >> 3) - ((lo - b) >> 3) + 1, 0)
            per_leaf.append(ng.max(axis=1))        # per env: the lane with the most groups decides the wave's trips (one-wave envs; an upper bound for two-wave ones)
        per_leaf = np.stack(per_leaf, 1)
        want_tail = ((ss + cc) > (R & ~7)) & (cc > 0)
        old = np.ceil(per_leaf / 2).clip(min=1).sum(1) + want_tail.any(1)          # depth 2 per leaf, leaves one after the other (+ the tail's load)
        new = np.maximum(1, np.ceil((per_leaf.max(1) - 0) / 1)) + want_tail.any(1)   # first group of every leaf together, then one group per trip in the fullest leaf
        hist_old.append(old.mean()); hist_new.append(new.mean()); tail_frac.append(want_tail.any(1).mean())
print(f"config {cfg}: memory round trips of the gather per env and TTI (mean over envs and {ttis - 20} TTIs): leaves one after the other, 2 groups in flight: "
      f"{np.mean(hist_old):.2f}; first groups of all leaves together, then depth 1: {np.mean(hist_new):.2f}; envs whose wave loads the tail group: {np.mean(tail_frac):.2f}")
s = env.views()["rb_start"].cpu().numpy(); c = env.views()["rb_count"].cpu().numpy()
print("rb_count of UEs that got any RB: percentiles 50/90/99/max:", np.percentile(c[c > 0], [50, 90, 99]).tolist(), int(c.max()), " UEs with RBs per env:", float((c > 0).sum(1).mean()))
