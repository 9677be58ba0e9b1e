"""Diagnostic: s_memtime phase stamps of the step kernel (build with -DRANENV_DIAG=9, run with
RANENV_LIB=tools/diag9.so).  Slots 0-4: wave 0 at entry / stream start / stream end /
UE step end / after the barrier; slots 5-8: the same for wave 1; slot 9: end of the obs tail."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intent_radio_sched_multi_slice_amd import _lib
from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
wl = make_mult_slice_workload(4096, torch.device("cuda", 0), policy=_lib.POLICY_MAPF, intra=_lib.INTRA_PF,
                              n_traces=100, trace_len=100)
env = wl.env
env.reset()
for _ in range(30):
    env.step()
torch.cuda.synchronize()
st = env.views()["policy_scores"].cpu().numpy()[:, :10] * 0.01      # us (100 MHz counter)
t0 = st[:, 0].min()
print("kernel span (first entry -> last end) %.1f us" % (st[:, 9].max() - t0))
print("block entry times: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(st[:, 0] - t0, [10, 50, 90, 100])))
names = ["w0 prologue", "w0 stream", "w0 ue step", "w0 barrier wait"]
for k, n in enumerate(names):
    d = st[:, k + 1] - st[:, k]
    print(f"{n:18s} median {np.median(d):7.2f}  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f}")
d = st[:, 9] - st[:, 4]
print(f"{'w0 obs tail':18s} median {np.median(d):7.2f}  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f}")
for k, n in enumerate(["w1 prologue", "w1 stream", "w1 ue step"]):
    d = st[:, k + 6] - st[:, k + 5]
    print(f"{n:18s} median {np.median(d):7.2f}  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f}")
d = st[:, 9] - st[:, 0]
print(f"{'block lifetime':18s} median {np.median(d):7.2f}  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f}")
