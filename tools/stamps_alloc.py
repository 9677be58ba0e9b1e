"""Diagnostic: s_memtime phase stamps of the alloc kernel (build with -DRANENV_DIAG=8, run with
RANENV_LIB=tools/diag8.so): ticks since the first stamp, per phase."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intent_radio_sched_multi_slice_amd import _lib
from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
wl = make_mult_slice_workload(4096, torch.device("cuda", 0), policy=_lib.POLICY_MAPF, intra=_lib.INTRA_PF)
env = wl.env
env.reset()
for _ in range(30):
    env.step()
torch.cuda.synchronize()
st = env.views()["policy_scores"].cpu().numpy()[:, :7]
st[:, 0] = st[:, 1]                       # slot 0 is unused; slot 1 is the first stamp (= 0)
names = ["(unused)", "1-2 rows written (slot / state loads landed)", "2-3 barrier", "3-4 inter-slice part",
         "4-5 barrier", "5-6 intra-slice part"]
tot = st[:, 6] - st[:, 1]
print("block lifetime ticks: median %.0f p10 %.0f p90 %.0f" % (np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
for k, n in enumerate(names):
    d = st[:, k + 1] - st[:, k]
    print(f"{n:46s} median {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}  share {100 * np.median(d) / np.median(tot):5.1f}%")
