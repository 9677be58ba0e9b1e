"""Phase stamps (build -DRANENV_DIAG=9: RANENV_LIB=tools/variants/stamps.so) of the LAST TTI of a rollout, any bench config:
    python tools/stamps_cfg.py [config] [ttis] [stream|gather]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ttis = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mode = sys.argv[3] if len(sys.argv) > 3 else "stream"
wl, _ = make_bench_workload(config, torch.device("cuda", 0), n_traces=100, trace_len=100)
env = wl.env
env.set_se_mode(mode)
env.set_partitions(3 if env.B >= 2048 else 1)
env.reset(); env.rollout(ttis); torch.cuda.synchronize()
st = env.views()["policy_scores"].cpu().numpy()[:, :8] * 0.01      # us
names = ["entry", "allocation", "SE stream / gather", "barrier", "UE step", "barrier", "obs tail"]
print(f"config {config} B {env.B} mode {mode}: TTI {ttis} of rollout({ttis}); phase medians us:",
      {n: round(float(np.median(st[:, k + 1] - st[:, k])), 2) for k, n in enumerate(names) if n != "barrier"},
      "lifetime median %.1f p90 %.1f" % tuple(np.percentile(st[:, 7] - st[:, 0], [50, 90])), flush=True)
