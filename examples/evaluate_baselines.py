"""The reference's baseline comparison (MARR + round-robin against MAPF + proportional fairness, `simu.py` test mode over the
`mult_slice` scenarios; the violation / distance figures of results/gen_results.py:874-1022) run for a whole batch of
environments on one MI355X, episode ends and metric sums on the device:

    python examples/evaluate_baselines.py [--batch 1024] [--episodes 4] [--steps 1000]

Every env plays `episodes` consecutive episodes (scenario = episode number mod n_scenarios like
associations/mult_slice.py:444-452, its own channel trace), all TTIs of all episodes in ONE ranenv_rollout call per agent.
Both agents see the same exogenous inputs (same seeds), as the reference requires of a fair comparison
(results/gen_results.py:1587-1635).  Synthetic scenarios / channels of the reference's laws: the real datasets are not shipped.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from intent_radio_sched_multi_slice_amd._lib import INTRA_PF, INTRA_RR, POLICY_MAPF, POLICY_MARR
from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--episodes", type=int, default=4)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--scenarios", type=int, default=200)
    ap.add_argument("--traces", type=int, default=50)
    ap.add_argument("--se-mode", choices=("stream", "gather"), default="gather",
                    help="how replayed SE tiles are read (gather: per-tile mean-SE sidecar + the allocated RBs only; same results)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, n_ep, T = args.batch, args.episodes, args.steps
    rows = []
    for name, policy, intra in (("marr + rr", POLICY_MARR, INTRA_RR), ("mapf + pf", POLICY_MAPF, INTRA_PF)):
        wl = make_mult_slice_workload(B, dev, policy=policy, intra=intra, n_scenarios=args.scenarios, n_traces=args.traces,
                                      trace_len=T, max_steps=T)
        env = wl.env
        env.set_se_mode(args.se_mode)
        # episode number n: scenario n mod n_scenarios, channel trace n mod n_traces, traffic trace of its scenario
        first, count = 0, args.scenarios * 4
        ep = np.arange(first, first + count)
        env.set_episode_table(scenario=ep % args.scenarios, se_base=(ep % args.traces) * T, se_len=T, se_offset=0,
                              trf_base=(ep % args.scenarios) * T, trf_len=T, trf_offset=0, first_episode=first)
        env.enable_autoreset(first, first + count, episode_numbers=first + (np.arange(B) * n_ep) % count)
        env.enable_metrics(n_ep)
        env.set_partitions(3 if B >= 2048 else 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = env.evaluate(n_ep)
        dt = time.perf_counter() - t0
        ttis = res["ttis"].sum()
        rows.append((name, res["reward"].sum() / ttis, res["violations"].sum() / ttis, res["priority_violations"].sum() / ttis,
                     res["distance"].sum() / ttis, res["priority_distance"].sum() / ttis,
                     res["pkts_dropped"].sum() / max(1.0, res["pkts_dropped"].sum() + res["pkts_sent"].sum()), ttis / dt))
        env.close()
    print(f"{B} envs x {n_ep} episodes x {T} TTIs per agent")
    print(f"{'agent':12s} {'reward/TTI':>11s} {'violations':>11s} {'prio viol.':>11s} {'distance':>10s} {'prio dist.':>11s} "
          f"{'drop rate':>10s} {'env-steps/s':>12s}")
    for r in rows:
        print(f"{r[0]:12s} {r[1]:11.4f} {r[2]:11.4f} {r[3]:11.4f} {r[4]:10.4f} {r[5]:11.4f} {r[6]:10.4f} {r[7]:12.3e}")


if __name__ == "__main__":
    main()
