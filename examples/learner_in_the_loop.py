"""A learner in the loop on one MI355X: a small torch policy acts on every TTI's observation of every env.

    python examples/learner_in_the_loop.py [--batch 4096] [--ttis 400] [--ranges 1] [--se-mode gather]

What the reference does with RLlib's PPO and 10 concurrent env runners (simu.py:555-566, agents/ray_agent.py:296-300) in
miniature: the inter-slice agent ("player_0") is a masked diagonal Gaussian over the S slices
(agents/masked_action_distribution.py:30-36) whose mean comes from an MLP on the 10*S inter-slice observation
(agents/ib_sched.py:160-173); PF runs inside the slices; the policy is improved by a plain policy-gradient step every
`--update-every` TTIs on the rewards collected meanwhile.  The batch is cut into ranges that are stepped alternately
(BatchedRanEnv.set_ranges / range_stream / step_wait / step_async): while the policy network runs on one range's
observations, the other range's TTI occupies the GPU.  Episode ends are handled on the device (enable_autoreset).

This is an example of the API, not a tuned trainer: it prints env-steps/s with the policy in the loop and the mean
inter-slice reward as training goes on.  Measured on one MI355X (B = 4096, gather mode): 3-7 M env-steps/s with one range
(it depends on the box's host cores), about half of that with two -- the loop is bound by the HOST: an eager torch policy is
~15 small kernels per decision (250-600 us of Python and launch time per range and TTI against ~45 us of env step), and every
further range adds that much host work per TTI.
Ranges pay off once the policy costs the host little (a fused or graph-captured forward) AND the env steps in the streaming
mode: `bench.py`'s `pipelined_step` runs the same schedule with a one-kernel policy at 56 M env-steps/s against 49 M for plain
`env.step()`.  In the gather mode the plain `env.step()` on one stream is the faster schedule since round 4 (a whole-batch step is
one launch of mixed blocks, the whole batch resident in one round: 45.5 us per TTI against 47.8 for two ranges,
tools/pipeprobe.py) -- one range, as this example defaults to.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from intent_radio_sched_multi_slice_amd._lib import INTRA_PF, POLICY_EXTERNAL
from intent_radio_sched_multi_slice_amd.adapters import masked_gaussian_params, sorted_action_mask
from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--ttis", type=int, default=400)
    ap.add_argument("--ranges", type=int, default=1)
    ap.add_argument("--episode-len", type=int, default=100)
    ap.add_argument("--update-every", type=int, default=20)
    ap.add_argument("--se-mode", choices=("stream", "gather"), default="gather")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, T = args.batch, args.episode_len
    n_ep = 64
    wl = make_mult_slice_workload(B, dev, policy=POLICY_EXTERNAL, intra=INTRA_PF, n_scenarios=n_ep, n_traces=n_ep, trace_len=T,
                                  max_steps=T)
    env = wl.env
    S = env.S
    env.set_se_mode(args.se_mode)
    ep = np.arange(n_ep)
    env.set_episode_table(scenario=ep, se_base=ep * T, se_len=T, trf_base=ep * T, trf_len=T)
    env.enable_autoreset(0, n_ep, random_episodes=True, seed=7, episode_numbers=np.arange(B) % n_ep)
    ranges = env.set_ranges(args.ranges)
    streams = [env.range_stream(k) for k in range(args.ranges)]

    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)      # forwards run on the ranges' streams by design
    torch.manual_seed(0)
    policy = torch.nn.Sequential(torch.nn.Linear(10 * S, 64), torch.nn.Tanh(), torch.nn.Linear(64, S)).to(dev)
    log_std = torch.nn.Parameter(torch.full((S,), -0.5, device=dev))
    opt = torch.optim.Adam(list(policy.parameters()) + [log_std], lr=3e-4)
    scores = torch.zeros((B, S), dtype=torch.float64, device=dev)
    mask_view = env.views()["mask_inter"]

    def act(k):
        """Scores of range k from its last observation, on the current (= the range's) stream; returns log-prob."""
        lo, hi = ranges[k]
        # CLONE what the autograd graph keeps: env.obs_inter and the mask are overwritten in place by the next TTI's kernel
        # (no version counter sees a HIP kernel's stores), and backward() runs update_every TTIs later -- a saved zero-copy view
        # would backpropagate every transition with the latest observation instead of its own.
        obs = env.obs_inter[lo:hi].clone()
        mask = mask_view[lo:hi].clone()
        mean, std = masked_gaussian_params(torch.tanh(policy(obs)), log_std.expand(hi - lo, S), sorted_action_mask(mask))
        eps = torch.randn_like(mean)
        scores[lo:hi].copy_(torch.clamp(mean.detach() + std.detach() * eps, -1.0, 1.0))     # float32 in, float64 scores out
        return -(0.5 * eps * eps + torch.log(std)).sum(dim=1)      # log-density of the draw, up to a constant

    env.reset()
    torch.cuda.synchronize()
    pending = [None] * args.ranges          # (log-prob of the action in flight) per range
    for k in range(args.ranges):
        with torch.cuda.stream(streams[k]):
            pending[k] = act(k)
            env.step_async(k, scores)
    losses, rewards = [[] for _ in ranges], []
    t0 = time.perf_counter()
    for t in range(args.ttis):
        for k, (lo, hi) in enumerate(ranges):
            with torch.cuda.stream(streams[k]):
                obs, reward, done = env.step_wait(k)                     # the TTI in flight for this range is done
                r = reward[:, 0].to(torch.float32)
                losses[k].append(-(pending[k] * (r - r.mean())).mean())  # REINFORCE with a batch-mean baseline
                if k == 0:
                    rewards.append(r.mean())
                pending[k] = act(k)
                env.step_async(k, scores)                                # the range's next TTI (and its episode advance)
        if (t + 1) % args.update_every == 0:
            for s in streams:                                            # one optimiser step on everything collected
                torch.cuda.current_stream(dev).wait_stream(s)
            loss = torch.stack([torch.stack(l).mean() for l in losses]).mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            for s in streams:
                s.wait_stream(torch.cuda.current_stream(dev))
            losses = [[] for _ in ranges]
            pending = [p.detach() for p in pending]
            if (t + 1) % (5 * args.update_every) == 0:
                print(f"TTI {t + 1:5d}: mean inter-slice reward {torch.stack(rewards).mean().item():+.4f}", flush=True)
                rewards = []
    for k in range(args.ranges):
        with torch.cuda.stream(streams[k]):
            env.step_wait(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{B} envs x {args.ttis} TTIs with the policy in the loop ({args.ranges} ranges, SE mode {args.se_mode}): "
          f"{B * args.ttis / dt / 1e6:.2f} M env-steps/s, {dt / args.ttis * 1e6:.0f} us per TTI")
    env.close()


if __name__ == "__main__":
    main()
