#!/usr/bin/env python3
"""bench.py -- env-steps/s of the MI355X-native RAN-slicing env step.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one TTI of every env of the batch (device MAPF policy -> inter-slice split -> PF
intra-slice -> UEs.step -> intent observation and reward).  The K timed steps are enqueued with
``ranenv_rollout`` over 3 batch partitions (ranges of independent envs, each stepped by its own launch on its own HIP
stream, joined with the caller's stream before the first and after the last TTI); the same K steps as one launch per
TTI on one stream are timed right after and reported beside it (``single_stream``).  Workload at N=1: BASELINE.json configs[2], the configuration the north_star's
throughput target is quoted on (mult_slice, 10 slices / 100 UEs / 135 RBGs, batch 4096, PF
intra-slice + ib_sched intent reward); with N GPUs every rank steps its own 4096 envs (weak
scaling; N=8 is configs[3], batch 32768 sharded 8x).  Inputs (scenario, SE and traffic pools)
are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def usable_cores() -> int:
    """Host threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(wl, sample_envs: int, sample_steps: int):
    """Time the CPU oracle (oracle/ranenv_oracle.c, OpenMP over envs) on a bounded sample of the
    same workload: the first `sample_envs` envs of this rank for `sample_steps` TTIs."""
    import torch
    from oracle import pyoracle
    env = wl.env
    S, U, R, G, Us = env.S, env.U, env.R, env.G, env.Us
    cores = usable_cores()
    n = min(sample_envs, env.B)
    cfg = pyoracle.make_cfg(S, U, R, G, Us, bandwidth_hz=env.bandwidth_hz, max_age_cap=env.max_age_cap,
                            max_steps=env.max_steps)
    oenvs = []
    for b in range(n):
        e = pyoracle.OracleEnv(cfg)
        e.set_scenario(wl.tables, int(wl.scenario[b]))
        oenvs.append(e)
    L = wl.trace_len
    W = min(L, 64)              # the timing sample replays a window of W tiles per env (host memory: n * W * 54 KB)
    eps = env.episodes
    # tiles the sample touches, transposed to the oracle's UE-major layout
    tile_idx = np.stack([eps["se_base"][:n] + (eps["se_offset"][:n] + t % W) % L for t in range(sample_steps)])
    uniq, inv = np.unique(tile_idx, return_inverse=True)
    inv = inv.reshape(tile_idx.shape)
    se_host = wl.se_pool[torch.as_tensor(uniq, device=env.device)].transpose(1, 2).contiguous().cpu().numpy()
    trf_host = wl.traffic_pool.cpu().numpy().astype(np.float64)
    intra = np.full((n, S), wl.intra, dtype=np.int32)
    pyoracle.batch_reset(oenvs, se_host, inv[0], cores)
    t0 = time.perf_counter()
    for t in range(sample_steps):
        rows = eps["trf_base"][:n] + (eps["trf_offset"][:n] + t) % L
        pyoracle.batch_step(oenvs, wl.policy, None, intra, se_host, inv[t], trf_host[rows], cores)
    dt = time.perf_counter() - t0
    return {"value": n * sample_steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} envs x {sample_steps} TTIs of the same workload (each env cycling through {W} tiles of its trace), "
                      f"oracle/ranenv_oracle.c, OpenMP over envs, {dt:.2f} s"}


def rank_env():
    """(world, rank, local_rank) from the launcher's environment (torch.distributed.run)."""
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def timed_steps(step_fn, n_steps: int, sync_fn, barrier_fn, max_over_ranks_fn):
    """EXACTLY n_steps calls of step_fn bracketed by barrier + device sync on both sides; wall-clock
    seconds, MAX over ranks.  Shared with tests/test_bench_logic.py (gloo, stub env)."""
    sync_fn(); barrier_fn(); sync_fn()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step_fn()
    sync_fn(); barrier_fn(); sync_fn()
    return max_over_ranks_fn(time.perf_counter() - t0)


def build_line(args, world, batch, label, env_sizes, alg_bytes_env_step, elapsed, launch_ms, n_launches, parts, traffic, metrics,
               workload_extra="", single_stream=None):
    """The ONE JSON line.  `value` and `roofline.frac` share the wall clock of the timed region (the whole TTI of the
    whole batch); the step kernel's per-launch duration (the dispatch's own timestamps, from the same K steps repeated
    with timing on) is listed beside it -- with partitions `parts` launches of batch/parts envs overlap."""
    S, U, R = env_sizes
    total_env_steps = batch * world * args.steps
    value = total_env_steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    alg_bytes = alg_bytes_env_step * batch                       # per TTI of one rank's batch
    achieved = alg_bytes / (ms_per_step * 1e-3) / 1e9            # per-GPU GB/s on the wall clock
    launch_bytes = alg_bytes / parts
    launch_gbs = launch_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    line = {
        "metric": "env-steps/s (batched TTIs) at mult_slice 10-slice/100-UE; 1->8 GPU scaling",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[{args.config}]: {label}; batch {batch} per GPU; SE replayed from "
                               f"an HBM pool of {args.traces}x{args.trace_len} float32 tiles{workload_extra}",
                   "batch_per_gpu": batch, "global_batch": batch * world, "n_slices": S, "n_ues": U,
                   "n_rbs": R, "parallelism": f"episodes sharded over {world} GPU(s), metrics all_gather only",
                   "launch": (f"ranenv_rollout: the K TTIs enqueued in one call, batch stepped as {parts} partitions on {parts} "
                              "HIP streams (one launch of the step kernel per partition and TTI)") if parts > 1 else
                             "one launch of the step kernel per TTI on one stream"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "clock": "wall clock of the timed region (same as value)",
                     "traffic": traffic.get("hbm_bytes_per_launch") if traffic else None,
                     "traffic_source": traffic.get("source") if traffic else None,
                     "kernel": "ranenv_core_kernel<STEP> (one TTI = %d concurrent launch(es) of it)" % parts,
                     "algorithmic_bytes_per_env_step": alg_bytes_env_step,
                     "algorithmic_bytes_per_tti": alg_bytes,
                     "dominant_kernel": {"name": "ranenv_core_kernel<STEP>", "ms": launch_ms, "n_launches": n_launches,
                                         "envs_per_launch": batch / parts, "algorithmic_bytes": launch_bytes,
                                         "achieved": launch_gbs, "frac": launch_gbs / HBM_PEAK_GBS,
                                         "concurrent_launches": parts,
                                         "source": "hipExtLaunchKernel start/stop events of every launch, the K steps repeated "
                                                   "right after the timed region under the same schedule"}},
        "metrics": metrics,
    }
    if single_stream is not None:
        line["single_stream"] = single_stream
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None, help="envs per GPU (default: the config's own)")
    ap.add_argument("--config", type=int, default=2, choices=(1, 2, 3, 4),
                    help="BASELINE.json configs index: 1 = B1024 MARR+RR, 2 = B4096 MAPF+PF (default; 3 = the same "
                         "per GPU, i.e. what --gpus 8 runs), 4 = mult_slice_seq sweep B8192, mixed masks")
    ap.add_argument("--traces", type=int, default=200)
    ap.add_argument("--trace-len", type=int, default=1000,
                    help="TTIs per channel trace (1000 = a whole episode: no env ever replays a tile; the pool is "
                         "traces x trace_len x 54 KB = 10.8 GB)")
    ap.add_argument("--traffic", choices=("pool", "philox"), default="pool",
                    help="offered traffic: replayed Poisson pool (parity mode) or the device counter-based generator")
    ap.add_argument("--partitions", type=int, default=None,
                    help="batch partitions on their own HIP streams (default: 3 on one GPU, 2 per rank in a multi-GPU run, "
                         "1 when the batch does not fill the CUs)")
    ap.add_argument("--cpu-envs", type=int, default=256)
    ap.add_argument("--cpu-steps", type=int, default=6000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-stream", action="store_true",
                    help="skip the one-launch-per-TTI comparison run (so that a rocprofv3 pass averages only the launches of "
                         "the measured schedule)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from intent_radio_sched_multi_slice_amd import _lib
    from intent_radio_sched_multi_slice_amd.dist import gather_metrics, local_metrics, summarize
    from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload

    world, rank, local_rank = rank_env()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the env step has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    wl, label = make_bench_workload(args.config, device, batch=args.batch, n_traces=args.traces,
                                    trace_len=args.trace_len, rank=rank, traffic=args.traffic)
    env = wl.env
    batch = env.B
    # partitions: 3 measured best on one GPU (caller's stream + 2); a process has 4 hardware queues and RCCL wants some
    # of them in a multi-rank run, so there 2 (caller's stream + 1; within 2 % of 3 on one GPU)
    parts = args.partitions if args.partitions is not None else ((3 if world == 1 else 2) if batch >= 2048 else 1)
    env.set_partitions(parts)
    env.reset()
    env.rollout(max(1, args.warmup))
    gather_metrics(local_metrics(env.reward, env.views(), env.done, 1))   # warm torch's reduction kernels / RCCL

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    barrier = dist.barrier if world > 1 else (lambda: None)
    # EXACTLY K steps: one ranenv_rollout call enqueues the K TTIs of every env
    elapsed = timed_steps(lambda: env.rollout(args.steps), 1, torch.cuda.synchronize, barrier, max_over_ranks)
    # metrics: the only collective, once per reporting interval, outside the timed K steps
    gathered = gather_metrics(local_metrics(env.reward, env.views(), env.done, args.steps))
    # the same K steps again with the dispatch timestamps of every launch
    env.profile_begin()
    env.rollout(args.steps)
    kms = env.profile_end()
    # ... and as one launch per TTI on one stream (what a caller that consumes every TTI's outputs gets)
    single = None
    if parts > 1 and not args.no_single_stream:
        env.set_partitions(1)
        el1 = timed_steps(env.step, args.steps, torch.cuda.synchronize, barrier, max_over_ranks)
        single = {"value": batch * world * args.steps / el1, "ms_per_step": el1 / args.steps * 1e3,
                  "roofline_frac": batch * env.algorithmic_bytes_per_env_step() / (el1 / args.steps) / (HBM_PEAK_GBS * 1e9),
                  "launch": "one launch of the step kernel per TTI on one stream (env.step() in a loop)"}

    if rank == 0:
        traffic = None
        tfile = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("batch") == batch and tj.get("config") == args.config:
                    traffic = {"hbm_bytes_per_launch": tj.get("hbm_bytes_per_launch"),
                               "source": "profiles/pmc_traffic.json (rocprofv3 --pmc passes of an earlier run of this "
                                         f"workload, {tj.get('date', 'undated')}; not measured in this run)"}
            except Exception:
                traffic = None
        line = build_line(args, world, batch, label, (env.S, env.U, env.R), env.algorithmic_bytes_per_env_step(),
                          elapsed, kms["step"], kms["n_launches"], parts, traffic, summarize(gathered.cpu()),
                          workload_extra=(", Poisson traffic pool" if args.traffic == "pool"
                                          else ", Poisson traffic drawn on the device (Philox4x32-10)"),
                          single_stream=single)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wl, args.cpu_envs, args.cpu_steps)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
