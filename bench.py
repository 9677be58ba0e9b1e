#!/usr/bin/env python3
"""bench.py -- env-steps/s of the MI355X-native RAN-slicing env step.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one TTI of every env of the batch = one launch of the fused HIP kernel
(device MAPF policy -> inter-slice split -> PF intra-slice -> UEs.step -> intent observation
and reward).  Workload at N=1: BASELINE.json configs[2], the configuration the north_star's
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
    eps = env.episodes
    # tiles the sample touches, transposed to the oracle's UE-major layout
    tile_idx = np.stack([eps["se_base"][:n] + (eps["se_offset"][:n] + t) % L for t in range(sample_steps)])
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
            "sample": f"{n} envs x {sample_steps} TTIs of the same workload, oracle/ranenv_oracle.c, OpenMP over envs, {dt:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--config", type=int, default=2, choices=(1, 2),
                    help="BASELINE.json configs index: 1 = B1024 MARR+RR, 2 = B4096 MAPF+PF (default)")
    ap.add_argument("--traces", type=int, default=200)
    ap.add_argument("--trace-len", type=int, default=200)
    ap.add_argument("--cpu-envs", type=int, default=256)
    ap.add_argument("--cpu-steps", type=int, default=6000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from intent_radio_sched_multi_slice_amd import _lib
    from intent_radio_sched_multi_slice_amd.dist import gather_metrics, local_metrics, summarize
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    if args.config == 1:
        batch = 1024 if args.batch == 4096 else args.batch
        policy, intra, label = _lib.POLICY_MARR, _lib.INTRA_RR, "MARR inter-slice + round-robin intra-slice"
    else:
        batch, policy, intra = args.batch, _lib.POLICY_MAPF, _lib.INTRA_PF
        label = "MAPF inter-slice + PF intra-slice + ib_sched intent observation/reward"
    wl = make_mult_slice_workload(batch, device, policy=policy, intra=intra, n_traces=args.traces,
                                  trace_len=args.trace_len, rank=rank)
    env = wl.env
    env.reset()
    for _ in range(args.warmup):
        obs, reward, done = env.step()
    gather_metrics(local_metrics(reward, env.views(), done, 1))   # warm torch's reduction kernels / RCCL
    ev_start = torch.cuda.Event(enable_timing=True)
    ev_end = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_start.record()               # torch's current stream = the stream the kernels are launched on
    for i in range(args.steps):
        obs, reward, done = env.step()
    ev_end.record()
    vec = local_metrics(reward, env.views(), done, args.steps)
    gathered = gather_metrics(vec)      # the only collective: metrics, RCCL all_gather
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    step_ms = ev_start.elapsed_time(ev_end) / args.steps      # device time of one TTI (3 kernels)
    prof = [env.step_profiled() for _ in range(24)][4:]       # per-kernel HIP events, after the timed region
    kms = {k: float(np.mean([q[k] for q in prof])) for k in ("alloc", "core")}
    total_env_steps = batch * world * args.steps
    value = total_env_steps / elapsed

    if rank == 0:
        alg_bytes = env.algorithmic_bytes_per_env_step() * batch          # per TTI of the whole batch
        achieved = alg_bytes / (step_ms * 1e-3) / 1e9                     # whole step: alloc + core + obs
        # the dominant kernel (core) moves everything except the action term of SURVEY 8(d)
        core_bytes = (env.algorithmic_bytes_per_env_step() - env.S * 5) * batch
        core_gbs = core_bytes / (kms["core"] * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("batch") == batch and tj.get("config") == args.config:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "env-steps/s (batched TTIs) at mult_slice 10-slice/100-UE; 1->8 GPU scaling",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{args.config}]: mult_slice, 10 slices, 100 UEs, 135 RBGs, "
                                   f"batch {batch} per GPU, {label}; SE replayed from an HBM pool of "
                                   f"{args.traces}x{args.trace_len} float32 tiles, Poisson traffic pool",
                       "batch_per_gpu": batch, "global_batch": batch * world, "n_slices": env.S, "n_ues": env.U,
                       "n_rbs": env.R, "parallelism": f"episodes sharded over {world} GPU(s), metrics all_gather only"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "one TTI = ranenv_alloc_kernel + ranenv_core_kernel<STEP>",
                         "kernel_ms": step_ms,
                         "algorithmic_bytes_per_env_step": env.algorithmic_bytes_per_env_step(),
                         "dominant_kernel": {"name": "ranenv_core_kernel<STEP>", "ms": kms["core"],
                                             "algorithmic_bytes": core_bytes, "achieved": core_gbs,
                                             "frac": core_gbs / HBM_PEAK_GBS},
                         "other_kernels_ms": {"alloc": kms["alloc"]}},
            "metrics": summarize(gathered.cpu()),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wl, args.cpu_envs, args.cpu_steps)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
