#!/usr/bin/env python3
"""bench.py -- env-steps/s of the MI355X-native RAN-slicing env step.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one TTI of every env of the batch (device MAPF policy -> inter-slice split -> PF
intra-slice -> UEs.step -> intent observation and reward).  Workload at N=1: BASELINE.json configs[2], the
configuration the north_star's throughput target is quoted on (mult_slice, 10 slices / 100 UEs / 135 RBGs, batch 4096,
PF intra-slice + ib_sched intent reward); with N GPUs every rank steps its own 4096 envs (weak scaling; N=8 is
configs[3], batch 32768 sharded 8x).  Inputs (scenario, SE and traffic pools) are resident in HBM before the timed
region.  Prints ONE JSON line on rank 0.

What is timed, each as blocks of EXACTLY K steps between barrier + device sync on both sides (MAX over ranks); a block
is repeated until ~20 ms of stepping are sampled, the median block is reported (`repeats`, min / max beside it):

  value / roofline   the headline, SURVEY 8(d)'s streaming step kernel: `ranenv_rollout` (the K TTIs of the device
                     policy enqueued in one call): launches of up to 10 TTIs over 3 batch partitions on 3 HIP streams
                     (RANENV_PERSIST=1: as persistent work-queue launches, a tie for this kernel; the line says which ran)
  single_stream      the same K TTIs as env.step() in a loop: one launch per TTI on one stream (mixed blocks: one block per env
                     of more than 64 slice members + one per two envs of at most 64, the whole batch resident in one round)
  pipelined_step     a learner in the loop: external inter-slice scores produced from each half's last observation
                     on that half's own stream, two half-batches stepped alternately (set_ranges / range_stream /
                     step_async / step_wait)
  se_gather          ranenv_rollout in the SE gather mode (ranenv_set_se_mode: per-tile mean-SE sidecar + reads of the
                     allocated RBs only) -- as persistent work-queue launches, one per workgroup class --, with its own byte
                     model and bound
  other_configs      (N = 1, default config only; --no-other-configs skips it) the other BASELINE configs as short blocks of the
                     same K steps, each under the schedule ranenv_rollout picks for it: "1" = configs[1] (B 1024, MARR + round-robin),
                     "4" = configs[4] (mult_slice_seq sweep, B 8192), "native" = the reference's own size (S 5 / U 25, B 16384, two
                     envs per wave); value, ms_per_step, roofline_frac against each config's own algorithmic bytes
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
N_SIMD, CLOCK_HZ, VALU_CYCLES = 256 * 4, 2.4e9, 4.0   # 256 CUs x 4 SIMDs, 2.4 GHz; a wave64 VALU instruction issues over >= 4 cycles
PMC_FILE = os.path.join("profiles", "r06_pmc.json")   # rocprofv3 --pmc passes of the schedules timed here (tools/profile_rollout.py, pmc_collect_r4.py, round6_measure.sh)
REFPY_FILE = os.path.join("profiles", "r04_reference_python.json")   # the reference's own Python, timed in the build container
SAMPLE_S = 0.020                                      # stepping time sampled per timed variant


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
    se_host = env.pooled_tiles(torch.as_tensor(uniq, device=env.device)).transpose(1, 2).contiguous().cpu().numpy()
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


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch_cmd(n_gpus: int, argv, port: int):
    """The child command of a plain `python bench.py --gpus N` (N > 1, no launcher environment): one rank per GPU under
    torch.distributed.run on 127.0.0.1, the same arguments passed through."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus: int, argv, popen=None) -> int:
    """Start the N ranks as a CHILD process (never exec: this must also work from a process that touched the GPU), relay
    rank 0's JSON line(s) on stdout and everything else on stderr, return the child's exit code.  Called before any
    torch.cuda / HIP call of this process.  `popen` is the test hook (tests/test_bench_logic.py)."""
    import subprocess
    cmd = self_launch_cmd(n_gpus, argv, free_port())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n_gpus)))
    print("bench.py: --gpus %d without a launcher environment: starting %s" % (n_gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    child = (popen or subprocess.Popen)(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in child.stdout:
        t = ln.strip()
        is_line = False
        if t.startswith("{") and t.endswith("}"):
            try:
                is_line = "metric" in json.loads(t)
            except ValueError:
                pass
        (sys.stdout if is_line else sys.stderr).write(ln)
        (sys.stdout if is_line else sys.stderr).flush()
    return child.wait()


def timed_steps(step_fn, n_steps: int, sync_fn, barrier_fn, max_over_ranks_fn, local_out=None):
    """EXACTLY n_steps calls of step_fn bracketed by barrier + device sync on both sides; wall-clock
    seconds, MAX over ranks.  Shared with tests/test_bench_logic.py (gloo, stub env).  `local_out` (a list) also
    receives this rank's own time up to its device sync, before the closing barrier: the per-GPU figure."""
    sync_fn(); barrier_fn(); sync_fn()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step_fn()
    sync_fn()
    if local_out is not None:
        local_out.append(time.perf_counter() - t0)
    barrier_fn(); sync_fn()
    return max_over_ranks_fn(time.perf_counter() - t0)


def timed_blocks(block_fn, sync_fn, barrier_fn, max_over_ranks_fn, sample_s: float = SAMPLE_S, max_repeats: int = 64, local_out=None):
    """block_fn() enqueues EXACTLY the K steps of one block.  One block is timed (bracketed as in timed_steps), then the
    block is repeated max(1, ceil(sample_s / that time)) times, each repeat bracketed and timed on its own; returns the
    list of per-block times (every rank derives the same repeat count from the MAX-over-ranks time)."""
    first = timed_steps(block_fn, 1, sync_fn, barrier_fn, max_over_ranks_fn)
    repeats = max(1, min(max_repeats, int(math.ceil(sample_s / max(first, 1e-9)))))
    return [timed_steps(block_fn, 1, sync_fn, barrier_fn, max_over_ranks_fn, local_out) for _ in range(repeats)]


def block_stats(times, env_steps_per_block: float, steps: int):
    """median / min / max of the per-block times as env-steps/s and ms per step."""
    med = float(np.median(times))
    return {"value": env_steps_per_block / med, "ms_per_step": med / steps * 1e3, "repeats": len(times),
            "value_min": env_steps_per_block / max(times), "value_max": env_steps_per_block / min(times)}


def build_line(args, world, batch, label, env_sizes, alg_bytes_env_step, times, kms, parts, pmc, metrics,
               workload_extra="", extras=None, persistent=False):
    """The ONE JSON line.  `value` and `roofline.frac` share the wall clock of the timed region (the median block: the
    whole TTI of the whole batch); the step kernel's per-launch duration (the dispatch's own timestamps, from the same
    K steps repeated with timing on) is listed beside it -- with partitions `parts` launches of batch/parts envs overlap."""
    S, U, R = env_sizes
    st = block_stats(times, batch * world * args.steps, args.steps)
    ms_per_step = st["ms_per_step"]
    alg_bytes = alg_bytes_env_step * batch                       # per TTI of one rank's batch
    achieved = alg_bytes / (ms_per_step * 1e-3) / 1e9            # per-GPU GB/s on the wall clock
    launch_ms, n_launches = kms["step"], kms["n_launches"]
    ttis_per_launch = kms.get("n_ttis", n_launches) / max(1, n_launches)      # inside a rollout a launch covers several TTIs
    env_ttis_per_launch = kms.get("n_env_ttis", batch / parts * kms.get("n_ttis", n_launches)) / max(1, n_launches)
    launch_bytes = alg_bytes_env_step * env_ttis_per_launch                   # algorithmic bytes of the average launch
    launch_gbs = launch_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    mode = "gather" if getattr(args, "only_gather", False) else "stream"
    traffic = (pmc or {}).get(f"{mode}_rollout" if (parts > 1 or persistent) else mode)
    moved_per_tti = traffic.get("hbm_bytes_per_tti") if traffic else None      # HBM bytes one TTI of the whole batch moves (counters)
    kname = f"ranenv_persist_kernel<{mode}>" if persistent else ("ranenv_core_kernel_gather<STEP>" if mode == "gather" else "ranenv_core_kernel<STEP>")
    if persistent:
        schedule = (f"ranenv_rollout as a persistent work-queue launch (option persist, auto): the K TTIs of the device policy in ONE "
                    f"launch per workgroup class ({n_launches} launch(es) for the timed call: envs sorted by the waves a compact step "
                    f"needs, one wave per 64 slice members), grids of what the chip holds, an env changes hands through its XCD's "
                    f"ready queue only when another env is waiting for a slot; no host between TTIs")
    elif parts > 1:
        schedule = (f"ranenv_rollout: the K TTIs of the device policy enqueued in one call, batch stepped as {parts} partitions on "
                    f"{parts} HIP streams, no host between TTIs; a launch of the step kernel takes its partition through "
                    f"{ttis_per_launch:.3g} TTIs on average (up to K/4, at most 10)")
    else:
        schedule = "one launch of the step kernel per TTI on one stream"
    conc = n_launches if persistent else parts
    line = {
        "metric": "env-steps/s (batched TTIs) at mult_slice 10-slice/100-UE; 1->8 GPU scaling"
                  + (" [device-policy rollout over batch partitions; step-by-step and learner-in-the-loop schedules beside it]"
                     if parts > 1 else ""),
        "value": st["value"], "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "repeats": st["repeats"], "value_min": st["value_min"], "value_max": st["value_max"],
        "timing": f"blocks of exactly {args.steps} steps, each between barrier + device sync (MAX over ranks); value = median of "
                  f"{st['repeats']} blocks",
        "config": {"workload": (f"BASELINE.json configs[{args.config}]" if args.config != 5 else "NOT a BASELINE config") + f": {label}; batch {batch} per GPU; SE replayed from "
                               f"an HBM pool of {args.traces}x{args.trace_len} float32 tiles{workload_extra}",
                   "batch_per_gpu": batch, "global_batch": batch * world, "n_slices": S, "n_ues": U,
                   "n_rbs": R, "parallelism": f"episodes sharded over {world} GPU(s), metrics all_gather only",
                   "launch": schedule, "se_mode": "stream", "se_layout": getattr(args, "se_layout", None)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "clock": "wall clock of the timed region (same as value)",
                     "frac_algorithmic_bytes": achieved / HBM_PEAK_GBS,
                     "frac_moved_bytes": (moved_per_tti / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS) if moved_per_tti else None,
                     "traffic": (moved_per_tti * env_ttis_per_launch / batch) if moved_per_tti else None,
                     "traffic_per_tti": moved_per_tti,
                     "traffic_over_algorithmic": (moved_per_tti / alg_bytes) if moved_per_tti else None,
                     "traffic_source": (f"{PMC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the launches of THIS schedule -- "
                                        f"compact steps, {traffic.get('ttis_per_launch_and_partition', 0):.3g} TTIs per launch on average -- summed and "
                                        f"divided by the TTIs stepped; `traffic` = that per-TTI figure scaled to the env-TTIs of the average launch; "
                                        f"{traffic.get('date', 'undated')}; not measured in this run)") if traffic else None,
                     "kernel": "%s (%d concurrent launch(es) of it, %.3g TTIs per launch)" % (kname, conc, ttis_per_launch),
                     "algorithmic_bytes_per_env_step": alg_bytes_env_step,
                     "algorithmic_bytes_per_tti": alg_bytes,
                     "dominant_kernel": {"name": kname, "ms": launch_ms, "n_launches": n_launches,
                                         "envs_per_launch": env_ttis_per_launch / max(1e-9, ttis_per_launch), "ttis_per_launch": ttis_per_launch,
                                         "env_ttis_per_launch": env_ttis_per_launch,
                                         "algorithmic_bytes": launch_bytes,
                                         "achieved": launch_gbs, "frac": launch_gbs / HBM_PEAK_GBS,
                                         "concurrent_launches": conc,
                                         "source": "hipExtLaunchKernel start/stop events of every launch, the K steps repeated "
                                                   "right after the timed region under the same schedule"}},
        "metrics": metrics,
    }
    line.update(extras or {})
    return line


def gather_block(env, batch, world, steps, times, kms, parts, pmc):
    """The `se_gather` object: the headline schedule in the SE gather mode, against the gather mode's own bytes and --
    it is not HBM-bound -- against the VALU issue rate."""
    st = block_stats(times, batch * world * steps, steps)
    b_env = env.algorithmic_bytes_per_env_step("gather")
    t = st["ms_per_step"] * 1e-3
    hbm_frac = b_env * batch / t / (HBM_PEAK_GBS * 1e9)
    out = dict(st)
    persistent = bool(kms.get("persistent"))
    out.update({"bytes_per_env_step": b_env,
                "bytes_model": "4*R (allocated RBs, each read once) + 8*U (sidecar row of per-UE mean SE) + 180*U + S*(85+8*Us) + 4",
                "hbm_frac": hbm_frac,
                "kernel": "ranenv_persist_kernel<gather> (one persistent work-queue launch per workgroup class and rollout call)"
                          if persistent else "ranenv_core_kernel_gather<STEP>",
                "env_ttis_per_launch": kms.get("n_env_ttis", 0) / max(1, kms["n_launches"]),
                "dominant_kernel_ms": kms["step"], "n_launches": kms["n_launches"],
                "ttis_per_launch": kms.get("n_ttis", kms["n_launches"]) / max(1, kms["n_launches"]), "concurrent_launches": parts})
    g = (pmc or {}).get("gather_rollout") or (pmc or {}).get("gather")
    if g and g.get("valu_insts_per_launch"):
        issue_s = g["valu_insts_per_launch"] * VALU_CYCLES / (N_SIMD * CLOCK_HZ)
        out.update({"valu_insts_per_tti": g["valu_insts_per_launch"], "issue_bound_s": issue_s, "issue_frac": issue_s / t,
                    "issue_model": f"SQ_INSTS_VALU per TTI x {VALU_CYCLES:g} cycles / ({N_SIMD} SIMDs x {CLOCK_HZ / 1e9:g} GHz)",
                    "traffic": g.get("hbm_bytes_per_launch"), "pmc_source": f"{PMC_FILE} ({g.get('date', 'undated')}; not measured in this run)"})
        out["bound"], out["frac"] = ("valu-issue", out["issue_frac"]) if out["issue_frac"] >= hbm_frac else ("hbm", hbm_frac)
    else:
        out["bound"], out["frac"] = "hbm", hbm_frac
    return out


def issue_fields(block, t_s):
    """VALU issue bound of one TTI from a PMC block (SQ_INSTS_VALU per TTI of the batch) against the measured time per TTI."""
    if not block or not block.get("valu_insts_per_tti"):
        return {}
    issue_s = block["valu_insts_per_tti"] * VALU_CYCLES / (N_SIMD * CLOCK_HZ)
    out = {"valu_insts_per_tti": block["valu_insts_per_tti"], "issue_bound_s": issue_s, "issue_frac": issue_s / t_s,
           "pmc_source": f"{PMC_FILE} ({block.get('date', 'undated')}; not measured in this run)"}
    if block.get("wait_any_per_tti") and block.get("wave_cycles_per_tti"):
        out["wait_any_frac_of_wave_cycles"] = block["wait_any_per_tti"] / block["wave_cycles_per_tti"]
    if block.get("hbm_bytes_per_tti"):
        out["traffic_per_tti"] = block["hbm_bytes_per_tti"]
    return out


def other_config_block(cfg, device, args, rank, se_pool, timing, pmc_block=None):
    """One of the other BASELINE configs as a short block: its workload (the resident SE pool re-used where the shape matches),
    warm-up, blocks of exactly K steps of ranenv_rollout under the schedule the library picks for it."""
    import torch
    from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
    sync, barrier, max_over_ranks = timing
    K = args.steps
    t0 = time.perf_counter()
    wl, label = make_bench_workload(cfg, device, n_traces=args.traces, trace_len=args.trace_len, rank=rank, traffic=args.traffic,
                                    se_pool=se_pool)
    env = wl.env
    parts = 3 if env.B >= 2048 else 1
    env.set_partitions(parts)
    env.reset()
    env.rollout(max(1, args.warmup))
    times = timed_blocks(lambda: env.rollout(K), sync, barrier, max_over_ranks)
    st = block_stats(times, env.B * K, K)
    alg = env.algorithmic_bytes_per_env_step("stream")
    persistent, launches = env.get_option("last_rollout_persistent"), env.get_option("last_rollout_launches")
    st.update({"workload": label, "batch": env.B, "se_layout": getattr(env, "se_layout", "rb"), "n_slices": env.S, "n_ues": env.U, "n_rbs": env.R,
               "algorithmic_bytes_per_env_step": alg,
               "roofline_frac": env.B * alg / (st["ms_per_step"] * 1e-3) / (HBM_PEAK_GBS * 1e9),
               "launch": (f"ranenv_rollout: {launches} persistent work-queue launch(es) for the K TTIs" if persistent else
                          f"ranenv_rollout: {launches} launches of the step kernel over {parts} partition(s) for the K TTIs")
                         + (", two envs per wave" if (env.U <= 32 and env.get_option("pack")) else ""),
               "persistent": bool(persistent), "partitions": parts, "se_mode": "stream",
               "setup_s": None})
    if pmc_block and pmc_block.get("batch") == env.B:       # (counters of THIS workload's rollout: which of HBM and VALU issue is nearer)
        st.update(issue_fields(pmc_block, st["ms_per_step"] * 1e-3))
        if "issue_frac" in st:
            st["bound"] = "valu-issue" if st["issue_frac"] >= st["roofline_frac"] else "hbm"
    sync()
    env.close()
    del wl, env
    torch.cuda.empty_cache()
    st["setup_s"] = round(time.perf_counter() - t0, 2)       # (workload build + warm-up + timed blocks: what the block adds to the run)
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None, help="envs per GPU (default: the config's own)")
    ap.add_argument("--config", default="2", choices=("1", "2", "3", "4", "native"),
                    help="BASELINE.json configs index: 1 = B1024 MARR+RR, 2 = B4096 MAPF+PF (default; 3 = the same "
                         "per GPU, i.e. what --gpus 8 runs), 4 = mult_slice_seq sweep B8192, mixed masks; native = the "
                         "reference's own size (S 5 / U 25 / 27 RBGs of 5), B 16384, MAPF+PF: not a BASELINE config")
    ap.add_argument("--traces", type=int, default=200)
    ap.add_argument("--trace-len", type=int, default=1000,
                    help="TTIs per channel trace (1000 = a whole episode: no env ever replays a tile; the pool is "
                         "traces x trace_len x 54 KB = 10.8 GB, + 11.0 GB of sidecars for the se_gather variant)")
    ap.add_argument("--traffic", choices=("pool", "philox"), default="pool",
                    help="offered traffic: replayed Poisson pool (parity mode) or the device counter-based generator")
    ap.add_argument("--partitions", type=int, default=None,
                    help="batch partitions on their own HIP streams (default: 3 on one GPU, 2 per rank in a multi-GPU run, "
                         "1 when the batch does not fill the CUs)")
    ap.add_argument("--ranges", type=int, default=2, help="ranges of the batch the pipelined_step variant alternates between")
    ap.add_argument("--cpu-envs", type=int, default=256)
    ap.add_argument("--cpu-steps", type=int, default=6000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-stream", action="store_true",
                    help="skip the comparison schedules (single_stream, pipelined_step), so that a rocprofv3 pass averages only "
                         "the launches of the measured schedule")
    ap.add_argument("--no-gather", action="store_true", help="skip the se_gather variant (and its sidecar build)")
    ap.add_argument("--only-gather", action="store_true", help="profiling aid: run the se_gather variant only (the line's value "
                                                                "is then the gather mode's, labelled so)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the other_configs blocks (configs[1], configs[4], native size) the default N = 1 run appends")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="plumbing rehearsal of a multi-rank launch on a box with ONE GPU: every rank uses cuda:0 and the process "
                         "group is gloo (RCCL refuses two ranks on one device); the line is labelled a rehearsal, not a measurement")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal aid: at N = 1 initialise the RCCL process group anyway and send the clock's MAX, the barrier and the metrics gather "
                         "through it (communicator creation and the three collectives of the multi-rank path on ONE rank)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the plain command: this process only starts the ranks (one per GPU) and relays rank 0's line; nothing here has
        # touched torch.cuda / HIP yet
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    args.config_name = args.config
    args.config = 5 if args.config == "native" else int(args.config)

    import torch
    import torch.distributed as dist
    from intent_radio_sched_multi_slice_amd import _lib
    from intent_radio_sched_multi_slice_amd.dist import gather_metrics, local_metrics, summarize
    from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload

    world, rank, local_rank = rank_env()
    if args.gpus != world:
        raise SystemExit(f"rank {rank}: --gpus {args.gpus} but the launcher started WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the env step has no CPU fallback")
    dev_index = 0 if args.rehearse_on_one_gpu else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible: one rank per GPU")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    try:
        # (keep_rb_major=False: the RB-major tensor the generator wrote is dropped once its RB-quad-major copy is bound -- one copy of
        # the pool in HBM, shared zero-copy with the other_configs blocks)
        wl, label = make_bench_workload(args.config, device, batch=args.batch, n_traces=args.traces,
                                        trace_len=args.trace_len, rank=rank, traffic=args.traffic, keep_rb_major=False)
    except (torch.OutOfMemoryError, _lib.RanEnvError) as e:       # the first run on a new box must not die on plumbing
        raise SystemExit(f"rank {rank}: building the workload failed ({e}).  The SE pool is traces x trace_len x 54 KB "
                         f"(= {args.traces * args.trace_len * 54e3 / 1e9:.1f} GB here, generated on the GPU in 110 MB bursts) plus "
                         "~1.8 GB of env state at batch 4096: pass a smaller --traces / --trace-len if this GPU is shared.")
    env = wl.env
    batch = env.B
    args.se_layout = ("RB-quad-major [ceil(R/4)][U][4] float32 (ranenv_bind_se_pool_quad: 16-byte loads; a copy made once at bind from the RB-major "
                      "pool the workload generates)" if getattr(env, "se_layout", "rb") == "quad" else "RB-major [R][U] float32")
    # partitions: 3 measured best on one GPU (caller's stream + 2); a process has 4 hardware queues and RCCL wants some
    # of them in a multi-rank run, so there 2 (caller's stream + 1; within 2 % of 3 on one GPU)
    parts = args.partitions if args.partitions is not None else ((3 if world == 1 else 2) if batch >= 2048 else 1)

    coll_dev = torch.device("cpu") if args.rehearse_on_one_gpu else device       # gloo reduces host tensors

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    gm = (lambda v: gather_metrics(v.cpu())) if args.rehearse_on_one_gpu else gather_metrics
    barrier = dist.barrier if use_dist else (lambda: None)
    sync = torch.cuda.synchronize
    K = args.steps
    extras = {}

    def rollout_variant():
        """warm-up, the timed blocks of one ranenv_rollout(K) each, then the same K steps with per-launch timestamps."""
        env.set_partitions(parts)
        env.reset()
        env.rollout(max(1, args.warmup))
        # (the timed schedule itself is warmed by timed_blocks: its first block of K steps -- same call, same launches, class lists and queues of the
        # persistent rollout already built -- only sizes the sample and is not among the blocks `value` is the median of)
        local = []
        times = timed_blocks(lambda: env.rollout(K), sync, barrier, max_over_ranks, local_out=local)
        env.profile_begin()
        env.rollout(K)
        kms = env.profile_end()
        kms["persistent"] = bool(env.get_option("last_rollout_persistent"))     # (reported by the library, not inferred)
        kms["local_block_s"] = float(np.median(local))                            # this rank's own median block, before the barrier
        return times, kms

    times = kms = None
    gathered = None
    if not args.only_gather:
        gm(local_metrics(env.reward, env.views(), env.done, 1))   # warm torch's reduction kernels / RCCL
        times, kms = rollout_variant()
        # metrics: the only collective, once per reporting interval, outside the timed K steps ...
        gathered = gm(local_metrics(env.reward, env.views(), env.done, K))
        if use_dist:
            # what the collective itself saw, and every GPU's own clock (its median block up to its own device sync)
            alg_tti = env.algorithmic_bytes_per_env_step("stream") * batch
            mine = torch.tensor([kms["local_block_s"]], dtype=torch.float64, device=device)
            per_rank_s = gm(mine)[:, 0].tolist()
            extras["ranks_seen"] = {"world_size": dist.get_world_size(), "all_gather_rows": int(gathered.shape[0]),
                                    "backend": dist.get_backend()}
            extras["per_gpu"] = {"median_block_s": per_rank_s,
                                 "roofline_frac": [alg_tti * K / t / (HBM_PEAK_GBS * 1e9) for t in per_rank_s],
                                 "note": "every rank's own median block of K steps up to its own device sync (before the closing "
                                         "barrier), gathered with the metrics collective; roofline.frac above uses the MAX-over-ranks clock"}
        if use_dist:      # ... and one variant with it inside: K steps + the all_gather of the interval's accumulators
            tg = timed_blocks(lambda: (env.rollout(K), gm(local_metrics(env.reward, env.views(), env.done, K))),
                              sync, barrier, max_over_ranks)
            extras["with_metrics_gather"] = dict(block_stats(tg, batch * world * K, K),
                                                 note="the K-step block plus one all_gather of the 8 float64 accumulators per rank")
        if not args.no_single_stream:
            alg = env.algorithmic_bytes_per_env_step("stream")
            if parts > 1:      # one launch per TTI on one stream (what a caller that joins every TTI gets)
                env.set_partitions(1)
                t1 = timed_blocks(lambda: [env.step() for _ in range(K)], sync, barrier, max_over_ranks)
                s1 = block_stats(t1, batch * world * K, K)
                s1.update({"roofline_frac": batch * alg / (s1["ms_per_step"] * 1e-3) / (HBM_PEAK_GBS * 1e9),
                           "launch": "one launch of the step kernel per TTI on one stream (env.step() in a loop); option mix = "
                                     f"{env.get_option('mix')}: a whole-batch step is one launch of mixed blocks -- one block per env of more "
                                     "than 64 slice members, one per two envs of at most 64 -- resident in one round"})
                extras["single_stream"] = s1
            # a learner in the loop: scores from the caller's stream, two halves alternating on their own streams
            env.set_partitions(1)
            env.set_policy(_lib.POLICY_EXTERNAL, wl.intra)
            nr = args.ranges
            ranges = env.set_ranges(nr)
            scores = torch.zeros((batch, env.S), dtype=torch.float64, device=device)
            S = env.S
            # stand-in policy: a function of the range's last inter-slice observation, one small kernel on the caller's
            # stream (tanh of every slice's first drift entry, float32 in, float64 scores out)
            p_in = [env.obs_inter[lo:hi].view(hi - lo, S, 10)[:, :, 0] for lo, hi in ranges]
            p_out = [scores[lo:hi] for lo, hi in ranges]

            def policy(k):
                torch.tanh(p_in[k], out=p_out[k])

            rs = [env.range_stream(k) for k in range(nr)]
            main_stream = torch.cuda.current_stream(device)

            def pipelined_block():      # every range an in-order chain TTI -> policy -> TTI on its own stream
                for _ in range(K):
                    for k in range(nr):
                        torch.cuda.set_stream(rs[k])
                        env.step_wait(k)
                        policy(k)
                        env.step_async(k, scores)
                torch.cuda.set_stream(main_stream)
            env.reset()
            sync()
            for k in range(nr):
                torch.cuda.set_stream(rs[k])
                policy(k); env.step_async(k, scores)
            torch.cuda.set_stream(main_stream)
            pipelined_block()
            tp = timed_blocks(pipelined_block, sync, barrier, max_over_ranks)
            sp = block_stats(tp, batch * world * K, K)
            sp.update({"roofline_frac": batch * alg / (sp["ms_per_step"] * 1e-3) / (HBM_PEAK_GBS * 1e9),
                       "launch": f"external inter-slice scores (a torch op computed from each range's last observation on that range's "
                                 f"own stream), PF intra-slice; the batch as {nr} ranges, each an in-order chain TTI -> policy -> TTI "
                                 "enqueued by ranenv_step_part on the range's stream (set_ranges / range_stream / step_wait / "
                                 "step_async), the host alternating between the ranges; a step = one TTI of every range"})
            extras["pipelined_step"] = sp
            for k in range(nr):
                env.step_wait(k)
            sync()
            env.set_policy(wl.policy, wl.intra)

    pmc = None
    if rank == 0 and os.path.exists(os.path.join(REPO, PMC_FILE)):
        try:
            pj = json.load(open(os.path.join(REPO, PMC_FILE)))
            if pj.get("batch") == batch and pj.get("config") == args.config:
                pmc = {m: dict(pj[m], date=pj.get("date")) for m in ("stream", "gather", "stream_rollout", "gather_rollout", "native_rollout", "config1_rollout") if m in pj}
        except Exception:
            pmc = None

    if not args.no_gather:
        try:
            env.set_se_mode("gather")
        except _lib.RanEnvError as e:
            extras["se_gather"] = {"skipped": f"sidecars not built: {e}"}
        else:
            tg, kg = rollout_variant()
            if rank == 0:
                extras["se_gather"] = gather_block(env, batch, world, K, tg, kg, parts, pmc)
            if args.only_gather:
                times, kms = tg, kg
                gathered = gm(local_metrics(env.reward, env.views(), env.done, K))
            env.set_se_mode("stream")

    if world == 1 and args.config == 2 and args.batch is None and not args.no_other_configs and not args.only_gather:
        sync()
        pool = wl.se_pool
        others = {}
        for name, cfg in (("1", 1), ("4", 4), ("native", 5)):
            try:
                others[name] = other_config_block(cfg, device, args, rank, pool if cfg != 5 else None, (sync, barrier, max_over_ranks),
                                                  (pmc or {}).get({"1": "config1_rollout", "native": "native_rollout"}.get(name, "")))
            except (torch.OutOfMemoryError, _lib.RanEnvError) as e:
                others[name] = {"skipped": str(e)}
        extras["other_configs"] = others

    if rank == 0:
        persistent = bool(kms.get("persistent"))        # option last_rollout_persistent, read right behind the profiled call
        line = build_line(args, world, batch, label, (env.S, env.U, env.R),
                          env.algorithmic_bytes_per_env_step("gather" if args.only_gather else "stream"),
                          times, kms, parts, pmc, summarize(gathered.cpu()), persistent=persistent,
                          workload_extra=(", Poisson traffic pool" if args.traffic == "pool"
                                          else ", Poisson traffic drawn on the device (Philox4x32-10)"),
                          extras=extras)
        if args.rehearse_on_one_gpu:
            line["rehearsal"] = f"{world} ranks sharing ONE GPU over gloo: a plumbing check of the multi-rank path, not a scaling measurement"
        if args.only_gather:
            line["config"]["se_mode"] = "gather (profiling aid: value and roofline are the gather mode's, not the headline)"
        if world == 1 and not args.no_cpu_baseline and not args.only_gather:
            line["cpu_baseline"] = cpu_baseline(wl, args.cpu_envs, args.cpu_steps)
            try:      # the reference's own Python for a full step: it never travels, so it is a recorded figure (tools/time_reference_python.py)
                rp = json.load(open(os.path.join(REPO, REFPY_FILE)))
                size = "scaled" if (env.S, env.U) == (10, 100) else ("native" if (env.S, env.U) == (5, 25) else None)
                if size:
                    line["cpu_baseline"]["reference_python"] = dict(
                        rp["summary"][size], unit="env-steps/s", size=size, cpu=rp.get("cpu"), cores_visible=rp.get("cores_visible"),
                        where="measured in the build container, not on the GPU box (the reference does not travel); "
                              "not measured in this run", what=rp.get("what"), source=REFPY_FILE)
            except Exception:
                pass
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
