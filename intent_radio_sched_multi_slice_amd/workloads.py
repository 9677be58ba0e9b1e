"""Synthetic workloads for the BASELINE.json configurations.

Everything an episode replays is exogenous and action-independent, as the reference requires
(results/gen_results.py:1587-1635): a scenario pool (association + slice intents), an SE pool
(channel traces) and a traffic pool (Poisson draws), all resident in HBM before stepping.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from ._lib import INTRA_PF, INTRA_RR, POLICY_MAPF, POLICY_MARR
from .batched_env import BatchedRanEnv
from .scenario import ScenarioTables, generate_scaled_scenarios


@dataclass
class Workload:
    name: str
    env: BatchedRanEnv
    tables: ScenarioTables
    se_pool: torch.Tensor        # [tiles, R, U] float32 (RB-major); with keep_rb_major=False the pool as it is bound: RB-quad-major [tiles, ceil(R/4), U, 4]
    traffic_pool: torch.Tensor   # [rows, U] int32
    scenario: np.ndarray         # [B]
    se_trace: np.ndarray
    se_offset: np.ndarray
    trace_len: int
    policy: int
    intra: int


def mimic_quadriga_pool(n_traces: int, trace_len: int, n_ues: int, n_rbs: int, seed: int,
                        device: torch.device, chunk: int = 2048) -> torch.Tensor:
    """SE pool following the MimicQuadriga law (channels/mimic_quadriga.py:37-56): per-trace
    per-UE mean |N(10, 7.5)|, per-TTI per-RB |N(mean, 1.5)|, stored float32 RB-major."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    pool = torch.empty((n_traces * trace_len, n_rbs, n_ues), dtype=torch.float32, device=device)
    mu = (torch.randn((n_traces, 1, 1, n_ues), generator=g, device=device) * 7.5 + 10.0).abs()
    per = max(1, chunk // trace_len)
    for t0 in range(0, n_traces, per):
        t1 = min(n_traces, t0 + per)
        z = torch.randn((t1 - t0, trace_len, n_rbs, n_ues), generator=g, device=device)
        pool[t0 * trace_len:t1 * trace_len] = (mu[t0:t1] + 1.5 * z).abs().reshape(-1, n_rbs, n_ues)
    return pool


def _se_pool_or(se_pool, n_traces, trace_len, n_ues, n_rbs, seed, device):
    """An SE pool of the wanted shape that is already resident (the same seed gives the same pool: bench.py hands the headline
    workload's pool to the other configs instead of generating 10.8 GB again), or a new one."""
    if se_pool is not None:
        shapes = ((n_traces * trace_len, n_rbs, n_ues), (n_traces * trace_len, (n_rbs + 3) // 4, n_ues, 4))     # RB-major, or already RB-quad-major
        if tuple(se_pool.shape) not in shapes or se_pool.dtype != torch.float32 or se_pool.device != device:
            raise ValueError("se_pool: float32 [n_traces * trace_len, R, U] (or RB-quad-major [.., ceil(R/4), U, 4]) on the env's device expected")
        return se_pool
    return mimic_quadriga_pool(n_traces, trace_len, n_ues, n_rbs, seed, device)


def quadriga_pool_from_power(power: torch.Tensor, n_rbs: int, transmission_power: float = 100.0,
                             thermal_noise_power: float = 10e-14) -> torch.Tensor:
    """SE pool from QuaDRiGa received power (channels/quadriga.py:56-69), on the GPU.

    ``power``: float64 [n_tiles, R, U] resident on a GPU -- the per-step slices of
    ``target_cell_power`` in the .mat file's own RB-major order (the reference transposes them for the
    agents, :70-72).  Returns the float32 [n_tiles, R, U] pool ``bind_se_pool`` takes.
    """
    import ctypes as C
    from . import _lib
    if not power.is_cuda or power.dtype != torch.float64 or power.dim() != 3 or power.shape[1] != n_rbs:
        raise _lib.RanEnvError("power must be a float64 [n_tiles, R, U] tensor on the GPU")
    power = power.contiguous()
    out = torch.empty(power.shape, dtype=torch.float32, device=power.device)
    lib = _lib.load()
    with torch.cuda.device(power.device):
        st = lib.ranenv_se_from_power(C.c_void_p(power.data_ptr()), C.c_void_p(out.data_ptr()), power.numel(),
                                      float(transmission_power) / float(n_rbs), float(thermal_noise_power),
                                      C.c_void_p(torch.cuda.current_stream(power.device).cuda_stream))
    _lib.check(lib, None, st, "ranenv_se_from_power")
    return out


def poisson_traffic_pool(tables: ScenarioTables, trace_len: int, seed: int) -> np.ndarray:
    """[n_scenarios*trace_len, U] int32 offered bits: Poisson(slice Mbps) * 1e6 for the UEs of each
    slice (traffics/mult_slice.py:26-32); one trace per scenario."""
    rng = np.random.default_rng(seed)
    ns, U = tables.n_scenarios, tables.n_ues
    lam = np.zeros((ns, U))
    for i in range(ns):
        for s in range(tables.n_slices):
            n = int(tables.slice_nues[i, s])
            if tables.slice_has_req[i, s] and n:
                lam[i, tables.slice_ues[i, s, :n]] = tables.slice_traffic[i, s]
    draws = rng.poisson(np.broadcast_to(lam[:, None, :], (ns, trace_len, U)))
    return (draws.astype(np.int64) * 1_000_000).astype(np.int32).reshape(ns * trace_len, U)


def make_mult_slice_workload(batch: int, device: torch.device, policy: int = POLICY_MAPF, intra: int = INTRA_PF,
                             n_scenarios: int = 200, n_traces: int = 200, trace_len: int = 200, seed: int = 10,
                             n_slices: int = 10, n_ues: int = 100, n_rbs: int = 135, rbs_per_rbg: int = 1,
                             max_ues_slice: int = 10, max_steps: int = 1000, rank: int = 0,
                             name: Optional[str] = None, flags: int = 0, min_slices: Optional[int] = None,
                             min_ues: Optional[int] = None, se_pool: Optional[torch.Tensor] = None,
                             se_layout: Optional[str] = None, keep_rb_major: bool = True) -> Workload:
    """BASELINE configs 2-4: S 10 / U 100 / R 135 allocation units, 6..10 active slices with
    distinct templates, 4..10 UEs per slice, MimicQuadriga-law SE replayed from HBM."""
    tables = generate_scaled_scenarios(n_scenarios, seed=seed, n_slices=n_slices, n_ues=n_ues,
                                       max_ues_slice=max_ues_slice,
                                       min_slices=min(6, n_slices) if min_slices is None else min_slices,
                                       min_ues=min(4, max_ues_slice) if min_ues is None else min_ues)
    env = BatchedRanEnv(batch=batch, n_slices=n_slices, n_ues=n_ues, n_rbs=n_rbs, rbs_per_rbg=rbs_per_rbg,
                        max_ues_slice=max_ues_slice, n_scenarios=n_scenarios, max_steps=max_steps,
                        device=device, flags=flags)
    env.load_scenarios(tables)
    se_pool = _se_pool_or(se_pool, n_traces, trace_len, n_ues, n_rbs, seed + 1000 * (rank + 1), env.device)
    trf = torch.from_numpy(poisson_traffic_pool(tables, trace_len, seed + 7)).to(env.device)
    env.bind_se_pool(se_pool, layout=se_layout)
    if not keep_rb_major:         # (the pool as bound -- one copy in HBM; env.pooled_tiles() still hands out RB-major tiles)
        se_pool = env.bound_se_pool
    env.bind_traffic_pool(trf)
    rng = np.random.default_rng(seed + 31 * (rank + 1))
    scenario = rng.integers(0, n_scenarios, batch)
    se_trace = (np.arange(batch) + rank * batch) % n_traces       # env -> trace id = e mod T
    se_offset = rng.integers(0, trace_len, batch)
    env.set_episodes(scenario=scenario, se_base=se_trace * trace_len, se_len=trace_len, se_offset=se_offset,
                     trf_base=scenario * trace_len, trf_len=trace_len, trf_offset=rng.integers(0, trace_len, batch))
    env.set_policy(policy, intra)
    return Workload(name or f"mult_slice S{n_slices}/U{n_ues}/R{n_rbs} B{batch}", env, tables, se_pool, trf,
                    scenario, se_trace, se_offset, trace_len, policy, intra)


def make_mult_slice_seq_workload(batch: int, device: torch.device, policy: int = POLICY_MAPF, intra: int = INTRA_PF,
                                 n_groups: int = 10, channels_per_scenario: int = 100, n_traces: int = 200,
                                 trace_len: int = 200, seed: int = 10, n_slices: int = 10, n_ues: int = 100,
                                 n_rbs: int = 135, rbs_per_rbg: int = 1, max_ues_slice: int = 10,
                                 max_steps: int = 1000, rank: int = 0, flags: int = 0,
                                 se_pool: Optional[torch.Tensor] = None, se_layout: Optional[str] = None,
                                 keep_rb_major: bool = True) -> Workload:
    """BASELINE configs[4]: the ``mult_slice_seq`` per-scenario sweep.  Env e plays episode number
    ``rank*batch + e``; like MultSliceAssociationSeq / QuadrigaChannelSeq (associations/mult_slice_seq.py:38-46,
    channels/quadriga_seq.py:28-39) the association scenario is ``episode // channels_per_scenario`` (mod
    n_groups) and the channel trace changes with every episode, so the envs of a group share the
    association row but not the SE trace.  Scenarios carry 3..S active slices with different intent metric
    sets (the branchy path)."""
    tables = generate_scaled_scenarios(n_groups, seed=seed, n_slices=n_slices, n_ues=n_ues,
                                       max_ues_slice=max_ues_slice, min_slices=min(3, n_slices),
                                       min_ues=min(2, max_ues_slice))
    env = BatchedRanEnv(batch=batch, n_slices=n_slices, n_ues=n_ues, n_rbs=n_rbs, rbs_per_rbg=rbs_per_rbg,
                        max_ues_slice=max_ues_slice, n_scenarios=n_groups, max_steps=max_steps,
                        device=device, flags=flags)
    env.load_scenarios(tables)
    se_pool = _se_pool_or(se_pool, n_traces, trace_len, n_ues, n_rbs, seed + 1000 * (rank + 1), env.device)
    trf = torch.from_numpy(poisson_traffic_pool(tables, trace_len, seed + 7)).to(env.device)
    env.bind_se_pool(se_pool, layout=se_layout)
    if not keep_rb_major:         # (the pool as bound -- one copy in HBM; env.pooled_tiles() still hands out RB-major tiles)
        se_pool = env.bound_se_pool
    env.bind_traffic_pool(trf)
    episode = np.arange(batch, dtype=np.int64) + rank * batch
    scenario = (episode // channels_per_scenario) % n_groups
    se_trace = episode % n_traces
    rng = np.random.default_rng(seed + 31 * (rank + 1))
    se_offset = rng.integers(0, trace_len, batch)
    env.set_episodes(scenario=scenario, se_base=se_trace * trace_len, se_len=trace_len, se_offset=se_offset,
                     trf_base=scenario * trace_len, trf_len=trace_len, trf_offset=rng.integers(0, trace_len, batch))
    env.set_policy(policy, intra)
    return Workload(f"mult_slice_seq S{n_slices}/U{n_ues}/R{n_rbs} B{batch}", env, tables, se_pool, trf,
                    scenario, se_trace, se_offset, trace_len, policy, intra)


def make_bench_workload(config: int, device: torch.device, batch: Optional[int] = None, n_traces: int = 200,
                        trace_len: int = 200, rank: int = 0, traffic: str = "pool", se_pool: Optional[torch.Tensor] = None,
                        keep_rb_major: bool = True):
    """The BASELINE.json ``configs[config]`` workload for one rank -> (Workload, label).  ``se_pool``: a resident pool of the
    config's shape to bind instead of generating one (configs 1-4 share shape and seed: the same pool; RB-major or already
    RB-quad-major, which is bound as it is).  ``keep_rb_major=False``: the generated RB-major tensor is dropped once the
    RB-quad-major copy is bound (``wl.se_pool`` is then that copy)."""
    if config == 1:
        wl = make_mult_slice_workload(batch or 1024, device, policy=POLICY_MARR, intra=INTRA_RR, n_traces=n_traces,
                                      trace_len=trace_len, rank=rank, se_pool=se_pool, keep_rb_major=keep_rb_major)
        label = ("mult_slice, 10 slices, 100 UEs, 135 RBGs, MARR inter-slice + round-robin intra-slice + ib_sched "
                 "intent observation/reward")
    elif config in (2, 3):
        wl = make_mult_slice_workload(batch or 4096, device, policy=POLICY_MAPF, intra=INTRA_PF, n_traces=n_traces,
                                      trace_len=trace_len, rank=rank, se_pool=se_pool, keep_rb_major=keep_rb_major)
        label = ("mult_slice, 10 slices, 100 UEs, 135 RBGs, MAPF inter-slice + PF intra-slice + ib_sched intent "
                 "observation/reward")
    elif config == 4:
        wl = make_mult_slice_seq_workload(batch or 8192, device, policy=POLICY_MAPF, intra=INTRA_PF,
                                          n_traces=n_traces, trace_len=trace_len, rank=rank, se_pool=se_pool, keep_rb_major=keep_rb_major)
        label = ("mult_slice_seq per-scenario sweep, 10 scenario groups with 3..10 active slices (mixed masks and "
                 "intent metric sets), 100 UEs, 135 RBGs, MAPF + PF + ib_sched intent observation/reward")
    elif config == 5:       # not a BASELINE config: the size every reference agent is actually trained at
        wl = make_mult_slice_workload(batch or 16384, device, policy=POLICY_MAPF, intra=INTRA_PF, n_scenarios=200, n_traces=n_traces,
                                      trace_len=trace_len, rank=rank, n_slices=5, n_ues=25, n_rbs=135, rbs_per_rbg=5, max_ues_slice=5,
                                      min_slices=3, min_ues=2, se_pool=se_pool, keep_rb_major=keep_rb_major)          # 3..5 active slices of 2..5 UEs (associations/mult_slice.py:359-423)
        label = ("reference-native size (env_config/mult_slice.yml:2-14, agents/ib_sched.py:50,56): 5 slices, 25 UEs, 27 RBGs of 5 "
                 "RBs, MAPF inter-slice + PF intra-slice + ib_sched intent observation/reward")
    else:
        raise ValueError(f"no bench workload for BASELINE configs[{config}]")
    if traffic == "philox":
        wl.env.set_traffic_generator(seed=1234 + rank)
    elif traffic != "pool":
        raise ValueError("traffic must be 'pool' or 'philox'")
    return wl, label
