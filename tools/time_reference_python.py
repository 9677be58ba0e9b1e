#!/usr/bin/env python3
"""Wall-clock of the REFERENCE's own Python for a full env.step() -- the loop of simu.py:547-566 -- in the build container.

    python tools/time_reference_python.py [--ttis 1000] [--warmup 100] [--out profiles/r04_reference_python.json]

Runs only where /root/reference is mounted (never on the GPU box: the reference does not travel).  What runs, unmodified, from
the reference: agents/ib_sched.py IBSched (obs_space_format, action_format, calculate_reward; agents/common.py under it),
agents/mapf.py MAPF / agents/marr.py MARR (``action = agent.step(obs)``), traffics/mult_slice.py MultSliceTraffic,
channels/mimic_quadriga.py MimicQuadriga, mobilities/simple.py SimpleMobility, associations/mult_slice.py MultSliceAssociation
(generator mode: the datasets are download links) -- attached by env_creator's sequence (simu.py:341-424) to this build's
MARLCommEnv facade exactly as tests/golden/gen_golden_agents.py does.  The one piece that is not the reference's is the env core
(``sixg_radio_mgmt`` is an un-vendored submodule): UEs.step is this build's C oracle behind the facade (a ctypes call per TTI),
i.e. the figure is a LOWER bound on the reference's own step time.

Sizes: the reference's native S 5 / U 25 / 27 RBGs of 5 RBs (env_config/mult_slice.yml:2-14, agents/ib_sched.py:50,56) and the
BASELINE's scaled S 10 / U 100 / 135 RBGs of 1.  Seeds 10 and 15 (simu.py:203-204).  One process on one core, then 8 processes
at once (the reference trains with 10 env runners, agents/ray_agent.py:296-300; this container has 8 cores).
A stated baseline, never a target; bench.py carries it as cpu_baseline.reference_python."""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import platform
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
sys.dont_write_bytecode = True

SIZES = {"native": dict(S=5, U=25, R=135, G=5, Us=5), "scaled": dict(S=10, U=100, R=135, G=1, Us=10)}


def run_one(size: str, agent_name: str, seed: int, ttis: int, warmup: int):
    """-> (seconds for `ttis` TTIs, per-phase seconds dict)."""
    import numpy as np
    import gen_golden_agents as gga                      # the harness that pins agents_on_facade.npz: same shims, same sequence
    from intent_radio_sched_multi_slice_amd import comm_env
    gga.install_shims()
    comm_env.BatchedRanEnv = gga.OracleDevice            # no GPU here: UEs.step by the CPU oracle
    from agents.mapf import MAPF
    from agents.marr import MARR
    from associations.mult_slice import MultSliceAssociation
    from channels.mimic_quadriga import MimicQuadriga
    from mobilities.simple import SimpleMobility
    from traffics.mult_slice import MultSliceTraffic
    sz = SIZES[size]

    class GenAssociation(MultSliceAssociation):
        def __init__(self, *a, **k):
            super().__init__(*a, generator_mode=True, **k)
            self.max_number_slices = sz["S"]             # (associations/mult_slice.py:32 hard-codes 5)

    steps = ttis + warmup
    cfg = dict(comm_env.DEFAULT_CONFIGS["mult_slice"], max_number_steps=steps, max_number_slices=sz["S"], max_number_ues=sz["U"],
               num_available_rbs=[sz["R"]])
    env = comm_env.MARLCommEnv(MimicQuadriga, MultSliceTraffic, SimpleMobility, GenAssociation, "mult_slice", agent_name, seed,
                               root_path="/nonexistent", initial_episode_number=0, simu_name="mult_slice", save_hist=False,
                               max_episode_number=2, enable_random_episodes=False, config=cfg, max_ues_slice=sz["Us"])
    ce = env.comm_env
    AgentCls = {"mapf": MAPF, "marr": MARR}[agent_name]
    agent = AgentCls(env, ce.max_number_ues, ce.max_number_slices, ce.max_number_basestations, ce.num_available_rbs, seed=seed)
    agent.rbs_per_rbg, agent.max_number_ues_slice = sz["G"], sz["Us"]
    for inner in (getattr(agent, "fake_agent", None),):
        if inner is not None:
            inner.rbs_per_rbg, inner.max_number_ues_slice = sz["G"], sz["Us"]
    env.set_agent_functions(agent.obs_space_format, agent.action_format, agent.calculate_reward,
                            agent.get_obs_space(), agent.get_action_space())
    agent.init_agent()
    obs, _ = env.reset(seed=seed, options={"initial_episode": 0})
    # phase clocks: wrap the callbacks the facade calls (the facade's own glue is the remainder)
    phase = {"agent.step": 0.0, "action_format": 0.0, "obs_space_format": 0.0, "calculate_reward": 0.0, "plugins": 0.0, "env_core": 0.0}

    def timed(name, fn):
        def w(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                phase[name] += time.perf_counter() - t
        return w
    env.action_format = timed("action_format", env.action_format)
    env.obs_space_format = timed("obs_space_format", env.obs_space_format)
    env.calculate_reward = timed("calculate_reward", env.calculate_reward)
    for plug in ("mobility", "channel", "traffic", "associations"):
        p = getattr(ce, plug)
        p.step = timed("plugins", p.step)
    env._dev.step_dense = timed("env_core", env._dev.step_dense)
    step_agent = timed("agent.step", agent.step)
    t0 = None
    for t in range(steps):
        if t == warmup:
            for k in phase:
                phase[k] = 0.0
            t0 = time.perf_counter()
        action = step_agent(obs)                          # simu.py:555-558
        obs, reward, term, trunc, info = env.step(action)
    dt = time.perf_counter() - t0
    assert term["__all__"] if isinstance(term, dict) else term
    return dt, phase


def _worker(args):
    size, agent_name, seed, ttis, warmup = args
    dt, _ = run_one(size, agent_name, seed, ttis, warmup)
    return dt


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return platform.processor() or "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ttis", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r04_reference_python.json"))
    args = ap.parse_args()
    if not os.path.isdir(os.environ.get("RANENV_REFERENCE", "/root/reference")):
        raise SystemExit("the reference is not mounted here: this tool runs in the build container only")
    import numpy as np
    cores = len(os.sched_getaffinity(0))
    res = {"what": "reference Python full env.step() (simu.py:547-566): IBSched + MAPF/MARR + MultSliceTraffic + MimicQuadriga + "
                   "MultSliceAssociation (generator mode) from /root/reference on this build's MARLCommEnv facade; UEs.step by the "
                   "build's C oracle (sixg_radio_mgmt is un-vendored): a lower bound on the reference's own step time",
           "where": "measured in the build container, NOT on the GPU box (the reference does not travel)",
           "cpu": cpu_model(), "cores_visible": cores, "python": platform.python_version(), "numpy": np.__version__,
           "ttis": args.ttis, "warmup": args.warmup, "unit": "env-steps/s", "runs": []}
    ctx = mp.get_context("spawn")
    for size in ("native", "scaled"):
        for agent_name in ("mapf", "marr"):
            for seed in (10, 15):
                with ctx.Pool(1) as pool:                 # a fresh interpreter per run: the shims patch module globals
                    dt1 = pool.map(_worker, [(size, agent_name, seed, args.ttis, args.warmup)])[0]
                with ctx.Pool(1) as pool:
                    ph = pool.apply(run_one, (size, agent_name, seed, args.ttis, args.warmup))[1]
                n = min(args.procs, cores)
                with ctx.Pool(n) as pool:
                    t0 = time.perf_counter()
                    dts = pool.map(_worker, [(size, agent_name, seed + 1000 * i, args.ttis, args.warmup) for i in range(n)])
                row = {"size": size, **SIZES[size], "agent": agent_name, "seed": seed,
                       "one_process_steps_per_s": args.ttis / dt1, "one_process_ms_per_step": dt1 / args.ttis * 1e3,
                       "n_processes": n, "n_process_steps_per_s": n * args.ttis / max(dts),
                       "phase_ms_per_step": {k: v / args.ttis * 1e3 for k, v in ph.items()}}
                res["runs"].append(row)
                print(json.dumps(row), flush=True)
    def med(size, key):
        return float(np.median([r[key] for r in res["runs"] if r["size"] == size]))
    res["summary"] = {size: {"one_process_steps_per_s": med(size, "one_process_steps_per_s"),
                             "n_process_steps_per_s": med(size, "n_process_steps_per_s"), "n_processes": min(args.procs, cores)}
                      for size in ("native", "scaled")}
    json.dump(res, open(args.out, "w"), indent=1)
    print("->", args.out, json.dumps(res["summary"]))


if __name__ == "__main__":
    main()
