#!/usr/bin/env python3
"""agents_on_facade.npz: the reference's REAL agent and plugin classes attached to this build's MARLCommEnv facade.

Run in the build container only (the reference lives at /root/reference and never travels):

    python tests/golden/gen_golden_agents.py

What runs here, unmodified, from /root/reference: agents/ib_sched.py IBSched, agents/marr.py MARR, agents/mapf.py MAPF
(and agents/common.py under them), associations/mult_slice.py MultSliceAssociation (generator mode),
traffics/mult_slice.py MultSliceTraffic, channels/mimic_quadriga.py MimicQuadriga, mobilities/simple.py SimpleMobility.
Their ``from sixg_radio_mgmt import ...`` resolves to INTEGRATION.md section 2's shim -- this build's comm_env.MARLCommEnv /
CommunicationEnv and plugins.{Agent, Association, Channel, Mobility, Traffic, UEs} -- and ``gymnasium.spaces`` to a
shape-holding stand-in (gymnasium is not installed).  The sequence is env_creator's (simu.py:341-424): MARLCommEnv(
ChannelCls, TrafficCls, MobilityCls, AssociationCls, "mult_slice", agent_name, seed, root_path=..., ...), the agent built
from ``marl_comm_env.comm_env.*``, ``set_agent_functions(obs_space_format, action_format, calculate_reward,
get_obs_space(), get_action_space())``, ``init_agent()``; then the test loop of simu.py:547-566 for one 50-TTI episode
(``action = agent.step(obs)`` for MARR / MAPF; IBSched is the learner's adapter -- its step() raises -- so its actions are
drawn like a policy's output).

There is no GPU in this container: under the facade, where the product has BatchedRanEnv (HIP), sits a stand-in with the
same five methods backed by the CPU oracle's env core (UEs.step only -- everything else on the path is the facade's own
code and the reference's classes).  The fixture stores what went in (seed, actions) and what came out (sched_decision
ranges, raw integers, dict observations, rewards, spaces); tests/test_gpu_reference_agents.py replays it through the real
facade on the GPU.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("RANENV_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

import torch                                                    # noqa: E402
from intent_radio_sched_multi_slice_amd import comm_env, plugins   # noqa: E402
from oracle import pyoracle                                     # noqa: E402


class OracleDevice:
    """The five methods the facade calls on BatchedRanEnv (comm_env.py), answered by the CPU oracle's env core."""

    def __init__(self, batch, n_slices, n_ues, n_rbs, rbs_per_rbg=1, max_ues_slice=None, n_scenarios=1, bandwidth_hz=100e6,
                 max_steps=1000, max_age_cap=400, device=None, **kw):
        assert batch == 1 and n_scenarios == 1
        self.S, self.U, self.R = n_slices, n_ues, n_rbs
        self.cfg = pyoracle.make_cfg(n_slices, n_ues, n_rbs, rbs_per_rbg, max_ues_slice, bandwidth_hz=bandwidth_hz,
                                     max_age_cap=max_age_cap, max_steps=max_steps)
        self.core = pyoracle.OracleEnv(self.cfg)

    def set_episodes(self, **kw):
        pass

    def load_scenarios(self, tables):
        self.core.set_scenario(tables, 0)

    def reset(self, se_tiles=None):
        self.core.reset(np.ascontiguousarray(np.asarray(se_tiles)[0].T))          # RB-major tile -> the oracle's (U, R)

    def step_dense(self, sched, traffic, se_tiles):
        self.core.core_step(np.asarray(sched)[0], np.ascontiguousarray(np.asarray(se_tiles)[0].T), np.asarray(traffic)[0])

    def raw_observation(self):
        return {k: torch.as_tensor(v)[None] for k, v in self.core.raw().items()}

    def close(self):
        pass


def install_shims():
    """INTEGRATION.md section 2's shim, and a shape-holding gymnasium.spaces."""
    m = types.ModuleType("sixg_radio_mgmt")
    for name in ("MARLCommEnv", "CommunicationEnv"):
        setattr(m, name, getattr(comm_env, name))
    for name in ("Agent", "Association", "Channel", "Mobility", "Traffic", "UEs"):
        setattr(m, name, getattr(plugins, name))
    sys.modules["sixg_radio_mgmt"] = m
    g, sp = types.ModuleType("gymnasium"), types.ModuleType("gymnasium.spaces")

    class _Space:
        def __init__(self, *a, **k):
            self.args, self.kwargs = a, k

    class Box(_Space):
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            super().__init__(low=low, high=high, shape=shape, dtype=dtype)
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    class Discrete(_Space):
        def __init__(self, n):
            super().__init__(n)
            self.n = n

    class Dict(_Space):
        def __init__(self, spaces):
            super().__init__(spaces)
            self.spaces = dict(spaces)

        def __getitem__(self, key):                          # gymnasium's Dict is subscriptable (agents/sched_twc.py:431)
            return self.spaces[key]

    sp.Box, sp.Discrete, sp.Dict = Box, Discrete, Dict
    g.spaces = sp
    sys.modules["gymnasium"], sys.modules["gymnasium.spaces"] = g, sp
    try:
        import matplotlib.pylab  # noqa: F401  (agents/mapf.py imports a name from it)
    except Exception:  # pragma: no cover
        mp, pl = types.ModuleType("matplotlib"), types.ModuleType("matplotlib.pylab")
        pl.f = None
        sys.modules["matplotlib"], sys.modules["matplotlib.pylab"] = mp, pl
    sys.path.insert(0, REF)


def install_sb3_standins():
    """agents/sched_twc.py and sched_colran.py import stable-baselines3 (absent here) for their trainers; the observation,
    reward and action code needs none of it.  Attribute-holding stand-ins (PPO(...) just stores its arguments)."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m

    class _Stub:
        def __init__(self, *a, **k):
            self.args, self.kwargs = a, k

    sys.modules["gymnasium"].Env = object
    sys.modules["gymnasium.spaces"].Space = object
    mod("stable_baselines3"); mod("stable_baselines3.common")
    mod("stable_baselines3.common.callbacks", CheckpointCallback=_Stub, BaseCallback=_Stub, EvalCallback=_Stub, EventCallback=_Stub)
    mod("stable_baselines3.common.evaluation", evaluate_policy=None)
    mod("stable_baselines3.common.vec_env", DummyVecEnv=_Stub, VecEnv=_Stub, sync_envs_normalization=None)
    mod("stable_baselines3.ppo"); mod("stable_baselines3.ppo.ppo", PPO=_Stub)
    mod("stable_baselines3.sac"); mod("stable_baselines3.sac.sac", SAC=_Stub)


def describe_space(sp):
    if hasattr(sp, "spaces"):
        return {k: describe_space(v) for k, v in sp.spaces.items()}
    if hasattr(sp, "n"):
        return {"discrete": int(sp.n)}
    return {"shape": list(sp.shape), "low": float(np.min(sp.low)), "high": float(np.max(sp.high)), "dtype": np.dtype(sp.dtype).name}


def main():
    install_shims()
    comm_env.BatchedRanEnv = OracleDevice                    # no GPU here: UEs.step by the CPU oracle
    import numpy as _np
    from agents import common as ref_common, ib_sched as ref_ib
    from agents.ib_sched import IBSched
    from agents.mapf import MAPF
    from agents.marr import MARR
    from associations.mult_slice import MultSliceAssociation
    from channels.mimic_quadriga import MimicQuadriga
    from mobilities.simple import SimpleMobility
    from traffics.mult_slice import MultSliceTraffic

    class _StableNumpy:                                     # the build's tie rule: np.argsort stable
        def __getattr__(self, k):
            return getattr(_np, k)

        @staticmethod
        def argsort(a, *args, **kw):
            kw.setdefault("kind", "stable")
            return _np.argsort(a, *args, **kw)
    ref_common.np = _StableNumpy(); ref_ib.np = _StableNumpy()

    class GenAssociation(MultSliceAssociation):              # the datasets a replay would read are download links
        def __init__(self, *a, **k):
            super().__init__(*a, generator_mode=True, **k)

    install_sb3_standins()
    from agents import sched_twc as ref_twc, sched_colran as ref_col
    from agents.sched_colran import SchedColORAN
    from agents.sched_twc import SchedTWC
    ref_twc.np = _StableNumpy(); ref_col.np = _StableNumpy()

    steps, seed = 50, 10
    out = {"cfg": np.array([5, 25, 135, 5, 5, seed, steps])}
    for agent_name, AgentCls in (("ib_sched", IBSched), ("marr", MARR), ("mapf", MAPF), ("sched_twc", SchedTWC),
                                 ("sched_colran", SchedColORAN)):
        env_config = {"seed": seed, "agent": agent_name, "root_path": "/nonexistent", "scenario": "mult_slice", "save_hist": False,
                      "enable_random_episodes": False}
        cfg = dict(comm_env.DEFAULT_CONFIGS["mult_slice"], max_number_steps=steps)
        env = comm_env.MARLCommEnv(MimicQuadriga, MultSliceTraffic, SimpleMobility, GenAssociation, "mult_slice",
                                   env_config["agent"], env_config["seed"], root_path=env_config["root_path"],
                                   initial_episode_number=0, simu_name=env_config["scenario"], save_hist=env_config["save_hist"],
                                   max_episode_number=2, enable_random_episodes=env_config["enable_random_episodes"], config=cfg,
                                   max_ues_slice=5)
        ce = env.comm_env
        head = agent_name.startswith("sched_")
        if head:        # env_creator's "rl" branch (simu.py:380-397): keyword arguments, no evaluation env
            agent = AgentCls(env=env, max_number_ues=ce.max_number_ues, max_number_slices=ce.max_number_slices,
                             max_number_basestations=ce.max_number_basestations, num_available_rbs=ce.num_available_rbs,
                             eval_env=None, agent_name=agent_name, seed=env_config["seed"], episode_evaluation_freq=None,
                             number_evaluation_episodes=None, checkpoint_episode_freq=10, eval_initial_env_episode=None)
        else:
            agent = AgentCls(env, ce.max_number_ues, ce.max_number_slices, ce.max_number_basestations, ce.num_available_rbs,
                             seed=env_config["seed"])
        env.set_agent_functions(agent.obs_space_format, agent.action_format, agent.calculate_reward,
                                agent.get_obs_space(), agent.get_action_space())
        agent.init_agent()
        out[f"{agent_name}_spaces"] = np.array(json.dumps({"obs": describe_space(env.observation_space),
                                                           "action": describe_space(env.action_space)}))
        rng = np.random.default_rng(77)
        obs, info = env.reset(seed=env_config["seed"], options={"initial_episode": 0})
        S, U, R = ce.max_number_slices, ce.max_number_ues, int(ce.num_available_rbs[0])
        is_marl = agent_name == "ib_sched"

        def flat_obs(o):
            if is_marl:
                return (np.concatenate([np.asarray(o["player_0"]["observations"], dtype=float)] +
                                       [np.asarray(o[f"player_{s + 1}"]["observations"], dtype=float) for s in range(S)]),
                        np.concatenate([np.asarray(o["player_0"]["action_mask"]).astype(np.int8)] +
                                       [np.asarray(o[f"player_{s + 1}"]["action_mask"]).astype(np.int8) for s in range(S)]))
            return np.asarray(o, dtype=float), np.zeros(0, dtype=np.int8)
        rec = {k: [] for k in ("action", "rb_start", "rb_count", "obs", "mask", "reward", "pkt_incoming", "pkt_throughputs",
                               "pkt_effective_thr", "dropped_pkts", "buffer_occupancies", "buffer_latencies", "traffic", "se_sum")}
        out[f"{agent_name}_reset_obs"], out[f"{agent_name}_reset_mask"] = flat_obs(obs)
        out[f"{agent_name}_slice_ue_assoc"] = ce.slice_ue_assoc.copy()
        out[f"{agent_name}_slice_names"] = np.array(json.dumps({k: (v["name"] if v else None) for k, v in ce.slice_req.items()}))
        out[f"{agent_name}_ues"] = np.stack([ce.ues.pkt_sizes, ce.ues.max_buffer_pkts, ce.ues.max_buffer_latencies])
        terminated, t = False, 0
        while not terminated:
            if is_marl:                                      # a policy's output: Box(-1, 1, (S,)) + Discrete(3) per slice
                action = {"player_0": rng.uniform(-1, 1, S)}
                action.update({f"player_{s + 1}": int(rng.integers(0, 3)) for s in range(S)})
                flat_a = np.concatenate([action["player_0"], [action[f"player_{s + 1}"] for s in range(S)]])
            elif head:                                       # the SB3 model's output: Box(-1, 1, (S,)) (its PPO is a stand-in here)
                action = rng.uniform(-1, 1, S)
                flat_a = action.copy()
            else:
                action = agent.step(obs)                     # simu.py:555-558
                flat_a = np.asarray(action, dtype=float)
            obs, reward, term, trunc, info = env.step(action)
            terminated = term["__all__"] if isinstance(term, dict) else bool(term)
            raw = env._last_raw
            sched = np.asarray(raw["sched_decision"])[0]
            cnt = sched.sum(axis=1).astype(np.int32)
            st = np.array([int(np.nonzero(sched[u])[0][0]) if cnt[u] else 0 for u in range(U)], dtype=np.int32)
            for u in range(U):
                assert sched[u, st[u]:st[u] + cnt[u]].sum() == cnt[u]          # contiguous ranges
            o, m = flat_obs(obs)
            rw = np.array([reward[f"player_{i}"] for i in range(S + 1)]) if is_marl else np.array([float(reward)])
            for k, v in (("action", flat_a), ("rb_start", st), ("rb_count", cnt), ("obs", o), ("mask", m), ("reward", rw),
                         ("traffic", env._last_traffic), ("se_sum", np.asarray(raw["spectral_efficiencies"])[0].sum(axis=1))):
                rec[k].append(np.array(v))
            for k in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts", "buffer_occupancies", "buffer_latencies"):
                rec[k].append(np.asarray(raw[k]).copy())
            t += 1
        assert t == steps
        for k, v in rec.items():
            out[f"{agent_name}_{k}"] = np.array(v)
        print(agent_name, "ok:", t, "TTIs; packets sent", int(out[f"{agent_name}_pkt_effective_thr"].sum()),
              "dropped", int(out[f"{agent_name}_dropped_pkts"].sum()), "reward[0] mean", float(out[f"{agent_name}_reward"][:, 0].mean()))
    out["meta"] = np.array(json.dumps({"numpy": np.__version__, "stable_argsort": 1,
                                       "reference": "lasseufpa/intent_radio_sched_multi_slice snapshot 2026-03-13"}))
    np.savez_compressed(os.path.join(HERE, "agents_on_facade.npz"), **out)
    print("agents_on_facade.npz", os.path.getsize(os.path.join(HERE, "agents_on_facade.npz")), "bytes")


if __name__ == "__main__":
    main()
