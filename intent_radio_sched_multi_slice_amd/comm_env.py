"""MARLCommEnv: the reference-shaped, single-environment face of the HIP env step.

Same construction, ``set_agent_functions`` / ``reset`` / ``step`` protocol and ``comm_env.*``
attributes the reference uses (simu.py:341-424, :539-566), so an agent written against
``sixg_radio_mgmt.MARLCommEnv`` (its ``obs_space_format`` / ``action_format`` / ``calculate_reward``
callbacks and spaces) attaches unchanged.  Per TTI (order of SURVEY.md section 3.1):

    sched_decision = action_format(action)            # agent callback, dense (1, U, R)
    mobility.step -> channel.step -> traffic.step     # host plugins (plugins.py)
    UEs.step                                           # GPU: ranenv_step_dense, B = 1
    obs = obs_space_format(raw observation dict)      # agent callback
    reward = calculate_reward(obs)                    # agent callback
    association.step                                  # next TTI's scenario

This is the compatibility path (one env, host callbacks every TTI).  Throughput comes from
BatchedRanEnv (batched_env.py), which keeps the same arithmetic on the device for thousands of envs.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Callable, Optional

import numpy as np
import torch

from .batched_env import BatchedRanEnv
from .plugins import UEs
from .scenario import IDLE_UE_MAX_AGE, IDLE_UE_MAX_PKTS, IDLE_UE_PKT_SIZE, MAX_AGE_CAP_DEFAULT, ScenarioTables

# env_config/mult_slice.yml of the reference
DEFAULT_CONFIGS = {
    "mult_slice": dict(bandwidths=[100e6], carrier_frequencies=[2.8e9], max_number_basestations=1,
                       num_available_rbs=[135], max_number_episodes=10000, max_number_steps=1000,
                       simu_name="mult_slice", max_number_slices=5, max_number_ues=25, hist_root_path="./"),
}


class CommunicationEnv:
    """Attribute holder for what the reference reads off ``marl_comm_env.comm_env``
    (simu.py:382-385,484-487,539-546; agents/common.py:573-591)."""


class MARLCommEnv:
    def __init__(self, ChannelClass, TrafficClass, MobilityClass, AssociationClass, config_name: str = "mult_slice",
                 agent_name: str = "agent", seed: int = 0, *, root_path: str = ".", initial_episode_number: int = 0,
                 simu_name: str = "mult_slice", save_hist: bool = False, max_episode_number: int = 1,
                 enable_random_episodes: bool = False, config: Optional[dict] = None,
                 max_ues_slice: Optional[int] = None, device: Optional[torch.device] = None, flags: int = 0):
        # flags: include/ranenv.h RANENV_F_* (e.g. _lib.F_SCALE_PER_ELEMENT: the other candidate rounding of UEs.get_pkt_throughputs)
        cfg = dict(config if config is not None else DEFAULT_CONFIGS[config_name])
        ce = CommunicationEnv()
        ce.bandwidths = np.array(cfg["bandwidths"], dtype=float)
        ce.num_available_rbs = np.array(cfg["num_available_rbs"], dtype=int)
        ce.max_number_basestations = int(cfg["max_number_basestations"])
        ce.max_number_slices = int(cfg["max_number_slices"])
        ce.max_number_ues = int(cfg["max_number_ues"])
        ce.max_number_steps = int(cfg["max_number_steps"])
        ce.max_number_episodes = int(max_episode_number)
        ce.initial_episode_number = int(initial_episode_number)
        ce.simu_name, ce.agent_name, ce.root_path = simu_name, agent_name, root_path
        ce.save_hist, ce.enable_random_episodes, ce.seed = bool(save_hist), bool(enable_random_episodes), seed
        ce.step_number, ce.episode_number = 0, ce.initial_episode_number
        ce.rng = np.random.default_rng(seed)
        U, S, R = ce.max_number_ues, ce.max_number_slices, int(ce.num_available_rbs[0])
        ce.ues = UEs(U, np.repeat(IDLE_UE_MAX_AGE, U), np.repeat(IDLE_UE_MAX_PKTS, U), np.repeat(IDLE_UE_PKT_SIZE, U))
        ce.mobility = MobilityClass(U, ce.rng, root_path)
        ce.channel = ChannelClass(U, ce.max_number_basestations, ce.num_available_rbs, ce.rng, root_path, simu_name)
        ce.traffic = TrafficClass(U, ce.rng, root_path)
        ce.associations = AssociationClass(ce.ues, U, ce.max_number_basestations, S, ce.rng, root_path)
        self.comm_env = ce
        self.max_ues_slice = int(max_ues_slice if max_ues_slice is not None else max(1, U // S))
        self._dev = BatchedRanEnv(batch=1, n_slices=S, n_ues=U, n_rbs=R, rbs_per_rbg=1, max_ues_slice=self.max_ues_slice,
                                  n_scenarios=1, bandwidth_hz=float(ce.bandwidths[0]), max_steps=ce.max_number_steps,
                                  max_age_cap=int(cfg.get("max_age_cap", MAX_AGE_CAP_DEFAULT)), device=device, flags=int(flags))
        self._dev.set_episodes(scenario=0)
        self._tables = ScenarioTables.empty(1, S, U, self.max_ues_slice)
        self._loaded_version = None
        self.obs_space_format: Optional[Callable] = None
        self.action_format: Optional[Callable] = None
        self.calculate_reward: Optional[Callable] = None
        self.observation_space = self.action_space = None
        self.hist = {}
        self._zero_assoc()

    # ------------------------------------------------------------------------------------------
    def _zero_assoc(self):
        ce = self.comm_env
        ce.basestation_ue_assoc = np.zeros((ce.max_number_basestations, ce.max_number_ues))
        ce.basestation_slice_assoc = np.zeros((ce.max_number_basestations, ce.max_number_slices))
        ce.slice_ue_assoc = np.zeros((ce.max_number_slices, ce.max_number_ues))
        ce.slice_req = {f"slice_{i}": {} for i in range(ce.max_number_slices)}

    def set_agent_functions(self, obs_space_format, action_format, calculate_reward, obs_space=None, action_space=None):
        """simu.py:405-418"""
        self.obs_space_format, self.action_format, self.calculate_reward = obs_space_format, action_format, calculate_reward
        self.observation_space, self.action_space = obs_space, action_space

    def _association_step(self):
        ce = self.comm_env
        ce.basestation_ue_assoc, ce.basestation_slice_assoc, ce.slice_ue_assoc, ce.slice_req = ce.associations.step(
            ce.basestation_ue_assoc, ce.basestation_slice_assoc, ce.slice_ue_assoc, ce.slice_req,
            ce.step_number, ce.episode_number)

    def _sync_scenario(self):
        """Push association + UE buffer parameters to the device when they changed."""
        ce = self.comm_env
        key = (ce.ues.version, ce.slice_ue_assoc.tobytes(), ce.basestation_slice_assoc.tobytes(),
               tuple(sorted(k for k, v in (ce.slice_req or {}).items() if v)))
        if key == self._loaded_version:
            return
        self._tables.set_from_reference(0, ce.basestation_slice_assoc, ce.slice_ue_assoc, ce.slice_req or {}, True,
                                        (ce.ues.pkt_sizes, ce.ues.max_buffer_pkts, ce.ues.max_buffer_latencies))
        self._dev.load_scenarios(self._tables)
        self._loaded_version = key

    def _se_tile(self, se: np.ndarray) -> np.ndarray:
        """(n_bs, U, R) float64 from a channel plugin -> the device's RB-major float32 tile."""
        return np.ascontiguousarray(np.asarray(se)[0].T, dtype=np.float32)[None]

    def _raw_obs(self, se, mobility, sched_decision):
        ce = self.comm_env
        m = {k: v[0].cpu().numpy() for k, v in self._dev.raw_observation().items()}
        self._last_se32 = np.asarray(se)[0].astype(np.float32)       # the (U, R) tile the device consumed (tests)
        m.update({
            "mobility": mobility, "spectral_efficiencies": np.asarray(se, dtype=np.float64),
            "basestation_ue_assoc": ce.basestation_ue_assoc, "basestation_slice_assoc": ce.basestation_slice_assoc,
            "slice_ue_assoc": ce.slice_ue_assoc, "sched_decision": sched_decision, "slice_req": ce.slice_req,
        })
        self._last_raw = m
        return m

    # ------------------------------------------------------------------------------------------
    def reset(self, seed: Optional[int] = None, options: Optional[dict] = None):
        """simu.py:547-554.  Episode selection: ``options['initial_episode']`` restarts the counter;
        otherwise the next episode (random in [initial, max) when enable_random_episodes)."""
        ce = self.comm_env
        if options and "initial_episode" in options:
            ce.episode_number = int(options["initial_episode"])
        elif ce.step_number > 0 or getattr(self, "_was_reset", False):
            if ce.enable_random_episodes:
                ce.episode_number = int(ce.rng.integers(ce.initial_episode_number, max(ce.max_number_episodes, ce.initial_episode_number + 1)))
            else:
                nxt = ce.episode_number + 1
                ce.episode_number = nxt if nxt < ce.max_number_episodes else ce.initial_episode_number
        if seed is not None:
            ce.seed = seed
            ce.rng = np.random.default_rng(seed)
            for plug in (ce.mobility, ce.channel, ce.traffic, ce.associations):
                plug.rng = ce.rng
        self._was_reset = True
        ce.step_number = 0
        self._zero_assoc()
        self._association_step()
        self._sync_scenario()
        mobility = ce.mobility.step(0, ce.episode_number)
        se = ce.channel.step(0, ce.episode_number, mobility)
        self._dev.reset(se_tiles=self._se_tile(se))
        raw = self._raw_obs(se, mobility, np.zeros((ce.max_number_basestations, ce.max_number_ues, int(ce.num_available_rbs[0]))))
        self.hist = {k: [] for k in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "buffer_occupancies",
                                     "buffer_latencies", "dropped_pkts", "mobility", "spectral_efficiencies",
                                     "basestation_ue_assoc", "basestation_slice_assoc", "slice_ue_assoc",
                                     "sched_decision", "reward", "slice_req", "obs", "agent_action")}
        return self.obs_space_format(raw), {}

    def step(self, action):
        """simu.py:559: returns (obs, reward, terminated, truncated, info)."""
        ce = self.comm_env
        sched = np.asarray(self.action_format(action))
        U, R = ce.max_number_ues, int(ce.num_available_rbs[0])
        if sched.shape != (ce.max_number_basestations, U, R):
            raise ValueError(f"action_format must return {(ce.max_number_basestations, U, R)}, got {sched.shape}")
        mobility = ce.mobility.step(ce.step_number, ce.episode_number)
        se = ce.channel.step(ce.step_number, ce.episode_number, mobility, sched)
        traffic = ce.traffic.step(ce.slice_ue_assoc, ce.slice_req, ce.step_number, ce.episode_number)
        self._last_traffic = np.asarray(traffic, dtype=np.float64)
        self._sync_scenario()
        self._dev.step_dense((sched[0] != 0).astype(np.uint8)[None], np.asarray(traffic, dtype=np.float64)[None],
                             self._se_tile(se))
        raw = self._raw_obs(se, mobility, sched)
        ce.step_number += 1
        obs = self.obs_space_format(raw)
        reward = self.calculate_reward(obs)
        if ce.save_hist:
            for k in self.hist:
                if k in raw:
                    self.hist[k].append(raw[k])
            self.hist["reward"].append(reward); self.hist["obs"].append(obs); self.hist["agent_action"].append(action)
        terminated = ce.step_number >= ce.max_number_steps
        if terminated and ce.save_hist:
            self.save_history()
        if not terminated:
            # next TTI's scenario; a replayed ep_N.npz holds exactly max_number_steps rows
            # (gen_assoc_mult_slice.py:110-119), so the terminal transition must not index one more
            self._association_step()
        if isinstance(reward, dict):
            term = {k: terminated for k in reward}
            term["__all__"] = terminated
            return obs, reward, term, {k: False for k in term}, {}
        return obs, reward, terminated, False, {}

    def save_history(self):
        """hist/{scenario}/{agent}/ep_{n}.npz with the 16 keys of results/gen_results.py:88-108."""
        from .history import hist_path, write_episode_npz
        ce = self.comm_env
        return write_episode_npz(hist_path(ce.root_path, ce.simu_name, ce.agent_name, ce.episode_number), self.hist)

    def close(self):
        self._dev.close()
