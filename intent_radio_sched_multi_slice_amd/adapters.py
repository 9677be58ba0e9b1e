"""Trainer-facing views of BatchedRanEnv.

* ``HeadVecEnv``: the single-agent, vectorised environment the reference's SB3 agents train on
  (``PPO("MlpPolicy", env)``, agents/sched_twc.py:112-118) -- observation and reward of SchedTWC or
  SchedColORAN from the head kernel, action = inter-slice scores with round-robin inside the slices
  (agents/sched_twc.py:415-422).  It follows the ``stable_baselines3.common.vec_env.VecEnv`` protocol
  (``num_envs``, ``reset``, ``step_async`` / ``step_wait`` / ``step``, auto-reset with
  ``infos[i]["terminal_observation"]``) and subclasses it when stable-baselines3 is importable.
* ``marl_obs_dict`` / ``marl_reward_dict``: one env of the batch in the dict layout RLlib's policies
  of the reference consume (``player_0`` = inter-slice agent, ``player_{s+1}`` = intra-slice agents,
  agents/ib_sched.py:160-200, simu.py:559-566).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from ._lib import INTRA_RR, POLICY_EXTERNAL
from .batched_env import BatchedRanEnv

try:  # optional: real base class and spaces when the trainer stack is installed
    from stable_baselines3.common.vec_env import VecEnv as _VecEnvBase   # type: ignore
except Exception:  # noqa: BLE001
    _VecEnvBase = object
try:
    from gymnasium import spaces as _spaces                             # type: ignore
except Exception:  # noqa: BLE001
    _spaces = None


class _Box:
    """Stand-in for gymnasium.spaces.Box when gymnasium is absent (shape / bounds / dtype holder)."""

    def __init__(self, low, high, shape, dtype):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype


def _box(low, high, shape, dtype):
    if _spaces is not None:
        return _spaces.Box(low=low, high=high, shape=tuple(shape), dtype=dtype)
    return _Box(low, high, shape, dtype)


class HeadVecEnv(_VecEnvBase):
    """B environments as one vector env for a single-agent trainer.

    ``reward``: "twc" (SchedTWC.calculate_reward) or "colran" (SchedColORAN.calculate_reward).
    ``slice_usecase``: [n_scenarios, S] eMBB / URLLC bits, needed by the "colran" reward
    (scenario.slice_usecase_from_req).  The env must already have scenarios, pools and episodes set.
    """

    def __init__(self, env: BatchedRanEnv, reward: str = "twc", slice_usecase=None):
        if reward not in ("twc", "colran"):
            raise ValueError("reward must be 'twc' or 'colran'")
        self.env = env
        self._col = 0 if reward == "twc" else 1
        env.enable_heads(slice_usecase)
        env.set_policy(POLICY_EXTERNAL, INTRA_RR)
        self.num_envs = env.B
        S = env.S
        self.observation_space = _box(-np.inf, np.inf, (10 * S,), np.float32)       # sched_twc.py:430-433
        self.action_space = _box(-1.0, 1.0, (S,), np.float64)                        # ib_sched.py:394-403
        if _VecEnvBase is not object:
            super().__init__(self.num_envs, self.observation_space, self.action_space)
        self._intra = torch.zeros((env.B, S), dtype=torch.uint8, device=env.device)
        self._actions: Optional[torch.Tensor] = None

    # -- VecEnv protocol ------------------------------------------------------------------------
    def reset(self):
        self.env.reset()
        return self.env.head_obs.cpu().numpy()

    def step_async(self, actions):
        a = torch.as_tensor(np.asarray(actions), dtype=torch.float64, device=self.env.device)
        if tuple(a.shape) != (self.env.B, self.env.S):
            raise ValueError(f"actions must be [{self.env.B}, {self.env.S}]")
        self._actions = a.clamp(-1.0, 1.0)

    def step_wait(self):
        if self._actions is None:
            raise RuntimeError("step_async was not called")
        _, _, done = self.env.step(self._actions, self._intra)
        self._actions = None
        obs = self.env.head_obs.cpu().numpy()
        rew = self.env.head_reward[:, self._col].cpu().numpy().astype(np.float32)
        dones = done.cpu().numpy().astype(bool)
        infos: List[Dict] = [{} for _ in range(self.num_envs)]
        if dones.any():                                         # auto-reset, as DummyVecEnv does
            for i in np.nonzero(dones)[0]:
                infos[i]["terminal_observation"] = obs[i].copy()
                infos[i]["TimeLimit.truncated"] = False
            self.env.reset(env_mask=dones.astype(np.uint8))
            obs[dones] = self.env.head_obs.cpu().numpy()[dones]
        return obs, rew, dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.env.close()

    # the rest of the VecEnv interface: nothing to delegate to (the envs live on the GPU)
    def get_attr(self, attr_name, indices=None):
        return [getattr(self.env, attr_name)] * self.num_envs

    def set_attr(self, attr_name, value, indices=None):
        setattr(self.env, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        return [getattr(self.env, method_name)(*args, **kwargs)] * self.num_envs

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.num_envs

    def seed(self, seed=None):
        return [None] * self.num_envs


def marl_obs_dict(env: BatchedRanEnv, b: int) -> Dict[str, Dict[str, np.ndarray]]:
    """Env ``b``'s last observation as IBSched.obs_space_format returns it (agents/ib_sched.py:160-200)."""
    v = env.views()
    out = {"player_0": {"observations": env.obs_inter[b].cpu().numpy().astype(np.float64),
                        "action_mask": v["mask_inter"][b].cpu().numpy().astype(np.int8)}}
    oa, ma = env.obs_intra[b].cpu().numpy(), v["mask_intra"][b].cpu().numpy()
    for s in range(env.S):
        out[f"player_{s + 1}"] = {"observations": oa[s].astype(np.float64), "action_mask": ma[s].astype(np.int8)}
    return out


def marl_reward_dict(env: BatchedRanEnv, b: int) -> Dict[str, float]:
    """Env ``b``'s last rewards as calculate_reward_no_mask returns them (agents/common.py:381-439)."""
    r = env.reward[b].cpu().numpy()
    return {f"player_{i}": float(r[i]) for i in range(env.S + 1)}
