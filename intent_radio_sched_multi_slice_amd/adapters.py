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


class _Discrete:
    """Stand-in for gymnasium.spaces.Discrete when gymnasium is absent."""

    def __init__(self, n):
        self.n = int(n)


class _Dict:
    """Stand-in for gymnasium.spaces.Dict when gymnasium is absent."""

    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def __getitem__(self, key):
        return self.spaces[key]


def _box(low, high, shape, dtype):
    if _spaces is not None:
        return _spaces.Box(low=low, high=high, shape=tuple(shape), dtype=dtype)
    return _Box(low, high, shape, dtype)


def _discrete(n):
    return _spaces.Discrete(n) if _spaces is not None else _Discrete(n)


def _dict(spaces):
    return _spaces.Dict(spaces) if _spaces is not None else _Dict(spaces)


def describe_space(sp) -> dict:
    """A space (gymnasium's or the stand-ins above) as plain data: nested dicts of {"shape", "low", "high", "dtype"} /
    {"discrete": n}."""
    if hasattr(sp, "spaces"):
        return {k: describe_space(v) for k, v in sp.spaces.items()}
    if hasattr(sp, "n"):
        return {"discrete": int(sp.n)}
    return {"shape": list(sp.shape), "low": float(np.min(sp.low)), "high": float(np.max(sp.high)), "dtype": np.dtype(sp.dtype).name}


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
    def enable_device_autoreset(self, initial_episode: int, max_episode: int, random_episodes: bool = False, seed: int = 0,
                                episode_numbers=None):
        """Finished envs move to their next episode and are reset on the device (BatchedRanEnv.enable_autoreset;
        the env needs an episode table): ``step_wait`` then costs one device-to-host copy per step."""
        self.env.enable_autoreset(initial_episode, max_episode, random_episodes, seed, episode_numbers)

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
        env = self.env
        _, _, done = env.step(self._actions, self._intra)
        self._actions = None
        infos: List[Dict] = [{} for _ in range(self.num_envs)]
        if env._autoreset:
            # one packed D2H: [observation (already the next episode's first one where done) | reward | done]
            n = env.head_obs.shape[1]
            packed = torch.cat([env.head_obs.to(torch.float64), env.head_reward[:, self._col:self._col + 1],
                                done.to(torch.float64)[:, None]], dim=1).cpu().numpy()
            obs, rew, dones = packed[:, :n].astype(np.float32), packed[:, n].astype(np.float32), packed[:, n + 1] != 0
            if dones.any():                                     # rare (once per episode and env): fetch the terminal rows
                idx = np.nonzero(dones)[0]
                term = env.term_head_obs[torch.as_tensor(idx, device=env.device)].cpu().numpy()
                for j, i in enumerate(idx):
                    infos[i]["terminal_observation"] = term[j]
                    infos[i]["TimeLimit.truncated"] = False
            return obs, rew, dones, infos
        obs = env.head_obs.cpu().numpy()
        rew = env.head_reward[:, self._col].cpu().numpy().astype(np.float32)
        dones = done.cpu().numpy().astype(bool)
        if dones.any():                                         # auto-reset from the host, as DummyVecEnv does
            for i in np.nonzero(dones)[0]:
                infos[i]["terminal_observation"] = obs[i].copy()
                infos[i]["TimeLimit.truncated"] = False
            env.reset(env_mask=dones.astype(np.uint8))
            obs[dones] = env.head_obs.cpu().numpy()[dones]
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


# --------------------------------------------------------------------------------------------
# the whole batch in the reference's RLlib layout, on the device
# --------------------------------------------------------------------------------------------
def sorted_action_mask(action_mask: torch.Tensor) -> torch.Tensor:
    """TorchActionMaskModel.forward's mask (agents/action_mask_model.py:46-50): the inter-slice observation lists the
    slices sorted by requested traffic, inactive ones first, so the mask handed to the action distribution has its
    last ``n_active`` entries set.  Batched: every row uses its own count (the reference takes row 0's for all)."""
    n = action_mask.to(torch.int64).sum(dim=-1, keepdim=True)
    S = action_mask.shape[-1]
    pos = torch.arange(S, device=action_mask.device).expand_as(action_mask)
    return (pos >= S - n).to(action_mask.dtype)


def masked_gaussian_params(mean: torch.Tensor, log_std: torch.Tensor, masks: torch.Tensor):
    """The inter-slice action distribution's parameters (agents/masked_action_distribution.py:30-36): std = exp(log_std);
    where the mask is 0 the mean is -1 and the std 1e-9, so an inactive slice always gets the score -1."""
    std = torch.exp(log_std)
    std = torch.where(masks == 0, torch.full_like(std, 1e-9), std)
    mean = torch.where(masks == 0, torch.full_like(mean, -1.0), mean)
    return mean, std


class MarlBatchEnv:
    """B environments in the multi-agent layout the reference trains RLlib policies on, as device tensors.

    Observation (IBSched.obs_space_format, agents/ib_sched.py:160-200; spaces :413-470):
        ``obs["player_0"]``     = {"observations": float32 [B, 10*S], "action_mask": int8 [B, S]}
        ``obs["player_{s+1}"]`` = {"observations": float32 [B, 2*Us+9], "action_mask": int8 [B, Us]}
    Action (IBSched.get_action_space :394-411): ``{"player_0": [B, S] scores in [-1, 1] (the masked diagonal Gaussian
    over S of masked_action_distribution.py), "player_{s+1}": [B] integers in {0, 1, 2} (Discrete(3): RR / PF / MT)}``.
    Reward: ``{"player_i": float64 [B]}``; terminated: ``{"player_i": bool [B], "__all__": bool [B]}`` (simu.py:559-564).
    All tensors are views of the env's buffers (zero copy); nothing here synchronises with the host.
    ``observation_space`` / ``action_space``: one env's spaces as IBSched.get_obs_space / get_action_space declare them
    (agents/ib_sched.py:394-470; gymnasium's classes when installed), with two differences: the per-slice observation has
    ``2 * max_ues_slice + 9`` entries (the reference hard-codes max_number_ues / max_number_slices UEs per slice) and
    observations are declared float32, which is what the device hands out.
    """

    def __init__(self, env: BatchedRanEnv):
        self.env = env
        env.set_policy(POLICY_EXTERNAL, 255)           # scores and schedulers come with every action
        self.players = [f"player_{i}" for i in range(env.S + 1)]
        self._intra = torch.zeros((env.B, env.S), dtype=torch.uint8, device=env.device)
        S, Us = env.S, env.Us
        self.action_space = _dict({p: (_box(-1, 1, (S,), np.float64) if i == 0 else _discrete(3))       # :394-411
                                   for i, p in enumerate(self.players)})
        self.observation_space = _dict({                                                                  # :413-470
            p: _dict({"observations": _box(-1, np.inf, (S * 10,) if i == 0 else (2 * Us + 9,), np.float32),
                      "action_mask": _box(0.0, 1.0, (S,) if i == 0 else (Us,), np.int8)})
            for i, p in enumerate(self.players)})

    def _obs(self):
        env, v = self.env, self.env.views()
        out = {"player_0": {"observations": env.obs_inter, "action_mask": v["mask_inter"]}}
        for s in range(env.S):
            out[f"player_{s + 1}"] = {"observations": env.obs_intra[:, s], "action_mask": v["mask_intra"][:, s]}
        return out

    def reset(self):
        self.env.reset()
        return self._obs(), {}

    def step(self, action: Dict[str, torch.Tensor]):
        env = self.env
        scores = action["player_0"].to(device=env.device, dtype=torch.float64)
        self._intra.copy_(torch.stack([action[f"player_{s + 1}"].to(env.device) for s in range(env.S)], dim=1))
        _, reward, done = env.step(scores, self._intra)
        rew = {p: reward[:, i] for i, p in enumerate(self.players)}
        d = done != 0
        term = {p: d for p in self.players}
        term["__all__"] = d
        trunc = {p: torch.zeros_like(d) for p in term}
        return self._obs(), rew, term, trunc, {}
