"""History files: ``hist/{scenario}/{agent}/ep_{n}.npz`` with the 16 keys the reference's result scripts read
(results/gen_results.py:88-108: ``np.load(..., allow_pickle=True)`` then ``data[key]`` for every key below; per-step
arrays are indexed ``[step, ...]``, ``spectral_efficiencies`` and ``sched_decision`` carry the base-station axis
``(steps, 1, U, R)`` (:262-265, :629), ``slice_ue_assoc`` is ``(steps, S, U)`` (:279), ``reward[idx]["player_0"]`` for
multi-agent runs (:162), ``slice_req[step]["slice_k"]`` dicts (:422-426)).

Two producers: the B = 1 facade (comm_env.MARLCommEnv, ``save_hist=True``) and the batched recorder below, which
keeps the traces of selected envs of a BatchedRanEnv on the device and writes one file per env at ``done``.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import numpy as np

HIST_KEYS = ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "buffer_occupancies", "buffer_latencies",
             "dropped_pkts", "mobility", "spectral_efficiencies", "basestation_ue_assoc", "basestation_slice_assoc",
             "slice_ue_assoc", "sched_decision", "reward", "slice_req", "obs", "agent_action")
OBJECT_KEYS = ("slice_req", "obs", "reward", "agent_action")


def hist_path(root_path: str, simu_name: str, agent_name: str, episode: int) -> str:
    return os.path.join(root_path, "hist", simu_name, agent_name, f"ep_{episode}.npz")


def write_episode_npz(path: str, hist: Dict[str, Sequence]) -> str:
    """``hist[key]`` = one entry per step.  Dict-valued keys go into object arrays (one dict per step), the rest
    become dense arrays ``(steps, ...)``."""
    missing = [k for k in HIST_KEYS if k not in hist]
    if missing:
        raise ValueError(f"history is missing {missing}")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    out = {}
    for k in HIST_KEYS:
        v = hist[k]
        if k in OBJECT_KEYS and len(v) and isinstance(v[0], dict):
            a = np.empty(len(v), dtype=object)
            for i, x in enumerate(v):
                a[i] = x
            out[k] = a
        else:
            out[k] = np.asarray(v)
    np.savez_compressed(path, **out)
    return path


class HistoryRecorder:
    """Traces of selected envs of a BatchedRanEnv, accumulated on the device (a few small gathers on the env's
    stream after every step), written as reference-format history files when the env reports ``done``.

    Every recorded env has its own step counter: envs may be reset under a mask, run episodes of different lengths
    (``set_max_steps``) or move on to their next episode on the device (``enable_autoreset``: the recorder then takes
    the episode number, the scenario and the channel trace of a new episode from the descriptors the device installed).
    ``episode_numbers[i]`` is the episode number env ``envs[i]`` is playing when recording starts; it names the file and,
    without device auto-reset, is advanced by one after every write.  Association / intent columns come from the
    scenario pool row the env replays; the agent action recorded is the inter-slice score vector the step used and the
    intra-slice scheduler choices.
    """

    def __init__(self, env, envs: Sequence[int], root_path: str = ".", simu_name: str = "mult_slice",
                 agent_name: str = "agent", episode_numbers: Optional[Sequence[int]] = None, marl: bool = True):
        import torch
        self._torch = torch
        self.env = env
        self.idx = torch.as_tensor(list(envs), dtype=torch.int64, device=env.device)
        self.envs = [int(e) for e in envs]
        if any(e < 0 or e >= env.B for e in self.envs):
            raise ValueError("recorded env index outside the batch")
        self.root_path, self.simu_name, self.agent_name, self.marl = root_path, simu_name, agent_name, marl
        self.episode_numbers = list(episode_numbers) if episode_numbers is not None else [0] * len(self.envs)
        me = getattr(env, "max_steps_env", None)
        n, U, S = len(self.envs), env.U, env.S
        T = self.T = int(env.max_steps if me is None else np.asarray(me)[self.envs].max())
        z = lambda *sh, dt=torch.int32: torch.zeros(sh, dtype=dt, device=env.device)
        self.buf = {
            "pkt_incoming": z(T, n, U), "pkt_throughputs": z(T, n, U), "pkt_effective_thr": z(T, n, U),
            "dropped_pkts": z(T, n, U), "queue_pkts": z(T, n, U), "queue_age_sum": z(T, n, U, dt=torch.int64),
            "rb_start": z(T, n, U), "rb_count": z(T, n, U),
            "se": z(T, n, env.R, U, dt=torch.float32),
            "reward": z(T, n, S + 1, dt=torch.float64), "scores": z(T, n, S, dt=torch.float64),
            "intra": z(T, n, S, dt=torch.uint8),
            "obs_inter": z(T, n, S * 10, dt=torch.float32), "obs_intra": z(T, n, S, env.W, dt=torch.float32),
        }
        self.t = np.zeros(n, dtype=np.int64)                 # steps recorded of every slot's current episode
        self._cols = torch.arange(n, device=env.device)
        self._stale = np.zeros(n, dtype=bool)                # the slot's episode changed on the device: re-read its descriptor
        self._desc = None
        self._refresh(np.arange(n))
        self.written: List[str] = []

    def _refresh(self, slots):
        """(Re-)read the episode descriptors -- scenario, channel trace -- of the given slots as they are on the device."""
        eps = self.env.episode_descriptors()
        if self._desc is None:
            self._desc = np.array(eps[self.envs])
        else:
            self._desc[slots] = eps[np.asarray(self.envs)[slots]]
        if self.env._autoreset:
            num = self.env.views()["episode_number"].index_select(0, self.idx).cpu().numpy()
            for k in slots:
                self.episode_numbers[k] = int(num[k])
        self._stale[slots] = False

    def on_reset(self, env_mask=None):
        """Called by BatchedRanEnv.reset: the masked envs (all without a mask) start an episode; what was recorded of
        their unfinished one is dropped."""
        if env_mask is None:
            slots = np.arange(len(self.envs))
        else:
            m = env_mask.index_select(0, self.idx).cpu().numpy() if hasattr(env_mask, "index_select") else np.asarray(env_mask)[self.envs]
            slots = np.nonzero(m)[0]
        self.t[slots] = 0
        if len(slots):
            self._refresh(slots)

    def on_step(self, se_tiles, intra_choice, done):
        """Called by BatchedRanEnv.step after the launch (and before an auto-reset is enqueued); ``se_tiles`` = explicit
        tiles of this step or None (pool)."""
        torch, env, i = self._torch, self.env, self.idx
        if self._stale.any():
            self._refresh(np.nonzero(self._stale)[0])
        if (self.t >= self.T).any():
            raise RuntimeError("recorder: an env ran past the longest episode length known when recording started "
                               "(set_max_steps after record()?)")
        tt, cols = torch.as_tensor(self.t, device=env.device), self._cols
        v = env.views()
        for k in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts", "queue_pkts", "queue_age_sum",
                  "rb_start", "rb_count"):
            self.buf[k][tt, cols] = v[k].index_select(0, i)
        self.buf["scores"][tt, cols] = v["policy_scores"].index_select(0, i)
        self.buf["reward"][tt, cols] = env.reward.index_select(0, i)
        self.buf["obs_inter"][tt, cols] = env.obs_inter.index_select(0, i)
        self.buf["obs_intra"][tt, cols] = env.obs_intra.index_select(0, i)
        if intra_choice is not None:
            self.buf["intra"][tt, cols] = intra_choice.index_select(0, i)
        else:
            self.buf["intra"][tt, cols] = int(env.fixed_intra if env.fixed_intra != 255 else 0)
        if se_tiles is not None:
            self.buf["se"][tt, cols] = se_tiles.index_select(0, i)
        else:
            d = self._desc
            tile = d["se_base"] + (d["se_offset"] + self.t) % d["se_len"]
            self.buf["se"][tt, cols] = env.pooled_tiles(torch.as_tensor(tile, device=env.device))
        self.t += 1
        d = done.index_select(0, i).cpu().numpy().astype(bool)         # recording is a diagnostic mode: one small sync
        if d.any():
            which = [k for k in range(len(self.envs)) if d[k]]
            self.flush(which)
            self.t[which] = 0
            if env._autoreset:
                self._stale[which] = True                              # the device installs the next episode after this call

    def flush(self, which: Optional[Sequence[int]] = None) -> List[str]:
        """Write the steps recorded so far of the current episode of recorder slots ``which`` (all by default)."""
        env = self.env
        which = list(range(len(self.envs))) if which is None else list(which)
        Tmax = int(self.t[which].max()) if which else 0
        host = {k: b[:Tmax].cpu().numpy() for k, b in self.buf.items()}
        S, U, R, Us = env.S, env.U, env.R, env.Us
        paths = []
        for k in which:
            T = int(self.t[k])
            scen = int(self._desc["scenario"][k])
            bua, bsa, sua, req = env.tables.to_reference(scen)
            max_pkts = env.tables.ue_max_pkts[scen].astype(np.float64)
            q = host["queue_pkts"][:T, k].astype(np.float64)
            age = host["queue_age_sum"][:T, k].astype(np.float64)
            lat = np.where(q > 0, age / np.maximum(q, 1.0), 0.0)
            st, cn = host["rb_start"][:T, k], host["rb_count"][:T, k]
            r = np.arange(R)[None, None, :]
            sched = ((r >= st[:, :, None]) & (r < (st + cn)[:, :, None])).astype(np.float64)[:, None]   # (T, 1, U, R)
            se = np.swapaxes(host["se"][:T, k], 1, 2).astype(np.float64)[:, None]                        # (T, 1, U, R)
            mask_inter = np.asarray(env.tables.slice_active[scen], dtype=np.int8)
            nues = env.tables.slice_nues[scen]
            obs, rew, act = [], [], []
            for t in range(T):
                if self.marl:
                    o = {"player_0": {"observations": host["obs_inter"][t, k].astype(np.float64), "action_mask": mask_inter}}
                    for s in range(S):
                        o[f"player_{s + 1}"] = {"observations": host["obs_intra"][t, k, s].astype(np.float64),
                                                "action_mask": (np.arange(Us) < nues[s]).astype(np.int8)}
                    obs.append(o)
                    rew.append({f"player_{j}": float(host["reward"][t, k, j]) for j in range(S + 1)})
                    a = {"player_0": host["scores"][t, k].copy()}
                    a.update({f"player_{s + 1}": int(host["intra"][t, k, s]) for s in range(S)})
                    act.append(a)
                else:
                    obs.append(host["obs_inter"][t, k].astype(np.float64))
                    rew.append(float(host["reward"][t, k, 0]))
                    act.append(host["scores"][t, k].copy())
            rep = lambda a: np.repeat(np.asarray(a)[None], T, axis=0)
            hist = {
                "pkt_incoming": host["pkt_incoming"][:T, k].astype(np.float64),
                "pkt_throughputs": host["pkt_throughputs"][:T, k].astype(np.float64),
                "pkt_effective_thr": host["pkt_effective_thr"][:T, k].astype(np.float64),
                "buffer_occupancies": q / max_pkts[None, :], "buffer_latencies": lat,
                "dropped_pkts": host["dropped_pkts"][:T, k].astype(np.float64),
                "mobility": np.ones((T, U, 2)), "spectral_efficiencies": se,
                "basestation_ue_assoc": rep(bua), "basestation_slice_assoc": rep(bsa), "slice_ue_assoc": rep(sua),
                "sched_decision": sched, "reward": rew, "slice_req": [req] * T, "obs": obs, "agent_action": act,
            }
            paths.append(write_episode_npz(hist_path(self.root_path, self.simu_name, self.agent_name,
                                                     self.episode_numbers[k]), hist))
            if not env._autoreset:
                self.episode_numbers[k] += 1
        self.written += paths
        return paths
