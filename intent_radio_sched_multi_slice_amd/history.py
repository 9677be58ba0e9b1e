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

    ``episode_numbers[i]`` is the episode number env ``envs[i]`` is playing (it names the file and is advanced
    by one after every write).  Association / intent columns come from the scenario pool row the env replays; the
    agent action recorded is the inter-slice score vector the step used and the intra-slice scheduler choices.
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
        n, T, U, S = len(self.envs), env.max_steps, env.U, env.S
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
        self.t = 0
        self._scen_at_start = np.array(env.episode_descriptors()["scenario"])[self.envs]
        self.written: List[str] = []

    def on_reset(self):
        self.t = 0
        self._scen_at_start = np.array(self.env.episode_descriptors()["scenario"])[self.envs]

    def on_step(self, se_tiles, intra_choice, done):
        """Called by BatchedRanEnv.step after the launch; ``se_tiles`` = explicit tiles of this step or None (pool)."""
        torch, env, i, t = self._torch, self.env, self.idx, self.t
        if t >= env.max_steps:
            raise RuntimeError("recorder: more steps than max_steps without a reset")
        v = env.views()
        for k in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts", "queue_pkts", "queue_age_sum",
                  "rb_start", "rb_count"):
            self.buf[k][t] = v[k].index_select(0, i)
        self.buf["scores"][t] = v["policy_scores"].index_select(0, i)
        self.buf["reward"][t] = env.reward.index_select(0, i)
        self.buf["obs_inter"][t] = env.obs_inter.index_select(0, i)
        self.buf["obs_intra"][t] = env.obs_intra.index_select(0, i)
        if intra_choice is not None:
            self.buf["intra"][t] = intra_choice.index_select(0, i)
        else:
            self.buf["intra"][t] = int(env.fixed_intra if env.fixed_intra != 255 else 0)
        if se_tiles is not None:
            self.buf["se"][t] = se_tiles.index_select(0, i)
        else:
            eps = env.episode_descriptors()
            tile = eps["se_base"][self.envs] + (eps["se_offset"][self.envs] + t) % eps["se_len"][self.envs]
            self.buf["se"][t] = env._keep["se_pool"].index_select(0, torch.as_tensor(tile, device=env.device))
        self.t = t + 1
        d = done.index_select(0, i).cpu().numpy().astype(bool)         # recording is a diagnostic mode: one small sync
        if d.any():
            self.flush([k for k in range(len(self.envs)) if d[k]])

    def flush(self, which: Optional[Sequence[int]] = None) -> List[str]:
        """Write the steps recorded so far for recorder slots ``which`` (all by default)."""
        env, T = self.env, self.t
        which = list(range(len(self.envs))) if which is None else list(which)
        host = {k: b[:T].cpu().numpy() for k, b in self.buf.items()}
        S, U, R, Us = env.S, env.U, env.R, env.Us
        paths = []
        for k in which:
            e = self.envs[k]
            scen = int(self._scen_at_start[k])
            bua, bsa, sua, req = env.tables.to_reference(scen)
            max_pkts = env.tables.ue_max_pkts[scen].astype(np.float64)
            q = host["queue_pkts"][:, k].astype(np.float64)
            age = host["queue_age_sum"][:, k].astype(np.float64)
            lat = np.where(q > 0, age / np.maximum(q, 1.0), 0.0)
            st, cn = host["rb_start"][:, k], host["rb_count"][:, k]
            r = np.arange(R)[None, None, :]
            sched = ((r >= st[:, :, None]) & (r < (st + cn)[:, :, None])).astype(np.float64)[:, None]   # (T, 1, U, R)
            se = np.swapaxes(host["se"][:, k], 1, 2).astype(np.float64)[:, None]                         # (T, 1, U, R)
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
                "pkt_incoming": host["pkt_incoming"][:, k].astype(np.float64),
                "pkt_throughputs": host["pkt_throughputs"][:, k].astype(np.float64),
                "pkt_effective_thr": host["pkt_effective_thr"][:, k].astype(np.float64),
                "buffer_occupancies": q / max_pkts[None, :], "buffer_latencies": lat,
                "dropped_pkts": host["dropped_pkts"][:, k].astype(np.float64),
                "mobility": np.ones((T, U, 2)), "spectral_efficiencies": se,
                "basestation_ue_assoc": rep(bua), "basestation_slice_assoc": rep(bsa), "slice_ue_assoc": rep(sua),
                "sched_decision": sched, "reward": rew, "slice_req": [req] * T, "obs": obs, "agent_action": act,
            }
            paths.append(write_episode_npz(hist_path(self.root_path, self.simu_name, self.agent_name,
                                                     self.episode_numbers[k]), hist))
            self.episode_numbers[k] += 1
        self.written += paths
        return paths
