"""ctypes binding of the CPU oracle (oracle/ranenv_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libranenv_oracle.so")


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("ranenv_oracle.c", "ranenv_oracle.h")]
    stale = not os.path.exists(_SO) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        if not all(os.path.exists(s) for s in src):
            if os.path.exists(_SO):
                return _SO
            raise FileNotFoundError("oracle sources missing")
        subprocess.run(["make", "-C", _HERE, "-s"], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
    return _SO


class _Cfg(C.Structure):
    _fields_ = [
        ("n_slices", C.c_int32), ("n_ues", C.c_int32), ("n_rbs", C.c_int32), ("rbs_per_rbg", C.c_int32),
        ("max_ues_slice", C.c_int32), ("hist_depth", C.c_int32), ("max_age_cap", C.c_int32),
        ("max_steps", C.c_int32),
        ("bandwidth_hz", C.c_double), ("overfulfill", C.c_double), ("norm_traffic", C.c_double),
        ("norm_ues", C.c_double), ("norm_se", C.c_double),
    ]


_SC_FIELDS = [
    ("slice_active", np.int32), ("slice_has_req", np.int32), ("slice_nues", np.int32),
    ("slice_ues", np.int32), ("slice_priority", np.float64), ("slice_traffic", np.float64),
    ("slice_buffer_size", np.int32), ("slice_buffer_latency", np.int32), ("slice_message_size", np.int32),
    ("slice_nparams", np.int32), ("param_metric", np.int32), ("param_op", np.int32),
    ("param_value", np.float64), ("sorted_slices", np.int32),
    ("ue_pkt_size", np.int32), ("ue_max_pkts", np.int32), ("ue_max_age", np.int32),
]


class _Scenario(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _ in _SC_FIELDS]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_np_sum.restype = C.c_double
        _lib.orc_np_sum.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        _lib.orc_env_create.restype = C.c_void_p
        _lib.orc_env_create.argtypes = [C.POINTER(_Cfg)]
        for name in ("orc_env_destroy", "orc_env_clear"):
            getattr(_lib, name).argtypes = [C.c_void_p]
            getattr(_lib, name).restype = None
        _lib.orc_env_set_scenario.argtypes = [C.c_void_p, C.POINTER(_Scenario)]
        _lib.orc_env_set_scale_per_element.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_env_set_scale_per_element.restype = None
        _lib.orc_env_step_number.argtypes = [C.c_void_p]
        _lib.orc_env_hist_len.argtypes = [C.c_void_p]
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else C.c_void_p(a.ctypes.data)


def np_sum(a: np.ndarray) -> float:
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().orc_np_sum(_p(a), a.size, 1)


def round_int_equal_sum(v: np.ndarray, target: int) -> np.ndarray:
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.zeros(v.size, dtype=np.int64)
    lib().orc_round_int_equal_sum(_p(v), C.c_int(v.size), C.c_int64(int(target)), _p(out))
    return out


def scores_to_rbs(action: np.ndarray, total_rbs: int, association: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(action, dtype=np.float64)
    assoc = np.ascontiguousarray(association, dtype=np.float64)
    out = np.zeros(a.size, dtype=np.int64)
    lib().orc_scores_to_rbs(_p(a), C.c_int(a.size), C.c_int64(int(total_rbs)), _p(assoc), _p(out))
    return out


def sort_slices(nues: np.ndarray, traffic: np.ndarray, has_req: np.ndarray) -> np.ndarray:
    nues = np.ascontiguousarray(nues, dtype=np.int32)
    traffic = np.ascontiguousarray(traffic, dtype=np.float64)
    has_req = np.ascontiguousarray(has_req, dtype=np.int32)
    out = np.zeros(nues.size, dtype=np.int32)
    lib().orc_sort_slices(_p(nues), _p(traffic), _p(has_req), C.c_int(nues.size), _p(out))
    return out


def quadriga_se_from_power(target_cell_power: np.ndarray, n_rbs: int, transmission_power: float = 100.0,
                           thermal_noise_power: float = 10e-14) -> np.ndarray:
    """Checker for the channel-ingest kernel (ranenv_se_from_power): QuadrigaChannel.step's transform,
    channels/quadriga.py:56-69 -- ``spectral_efficiencies = log2(1 + (transmission_power / num_available_rbs)
    * target_cell_power / (intercell_interference + thermal_noise_power))`` with intercell_interference an
    all-zeros array (:62-66), float64 numpy, elementwise."""
    g = np.asarray(target_cell_power, dtype=np.float64)
    interference = np.zeros(g.shape)
    return np.log2(1 + np.divide((transmission_power / n_rbs) * g, interference + thermal_noise_power))


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox-4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), the
    checker for the device generator: counter words c0..c3 (arrays broadcast together), key (k0, k1) -> 4 uint32
    arrays.  Round: (c0, c1, c2, c3) <- (hi(M1*c2) ^ c1 ^ k0, lo(M1*c2), hi(M0*c0) ^ c3 ^ k1, lo(M0*c0)); key += (W0, W1)."""
    m32 = np.uint64(0xFFFFFFFF)
    c = [np.asarray(x, dtype=np.uint64) & m32 for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = np.uint64(k0) & m32, np.uint64(k1) & m32
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & m32, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & m32]
        k0, k1 = (k0 + W0) & m32, (k1 + W1) & m32
    return c


def poisson_from_tables(cdf_row: np.ndarray, u: np.ndarray) -> np.ndarray:
    """Inversion as the device does it: smallest k with u < cdf[k] (k capped at 255)."""
    return np.minimum(np.searchsorted(cdf_row, u, side="right"), 255)


def generator_traffic(cdf, tables, scenario: int, seed: int, env_id: int, episode: int, step: int) -> np.ndarray:
    """[U] offered bits of one env and TTI as ranenv_set_traffic_generator defines them: Poisson(slice Mbps) * 1e6
    for the UEs of slices with a request (traffics/mult_slice.py:24-32), u = (x1 << 32 | x0) of
    Philox-4x32-10(counter (env_id, episode, step, ue), key (seed lo, seed hi))."""
    U, S = tables.n_ues, tables.n_slices
    out = np.zeros(U)
    ue = np.arange(U, dtype=np.uint64)
    x = philox4x32_10(env_id, episode, step, ue, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = (x[1] << np.uint64(32)) | x[0]
    for s in range(S):
        n = int(tables.slice_nues[scenario, s])
        if not tables.slice_has_req[scenario, s] or n == 0 or not tables.slice_traffic[scenario, s] > 0:
            continue
        ues = tables.slice_ues[scenario, s, :n]
        out[ues] = poisson_from_tables(cdf[scenario, s], u[ues]).astype(np.float64) * 1e6
    return out


def make_cfg(S, U, R, G, Us, bandwidth_hz=100e6, hist_depth=10, max_age_cap=400, max_steps=1000,
             overfulfill=0.2, norm_traffic=120.0, norm_ues=5.0, norm_se=40.0) -> _Cfg:
    return _Cfg(S, U, R, G, Us, hist_depth, max_age_cap, max_steps, bandwidth_hz, overfulfill,
                norm_traffic, norm_ues, norm_se)


class OracleEnv:
    """One reference-shaped env (CommunicationEnv + IBSched) on the CPU."""

    def __init__(self, cfg: _Cfg):
        self.cfg = cfg
        self.S, self.U, self.R, self.Us = cfg.n_slices, cfg.n_ues, cfg.n_rbs, cfg.max_ues_slice
        self._h = C.c_void_p(lib().orc_env_create(C.byref(cfg)))
        self._sc_keep = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_env_destroy(self._h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def clear(self):
        lib().orc_env_clear(self._h)

    def set_scale_per_element(self, on: bool):
        """RANENV_F_SCALE_PER_ELEMENT's convention (include/ranenv.h): every sched * se element is scaled by BW / R before it is added."""
        lib().orc_env_set_scale_per_element(self._h, 1 if on else 0)

    def set_scenario(self, tables, idx: int):
        """``tables``: any object with the ScenarioTables array attributes; row ``idx``."""
        arrs, sc = [], _Scenario()
        for name, dt in _SC_FIELDS:
            a = np.ascontiguousarray(np.asarray(getattr(tables, name))[idx], dtype=dt)
            arrs.append(a)
            setattr(sc, name, a.ctypes.data)
        self._sc_keep = (arrs, sc)
        lib().orc_env_set_scenario(self._h, C.byref(sc))

    def reset(self, se_tile: np.ndarray):
        se = np.ascontiguousarray(se_tile, dtype=np.float32)
        assert se.size == self.U * self.R
        lib().orc_env_reset(self._h, _p(se))

    def action_format(self, inter_scores, intra_choice, want_dense=True):
        sc = np.ascontiguousarray(inter_scores, dtype=np.float64)
        ic = np.ascontiguousarray(intra_choice, dtype=np.int32)
        start = np.zeros(self.U, dtype=np.int32)
        count = np.zeros(self.U, dtype=np.int32)
        dense = np.zeros((self.U, self.R), dtype=np.uint8) if want_dense else None
        lib().orc_action_format(self._h, _p(sc), _p(ic), _p(start), _p(count), _p(dense))
        return start, count, dense

    def core_step(self, dense_mask, se_tile, traffic_bits):
        d = np.ascontiguousarray(dense_mask, dtype=np.uint8)
        se = np.ascontiguousarray(se_tile, dtype=np.float32)
        tr = np.ascontiguousarray(traffic_bits, dtype=np.float64)
        assert d.size == self.U * self.R and se.size == self.U * self.R and tr.size == self.U
        lib().orc_env_core_step(self._h, _p(d), _p(se), _p(tr))

    def step(self, inter_scores, intra_choice, se_tile, traffic_bits):
        sc = np.ascontiguousarray(inter_scores, dtype=np.float64)
        ic = np.ascontiguousarray(intra_choice, dtype=np.int32)
        se = np.ascontiguousarray(se_tile, dtype=np.float32)
        tr = np.ascontiguousarray(traffic_bits, dtype=np.float64)
        assert sc.size == self.S and ic.size == self.S and se.size == self.U * self.R and tr.size == self.U
        lib().orc_env_step(self._h, _p(sc), _p(ic), _p(se), _p(tr))

    def agent_observe(self, sent, dropped, occ, lat, se_tile, sched_rowsum):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        a = [f(sent), f(dropped), f(occ), f(lat)]
        se = np.ascontiguousarray(se_tile, dtype=np.float32)
        rs = f(sched_rowsum)
        lib().orc_agent_observe(self._h, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(se), _p(rs))

    def policy_marr(self):
        out = np.zeros(self.S)
        lib().orc_policy_marr(self._h, _p(out))
        return out

    def policy_mapf(self):
        out = np.zeros(self.S)
        lib().orc_policy_mapf(self._h, _p(out))
        return out

    def raw(self):
        names = ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts",
                 "buffer_occupancies", "buffer_latencies")
        arrs = [np.zeros(self.U) for _ in names]
        lib().orc_env_get_raw(self._h, *[_p(a) for a in arrs])
        return dict(zip(names, arrs))

    def obs(self):
        S, Us = self.S, self.Us
        oi = np.zeros(S * 10); mi = np.zeros(S, dtype=np.int8)
        oa = np.zeros((S, 2 * Us + 9)); ma = np.zeros((S, Us), dtype=np.int8)
        rw = np.zeros(S + 1)
        lib().orc_env_get_obs(self._h, _p(oi), _p(mi), _p(oa), _p(ma), _p(rw))
        return {"obs_inter": oi, "mask_inter": mi, "obs_intra": oa, "mask_intra": ma, "reward": rw}

    def set_pkt_throughputs(self, thr):
        a = np.ascontiguousarray(thr, dtype=np.float64)
        assert a.size == self.U
        lib().orc_env_set_pkt_throughputs(self._h, _p(a))

    def heads(self, usecase):
        """SchedTWC / SchedColORAN: (observation [10*S], reward_twc, reward_colran)."""
        uc = np.ascontiguousarray(usecase, dtype=np.int32)
        assert uc.size == self.S
        obs = np.zeros(10 * self.S)
        r1, r2 = C.c_double(0.0), C.c_double(0.0)
        lib().orc_env_get_heads(self._h, _p(uc), _p(obs), C.byref(r1), C.byref(r2))
        return obs, r1.value, r2.value

    def drift(self):
        d = np.zeros((self.S, self.Us, 3))
        lib().orc_env_get_drift(self._h, _p(d))
        return d

    def buffer(self, u: int):
        h = np.zeros(self.cfg.max_age_cap + 1, dtype=np.int64)
        lib().orc_env_get_buffer(self._h, C.c_int(u), _p(h))
        return h

    @property
    def step_number(self):
        return lib().orc_env_step_number(self._h)

    @property
    def hist_len(self):
        return lib().orc_env_hist_len(self._h)


def batch_step(envs: Sequence[OracleEnv], policy: int, scores, intra, se_pool, tile_index, traffic,
               n_threads: int = 1):
    n = len(envs)
    hs = (C.c_void_p * n)(*[e.handle for e in envs])
    sc = None if scores is None else np.ascontiguousarray(scores, dtype=np.float64)
    ic = np.ascontiguousarray(intra, dtype=np.int32)
    ti = np.ascontiguousarray(tile_index, dtype=np.int64)
    tr = np.ascontiguousarray(traffic, dtype=np.float64)
    assert se_pool.dtype == np.float32 and se_pool.flags.c_contiguous
    lib().orc_batch_step(hs, C.c_int(n), C.c_int(policy), _p(sc), _p(ic), _p(se_pool), _p(ti), _p(tr),
                         C.c_int(n_threads))


def batch_reset(envs: Sequence[OracleEnv], se_pool, tile_index, n_threads: int = 1):
    n = len(envs)
    hs = (C.c_void_p * n)(*[e.handle for e in envs])
    ti = np.ascontiguousarray(tile_index, dtype=np.int64)
    assert se_pool.dtype == np.float32 and se_pool.flags.c_contiguous
    lib().orc_batch_reset(hs, C.c_int(n), _p(se_pool), _p(ti), C.c_int(n_threads))
