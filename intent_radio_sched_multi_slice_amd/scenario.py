"""Scenario tables: association + slice intents of an episode, flattened for HBM.

The reference carries a scenario as Python objects: ``basestation_slice_assoc (1,S)``,
``slice_ue_assoc (S,U)`` and the ``slice_req`` dict of dicts
(associations/mult_slice.py:58-347, :350-442).  The device path needs the same facts as
structure-of-arrays tables, one row per scenario, so that thousands of envs can share a
few hundred scenarios that stay L2-resident (the reference itself cycles 200 association
files, associations/mult_slice.py:33,444-452).

Host-side only (numpy); nothing here touches the GPU.
"""
from __future__ import annotations

from dataclasses import dataclass, fields
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

METRIC_THROUGHPUT, METRIC_RELIABILITY, METRIC_LATENCY = 0, 1, 2
METRIC_CODE = {"throughput": METRIC_THROUGHPUT, "reliability": METRIC_RELIABILITY, "latency": METRIC_LATENCY}
OP_GE, OP_LE, OP_EQ, OP_GT, OP_LT = 0, 1, 2, 3, 4
# expectation_params of associations/mult_slice.py:48-55 (np.isin on a scalar is equality)
_OP_BY_UFUNC = {
    "greater_equal": OP_GE, "less_equal": OP_LE, "equal": OP_EQ,
    "greater": OP_GT, "less": OP_LT, "isin": OP_EQ,
}
OP_NAME = {"at_least": OP_GE, "at_most": OP_LE, "exactly": OP_EQ, "greater": OP_GT, "smaller": OP_LT, "one_of": OP_EQ}
OP_UFUNC = {OP_GE: np.greater_equal, OP_LE: np.less_equal, OP_EQ: np.equal, OP_GT: np.greater, OP_LT: np.less}

INTRA_RR, INTRA_PF, INTRA_MT = 0, 1, 2

# Defaults of a UE that belongs to no slice.  gen_assoc_mult_slice.py:92-97 builds
# UEs(max_number_ues, latency=100, buffer=1024, pkt_size=100) before any association.
IDLE_UE_MAX_AGE, IDLE_UE_MAX_PKTS, IDLE_UE_PKT_SIZE = 100, 1024, 100


def _op_code(op) -> int:
    if isinstance(op, (int, np.integer)):
        return int(op)
    if isinstance(op, str):
        return OP_NAME[op]
    name = getattr(op, "__name__", None)
    if name in _OP_BY_UFUNC:
        return _OP_BY_UFUNC[name]
    raise ValueError(f"unsupported intent operator {op!r}")


# The ten slice templates of associations/mult_slice.py:58-347, as rows:
# (name, priority, [(metric, value, op)...], buffer_size pkts, buffer_latency TTIs,
#  message_size bits, mobility km/h, traffic Mbps, min_ues, max_ues)
_KB = 1024
SLICE_TEMPLATES: Tuple[tuple, ...] = (
    ("control_case_2", 1, (("reliability", 99.999999, "at_least"), ("latency", 50, "at_most")),
     10 * _KB, 100, 8 * _KB, 0, 5, 4, 5),
    ("monitoring_case_1", 0, (("throughput", 10, "at_least"),),
     10 * _KB, 100, 8 * _KB, 72, 10, 4, 5),
    ("robotic_surgery_case_1", 1,
     (("reliability", 99.9999, "at_least"), ("latency", 20, "at_most"), ("throughput", 30, "at_least")),
     1000 * _KB, 40, 2000 * 8, 0, 30, 4, 5),
    ("robotic_diagnosis", 0,
     (("reliability", 99.999, "at_least"), ("latency", 20, "at_most"), ("throughput", 15, "at_least")),
     1000 * _KB, 40, 80 * 8, 0, 15, 4, 5),
    ("medical_monitoring", 0,
     (("reliability", 99.9999, "at_least"), ("latency", 100, "at_most"), ("throughput", 10, "at_least")),
     10 * _KB, 200, 1000 * 8, 0, 10, 4, 5),
    ("uav_app_case_1", 1, (("latency", 200, "at_most"), ("throughput", 100, "at_least")),
     1000 * _KB, 400, 8192 * 8, 30, 100, 2, 4),
    ("uav_control_non_vlos", 1,
     (("reliability", 99.99, "at_least"), ("latency", 140, "at_most"), ("throughput", 20, "at_least")),
     10 * _KB, 300, 8192 * 8, 30, 20, 4, 5),
    ("vr_gaming", 0,
     (("reliability", 99.99, "at_least"), ("latency", 10, "at_most"), ("throughput", 100, "at_least")),
     1000 * _KB, 20, 8192 * 8, 0, 100, 2, 4),
    ("cloud_gaming", 0, (("latency", 80, "at_most"), ("throughput", 50, "at_least")),
     10 * _KB, 160, 8192 * 8, 0, 50, 2, 5),
    ("video_streaming_4k", 0, (("throughput", 30, "at_least"),),
     10 * _KB, 100, 8192 * 8, 0, 30, 2, 5),
)
SLICE_TYPE_NAMES = tuple(t[0] for t in SLICE_TEMPLATES)
# SchedColORAN's slice-name -> use-case table (agents/sched_colran.py:356-367) as a bitmask:
# bit 0 = eMBB (throughput rewarded), bit 1 = URLLC (buffered Mbit penalised)
USECASE_EMBB, USECASE_URLLC = 1, 2
SLICE_USECASE = {
    "control_case_2": 2, "monitoring_case_1": 1, "robotic_surgery_case_1": 3, "robotic_diagnosis": 2,
    "medical_monitoring": 1, "uav_app_case_1": 1, "uav_control_non_vlos": 1, "vr_gaming": 3,
    "cloud_gaming": 1, "video_streaming_4k": 1,
}


def slice_usecase_from_req(slice_req: dict, n_slices: int) -> np.ndarray:
    """[S] int32 use-case bits of a reference ``slice_req`` dict (0 for empty or unknown slices)."""
    out = np.zeros(n_slices, dtype=np.int32)
    for s in range(n_slices):
        req = (slice_req or {}).get(f"slice_{s}", {})
        if req:
            out[s] = SLICE_USECASE.get(req.get("name", ""), 0)
    return out
MAX_AGE_CAP_DEFAULT = max(t[4] for t in SLICE_TEMPLATES)  # 400 TTIs (uav_app_case_1)


def slice_template_dict(type_idx: int) -> dict:
    """One template in the reference's ``slice_req[slice]`` dict form."""
    name, prio, params, bsize, blat, msg, mob, traffic, mn, mx = SLICE_TEMPLATES[type_idx]
    return {
        "name": name,
        "priority": prio,
        "parameters": {
            f"par{i + 1}": {"name": m, "value": v, "unit": "", "operator": OP_UFUNC[OP_NAME[op]]}
            for i, (m, v, op) in enumerate(params)
        },
        "ues": {
            "buffer_size": bsize, "buffer_latency": blat, "message_size": msg, "mobility": mob,
            "traffic": traffic, "min_number_ues": mn, "max_number_ues": mx,
        },
    }


def stable_sort_slices(slice_nues: np.ndarray, slice_traffic: np.ndarray, slice_has_req: np.ndarray) -> np.ndarray:
    """IBSched.sort_slices (agents/ib_sched.py:351-370) with the build's tie rule.

    The reference calls ``np.argsort`` with the default kind, whose order among equal
    keys depends on the numpy build; the canonical rule here is the stable one.
    """
    key = slice_nues.astype(np.float64) * np.where(slice_has_req != 0, slice_traffic, 0.0)
    return np.argsort(key, kind="stable").astype(np.int32)


_I32_SLICE = ("slice_active", "slice_has_req", "slice_nues", "slice_buffer_size",
              "slice_buffer_latency", "slice_message_size", "slice_nparams", "sorted_slices")
_F64_SLICE = ("slice_priority", "slice_traffic")


@dataclass
class ScenarioTables:
    """``NS`` scenarios as contiguous numpy tables (row ``i`` = scenario ``i``)."""
    slice_active: np.ndarray          # (NS,S)    i32  basestation_slice_assoc[0]
    slice_has_req: np.ndarray         # (NS,S)    i32  slice_req[s] != {}
    slice_nues: np.ndarray            # (NS,S)    i32
    slice_ues: np.ndarray             # (NS,S,Us) i32  ascending UE ids, -1 pad
    slice_priority: np.ndarray        # (NS,S)    f64
    slice_traffic: np.ndarray         # (NS,S)    f64  Mbps
    slice_buffer_size: np.ndarray     # (NS,S)    i32  pkts
    slice_buffer_latency: np.ndarray  # (NS,S)    i32  TTIs
    slice_message_size: np.ndarray    # (NS,S)    i32  bits
    slice_nparams: np.ndarray         # (NS,S)    i32
    param_metric: np.ndarray          # (NS,S,3)  i32
    param_op: np.ndarray              # (NS,S,3)  i32
    param_value: np.ndarray           # (NS,S,3)  f64
    sorted_slices: np.ndarray         # (NS,S)    i32  slice index at sorted position
    ue_slice: np.ndarray              # (NS,U)    i32  -1 = idle
    ue_pos: np.ndarray                # (NS,U)    i32  position inside its slice
    ue_pkt_size: np.ndarray           # (NS,U)    i32
    ue_max_pkts: np.ndarray           # (NS,U)    i32
    ue_max_age: np.ndarray            # (NS,U)    i32

    @property
    def n_scenarios(self) -> int:
        return self.slice_active.shape[0]

    @property
    def n_slices(self) -> int:
        return self.slice_active.shape[1]

    @property
    def n_ues(self) -> int:
        return self.ue_slice.shape[1]

    @property
    def max_ues_slice(self) -> int:
        return self.slice_ues.shape[2]

    @staticmethod
    def empty(n_scenarios: int, n_slices: int, n_ues: int, max_ues_slice: int) -> "ScenarioTables":
        ns, s, u, us = n_scenarios, n_slices, n_ues, max_ues_slice
        z32 = lambda *sh: np.zeros(sh, dtype=np.int32)
        z64 = lambda *sh: np.zeros(sh, dtype=np.float64)
        t = ScenarioTables(
            slice_active=z32(ns, s), slice_has_req=z32(ns, s), slice_nues=z32(ns, s),
            slice_ues=np.full((ns, s, us), -1, dtype=np.int32),
            slice_priority=z64(ns, s), slice_traffic=z64(ns, s),
            slice_buffer_size=z32(ns, s), slice_buffer_latency=z32(ns, s), slice_message_size=z32(ns, s),
            slice_nparams=z32(ns, s), param_metric=z32(ns, s, 3), param_op=z32(ns, s, 3),
            param_value=z64(ns, s, 3),
            sorted_slices=np.tile(np.arange(s, dtype=np.int32), (ns, 1)),
            ue_slice=np.full((ns, u), -1, dtype=np.int32), ue_pos=z32(ns, u),
            ue_pkt_size=np.full((ns, u), IDLE_UE_PKT_SIZE, dtype=np.int32),
            ue_max_pkts=np.full((ns, u), IDLE_UE_MAX_PKTS, dtype=np.int32),
            ue_max_age=np.full((ns, u), IDLE_UE_MAX_AGE, dtype=np.int32),
        )
        return t

    def set_from_reference(
        self,
        idx: int,
        basestation_slice_assoc: np.ndarray,
        slice_ue_assoc: np.ndarray,
        slice_req: dict,
        enable_sort_slices: bool = True,
        ue_params: Optional[Tuple[np.ndarray, np.ndarray, np.ndarray]] = None,
    ) -> None:
        """Fill row ``idx`` from the reference's objects.

        ``ue_params`` = (pkt_sizes, max_buffer_pkts, max_buffer_latencies) as the env core's
        ``UEs`` object holds them; when None they follow
        MultSliceAssociation.update_ues (associations/mult_slice.py:468-488): every UE of a
        slice gets the slice's message_size / buffer_size / buffer_latency.
        """
        S, U, Us = self.n_slices, self.n_ues, self.max_ues_slice
        bsa = np.asarray(basestation_slice_assoc).reshape(-1, S)[0]
        sua = np.asarray(slice_ue_assoc).reshape(S, U)
        if np.any(sua.sum(axis=0) > 1):
            raise ValueError("UE associated with more than one slice")  # gen_assoc_mult_slice.py:194-195
        self.slice_active[idx] = (bsa != 0).astype(np.int32)
        self.slice_ues[idx] = -1
        self.ue_slice[idx] = -1
        self.ue_pos[idx] = 0
        if ue_params is not None:
            self.ue_pkt_size[idx] = np.asarray(ue_params[0], dtype=np.int64)
            self.ue_max_pkts[idx] = np.asarray(ue_params[1], dtype=np.int64)
            self.ue_max_age[idx] = np.asarray(ue_params[2], dtype=np.int64)
        else:
            self.ue_pkt_size[idx] = IDLE_UE_PKT_SIZE
            self.ue_max_pkts[idx] = IDLE_UE_MAX_PKTS
            self.ue_max_age[idx] = IDLE_UE_MAX_AGE
        for s in range(S):
            req = (slice_req or {}).get(f"slice_{s}", {})
            ues = np.nonzero(sua[s])[0]
            n = len(ues)
            if n > Us:
                raise ValueError(f"slice {s} has {n} UEs, max_number_ues_slice is {Us}")
            self.slice_nues[idx, s] = n
            self.slice_ues[idx, s, :n] = ues
            self.ue_slice[idx, ues] = s
            self.ue_pos[idx, ues] = np.arange(n)
            self.slice_has_req[idx, s] = 1 if req else 0
            self.slice_nparams[idx, s] = 0
            self.param_metric[idx, s] = 0
            self.param_op[idx, s] = 0
            self.param_value[idx, s] = 0.0
            for name in ("slice_priority", "slice_traffic"):
                getattr(self, name)[idx, s] = 0.0
            for name in ("slice_buffer_size", "slice_buffer_latency", "slice_message_size"):
                getattr(self, name)[idx, s] = 0
            if not req:
                continue
            self.slice_priority[idx, s] = float(req.get("priority", 0))
            u = req["ues"]
            self.slice_traffic[idx, s] = float(u["traffic"])
            self.slice_buffer_size[idx, s] = int(u["buffer_size"])
            self.slice_buffer_latency[idx, s] = int(u["buffer_latency"])
            self.slice_message_size[idx, s] = int(u["message_size"])
            params = list(req.get("parameters", {}).values())
            if len(params) > 3:
                raise ValueError("at most 3 intent parameters per slice")
            seen = set()
            for p, par in enumerate(params):
                m = METRIC_CODE[par["name"]]
                if m in seen:
                    raise ValueError("an intent metric may be declared once per slice")
                seen.add(m)
                self.param_metric[idx, s, p] = m
                self.param_op[idx, s, p] = _op_code(par["operator"])
                self.param_value[idx, s, p] = float(par["value"])
            self.slice_nparams[idx, s] = len(params)
            if ue_params is None and n:
                self.ue_pkt_size[idx, ues] = int(u["message_size"])
                self.ue_max_pkts[idx, ues] = int(u["buffer_size"])
                self.ue_max_age[idx, ues] = int(u["buffer_latency"])
        if enable_sort_slices:
            self.sorted_slices[idx] = stable_sort_slices(
                self.slice_nues[idx], self.slice_traffic[idx], self.slice_has_req[idx])
        else:
            self.sorted_slices[idx] = np.arange(S, dtype=np.int32)

    def to_reference(self, idx: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray, dict]:
        """Row ``idx`` back as (basestation_ue_assoc, basestation_slice_assoc, slice_ue_assoc, slice_req)."""
        S, U = self.n_slices, self.n_ues
        sua = np.zeros((S, U))
        slice_req: Dict[str, dict] = {}
        for s in range(S):
            n = int(self.slice_nues[idx, s])
            sua[s, self.slice_ues[idx, s, :n]] = 1
            if not self.slice_has_req[idx, s]:
                slice_req[f"slice_{s}"] = {}
                continue
            params = {}
            for p in range(int(self.slice_nparams[idx, s])):
                m = int(self.param_metric[idx, s, p])
                params[f"par{p + 1}"] = {
                    "name": [k for k, v in METRIC_CODE.items() if v == m][0],
                    "value": float(self.param_value[idx, s, p]),
                    "unit": "",
                    "operator": OP_UFUNC[int(self.param_op[idx, s, p])],
                }
            slice_req[f"slice_{s}"] = {
                "name": f"slice_{s}",
                "priority": float(self.slice_priority[idx, s]),
                "parameters": params,
                "ues": {
                    "buffer_size": int(self.slice_buffer_size[idx, s]),
                    "buffer_latency": int(self.slice_buffer_latency[idx, s]),
                    "message_size": int(self.slice_message_size[idx, s]),
                    "mobility": 0,
                    "traffic": float(self.slice_traffic[idx, s]),
                },
            }
        bsa = self.slice_active[idx].astype(np.float64)[None, :]
        bua = sua.sum(axis=0)[None, :]
        return bua, bsa, sua, slice_req

    def arrays(self) -> Dict[str, np.ndarray]:
        return {f.name: getattr(self, f.name) for f in fields(self)}

    @staticmethod
    def from_arrays(d: Dict[str, np.ndarray]) -> "ScenarioTables":
        return ScenarioTables(**{f.name: np.ascontiguousarray(d[f.name]) for f in fields(ScenarioTables)})

    def validate(self, max_age_cap: int, n_rbs: int, bandwidth_hz: float, max_se: float = 64.0) -> None:
        """Refuse tables the int32 device state cannot represent."""
        if self.n_scenarios == 0:
            return
        if int(self.ue_max_age.max()) > max_age_cap:
            raise ValueError(f"buffer_latency {int(self.ue_max_age.max())} exceeds max_age_cap {max_age_cap}")
        if int(self.ue_pkt_size.min()) <= 0 or int(self.ue_max_pkts.min()) <= 0:
            raise ValueError("pkt_size and max_buffer_pkts must be positive")
        worst = bandwidth_hz * max_se / float(self.ue_pkt_size.min())
        if worst >= 2 ** 31:
            raise ValueError("per-TTI packet capacity may overflow int32; raise message_size")
        if int(self.slice_nues.max()) > self.max_ues_slice:
            raise ValueError("slice_nues exceeds max_ues_slice")


def generate_reference_scenario(
    rng: np.random.Generator, n_slices: int, n_ues: int, min_slices: int = 3,
) -> Tuple[np.ndarray, np.ndarray, np.ndarray, dict, np.ndarray]:
    """MultSliceAssociation generator mode, step 0 (associations/mult_slice.py:359-423).

    Consumes ``rng`` in exactly the reference's call order, so the same seed yields the
    same scenario.  Returns (basestation_ue_assoc, basestation_slice_assoc, slice_ue_assoc,
    slice_req, slices_to_use).
    """
    bsa = np.zeros((1, n_slices))
    sua = np.zeros((n_slices, n_ues))
    bua = np.zeros((1, n_ues))
    number_slices = rng.integers(low=min_slices, high=n_slices, endpoint=True)          # :361-365
    slices_to_use = rng.choice(np.arange(n_slices), number_slices, replace=False)       # :366-370
    bsa[0, slices_to_use] = 1
    slice_req: Dict[str, dict] = {f"slice_{i}": {} for i in range(n_slices)}
    types = rng.choice(len(SLICE_TEMPLATES), len(slices_to_use), replace=False)          # :457-459
    for i, t in enumerate(types):
        slice_req[f"slice_{slices_to_use[i]}"] = slice_template_dict(int(t))
    ues_per_slice = np.array([                                                           # :377-391
        rng.integers(slice_req[f"slice_{s}"]["ues"]["min_number_ues"],
                     slice_req[f"slice_{s}"]["ues"]["max_number_ues"], 1, endpoint=True)
        for s in slices_to_use
    ]).flatten()
    active_ues = np.array(rng.choice((bua[0] == 0).nonzero()[0], int(np.sum(ues_per_slice)), replace=False))
    used = 0
    for k, s in enumerate(slices_to_use):                                                # :399-411
        sua[s, active_ues[used:used + ues_per_slice[k]]] = 1
        used += ues_per_slice[k]
    bua = np.array([np.sum(sua, axis=0)])
    return bua, bsa, sua, slice_req, slices_to_use


def generate_scaled_scenarios(
    n_scenarios: int, seed: int, n_slices: int = 10, n_ues: int = 100, max_ues_slice: int = 10,
    min_slices: int = 6, min_ues: int = 4, enable_sort_slices: bool = True,
) -> ScenarioTables:
    """BASELINE configs 2-5: ``min_slices..n_slices`` active slices with distinct templates,
    ``min_ues..max_ues_slice`` UEs each, remaining UEs idle."""
    rng = np.random.default_rng(seed)
    t = ScenarioTables.empty(n_scenarios, n_slices, n_ues, max_ues_slice)
    for i in range(n_scenarios):
        n_act = int(rng.integers(min_slices, n_slices, endpoint=True))
        slices = rng.choice(n_slices, n_act, replace=False)
        types = rng.choice(len(SLICE_TEMPLATES), n_act, replace=n_act > len(SLICE_TEMPLATES))
        counts = rng.integers(min_ues, max_ues_slice, n_act, endpoint=True)
        while counts.sum() > n_ues:
            counts[np.argmax(counts)] -= 1
        ues = rng.choice(n_ues, int(counts.sum()), replace=False)
        bsa = np.zeros((1, n_slices)); sua = np.zeros((n_slices, n_ues))
        req = {f"slice_{s}": {} for s in range(n_slices)}
        used = 0
        for k, s in enumerate(slices):
            bsa[0, s] = 1
            req[f"slice_{s}"] = slice_template_dict(int(types[k]))
            sua[s, ues[used:used + counts[k]]] = 1
            used += counts[k]
        t.set_from_reference(i, bsa, sua, req, enable_sort_slices)
    return t


# --------------------------------------------------------------------------------------------
# the reference's scenario files (associations/data/{scenario}/ep_N.npz)
# --------------------------------------------------------------------------------------------
EPISODE_FILE_KEYS = ("hist_basestation_ue_assoc", "hist_basestation_slice_assoc", "hist_slice_ue_assoc",
                     "hist_slice_req", "hist_slices_lifetime", "hist_slices_to_use")


def save_episode_npz(path: str, basestation_ue_assoc: np.ndarray, basestation_slice_assoc: np.ndarray,
                     slice_ue_assoc: np.ndarray, slice_req: dict, slices_to_use: np.ndarray, n_steps: int) -> None:
    """Write one scenario in the schema of gen_assoc_mult_slice.py:229-237: per-step histories (the
    scenario is constant over the episode, so the step-0 objects are repeated), dicts and ragged
    arrays pickled inside object arrays."""
    hist_req = np.empty(n_steps, dtype=object)
    hist_use = np.empty(n_steps, dtype=object)
    for t in range(n_steps):
        hist_req[t] = slice_req
        hist_use[t] = np.asarray(slices_to_use)
    np.savez_compressed(
        path,
        hist_basestation_ue_assoc=np.repeat(np.asarray(basestation_ue_assoc)[None], n_steps, axis=0),
        hist_basestation_slice_assoc=np.repeat(np.asarray(basestation_slice_assoc)[None], n_steps, axis=0),
        hist_slice_ue_assoc=np.repeat(np.asarray(slice_ue_assoc)[None], n_steps, axis=0),
        hist_slice_req=hist_req,
        hist_slices_lifetime=np.zeros((n_steps, np.asarray(slice_ue_assoc).shape[0])),
        hist_slices_to_use=hist_use,
    )


def load_episode_npz(path: str) -> Dict[str, np.ndarray]:
    """Read a scenario file the way MultSliceAssociation.load_episode_data does
    (associations/mult_slice.py:490-508).  The file holds pickled objects: only open files you trust."""
    with np.load(path, allow_pickle=True, mmap_mode=None) as f:
        missing = [k for k in EPISODE_FILE_KEYS if k not in f.files]
        if missing:
            raise ValueError(f"{path}: not a scenario file, missing {missing}")
        return {k: f[k] for k in EPISODE_FILE_KEYS}


def tables_from_episode_files(paths, n_slices: int, n_ues: int, max_ues_slice: int, step: int = 0,
                              enable_sort_slices: bool = True) -> ScenarioTables:
    """One scenario-pool row per ep_N.npz (its ``step``-th entry; mult_slice scenarios do not change
    inside an episode) -- the ingest of SURVEY section 8f-1 for the association side."""
    paths = list(paths)
    tabs = ScenarioTables.empty(len(paths), n_slices, n_ues, max_ues_slice)
    for i, path in enumerate(paths):
        ep = load_episode_npz(path)
        sua = np.asarray(ep["hist_slice_ue_assoc"][step], dtype=np.float64)
        bsa = np.asarray(ep["hist_basestation_slice_assoc"][step], dtype=np.float64)
        if sua.shape != (n_slices, n_ues):
            raise ValueError(f"{path}: slice_ue_assoc is {sua.shape}, expected {(n_slices, n_ues)}")
        tabs.set_from_reference(i, bsa, sua, ep["hist_slice_req"][step], enable_sort_slices)
    return tabs
