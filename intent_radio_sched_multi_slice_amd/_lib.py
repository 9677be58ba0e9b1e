"""ctypes binding of the C ABI in include/ranenv.h (libranenv_hip.so).

There is no fallback: if the HIP library is missing the import of the product path fails
loudly, and every non-zero status from the library raises RanEnvError.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RANENV_LIB") or os.path.join(_HERE, "csrc", "libranenv_hip.so")

ABI_VERSION = 9
POLICY_EXTERNAL, POLICY_MARR, POLICY_MAPF = 0, 1, 2
INTRA_RR, INTRA_PF, INTRA_MT, INTRA_PER_SLICE = 0, 1, 2, 255
F_CLEAR_HISTORY_ON_RESET, F_NO_RAW_OUTPUT, F_SYNC_CHECK, F_SCALE_PER_ELEMENT = 0x1, 0x2, 0x4, 0x8
SE_STREAM, SE_GATHER = 0, 1

EXPORTS = (
    "ranenv_last_error", "ranenv_abi_version", "ranenv_create", "ranenv_destroy",
    "ranenv_load_scenarios", "ranenv_bind_se_pool", "ranenv_bind_traffic_pool", "ranenv_set_episodes",
    "ranenv_set_policy", "ranenv_reset", "ranenv_step", "ranenv_step_dense", "ranenv_profile_begin", "ranenv_profile_end", "ranenv_profile_ttis",
    "ranenv_get_views",
    "ranenv_launch_info", "ranenv_se_from_power", "ranenv_bind_head_outputs", "ranenv_set_slice_usecase",
    "ranenv_set_traffic_generator", "ranenv_set_max_steps", "ranenv_set_episode_table", "ranenv_set_autoreset",
    "ranenv_autoreset", "ranenv_get_poisson_tables", "ranenv_set_partitions", "ranenv_rollout", "ranenv_enable_metrics", "ranenv_get_metrics",
    "ranenv_step_range", "ranenv_set_se_mode", "ranenv_get_se_sidecars", "ranenv_step_part", "ranenv_wait_part",
    "ranenv_get_partition", "ranenv_get_part_stream", "ranenv_autoreset_part", "ranenv_set_option", "ranenv_get_option", "ranenv_profile_work", "ranenv_bind_se_gather_from_power",
    "ranenv_bind_se_pool_quad", "ranenv_se_retile_quad", "ranenv_packed_step_fits", "ranenv_selftest_ddiv",
)


class RanEnvError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("batch", C.c_int32), ("n_slices", C.c_int32),
        ("n_ues", C.c_int32), ("n_rbs", C.c_int32), ("rbs_per_rbg", C.c_int32), ("max_ues_slice", C.c_int32),
        ("hist_depth", C.c_int32), ("max_age_cap", C.c_int32), ("max_steps", C.c_int32),
        ("n_scenarios", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32),
        ("bandwidth_hz", C.c_double), ("overfulfill", C.c_double), ("norm_traffic", C.c_double),
        ("norm_ues", C.c_double), ("norm_se", C.c_double),
    ]


SCENARIO_FIELDS = (
    "slice_active", "slice_has_req", "slice_nues", "slice_ues", "slice_priority", "slice_traffic",
    "slice_buffer_size", "slice_buffer_latency", "slice_message_size", "slice_nparams", "param_metric",
    "param_op", "param_value", "sorted_slices", "ue_slice", "ue_pos", "ue_pkt_size", "ue_max_pkts", "ue_max_age",
)
SCENARIO_F64 = {"slice_priority", "slice_traffic", "param_value"}


class ScenarioTablesC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in SCENARIO_FIELDS]


class Episode(C.Structure):
    _fields_ = [
        ("scenario", C.c_int32), ("se_len", C.c_int32), ("se_base", C.c_int64), ("se_offset", C.c_int32),
        ("trf_len", C.c_int32), ("trf_base", C.c_int64), ("trf_offset", C.c_int32), ("reserved", C.c_int32),
    ]


VIEW_FIELDS = (
    ("pkt_incoming", "i4", "BU"), ("pkt_throughputs", "i4", "BU"), ("pkt_effective_thr", "i4", "BU"),
    ("dropped_pkts", "i4", "BU"), ("queue_pkts", "i4", "BU"), ("queue_age_sum", "i8", "BU"),
    ("rb_start", "i4", "BU"), ("rb_count", "i4", "BU"), ("se_mean", "f8", "BU"), ("win_sent", "i8", "BU"),
    ("win_dropped", "i8", "BU"), ("step_number", "i4", "B"), ("hist_len", "i4", "B"),
    ("mask_inter", "i1", "BS"), ("mask_intra", "i1", "BSK"), ("policy_scores", "f8", "BS"),
    ("episode_number", "i4", "B"), ("episodes", "i4", "BE"),
)


class Views(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _, _ in VIEW_FIELDS]


_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raises if it was not built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime; it has to be in the process before this library's
    # DT_NEEDED libamdhip64 is resolved, or two runtimes end up loaded and the second sees no GPU.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RanEnvError(
            f"{LIB_PATH} is missing: build it with `python -m intent_radio_sched_multi_slice_amd.csrc.build` "
            "(or __graft_entry__.build()). The env step has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise RanEnvError(f"{LIB_PATH} does not export {name}")
    lib.ranenv_last_error.restype = C.c_char_p
    lib.ranenv_last_error.argtypes = [C.c_void_p]
    lib.ranenv_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    lib.ranenv_destroy.argtypes = [C.c_void_p]
    lib.ranenv_load_scenarios.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(ScenarioTablesC), C.c_void_p]
    lib.ranenv_bind_se_pool.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
    lib.ranenv_bind_se_pool_quad.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
    lib.ranenv_se_retile_quad.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
    lib.ranenv_bind_traffic_pool.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.ranenv_set_episodes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ranenv_set_policy.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.ranenv_reset.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    lib.ranenv_step.argtypes = [C.c_void_p] + [C.c_void_p] * 9
    lib.ranenv_step_range.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 9
    lib.ranenv_step_part.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 9
    lib.ranenv_wait_part.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.ranenv_get_part_stream.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.ranenv_get_partition.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.ranenv_set_se_mode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.ranenv_get_se_sidecars.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]
    lib.ranenv_set_partitions.argtypes = [C.c_void_p, C.c_int32]
    lib.ranenv_rollout.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 5
    lib.ranenv_enable_metrics.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.ranenv_get_metrics.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]
    lib.ranenv_step_dense.argtypes = [C.c_void_p] + [C.c_void_p] * 8
    lib.ranenv_profile_begin.argtypes = [C.c_void_p]
    lib.ranenv_profile_end.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    lib.ranenv_profile_ttis.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.ranenv_get_views.argtypes = [C.c_void_p, C.POINTER(Views)]
    lib.ranenv_launch_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.ranenv_se_from_power.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_void_p]
    lib.ranenv_bind_head_outputs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ranenv_set_slice_usecase.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.ranenv_set_traffic_generator.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_void_p]
    lib.ranenv_get_poisson_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ranenv_set_max_steps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ranenv_set_episode_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.ranenv_set_autoreset.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.ranenv_autoreset.argtypes = [C.c_void_p] + [C.c_void_p] * 7
    lib.ranenv_autoreset_part.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 7
    lib.ranenv_bind_se_gather_from_power.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_void_p]
    lib.ranenv_profile_work.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.ranenv_selftest_ddiv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.ranenv_packed_step_fits.argtypes = [C.POINTER(Config), C.c_int64, C.c_int64]
    lib.ranenv_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.ranenv_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
    if lib.ranenv_abi_version() != ABI_VERSION:
        raise RanEnvError(f"ABI mismatch: library {lib.ranenv_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(lib, handle, status: int, what: str) -> None:
    if status != 0:
        msg = lib.ranenv_last_error(handle)
        raise RanEnvError(f"{what} failed ({status}): {msg.decode() if msg else 'unknown error'}")
