"""BatchedRanEnv: B independent RAN-slicing environments stepped by one HIP launch.

Host side of the C ABI in include/ranenv.h.  PyTorch is plumbing here (device memory,
streams); every number is produced by the gfx950 kernels in csrc/ranenv_step_body.hpp (+ ranenv_aux.hip).

gymnasium-style surface, batched (reference: ``env.reset`` / ``env.step`` simu.py:547-566):

    env = BatchedRanEnv(batch=4096, n_slices=10, n_ues=100, n_rbs=135, rbs_per_rbg=1,
                        max_ues_slice=10, n_scenarios=200)
    env.load_scenarios(tables)                     # association + slice intents
    env.bind_se_pool(se)                           # float32 [tiles, R, U] on the GPU
    env.bind_traffic_pool(bits)                    # int32   [rows, U]    on the GPU
    env.set_episodes(scenario=..., se_base=..., se_len=..., trf_base=..., trf_len=...)
    obs = env.reset()
    obs, reward, done = env.step(inter_scores, intra_choice)   # or env.step() with a device policy
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import (F_CLEAR_HISTORY_ON_RESET, F_NO_RAW_OUTPUT, F_SYNC_CHECK, INTRA_MT, INTRA_PER_SLICE, INTRA_PF, INTRA_RR,
                   POLICY_EXTERNAL, POLICY_MAPF, POLICY_MARR, SE_GATHER, SE_STREAM, RanEnvError)
from .scenario import MAX_AGE_CAP_DEFAULT, ScenarioTables

_TORCH_DT = {"i1": torch.int8, "i4": torch.int32, "i8": torch.int64, "f8": torch.float64, "f4": torch.float32}


class _DevArray:
    """Zero-copy window on handle-owned device memory (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, shape, typestr: str, owner):
        self.__cuda_array_interface__ = {
            "shape": tuple(int(x) for x in shape), "typestr": "<" + typestr if typestr != "i1" else "|i1",
            "data": (int(ptr), False), "version": 2, "strides": None,
        }
        self._owner = owner


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else C.c_void_p(t.data_ptr())


class BatchedRanEnv:
    def __init__(self, batch: int, n_slices: int, n_ues: int, n_rbs: int, rbs_per_rbg: int = 1,
                 max_ues_slice: Optional[int] = None, n_scenarios: int = 1, bandwidth_hz: float = 100e6,
                 max_steps: int = 1000, hist_depth: int = 10, max_age_cap: int = MAX_AGE_CAP_DEFAULT,
                 overfulfill: float = 0.2, norm_traffic: float = 120.0, norm_ues: float = 5.0,
                 norm_se: float = 40.0, device: Optional[torch.device] = None, flags: int = 0,
                 strict_inputs: bool = False):
        """``strict_inputs``: refuse per-step inputs that are not already contiguous tensors of the right dtype on
        the env's GPU, instead of converting them (a conversion is a hidden host-to-device copy every TTI)."""
        if not torch.cuda.is_available():
            raise RanEnvError("BatchedRanEnv needs a ROCm GPU (there is no CPU fallback)")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise RanEnvError(f"device must be a GPU, got {self.device}")
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        self.B, self.S, self.U, self.R = int(batch), int(n_slices), int(n_ues), int(n_rbs)
        self.G = int(rbs_per_rbg)
        self.Us = int(max_ues_slice if max_ues_slice is not None else max(1, n_ues // n_slices))
        self.W = 2 * self.Us + 9
        self.max_steps, self.hist_depth, self.max_age_cap = int(max_steps), int(hist_depth), int(max_age_cap)
        self.bandwidth_hz = float(bandwidth_hz)
        self.n_scenarios = int(n_scenarios)
        self.strict_inputs = bool(strict_inputs)
        self._lib = _lib.load()
        cfg = _lib.Config(_lib.ABI_VERSION, dev_index, self.B, self.S, self.U, self.R, self.G, self.Us,
                          self.hist_depth, self.max_age_cap, self.max_steps, self.n_scenarios, int(flags), 0,
                          self.bandwidth_hz, overfulfill, norm_traffic, norm_ues, norm_se)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            st = self._lib.ranenv_create(C.byref(cfg), C.byref(self._h))
        if st != 0:
            msg = self._lib.ranenv_last_error(None)
            raise RanEnvError(f"ranenv_create failed ({st}): {msg.decode() if msg else ''}")
        self._keep: Dict[str, object] = {}
        self._views: Optional[Dict[str, torch.Tensor]] = None
        B, S = self.B, self.S
        dev = self.device
        self.obs_inter = torch.zeros((B, S * 10), dtype=torch.float32, device=dev)
        self.obs_intra = torch.zeros((B, S, self.W), dtype=torch.float32, device=dev)
        self.reward = torch.zeros((B, S + 1), dtype=torch.float64, device=dev)
        self.done = torch.zeros((B,), dtype=torch.uint8, device=dev)
        self.tables: Optional[ScenarioTables] = None
        self.episodes: Optional[np.ndarray] = None
        # cached ctypes views of the output buffers (the per-step call is on the hot path)
        self._obs_dict = {"obs_inter": self.obs_inter, "obs_intra": self.obs_intra}
        self._p_out = (_ptr(self.obs_inter), _ptr(self.obs_intra), _ptr(self.reward), _ptr(self.done))
        self._step_fn = self._lib.ranenv_step
        self.policy, self.fixed_intra = POLICY_MARR, INTRA_RR      # the library's defaults (ranenv_create)
        self._recorder = None
        self._autoreset = False
        self.term_obs_inter = self.term_obs_intra = self.term_head_obs = None
        self.se_mode = "stream"
        self._ranges = None          # set_ranges(): [(lo, hi)] for step_async / step_wait
        # (the views are handed out once, here: ranenv_get_views ends the library's host shadow of the step counters -- the views are writable --,
        # and a first views() call in the middle of an auto-reset loop would switch the shortcut of enable_autoreset off until the next full reset)
        self.views()

    # ------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.ranenv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, status: int, what: str):
        _lib.check(self._lib, self._h, status, what)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, x, dtype, shape, name):
        if x is None:
            return None
        if self.strict_inputs and not (isinstance(x, torch.Tensor) and x.device == self.device and x.dtype == dtype
                                       and x.is_contiguous()):
            raise RanEnvError(f"{name}: strict_inputs needs a contiguous {dtype} tensor on {self.device}")
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.array(x, copy=True))
        t = t.to(device=self.device, dtype=dtype).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise RanEnvError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    # ------------------------------------------------------------------------------------------
    def load_scenarios(self, tables: ScenarioTables, first: int = 0):
        """Association + slice intents (associations/mult_slice.py:350-488) into the pool."""
        if (tables.n_slices, tables.n_ues, tables.max_ues_slice) != (self.S, self.U, self.Us):
            raise RanEnvError("scenario tables do not match the env sizes")
        tables.validate(self.max_age_cap, self.R, self.bandwidth_hz)
        ct = _lib.ScenarioTablesC()
        keep = []
        for name in _lib.SCENARIO_FIELDS:
            dt = np.float64 if name in _lib.SCENARIO_F64 else np.int32
            a = np.ascontiguousarray(getattr(tables, name), dtype=dt)
            keep.append(a)
            setattr(ct, name, a.ctypes.data)
        with torch.cuda.device(self.device):
            self._check(self._lib.ranenv_load_scenarios(self._h, int(first), tables.n_scenarios, C.byref(ct),
                                                         self._stream()), "ranenv_load_scenarios")
        if first == 0 and tables.n_scenarios == self.n_scenarios:
            self.tables = tables

    def bind_se_pool(self, se_pool: torch.Tensor, layout: Optional[str] = None, keep_rb_major: bool = False):
        """float32 [n_tiles, R, U] (RB-major tiles, the order of the reference's .mat: channels/quadriga.py:70-72) resident on
        this GPU.  ``layout``: how the library replays it -- ``"quad"`` (default; ``RANENV_SE_LAYOUT`` overrides): a COPY in
        RB-quad-major order [n_tiles, ceil(R/4), U, 4] is made once (ranenv_se_retile_quad, the same size again) and bound
        (ranenv_bind_se_pool_quad): 16-byte loads, a quarter of the memory instructions, bit-identical results.  The handle
        then does NOT alias the caller's tensor: writes to ``se_pool`` after this call are not seen (bind again), and this
        object keeps no reference to the RB-major tensor -- the caller may drop it and halve the pool's footprint -- unless
        ``keep_rb_major`` (then it is kept in ``self.se_pool_rb_major``).  ``"rb"``: the tensor itself is bound, zero-copy
        (ranenv_bind_se_pool).  A 4-D [n_tiles, ceil(R/4), U, 4] tensor is taken as already re-tiled (zero-copy)."""
        if se_pool.dtype != torch.float32 or se_pool.device != self.device or not se_pool.is_contiguous():
            raise RanEnvError("SE pool must be a contiguous float32 tensor on the env's GPU")
        Rq = (self.R + 3) // 4
        if se_pool.dim() == 4:
            if tuple(se_pool.shape[1:]) != (Rq, self.U, 4):
                raise RanEnvError(f"RB-quad-major SE pool must be [tiles, {Rq}, U={self.U}, 4], got {tuple(se_pool.shape)}")
            quad = se_pool
        else:
            if se_pool.dim() != 3 or se_pool.shape[1] != self.R or se_pool.shape[2] != self.U:
                raise RanEnvError(f"SE pool must be [tiles, R={self.R}, U={self.U}], got {tuple(se_pool.shape)}")
            layout = layout or os.environ.get("RANENV_SE_LAYOUT") or "quad"
            if layout not in ("quad", "rb"):
                raise RanEnvError("layout must be 'quad' or 'rb'")
            quad = None
            if layout == "quad":
                quad = torch.empty((se_pool.shape[0], Rq, self.U, 4), dtype=torch.float32, device=self.device)
                with torch.cuda.device(self.device):
                    st = self._lib.ranenv_se_retile_quad(_ptr(se_pool), _ptr(quad), se_pool.shape[0], self.U, self.R, self._stream())
                if st != 0:
                    raise RanEnvError("ranenv_se_retile_quad failed: " + (self._lib.ranenv_last_error(None) or b"").decode())
                # the copy is read by launches on other streams (partitions, ranges): a one-off build ends with a synchronisation, like
                # the gather sidecars', instead of an event every such stream would have to wait for
                torch.cuda.current_stream(self.device).synchronize()
        self.se_pool_rb_major = se_pool if (keep_rb_major and se_pool.dim() == 3) else None
        if quad is not None:
            self._keep["se_pool"] = quad
            self._check(self._lib.ranenv_bind_se_pool_quad(self._h, _ptr(quad), quad.shape[0], Rq * self.U * 4), "ranenv_bind_se_pool_quad")
            self.se_layout = "quad"
        else:
            self._keep["se_pool"] = se_pool
            self._check(self._lib.ranenv_bind_se_pool(self._h, _ptr(se_pool), se_pool.shape[0], self.R * self.U),
                        "ranenv_bind_se_pool")
            self.se_layout = "rb"
        self.se_mode = "stream"
        if os.environ.get("RANENV_SE_MODE") == "gather":       # experiment / test knob, like RANENV_SMALL_BATCH
            self.set_se_mode("gather")

    @property
    def bound_se_pool(self) -> Optional[torch.Tensor]:
        """The SE pool tensor the handle replays (RB-quad-major 4-D, or the caller's RB-major 3-D tensor under layout "rb")."""
        return self._keep.get("se_pool")

    def pooled_tiles(self, tile_index: torch.Tensor) -> torch.Tensor:
        """Tiles of the bound pool as float32 [n, R, U] (RB-major, the reference's order) whatever layout is bound."""
        pool = self._keep["se_pool"]
        t = pool.index_select(0, tile_index.to(torch.int64))
        if t.dim() == 4:                              # RB-quad-major [n, Rq, U, 4] -> [n, 4 Rq, U] -> the first R RBs
            t = t.permute(0, 1, 3, 2).reshape(t.shape[0], -1, self.U)[:, :self.R, :]
        return t

    def set_se_mode(self, mode: str):
        """``"stream"`` (default): every TTI streams the env's whole U x R tile.  ``"gather"``: the per-UE mean over all
        RBs -- all that the observation, PF / MT and MAPF read of a tile (agents/ib_sched.py:110-116,146-157,
        agents/common.py:567-573,648-654), a function of the tile alone -- is computed once per pooled tile, and a TTI reads
        that row plus, from a UE-major copy of the pool, only the RBs each UE was allocated.  Bit-identical results; the two
        sidecars (n_tiles * U * (8 + 4 * roundup(R, 8)) bytes) are built here for the pool bound now."""
        if mode not in ("stream", "gather"):
            raise RanEnvError("SE mode must be 'stream' or 'gather'")
        with torch.cuda.device(self.device):
            self._check(self._lib.ranenv_set_se_mode(self._h, SE_GATHER if mode == "gather" else SE_STREAM, self._stream()),
                        "ranenv_set_se_mode")
        self.se_mode = mode

    def bind_se_gather_from_power(self, power: torch.Tensor, transmission_power: float = 100.0,
                                  thermal_noise_power: float = 10e-14):
        """Gather-only channel ingest (channels/quadriga.py:56-76): QuaDRiGa received power, float64 [n_tiles, R, U] on this GPU
        (the .mat's own RB-major order), straight into the gather mode's sidecars -- no RB-major float32 pool is built or kept.
        Same sidecars bit for bit as ``bind_se_pool(quadriga_pool_from_power(power, R))`` + ``set_se_mode("gather")``."""
        if power.dtype != torch.float64 or power.device != self.device or power.dim() != 3 or tuple(power.shape[1:]) != (self.R, self.U):
            raise RanEnvError(f"power must be a float64 [n_tiles, R={self.R}, U={self.U}] tensor on {self.device}")
        power = power.contiguous()
        with torch.cuda.device(self.device):
            self._check(self._lib.ranenv_bind_se_gather_from_power(self._h, _ptr(power), power.shape[0],
                                                                  float(transmission_power) / float(self.R),
                                                                  float(thermal_noise_power), self._stream()),
                        "ranenv_bind_se_gather_from_power")
        self._keep.pop("se_pool", None)
        self._n_se_tiles = int(power.shape[0])
        self.se_mode = "gather"

    def se_sidecars(self) -> Dict[str, torch.Tensor]:
        """Diagnostic: the gather mode's sidecars, zero copy: ``row_mean`` float64 [tiles, U], ``ue_major`` float32
        [tiles, U, roundup(R, 8)]."""
        mean, um, rp = C.c_void_p(), C.c_void_p(), C.c_int32()
        self._check(self._lib.ranenv_get_se_sidecars(self._h, C.byref(mean), C.byref(um), C.byref(rp)), "ranenv_get_se_sidecars")
        n = self._keep["se_pool"].shape[0] if "se_pool" in self._keep else self._n_se_tiles
        return {"row_mean": torch.as_tensor(_DevArray(mean.value, (n, self.U), "f8", self), device=self.device),
                "ue_major": torch.as_tensor(_DevArray(um.value, (n, self.U, rp.value), "f4", self), device=self.device)}

    def bind_traffic_pool(self, traffic_pool: torch.Tensor):
        """int32 [rows, U] offered bits per UE and TTI (traffics/mult_slice.py:26-32)."""
        if traffic_pool.dtype != torch.int32 or traffic_pool.device != self.device or not traffic_pool.is_contiguous():
            raise RanEnvError("traffic pool must be a contiguous int32 tensor on the env's GPU")
        if traffic_pool.dim() != 2 or traffic_pool.shape[1] != self.U:
            raise RanEnvError(f"traffic pool must be [rows, U={self.U}]")
        self._keep["trf_pool"] = traffic_pool
        self._check(self._lib.ranenv_bind_traffic_pool(self._h, _ptr(traffic_pool), traffic_pool.shape[0]),
                    "ranenv_bind_traffic_pool")

    def set_episodes(self, scenario, se_base=0, se_len=1, se_offset=0, trf_base=0, trf_len=1, trf_offset=0):
        """Which scenario / channel trace / traffic trace each env replays (arrays of [B] or scalars)."""
        eps = self._episode_array(self.B, scenario, se_base, se_len, se_offset, trf_base, trf_len, trf_offset)
        with torch.cuda.device(self.device):
            self._check(self._lib.ranenv_set_episodes(self._h, C.c_void_p(eps.ctypes.data), self._stream()),
                        "ranenv_set_episodes")
        self.episodes = eps

    def set_policy(self, policy: int = POLICY_EXTERNAL, fixed_intra: int = INTRA_PER_SLICE):
        self._check(self._lib.ranenv_set_policy(self._h, int(policy), int(fixed_intra)), "ranenv_set_policy")
        self.policy, self.fixed_intra = int(policy), int(fixed_intra)

    _EP_DTYPE = [("scenario", "<i4"), ("se_len", "<i4"), ("se_base", "<i8"), ("se_offset", "<i4"), ("trf_len", "<i4"),
                 ("trf_base", "<i8"), ("trf_offset", "<i4"), ("reserved", "<i4")]

    def _episode_array(self, n, scenario, se_base, se_len, se_offset, trf_base, trf_len, trf_offset):
        eps = np.zeros(n, dtype=self._EP_DTYPE)
        assert eps.dtype.itemsize == C.sizeof(_lib.Episode)
        for k, v in (("scenario", scenario), ("se_len", se_len), ("se_base", se_base), ("se_offset", se_offset),
                     ("trf_len", trf_len), ("trf_base", trf_base), ("trf_offset", trf_offset)):
            eps[k] = np.broadcast_to(np.asarray(v), (n,))
        return eps

    def episode_descriptors(self) -> np.ndarray:
        """The per-env episode descriptors as they are on the device now (structured array; one small D2H)."""
        if not self._autoreset:
            return self.episodes
        raw = self.views()["episodes"].cpu().numpy()
        return np.ascontiguousarray(raw).view(self._EP_DTYPE).reshape(self.B)

    def set_traffic_generator(self, seed: int, env_id_base: int = 0, enable: bool = True):
        """Draw the offered traffic on the device -- Poisson(slice Mbps) * 1e6 bits per UE and TTI
        (traffics/mult_slice.py:24-32), Philox-4x32-10 keyed (seed; env_id_base + env, episode, step, UE) --
        instead of replaying the traffic pool."""
        with torch.cuda.device(self.device):
            self._check(self._lib.ranenv_set_traffic_generator(self._h, 1 if enable else 0, int(seed) & (2 ** 64 - 1),
                                                               int(env_id_base), self._stream()), "ranenv_set_traffic_generator")
        self.traffic_seed, self.env_id_base = (int(seed) & (2 ** 64 - 1), int(env_id_base)) if enable else (None, 0)

    def poisson_tables(self):
        """Diagnostic: the traffic generator's inversion tables -> (cdf uint64 [NS, S, 256], guide uint8 [NS, S, 64])."""
        cdf = np.zeros((self.n_scenarios, self.S, 256), dtype=np.uint64)
        guide = np.zeros((self.n_scenarios, self.S, 64), dtype=np.uint8)
        with torch.cuda.device(self.device):
            self._check(self._lib.ranenv_get_poisson_tables(self._h, C.c_void_p(cdf.ctypes.data), C.c_void_p(guide.ctypes.data)),
                        "ranenv_get_poisson_tables")
        return cdf, guide

    def set_max_steps(self, max_steps=None):
        """Per-env episode length ([B] ints) or None for the constructor's max_steps everywhere."""
        if max_steps is None:
            self._check(self._lib.ranenv_set_max_steps(self._h, None, self._stream()), "ranenv_set_max_steps")
            self.max_steps_env = None
            return
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(max_steps, dtype=np.int32), (self.B,)))
        with torch.cuda.device(self.device):
            self._check(self._lib.ranenv_set_max_steps(self._h, C.c_void_p(a.ctypes.data), self._stream()), "ranenv_set_max_steps")
        self.max_steps_env = a.copy()

    def set_episode_table(self, scenario, se_base=0, se_len=1, se_offset=0, trf_base=0, trf_len=1, trf_offset=0,
                          first_episode: int = 0):
        """Descriptor of every episode number in [first_episode, first_episode + len(scenario)): what the plugins'
        choose_episode resolves per episode (associations/mult_slice.py:444-452, channels/quadriga.py:78-87)."""
        n = len(np.atleast_1d(scenario))
        tab = self._episode_array(n, scenario, se_base, se_len, se_offset, trf_base, trf_len, trf_offset)
        with torch.cuda.device(self.device):
            self._check(self._lib.ranenv_set_episode_table(self._h, C.c_void_p(tab.ctypes.data), int(first_episode), n,
                                                           self._stream()), "ranenv_set_episode_table")
        self.episode_table, self.episode_table_first = tab, int(first_episode)

    def enable_autoreset(self, initial_episode: int, max_episode: int, random_episodes: bool = False, seed: int = 0,
                         episode_numbers=None, shortcut: bool = True):
        """After every step, envs that reported ``done`` move to their next episode on the device (sequential, or
        random in [initial, max) like enable_random_episodes, simu.py:361,377) and are reset, without a host sync.
        ``episode_numbers`` [B]: the episode every env plays now (its descriptor is installed here).  The terminal
        observation of those envs is kept in ``term_obs_inter`` / ``term_obs_intra`` (/ ``term_head_obs``).
        ``shortcut`` (library option ``autoreset_shortcut``): this class calls ranenv_autoreset right behind the step, inside
        ``step()`` / ``step_async()``, with its own ``done`` buffer -- nobody can have touched the flags in between -- so the
        library may follow the step counters on the host and enqueue nothing at a TTI at which no episode ended.  Pass
        ``False`` if you write to ``views()["step_number"]`` (the views are cached here: the library cannot see later writes)."""
        ep = None
        if episode_numbers is not None:
            ep = np.ascontiguousarray(np.broadcast_to(np.asarray(episode_numbers, dtype=np.int32), (self.B,)))
            rows = ep - self.episode_table_first
            if rows.min() < 0 or rows.max() >= len(self.episode_table):
                raise RanEnvError("episode_numbers outside the episode table")
            t = self.episode_table[rows]
            self.set_episodes(scenario=t["scenario"], se_base=t["se_base"], se_len=t["se_len"], se_offset=t["se_offset"],
                              trf_base=t["trf_base"], trf_len=t["trf_len"], trf_offset=t["trf_offset"])
        with torch.cuda.device(self.device):
            self._check(self._lib.ranenv_set_autoreset(self._h, 1, int(initial_episode), int(max_episode),
                                                       1 if random_episodes else 0, int(seed) & (2 ** 64 - 1),
                                                       None if ep is None else C.c_void_p(ep.ctypes.data), self._stream()),
                        "ranenv_set_autoreset")
        self.set_option("autoreset_shortcut", 1 if shortcut else 0)
        self.term_obs_inter = torch.zeros_like(self.obs_inter)
        self.term_obs_intra = torch.zeros_like(self.obs_intra)
        if getattr(self, "head_obs", None) is not None:
            self.term_head_obs = torch.zeros_like(self.head_obs)
        self._autoreset = True

    def disable_autoreset(self):
        self._check(self._lib.ranenv_set_autoreset(self._h, 0, 0, 0, 0, 0, None, self._stream()), "ranenv_set_autoreset")
        self._autoreset = False

    def _after_step(self, se, ic):
        if self._recorder is not None:
            self._recorder.on_step(se, ic, self.done)
        if self._autoreset:
            st = self._lib.ranenv_autoreset(self._h, _ptr(self.done), _ptr(self.obs_inter), _ptr(self.obs_intra),
                                            _ptr(self.term_obs_inter), _ptr(self.term_obs_intra), _ptr(self.term_head_obs),
                                            self._stream())
            if st != 0:
                self._check(st, "ranenv_autoreset")

    def record(self, envs, root_path: str = ".", simu_name: str = "mult_slice", agent_name: str = "agent",
               episode_numbers=None, marl: bool = True):
        """Keep the traces of the listed envs on the device and write ``hist/{simu_name}/{agent_name}/ep_{n}.npz``
        (the 16 keys of results/gen_results.py:88-108) whenever one of them reports ``done``.  ``record(None)``
        stops recording.  Returns the recorder (``.written`` lists the files)."""
        if envs is None:
            self._recorder = None
            return None
        from .history import HistoryRecorder
        if self.tables is None or self.episodes is None:
            raise RanEnvError("record() needs load_scenarios + set_episodes first")
        self._recorder = HistoryRecorder(self, envs, root_path, simu_name, agent_name, episode_numbers, marl)
        return self._recorder

    # ------------------------------------------------------------------------------------------
    def _obs(self):
        return self._obs_dict

    def reset(self, env_mask=None, se_tiles=None):
        """CommunicationEnv.reset for the masked envs (all when None); returns the formatted obs."""
        m = self._dev(env_mask, torch.uint8, (self.B,), "env_mask")
        se = self._dev(se_tiles, torch.float32, (self.B, self.R, self.U), "se_tiles")
        with torch.cuda.device(self.device):
            self._check(self._lib.ranenv_reset(self._h, _ptr(m), _ptr(se), _ptr(self.obs_inter),
                                                _ptr(self.obs_intra), _ptr(self.reward), self._stream()),
                        "ranenv_reset")
        self._keep["last_inputs"] = (m, se)
        if self._recorder is not None:
            self._recorder.on_reset(m)
        return self._obs()

    def step(self, inter_scores=None, intra_choice=None, traffic_bits=None, se_tiles=None):
        """One TTI for all envs.  Returns (obs, reward [B,S+1] float64, done [B] uint8)."""
        if inter_scores is None and intra_choice is None and traffic_bits is None and se_tiles is None:
            # device policy + pools: nothing to marshal, just enqueue the launch
            st = self._step_fn(self._h, None, None, None, None, *self._p_out,
                               C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
            if st != 0:
                self._check(st, "ranenv_step")
            if self._recorder is not None or self._autoreset:
                self._after_step(None, None)
            return self._obs(), self.reward, self.done
        sc = self._dev(inter_scores, torch.float64, (self.B, self.S), "inter_scores")
        ic = self._dev(intra_choice, torch.uint8, (self.B, self.S), "intra_choice")
        tr = self._dev(traffic_bits, torch.float64, (self.B, self.U), "traffic_bits")
        se = self._dev(se_tiles, torch.float32, (self.B, self.R, self.U), "se_tiles")
        with torch.cuda.device(self.device):
            self._check(self._lib.ranenv_step(self._h, _ptr(sc), _ptr(ic), _ptr(tr), _ptr(se), _ptr(self.obs_inter),
                                               _ptr(self.obs_intra), _ptr(self.reward), _ptr(self.done),
                                               self._stream()), "ranenv_step")
        self._keep["last_inputs"] = (sc, ic, tr, se)
        if self._recorder is not None or self._autoreset:
            self._after_step(se, ic)
        return self._obs(), self.reward, self.done

    # -- a learner in the loop: ranges of the batch stepped alternately on their own streams -------------------------
    def set_ranges(self, n_ranges: int = 2):
        """Cut the batch into ``n_ranges`` contiguous ranges (the batch partitions of ``set_partitions``), each with its
        own HIP stream, for ``step_async`` / ``step_wait``: while the policy consumes one range's observations, the other
        ranges' TTIs occupy the GPU (the reference trains through env.step with 10 concurrent env runners,
        simu.py:555-566, agents/ray_agent.py:296-300).  Returns the ranges [(lo, hi), ...]."""
        self.set_partitions(n_ranges)
        lo, n = C.c_int32(), C.c_int32()
        self._ranges = []
        for k in range(n_ranges):
            self._check(self._lib.ranenv_get_partition(self._h, k, C.byref(lo), C.byref(n)), "ranenv_get_partition")
            self._ranges.append((lo.value, lo.value + n.value))
        self._range_out = [({"obs_inter": self.obs_inter[lo:hi], "obs_intra": self.obs_intra[lo:hi]}, self.reward[lo:hi],
                            self.done[lo:hi]) for lo, hi in self._ranges]
        self._range_streams = {}
        return list(self._ranges)

    def range_stream(self, k: int) -> "torch.cuda.Stream":
        """Range ``k``'s own HIP stream as a torch stream.  A learner that runs range k's policy inside
        ``with torch.cuda.stream(env.range_stream(k)):`` and calls ``step_wait(k)`` / ``step_async(k, ...)`` there makes
        range k one in-order chain TTI -> policy -> TTI on one hardware queue: no event crosses between queues (each such
        hop costs ~15 us on this GPU), and the ranges' chains overlap on the GPU like concurrent env runners."""
        if self._ranges is None:
            raise RanEnvError("range_stream needs set_ranges() first")
        if k not in self._range_streams:
            p = C.c_void_p()
            self._check(self._lib.ranenv_get_part_stream(self._h, int(k), C.byref(p)), "ranenv_get_part_stream")
            self._range_streams[k] = torch.cuda.ExternalStream(p.value, device=self.device)
        return self._range_streams[k]

    def step_async(self, k: int, inter_scores=None, intra_choice=None, traffic_bits=None, se_tiles=None):
        """Enqueue one TTI of range ``k`` on that range's stream, ordered behind what the caller's current stream holds
        now (the kernels that produced the scores).  The arguments are whole-batch tensors ([B, ...], already on the
        device: nothing is converted here); only range k's rows are read and written, and the caller must leave those
        rows alone until ``step_wait(k)``.  Returns at once (one library call: ranenv_step_part; with ``enable_autoreset``
        a second one, ranenv_autoreset_part: finished envs of the range restart behind the step, ``done`` / ``reward`` keep
        the terminal transition, ``term_obs_*`` the terminal observation)."""
        if self._ranges is None:
            raise RanEnvError("step_async needs set_ranges() first")
        if self._recorder is not None:
            raise RanEnvError("step_async does not run the history recorder: use step()")
        for name, x, dt in (("inter_scores", inter_scores, torch.float64), ("intra_choice", intra_choice, torch.uint8),
                            ("traffic_bits", traffic_bits, torch.float64), ("se_tiles", se_tiles, torch.float32)):
            if x is not None and not (x.dtype == dt and x.device == self.device and x.shape[0] == self.B and x.is_contiguous()):
                raise RanEnvError(f"step_async: {name} must be a contiguous {dt} tensor [B, ...] on {self.device}")
        cur = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        st = self._lib.ranenv_step_part(self._h, k, _ptr(inter_scores), _ptr(intra_choice), _ptr(traffic_bits), _ptr(se_tiles),
                                        *self._p_out, cur)
        if st != 0:
            self._check(st, "ranenv_step_part")
        if self._autoreset:       # the range's finished envs move on to their next episode behind the step, on the range's stream
            st = self._lib.ranenv_autoreset_part(self._h, k, _ptr(self.done), _ptr(self.obs_inter), _ptr(self.obs_intra),
                                                 _ptr(self.term_obs_inter), _ptr(self.term_obs_intra), _ptr(self.term_head_obs), cur)
            if st != 0:
                self._check(st, "ranenv_autoreset_part")
        # the inputs are read on the range's stream: they stay referenced here until the range's next launch (their
        # memory must not go back to the caching allocator meanwhile)
        self._keep[("async_inputs", k)] = (inter_scores, intra_choice, traffic_bits, se_tiles)

    def step_wait(self, k: int):
        """Order the caller's current stream behind range ``k``'s last ``step_async`` (no host sync) and return views of
        that range's rows: ({"obs_inter", "obs_intra"}, reward, done).  They are zero-copy views of buffers the range's NEXT
        step_async overwrites in place from a HIP kernel (autograd's version counters do not see it): clone whatever a
        graph or a replay buffer keeps beyond that call."""
        st = self._lib.ranenv_wait_part(self._h, k, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if st != 0:
            self._check(st, "ranenv_wait_part")
        return self._range_out[k]

    def step_dense(self, sched_decision, traffic_bits=None, se_tiles=None):
        """One TTI with a caller-made dense sched_decision [B,U,R] (any agent's action_format)."""
        sd = self._dev(sched_decision, torch.uint8, (self.B, self.U, self.R), "sched_decision")
        tr = self._dev(traffic_bits, torch.float64, (self.B, self.U), "traffic_bits")
        se = self._dev(se_tiles, torch.float32, (self.B, self.R, self.U), "se_tiles")
        with torch.cuda.device(self.device):
            self._check(self._lib.ranenv_step_dense(self._h, _ptr(sd), _ptr(tr), _ptr(se), _ptr(self.obs_inter),
                                                     _ptr(self.obs_intra), _ptr(self.reward), _ptr(self.done),
                                                     self._stream()), "ranenv_step_dense")
        self._keep["last_inputs"] = (sd, tr, se)
        return self._obs(), self.reward, self.done

    # ------------------------------------------------------------------------------------------
    def enable_heads(self, slice_usecase=None):
        """Also compute the SchedTWC / SchedColORAN observation and rewards every TTI
        (agents/sched_twc.py:165-413, agents/sched_colran.py:348-419).

        ``self.head_obs``  float32 [B, 10*S]: per slice 3 requirement values, then the slice means of SE,
        served Mbps, effective Mbps, buffer occupancy, buffer latency, loss rate and the requested Mbps
        (metric-major, slices in index order); ``self.head_reward`` float64 [B, 2] = (SchedTWC, SchedColORAN).
        ``slice_usecase``: int [n_scenarios, S], bit 0 eMBB / bit 1 URLLC (scenario.slice_usecase_from_req).
        Their action is IBSched's with round-robin inside the slices: ``set_policy(POLICY_EXTERNAL, INTRA_RR)``.
        """
        self.head_obs = torch.zeros((self.B, 10 * self.S), dtype=torch.float32, device=self.device)
        self.head_reward = torch.zeros((self.B, 2), dtype=torch.float64, device=self.device)
        self._check(self._lib.ranenv_bind_head_outputs(self._h, _ptr(self.head_obs), _ptr(self.head_reward)),
                    "ranenv_bind_head_outputs")
        if self._autoreset:
            self.term_head_obs = torch.zeros_like(self.head_obs)
        if slice_usecase is not None:
            self.set_slice_usecase(slice_usecase)

    def set_slice_usecase(self, slice_usecase, first: int = 0):
        uc = np.ascontiguousarray(slice_usecase, dtype=np.int32).reshape(-1, self.S)
        with torch.cuda.device(self.device):
            self._check(self._lib.ranenv_set_slice_usecase(self._h, int(first), uc.shape[0], C.c_void_p(uc.ctypes.data),
                                                           self._stream()), "ranenv_set_slice_usecase")

    # ------------------------------------------------------------------------------------------
    def views(self) -> Dict[str, torch.Tensor]:
        """Zero-copy torch views of the handle's raw-observation and state arrays."""
        if self._views is None:
            v = _lib.Views()
            self._check(self._lib.ranenv_get_views(self._h, C.byref(v)), "ranenv_get_views")
            dims = {"B": self.B, "U": self.U, "S": self.S, "K": self.Us, "E": C.sizeof(_lib.Episode) // 4}
            out = {}
            for name, ts, shp in _lib.VIEW_FIELDS:
                shape = tuple(dims[c] for c in shp)
                arr = _DevArray(getattr(v, name), shape, ts, self)
                out[name] = torch.as_tensor(arr, device=self.device)
                assert out[name].dtype == _TORCH_DT[ts]
            self._views = out
        return self._views

    def raw_observation(self) -> Dict[str, torch.Tensor]:
        """The reference's raw-observation metric fields (agents/ib_sched.py:78-181), batched."""
        v = self.views()
        if self.tables is None or self.episodes is None:
            raise RanEnvError("raw_observation needs load_scenarios + set_episodes")
        scen = v["episodes"][:, 0].to(torch.int64)           # as on the device: auto-reset may have moved on
        max_pkts = torch.as_tensor(self.tables.ue_max_pkts, device=self.device)[scen].to(torch.float64)
        q = v["queue_pkts"].to(torch.float64)
        lat = torch.where(q > 0, v["queue_age_sum"].to(torch.float64) / q.clamp(min=1), torch.zeros_like(q))
        return {
            "pkt_incoming": v["pkt_incoming"].to(torch.float64),
            "pkt_throughputs": v["pkt_throughputs"].to(torch.float64),
            "pkt_effective_thr": v["pkt_effective_thr"].to(torch.float64),
            "dropped_pkts": v["dropped_pkts"].to(torch.float64),
            "buffer_occupancies": q / max_pkts,
            "buffer_latencies": lat,
        }

    def profile_begin(self):
        """Time every launch of the step kernel from here on (the dispatch's own start / stop timestamps)."""
        self._check(self._lib.ranenv_profile_begin(self._h), "ranenv_profile_begin")

    def profile_end(self) -> Dict[str, float]:
        """Average duration in ms of the step-kernel launches since profile_begin: {'step', 'n_launches', 'n_ttis'}
        (inside rollout() a launch may cover several TTIs: n_ttis / n_launches of them on average; n_env_ttis = envs x TTIs
        summed over the launches)."""
        ms, n, nt, ne = C.c_double(), C.c_int32(), C.c_int64(), C.c_int64()
        self._check(self._lib.ranenv_profile_end(self._h, C.byref(ms), C.byref(n)), "ranenv_profile_end")
        self._check(self._lib.ranenv_profile_work(self._h, C.byref(nt), C.byref(ne)), "ranenv_profile_work")
        return {"step": ms.value, "n_launches": n.value, "n_ttis": nt.value, "n_env_ttis": ne.value}

    def set_partitions(self, n_parts: int):
        """Step the batch as ``n_parts`` contiguous ranges of envs, each by its own launch on its own stream
        (ranenv_set_partitions): one range's launch ramp and tail then run under the others' steady state."""
        with torch.cuda.device(self.device):
            self._check(self._lib.ranenv_set_partitions(self._h, int(n_parts)), "ranenv_set_partitions")
        self.n_parts = int(n_parts)

    def rollout(self, n_steps: int):
        """``n_steps`` TTIs under the device policy enqueued in one call (MARR / MAPF evaluation runs): the launches of
        ``n_steps`` calls of ``step()``, joined with the current stream only before the first and after the last TTI
        -- except that one launch takes its envs through up to n_steps / 4 (at most 10) TTIs where nothing has to happen in
        between (no head kernel; with auto-reset: up to the TTI at which an episode of the batch ends).  Same results bit
        for bit.  Returns the last TTI's (obs, reward, done)."""
        if self._recorder is not None:
            raise RanEnvError("rollout() does not return between TTIs: the recorder needs step()")
        st = self._lib.ranenv_rollout(self._h, int(n_steps), *self._p_out,
                                      C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if st != 0:
            self._check(st, "ranenv_rollout")
        return self._obs(), self.reward, self.done

    METRIC_NAMES = ("ttis", "reward", "violations", "priority_violations", "distance", "priority_distance",
                    "pkts_sent", "pkts_dropped")

    def enable_metrics(self, episode_slots: int = 0) -> None:
        """Per-env running sums over the TTIs of the current episode, kept by the step kernel (include/ranenv.h:
        TTIs, inter-slice reward, slices in violation and distance to fulfilment for all / priority slices -- the
        quantities results/gen_results.py:874-1022 derives from the history files -- and packets sent / dropped).
        With auto-reset a finished episode's sums are appended to the env's log of ``episode_slots`` rows."""
        self._check(self._lib.ranenv_enable_metrics(self._h, int(episode_slots), self._stream()), "ranenv_enable_metrics")
        self._metric_views = None

    def disable_metrics(self) -> None:
        self._check(self._lib.ranenv_enable_metrics(self._h, -1, self._stream()), "ranenv_enable_metrics")

    def episode_metrics(self) -> Dict[str, torch.Tensor]:
        """Zero-copy views: ``running`` [B, 8] (current episode), ``episode_log`` [B, slots, 8] (finished episodes, in
        order; absent with 0 slots), ``episodes_done`` [B]; columns as METRIC_NAMES."""
        if getattr(self, "_metric_views", None) is None:
            run, log, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
            slots = C.c_int32()
            self._check(self._lib.ranenv_get_metrics(self._h, C.byref(run), C.byref(log), C.byref(n), C.byref(slots)), "ranenv_get_metrics")
            out = {"running": torch.as_tensor(_DevArray(run.value, (self.B, 8), "f8", self), device=self.device),
                   "episodes_done": torch.as_tensor(_DevArray(n.value, (self.B,), "i4", self), device=self.device)}
            if slots.value > 0:
                out["episode_log"] = torch.as_tensor(_DevArray(log.value, (self.B, slots.value, 8), "f8", self), device=self.device)
            self._metric_views = out
        return self._metric_views

    def evaluate(self, n_episodes: int, max_steps=None) -> Dict[str, np.ndarray]:
        """The reference's test loop for its baseline agents (simu.py:547-566 over ``max_episode - initial_episode``
        episodes; the numbers results/gen_results.py:874-1022 turns into the paper's violation / distance figures), for
        the whole batch on the device: reset, then one rollout long enough for every env to finish ``n_episodes``
        episodes under the device policy, episode ends handled by auto-reset.  Needs enable_autoreset(...) and
        enable_metrics(slots >= n_episodes).  Returns {metric: float64 [B, n_episodes]} with the names of METRIC_NAMES;
        row b holds env b's episodes in the order it played them (from the episode number given to enable_autoreset)."""
        if not self._autoreset:
            raise RanEnvError("evaluate() needs enable_autoreset(): it runs through episode ends on the device")
        m = self.episode_metrics()
        if "episode_log" not in m or m["episode_log"].shape[1] < n_episodes:
            raise RanEnvError(f"evaluate({n_episodes}) needs enable_metrics(episode_slots >= {n_episodes})")
        if max_steps is not None:
            self.set_max_steps(max_steps)
        me = getattr(self, "max_steps_env", None)
        longest = int(self.max_steps) if me is None else int(me.max())
        self.enable_metrics(m["episode_log"].shape[1])      # zero the sums and the log
        self.reset()
        self.rollout(n_episodes * longest)
        torch.cuda.synchronize(self.device)
        done = m["episodes_done"].cpu().numpy()
        if done.min() < n_episodes:
            raise RanEnvError(f"an env finished only {int(done.min())} of {n_episodes} episodes: per-env max_steps longer than assumed")
        log = m["episode_log"][:, :n_episodes].cpu().numpy()
        return {name: log[:, :, k].copy() for k, name in enumerate(self.METRIC_NAMES)}

    def set_option(self, key: str, value: int) -> None:
        """A tuning / debug knob of the launch schedule (include/ranenv.h, "Options": compact, fuse, fuse_first0..9, late,
        row_width, small_batch, persist).  None of them changes a result."""
        self._check(self._lib.ranenv_set_option(self._h, key.encode(), int(value)), f"ranenv_set_option({key})")

    def get_option(self, key: str) -> int:
        v = C.c_int64()
        self._check(self._lib.ranenv_get_option(self._h, key.encode(), C.byref(v)), f"ranenv_get_option({key})")
        return int(v.value)

    def launch_info(self):
        g, b, l = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._lib.ranenv_launch_info(self._h, C.byref(g), C.byref(b), C.byref(l)), "ranenv_launch_info")
        return {"grid": g.value, "block": b.value, "lds_bytes": l.value}

    def algorithmic_bytes_per_env_step(self, se_mode: Optional[str] = None) -> int:
        """4*U*R + 180*U + S*(85 + 8*Us) + 4 for the streaming step.  The gather mode replaces the
        tile term 4*U*R by what it reads of a tile: the R allocated elements (every RB belongs to one UE) and the per-UE
        mean row of the sidecar, 4*R + 8*U."""
        rest = 180 * self.U + self.S * (85 + 8 * self.Us) + 4
        if (se_mode or self.se_mode) == "gather":
            return 4 * self.R + 8 * self.U + rest
        return 4 * self.U * self.R + rest
