"""Plugin surfaces of the env core, mirrored from the reference's call sites.

The reference's plugins subclass four base classes of ``sixg_radio_mgmt`` (an un-vendored git
submodule).  Their constructor orders and ``step`` signatures are pinned by the subclasses in the
reference tree:

    Association(ues, max_number_ues, max_number_basestations, max_number_slices, rng, root_path)
        associations/mult_slice.py:21-28        step(...)  :350-358
    Traffic(max_number_ues, rng, root_path)     traffics/mult_slice.py:13   step(...) :15-21
    Channel(max_number_ues, max_number_basestations, num_available_rbs, rng, root_path, scenario_name)
        channels/quadriga.py:19-26              step(...)  :38-44
    Mobility(max_number_ues, rng, root_path)    mobilities/simple.py:13     step(...) :15
    Agent(env, max_number_ues, max_number_slices, max_number_basestations, num_available_rbs, seed)
        agents/ib_sched.py:39-45

so a plugin written against the reference attaches to ``MARLCommEnv`` (comm_env.py) unchanged.
The concrete plugins below restate the reference's own ones; they are host-side numpy, called
once per TTI by the B=1 facade only (the batched path replays pools from HBM instead).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from .scenario import SLICE_TEMPLATES, generate_reference_scenario, slice_template_dict


# --------------------------------------------------------------------------------------------
# base classes (attribute holders, like the reference's)
# --------------------------------------------------------------------------------------------
class _Buffer:
    """What the reference reads off ``ues.buffers[i]`` (gen_assoc_mult_slice.py:218-224)."""

    def __init__(self, max_packets_buffer: int, max_packets_age: int):
        self.max_packets_buffer = int(max_packets_buffer)
        self.max_packets_age = int(max_packets_age)


class UEs:
    """Per-UE buffer parameters; the queues themselves live on the GPU.

    Constructor order from gen_assoc_mult_slice.py:92-97; ``update_ues`` argument order from
    associations/mult_slice.py:483-488; attributes ``max_buffer_pkts`` / ``pkt_sizes`` read by
    agents/common.py:581-582,591.
    """

    def __init__(self, max_number_ues, max_buffer_latencies, max_buffer_pkts, pkt_sizes):
        self.max_number_ues = int(max_number_ues)
        self.max_buffer_latencies = np.array(max_buffer_latencies, dtype=np.int64).copy()
        self.max_buffer_pkts = np.array(max_buffer_pkts, dtype=np.int64).copy()
        self.pkt_sizes = np.array(pkt_sizes, dtype=np.int64).copy()
        self.version = 0   # bumped on every update: the env reloads the device tables
        self._rebuild()

    def _rebuild(self):
        self.buffers = [_Buffer(self.max_buffer_pkts[i], self.max_buffer_latencies[i])
                        for i in range(self.max_number_ues)]

    def update_ues(self, ue_indexes, max_buffer_latencies, max_buffer_pkts, pkt_sizes):
        self.max_buffer_latencies[ue_indexes] = max_buffer_latencies
        self.max_buffer_pkts[ue_indexes] = max_buffer_pkts
        self.pkt_sizes[ue_indexes] = pkt_sizes
        self.version += 1
        self._rebuild()


class Agent:
    def __init__(self, env, max_number_ues, max_number_slices, max_number_basestations, num_available_rbs, seed=0):
        self.env = env
        self.max_number_ues = max_number_ues
        self.max_number_slices = max_number_slices
        self.max_number_basestations = max_number_basestations
        self.num_available_rbs = num_available_rbs
        self.seed = seed


class Association:
    def __init__(self, ues, max_number_ues, max_number_basestations, max_number_slices,
                 rng=None, root_path: str = ""):
        self.ues = ues
        self.max_number_ues = max_number_ues
        self.max_number_basestations = max_number_basestations
        self.max_number_slices = max_number_slices
        self.rng = rng if rng is not None else np.random.default_rng()
        self.root_path = root_path

    def step(self, basestation_ue_assoc, basestation_slice_assoc, slice_ue_assoc, slice_req,
             step_number, episode_number):
        raise NotImplementedError


class Traffic:
    def __init__(self, max_number_ues, rng=None, root_path: str = ""):
        self.max_number_ues = max_number_ues
        self.rng = rng if rng is not None else np.random.default_rng()
        self.root_path = root_path

    def step(self, slice_ue_assoc, slice_req, step_number, episode_number):
        raise NotImplementedError


class Channel:
    def __init__(self, max_number_ues, max_number_basestations, num_available_rbs, rng=None,
                 root_path: str = "", scenario_name: str = ""):
        self.max_number_ues = max_number_ues
        self.max_number_basestations = max_number_basestations
        self.num_available_rbs = num_available_rbs
        self.rng = rng if rng is not None else np.random.default_rng()
        self.root_path = root_path
        self.scenario_name = scenario_name

    def step(self, step_number, episode_number, mobilities, sched_decision=None):
        raise NotImplementedError


class Mobility:
    def __init__(self, max_number_ues, rng=None, root_path: str = ""):
        self.max_number_ues = max_number_ues
        self.rng = rng if rng is not None else np.random.default_rng()
        self.root_path = root_path

    def step(self, step_number, episode_number):
        raise NotImplementedError


# --------------------------------------------------------------------------------------------
# the reference's own plugins, restated
# --------------------------------------------------------------------------------------------
class SimpleMobility(Mobility):
    """mobilities/simple.py:15-16"""

    def step(self, step_number, episode_number):
        return np.ones((self.max_number_ues, 2))


class SimpleTraffic(Traffic):
    """traffics/simple.py:15-22: constant 4 bits per UE."""

    def step(self, slice_ue_assoc, slice_req, step_number, episode_number):
        return np.ones(self.max_number_ues) * 4


class MultSliceTraffic(Traffic):
    """traffics/mult_slice.py:15-34: Poisson(slice Mbps) * 1e6 bits for the UEs of each slice,
    slices in dict order, one generator call per slice."""

    def step(self, slice_ue_assoc, slice_req, step_number, episode_number):
        traffic_per_ue = np.zeros(self.max_number_ues)
        for name, req in slice_req.items():
            if req != {}:
                s = int(name.split("_")[1])
                idx_ues = (slice_ue_assoc[s, :] == 1).nonzero()[0]
                traffic_per_ue[idx_ues] = self.rng.poisson(req["ues"]["traffic"], len(idx_ues)) * 1e6
        return traffic_per_ue


class FixedSE(Channel):
    """channels/fixed_se.py:26,35-41: SE = 2.0 everywhere."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.fixed_se = 2.0

    def step(self, step_number, episode_number, mobilities, sched_decision=None):
        return np.array([self.fixed_se * np.ones((self.max_number_ues, self.num_available_rbs[i]))
                         for i in range(self.max_number_basestations)])


class MimicQuadriga(Channel):
    """channels/mimic_quadriga.py:30-58: per-episode UE means |N(10, 7.5)|, per-TTI |N(mean, 1.5)|."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.current_episode_number = -1
        self.ues_mean_se = np.array([])
        self.default_std = 1.5

    def step(self, step_number, episode_number, mobilities, sched_decision=None):
        if episode_number != self.current_episode_number:
            self.current_episode_number = episode_number
            self.ues_mean_se = np.abs(self.rng.normal(10, 7.5, size=(self.max_number_ues,)))
        se = np.ones((self.max_number_basestations, self.max_number_ues, self.num_available_rbs[0]))
        for ue_idx, ue_mean in enumerate(self.ues_mean_se):
            se[0, ue_idx, :] = np.abs(self.rng.normal(ue_mean, self.default_std, size=(self.num_available_rbs[0],)))
        return se


class PoolChannel(Channel):
    """Replay of a precomputed SE array [episodes][steps][U][R] (what channels/quadriga.py:38-76 does
    from target_cell_power.mat after log2(1 + P*g/N); QuadrigaChannel below reads the files themselves)."""

    def __init__(self, *a, pool: Optional[np.ndarray] = None, **k):
        super().__init__(*a, **k)
        self.pool = pool

    def step(self, step_number, episode_number, mobilities, sched_decision=None):
        ep = self.pool[episode_number % self.pool.shape[0]]
        return ep[step_number % ep.shape[0]][None, :, :]


class QuadrigaChannel(Channel):
    """channels/quadriga.py:9-87: replay of QuaDRiGa channel files.

    ``{root_path}/mult_slice_channel_generation/results/mult_slice/freq_channel/assoc_{A}/ep_{E}/
    target_cell_power.mat`` is opened when the episode changes (:45-54); per TTI the step-th slice of
    ``target_cell_power`` goes through ``log2(1 + (P/R)*g / (0 + noise))`` and is transposed to
    ``(1, U, R)`` (:56-76).  (A, E) = (episode, 0) (:78-87).  The files are MATLAB v7.3 = HDF5: ``h5py`` is
    imported when the first file is opened and its absence is an error here, not a silent fallback.
    """

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.scenario_name = "mult_slice"          # the reference always uses this folder (:27-29)
        self.current_episode_number = -1
        self.file = None
        self.channels_path = f"{self.root_path}/mult_slice_channel_generation/results/{self.scenario_name}/freq_channel/"
        self.spectral_efficiencies = np.array([])
        self.transmission_power = 100              # Watts
        self.thermal_noise_power = 10e-14
        self.channel_eps_per_scenario = 100

    def _open(self, path: str):
        try:
            import h5py
        except ImportError as exc:
            raise ImportError("QuadrigaChannel reads MATLAB v7.3 (HDF5) files and needs h5py") from exc
        return h5py.File(path, "r")

    def choose_episode(self, episode_number: int, current_episode: int):
        if episode_number != current_episode:
            return episode_number, 0, True
        return 0, 0, False

    def step(self, step_number, episode_number, mobilities, sched_decision=None):
        assoc, ep, changed = self.choose_episode(episode_number, self.current_episode_number)
        if changed:
            self.current_episode_number = episode_number
            if self.file is not None:
                self.file.close()
            self.file = self._open(f"{self.channels_path}assoc_{assoc}/ep_{ep}/target_cell_power.mat")
        if self.file is None:
            raise ValueError("File is None")
        power = np.array(self.file.get("target_cell_power")[step_number, :, :, :, :])
        se = quadriga_se_from_power(power, int(self.num_available_rbs[0]), self.transmission_power,
                                    self.thermal_noise_power)
        self.spectral_efficiencies = np.squeeze(se.transpose())
        return np.array([self.spectral_efficiencies])


class QuadrigaChannelSeq(QuadrigaChannel):
    """channels/quadriga_seq.py:28-39: (A, E) = (episode // 100, episode % 100)."""

    def choose_episode(self, episode_number: int, current_episode: int):
        if episode_number != current_episode:
            return episode_number // self.channel_eps_per_scenario, episode_number % self.channel_eps_per_scenario, True
        return 0, 0, False


def quadriga_se_from_power(target_cell_power: np.ndarray, n_rbs: int, transmission_power: float = 100.0,
                           thermal_noise_power: float = 10e-14) -> np.ndarray:
    """channels/quadriga.py:56-69: log2(1 + (P / n_rbs) * g / (0 + noise))."""
    return np.log2(1 + np.divide((transmission_power / n_rbs) * target_cell_power,
                                 np.zeros_like(target_cell_power) + thermal_noise_power))


class SimpleAssociation(Association):
    """associations/simple.py:27-41: pass-through."""

    def step(self, basestation_ue_assoc, basestation_slice_assoc, slice_ue_assoc, slice_req,
             step_number, episode_number):
        return basestation_ue_assoc, basestation_slice_assoc, slice_ue_assoc, slice_req


class MultSliceAssociation(Association):
    """associations/mult_slice.py: generator mode (:359-423), replay of ``ep_N.npz`` scenario files
    (:424-442, :490-508; scenario = episode % 200, :444-452) and update_ues (:468-488).

    Replay reads ``{root_path}/associations/data/{scenario_name}/ep_{n}.npz`` (pickled dicts: trusted
    files only).  In generator mode a scenario is drawn per episode from the same law, consuming the
    shared rng in the reference's call order.
    """

    def __init__(self, ues, max_number_ues, max_number_basestations, max_number_slices, rng=None,
                 root_path: str = ".", generator_mode: bool = True, slice_req_changed: bool = True,
                 scenario_name: str = "mult_slice"):
        # positional order of associations/mult_slice.py:9-20 (the reference defaults generator_mode to
        # False; the datasets it would replay are download links, so the default here is the generator)
        super().__init__(ues, max_number_ues, max_number_basestations, max_number_slices, rng, root_path)
        self.slice_req_changed = slice_req_changed
        self.min_number_slices = 3
        self.maximum_number_scenarios = 200
        self.generator_mode = generator_mode
        self.scenario_name = scenario_name
        self.slice_types = [t[0] for t in SLICE_TEMPLATES]
        self.slice_type_model = {t[0]: slice_template_dict(i) for i, t in enumerate(SLICE_TEMPLATES)}
        self.slices_to_use = np.array([])
        self.current_episode = -1
        self._ep = None

    def choose_episode(self, episode_number: int, current_episode: int):
        episode_to_use = episode_number % self.maximum_number_scenarios
        if episode_to_use != current_episode:
            return episode_to_use, True
        return 0, False

    def load_episode_data(self, episode_number: int):
        from .scenario import load_episode_npz
        self._ep = load_episode_npz(f"{self.root_path}/associations/data/{self.scenario_name}/ep_{episode_number}.npz")
        self.current_episode = episode_number

    def step(self, basestation_ue_assoc, basestation_slice_assoc, slice_ue_assoc, slice_req,
             step_number, episode_number):
        if self.generator_mode:
            if step_number == 0:
                bua, bsa, sua, req, slices = generate_reference_scenario(
                    self.rng, self.max_number_slices, self.max_number_ues, self.min_number_slices)
                self.slices_to_use = slices
                self.update_ues(sua, slices, req)
                return bua, bsa, sua, req
            return basestation_ue_assoc, basestation_slice_assoc, slice_ue_assoc, slice_req
        episode_to_use, condition = self.choose_episode(episode_number, self.current_episode)
        if condition:
            self.load_episode_data(episode_to_use)
            self.update_ues(self._ep["hist_slice_ue_assoc"][step_number], self._ep["hist_slices_to_use"][step_number],
                            self._ep["hist_slice_req"][step_number])
        return (self._ep["hist_basestation_ue_assoc"][step_number], self._ep["hist_basestation_slice_assoc"][step_number],
                self._ep["hist_slice_ue_assoc"][step_number], self._ep["hist_slice_req"][step_number])

    def update_ues(self, slice_ue_assoc, slices_to_use, slice_req):
        for s in slices_to_use:
            ues = (slice_ue_assoc[s] == 1).nonzero()[0]
            u = slice_req[f"slice_{s}"]["ues"]
            self.ues.update_ues(ues, np.repeat(u["buffer_latency"], len(ues)),
                                np.repeat(u["buffer_size"], len(ues)), np.repeat(u["message_size"], len(ues)))


class MultSliceAssociationSeq(MultSliceAssociation):
    """associations/mult_slice_seq.py:9-46: the ``mult_slice_seq`` sweep -- 100 consecutive episodes replay
    one association scenario (``ep_{episode // 100}.npz`` of the mult_slice folder) while the channel
    changes every episode (QuadrigaChannelSeq, channels/quadriga_seq.py:28-39)."""

    def __init__(self, ues, max_number_ues, max_number_basestations, max_number_slices, rng=None,
                 root_path: str = ".", generator_mode: bool = False, slice_req_changed: bool = True,
                 scenario_name: str = "mult_slice"):
        super().__init__(ues, max_number_ues, max_number_basestations, max_number_slices, rng, root_path,
                         generator_mode, slice_req_changed, scenario_name)
        self.scenario_name = "mult_slice"          # reads the associations of the mult_slice folder (:33-35)
        self.channels_per_scenario = 100

    def choose_episode(self, episode_number: int, current_episode: int):
        episode_to_use = episode_number // self.channels_per_scenario
        if episode_to_use != current_episode:
            return episode_to_use, True
        return 0, False


def simple_slice_req() -> dict:
    """The two toy slice intents SimpleSliceAssociation installs (associations/simple_slice.py:46-105)."""
    from .scenario import OP_NAME, OP_UFUNC
    ge, le = OP_UFUNC[OP_NAME["at_least"]], OP_UFUNC[OP_NAME["at_most"]]
    ues = {"buffer_size": 10, "buffer_latency": 10, "message_size": 1, "mobility": 0, "traffic": 2,
           "min_number_ues": 8, "max_number_ues": 10}
    return {
        "slice_0": {
            "name": "robotic_surgery_case_1",
            "parameters": {
                "par1": {"name": "reliability", "value": 99.00, "unit": "rate", "operator": ge},
                "par2": {"name": "latency", "value": 20, "unit": "ms", "operator": le},
                "par3": {"name": "throughput", "value": 1, "unit": "Mbps", "operator": ge},
            },
            "ues": dict(ues),
        },
        "slice_1": {
            "name": "control_case_2",
            "parameters": {
                "par1": {"name": "reliability", "value": 1.0, "unit": "rate", "operator": ge},
                "par2": {"name": "latency", "value": 20, "unit": "ms", "operator": le},
            },
            "ues": dict(ues),
        },
    }


class SimpleSliceAssociation(Association):
    """associations/simple_slice.py:27-112: associations pass through unchanged; at step 0 the two toy
    slice intents replace ``slice_req`` (the reference never pushes their buffer parameters to the UEs:
    it has no update_ues call, so the UEs keep what the env was built with)."""

    def step(self, basestation_ue_assoc, basestation_slice_assoc, slice_ue_assoc, slice_req,
             step_number, episode_number):
        if step_number == 0:
            slice_req = simple_slice_req()
        return basestation_ue_assoc, basestation_slice_assoc, slice_ue_assoc, slice_req
