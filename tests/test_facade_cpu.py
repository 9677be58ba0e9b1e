"""Host logic of the MARLCommEnv facade with a stand-in for the device env (no GPU): a full replayed episode
(scenario file with exactly max_number_steps rows: the terminal transition must not read one more), episode
advance, history file."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from intent_radio_sched_multi_slice_amd import comm_env, plugins
from intent_radio_sched_multi_slice_amd.history import HIST_KEYS
from intent_radio_sched_multi_slice_amd.scenario import generate_reference_scenario, save_episode_npz


class _StubDevice:
    """What the facade calls on BatchedRanEnv, answering with zeros."""

    def __init__(self, batch, n_slices, n_ues, n_rbs, **kw):
        self.U = n_ues
        self.calls = []

    def set_episodes(self, **kw): self.calls.append("set_episodes")
    def load_scenarios(self, tables): self.calls.append("load_scenarios")
    def reset(self, se_tiles=None): self.calls.append("reset")
    def step_dense(self, sched, traffic, se): self.calls.append("step_dense")
    def close(self): pass

    def raw_observation(self):
        z = torch.zeros((1, self.U), dtype=torch.float64)
        return {k: z.clone() for k in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts",
                                       "buffer_occupancies", "buffer_latencies")}


class _Agent:
    def __init__(self, env): self.env = env
    def obs_space_format(self, raw): return {"player_0": {"observations": raw["buffer_occupancies"], "action_mask": np.ones(5)}}
    def calculate_reward(self, obs): return {"player_0": 0.5, "player_1": -1.0}
    def action_format(self, action):
        ce = self.env.comm_env
        return np.zeros((1, ce.max_number_ues, int(ce.num_available_rbs[0])))


def test_replayed_episode_runs_to_its_terminal_step_and_writes_history(tmp_path, monkeypatch):
    monkeypatch.setattr(comm_env, "BatchedRanEnv", _StubDevice)
    S, U, steps = 5, 25, 6
    d = tmp_path / "associations" / "data" / "mult_slice"
    d.mkdir(parents=True)
    gen = np.random.default_rng(4)
    for n in range(2):
        bua, bsa, sua, req, use = generate_reference_scenario(gen, S, U)
        save_episode_npz(str(d / f"ep_{n}.npz"), bua, bsa, sua, req, use, steps)     # exactly `steps` rows

    class Replay(plugins.MultSliceAssociation):
        def __init__(self, *a, **k):
            super().__init__(*a, generator_mode=False, **k)

    cfg = dict(comm_env.DEFAULT_CONFIGS["mult_slice"], max_number_steps=steps)
    env = comm_env.MARLCommEnv(plugins.FixedSE, plugins.MultSliceTraffic, plugins.SimpleMobility, Replay, "mult_slice",
                               "stub_agent", 10, root_path=str(tmp_path), config=cfg, max_episode_number=2, save_hist=True)
    agent = _Agent(env)
    env.set_agent_functions(agent.obs_space_format, agent.action_format, agent.calculate_reward)
    for episode in range(2):
        obs, info = env.reset(options={"initial_episode": 0}) if episode == 0 else env.reset()
        assert env.comm_env.episode_number == episode
        for t in range(steps):
            obs, reward, term, trunc, info = env.step({"player_0": np.zeros(S)})
            assert term["__all__"] == (t == steps - 1) and term["player_1"] == term["__all__"] and not trunc["__all__"]
        data = np.load(tmp_path / "hist" / "mult_slice" / "stub_agent" / f"ep_{episode}.npz", allow_pickle=True)
        assert set(data.files) == set(HIST_KEYS)
        assert data["pkt_incoming"].shape == (steps, U) and data["sched_decision"].shape == (steps, 1, U, 135)
        assert data["reward"][steps - 1]["player_1"] == -1.0
        assert data["slice_ue_assoc"].shape == (steps, S, U)
    assert env._dev.calls.count("step_dense") == 2 * steps and env._dev.calls.count("reset") == 2
