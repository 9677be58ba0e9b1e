"""History files (history.py): schema of results/gen_results.py:88-108, read back with that script's access pattern."""
import numpy as np
import pytest

from intent_radio_sched_multi_slice_amd.history import HIST_KEYS, hist_path, write_episode_npz


def _fake_hist(T=7, S=3, U=6, R=9, marl=True):
    rng = np.random.default_rng(0)
    req = {f"slice_{s}": ({"name": "x", "ues": {"traffic": 5}} if s < 2 else {}) for s in range(S)}
    h = {k: [] for k in HIST_KEYS}
    for t in range(T):
        for k in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "buffer_occupancies", "buffer_latencies", "dropped_pkts"):
            h[k].append(rng.random(U))
        h["mobility"].append(np.ones((U, 2)))
        h["spectral_efficiencies"].append(rng.random((1, U, R)))
        h["basestation_ue_assoc"].append(np.ones((1, U))); h["basestation_slice_assoc"].append(np.ones((1, S)))
        h["slice_ue_assoc"].append(np.zeros((S, U))); h["sched_decision"].append(np.zeros((1, U, R)))
        h["slice_req"].append(req)
        if marl:
            h["reward"].append({f"player_{i}": float(t + i) for i in range(S + 1)})
            h["obs"].append({"player_0": {"observations": rng.random(10 * S), "action_mask": np.ones(S, dtype=np.int8)}})
            h["agent_action"].append({"player_0": rng.random(S), "player_1": 1})
        else:
            h["reward"].append(float(t)); h["obs"].append(rng.random(10 * S)); h["agent_action"].append(rng.random(S))
    return h


@pytest.mark.parametrize("marl", [True, False])
def test_history_file_reads_back_like_gen_results(tmp_path, marl):
    T, S, U, R = 7, 3, 6, 9
    path = hist_path(str(tmp_path), "mult_slice", "ib_sched", 4)
    assert path.endswith("hist/mult_slice/ib_sched/ep_4.npz")
    write_episode_npz(path, _fake_hist(T, S, U, R, marl))
    data = np.load(path, allow_pickle=True)                                   # gen_results.py:88-91
    data_metrics = {k: data[k] for k in HIST_KEYS}                            # :92-108
    assert set(data.files) == set(HIST_KEYS)
    assert data_metrics["pkt_incoming"].shape == (T, U)
    assert np.squeeze(data_metrics["spectral_efficiencies"]).shape == (T, U, R)          # :262
    assert data_metrics["sched_decision"][:, 0, :, :].shape == (T, U, R)                 # :629
    assert data_metrics["slice_ue_assoc"][:, 1, :].shape == (T, U)                        # :279
    assert data_metrics["slice_req"][2]["slice_0"]["ues"]["traffic"] == 5                 # :818
    assert data_metrics["slice_req"][2]["slice_2"] == {}                                  # :821
    if marl:
        assert [data_metrics["reward"][i]["player_0"] for i in range(data_metrics["reward"].shape[0])] == list(map(float, range(T)))  # :162
    else:
        assert data_metrics["reward"].tolist() == list(map(float, range(T)))                                                     # :167
        assert data_metrics["obs"].shape == (T, 10 * S) and data_metrics["agent_action"][:, 1].shape == (T,)                     # :677-700


def test_history_rejects_missing_keys(tmp_path):
    h = _fake_hist()
    del h["obs"]
    with pytest.raises(ValueError, match="missing"):
        write_episode_npz(str(tmp_path / "x.npz"), h)
