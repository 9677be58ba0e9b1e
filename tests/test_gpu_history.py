"""Batched history writer: BatchedRanEnv.record(envs) keeps the traces of selected envs on the device and writes
hist/{scenario}/{agent}/ep_{n}.npz (results/gen_results.py:88-108) at done; values against the oracle's trace."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


def test_batched_recorder_writes_reference_history_files(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from intent_radio_sched_multi_slice_amd.history import HIST_KEYS
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    from oracle import pyoracle
    steps, B, rec_envs = 12, 6, [1, 4]
    wl = make_mult_slice_workload(B, torch.device("cuda", 0), policy=2, intra=1, n_scenarios=4, n_traces=3, trace_len=8,
                                  n_slices=5, n_ues=25, n_rbs=135, rbs_per_rbg=5, max_ues_slice=10, max_steps=steps)
    env = wl.env
    S, U, R = env.S, env.U, env.R
    rec = env.record(rec_envs, root_path=str(tmp_path), simu_name="mult_slice", agent_name="mapf", episode_numbers=[7, 30])
    cfg = pyoracle.make_cfg(S, U, R, env.G, env.Us, max_steps=steps)
    se_host = wl.se_pool.transpose(1, 2).contiguous().cpu().numpy()
    trf_host = wl.traffic_pool.cpu().numpy().astype(np.float64)
    eps, L = env.episodes, wl.trace_len
    intra = np.full(S, 1, dtype=np.int32)
    oenvs = {}
    for b in rec_envs:
        o = pyoracle.OracleEnv(cfg); o.set_scenario(wl.tables, int(wl.scenario[b])); oenvs[b] = o
    for episode in range(2):
        env.reset()
        trace = {b: [] for b in rec_envs}
        for b, o in oenvs.items():
            o.reset(se_host[int(eps["se_base"][b] + eps["se_offset"][b] % L)])
        for t in range(steps):
            env.step()
            for b, o in oenvs.items():
                tile = int(eps["se_base"][b] + (eps["se_offset"][b] + t) % L)
                row = int(eps["trf_base"][b] + (eps["trf_offset"][b] + t) % L)
                sc = o.policy_mapf()
                _, _, dense = o.action_format(sc, intra, want_dense=True)
                o.step(sc, intra, se_host[tile], trf_host[row])
                trace[b].append((o.raw(), o.obs(), dense.copy(), sc.copy(), se_host[tile].astype(np.float64)))
        assert len(rec.written) == 2 * (episode + 1)
        for k, b in enumerate(rec_envs):
            ep_no = [7, 30][k] + episode
            data = np.load(tmp_path / "hist" / "mult_slice" / "mapf" / f"ep_{ep_no}.npz", allow_pickle=True)
            assert set(data.files) == set(HIST_KEYS)
            scen = int(wl.scenario[b])
            bua, bsa, sua, req = wl.tables.to_reference(scen)
            assert data["slice_ue_assoc"].shape == (steps, S, U) and np.array_equal(data["slice_ue_assoc"][3], sua)
            assert data["mobility"].shape == (steps, U, 2)
            assert data["basestation_slice_assoc"].shape == (steps, 1, S)
            assert data["slice_req"][0].keys() == req.keys()
            for t in range(steps):
                raw, oo, dense, sc, se = trace[b][t]
                for name in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts", "buffer_occupancies",
                             "buffer_latencies"):
                    assert np.array_equal(data[name][t], raw[name]), (episode, b, t, name)
                assert np.array_equal(data["sched_decision"][t, 0], dense.astype(np.float64)), (episode, b, t)
                assert np.array_equal(data["spectral_efficiencies"][t, 0], se)
                np.testing.assert_allclose(data["obs"][t]["player_0"]["observations"], oo["obs_inter"], rtol=0, atol=1e-5)
                assert np.array_equal(data["obs"][t]["player_0"]["action_mask"], oo["mask_inter"])
                np.testing.assert_allclose(data["obs"][t]["player_2"]["observations"], oo["obs_intra"][1], rtol=0, atol=1e-5)
                np.testing.assert_allclose([data["reward"][t][f"player_{i}"] for i in range(S + 1)], oo["reward"], rtol=0, atol=1e-9)
                np.testing.assert_allclose(data["agent_action"][t]["player_0"], sc, rtol=0, atol=1e-12)
                assert data["agent_action"][t]["player_1"] == 1
    env.record(None)
    env.close()
