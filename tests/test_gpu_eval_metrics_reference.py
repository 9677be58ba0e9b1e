"""The device's episode metrics (ranenv_enable_metrics, BatchedRanEnv.evaluate) and the batched history writer against
tests/golden/eval_metrics.npz: the reference's own results/gen_results.py:845-1022 (calc_slice_violations,
calc_intent_distance) run on history files of the same closed loop (3 consecutive episodes of one env, MAPF + PF,
written by this build's history.py from the CPU oracle; tests/golden/gen_golden_r3.py).

* default flags: the 10-TTI window is never cleared, like the reference agent's deque (agents/ib_sched.py:51,64) -> the
  fixture's ``live_deque`` (the reference's functions over the whole run, reset observations included, as one sequence)
* RANENV_F_CLEAR_HISTORY_ON_RESET: the window restarts at every reset -> ``restarted_with_reset`` (per episode, the
  reset observation in front)
What gen_results.py gives per history FILE (a fresh deque per file that never sees a reset observation) differs from both
at an episode's first TTIs only; tests/test_reference_goldens_r3.py pins that difference on the fixture itself.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from tests.common import load_golden, tables_from
from tests.synth import se_tile

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _env_from_fixture(flags=0, B=2):
    from intent_radio_sched_multi_slice_amd.batched_env import BatchedRanEnv
    fx = load_golden("eval_metrics")
    S, U, R, G, Us, seed, steps, n_ep = (int(x) for x in fx["cfg"])
    tabs = tables_from(fx)
    env = BatchedRanEnv(batch=B, n_slices=S, n_ues=U, n_rbs=R, rbs_per_rbg=G, max_ues_slice=Us, n_scenarios=tabs.n_scenarios,
                        max_steps=steps, flags=flags)
    env.load_scenarios(tabs)
    se = np.stack([np.ascontiguousarray(se_tile(seed + ep, t, U, R).T) for ep in range(n_ep) for t in range(steps)])
    env.bind_se_pool(torch.as_tensor(se, device=env.device))
    env.bind_traffic_pool(torch.as_tensor(fx["traffic"].reshape(n_ep * steps, U).astype(np.int32), device=env.device))
    ep = np.arange(n_ep)
    env.set_episode_table(scenario=fx["scen_ids"], se_base=ep * steps, se_len=steps, trf_base=ep * steps, trf_len=steps)
    env.set_policy(2, 1)                                   # MAPF + PF on the device
    env.enable_autoreset(0, n_ep, episode_numbers=np.zeros(B, dtype=np.int32))
    return fx, env, (S, U, R, steps, n_ep)


@pytest.mark.parametrize("window", ["live", "restarted"])
def test_per_tti_device_metrics_equal_the_reference_evaluation_code(window):
    _need_gpu()
    from intent_radio_sched_multi_slice_amd._lib import F_CLEAR_HISTORY_ON_RESET
    fx, env, (S, U, R, steps, n_ep) = _env_from_fixture(F_CLEAR_HISTORY_ON_RESET if window == "restarted" else 0)
    want = fx["live_deque" if window == "live" else "restarted_with_reset"]          # [ep, t, (viol, prio viol, dist, prio dist)]
    env.enable_metrics(n_ep)
    env.reset()
    m = env.episode_metrics()
    prev = np.zeros(8)
    for ep in range(n_ep):
        for t in range(steps):
            obs, rew, done = env.step()
            v = {k: x[0].cpu().numpy() for k, x in env.views().items()}
            last = t == steps - 1
            assert bool(done[0]) == last
            # the closed loop itself: integers bit for bit (after a terminal TTI the views already show the reset state
            # of the next episode: the step's rewards and metric sums are what survives the auto-reset)
            if not last:
                for name in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts", "rb_count"):
                    assert np.array_equal(v[name].astype(np.float64), fx[f"{window}_{name}"][ep, t].astype(np.float64)), (window, ep, t, name)
                np.testing.assert_allclose(v["policy_scores"], fx[f"{window}_scores"][ep, t], rtol=0, atol=1e-12)
            np.testing.assert_allclose(rew[0].cpu().numpy(), fx[f"{window}_reward"][ep, t], rtol=0, atol=1e-9)
            # this TTI's share of the running sums (at a terminal TTI the sums moved to the episode log and were zeroed)
            now = (m["episode_log"][0, ep] if last else m["running"][0]).cpu().numpy()
            d = now - prev
            prev = np.zeros(8) if last else now
            assert d[0] == 1.0
            assert d[2] == want[ep, t, 0] and d[3] == want[ep, t, 1], (window, ep, t, d, want[ep, t])
            np.testing.assert_allclose(d[4:6], want[ep, t, 2:4], rtol=0, atol=1e-9)
            np.testing.assert_allclose(d[1], fx[f"{window}_reward"][ep, t, 0], rtol=0, atol=1e-9)
    assert m["episodes_done"].cpu().numpy().tolist() == [n_ep, n_ep]
    env.close()


def test_evaluate_sums_equal_the_reference_evaluation_code():
    """BatchedRanEnv.evaluate (one rollout through 3 episodes per env, over 2 partitions) -> per-episode sums."""
    _need_gpu()
    fx, env, (S, U, R, steps, n_ep) = _env_from_fixture()
    env.enable_metrics(n_ep)
    env.set_partitions(2)
    res = env.evaluate(n_ep)
    want = fx["live_deque"].sum(axis=1)                                              # [ep, 4]
    for b in range(2):
        assert np.array_equal(res["violations"][b], want[:, 0]) and np.array_equal(res["priority_violations"][b], want[:, 1])
        np.testing.assert_allclose(res["distance"][b], want[:, 2], rtol=0, atol=1e-8)
        np.testing.assert_allclose(res["priority_distance"][b], want[:, 3], rtol=0, atol=1e-8)
        np.testing.assert_allclose(res["reward"][b], fx["live_reward"][:, :, 0].sum(axis=1), rtol=0, atol=1e-8)
        assert np.array_equal(res["pkts_sent"][b], fx["live_pkt_effective_thr"].sum(axis=(1, 2)))
        assert np.array_equal(res["pkts_dropped"][b], fx["live_dropped_pkts"].sum(axis=(1, 2)))
    # for the record: what gen_results.py reports from one file per episode differs by the episodes' first TTIs only
    per_file = fx["live_per_file"].sum(axis=1)
    assert np.abs(per_file[:, 0] - want[:, 0]).max() <= 5 and np.abs(per_file[:, 2] - want[:, 2]).max() < 5.0
    env.close()


def test_recorder_follows_the_device_through_three_episodes(tmp_path):
    """BatchedRanEnv.record + device auto-reset: one history file per episode, named by the episode number the device
    moved to, each holding that episode's own scenario / channel trace / steps; the arrays are the fixture's (the inputs
    the reference's evaluation code was run on)."""
    _need_gpu()
    fx, env, (S, U, R, steps, n_ep) = _env_from_fixture()
    seed = int(fx["cfg"][5])
    tabs = tables_from(fx)
    rec = env.record([1], root_path=str(tmp_path), simu_name="mult_slice", agent_name="mapf", episode_numbers=[0])
    env.reset()
    for _ in range(n_ep * steps):
        env.step()
    assert len(rec.written) == n_ep and [p.split("ep_")[-1] for p in rec.written] == ["0.npz", "1.npz", "2.npz"]
    for ep, path in enumerate(rec.written):
        d = np.load(path, allow_pickle=True)
        scen = int(fx["scen_ids"][ep])
        bua, bsa, sua, req = tabs.to_reference(scen)
        assert d["obs"].shape[0] == steps
        for name in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts", "buffer_occupancies", "buffer_latencies"):
            assert np.array_equal(d[name], fx["live_" + name][ep]), (ep, name)
        assert np.array_equal(d["slice_ue_assoc"][0], sua) and np.array_equal(d["slice_ue_assoc"][-1], sua)
        assert np.array_equal(d["basestation_slice_assoc"][5], bsa)
        assert np.array_equal(d["sched_decision"][:, 0].sum(axis=2), fx["live_rb_count"][ep])
        for t in (0, 17, steps - 1):
            assert np.array_equal(d["spectral_efficiencies"][t, 0], se_tile(seed + ep, t, U, R).astype(np.float64))
            np.testing.assert_allclose(d["agent_action"][t]["player_0"], fx["live_scores"][ep, t], rtol=0, atol=1e-12)
            np.testing.assert_allclose(d["reward"][t]["player_0"], fx["live_reward"][ep, t, 0], rtol=0, atol=1e-9)
        names = {k: (v["name"] if v else None) for k, v in d["slice_req"][3].items()}
        assert names == {k: (v["name"] if v else None) for k, v in req.items()}
    env.close()


def test_recorder_with_masked_resets_and_per_env_episode_lengths(tmp_path):
    """Every recorded env keeps its own step counter: a masked reset restarts only the masked env's trace, envs with
    different max_steps write files of their own length."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    wl = make_mult_slice_workload(4, torch.device("cuda", 0), policy=2, intra=1, n_scenarios=6, n_traces=12, trace_len=10,
                                  n_slices=5, n_ues=25, n_rbs=135, rbs_per_rbg=5, max_ues_slice=10, max_steps=9)
    env = wl.env
    env.set_max_steps([9, 5, 7, 9])
    rec = env.record([0, 1, 2], root_path=str(tmp_path), agent_name="a", episode_numbers=[10, 20, 30])
    env.reset()
    lens = {}
    for t in range(9):
        obs, rew, done = env.step()
        d = done.cpu().numpy().astype(bool)
        if t == 2:
            env.reset(env_mask=np.array([1, 0, 0, 0], dtype=np.uint8))     # env 0 starts over: 6 more steps fit before t = 8
        if d.any():
            env.reset(env_mask=d.astype(np.uint8))
    for p in rec.written:
        lens[p.split("ep_")[-1]] = np.load(p, allow_pickle=True)["obs"].shape[0]
    # env 1: 5 steps (episode 20), then 4 more of episode 21 (unfinished, not written); env 2: 7 steps; env 0: reset at t = 2,
    # so its 9-step episode is not over at t = 8
    assert lens == {"20.npz": 5, "30.npz": 7}
    assert rec.t.tolist() == [6, 4, 2]
    env.close()
