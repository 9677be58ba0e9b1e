"""Episode metrics kept on the device (ranenv_enable_metrics) and rollouts that run through episode ends
(ranenv_rollout with auto-reset): against per-TTI sums formed from the oracle's observations, and against the
step-by-step auto-reset path that tests/test_gpu_autoreset_traffic.py checks against the oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _small_workload(B, steps, **kw):
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    return make_mult_slice_workload(B, torch.device("cuda", 0), policy=2, intra=1, n_scenarios=6, n_traces=12, trace_len=10,
                                    n_slices=5, n_ues=25, n_rbs=135, rbs_per_rbg=5, max_ues_slice=10, max_steps=steps, **kw)


def _tti_metrics(o):
    """What one TTI adds to the running sums, from the oracle's formatted observation (agents/common.py:389-427 names:
    active_observations = per-slice minimum drift, undeclared metrics count as 0) and raw metrics."""
    oo, raw = o.obs(), o.raw()
    rows = oo["obs_inter"].reshape(-1, 10)
    ao, prio = rows[:, 0:3].min(axis=1), rows[:, 6]
    neg, pneg = ao < 0.0, prio * ao < 0.0
    return np.array([1.0, oo["reward"][0], neg.sum(), pneg.sum(), ao[neg].sum(), ao[pneg].sum(),
                     raw["pkt_effective_thr"].sum(), raw["dropped_pkts"].sum()])


def test_running_sums_match_per_tti_sums_from_the_oracle():
    _need_gpu()
    from oracle import pyoracle
    B, steps = 12, 40
    wl = _small_workload(B, steps)
    env, tabs = wl.env, wl.tables
    S, U, R = env.S, env.U, env.R
    env.enable_metrics(0)
    cfg = pyoracle.make_cfg(S, U, R, env.G, env.Us, max_steps=steps)
    se_host = wl.se_pool.transpose(1, 2).contiguous().cpu().numpy()
    trf_host = wl.traffic_pool.cpu().numpy().astype(np.float64)
    ep = env.episodes
    tile = lambda b, t: int(ep[b]["se_base"] + (ep[b]["se_offset"] + t) % ep[b]["se_len"])
    trow = lambda b, t: int(ep[b]["trf_base"] + (ep[b]["trf_offset"] + t) % ep[b]["trf_len"])
    oenvs = []
    for b in range(B):
        o = pyoracle.OracleEnv(cfg); o.set_scenario(tabs, int(ep[b]["scenario"])); o.reset(se_host[tile(b, 0)]); oenvs.append(o)
    env.reset()
    m = env.episode_metrics()
    assert "episode_log" not in m and float(m["running"].abs().sum()) == 0.0
    intra = np.full(S, 1, dtype=np.int32)
    exp = np.zeros((B, 8))
    saw_violation = False
    for t in range(steps):
        env.step()
        for b, o in enumerate(oenvs):
            o.step(o.policy_mapf(), intra, se_host[tile(b, t)], trf_host[trow(b, t)])
            exp[b] += _tti_metrics(o)
        got = m["running"].cpu().numpy()
        assert np.array_equal(got[:, [0, 2, 3, 6, 7]], exp[:, [0, 2, 3, 6, 7]]), t          # counts: exact
        np.testing.assert_allclose(got[:, [1, 4, 5]], exp[:, [1, 4, 5]], rtol=0, atol=1e-9 * (t + 1))
        saw_violation |= bool(exp[:, 2].sum() > 0)
    assert saw_violation and exp[:, 6].sum() > 0
    # a masked reset starts the reset envs' sums again, the others keep theirs
    mask = (np.arange(B) % 2 == 0).astype(np.uint8)
    env.reset(env_mask=mask)
    got = m["running"].cpu().numpy()
    assert np.all(got[mask == 1] == 0.0) and np.array_equal(got[mask == 0][:, 0], exp[mask == 0][:, 0])
    # switched off: the sums stand still
    env.disable_metrics()
    env.step()
    assert np.array_equal(m["running"].cpu().numpy(), got)
    env.close()


@pytest.mark.parametrize("parts,random_episodes", [(1, False), (3, False), (3, True)])
def test_rollout_through_episode_ends_equals_stepwise_autoreset(parts, random_episodes):
    """Two identical batches, staggered episode lengths (17..29 TTIs), 100 TTIs = 3 to 5 episodes per env: one stepped
    with step() (auto-reset after every step, the path the oracle test covers), the other by two rollout() calls that run
    through the episode ends.  State, outputs, episode numbers and the per-episode metric logs must be identical."""
    _need_gpu()
    B, L, n_ep, first, total = 48, 10, 12, 3, 100
    envs = []
    for k in range(2):
        wl = _small_workload(B, 50)
        env = wl.env
        ep_no = np.arange(first, first + n_ep)
        env.set_episode_table(scenario=ep_no % 6, se_base=(ep_no % 12) * L, se_len=L, se_offset=ep_no % L,
                              trf_base=(ep_no % 6) * L, trf_len=L, trf_offset=(ep_no * 3) % L, first_episode=first)
        env.set_max_steps(17 + np.arange(B) % 13)
        env.enable_autoreset(first, first + n_ep, random_episodes=random_episodes, seed=7, episode_numbers=first + np.arange(B) % n_ep)
        env.enable_metrics(6)
        env.reset()
        envs.append(env)
    a, b = envs
    for _ in range(total):
        a.step()
    b.set_partitions(parts)
    b.rollout(37)
    obs_b, rew_b, done_b = b.rollout(total - 37)
    torch.cuda.synchronize()
    va, vb = a.views(), b.views()
    for name in va:
        assert torch.equal(va[name], vb[name]), name
    ma, mb = a.episode_metrics(), b.episode_metrics()
    for name in ma:
        assert torch.equal(ma[name], mb[name]), name
    n_done = mb["episodes_done"].cpu().numpy()
    assert n_done.min() >= 3 and n_done.max() <= 5
    log = mb["episode_log"].cpu().numpy()
    max_steps = 17 + np.arange(B) % 13
    for e in range(B):
        assert np.array_equal(log[e, :n_done[e], 0], np.full(n_done[e], max_steps[e]))      # every logged episode ran its full length
        assert np.all(log[e, n_done[e]:] == 0.0)
    run = mb["running"].cpu().numpy()
    assert np.array_equal(run[:, 0], total - n_done * max_steps)                              # TTIs into the current episode
    assert np.array_equal(run[:, 0], vb["step_number"].cpu().numpy())
    assert torch.equal(a.obs_inter, b.obs_inter) and torch.equal(a.obs_intra, b.obs_intra)
    assert torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done)
    a.close(); b.close()


def test_evaluate_runs_whole_episodes_on_the_device_and_matches_the_oracle():
    """evaluate(): reset + one rollout through n episodes per env; the per-episode sums against an oracle that plays the
    same episodes (sequential episode numbers, equal lengths)."""
    _need_gpu()
    from oracle import pyoracle
    B, L, n_ep, first, steps, n_eval = 8, 10, 12, 3, 20, 3
    wl = _small_workload(B, steps)
    env, tabs = wl.env, wl.tables
    S, U, R = env.S, env.U, env.R
    ep_no = np.arange(first, first + n_ep)
    env.set_episode_table(scenario=ep_no % 6, se_base=(ep_no % 12) * L, se_len=L, se_offset=ep_no % L,
                          trf_base=(ep_no % 6) * L, trf_len=L, trf_offset=(ep_no * 3) % L, first_episode=first)
    start = first + np.arange(B) % n_ep
    env.enable_autoreset(first, first + n_ep, episode_numbers=start)
    env.enable_metrics(n_eval)
    env.set_partitions(2)
    res = env.evaluate(n_eval)
    assert set(res) == set(env.METRIC_NAMES) and res["ttis"].shape == (B, n_eval) and np.all(res["ttis"] == steps)
    cfg = pyoracle.make_cfg(S, U, R, env.G, env.Us, max_steps=steps)
    se_host = wl.se_pool.transpose(1, 2).contiguous().cpu().numpy()
    trf_host = wl.traffic_pool.cpu().numpy().astype(np.float64)
    tab = env.episode_table
    intra = np.full(S, 1, dtype=np.int32)
    for b in range(B):
        ep = int(start[b])
        o = pyoracle.OracleEnv(cfg)                 # one object per env: the 10-TTI window survives a reset
        for k in range(n_eval):
            r = tab[ep - first]
            o.set_scenario(tabs, int(r["scenario"]))
            o.reset(se_host[int(r["se_base"] + r["se_offset"] % r["se_len"])])
            exp = np.zeros(8)
            for t in range(steps):
                o.step(o.policy_mapf(), intra, se_host[int(r["se_base"] + (r["se_offset"] + t) % r["se_len"])],
                       trf_host[int(r["trf_base"] + (r["trf_offset"] + t) % r["trf_len"])])
                exp += _tti_metrics(o)
            got = np.array([res[n][b, k] for n in env.METRIC_NAMES])
            assert np.array_equal(got[[0, 2, 3, 6, 7]], exp[[0, 2, 3, 6, 7]]), (b, k, got, exp)
            np.testing.assert_allclose(got[[1, 4, 5]], exp[[1, 4, 5]], rtol=0, atol=1e-8)
            ep = ep + 1 if ep + 1 < first + n_ep else first
    with pytest.raises(Exception, match="episode_slots"):
        env.evaluate(n_eval + 1)
    env.close()


def test_metrics_api_errors():
    _need_gpu()
    from intent_radio_sched_multi_slice_amd._lib import RanEnvError
    wl = _small_workload(4, 10)
    env = wl.env
    with pytest.raises(RanEnvError, match="not enabled"):
        env.episode_metrics()
    env.enable_metrics(2)
    with pytest.raises(RanEnvError, match="2 slots"):
        env.enable_metrics(3)
    env.enable_metrics(2)            # same size: zeroes and switches on again
    env.close()
