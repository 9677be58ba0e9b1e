"""C-ABI behaviour besides the step arithmetic: config flags, the done flag, error reporting.

All through BatchedRanEnv -> libranenv_hip.so on the GPU; the oracle is the checker.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
from tests.common import poisson_traffic_rows
from tests.synth import se_tile

pytestmark = pytest.mark.gpu
OBS_TOL, REW_TOL = 1e-5, 1e-9


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def _setup(flags=0, steps=12, B=4, seed=3, hist_depth=10):
    from intent_radio_sched_multi_slice_amd.batched_env import BatchedRanEnv
    from intent_radio_sched_multi_slice_amd.scenario import generate_scaled_scenarios
    S, U, R, G, Us = 5, 25, 135, 1, 5
    tabs = generate_scaled_scenarios(3, seed=seed, n_slices=S, n_ues=U, max_ues_slice=Us, min_slices=3, min_ues=2)
    rng = np.random.default_rng(seed)
    scen = rng.integers(0, tabs.n_scenarios, B)
    se_pool = np.stack([se_tile(33, t, U, R) for t in range(B * steps)])
    trf = np.concatenate([poisson_traffic_rows(tabs, int(scen[b]), rng, steps) for b in range(B)])
    env = BatchedRanEnv(batch=B, n_slices=S, n_ues=U, n_rbs=R, rbs_per_rbg=G, max_ues_slice=Us,
                        n_scenarios=tabs.n_scenarios, max_steps=steps, flags=flags, hist_depth=hist_depth)
    env.load_scenarios(tabs)
    env.bind_se_pool(torch.as_tensor(np.ascontiguousarray(np.swapaxes(se_pool, -1, -2)), device=env.device))
    env.bind_traffic_pool(torch.as_tensor(trf.astype(np.int32), device=env.device))
    env.set_episodes(scenario=scen, se_base=np.arange(B) * steps, se_len=steps, trf_base=np.arange(B) * steps, trf_len=steps)
    return env, tabs, scen, se_pool, trf, (S, U, R, G, Us)


def _oracles(tabs, scen, dims, steps, hist_depth=10):
    from oracle import pyoracle
    S, U, R, G, Us = dims
    cfg = pyoracle.make_cfg(S, U, R, G, Us, max_steps=steps, hist_depth=hist_depth)
    out = []
    for b in range(len(scen)):
        o = pyoracle.OracleEnv(cfg); o.set_scenario(tabs, int(scen[b])); out.append(o)
    return out


def test_clear_history_on_reset_matches_a_fresh_env():
    """RANENV_F_CLEAR_HISTORY_ON_RESET: after reset the 10-TTI window is empty, i.e. the env behaves like a
    newly created one (the reference's deque is never cleared, agents/ib_sched.py:51; this flag is the opt-in)."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd import _lib
    steps = 12
    env, tabs, scen, se_pool, trf, dims = _setup(flags=_lib.F_CLEAR_HISTORY_ON_RESET, steps=steps)
    S = dims[0]
    env.set_policy(2, 1)
    env.reset()
    for _ in range(5):
        env.step()
    env.reset()
    v = env.views()
    assert int(v["hist_len"].max()) == 1 and int(v["win_sent"].abs().max()) == 0 and int(v["win_dropped"].abs().max()) == 0
    oenvs = _oracles(tabs, scen, dims, steps)          # fresh oracles: empty deque
    for b, o in enumerate(oenvs):
        o.reset(se_pool[b * steps])
    for t in range(6):
        sc = np.stack([o.policy_mapf() for o in oenvs]); ic = np.ones((len(oenvs), S), dtype=np.uint8)
        obs, rew, done = env.step()
        for b, o in enumerate(oenvs):
            o.step(sc[b], ic[b], se_pool[b * steps + t], trf[b * steps + t])
            oo = o.obs()
            np.testing.assert_allclose(obs["obs_inter"][b].cpu().numpy(), oo["obs_inter"], rtol=0, atol=OBS_TOL)
            np.testing.assert_allclose(rew[b].cpu().numpy(), oo["reward"], rtol=0, atol=REW_TOL)
    env.close()


def test_no_raw_output_flag_only_skips_the_two_arrays():
    _need_gpu()
    from intent_radio_sched_multi_slice_amd import _lib
    a, *_ = _setup(flags=0)
    b, *_ = _setup(flags=_lib.F_NO_RAW_OUTPUT)
    for env in (a, b):
        env.set_policy(1, 0)
        env.reset()
    for _ in range(6):
        oa, ra, _ = a.step()
        ob, rb, _ = b.step()
        assert torch.equal(oa["obs_inter"], ob["obs_inter"]) and torch.equal(oa["obs_intra"], ob["obs_intra"]) and torch.equal(ra, rb)
    va, vb = a.views(), b.views()
    for k in ("pkt_effective_thr", "dropped_pkts", "queue_pkts", "queue_age_sum", "rb_count"):
        assert torch.equal(va[k], vb[k]), k
    assert int(va["pkt_incoming"].abs().sum()) > 0 and int(vb["pkt_incoming"].abs().sum()) == 0
    assert int(vb["pkt_throughputs"].abs().sum()) == 0
    a.close(); b.close()


def test_sync_check_flag_changes_nothing_but_the_waiting():
    """RANENV_F_SYNC_CHECK: every launch waits for its kernels (an asynchronous fault would be reported by the call that
    caused it); results are those of the ordinary, enqueue-only mode."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd import _lib
    a, *_ = _setup(flags=0)
    b, *_ = _setup(flags=_lib.F_SYNC_CHECK)
    for env in (a, b):
        env.set_policy(2, 1)
        env.reset()
    for _ in range(5):
        oa, ra, _ = a.step()
        ob, rb, _ = b.step()
    torch.cuda.synchronize()
    assert torch.equal(oa["obs_inter"], ob["obs_inter"]) and torch.equal(ra, rb)
    for k, v in a.views().items():
        assert torch.equal(v, b.views()[k]), k
    a.close(); b.close()


def test_shorter_trace_mid_episode_cannot_read_past_the_pool():
    """A trace position persisted under a longer trace is reduced when set_episodes installs a shorter one mid-episode
    (no reset in between): the next step reads tile 0 of the new trace -- inside the bound pool -- not position 9 of a
    4-tile trace."""
    _need_gpu()
    steps, B = 12, 4
    env, tabs, scen, se_pool, trf, dims = _setup(steps=steps, B=B)
    S, U, R, G, Us = dims
    env.set_policy(1, 0)
    env.reset()
    for _ in range(9):
        env.step()                                    # positions are now 9 in 12-tile traces
    short = 4
    base = (np.arange(B) * steps + steps - short).astype(np.int64)          # the LAST 4 tiles of each env's range:
    env.set_episodes(scenario=scen, se_base=base, se_len=short,             # position 9 would lie beyond the pool for env B-1
                     trf_base=np.arange(B) * steps, trf_len=short)
    v = env.views()
    q_before = v["queue_pkts"].clone()
    env.step()
    torch.cuda.synchronize()
    # the step used tile base + 0: the mean SE it stored is that tile's
    # (of the UEs in a slice: a step leaves UEs outside every slice alone when their traffic traces are empty)
    want = np.stack([se_pool[int(base[b])].astype(np.float64).mean(axis=1) for b in range(B)])
    in_slice = np.stack([tabs.ue_slice[int(scen[b])] >= 0 for b in range(B)])
    np.testing.assert_allclose(v["se_mean"].cpu().numpy()[in_slice], want[in_slice], rtol=1e-12, atol=0)
    assert int(v["step_number"][0]) == 10 and torch.isfinite(env.reward).all()
    env.close()


def test_done_flag_and_step_counter():
    _need_gpu()
    steps = 5
    env, *_ = _setup(steps=steps)
    env.set_policy(1, 0)
    env.reset()
    for t in range(steps):
        _, _, done = env.step()
        assert int(env.views()["step_number"].min()) == t + 1
        assert bool(done.all()) == (t + 1 >= steps)
    env.reset()
    assert int(env.views()["step_number"].max()) == 0
    env.close()


def test_errors_are_reported_not_swallowed():
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.batched_env import BatchedRanEnv, RanEnvError
    from intent_radio_sched_multi_slice_amd.scenario import generate_scaled_scenarios
    with pytest.raises(RanEnvError, match="unsupported sizes"):
        BatchedRanEnv(batch=1, n_slices=17, n_ues=10, n_rbs=20)
    with pytest.raises(RanEnvError, match="unsupported sizes"):
        BatchedRanEnv(batch=1, n_slices=4, n_ues=300, n_rbs=20, max_ues_slice=16)
    env = BatchedRanEnv(batch=2, n_slices=3, n_ues=9, n_rbs=20, max_ues_slice=3, n_scenarios=1)
    with pytest.raises(RanEnvError, match="no scenarios loaded"):
        env.reset()
    tabs = generate_scaled_scenarios(1, seed=1, n_slices=3, n_ues=9, max_ues_slice=3, min_slices=2, min_ues=1)
    env.load_scenarios(tabs)
    with pytest.raises(RanEnvError, match="no episode descriptors"):
        env.reset()
    with pytest.raises(RanEnvError, match="scenario 5 outside pool"):
        env.set_episodes(scenario=np.array([0, 5]))
    env.set_episodes(scenario=np.array([0, 0]))
    with pytest.raises(RanEnvError, match="no SE tiles given and no SE pool bound"):
        env.reset()
    pool = torch.ones((4, 20, 9), dtype=torch.float32, device=env.device)
    env.bind_se_pool(pool)
    with pytest.raises(RanEnvError, match="exceeds the bound pool"):
        env.set_episodes(scenario=np.array([0, 0]), se_base=np.array([0, 3]), se_len=2)
    env.set_episodes(scenario=np.array([0, 0]), se_base=np.array([0, 2]), se_len=2)
    env.reset()
    with pytest.raises(RanEnvError, match="no traffic given, no traffic pool bound and no traffic generator set"):
        env.step()
    env.bind_traffic_pool(torch.zeros((2, 9), dtype=torch.int32, device=env.device))
    env.set_episodes(scenario=np.array([0, 0]), se_base=np.array([0, 2]), se_len=2, trf_base=np.array([0, 1]), trf_len=1)
    env.set_policy(0, 255)
    with pytest.raises(RanEnvError, match="policy is EXTERNAL"):
        env.step()
    with pytest.raises(RanEnvError, match="expected shape"):
        env.step(np.zeros((2, 4)), np.zeros((2, 3), dtype=np.uint8))
    env.step(np.zeros((2, 3)), np.zeros((2, 3), dtype=np.uint8))     # and a good call still works
    # strict inputs: no hidden per-step conversions
    env.strict_inputs = True
    with pytest.raises(RanEnvError, match="strict_inputs"):
        env.step(np.zeros((2, 3)), np.zeros((2, 3), dtype=np.uint8))
    env.step(torch.zeros((2, 3), dtype=torch.float64, device=env.device), torch.zeros((2, 3), dtype=torch.uint8, device=env.device))
    env.strict_inputs = False
    # episode advance / traffic generator misuse
    with pytest.raises(RanEnvError, match="no episode table"):
        env.enable_autoreset(0, 2)
    with pytest.raises(RanEnvError, match="exceeds the bound pool"):
        env.set_episode_table(scenario=[0, 0], se_base=[0, 3], se_len=2)
    env.set_episode_table(scenario=[0, 0], se_base=[0, 2], se_len=2, first_episode=4)
    with pytest.raises(RanEnvError, match="inside the table"):
        env.enable_autoreset(0, 2)
    with pytest.raises(RanEnvError, match="max_steps must be >= 1"):
        env.set_max_steps([3, 0])
    env.enable_autoreset(4, 6, episode_numbers=[4, 5])
    env.step(np.zeros((2, 3)), np.zeros((2, 3), dtype=np.uint8))
    env.disable_autoreset()
    hot = copy_tabs = generate_scaled_scenarios(1, seed=1, n_slices=3, n_ues=9, max_ues_slice=3, min_slices=2, min_ues=1)
    hot.slice_traffic[0, hot.slice_has_req[0] != 0] = 500.0
    env.set_traffic_generator(1)
    with pytest.raises(RanEnvError, match="outside \\(0, 128\\]"):
        env.load_scenarios(hot)
    env.load_scenarios(tabs)
    # a scenario row that is not self-consistent is refused at load time
    import copy
    broken = copy.deepcopy(tabs)
    broken.sorted_slices[0, :] = 0
    with pytest.raises(RanEnvError, match="not a permutation"):
        env.load_scenarios(broken)
    env.close()


def test_se_from_power_matches_the_reference_formula():
    """Channel ingest kernel vs the expression of channels/quadriga.py:56-69 in numpy float64, stored float32.
    log2 may differ from numpy's in the last float64 bit, so the float32 results are allowed to differ by one
    float32 ulp in rare rounding ties; everything else must be identical."""
    _need_gpu()
    from oracle.pyoracle import quadriga_se_from_power          # the checker lives on the oracle side of the wall
    from intent_radio_sched_multi_slice_amd.workloads import quadriga_pool_from_power
    rng = np.random.default_rng(5)
    for shape in ((7, 135, 100), (3, 25, 4), (1, 1, 1), (5, 33, 7)):
        g = 10.0 ** rng.uniform(-16, -7, size=shape)            # received power per RB, W
        g[0, 0, 0] = 0.0                                        # no signal -> SE 0
        ref = quadriga_se_from_power(g, shape[1]).astype(np.float32)
        got = quadriga_pool_from_power(torch.as_tensor(g, device="cuda"), shape[1]).cpu().numpy()
        assert got.shape == ref.shape and got.dtype == np.float32
        ulp = np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
        assert ulp.max() <= 1, ulp.max()
        assert (ulp != 0).mean() <= 1e-3
        assert got[0, 0, 0] == 0.0


@pytest.mark.parametrize("hist_depth", [10, 5])
def test_alternative_heads_vs_oracle(hist_depth):
    """SchedTWC / SchedColORAN observation and rewards from the head kernel against the oracle (itself pinned
    to the reference's classes by tests/golden/heads_*.npz), in closed loop: external scores, round-robin
    inside the slices as their action_format does, a masked reset in the middle.  hist_depth 5: an odd deque
    length, where the oldest TTI of the heads' doubled window counts once."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.scenario import SLICE_TEMPLATES, SLICE_USECASE
    steps = 16
    env, tabs, scen, se_pool, trf, dims = _setup(steps=steps, B=6, seed=9, hist_depth=hist_depth)
    S, U, R, G, Us = dims
    # use-case bits per scenario row, recovered from the template numbers
    uc = np.zeros((tabs.n_scenarios, S), dtype=np.int32)
    for i in range(tabs.n_scenarios):
        for s in range(S):
            if tabs.slice_has_req[i, s]:
                key = (int(tabs.slice_buffer_size[i, s]), int(tabs.slice_buffer_latency[i, s]),
                       int(tabs.slice_message_size[i, s]), float(tabs.slice_traffic[i, s]))
                name = [t[0] for t in SLICE_TEMPLATES if (t[3], t[4], t[5], float(t[7])) == key][0]
                uc[i, s] = SLICE_USECASE[name]
    env.enable_heads(uc)
    env.set_policy(0, 0)                       # scores from the caller, round-robin intra-slice
    oenvs = _oracles(tabs, scen, dims, steps, hist_depth)
    env.reset()
    for b, o in enumerate(oenvs):
        o.reset(se_pool[b * steps])
    rng = np.random.default_rng(4)
    t0 = np.zeros(len(oenvs), dtype=np.int64)
    saw_neg = saw_col = False

    def check(tag):
        nonlocal saw_neg, saw_col
        ho, hr = env.head_obs.cpu().numpy(), env.head_reward.cpu().numpy()
        for b, o in enumerate(oenvs):
            obs, r_twc, r_col = o.heads(uc[scen[b]])
            np.testing.assert_allclose(ho[b], obs, rtol=1e-6, atol=OBS_TOL, err_msg=f"{tag} env {b}")
            np.testing.assert_allclose(hr[b], [r_twc, r_col], rtol=0, atol=REW_TOL, err_msg=f"{tag} env {b}")
            saw_neg |= r_twc < 0; saw_col |= r_col != 0

    check("reset")
    for t in range(12):
        if t == 7:
            mask = (np.arange(len(oenvs)) % 2 == 0).astype(np.uint8)
            env.reset(env_mask=mask)
            for b in np.nonzero(mask)[0]:
                t0[b] = t
                oenvs[b].reset(se_pool[b * steps])
            check("masked reset")
        sc = rng.uniform(-1, 1, (len(oenvs), S)); ic = np.zeros((len(oenvs), S), dtype=np.uint8)
        env.step(sc, ic)
        for b, o in enumerate(oenvs):
            k = b * steps + int(t - t0[b])
            o.step(sc[b], ic[b], se_pool[k], trf[k])
        check(f"t={t}")
    assert saw_neg and saw_col
    env.close()


def test_trainer_adapters():
    """HeadVecEnv follows the VecEnv protocol (shapes, dtypes, auto-reset with terminal_observation) and hands
    out the head kernel's numbers; the MARL dicts are the batched tensors of one env under the reference's keys."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.adapters import HeadVecEnv, marl_obs_dict, marl_reward_dict
    steps = 5
    env, tabs, scen, se_pool, trf, dims = _setup(steps=steps, B=4, seed=13)
    S, U, R, G, Us = dims
    venv = HeadVecEnv(env, reward="colran", slice_usecase=np.full((tabs.n_scenarios, S), 3, dtype=np.int32))
    obs = venv.reset()
    assert obs.shape == (4, 10 * S) and obs.dtype == np.float32 and venv.num_envs == 4
    assert venv.action_space.shape == (S,) and venv.observation_space.shape == (10 * S,)
    rng = np.random.default_rng(0)
    for t in range(2 * steps):
        obs, rew, dones, infos = venv.step(rng.uniform(-1, 1, (4, S)))
        assert obs.shape == (4, 10 * S) and rew.shape == (4,) and rew.dtype == np.float32 and dones.dtype == bool
        end = (t + 1) % steps == 0
        assert bool(dones.all()) == end and bool(dones.any()) == end
        if end:
            assert all("terminal_observation" in i for i in infos)
            assert int(env.views()["step_number"].max()) == 0          # auto-reset happened
            np.testing.assert_array_equal(obs, env.head_obs.cpu().numpy())
        else:
            np.testing.assert_allclose(rew, env.head_reward[:, 1].cpu().numpy().astype(np.float32))
    d = marl_obs_dict(env, 2); r = marl_reward_dict(env, 2)
    assert set(d) == {f"player_{i}" for i in range(S + 1)} == set(r)
    assert d["player_0"]["observations"].shape == (10 * S,) and d["player_0"]["action_mask"].shape == (S,)
    assert d["player_3"]["observations"].shape == (2 * Us + 9,) and d["player_3"]["action_mask"].shape == (Us,)
    with pytest.raises(ValueError):
        venv.step(np.zeros((3, S)))
    venv.close()


def test_degenerate_scenarios_vs_oracle():
    """No active slice at all (IBSched.action_format's `if np.sum(basestation_slice_assoc) != 0` guard,
    agents/ib_sched.py:240-246: the allocation stays all-zero), an active slice without UEs, a single UE owning every
    RB, and all of it next to an ordinary scenario in one batch: against the oracle, all four policies."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.batched_env import BatchedRanEnv
    from intent_radio_sched_multi_slice_amd.scenario import ScenarioTables, generate_scaled_scenarios, slice_template_dict
    from oracle import pyoracle
    S, U, R, G, Us, steps = 5, 25, 135, 5, 5, 8
    tabs = ScenarioTables.empty(4, S, U, Us)                                    # row 0: nothing active
    bsa = np.zeros((1, S)); sua = np.zeros((S, U)); bsa[0, 2] = 1               # row 1: slice 2 active, but no UE in it
    req = {f"slice_{s}": {} for s in range(S)}; req["slice_2"] = slice_template_dict(1)
    tabs.set_from_reference(1, bsa, sua, req, True)
    bsa = np.zeros((1, S)); sua = np.zeros((S, U)); bsa[0, 4] = 1; sua[4, 7] = 1  # row 2: one slice, one UE
    req = {f"slice_{s}": {} for s in range(S)}; req["slice_4"] = slice_template_dict(5)
    tabs.set_from_reference(2, bsa, sua, req, True)
    ordinary = generate_scaled_scenarios(1, seed=9, n_slices=S, n_ues=U, max_ues_slice=Us, min_slices=3, min_ues=2)
    for k, v in ordinary.arrays().items():
        getattr(tabs, k)[3] = v[0]
    B = 4
    rng = np.random.default_rng(2)
    se_pool = np.stack([se_tile(21, t, U, R) for t in range(steps)])
    trf = np.stack([poisson_traffic_rows(tabs, b, rng, steps) for b in range(B)])       # [B, steps, U]
    for policy, intra in ((1, 0), (2, 1), (2, 2), (0, 255)):
        env = BatchedRanEnv(batch=B, n_slices=S, n_ues=U, n_rbs=R, rbs_per_rbg=G, max_ues_slice=Us, n_scenarios=4, max_steps=steps)
        env.load_scenarios(tabs)
        env.bind_se_pool(torch.as_tensor(np.ascontiguousarray(np.swapaxes(se_pool, -1, -2)), device=env.device))
        env.bind_traffic_pool(torch.as_tensor(trf.reshape(B * steps, U).astype(np.int32), device=env.device))
        env.set_episodes(scenario=np.arange(B), se_base=0, se_len=steps, trf_base=np.arange(B) * steps, trf_len=steps)
        env.set_policy(policy, intra)
        cfg = pyoracle.make_cfg(S, U, R, G, Us, max_steps=steps)
        oenvs = []
        for b in range(B):
            o = pyoracle.OracleEnv(cfg); o.set_scenario(tabs, b); o.reset(se_pool[0]); oenvs.append(o)
        env.reset()
        for t in range(steps):
            if policy == 0:
                sc = rng.uniform(-1, 1, (B, S)); ic = rng.integers(0, 3, (B, S)).astype(np.uint8)
                obs, rew, done = env.step(sc, ic)
            else:
                obs, rew, done = env.step()
            g = {k: x.cpu().numpy() for k, x in env.views().items()}
            for b, o in enumerate(oenvs):
                if policy == 0:
                    scb, icb = sc[b], ic[b].astype(np.int32)
                else:
                    scb = o.policy_mapf() if policy == 2 else o.policy_marr()
                    icb = np.full(S, intra, dtype=np.int32)
                _, count, _ = o.action_format(scb, icb, want_dense=False)
                assert np.array_equal(g["rb_count"][b], count), (policy, t, b)
                o.step(scb, icb, se_pool[t], trf[b, t])
                raw, oo = o.raw(), o.obs()
                for name in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts"):
                    assert np.array_equal(g[name][b].astype(np.float64), raw[name]), (policy, t, b, name)
                np.testing.assert_allclose(obs["obs_inter"][b].cpu().numpy(), oo["obs_inter"], rtol=0, atol=1e-5)
                np.testing.assert_allclose(obs["obs_intra"][b].cpu().numpy(), oo["obs_intra"], rtol=0, atol=1e-5)
                np.testing.assert_allclose(rew[b].cpu().numpy(), oo["reward"], rtol=0, atol=1e-9)
            assert int(g["rb_count"][0].sum()) == 0 and int(g["rb_count"][1].sum()) == 0      # nobody to give RBs to
            assert int(g["rb_count"][2, 7]) == R and int(g["rb_count"][2].sum()) == R          # one UE owns the carrier
        env.close()


def test_partitioned_step_survives_stream_capture():
    """the partition join asked hipStreamQuery about the caller's stream, which is illegal while that stream is
    capturing.  A learner that graph-captures `env.step()` (partitions on handle-owned streams joined by events: a fork and
    a join inside the capture) must get a graph that replays to the same state as eager steps."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    dev = torch.device("cuda", 0)
    outs = []
    for mode in ("eager", "graph"):
        wl = make_mult_slice_workload(256, dev, n_scenarios=16, n_traces=8, trace_len=16, max_steps=1000)
        env = wl.env
        env.set_partitions(3)
        env.reset(); env.step(); torch.cuda.synchronize()
        if mode == "eager":
            for _ in range(6):
                env.step()
        else:
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
                    env.step(); env.step()
            for _ in range(3):
                g.replay()
        torch.cuda.synchronize()
        outs.append((env.obs_inter.clone(), env.obs_intra.clone(), env.reward.clone(), env.views()["step_number"].clone(),
                     env.views()["queue_pkts"].clone()))
        env.close()
    assert int(outs[1][3].min()) == 7
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def _captured_step_graph(env, dev, n=1):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
            for _ in range(n):
                env.step()
    return g


def test_graph_captured_mixed_step_follows_scenario_changes_between_replays():
    """a whole-batch step of two-wave workgroups runs as MIXED blocks (one block per env of more than 64 slice members,
    one per two envs of at most 64) from class lists that the HOST re-sorts when it knows the scenarios changed.  A graph
    captured while the lists were clean holds no sort; when set_episodes / reset change the envs' scenarios between replays the
    replayed launch would step an env that now has more than 64 slice members with ONE narrow wave (lanes >= 64 never stepped).
    Inside a capture the sort is therefore always enqueued in front of the mixed launch."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    dev = torch.device("cuda", 0)
    B = 512
    outs = []
    for mode in ("eager", "graph"):
        wl = make_mult_slice_workload(B, dev, n_scenarios=32, n_traces=8, trace_len=16, max_steps=1000)
        env = wl.env
        env.set_option("mix", 2); env.set_option("compact", 1)          # mixed blocks also for a batch that fits the chip
        members = (wl.tables.ue_slice >= 0).sum(axis=1)
        assert (members > 64).any() and (members <= 64).any()
        env.reset(); env.step(); torch.cuda.synchronize()
        g = _captured_step_graph(env, dev) if mode == "graph" else None
        stages = []
        for stage in range(2):
            if stage == 1:
                # every env moves to a scenario of the OTHER class where there is one
                wide = np.flatnonzero(members > 64); narrow = np.flatnonzero(members <= 64)
                cur = wl.scenario
                new = np.where(members[cur] > 64, narrow[np.arange(B) % len(narrow)], wide[np.arange(B) % len(wide)])
                env.set_episodes(scenario=new, se_base=wl.se_trace * wl.trace_len, se_len=wl.trace_len, se_offset=wl.se_offset,
                                 trf_base=new * wl.trace_len, trf_len=wl.trace_len, trf_offset=0)
                env.reset()
            for _ in range(4):
                if g is not None:
                    g.replay()
                else:
                    env.step()
            torch.cuda.synchronize()
            v = env.views()
            stages.append((env.obs_inter.clone(), env.obs_intra.clone(), env.reward.clone(), v["queue_pkts"].clone(),
                           v["pkt_effective_thr"].clone(), v["rb_count"].clone(), v["step_number"].clone()))
        outs.append(stages)
        env.close()
    for stage in range(2):
        for a, b in zip(outs[0][stage], outs[1][stage]):
            assert torch.equal(a, b), stage


def test_autoreset_loop_resorts_the_mixed_blocks_only_when_an_env_restarted():
    """The same lists in an eager auto-reset loop: the advance kernel raises a device flag when an env restarts, the (one-block)
    sort in front of the next mixed launch looks at it -- no host read-back of `done`.  Episodes of 5 TTIs whose scenarios change
    class at every restart, against the same loop with mixed blocks off."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    dev = torch.device("cuda", 0)
    B, n_ep, L = 300, 12, 16
    outs = []
    for mix in (0, 2):
        wl = make_mult_slice_workload(B, dev, n_scenarios=32, n_traces=8, trace_len=L, max_steps=1000)
        env = wl.env
        env.set_option("mix", mix); env.set_option("compact", 1)
        members = (wl.tables.ue_slice >= 0).sum(axis=1)
        wide = np.flatnonzero(members > 64); narrow = np.flatnonzero(members <= 64)
        ep = np.arange(n_ep)
        scen = np.where(ep % 2 == 0, wide[ep % len(wide)], narrow[ep % len(narrow)])        # alternating classes
        env.set_episode_table(scenario=scen, se_base=(ep % 8) * L, se_len=L, se_offset=ep % L, trf_base=scen * L, trf_len=L, trf_offset=(ep * 3) % L)
        env.set_max_steps(5 + (np.arange(B) % 3))
        env.enable_autoreset(0, n_ep, episode_numbers=np.arange(B) % n_ep)
        env.reset()
        trace = []
        for t in range(23):
            env.step()
            trace.append((env.reward.clone(), env.done.clone(), env.views()["queue_pkts"].clone(), env.views()["episode_number"].clone()))
        torch.cuda.synchronize()
        outs.append(trace)
        env.close()
    for t, (a, b) in enumerate(zip(*outs)):
        for x, y in zip(a, b):
            assert torch.equal(x, y), t
    assert int(outs[0][-1][3].max()) > 0                                  # episodes did end


def test_a_persistent_launch_that_gave_up_is_reported_cleared_and_the_rollout_falls_back():
    """PersistCtl::abort was set by a timed-out wait and never cleared; every later persistent launch then dropped its
    envs after their first chunk while ranenv_rollout kept returning RANENV_OK.  Now the next rollout call sees the sticky error word
    (host-visible), clears the queues, switches the persistent rollout off for the handle and FAILS; after a reset the handle steps
    again -- through the launch-per-chunk rollout -- and agrees with a handle that never had the fault."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from intent_radio_sched_multi_slice_amd import _lib
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    dev = torch.device("cuda", 0)
    # (a clearing reset: the 10-TTI window does not carry b's broken rollout into the comparison)
    a = make_mult_slice_workload(600, dev, n_scenarios=16, n_traces=8, trace_len=16, max_steps=1000, flags=_lib.F_CLEAR_HISTORY_ON_RESET)
    b = make_mult_slice_workload(600, dev, n_scenarios=16, n_traces=8, trace_len=16, max_steps=1000, flags=_lib.F_CLEAR_HISTORY_ON_RESET)
    for wl in (a, b):
        wl.env.set_option("compact", 1)
    a.env.set_option("persist", 0)
    b.env.set_option("persist", 1); b.env.set_option("persist_grid", 128); b.env.set_option("persist_chunk", 2)
    b.env.reset(); b.env.rollout(9); torch.cuda.synchronize()
    assert b.env.get_option("last_rollout_persistent") == 1 and b.env.get_option("persist_errors") == 0
    b.env.set_option("persist_inject_abort", 1)
    b.env.rollout(12)                                          # this launch finds a wait given up: envs dropped after a chunk
    torch.cuda.synchronize()
    steps = b.env.views()["step_number"]
    assert int(steps.min()) < 21                               # (not every env got its 12 TTIs)
    assert b.env.get_option("persist_errors") == 1
    with pytest.raises(_lib.RanEnvError, match="gave up"):
        b.env.rollout(5)
    assert b.env.get_option("persist") == 0 and b.env.get_option("persist_errors") == 1
    a.env.reset(); b.env.reset()
    for k in (7, 20, 3):
        a.env.rollout(k); b.env.rollout(k)
        assert b.env.get_option("last_rollout_persistent") == 0
    torch.cuda.synchronize()
    va, vb = a.env.views(), b.env.views()
    for k in ("queue_pkts", "pkt_effective_thr", "dropped_pkts", "rb_count", "step_number", "win_sent"):
        assert torch.equal(va[k], vb[k]), k
    assert torch.equal(a.env.reward, b.env.reward) and torch.equal(a.env.obs_inter, b.env.obs_inter)
    a.env.close(); b.env.close()


def test_the_very_first_step_of_a_handle_can_be_captured():
    """The examination of the traffic pool for compact steps reads a flag back (a stream synchronisation): a step captured BEFORE any eager
    step must not run it inside the capture -- it steps at full width instead, with the same results."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    dev = torch.device("cuda", 0)
    outs = []
    for mode in ("eager", "graph"):
        wl = make_mult_slice_workload(200, dev, n_scenarios=8, n_traces=4, trace_len=12, max_steps=1000)
        env = wl.env
        env.reset(); torch.cuda.synchronize()
        if mode == "eager":
            for _ in range(5):
                env.step()
        else:
            g = _captured_step_graph(env, dev)
            for _ in range(5):
                g.replay()
        torch.cuda.synchronize()
        v = env.views()
        outs.append((env.obs_inter.clone(), env.reward.clone(), v["queue_pkts"].clone(), v["step_number"].clone(), v["win_sent"].clone()))
        env.close()
    assert int(outs[1][3].min()) == 5
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_graph_captured_step_with_device_autoreset_replays_through_episode_ends():
    """A captured `env.step()` with auto-reset on holds the step, the advance kernel, the masked RESET launch and the sort of the class
    lists -- the host's shortcut (nothing enqueued when its shadow of the step counters says no episode ended) must not apply inside a
    capture, and the shadow ends there: replays through several episode ends equal eager steps, and eager steps behind the replays
    carry on correctly."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    dev = torch.device("cuda", 0)
    outs = []
    for mode in ("eager", "graph"):
        wl = make_mult_slice_workload(300, dev, n_scenarios=16, n_traces=8, trace_len=16, max_steps=1000)
        env = wl.env
        env.set_option("mix", 2); env.set_option("compact", 1)
        n_ep, L = 10, wl.trace_len
        ep = np.arange(n_ep)
        env.set_episode_table(scenario=(ep * 3) % 16, se_base=(ep % 8) * L, se_len=L, se_offset=ep % L, trf_base=((ep * 3) % 16) * L, trf_len=L, trf_offset=(ep * 5) % L)
        env.set_max_steps(4 + (np.arange(env.B) % 3))                  # episodes of 4, 5, 6 TTIs
        env.enable_autoreset(0, n_ep, episode_numbers=np.arange(env.B) % n_ep)
        env.reset(); env.step(); torch.cuda.synchronize()
        if mode == "eager":
            for _ in range(17):
                env.step()
        else:
            g = _captured_step_graph(env, dev)
            for _ in range(14):
                g.replay()
            torch.cuda.synchronize()
            for _ in range(3):                                          # eager again behind the replays
                env.step()
        torch.cuda.synchronize()
        v = env.views()
        outs.append((env.obs_inter.clone(), env.reward.clone(), env.done.clone(), v["queue_pkts"].clone(), v["step_number"].clone(),
                     v["episode_number"].clone(), v["win_sent"].clone()))
        env.close()
    assert int(outs[0][5].max()) >= 3
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def _autoreset_workload(dev, B=300, n_ep=12, L=16, flags=0):
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    wl = make_mult_slice_workload(B, dev, n_scenarios=16, n_traces=8, trace_len=L, max_steps=1000, flags=flags)
    env = wl.env
    ep = np.arange(n_ep)
    env.set_episode_table(scenario=ep % 16, se_base=(ep % 8) * L, se_len=L, se_offset=ep % L, trf_base=(ep % 16) * L, trf_len=L,
                          trf_offset=(ep * 3) % L)
    env.set_max_steps(6 + (np.arange(B) % 4))
    return wl, env, n_ep


def test_autoreset_reads_the_callers_done_flags_unless_the_shortcut_was_opted_into():
    """ranenv_autoreset's contract is "every env with dev_done != 0 restarts".  The host-shadow shortcut (nothing enqueued
    when the host's copy of the step counters says no episode ended) is opt-in (option autoreset_shortcut, default 0): a caller that
    ORs its own truncation flags into the buffer between the step and the call gets those envs restarted."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctypes as C
    from intent_radio_sched_multi_slice_amd.batched_env import _ptr
    dev = torch.device("cuda", 0)
    wl, env, n_ep = _autoreset_workload(dev)
    B = env.B
    assert env.get_option("autoreset_shortcut") == 0                    # the library's default
    env.enable_autoreset(0, n_ep, episode_numbers=np.arange(B) % n_ep, shortcut=False)
    assert env.get_option("autoreset_shortcut") == 0
    env._autoreset = False                                              # this test calls ranenv_autoreset itself
    env.reset()
    for _ in range(3):
        env.step()
    torch.cuda.synchronize()
    assert int(env.done.sum()) == 0                                     # no episode is over yet (lengths 6..9)
    ep_before = env.views()["episode_number"].clone()
    mine = torch.tensor([1, 7, 42, 299], device=dev)
    env.done[mine] = 1                                                  # the caller's own truncation

    def autoreset():
        st = env._lib.ranenv_autoreset(env._h, _ptr(env.done), _ptr(env.obs_inter), _ptr(env.obs_intra), _ptr(env.term_obs_inter),
                                       _ptr(env.term_obs_intra), None, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        assert st == 0
    autoreset()
    torch.cuda.synchronize()
    v = env.views()
    restarted = torch.zeros(B, dtype=torch.bool, device=dev); restarted[mine] = True
    assert torch.equal(v["step_number"][restarted], torch.zeros(4, dtype=torch.int32, device=dev))
    assert torch.equal(v["step_number"][~restarted], torch.full((B - 4,), 3, dtype=torch.int32, device=dev))
    assert bool((v["episode_number"][restarted] != ep_before[restarted]).all())
    assert torch.equal(v["episode_number"][~restarted], ep_before[~restarted])
    # opted in, the same call trusts the host's counters: flags the caller added are NOT looked at (documented)
    env.reset()                                                         # (a reset of the whole batch restarts the shadow)
    env.set_option("autoreset_shortcut", 1)
    for _ in range(3):
        env.step()
    env.done[mine] = 1
    autoreset()
    torch.cuda.synchronize()
    assert int(env.views()["step_number"].min()) == 3
    env.close()


def test_masked_reset_after_a_persistent_abort_does_not_leave_a_stale_step_shadow():
    """persist_check_errors switched the persistent rollout off but left the host's shadow of the step counters
    valid although the envs had advanced different numbers of TTIs.  A caller that answers the error with a MASKED reset (or none) and
    carries on with env.step() + ranenv_autoreset (shortcut on) must get resets at the TTIs the DEVICE's counters say -- the same as a
    handle that reached the same device state without the fault."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from intent_radio_sched_multi_slice_amd import _lib
    dev = torch.device("cuda", 0)
    wa, a, n_ep = _autoreset_workload(dev, flags=_lib.F_CLEAR_HISTORY_ON_RESET)
    wb, b, _ = _autoreset_workload(dev, flags=_lib.F_CLEAR_HISTORY_ON_RESET)
    B = a.B
    for env in (a, b):
        env.set_option("compact", 1)
        env.set_max_steps(30 + (np.arange(B) % 4))                          # (no episode ends inside the broken rollout)
        env.enable_autoreset(0, n_ep, episode_numbers=np.arange(B) % n_ep)  # shortcut on (the wrapper's default)
    b.set_option("persist", 1); b.set_option("persist_grid", 128); b.set_option("persist_chunk", 2)
    b.reset(); b.rollout(4); torch.cuda.synchronize()
    assert b.get_option("last_rollout_persistent") == 1
    b.set_option("persist_inject_abort", 1)
    b.rollout(12); torch.cuda.synchronize()                                 # envs dropped after their first chunk: uneven step counters
    with pytest.raises(_lib.RanEnvError, match="gave up"):
        b.rollout(1)
    steps_b = b.views()["step_number"].clone()
    assert int(steps_b.min()) < int(steps_b.max())
    # the caller's answer: reset only the envs that fell behind the most (a MASKED reset), keep the others where they are
    mask = (steps_b == steps_b.min()).to(torch.uint8)
    b.reset(env_mask=mask)
    torch.cuda.synchronize()
    steps_b = b.views()["step_number"].clone()
    # handle a reaches the same device state honestly: full reset, then every env stepped to b's counter with a masked ... there is no
    # masked step, so compare the BEHAVIOUR instead: from here on b must restart env e exactly when its device counter reaches its length
    lens = torch.as_tensor(30 + (np.arange(B) % 4), device=dev, dtype=torch.int32)
    ep0 = b.views()["episode_number"].clone()
    n_resets = torch.zeros(B, dtype=torch.int32, device=dev)
    cur = steps_b.clone()
    for t in range(40):
        b.step()
        cur += 1
        due = cur >= lens
        torch.cuda.synchronize()
        assert torch.equal(b.done.to(torch.bool), due), t                   # done as the device's counters say ...
        cur[due] = 0
        n_resets += due.to(torch.int32)
        assert torch.equal(b.views()["step_number"], cur), t                # ... and those envs, only those, were restarted
    assert int(n_resets.max()) >= 1 and int(n_resets.min()) >= 0
    moved = b.views()["episode_number"] != ep0
    assert torch.equal(moved, n_resets > 0) or n_ep == 1
    a.close(); b.close()


def test_guard_free_division_equals_ieee_division_on_the_kernels_operand_domain():
    """ddiv() -- the compiler's f64 division without v_div_scale / v_div_fixup -- must give the bits of the plain operator wherever the
    kernels use it: divisors that are positive normal numbers (packet sizes, UE / RB / slice counts, window lengths, 1e6, sums of weights, the
    validated normalisers in [1e-30, 1e30]), dividends that are finite and >= 0.  Checked inside the shipped build (ranenv_selftest_ddiv) on
    random and edge operands; explicit traffic -- the one caller-supplied dividend that may be inf / huge -- does not go through ddiv at all."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctypes as C
    from intent_radio_sched_multi_slice_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    n = 1 << 18
    a = np.concatenate([
        rng.integers(0, 2 ** 31, n).astype(np.float64),                          # packet / bit counts
        rng.random(n) * 10.0 ** rng.integers(-9, 13, n),                         # sums, rates, weights: 1e-9 ... 1e12
        np.array([0.0, 1.0, 2.0 ** -900, 2.0 ** 900, 1e-250, 1e250, 135.0, 1e8, 1e6, 2.0 ** -899, np.nextafter(1.0, 2.0), 2.0 ** 52 + 1.0]),
    ])
    b = np.concatenate([
        rng.integers(1, 2 ** 20, n).astype(np.float64),                          # integer divisors: sizes, counts
        rng.random(n) * 10.0 ** rng.integers(-9, 13, n) + 1e-12,
        np.array([1.0, 3.0, 1e6, 135.0, 7.0, 1e-30, 1e30, 2.0 ** -100, 2.0 ** 100, 640.0, np.nextafter(1.0, 0.0), 3.0]),
    ])
    # (the domain: dividend 0 or in [2^-900, 2^900] -- below that the residual of the correction step is a denormal and the last bit can differ,
    # which is what v_div_scale is for; the kernels' dividends are counts and sums of 1e-9 ... 1e12 -- and a quotient in the same range)
    # every pairing of the edge dividends with the edge divisors whose quotient stays in that range
    ea, eb = np.meshgrid(a[-12:], b[-12:])
    ea, eb = ea.ravel(), eb.ravel()
    with np.errstate(over="ignore", under="ignore"):
        q = ea / eb
    ok = (q == 0.0) | ((np.abs(q) >= 2.0 ** -900) & (np.abs(q) <= 2.0 ** 900))
    a, b = np.concatenate([a[:-12], ea[ok]]), np.concatenate([b[:-12], eb[ok]])
    ta, tb = torch.as_tensor(a, device=dev), torch.as_tensor(b, device=dev)
    fast, ieee = torch.empty_like(ta), torch.empty_like(ta)
    st = lib.ranenv_selftest_ddiv(C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr()), C.c_void_p(fast.data_ptr()), C.c_void_p(ieee.data_ptr()),
                                  ta.numel(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert st == 0
    torch.cuda.synchronize()
    f, i = fast.cpu().numpy(), ieee.cpu().numpy()
    with np.errstate(over="ignore", under="ignore"):
        assert np.array_equal(i.view(np.uint64), (a / b).view(np.uint64))      # the device's IEEE division is numpy's
    bad = np.flatnonzero(f.view(np.uint64) != i.view(np.uint64))
    assert bad.size == 0, [(a[k], b[k], f[k], i[k]) for k in bad[:5]]
