"""On-device episode advance + auto-reset (ranenv_autoreset), per-env episode lengths, the counter-based traffic
generator (ranenv_set_traffic_generator) and the batched RLlib-shaped view, against the oracle / numpy restatements."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

OBS_TOL, REW_TOL = 1e-5, 1e-9


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _small_workload(B, steps, **kw):
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    return make_mult_slice_workload(B, torch.device("cuda", 0), policy=2, intra=1, n_scenarios=6, n_traces=12, trace_len=10,
                                    n_slices=5, n_ues=25, n_rbs=135, rbs_per_rbg=5, max_ues_slice=10, max_steps=steps, **kw)


@pytest.mark.parametrize("random_episodes", [False, True])
def test_three_consecutive_episodes_with_device_autoreset_vs_oracle(random_episodes):
    """B = 64, episode lengths staggered per env (30..50 TTIs), three episodes each, no host-side reset: at `done` the
    device picks the next episode (sequential / counter-based random), installs its descriptor and resets.  The oracle
    side replays the same rule on the host."""
    _need_gpu()
    from oracle import pyoracle
    B, L, n_ep, first = 64, 10, 12, 3
    wl = _small_workload(B, 50)
    env, tabs = wl.env, wl.tables
    S, U, R = env.S, env.U, env.R
    # episode table: episode n -> scenario n % 6 (associations/mult_slice.py:444-452 with 6 scenarios), its own channel trace
    ep_no = np.arange(first, first + n_ep)
    env.set_episode_table(scenario=ep_no % 6, se_base=(ep_no % 12) * L, se_len=L, se_offset=ep_no % L,
                          trf_base=(ep_no % 6) * L, trf_len=L, trf_offset=(ep_no * 3) % L, first_episode=first)
    max_steps = 30 + np.arange(B) % 21
    env.set_max_steps(max_steps)
    start = first + np.arange(B) % n_ep
    seed = 99
    env.enable_autoreset(first, first + n_ep, random_episodes=random_episodes, seed=seed, episode_numbers=start)
    cfg = pyoracle.make_cfg(S, U, R, env.G, env.Us, max_steps=10 ** 6)
    se_host = wl.se_pool.transpose(1, 2).contiguous().cpu().numpy()
    trf_host = wl.traffic_pool.cpu().numpy().astype(np.float64)
    tab = env.episode_table
    intra = np.full(S, 1, dtype=np.int32)

    def tile(ep, t): r = tab[ep - first]; return int(r["se_base"] + (r["se_offset"] + t) % r["se_len"])
    def trow(ep, t): r = tab[ep - first]; return int(r["trf_base"] + (r["trf_offset"] + t) % r["trf_len"])

    oenvs, cur, tstep, nreset = [], start.copy(), np.zeros(B, dtype=int), np.zeros(B, dtype=int)
    for b in range(B):
        o = pyoracle.OracleEnv(cfg); o.set_scenario(tabs, int(tab[cur[b] - first]["scenario"])); o.reset(se_host[tile(cur[b], 0)])
        oenvs.append(o)
    env.reset()
    episodes_done = np.zeros(B, dtype=int)
    for it in range(3 * 50 + 5):
        obs, rew, done = env.step()
        g = {k: x.cpu().numpy() for k, x in env.views().items()}
        oi, oa, rw, dn = obs["obs_inter"].cpu().numpy(), obs["obs_intra"].cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
        ti, ta = env.term_obs_inter.cpu().numpy(), env.term_obs_intra.cpu().numpy()
        for b, o in enumerate(oenvs):
            o.step(o.policy_mapf(), intra, se_host[tile(cur[b], tstep[b])], trf_host[trow(cur[b], tstep[b])])
            tstep[b] += 1
            oo = o.obs()
            np.testing.assert_allclose(rw[b], oo["reward"], rtol=0, atol=REW_TOL)
            is_done = tstep[b] >= max_steps[b]
            assert bool(dn[b]) == is_done, (it, b)
            if not is_done:
                raw = o.raw()
                for name in ("pkt_incoming", "pkt_effective_thr", "dropped_pkts"):
                    assert np.array_equal(g[name][b].astype(np.float64), raw[name]), (it, b, name)
                np.testing.assert_allclose(oi[b], oo["obs_inter"], rtol=0, atol=OBS_TOL)
                np.testing.assert_allclose(oa[b], oo["obs_intra"], rtol=0, atol=OBS_TOL)
                continue
            # terminal observation kept, next episode chosen by the reference's rule, env reset: all on the device
            np.testing.assert_allclose(ti[b], oo["obs_inter"], rtol=0, atol=OBS_TOL)
            np.testing.assert_allclose(ta[b], oo["obs_intra"], rtol=0, atol=OBS_TOL)
            nreset[b] += 1
            if random_episodes:
                x = pyoracle.philox4x32_10(b, nreset[b], 0x45504953, 0, seed & 0xFFFFFFFF, seed >> 32)
                nxt = first + int(x[0]) % n_ep
            else:
                nxt = cur[b] + 1 if cur[b] + 1 < first + n_ep else first
            cur[b], tstep[b] = nxt, 0
            episodes_done[b] += 1
            assert int(g["episode_number"][b]) == nxt and int(g["episodes"][b, 0]) == int(tab[nxt - first]["scenario"])
            o.set_scenario(tabs, int(tab[nxt - first]["scenario"]))
            o.reset(se_host[tile(nxt, 0)])
            ro = o.obs()
            np.testing.assert_allclose(oi[b], ro["obs_inter"], rtol=0, atol=OBS_TOL)         # first observation of the new episode
            np.testing.assert_allclose(oa[b], ro["obs_intra"], rtol=0, atol=OBS_TOL)
            assert int(g["step_number"][b]) == 0 and int(g["queue_pkts"][b].sum()) == 0
        if episodes_done.min() >= 3:
            break
    assert episodes_done.min() >= 3
    if random_episodes:
        assert len(set(cur.tolist())) > 3
    env.close()


def test_traffic_generator_tables_and_draws_match_the_numpy_restatement():
    """The device draws equal the numpy restatement (Philox KAT-checked, same inversion tables) bit for bit; the tables
    are Poisson CDFs (against scipy.stats) and the draws have Poisson mean and variance."""
    _need_gpu()
    from scipy import stats
    from oracle import pyoracle
    B, steps, seed, base = 48, 40, 0x1234567890ABCDEF, 1000
    wl = _small_workload(B, steps)
    env, tabs = wl.env, wl.tables
    env.set_traffic_generator(seed, env_id_base=base)
    cdf, guide = env.poisson_tables()
    for i in range(tabs.n_scenarios):
        for s in range(tabs.n_slices):
            if not tabs.slice_has_req[i, s]:
                continue
            lam = float(tabs.slice_traffic[i, s])
            c = cdf[i, s].astype(np.float64) / 2.0 ** 64
            assert np.all(np.diff(cdf[i, s].astype(object)) >= 0)
            np.testing.assert_allclose(c, np.minimum(stats.poisson.cdf(np.arange(256), lam), 1.0), rtol=0, atol=2e-15)
            k = guide[i, s].astype(int)
            lo = [(j << 58) for j in range(64)]
            assert all(int(cdf[i, s, k[j]]) > lo[j] and (k[j] == 0 or int(cdf[i, s, k[j] - 1]) <= lo[j]) for j in range(64))
    env.reset()
    v = env.views()
    pk = np.asarray(tabs.ue_pkt_size)[wl.scenario].astype(np.float64)
    draws = []
    for t in range(steps):
        env.step()
        inc = v["pkt_incoming"].cpu().numpy().astype(np.float64)
        for b in range(B):
            bits = pyoracle.generator_traffic(cdf, tabs, int(wl.scenario[b]), seed, base + b, 0, t)
            assert np.array_equal(inc[b], np.floor(bits / pk[b])), (t, b)
        draws.append(np.ceil(inc * pk / 1e6))                   # k = ceil(pkt_in * pkt_size / 1e6): exact since pkt_size < 1e6
    draws = np.stack(draws)                                     # [steps, B, U]
    for s in range(tabs.n_slices):
        sel = [(b, u) for b in range(B) for u in tabs.slice_ues[wl.scenario[b], s, :tabs.slice_nues[wl.scenario[b], s]]
               if tabs.slice_traffic[wl.scenario[b], s] == tabs.slice_traffic[wl.scenario[0], s] and tabs.slice_has_req[wl.scenario[b], s]]
    # pooled statistics per distinct lambda
    lam_of = np.zeros((B, tabs.n_ues))
    for b in range(B):
        sc = wl.scenario[b]
        for s in range(tabs.n_slices):
            if tabs.slice_has_req[sc, s]:
                lam_of[b, tabs.slice_ues[sc, s, :tabs.slice_nues[sc, s]]] = tabs.slice_traffic[sc, s]
    assert np.all(draws[:, lam_of == 0] == 0)
    for lam in np.unique(lam_of[lam_of > 0]):
        x = draws[:, lam_of == lam].ravel()
        n = x.size
        assert abs(x.mean() - lam) < 5 * np.sqrt(lam / n), (lam, x.mean(), n)
        assert abs(x.var() - lam) < 6 * lam * np.sqrt(2.0 / n) + 0.05 * lam, (lam, x.var(), n)
    env.close()


def test_generated_traffic_is_deterministic_and_independent_of_the_actions():
    """results/gen_results.py:1587-1635: pkt_incoming must be identical across agents for the same (seed, episode).
    Same seed, different policies -> identical offered traffic; another seed -> different; a re-run -> identical."""
    _need_gpu()
    runs = {}
    for name, policy, intra, seed in (("mapf", 2, 1, 7), ("marr", 1, 0, 7), ("mapf_again", 2, 1, 7), ("other_seed", 2, 1, 8)):
        wl = _small_workload(32, 25)
        env = wl.env
        env.set_policy(policy, intra)
        env.set_traffic_generator(seed)
        env.reset()
        inc, sent = [], []
        for t in range(25):
            env.step()
            inc.append(env.views()["pkt_incoming"].clone()); sent.append(env.views()["pkt_effective_thr"].clone())
        runs[name] = (torch.stack(inc), torch.stack(sent))
        env.close()
    assert torch.equal(runs["mapf"][0], runs["marr"][0])                   # exogenous: same under another agent
    assert not torch.equal(runs["mapf"][1], runs["marr"][1])              # ... although the agents behave differently
    assert torch.equal(runs["mapf"][0], runs["mapf_again"][0]) and torch.equal(runs["mapf"][1], runs["mapf_again"][1])
    assert not torch.equal(runs["mapf"][0], runs["other_seed"][0])


def test_marl_batch_env_is_the_reference_layout_on_device_tensors():
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.adapters import MarlBatchEnv, masked_gaussian_params, sorted_action_mask
    from oracle import pyoracle
    B, steps = 16, 12
    wl = _small_workload(B, steps)
    env, tabs = wl.env, wl.tables
    S, U, R, Us = env.S, env.U, env.R, env.Us
    menv = MarlBatchEnv(env)
    obs, _ = menv.reset()
    assert set(obs) == {f"player_{i}" for i in range(S + 1)}
    assert obs["player_0"]["observations"].shape == (B, 10 * S) and obs["player_0"]["action_mask"].shape == (B, S)
    assert obs["player_2"]["observations"].shape == (B, 2 * Us + 9) and obs["player_2"]["action_mask"].shape == (B, Us)
    assert obs["player_0"]["observations"].is_cuda and obs["player_0"]["action_mask"].dtype == torch.int8
    cfg = pyoracle.make_cfg(S, U, R, env.G, Us, max_steps=steps)
    se_host = wl.se_pool.transpose(1, 2).contiguous().cpu().numpy()
    trf_host = wl.traffic_pool.cpu().numpy().astype(np.float64)
    eps, L = env.episodes, wl.trace_len
    oenvs = []
    for b in range(B):
        o = pyoracle.OracleEnv(cfg); o.set_scenario(tabs, int(wl.scenario[b]))
        o.reset(se_host[int(eps["se_base"][b] + eps["se_offset"][b] % L)]); oenvs.append(o)
    g = torch.Generator(device="cpu"); g.manual_seed(3)
    for t in range(steps):
        mask = sorted_action_mask(obs["player_0"]["action_mask"])
        assert torch.equal(mask.sum(1), obs["player_0"]["action_mask"].sum(1))
        mean, std = masked_gaussian_params(torch.zeros(B, S, device=env.device), torch.full((B, S), -1.0, device=env.device), mask)
        scores = (mean + std * torch.randn(B, S, generator=g).to(env.device)).clamp(-1, 1).to(torch.float64)
        assert torch.all(scores[mask == 0] == -1.0)                         # inactive positions pinned at -1 (:33-35)
        action = {"player_0": scores}
        action.update({f"player_{s + 1}": torch.randint(0, 3, (B,), generator=g) for s in range(S)})
        obs, rew, term, trunc, info = menv.step(action)
        assert set(rew) == set(obs) and set(term) == set(obs) | {"__all__"}
        sc, ic = scores.cpu().numpy(), np.stack([action[f"player_{s + 1}"].numpy() for s in range(S)], axis=1)
        for b, o in enumerate(oenvs):
            tile = int(eps["se_base"][b] + (eps["se_offset"][b] + t) % L)
            row = int(eps["trf_base"][b] + (eps["trf_offset"][b] + t) % L)
            o.step(sc[b], ic[b].astype(np.int32), se_host[tile], trf_host[row])
            oo = o.obs()
            np.testing.assert_allclose(obs["player_0"]["observations"][b].cpu().numpy(), oo["obs_inter"], rtol=0, atol=OBS_TOL)
            assert np.array_equal(obs["player_0"]["action_mask"][b].cpu().numpy(), oo["mask_inter"])
            for s in range(S):
                np.testing.assert_allclose(obs[f"player_{s + 1}"]["observations"][b].cpu().numpy(), oo["obs_intra"][s], rtol=0, atol=OBS_TOL)
                assert np.array_equal(obs[f"player_{s + 1}"]["action_mask"][b].cpu().numpy(), oo["mask_intra"][s])
            np.testing.assert_allclose([float(rew[f"player_{i}"][b]) for i in range(S + 1)], oo["reward"], rtol=0, atol=REW_TOL)
        assert bool(term["__all__"].all()) == (t == steps - 1)
    env.close()


def test_head_vec_env_with_device_autoreset_costs_one_copy_per_step():
    """HeadVecEnv over an env with an episode table: finished envs restart on the device, infos carry the terminal
    observation, the next observation is the new episode's first one."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.adapters import HeadVecEnv
    B, steps = 8, 6
    wl = _small_workload(B, steps)
    env = wl.env
    ep_no = np.arange(0, 6)
    env.set_episode_table(scenario=ep_no % 6, se_base=ep_no * 10, se_len=10, trf_base=(ep_no % 6) * 10, trf_len=10)
    venv = HeadVecEnv(env, reward="twc")
    venv.enable_device_autoreset(0, 6, episode_numbers=np.arange(B) % 6)
    obs = venv.reset()
    assert obs.shape == (B, 10 * env.S)
    for t in range(2 * steps):
        obs, rew, dones, infos = venv.step(np.zeros((B, env.S)))
        assert obs.shape == (B, 10 * env.S) and rew.shape == (B,) and dones.dtype == bool
        if (t + 1) % steps == 0:
            assert dones.all() and all("terminal_observation" in i for i in infos)
            assert not np.array_equal(infos[0]["terminal_observation"], obs[0])
            assert int(env.views()["step_number"].max()) == 0
        else:
            assert not dones.any() and all(i == {} for i in infos)
    assert env.views()["episode_number"].cpu().numpy().tolist() == [(b % 6 + 2) % 6 for b in range(B)]
    venv.close()


def test_autoreset_enqueues_nothing_when_no_episode_ended_and_the_same_as_ever_when_one_did():
    """`done` is a function of the step counter alone, and the host follows the counters (include/ranenv.h, ranenv_autoreset): behind
    a TTI at which no episode ended the call enqueues nothing; with flags the host cannot follow (another buffer than the steps'
    `done`) it asks the device as before.  Same state either way, and the launch counts say which path ran."""
    _need_gpu()
    import ctypes as C
    B, steps = 64, 26
    outs, launches = [], []
    for follow in (True, False):
        wl = _small_workload(B, steps)
        env, tabs = wl.env, wl.tables
        n_ep = 12
        ep = np.arange(n_ep)
        env.set_episode_table(scenario=(ep * 5) % tabs.n_scenarios, se_base=(ep % 4) * wl.trace_len, se_len=wl.trace_len, se_offset=ep % wl.trace_len,
                              trf_base=((ep * 5) % tabs.n_scenarios) * wl.trace_len, trf_len=wl.trace_len, trf_offset=(ep * 3) % wl.trace_len)
        env.set_max_steps(np.where(np.arange(B) % 2 == 0, 9, 13).astype(np.int32))       # episodes end at TTIs 9, 13, 18, 26
        env.enable_autoreset(0, n_ep, episode_numbers=np.arange(B) % n_ep)
        if not follow:
            env._autoreset = False                                                     # (the test calls ranenv_autoreset itself, with a COPY of done)
        env.reset()
        env.profile_begin()
        for t in range(steps):
            env.step()
            if not follow:
                mine = env.done.clone()
                st = env._lib.ranenv_autoreset(env._h, C.c_void_p(mine.data_ptr()), C.c_void_p(env.obs_inter.data_ptr()), C.c_void_p(env.obs_intra.data_ptr()),
                                               C.c_void_p(env.term_obs_inter.data_ptr()), C.c_void_p(env.term_obs_intra.data_ptr()), None,
                                               C.c_void_p(torch.cuda.current_stream(env.device).cuda_stream))
                assert st == 0
        kms = env.profile_end()
        launches.append(kms["n_launches"])
        v = env.views()
        outs.append({k: v[k].clone() for k in ("queue_pkts", "step_number", "episode_number", "win_sent", "pkt_effective_thr")} | {"obs": env.obs_inter.clone(), "rew": env.reward.clone()})
        env.close()
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert int(outs[0]["episode_number"].max()) >= 2
    assert launches[1] == 2 * steps                       # a step + a masked RESET launch every TTI
    assert launches[0] == steps + 4, launches             # a RESET launch only behind TTIs 9, 13, 18 and 26


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_autoreset_shadow_follows_random_call_sequences(seed):
    """The host's copy of the step counters (what lets ranenv_autoreset enqueue nothing behind a TTI at which no episode ended) against the
    device's own `done` flags over random sequences of everything that moves or invalidates the counters: whole-batch steps, ranges
    stepped different numbers of TTIs (ranenv_step_part / ranenv_autoreset_part), rollouts (which follow the episode ends
    themselves and hand the counters back), full resets, masked resets and new per-env episode lengths (after which the host
    cannot know the counters until the next full reset or rollout).  Handle A uses the library as a trainer would; handle B hands
    every auto-reset a COPY of `done`, which the host cannot follow, so the device decides.  Same state after every call."""
    _need_gpu()
    import ctypes as C
    B, n_ep, n_ranges = 48, 12, 3
    rng = np.random.default_rng(seed)
    plan = []
    for _ in range(110):
        r = rng.random()
        if r < 0.45: plan.append(("step",))
        elif r < 0.70: plan.append(("range", int(rng.integers(0, n_ranges)), int(rng.integers(1, 4))))
        elif r < 0.80: plan.append(("rollout", int(rng.integers(1, 9))))
        elif r < 0.87: plan.append(("reset",))
        elif r < 0.93: plan.append(("masked_reset", rng.integers(0, 2, B).astype(np.uint8)))
        else: plan.append(("max_steps", rng.integers(3, 10, B).astype(np.int32)))
    envs = []
    for follow in (True, False):
        wl = _small_workload(B, 7)
        env, tabs = wl.env, wl.tables
        ep = np.arange(n_ep)
        env.set_episode_table(scenario=(ep * 5) % tabs.n_scenarios, se_base=(ep % 4) * wl.trace_len, se_len=wl.trace_len, se_offset=ep % wl.trace_len,
                              trf_base=((ep * 5) % tabs.n_scenarios) * wl.trace_len, trf_len=wl.trace_len, trf_offset=(ep * 3) % wl.trace_len)
        env.set_max_steps(np.where(np.arange(B) % 2 == 0, 5, 8).astype(np.int32))
        env.enable_autoreset(0, n_ep, episode_numbers=np.arange(B) % n_ep)
        env.set_ranges(n_ranges)
        if not follow:
            env._autoreset = False              # (handle B calls ranenv_autoreset / _part itself, with a copy of `done`)
        env.reset()
        env.profile_begin()
        envs.append(env)

    def outs(env):
        return (C.c_void_p(env.obs_inter.data_ptr()), C.c_void_p(env.obs_intra.data_ptr()), C.c_void_p(env.term_obs_inter.data_ptr()),
                C.c_void_p(env.term_obs_intra.data_ptr()), None, C.c_void_p(torch.cuda.current_stream(env.device).cuda_stream))

    for i, op in enumerate(plan):
        for env, follow in zip(envs, (True, False)):
            if op[0] == "step":
                env.step()
                if not follow:
                    mine = env.done.clone()
                    assert env._lib.ranenv_autoreset(env._h, C.c_void_p(mine.data_ptr()), *outs(env)) == 0
            elif op[0] == "range":
                for _ in range(op[2]):
                    env.step_async(op[1])
                    env.step_wait(op[1])
                    if not follow:
                        mine = env.done.clone()
                        assert env._lib.ranenv_autoreset_part(env._h, op[1], C.c_void_p(mine.data_ptr()), *outs(env)) == 0
                        env.step_wait(op[1])
            elif op[0] == "rollout":
                env._autoreset, keep = True, env._autoreset        # (a rollout with auto-reset follows the episode ends itself, on either handle)
                env.rollout(op[1])
                env._autoreset = keep
            elif op[0] == "reset":
                env.reset()
            elif op[0] == "masked_reset":
                env.reset(env_mask=op[1])
            else:
                env.set_max_steps(op[1])
        torch.cuda.synchronize()
        va, vb = envs[0].views(), envs[1].views()
        for k in ("step_number", "episode_number", "queue_pkts", "win_sent"):
            assert torch.equal(va[k], vb[k]), (seed, i, op[0], k)
        assert torch.equal(envs[0].obs_inter, envs[1].obs_inter), (seed, i, op[0])
    assert int(envs[0].views()["episode_number"].max()) >= 3
    launches = [env.profile_end()["n_launches"] for env in envs]
    assert launches[0] < launches[1] - 20, launches          # handle A really skipped the auto-reset launches where it could
    for env in envs:
        env.close()
