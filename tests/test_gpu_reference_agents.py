"""tests/golden/agents_on_facade.npz replayed through the real facade on the GPU.

The fixture was produced in the build container by the reference's REAL classes -- agents/ib_sched.py IBSched,
agents/marr.py MARR, agents/mapf.py MAPF, associations/mult_slice.py, traffics/mult_slice.py, channels/mimic_quadriga.py,
mobilities/simple.py -- attached to comm_env.MARLCommEnv by env_creator's sequence (simu.py:341-424) with a CPU stand-in
under the facade (tests/golden/gen_golden_agents.py).  Here the same facade runs on the HIP env step with this build's
plugins (same seed: the four plugins share the env's one rng, so traffic and channel only agree if every plugin draws what
and when the reference's does) and the fixture's actions; an agent that speaks the reference's protocol with the oracle's
agent-side arithmetic formats observations.  Integers (RB ranges, traffic, packets in / out / dropped) must match exactly,
dict observations at 1e-5, rewards at 1e-9."""
import json

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from tests.common import load_golden
from tests.test_gpu_protocol_agent import OracleIBSched

pytestmark = pytest.mark.gpu

OBS_TOL, REW_TOL = 1e-5, 1e-9


class _Agent(OracleIBSched):
    """IBSched's protocol; ``sort`` = enable_sort_slices (MARR / MAPF wrap an IBSched built with it off, agents/marr.py:30-37)."""

    def __init__(self, *a, sort=True, **k):
        super().__init__(*a, **k)
        self._sort = sort

    def _sync_scenario(self, raw):
        from intent_radio_sched_multi_slice_amd.scenario import ScenarioTables
        ues = self.env.comm_env.ues
        key = (raw["slice_ue_assoc"].tobytes(), ues.pkt_sizes.tobytes(), ues.max_buffer_pkts.tobytes())
        if key != self._scenario_key:
            t = ScenarioTables.empty(1, self.S, self.U, self.Us)
            t.set_from_reference(0, raw["basestation_slice_assoc"], raw["slice_ue_assoc"], raw["slice_req"], self._sort,
                                 (ues.pkt_sizes, ues.max_buffer_pkts, np.array([b.max_packets_age for b in ues.buffers])))
            self.tables = t
            self.orc.set_scenario(t, 0)
            self._scenario_key = key


@pytest.mark.parametrize("name", ["ib_sched", "marr", "mapf"])
def test_fixture_of_the_real_reference_agents_replays_on_the_gpu_facade(name, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    replay_fixture(name, tmp_path)


def replay_fixture(name, tmp_path):
    """(also run by tests/test_reference_agents_cpu.py with the CPU stand-in under the facade)"""
    from intent_radio_sched_multi_slice_amd import plugins
    from intent_radio_sched_multi_slice_amd.comm_env import DEFAULT_CONFIGS, MARLCommEnv
    fx = load_golden("agents_on_facade")
    S, U, R, G, Us, seed, steps = (int(x) for x in fx["cfg"])
    cfg = dict(DEFAULT_CONFIGS["mult_slice"], max_number_steps=steps)
    env = MARLCommEnv(plugins.MimicQuadriga, plugins.MultSliceTraffic, plugins.SimpleMobility, plugins.MultSliceAssociation,
                      "mult_slice", name, seed, root_path=str(tmp_path), config=cfg, max_episode_number=2, max_ues_slice=Us)
    ce = env.comm_env
    marl = name == "ib_sched"
    agent = _Agent(env, ce.max_number_ues, ce.max_number_slices, ce.max_number_basestations, ce.num_available_rbs,
                   max_ues_slice=Us, rbs_per_rbg=G, sort=marl)
    env.set_agent_functions(agent.obs_space_format, agent.action_format, agent.calculate_reward, None, None)
    fixed = {"ib_sched": None, "marr": 0, "mapf": 1}[name]      # MARR: fixed_intra "rr" (marr.py:62-70), MAPF: "pf" (mapf.py:126-134)

    def flat(o):
        if marl:
            return (np.concatenate([o["player_0"]["observations"]] + [o[f"player_{s + 1}"]["observations"] for s in range(S)]),
                    np.concatenate([o["player_0"]["action_mask"]] + [o[f"player_{s + 1}"]["action_mask"] for s in range(S)]))
        return np.asarray(o["player_0"]["observations"]), None

    obs, _ = env.reset(seed=seed, options={"initial_episode": 0})
    assert np.array_equal(ce.slice_ue_assoc, fx[f"{name}_slice_ue_assoc"])
    assert {k: (v["name"] if v else None) for k, v in ce.slice_req.items()} == json.loads(str(fx[f"{name}_slice_names"]))
    assert np.array_equal(np.stack([ce.ues.pkt_sizes, ce.ues.max_buffer_pkts, ce.ues.max_buffer_latencies]), fx[f"{name}_ues"])
    o, m = flat(obs)
    np.testing.assert_allclose(o, fx[f"{name}_reset_obs"], rtol=0, atol=OBS_TOL)
    if marl:
        assert np.array_equal(m, fx[f"{name}_reset_mask"])
    for t in range(steps):
        a = fx[f"{name}_action"][t]
        action = {"player_0": a[:S].copy()}
        action.update({f"player_{s + 1}": int(a[S + s]) if marl else fixed for s in range(S)})
        obs, reward, term, trunc, info = env.step(action)
        raw = env._last_raw
        sched = np.asarray(raw["sched_decision"])[0]
        cnt = sched.sum(axis=1).astype(np.int32)
        assert np.array_equal(cnt, fx[f"{name}_rb_count"][t]), (name, t)
        st = np.array([int(np.nonzero(sched[u])[0][0]) if cnt[u] else 0 for u in range(U)])
        assert np.array_equal(st, fx[f"{name}_rb_start"][t]), (name, t)
        assert np.array_equal(env._last_traffic, fx[f"{name}_traffic"][t]), (name, t)
        np.testing.assert_array_equal(np.asarray(raw["spectral_efficiencies"])[0].sum(axis=1), fx[f"{name}_se_sum"][t])
        for k in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts", "buffer_occupancies", "buffer_latencies"):
            assert np.array_equal(raw[k], fx[f"{name}_{k}"][t]), (name, t, k)
        o, m = flat(obs)
        np.testing.assert_allclose(o, fx[f"{name}_obs"][t], rtol=0, atol=OBS_TOL, err_msg=str((name, t)))
        if marl:
            assert np.array_equal(m, fx[f"{name}_mask"][t])
            rw = np.array([reward[f"player_{i}"] for i in range(S + 1)])
        else:
            rw = np.array([reward["player_0"]])
        np.testing.assert_allclose(rw, fx[f"{name}_reward"][t], rtol=0, atol=REW_TOL, err_msg=str((name, t)))
    assert term["__all__"]
    env.close()


def test_batched_marl_view_carries_the_reference_spaces():
    """adapters.MarlBatchEnv.observation_space / action_space against IBSched.get_obs_space / get_action_space as the
    reference's class returned them (agents/ib_sched.py:394-470)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from intent_radio_sched_multi_slice_amd.adapters import MarlBatchEnv, describe_space
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    fx = load_golden("agents_on_facade")
    want = json.loads(str(fx["ib_sched_spaces"]))
    wl = make_mult_slice_workload(4, torch.device("cuda", 0), n_scenarios=4, n_traces=2, trace_len=4, n_slices=5, n_ues=25,
                                  n_rbs=135, rbs_per_rbg=5, max_ues_slice=5, max_steps=4)
    menv = MarlBatchEnv(wl.env)
    for p in want["obs"]:                 # the reference declares float64 observations; the device hands out float32
        assert want["obs"][p]["observations"]["dtype"] == "float64"
        want["obs"][p]["observations"]["dtype"] = "float32"
    assert describe_space(menv.observation_space) == want["obs"]
    assert describe_space(menv.action_space) == want["action"]
    obs, _ = menv.reset()
    for p, sp in menv.observation_space.spaces.items():
        assert tuple(obs[p]["observations"].shape[1:]) == tuple(sp.spaces["observations"].shape)
        assert tuple(obs[p]["action_mask"].shape[1:]) == tuple(sp.spaces["action_mask"].shape)
    wl.env.close()


@pytest.mark.parametrize("name,col", [("sched_twc", 0), ("sched_colran", 1)])
def test_fixture_of_the_real_head_agents_replays_on_the_gpu_facade(name, col, tmp_path):
    """The reference's real SchedTWC / SchedColORAN classes (agents/sched_twc.py, agents/sched_colran.py: their own
    observation of 10 values per slice and their own rewards, IBSched's action_format with round-robin inside the slices)
    ran on the facade when the fixture was made.  Replayed here: the fixture's actions through the GPU facade with this build's
    plugins, the head kernel (ranenv_bind_head_outputs) producing the observation and both rewards every TTI."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from intent_radio_sched_multi_slice_amd import plugins
    from intent_radio_sched_multi_slice_amd.comm_env import DEFAULT_CONFIGS, MARLCommEnv
    from intent_radio_sched_multi_slice_amd.scenario import slice_usecase_from_req
    fx = load_golden("agents_on_facade")
    S, U, R, G, Us, seed, steps = (int(x) for x in fx["cfg"])
    cfg = dict(DEFAULT_CONFIGS["mult_slice"], max_number_steps=steps)
    env = MARLCommEnv(plugins.MimicQuadriga, plugins.MultSliceTraffic, plugins.SimpleMobility, plugins.MultSliceAssociation,
                      "mult_slice", name, seed, root_path=str(tmp_path), config=cfg, max_episode_number=2, max_ues_slice=Us)
    ce = env.comm_env
    agent = _Agent(env, ce.max_number_ues, ce.max_number_slices, ce.max_number_basestations, ce.num_available_rbs,
                   max_ues_slice=Us, rbs_per_rbg=G, sort=False)       # their IBSched is built with enable_sort_slices=False
    env.set_agent_functions(agent.obs_space_format, agent.action_format, agent.calculate_reward, None, None)
    dev = env._dev
    dev.enable_heads(None)                   # outputs bound before the reset: the reset observation is a head observation too
    env.reset(seed=seed, options={"initial_episode": 0})
    dev.set_slice_usecase(slice_usecase_from_req(ce.slice_req, S)[None])      # SchedColORAN's slice-name table as data
    np.testing.assert_allclose(dev.head_obs[0].cpu().numpy(), fx[f"{name}_reset_obs"], rtol=2e-6, atol=OBS_TOL)
    for t in range(steps):
        a = fx[f"{name}_action"][t]
        action = {"player_0": a.copy()}
        action.update({f"player_{s + 1}": 0 for s in range(S)})                # fixed_intra = "rr" (sched_twc.py:415-422)
        env.step(action)
        raw = env._last_raw
        sched = np.asarray(raw["sched_decision"])[0]
        assert np.array_equal(sched.sum(axis=1).astype(np.int32), fx[f"{name}_rb_count"][t]), (name, t)
        assert np.array_equal(env._last_traffic, fx[f"{name}_traffic"][t]), (name, t)
        for k in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts", "buffer_occupancies", "buffer_latencies"):
            assert np.array_equal(raw[k], fx[f"{name}_{k}"][t]), (name, t, k)
        np.testing.assert_allclose(dev.head_obs[0].cpu().numpy(), fx[f"{name}_obs"][t], rtol=2e-6, atol=OBS_TOL, err_msg=str((name, t)))
        np.testing.assert_allclose(dev.head_reward[0, col].item(), fx[f"{name}_reward"][t, 0], rtol=0, atol=REW_TOL, err_msg=str((name, t)))
    env.close()


def _exogenous_inputs_of_the_fixture(fx, name):
    """The episode's scenario, SE tiles and traffic as the facade's plugins drew them from the env's one rng (reset:
    association, mobility, channel; every TTI: mobility, channel, traffic), regenerated with this build's plugins and checked
    against what the fixture recorded."""
    from intent_radio_sched_multi_slice_amd import plugins
    from intent_radio_sched_multi_slice_amd.scenario import IDLE_UE_MAX_AGE, IDLE_UE_MAX_PKTS, IDLE_UE_PKT_SIZE, ScenarioTables
    S, U, R, G, Us, seed, steps = (int(x) for x in fx["cfg"])
    rng = np.random.default_rng(seed)
    ues = plugins.UEs(U, np.repeat(IDLE_UE_MAX_AGE, U), np.repeat(IDLE_UE_MAX_PKTS, U), np.repeat(IDLE_UE_PKT_SIZE, U))
    mob = plugins.SimpleMobility(U, rng, "")
    ch = plugins.MimicQuadriga(U, 1, np.array([R]), rng, "", "mult_slice")
    tr = plugins.MultSliceTraffic(U, rng, "")
    assoc = plugins.MultSliceAssociation(ues, U, 1, S, rng, "")
    bua, bsa, sua, req = assoc.step(np.zeros((1, U)), np.zeros((1, S)), np.zeros((S, U)), {f"slice_{i}": {} for i in range(S)}, 0, 0)
    assert np.array_equal(sua, fx[f"{name}_slice_ue_assoc"])
    se = [np.asarray(ch.step(0, 0, mob.step(0, 0)))[0]]
    traffic = []
    for t in range(steps):
        m = mob.step(t, 0)
        se.append(np.asarray(ch.step(t, 0, m))[0])
        traffic.append(np.asarray(tr.step(sua, req, t, 0)))
        np.testing.assert_array_equal(se[-1].sum(axis=1), fx[f"{name}_se_sum"][t])
        np.testing.assert_array_equal(traffic[-1], fx[f"{name}_traffic"][t])
    return bua, bsa, sua, req, ues, np.stack(se).astype(np.float32), np.stack(traffic)


@pytest.mark.parametrize("B", [3, 4], ids=["one-env-per-wave", "packed"])
@pytest.mark.parametrize("name", ["ib_sched", "marr", "mapf"])
def test_fused_step_kernel_against_the_real_agents_episode(name, B):
    """The same fixture through the BATCHED path: the fused step kernel does action_format, UEs.step, obs_space_format and
    calculate_reward itself.  For the real IBSched's episode it gets the fixture's scores and per-slice scheduler choices; for
    the real MARR's / MAPF's it gets nothing -- the device policy must produce the very scores the reference's agent.step
    returned, TTI after TTI, in closed loop.  RB ranges, packets and occupancies exact, observations 1e-5, rewards 1e-9.
    B = 4: at this size -- the reference's own -- an even batch is stepped two envs per wave (ranenv_core_kernel_packed); the envs
    compared are the second half of wave 0 (1), the first half of wave 0 (0) and the first half of wave 1 (2)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from intent_radio_sched_multi_slice_amd.batched_env import BatchedRanEnv
    from intent_radio_sched_multi_slice_amd.scenario import ScenarioTables
    fx = load_golden("agents_on_facade")
    S, U, R, G, Us, seed, steps = (int(x) for x in fx["cfg"])
    bua, bsa, sua, req, ues, se, traffic = _exogenous_inputs_of_the_fixture(fx, name)
    marl = name == "ib_sched"
    tabs = ScenarioTables.empty(1, S, U, Us)
    tabs.set_from_reference(0, bsa, sua, req, marl, (ues.pkt_sizes, ues.max_buffer_pkts, ues.max_buffer_latencies))
    env = BatchedRanEnv(batch=B, n_slices=S, n_ues=U, n_rbs=R, rbs_per_rbg=G, max_ues_slice=Us, n_scenarios=1, max_steps=steps)
    env.load_scenarios(tabs)
    env.set_episodes(scenario=0)
    env.set_policy({"ib_sched": 0, "marr": 1, "mapf": 2}[name], {"ib_sched": 255, "marr": 0, "mapf": 1}[name])
    tile = lambda t: np.broadcast_to(np.ascontiguousarray(se[t].T), (B, R, U))
    obs = env.reset(se_tiles=tile(0))

    def flat(o, b):
        oi, oa = o["obs_inter"][b].cpu().numpy(), o["obs_intra"][b].cpu().numpy()
        return np.concatenate([oi, oa.ravel()]) if marl else oi
    np.testing.assert_allclose(flat(obs, 1), fx[f"{name}_reset_obs"], rtol=0, atol=OBS_TOL)
    for t in range(steps):
        a = fx[f"{name}_action"][t]
        tb = np.broadcast_to(traffic[t], (B, U))
        if marl:
            sc = np.broadcast_to(a[:S], (B, S))
            ic = np.broadcast_to(a[S:].astype(np.uint8), (B, S))
            obs, rew, done = env.step(sc, ic, tb, tile(t + 1))
        else:
            obs, rew, done = env.step(None, None, tb, tile(t + 1))
            # the device's MARR / MAPF against the real agent.step(obs) of this TTI
            np.testing.assert_allclose(env.views()["policy_scores"][2].cpu().numpy(), a, rtol=0, atol=1e-12, err_msg=str((name, t)))
        v = {k: x[1].cpu().numpy() for k, x in env.views().items()}
        ro = {k: x[1].cpu().numpy() for k, x in env.raw_observation().items()}
        cnt = fx[f"{name}_rb_count"][t]
        assert np.array_equal(v["rb_count"], cnt), (name, t)
        assert np.array_equal(v["rb_start"][cnt > 0], fx[f"{name}_rb_start"][t][cnt > 0]), (name, t)
        for k in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts", "buffer_occupancies", "buffer_latencies"):
            assert np.array_equal(ro[k], fx[f"{name}_{k}"][t]), (name, t, k)
        np.testing.assert_allclose(flat(obs, 0), fx[f"{name}_obs"][t], rtol=0, atol=OBS_TOL, err_msg=str((name, t)))
        want = fx[f"{name}_reward"][t]
        got = rew[2].cpu().numpy() if marl else rew[2, :1].cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=REW_TOL, err_msg=str((name, t)))
    assert bool(done.all())
    env.close()


@pytest.mark.parametrize("name,reward", [("sched_twc", "twc"), ("sched_colran", "colran")])
def test_head_vec_env_against_the_real_head_agents_episode(name, reward):
    """adapters.HeadVecEnv (the SB3 VecEnv view: head kernel + IBSched's action_format with round-robin on the device, SE and
    traffic replayed from HBM pools) on the episode the real SchedTWC / SchedColORAN played: their observation and reward
    at every TTI."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from intent_radio_sched_multi_slice_amd.adapters import HeadVecEnv
    from intent_radio_sched_multi_slice_amd.batched_env import BatchedRanEnv
    from intent_radio_sched_multi_slice_amd.scenario import ScenarioTables, slice_usecase_from_req
    fx = load_golden("agents_on_facade")
    S, U, R, G, Us, seed, steps = (int(x) for x in fx["cfg"])
    bua, bsa, sua, req, ues, se, traffic = _exogenous_inputs_of_the_fixture(fx, name)
    tabs = ScenarioTables.empty(1, S, U, Us)
    tabs.set_from_reference(0, bsa, sua, req, False, (ues.pkt_sizes, ues.max_buffer_pkts, ues.max_buffer_latencies))
    B = 2
    env = BatchedRanEnv(batch=B, n_slices=S, n_ues=U, n_rbs=R, rbs_per_rbg=G, max_ues_slice=Us, n_scenarios=1, max_steps=steps)
    env.load_scenarios(tabs)
    env.bind_se_pool(torch.as_tensor(np.ascontiguousarray(np.swapaxes(se[1:], 1, 2)), device=env.device))    # the steps' tiles
    env.bind_traffic_pool(torch.as_tensor(traffic.astype(np.int32), device=env.device))
    env.set_episodes(scenario=0, se_base=0, se_len=steps, trf_base=0, trf_len=steps)
    venv = HeadVecEnv(env, reward=reward, slice_usecase=slice_usecase_from_req(req, S)[None])
    env.reset(se_tiles=np.broadcast_to(np.ascontiguousarray(se[0].T), (B, R, U)))          # the reset's own channel draw
    np.testing.assert_allclose(env.head_obs[1].cpu().numpy(), fx[f"{name}_reset_obs"], rtol=2e-6, atol=OBS_TOL)
    for t in range(steps):
        obs, rew, dones, infos = venv.step(np.broadcast_to(fx[f"{name}_action"][t], (B, S)))
        if t < steps - 1:          # (at the terminal TTI the VecEnv protocol hands back the next episode's first observation)
            np.testing.assert_allclose(obs[0], fx[f"{name}_obs"][t], rtol=2e-6, atol=OBS_TOL, err_msg=str((name, t)))
        else:
            np.testing.assert_allclose(infos[0]["terminal_observation"], fx[f"{name}_obs"][t], rtol=2e-6, atol=OBS_TOL)
        np.testing.assert_allclose(rew[1], fx[f"{name}_reward"][t, 0], rtol=2e-6, atol=1e-6, err_msg=str((name, t)))
        assert bool(dones[0]) == (t == steps - 1)
    venv.close()
