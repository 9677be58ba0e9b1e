"""CPU oracle vs the golden vectors captured from the reference's own functions.

This is what pins the oracle: every fixture in tests/golden was
produced by tests/golden/gen_golden.py importing agents/common.py, agents/ib_sched.py,
agents/marr.py, agents/mapf.py, associations/mult_slice.py and traffics/mult_slice.py.
"""
import numpy as np
import pytest

from oracle import pyoracle
from tests.common import HEAD_CASES, AGENT_CASES, ATOL, RTOL, TRACE_CASES, load_golden, tables_from
from tests.synth import se_tile


def test_np_sum_matches_numpy_bitwise():
    rng = np.random.default_rng(0)
    for n in list(range(0, 200)) + [255, 256, 257, 300, 1000, 4097]:
        a = rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 3, n)
        assert pyoracle.np_sum(a) == (np.sum(a) if n else 0.0), n
        b = np.abs(rng.normal(10, 1.5, n)).astype(np.float32).astype(np.float64)
        assert pyoracle.np_sum(b) == (np.sum(b) if n else 0.0), n


def test_round_int_equal_sum_golden():
    fx = load_golden("functions")
    for vals, tgt, exp in ((fx["rie_values"], fx["rie_target"], fx["rie_expected"]),
                           (fx["rie_tie_values"], fx["rie_tie_target"], fx["rie_tie_expected"])):
        for v, t, e in zip(vals, tgt, exp):
            n = int(np.sum(~np.isnan(v)))
            got = pyoracle.round_int_equal_sum(v[:n], int(t))
            assert np.array_equal(got, e[:n]), (v[:n], t, got, e[:n])
            assert got.sum() == t and (got >= 0).all()   # agents/common.py:456-459 invariants


def test_scores_to_rbs_golden():
    fx = load_golden("functions")
    for a, assoc, t, e in zip(fx["s2r_action"], fx["s2r_assoc"], fx["s2r_target"], fx["s2r_expected"]):
        n = int(np.sum(~np.isnan(a)))
        got = pyoracle.scores_to_rbs(a[:n], int(t), assoc[:n])
        assert np.array_equal(got, e[:n]), (a[:n], assoc[:n], t, got, e[:n])
        assert int(np.sum(got * assoc[:n])) == t


def test_sort_slices_golden():
    fx = load_golden("assoc_traffic")
    tabs = tables_from(fx, "assoc_tab_")
    for i in range(tabs.n_scenarios):
        got = pyoracle.sort_slices(tabs.slice_nues[i], tabs.slice_traffic[i], tabs.slice_has_req[i])
        assert np.array_equal(got, tabs.sorted_slices[i])


def _make_env(fx):
    cfg = fx["cfg"]
    S, U, R, G, Us = (int(x) for x in cfg[:5])
    cap = int(cfg[8]) if len(cfg) > 8 else 400
    steps = int(cfg[7])
    ocfg = pyoracle.make_cfg(S, U, R, G, Us, bandwidth_hz=float(fx["bw"]), max_age_cap=cap, max_steps=steps)
    return pyoracle.OracleEnv(ocfg), (S, U, R, G, Us)


@pytest.mark.parametrize("case", AGENT_CASES)
def test_agent_side_golden(case):
    """obs_space_format -> calculate_reward -> action_format (+ MARR/MAPF actions) against the
    reference on synthetic raw observations (empty buffers, zero history, heavy loss...)."""
    fx = load_golden(case)
    env, (S, U, R, G, Us) = _make_env(fx)
    tabs = tables_from(fx)
    seed = int(fx["cfg"][5])
    steps = len(fx["scen"])
    for t in range(steps):
        env.set_scenario(tabs, int(fx["scen"][t]))
        se = se_tile(seed, t, U, R, low_se_every=7)
        env.agent_observe(fx["sent"][t], fx["dropped"][t], fx["occ"][t], fx["lat"][t], se, fx["rowsum"][t])
        o = env.obs()
        np.testing.assert_allclose(env.drift(), fx["drift"][t], rtol=RTOL, atol=ATOL, err_msg=f"drift t={t}")
        np.testing.assert_allclose(o["obs_inter"], fx["obs_inter"][t], rtol=RTOL, atol=ATOL, err_msg=f"t={t}")
        np.testing.assert_allclose(o["obs_intra"], fx["obs_intra"][t], rtol=RTOL, atol=ATOL, err_msg=f"t={t}")
        assert np.array_equal(o["mask_inter"], fx["mask_inter"][t])
        assert np.array_equal(o["mask_intra"], fx["mask_intra"][t])
        np.testing.assert_allclose(o["reward"], fx["reward"][t], rtol=RTOL, atol=ATOL, err_msg=f"reward t={t}")
        np.testing.assert_allclose(env.policy_marr(), fx["marr"][t], rtol=0, atol=0)
        np.testing.assert_allclose(env.policy_mapf(), fx["mapf"][t], rtol=RTOL, atol=ATOL, err_msg=f"mapf t={t}")
        start, count, dense = env.action_format(fx["scores"][t], fx["intra"][t])
        assert np.array_equal(count, fx["rb_count"][t]), (t, count, fx["rb_count"][t])
        used = count > 0
        assert np.array_equal(start[used], fx["rb_start"][t][used]), t
        assert dense.sum() == count.sum()
        if fx["mask_inter"][t].any():
            assert count.sum() == (R // G) * G            # ib_sched.py:345-347


@pytest.mark.parametrize("case", HEAD_CASES)
def test_alternative_heads_golden(case):
    """SchedTWC / SchedColORAN observation (10*S values) and rewards against the reference's own classes
    (agents/sched_twc.py:165-413, agents/sched_colran.py:348-419) on synthetic raw observations; their
    deque holds every TTI twice, which the oracle restates."""
    fx = load_golden(case)
    env, (S, U, R, G, Us) = _make_env(fx)
    tabs = tables_from(fx)
    seed = int(fx["cfg"][5])
    neg = 0
    for t in range(len(fx["scen"])):
        env.set_scenario(tabs, int(fx["scen"][t]))
        se = se_tile(seed, t, U, R, low_se_every=7)
        env.agent_observe(fx["sent"][t], fx["dropped"][t], fx["occ"][t], fx["lat"][t], se, fx["rowsum"][t])
        env.set_pkt_throughputs(fx["pkt_thr"][t])
        obs, r_twc, r_col = env.heads(fx["usecase"][t])
        np.testing.assert_allclose(obs, fx["head_obs"][t], rtol=RTOL, atol=ATOL, err_msg=f"t={t}")
        np.testing.assert_allclose(r_twc, fx["reward_twc"][t], rtol=RTOL, atol=ATOL, err_msg=f"twc t={t}")
        np.testing.assert_allclose(r_col, fx["reward_colran"][t], rtol=RTOL, atol=ATOL, err_msg=f"colran t={t}")
        neg += fx["reward_twc"][t] < 0
    assert neg > 3 and np.abs(fx["reward_colran"]).max() > 0     # the fixtures exercise both rewards


@pytest.mark.parametrize("case", TRACE_CASES)
def test_closed_loop_trace_golden(case):
    """Whole oracle step (action_format -> UEs.step -> obs -> reward) in closed loop; the agent
    side of every step was produced by the reference's code, the env core by this oracle."""
    fx = load_golden(case)
    env, (S, U, R, G, Us) = _make_env(fx)
    tabs = tables_from(fx)
    cfg = fx["cfg"]
    seed, steps_per_ep = int(cfg[5]), int(cfg[7])
    plumbing = len(cfg) > 9 and int(cfg[9]) == 1
    k = 0
    for ep, idx in enumerate(fx["scen_ids"]):
        get_se = (lambda t: np.full((U, R), 2.0, dtype=np.float32)) if plumbing else \
            (lambda t: se_tile(seed + ep, t, U, R))
        env.set_scenario(tabs, int(idx))
        env.reset(get_se(0))
        o = env.obs()
        got = np.concatenate([o["obs_inter"], o["obs_intra"].ravel()])
        np.testing.assert_allclose(got, fx["reset_obs"][ep], rtol=RTOL, atol=ATOL)
        for t in range(steps_per_ep):
            start, count, _ = env.action_format(fx["scores"][k], fx["intra"][k])
            assert np.array_equal(count, fx["rb_count"][k]), (case, ep, t)
            used = count > 0
            assert np.array_equal(start[used], fx["rb_start"][k][used]), (case, ep, t)
            env.step(fx["scores"][k], fx["intra"][k], get_se(t), fx["traffic"][k])
            raw = env.raw()
            for name in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts"):
                assert np.array_equal(raw[name], fx[name][k]), (case, name, ep, t)
            assert np.array_equal(raw["buffer_occupancies"], fx["buffer_occupancies"][k])
            assert np.array_equal(raw["buffer_latencies"], fx["buffer_latencies"][k])
            o = env.obs()
            np.testing.assert_allclose(o["obs_inter"], fx["obs_inter"][k], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(o["obs_intra"], fx["obs_intra"][k], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(o["reward"], fx["reward"][k], rtol=RTOL, atol=ATOL)
            assert np.array_equal(o["mask_inter"], fx["mask_inter"][k])
            assert np.array_equal(o["mask_intra"], fx["mask_intra"][k])
            k += 1
        assert env.step_number == steps_per_ep


def test_trace_exercises_the_interesting_regimes():
    """The fixtures must actually contain drops, backlog, empty buffers and negative rewards."""
    fx = load_golden("trace_ref_random")
    assert fx["dropped_pkts"].sum() > 0
    assert (fx["buffer_occupancies"] > 0).any() and (fx["buffer_occupancies"] == 0).any()
    assert (fx["reward"][:, 0] < 0).any()
    assert (fx["buffer_latencies"] > 0).any()
    assert (fx["pkt_effective_thr"] < fx["pkt_throughputs"]).any()


def test_pkt_throughputs_two_roundings_against_numpy():
    """UEs.get_pkt_throughputs is unpinned (the env core is absent from the reference snapshot): the oracle's default scales the
    SUM by BW / R, RANENV_F_SCALE_PER_ELEMENT's convention scales every ELEMENT first.  Both against numpy itself, on random
    masks / SE and on integer-valued SE, where they differ by a packet now and then (the named case: 54 RBs of SE 1.0 from RB 77,
    512-bit packets: 78 125 against 78 124)."""
    from intent_radio_sched_multi_slice_amd.scenario import generate_scaled_scenarios
    S, U, R, G, Us = 5, 25, 135, 5, 5
    tabs = generate_scaled_scenarios(2, seed=3, n_slices=S, n_ues=U, max_ues_slice=Us, min_slices=3, min_ues=2)
    tabs.ue_pkt_size[:] = 512
    bw = 100e6 / R
    rng = np.random.default_rng(5)
    differ = 0
    for trial in range(40):
        se = (rng.integers(1, 7, (U, R)) if trial % 2 else rng.uniform(0.1, 7.5, (U, R))).astype(np.float32)
        dense = np.zeros((U, R), dtype=np.uint8)
        for u in range(U):
            s = int(rng.integers(0, R)); c = int(rng.integers(0, R - s + 1))
            dense[u, s:s + c] = 1
        if trial == 1:
            se[0, :] = 1.0; dense[0, :] = 0; dense[0, 77:131] = 1
        got = {}
        for flagged in (False, True):
            o = pyoracle.OracleEnv(pyoracle.make_cfg(S, U, R, G, Us))
            o.set_scale_per_element(flagged); o.set_scenario(tabs, 0); o.reset(se)
            o.core_step(dense, se, np.zeros(U))
            got[flagged] = o.raw()["pkt_throughputs"]
        row = dense.astype(np.float64) * se.astype(np.float64)
        assert np.array_equal(got[False], np.floor(np.sum(row, axis=1) * bw / 512.0))
        assert np.array_equal(got[True], np.floor(np.sum(row * bw, axis=1) / 512.0))
        if trial == 1:
            assert got[False][0] == 78125 and got[True][0] == 78124
        differ += int(np.sum(got[False] != got[True]))
    assert differ >= 1
