"""The small scenario plugins (plugins.py) against tests/golden/plugins.npz, which the reference's own
classes produced (tests/golden/gen_golden.py plugins): associations/simple_slice.py:27-112,
associations/mult_slice_seq.py:38-46, associations/mult_slice.py:444-452, associations/simple.py:27-41,
channels/quadriga.py:78-87, channels/quadriga_seq.py:28-39, channels/fixed_se.py, traffics/simple.py,
mobilities/simple.py."""
import json

import numpy as np

from intent_radio_sched_multi_slice_amd import plugins
from tests.common import load_golden


def _req_json(req):
    def conv(o):
        if isinstance(o, dict):
            return {k: conv(v) for k, v in o.items()}
        if callable(o):
            return "ufunc:" + o.__name__
        if isinstance(o, np.integer):
            return int(o)
        if isinstance(o, np.floating):
            return float(o)
        return o
    return json.dumps(conv(req), sort_keys=True)


def _mk():
    U, S = 4, 2
    ues = plugins.UEs(U, np.repeat(100, U), np.repeat(1024, U), np.repeat(100, U))
    return U, S, ues, np.random.default_rng(10)


def test_simple_slice_association_matches_the_reference():
    fx = load_golden("plugins")
    U, S, ues, rng = _mk()
    a = plugins.SimpleSliceAssociation(ues, U, 1, S, rng, "")
    bua = np.ones((1, U)); bsa = np.ones((1, S)); sua = np.zeros((S, U)); sua[0, :2] = 1; sua[1, 2:] = 1
    r0 = a.step(bua, bsa, sua, {"old": 1}, 0, 0)
    r5 = a.step(bua, bsa, sua, {"kept": 1}, 5, 0)
    assert _req_json(r0[3]) == str(fx["simple_slice_req_step0"])
    assert _req_json(r5[3]) == str(fx["simple_slice_req_step5"])
    assert int(r0[0] is bua and r0[1] is bsa and r0[2] is sua) == int(fx["simple_slice_passthrough"][0])
    s = plugins.SimpleAssociation(ues, U, 1, S, rng, "")
    rs = s.step(bua, bsa, sua, {"x": 2}, 3, 1)
    assert int(rs[0] is bua and rs[1] is bsa and rs[2] is sua and rs[3] == {"x": 2}) == int(fx["simple_passthrough"][0])


def test_simple_slice_intents_load_into_scenario_tables():
    """The toy intents are a valid scenario for the device tables (BASELINE configs[0] plumbing)."""
    from intent_radio_sched_multi_slice_amd.scenario import ScenarioTables
    t = ScenarioTables.empty(1, 2, 4, 4)
    sua = np.zeros((2, 4)); sua[0, [0, 2]] = 1; sua[1, [1, 3]] = 1
    t.set_from_reference(0, np.ones((1, 2)), sua, plugins.simple_slice_req(), True)
    assert t.slice_nparams[0].tolist() == [3, 2]
    assert t.slice_buffer_size[0].tolist() == [10, 10] and t.slice_message_size[0].tolist() == [1, 1]
    assert t.param_value[0, 0].tolist() == [99.0, 20.0, 1.0]


def test_episode_choice_rules_match_the_reference():
    fx = load_golden("plugins")
    U, S, ues, rng = _mk()
    seq = plugins.MultSliceAssociationSeq(ues, U, 1, S, rng, ".")
    ms = plugins.MultSliceAssociation(ues, U, 1, S, rng, ".", generator_mode=False)
    qc = plugins.QuadrigaChannel(U, 1, np.array([25]), rng, ".", "x")
    qs = plugins.QuadrigaChannelSeq(U, 1, np.array([25]), rng, ".", "x")
    for k, (e, c) in enumerate(fx["choose_pairs"]):
        e, c = int(e), int(c)
        assert [int(v) for v in seq.choose_episode(e, c)] == fx["choose_mult_slice_seq"][k].tolist(), (e, c)
        assert [int(v) for v in ms.choose_episode(e, c)] == fx["choose_mult_slice"][k].tolist(), (e, c)
        assert [int(v) for v in qc.choose_episode(e, c)] == fx["choose_quadriga"][k].tolist(), (e, c)
        assert [int(v) for v in qs.choose_episode(e, c)] == fx["choose_quadriga_seq"][k].tolist(), (e, c)
    attrs = json.loads(str(fx["seq_attrs"]))
    assert seq.scenario_name == attrs["scenario_name"] and seq.channels_per_scenario == attrs["channels_per_scenario"]
    assert bool(seq.generator_mode) == attrs["generator_mode"]
    assert qs.channel_eps_per_scenario == attrs["channel_eps_per_scenario"]


def test_trivial_plugins_match_the_reference():
    fx = load_golden("plugins")
    U, S, ues, rng = _mk()
    sua = np.zeros((S, U))
    fse = plugins.FixedSE(U, 1, np.array([25]), rng, "", "")
    assert np.array_equal(np.asarray(fse.step(3, 1, np.ones((U, 2)), None)), fx["fixed_se"])
    assert np.array_equal(np.asarray(plugins.SimpleTraffic(U, rng, "").step(sua, {}, 2, 0)), fx["simple_traffic"])
    assert np.array_equal(np.asarray(plugins.SimpleMobility(U, rng, "").step(2, 0)), fx["simple_mobility"])


def test_mult_slice_seq_replays_one_scenario_file_per_hundred_episodes(tmp_path):
    """Replay mode: episodes 0..99 read ep_0.npz, 100..199 read ep_1.npz (associations/mult_slice_seq.py:38-46)."""
    from intent_radio_sched_multi_slice_amd.scenario import generate_reference_scenario, save_episode_npz
    S, U, steps = 5, 25, 4
    d = tmp_path / "associations" / "data" / "mult_slice"
    d.mkdir(parents=True)
    gen = np.random.default_rng(3)
    want = []
    for n in range(2):
        bua, bsa, sua, req, use = generate_reference_scenario(gen, S, U)
        save_episode_npz(str(d / f"ep_{n}.npz"), bua, bsa, sua, req, use, steps)
        want.append(sua)
    ues = plugins.UEs(U, np.repeat(100, U), np.repeat(1024, U), np.repeat(100, U))
    a = plugins.MultSliceAssociationSeq(ues, U, 1, S, np.random.default_rng(0), str(tmp_path))
    for ep, idx in ((0, 0), (57, 0), (99, 0), (100, 1), (199, 1)):
        out = a.step(None, None, None, None, 0, ep)
        assert np.array_equal(out[2], want[idx]), ep
        assert a.current_episode == idx
