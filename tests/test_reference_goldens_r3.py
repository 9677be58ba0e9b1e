"""CPU checks against fixtures produced by running the reference's own classes (tests/golden/gen_golden_r3.py):
channel plugins (QuadrigaChannel / QuadrigaChannelSeq / MimicQuadriga), the scenario file in the reference writer's
layout read back by the reference's replay mode, and the paper's evaluation metrics (results/gen_results.py:845-1022)
against the oracle's live observation."""
import json
import os

import numpy as np
import pytest

from intent_radio_sched_multi_slice_amd import plugins, scenario
from tests.common import GOLDEN, load_golden, tables_from
from tests.synth import se_tile


# ---------------------------------------------------------------------------------------------- channels
class _FixtureFile:
    """What QuadrigaChannel gets from h5py.File: an object with get("target_cell_power") and close()."""

    def __init__(self, fx, path, opened):
        import re
        opened.append(path)
        m = re.search(r"assoc_(\d+)/ep_(\d+)/target_cell_power\.mat$", path)
        self.power = fx[f"power_{int(m.group(1))}_{int(m.group(2))}"]

    def get(self, name):
        assert name == "target_cell_power"
        return self.power

    def close(self):
        pass


@pytest.mark.parametrize("name,cls", [("quadriga", plugins.QuadrigaChannel), ("quadriga_seq", plugins.QuadrigaChannelSeq)])
def test_quadriga_channel_step_reproduces_the_reference_class(name, cls, monkeypatch):
    fx = load_golden("channels_ref")
    U, R, _ = (int(x) for x in fx["sizes"])
    opened = []
    monkeypatch.setattr(cls, "_open", lambda self, path: _FixtureFile(fx, path, opened))
    ch = cls(U, 1, np.array([R]), np.random.default_rng(0), str(fx["root_path"]), "whatever")
    for (step, ep), want in zip(fx[f"{name}_calls"], fx[f"{name}_se"]):
        got = ch.step(int(step), int(ep), np.ones((U, 2)))
        assert got.shape == (1, U, R) and got.dtype == np.float64
        np.testing.assert_array_equal(got, want)
    assert opened == json.loads(str(fx[f"{name}_opened"]))          # which file, and only when the episode changes


def test_the_oracle_side_transform_reproduces_the_reference_class():
    """oracle/pyoracle.quadriga_se_from_power is the checker of the ingest kernel (ranenv_se_from_power): pin it."""
    from oracle import pyoracle
    fx = load_golden("channels_ref")
    U, R, _ = (int(x) for x in fx["sizes"])
    for (step, ep), want in zip(fx["quadriga_calls"], fx["quadriga_se"]):
        power = fx[f"power_{int(ep)}_0"][int(step)]                   # (1, 1, R, U)
        se = pyoracle.quadriga_se_from_power(power, R)
        np.testing.assert_array_equal(np.squeeze(se.transpose()), want[0])


@pytest.mark.parametrize("seed", [10, 15])
def test_mimic_quadriga_draw_order(seed):
    fx = load_golden("channels_ref")
    U, R = (int(x) for x in fx["mimic_sizes"])
    ch = plugins.MimicQuadriga(U, 1, np.array([R]), np.random.default_rng(seed), "", "")
    want = fx[f"mimic_seed{seed}"]
    for ep in range(want.shape[0]):
        for t in range(want.shape[1]):
            np.testing.assert_array_equal(ch.step(t, ep, np.ones((U, 2))), want[ep, t])


# ---------------------------------------------------------------------------------------------- scenario file
def test_scenario_file_in_the_reference_writers_layout():
    root = os.path.join(GOLDEN, "ref_layout")
    path = os.path.join(root, "associations", "data", "mult_slice", "ep_0.npz")
    exp = load_golden("ref_layout_expected")
    ep = scenario.load_episode_npz(path)
    assert set(ep) == set(scenario.EPISODE_FILE_KEYS) and ep["hist_slices_to_use"].dtype == object
    assert ep["hist_slices_lifetime"].shape == (ep["hist_slice_ue_assoc"].shape[0], 5)
    S, U = 5, 25
    tabs = scenario.tables_from_episode_files([path], S, U, 5)
    # buffer parameters come from update_ues in the reference; the reader takes them from the slice templates
    for k, v in tabs.arrays().items():
        np.testing.assert_array_equal(v, exp["tab_" + k], err_msg=k)
    # replay mode: same outputs as the reference's own replay-mode class reading this file
    ues = plugins.UEs(U, np.repeat(100, U), np.repeat(1024, U), np.repeat(100, U))
    replay = plugins.MultSliceAssociation(ues, U, 1, S, np.random.default_rng(0), root, generator_mode=False)
    names = json.loads(str(exp["req_names"]))
    for i, t in enumerate(exp["steps"]):
        bua, bsa, sua, req = replay.step(np.zeros((1, U)), np.zeros((1, S)), np.zeros((S, U)), {}, int(t), 0)
        np.testing.assert_array_equal(bua, exp["bua"][i]); np.testing.assert_array_equal(bsa, exp["bsa"][i])
        np.testing.assert_array_equal(sua, exp["sua"][i])
        assert {k: (v["name"] if v else None) for k, v in req.items()} == names[i]
    np.testing.assert_array_equal(ues.pkt_sizes, exp["ues_pkt_sizes"])
    np.testing.assert_array_equal(ues.max_buffer_pkts, exp["ues_max_buffer_pkts"])
    np.testing.assert_array_equal(ues.max_buffer_latencies, exp["ues_max_buffer_latencies"])


# ---------------------------------------------------------------------------------------------- evaluation metrics
def tti_metrics_from_obs(oo):
    """[violations, priority violations, distance, priority distance] of one TTI from a formatted observation: per
    active slice the minimum declared drift (undeclared metrics are 0 in the observation, flagged by entries 3..5)."""
    rows = np.asarray(oo["obs_inter"]).reshape(-1, 10)       # one row per slice, in the observation's (sorted) slice order
    drift, declared, prio = rows[:, 0:3], rows[:, 3:6] > 0, rows[:, 6]
    active = rows[:, 8] > 0                                  # slices with UEs (entry 8 = n_ues / 5)
    m = np.where(declared, drift, np.inf).min(axis=1)
    m = np.where(np.isfinite(m), m, 1.0)
    neg = active & (m < 0)
    pneg = neg & (prio > 0)
    return np.array([neg.sum(), pneg.sum(), m[neg].sum(), m[pneg].sum()])


def replay_eval_fixture(fx, on_reset, on_step):
    """The fixture's closed loop again: 3 episodes of one env, its scores / traffic replayed; callbacks get (ep, t)."""
    from oracle import pyoracle
    S, U, R, G, Us, seed, steps, n_ep = (int(x) for x in fx["cfg"])
    tabs = tables_from(fx)
    for ep, idx in enumerate(fx["scen_ids"]):
        on_reset(ep, int(idx), se_tile(seed + ep, 0, U, R))
        for t in range(steps):
            on_step(ep, t, se_tile(seed + ep, t, U, R), fx["traffic"][ep, t])
    return tabs


@pytest.mark.parametrize("window", ["live", "restarted"])
def test_oracle_observation_gives_the_reference_evaluation_metrics(window):
    """results/gen_results.py's calc_slice_violations / calc_intent_distance (run on history files of this very loop when
    the fixture was made) against what the oracle's live observation says per TTI: exactly the reference's numbers
    when its deque sees what the env's agent saw -- never cleared, reset observations included ("live") -- or, with the
    window cleared at every reset, what it gives per episode with the reset observation in front ("restarted")."""
    from oracle import pyoracle
    fx = load_golden("eval_metrics")
    S, U, R, G, Us, seed, steps, n_ep = (int(x) for x in fx["cfg"])
    tabs = tables_from(fx)
    core = pyoracle.OracleEnv(pyoracle.make_cfg(S, U, R, G, Us, max_steps=steps))
    intra = np.ones(S, dtype=np.int32)
    got = np.zeros((n_ep, steps, 4))

    def on_reset(ep, idx, se0):
        if window == "restarted":
            core.clear()
        core.set_scenario(tabs, idx)
        core.reset(se0)

    def on_step(ep, t, se, traffic):
        sc = core.policy_mapf()
        np.testing.assert_allclose(sc, fx[window + "_scores"][ep, t], rtol=0, atol=1e-12)
        core.step(sc, intra, se, traffic)
        raw = core.raw()
        assert np.array_equal(raw["pkt_effective_thr"], fx[window + "_pkt_effective_thr"][ep, t])
        got[ep, t] = tti_metrics_from_obs(core.obs())

    replay_eval_fixture(fx, on_reset, on_step)
    want = fx["live_deque" if window == "live" else "restarted_with_reset"]
    np.testing.assert_array_equal(got[:, :, :2], want[:, :, :2])
    np.testing.assert_allclose(got[:, :, 2:], want[:, :, 2:], rtol=0, atol=1e-9)


def test_what_separates_the_per_file_metrics_from_the_live_ones():
    """gen_results.py starts a fresh deque per history file and never sees the reset observation.  In a run whose window
    restarts at every reset that changes TTI 0 only (the "previous buffer was empty" rule of the throughput intent,
    agents/common.py:100-119, finds the reset's empty buffers there); against the never-cleared deque of the reference's
    agent it additionally changes the packet-loss window of an episode's first 9 TTIs."""
    fx = load_golden("eval_metrics")
    A, B = fx["live_per_file"], fx["live_deque"]
    A2, C2 = fx["restarted_per_file"], fx["restarted_with_reset"]
    assert set(np.nonzero(np.abs(A2 - C2).sum(axis=2))[1].tolist()) <= {0}
    assert set(np.nonzero(np.abs(A - B).sum(axis=2))[1].tolist()) <= set(range(10))
    # at TTI 0 the per-file numbers take every throughput intent as it is; a window that holds the reset observation counts
    # the UEs whose buffer was empty before as over-fulfilled: never more violations, never a larger distance
    assert (C2[:, 0, 0] <= A2[:, 0, 0]).all() and (C2[:, 0, 2] >= A2[:, 0, 2] - 1e-12).all()
    assert A.shape == (3, 60, 4) and (A[:, :, 0] > 0).sum() > 100 and (A[:, :, 0] < A[:, :, 0].max()).sum() > 20


@pytest.mark.skipif(not os.path.isdir("/root/reference/results"), reason="the reference is mounted in the build container only")
def test_reference_result_scripts_read_this_builds_history_files():
    """tests/golden/check_reference_readers.py in a subprocess (it puts the reference's packages and stand-ins into
    sys.modules): the real MARR and MAPF play the same episodes on the facade with save_hist; the reference's own
    fair_comparison_check (results/gen_results.py:1587-1635) then finds the exogenous inputs identical across the agents' files,
    and its violation / distance / throughput functions read every file."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(GOLDEN, "check_reference_readers.py")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "fair_comparison_check passed" in out.stdout
