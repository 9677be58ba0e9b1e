#!/usr/bin/env python3
"""History files of this build read by the reference's own result scripts (build container only: needs /root/reference).

    python tests/golden/check_reference_readers.py

Two of the reference's real agents (agents/marr.py MARR, agents/mapf.py MAPF) play the same two 30-TTI episodes on this
build's MARLCommEnv facade with ``save_hist=True`` (the reference's plugins, INTEGRATION.md section 2's shim, the CPU stand-in
of gen_golden_agents.py under the facade).  Then, from the directory that holds ``hist/``, the reference's
results/gen_results.py is asked what it asks of its own runs:
  * fair_comparison_check (:1587-1635): pkt_incoming, mobility, spectral_efficiencies, the three association arrays and
    slice_req must be identical across the agents' files of an episode -- i.e. this build's exogenous inputs do not depend on
    the actions, and its files compare equal under np.array_equal the way the reference compares them;
  * calc_slice_violations / calc_intent_distance / calc_throughput_slice / calc_total_throughput (:770-1022) run on every file.
Exit status 0 = all of it held.  tests/test_reference_goldens_r3.py runs this in a subprocess when the reference is there.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_agents as gga          # noqa: E402  (OracleDevice, the shims; its main() is not run)


def main():
    gga.install_shims()
    from intent_radio_sched_multi_slice_amd import comm_env
    comm_env.BatchedRanEnv = gga.OracleDevice
    from agents.mapf import MAPF
    from agents.marr import MARR
    from associations.mult_slice import MultSliceAssociation
    from channels.mimic_quadriga import MimicQuadriga
    from mobilities.simple import SimpleMobility
    from traffics.mult_slice import MultSliceTraffic

    class GenAssociation(MultSliceAssociation):
        def __init__(self, *a, **k):
            super().__init__(*a, generator_mode=True, **k)

    root = tempfile.mkdtemp()
    steps, n_ep = 30, 2
    for name, cls in (("marr", MARR), ("mapf", MAPF)):
        cfg = dict(comm_env.DEFAULT_CONFIGS["mult_slice"], max_number_steps=steps)
        env = comm_env.MARLCommEnv(MimicQuadriga, MultSliceTraffic, SimpleMobility, GenAssociation, "mult_slice", name, 10,
                                   root_path=root, initial_episode_number=0, simu_name="mult_slice", save_hist=True,
                                   max_episode_number=n_ep, enable_random_episodes=False, config=cfg, max_ues_slice=5)
        ce = env.comm_env
        agent = cls(env, ce.max_number_ues, ce.max_number_slices, ce.max_number_basestations, ce.num_available_rbs, seed=10)
        env.set_agent_functions(agent.obs_space_format, agent.action_format, agent.calculate_reward, agent.get_obs_space(),
                                agent.get_action_space())
        agent.init_agent()
        obs, _ = env.reset(seed=10, options={"initial_episode": 0})           # simu.py:547-566
        for ep in range(n_ep):
            terminated = False
            while not terminated:
                obs, reward, terminated, truncated, info = env.step(agent.step(obs))
            if ep + 1 < n_ep:
                obs, _ = env.reset()
    os.chdir(root)                                                           # gen_results.py reads hist/... relative to the cwd
    sys.path.insert(0, os.path.join(HERE))
    import gen_golden_r3 as r3                                               # load_gen_results (its generators are not run)
    grs = r3.load_gen_results()
    assert grs.fair_comparison_check(["marr", "mapf"], np.arange(n_ep), ["mult_slice"]) is True
    n = 0
    for name in ("marr", "mapf"):
        for ep in range(n_ep):
            data = np.load(f"hist/mult_slice/{name}/ep_{ep}.npz", allow_pickle=True)
            v = grs.calc_slice_violations(data)[0]
            d = grs.calc_intent_distance(data)
            assert v.shape == (steps,) and d.shape == (steps,) and (d <= 0).all() and (v >= 0).all()
            for s in range(5):
                assert np.asarray(grs.calc_throughput_slice(data, "pkt_effective_thr", s)).shape[0] == steps
            assert np.asarray(grs.calc_total_throughput(data, "pkt_effective_thr", np.arange(5))).shape[0] == steps
            n += 1
    print(f"fair_comparison_check passed for marr / mapf over {n_ep} episodes; {n} history files read by the reference's functions")


if __name__ == "__main__":
    main()
