"""BASELINE.json configs[1] (B 1024, MARR + round-robin) and configs[4] (mult_slice_seq sweep, B 8192, mixed
active-slice masks) as the workloads bench.py runs: a sample of their envs against the CPU oracle at oracle-sized
step counts, and the full batch through size-independent properties (RB and packet conservation, bounds)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from tests.common import comparable_views

pytestmark = pytest.mark.gpu

OBS_TOL, REW_TOL = 1e-5, 1e-9


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _follow_oracle(wl, sample, steps):
    """Step the whole workload `steps` TTIs; the envs in `sample` are mirrored by the oracle every TTI."""
    from oracle import pyoracle
    env = wl.env
    S, U, R, G, Us = env.S, env.U, env.R, env.G, env.Us
    cfg = pyoracle.make_cfg(S, U, R, G, Us, bandwidth_hz=env.bandwidth_hz, max_age_cap=env.max_age_cap, max_steps=env.max_steps)
    se_host = wl.se_pool.transpose(1, 2).contiguous().cpu().numpy()          # oracle layout: [tile][U][R]
    trf_host = wl.traffic_pool.cpu().numpy().astype(np.float64)
    eps, L = env.episodes, wl.trace_len
    intra = np.full(S, wl.intra, dtype=np.int32)
    oenvs = {}
    for b in sample:
        o = pyoracle.OracleEnv(cfg)
        o.set_scenario(wl.tables, int(wl.scenario[b]))
        o.reset(se_host[int(eps["se_base"][b] + eps["se_offset"][b] % L)])
        oenvs[b] = o
    env.reset()
    for t in range(steps):
        obs, rew, done = env.step()
        g = {k: x.cpu().numpy() for k, x in env.views().items()}
        oi, oa, rw = obs["obs_inter"].cpu().numpy(), obs["obs_intra"].cpu().numpy(), rew.cpu().numpy()
        for b, o in oenvs.items():
            tile = int(eps["se_base"][b] + (eps["se_offset"][b] + t) % L)
            row = int(eps["trf_base"][b] + (eps["trf_offset"][b] + t) % L)
            sc = o.policy_mapf() if wl.policy == 2 else o.policy_marr()
            np.testing.assert_allclose(g["policy_scores"][b], sc, rtol=0, atol=1e-12)
            _, count, _ = o.action_format(sc, intra, want_dense=False)
            assert np.array_equal(g["rb_count"][b], count), (t, b)
            o.step(sc, intra, se_host[tile], trf_host[row])
            raw, oo = o.raw(), o.obs()
            for name in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts"):
                assert np.array_equal(g[name][b].astype(np.float64), raw[name]), (t, b, name)
            np.testing.assert_allclose(oi[b], oo["obs_inter"], rtol=0, atol=OBS_TOL)
            np.testing.assert_allclose(oa[b], oo["obs_intra"], rtol=0, atol=OBS_TOL)
            np.testing.assert_allclose(rw[b], oo["reward"], rtol=0, atol=REW_TOL)
    return oenvs


def _conservation(wl, steps):
    env, dev = wl.env, wl.env.device
    env.reset()
    v = env.views()
    scen = torch.as_tensor(wl.scenario, device=dev)
    active_any = torch.as_tensor(wl.tables.slice_active.sum(axis=1) > 0, device=dev)[scen]
    max_pkts = torch.as_tensor(wl.tables.ue_max_pkts, device=dev)[scen].to(torch.int64)
    ue_slice = torch.as_tensor(wl.tables.ue_slice, device=dev)[scen]
    prev_q = v["queue_pkts"].clone().to(torch.int64)
    for t in range(steps):
        obs, rew, done = env.step()
        cnt = v["rb_count"].to(torch.int64)
        assert torch.all(cnt.sum(dim=1)[active_any] == env.R)                     # agents/ib_sched.py:345-347
        assert torch.all(cnt[ue_slice < 0] == 0)
        q = v["queue_pkts"].to(torch.int64)
        inc, sent, drop = (v[k].to(torch.int64) for k in ("pkt_incoming", "pkt_effective_thr", "dropped_pkts"))
        assert torch.all(q == prev_q + inc - drop - sent)
        assert torch.all((q >= 0) & (q <= max_pkts)) and torch.all(sent <= v["pkt_throughputs"].to(torch.int64))
        assert torch.isfinite(rew).all() and torch.isfinite(obs["obs_inter"]).all() and torch.isfinite(obs["obs_intra"]).all()
        prev_q = q


def test_config4_seq_sweep_vs_oracle():
    """mult_slice_seq: 10 scenario groups of consecutive episodes; every env of a group replays the group's
    association but its own channel trace (associations/mult_slice_seq.py:38-46, channels/quadriga_seq.py:28-39)."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_seq_workload
    wl = make_mult_slice_seq_workload(40, torch.device("cuda", 0), channels_per_scenario=4, n_traces=40, trace_len=12,
                                      max_steps=30)
    assert wl.scenario.tolist() == [e // 4 for e in range(40)]
    assert len(set(wl.se_trace.tolist())) == 40                               # no two envs share a channel trace
    act = wl.tables.slice_active.sum(axis=1)
    assert act.min() >= 3 and act.max() <= 10 and len(set(act.tolist())) > 2  # mixed masks
    nmet = {tuple(sorted(wl.tables.param_metric[i, s, :wl.tables.slice_nparams[i, s]].tolist()))
            for i in range(10) for s in range(10) if wl.tables.slice_has_req[i, s]}
    assert len(nmet) >= 4                                                     # different intent metric sets: the branchy path
    _follow_oracle(wl, list(range(40)), 30)
    wl.env.close()


def test_config4_full_batch_properties():
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
    wl, label = make_bench_workload(4, torch.device("cuda", 0), n_traces=40, trace_len=30)
    assert wl.env.B == 8192 and "mult_slice_seq" in label
    assert np.array_equal(wl.scenario, (np.arange(8192) // 100) % 10)          # 100 consecutive episodes per scenario
    assert len(set(wl.se_trace[:40].tolist())) == 40                           # ... each with its own channel trace
    _conservation(wl, 20)
    wl.env.close()


def test_config1_marr_rr_vs_oracle_and_properties():
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
    wl, label = make_bench_workload(1, torch.device("cuda", 0), n_traces=20, trace_len=16)
    assert wl.env.B == 1024 and "round-robin" in label
    _follow_oracle(wl, list(range(0, 1024, 64)), 24)
    _conservation(wl, 20)
    wl.env.close()


@pytest.mark.parametrize("policy,intra,parts", [(2, 1, 1), (2, 1, 3), (1, 0, 4)])
def test_rollout_and_partitions_equal_repeated_steps(policy, intra, parts):
    """ranenv_rollout(n) over batch partitions = n x ranenv_step on one stream: every state array, the raw outputs and
    the last TTI's observation / reward bit for bit (the partitions are ranges of independent envs on their own HIP
    streams; joined steps in between must keep the order with the caller's stream)."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    outs = []
    for mode in ("steps", "rollout"):
        wl = make_mult_slice_workload(100, torch.device("cuda", 0), policy=policy, intra=intra, n_scenarios=8, n_traces=10, trace_len=9,
                                      max_steps=60)
        env = wl.env
        if mode == "rollout":
            env.set_partitions(parts)
        env.reset()
        if mode == "steps":
            for t in range(13 + 1 + 25):
                env.step()
            mid = None
        else:
            env.rollout(13)
            env.step()                                        # a joined step between two rollouts
            mid = env.views()["step_number"].clone()           # read on the caller's stream: ordered behind the partitions
            env.rollout(25)
        v = comparable_views(wl)
        outs.append((v, env.obs_inter.clone(), env.obs_intra.clone(), env.reward.clone(), env.done.clone(), mid))
        env.close()
    (va, oia, oaa, ra, da, _), (vb, oib, oab, rb, db, mid) = outs
    assert int(mid.min()) == 14 and int(mid.max()) == 14
    for k in va:
        assert torch.equal(va[k], vb[k]), k
    assert torch.equal(oia, oib) and torch.equal(oaa, oab) and torch.equal(ra, rb) and torch.equal(da, db)


def test_partitioned_steps_with_external_inputs_vs_oracle():
    """Partitions under the ordinary step(): inputs produced on the caller's stream right before the call, outputs
    consumed right after it; masked reset and the head kernel go through the same partitioned launch."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    from oracle import pyoracle
    B, steps = 24, 10
    wl = make_mult_slice_workload(B, torch.device("cuda", 0), policy=0, intra=255, n_scenarios=6, n_traces=8, trace_len=10,
                                  n_slices=5, n_ues=25, n_rbs=135, rbs_per_rbg=5, max_ues_slice=10, max_steps=steps)
    env, tabs = wl.env, wl.tables
    env.set_partitions(5)
    S, U, R = env.S, env.U, env.R
    cfg = pyoracle.make_cfg(S, U, R, env.G, env.Us, max_steps=steps)
    se_host = wl.se_pool.transpose(1, 2).contiguous().cpu().numpy()
    trf_host = wl.traffic_pool.cpu().numpy().astype(np.float64)
    eps, L = env.episodes, wl.trace_len
    oenvs = []
    for b in range(B):
        o = pyoracle.OracleEnv(cfg); o.set_scenario(tabs, int(wl.scenario[b]))
        o.reset(se_host[int(eps["se_base"][b] + eps["se_offset"][b] % L)]); oenvs.append(o)
    env.reset()
    g = torch.Generator(device=env.device); g.manual_seed(5)
    for t in range(steps):
        sc = torch.rand((B, S), generator=g, device=env.device, dtype=torch.float64) * 2 - 1     # produced on the stream
        ic = torch.randint(0, 3, (B, S), generator=g, device=env.device, dtype=torch.uint8)
        obs, rew, done = env.step(sc, ic)
        r = rew.cpu().numpy(); oi = obs["obs_inter"].cpu().numpy()                                 # consumed right away
        scn, icn = sc.cpu().numpy(), ic.cpu().numpy()
        for b, o in enumerate(oenvs):
            tile = int(eps["se_base"][b] + (eps["se_offset"][b] + t) % L)
            row = int(eps["trf_base"][b] + (eps["trf_offset"][b] + t) % L)
            o.step(scn[b], icn[b].astype(np.int32), se_host[tile], trf_host[row])
            oo = o.obs()
            np.testing.assert_allclose(oi[b], oo["obs_inter"], rtol=0, atol=OBS_TOL)
            np.testing.assert_allclose(r[b], oo["reward"], rtol=0, atol=REW_TOL)
    assert bool(done.all())
    mask = (np.arange(B) % 2).astype(np.uint8)
    env.reset(env_mask=mask)
    sn = env.views()["step_number"].cpu().numpy()
    assert np.array_equal(sn == 0, mask == 1)
    env.close()


def test_full_batch_episode_rollout_over_partitions_equals_single_stream_steps():
    """BASELINE configs[2] at full size for a whole 1000-TTI episode: four ranenv_rollout(250) calls over 3 batch
    partitions (the schedule bench.py times) leave exactly the state 1000 single-stream step() calls leave."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd import _lib
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    dev = torch.device("cuda", 0)
    finals = []
    for mode in ("steps", "rollout"):
        wl = make_mult_slice_workload(4096, dev, policy=_lib.POLICY_MAPF, intra=_lib.INTRA_PF, n_traces=40, trace_len=50)
        env = wl.env
        env.reset()
        if mode == "steps":
            for _ in range(1000):
                env.step()
        else:
            env.set_partitions(3)
            for _ in range(4):
                env.rollout(250)
        torch.cuda.synchronize()
        finals.append((comparable_views(wl), env.obs_inter.clone(), env.obs_intra.clone(),
                       env.reward.clone(), env.done.clone()))
        env.close()
    (va, oia, oaa, ra, da), (vb, oib, oab, rb, db) = finals
    for k in va:
        assert torch.equal(va[k], vb[k]), k
    assert torch.equal(oia, oib) and torch.equal(oaa, oab) and torch.equal(ra, rb) and torch.equal(da, db)
    assert bool(db.all()) and int(vb["step_number"].min()) == 1000


# ------------------------------------------------------------------------------------------------------------------
# The headline schedule itself against the oracle at full size: BASELINE configs[2] at B = 4096,
# three batch partitions on three HIP streams, ranenv_rollout with its default launch lengths (several TTIs per launch,
# staggered first launches, warm entries), compact steps on.  A sample of envs -- from every partition, with at most 64
# and with more than 64 UEs in slices (one wave / two waves per env in a compact step) -- is mirrored by the CPU oracle
# and compared after every rollout call: simu.py:555-566 (the MAPF loop), agents/mapf.py:41-111, agents/common.py:558-636.
# ------------------------------------------------------------------------------------------------------------------
ROLLOUT_CALLS = (1, 3, 10, 7, 20, 4, 15, 2, 9)            # 71 TTIs: launch boundaries fall differently in every call


def _headline_sample(wl, per_class=6):
    """Env indices: from each of the 3 partitions `per_class` envs whose scenario has <= 64 slice members and
    `per_class` with > 64, the partitions' first and last envs included."""
    B = wl.env.B
    members = (wl.tables.ue_slice[wl.scenario] >= 0).sum(axis=1)
    base, rem = divmod(B, 3)
    lo = [0]
    for k in range(3):
        lo.append(lo[-1] + base + (1 if k < rem else 0))
    sample = []
    for k in range(3):
        idx = np.arange(lo[k], lo[k + 1])
        small, big = idx[members[idx] <= 64], idx[members[idx] > 64]
        assert len(small) >= per_class and len(big) >= per_class
        pick = lambda a: a[np.unique(np.linspace(0, len(a) - 1, per_class).astype(int))]
        sample += pick(small).tolist() + pick(big).tolist() + [int(idx[0]), int(idx[-1])]
    return sorted(set(sample)), members


def _compare_with_oracle(env, oenvs, max_pkts, where):
    """views (raw outputs, queue lengths), both observations and the rewards of the mirrored envs; `max_pkts[b]`: [U]."""
    g = {k: x.cpu().numpy() for k, x in env.views().items()}
    oi, oa, rw = env.obs_inter.cpu().numpy(), env.obs_intra.cpu().numpy(), env.reward.cpu().numpy()
    for b, o in oenvs.items():
        raw, oo = o.raw(), o.obs()
        for name in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts"):
            assert np.array_equal(g[name][b].astype(np.float64), raw[name]), (where, b, name)
        mp = max_pkts[b].astype(np.float64)
        assert np.array_equal(g["queue_pkts"][b].astype(np.float64), np.rint(raw["buffer_occupancies"] * mp)), (where, b, "queue_pkts")
        np.testing.assert_allclose(oi[b], oo["obs_inter"], rtol=0, atol=OBS_TOL, err_msg=str((where, b)))
        np.testing.assert_allclose(oa[b].reshape(-1), np.asarray(oo["obs_intra"]).reshape(-1), rtol=0, atol=OBS_TOL, err_msg=str((where, b)))
        np.testing.assert_allclose(rw[b], oo["reward"], rtol=0, atol=REW_TOL, err_msg=str((where, b)))


def _partition_bounds(B, parts, unit=1):
    """ranenv_set_partitions' cut of the batch (an even batch into even ranges where possible: unit 2)."""
    base, rem = divmod(B // unit, parts)
    lo = [0]
    for k in range(parts):
        lo.append(lo[-1] + unit * (base + (1 if k < rem else 0)))
    return lo


def _mirror_rollouts_with_the_oracle(wl, sample, calls, se_mode, where, after_call=None, traffic_of=None):
    """The bench schedule of `wl` (whatever its options select) against the CPU oracle: the envs in `sample` are mirrored by
    pyoracle.OracleEnv under the workload's own device policy (MARR agents/marr.py:40-47 / MAPF agents/mapf.py:41-111, scores from the
    oracle's own state, simu.py:555-566) and intra-slice scheduler (agents/common.py:508-636) and compared after every
    ranenv_rollout call of `calls`: integers exact, observations 1e-5, rewards 1e-9.  after_call(k): extra assertions on the launch."""
    from oracle import pyoracle
    env = wl.env
    S, U, R = env.S, env.U, env.R
    cfg = pyoracle.make_cfg(S, U, R, env.G, env.Us, bandwidth_hz=env.bandwidth_hz, max_age_cap=env.max_age_cap, max_steps=env.max_steps)
    eps, L = env.episodes, wl.trace_len
    tiles = sorted({int(eps["se_base"][b] + (eps["se_offset"][b] + t) % L) for b in sample for t in range(sum(calls) + 1)})
    tpos = {t: i for i, t in enumerate(tiles)}
    se_host = wl.se_pool[torch.as_tensor(tiles, device=env.device)].transpose(1, 2).contiguous().cpu().numpy()   # oracle layout [U][R]
    trf_host = wl.traffic_pool.cpu().numpy().astype(np.float64)
    intra = np.full(S, wl.intra, dtype=np.int32)
    oenvs = {}
    for b in sample:
        o = pyoracle.OracleEnv(cfg)
        o.set_scenario(wl.tables, int(wl.scenario[b]))
        o.reset(se_host[tpos[int(eps["se_base"][b] + eps["se_offset"][b] % L)]])
        oenvs[b] = o
    env.reset()
    max_pkts = {b: np.asarray(wl.tables.ue_max_pkts)[int(wl.scenario[b])] for b in sample}
    score = (lambda o: o.policy_mapf()) if wl.policy == 2 else (lambda o: o.policy_marr())
    t = 0
    for k in calls:
        env.rollout(k)
        if after_call is not None:
            after_call(k)
        for _ in range(k):
            for b, o in oenvs.items():
                bits = traffic_of(b, t) if traffic_of is not None else trf_host[int(eps["trf_base"][b] + (eps["trf_offset"][b] + t) % L)]
                o.step(score(o), intra, se_host[tpos[int(eps["se_base"][b] + (eps["se_offset"][b] + t) % L)]], bits)
            t += 1
        torch.cuda.synchronize()
        _compare_with_oracle(env, oenvs, max_pkts, (where, se_mode, "after TTI", t))
    assert int(env.views()["step_number"].min()) == t == sum(calls)


@pytest.mark.parametrize("se_mode", ["stream", "gather"])
def test_config2_headline_schedule_vs_oracle(se_mode):
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
    wl, label = make_bench_workload(2, torch.device("cuda", 0), n_traces=64, trace_len=80)
    env = wl.env
    assert env.B == 4096 and "MAPF" in label and wl.policy == 2 and wl.intra == 1
    env.set_se_mode(se_mode)
    env.set_partitions(3)
    env.set_option("compact", 1); env.set_option("fuse", 0)      # the headline's settings, whatever knob the suite runs under
    for i in range(3):
        env.set_option(f"fuse_first{i}", 0)
    sample, members = _headline_sample(wl)
    assert len(sample) >= 32 and (members[sample] <= 64).sum() >= 12 and (members[sample] > 64).sum() >= 12

    def schedule(k):
        # the auto rule (include/ranenv.h, option persist = -1): the gather mode always runs persistent launches at this size, the
        # streaming kernel for rollouts of 4...64 TTIs -- the driver's blocks of 20 -- and launch-per-chunk otherwise
        if env.get_option("persist") == -1:
            assert env.get_option("last_rollout_persistent") == (1 if se_mode == "gather" or 4 <= k <= 64 else 0), (se_mode, k)
    _mirror_rollouts_with_the_oracle(wl, sample, ROLLOUT_CALLS, se_mode, "configs[2]", after_call=schedule)
    env.close()


def test_config2_headline_schedule_with_the_device_traffic_generator_vs_oracle():
    """`bench.py --traffic philox`: the headline schedule with the offered traffic drawn on the device (Poisson by table inversion of
    Philox-4x32-10 keyed (seed; env, episode, TTI, UE), traffics/mult_slice.py:24-32 in distribution); the oracle is fed the numpy
    restatement's draws for the mirrored envs."""
    _need_gpu()
    from oracle import pyoracle
    from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
    wl, _ = make_bench_workload(2, torch.device("cuda", 0), n_traces=64, trace_len=80, traffic="philox")
    env = wl.env
    env.set_partitions(3)
    _bench_options(env)
    sample, members = _headline_sample(wl, per_class=4)
    cdf, _ = env.poisson_tables()
    seed = 1234                                                    # make_bench_workload: set_traffic_generator(seed=1234 + rank), env ids from 0
    _mirror_rollouts_with_the_oracle(wl, sample, ROLLOUT_CALLS[:6], "stream", "configs[2] philox",
                                     traffic_of=lambda b, t: pyoracle.generator_traffic(cdf, wl.tables, int(wl.scenario[b]), seed, b, 0, t))
    env.close()


# ------------------------------------------------------------------------------------------------------------------
# The OTHER benchmarked schedules against the oracle at their bench size: until round 5 they met the oracle
# in small batches only and their bench-size runs were compared with another build of the same kernel.
# ------------------------------------------------------------------------------------------------------------------
def _bench_options(env):
    """the defaults bench.py runs under, whatever knob the suite runs under"""
    for k, v in (("compact", 1), ("fuse", 0), ("persist", -1), ("persist_chunk", 10), ("persist_grid", 0), ("pack", 1), ("mix", 1)):
        env.set_option(k, v)
    for i in range(3):
        env.set_option(f"fuse_first{i}", 0)


@pytest.mark.parametrize("se_mode", ["stream", "gather"])
def test_config1_bench_schedule_vs_oracle(se_mode):
    """BASELINE configs[1] (B 1024, MARR + round-robin) as bench.py runs it: a batch at <= 2 waves per SIMD, so ranenv_rollout is ONE
    persistent launch of one chunk -- streaming: ranenv_persist_kernel_tiny, the whole SE row in flight and the next TTI's tile
    requested a TTI ahead.  The test fails if the auto rule stops selecting that launch."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
    wl, label = make_bench_workload(1, torch.device("cuda", 0), n_traces=64, trace_len=80)
    env = wl.env
    assert env.B == 1024 and "round-robin" in label and wl.policy == 1 and wl.intra == 0
    env.set_se_mode(se_mode)
    _bench_options(env)
    members = (wl.tables.ue_slice[wl.scenario] >= 0).sum(axis=1)
    small, big = np.flatnonzero(members <= 64), np.flatnonzero(members > 64)
    sample = sorted(set(small[np.linspace(0, len(small) - 1, 18).astype(int)].tolist() + big[np.linspace(0, len(big) - 1, 18).astype(int)].tolist()
                        + [0, 1023]))
    assert len(sample) >= 32

    launches = []

    def one_persistent_launch(k):
        assert env.get_option("last_rollout_persistent") == 1, "configs[1] no longer runs the persistent launch"
        launches.append(env.get_option("last_rollout_launches"))
        assert launches[-1] == 1, launches              # one class, one launch for all k TTIs
    _mirror_rollouts_with_the_oracle(wl, sample, ROLLOUT_CALLS, se_mode, "configs[1]", after_call=one_persistent_launch)
    # ... and by the dispatch's own bookkeeping: one launch covered all the TTIs of a call
    env.profile_begin(); env.rollout(13); kms = env.profile_end()
    assert kms["n_launches"] == 1 and kms["n_ttis"] == 13 and kms["n_env_ttis"] == 13 * 1024, kms
    env.close()


@pytest.mark.parametrize("se_mode", ["stream", "gather"])
def test_config4_bench_schedule_vs_oracle(se_mode):
    """BASELINE configs[4] (mult_slice_seq sweep, B 8192, mixed active-slice masks) as bench.py runs it: three partitions, launches of up
    to 10 TTIs (streaming) / persistent launches per class (gather).  Mirrored: an env of each of the 10 scenario groups from every
    partition, and the partitions' first and last envs."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
    wl, label = make_bench_workload(4, torch.device("cuda", 0), n_traces=64, trace_len=80)
    env = wl.env
    assert env.B == 8192 and "mult_slice_seq" in label
    env.set_se_mode(se_mode)
    env.set_partitions(3)
    _bench_options(env)
    lo = _partition_bounds(8192, 3, unit=2)
    sample = []
    for k in range(3):
        idx = np.arange(lo[k], lo[k + 1])
        for g in range(10):
            sample.append(int(idx[wl.scenario[idx] == g][k]))          # (a different env of the group in every partition)
        sample += [int(idx[0]), int(idx[-1])]
    sample = sorted(set(sample))
    assert len(sample) >= 32 and set(wl.scenario[sample].tolist()) == set(range(10))
    _mirror_rollouts_with_the_oracle(wl, sample, ROLLOUT_CALLS, se_mode, "configs[4]")
    assert env.get_option("last_rollout_persistent") == (1 if se_mode == "gather" else 0)
    env.close()


def _native_sample(B, parts):
    """both envs of several packed waves (pairs 2i, 2i + 1) of every partition, the partitions' edges included"""
    lo = _partition_bounds(B, parts, unit=2)
    sample = []
    for k in range(parts):
        n = lo[k + 1] - lo[k]
        for off in (0, 2, n // 3 & ~1, n // 2 & ~1, n - 4, n - 2):
            sample += [lo[k] + off, lo[k] + off + 1]
    return sorted(set(sample)), lo


@pytest.mark.parametrize("se_mode", ["stream", "gather"])
def test_native_size_bench_schedule_vs_oracle(se_mode):
    """The reference's own size (env_config/mult_slice.yml:2-14, agents/ib_sched.py:50,56: what every reference agent trains at) as
    bench.py --config native runs it: B 16 384, three partitions, TWO envs per wave (ranenv_core_kernel_packed), MAPF + PF."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
    wl, label = make_bench_workload(5, torch.device("cuda", 0), n_traces=64, trace_len=80)
    env = wl.env
    assert env.B == 16384 and (env.S, env.U, env.R, env.G) == (5, 25, 135, 5)
    env.set_se_mode(se_mode)
    env.set_partitions(3)
    _bench_options(env)
    sample, lo = _native_sample(16384, 3)
    assert lo == [0, 5462, 10924, 16384] and len(sample) >= 32
    _mirror_rollouts_with_the_oracle(wl, sample, ROLLOUT_CALLS, se_mode, "native")
    assert env.get_option("last_rollout_persistent") == 0 and env.get_option("pack") == 1
    env.close()


@pytest.mark.parametrize("config", [2, 5])
def test_bench_schedule_across_an_episode_end_vs_oracle(config):
    """The headline schedule (configs[2]) and the native-size one (two envs per wave) with device auto-reset: every env's episode (37
    TTIs here) ends inside the rollouts, the advance + RESET launches follow that TTI on the partition's own stream, the fused
    launches end there."""
    _need_gpu()
    from oracle import pyoracle
    from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
    wl, _ = make_bench_workload(config, torch.device("cuda", 0), n_traces=64, trace_len=80)
    env, tabs = wl.env, wl.tables
    env.set_partitions(3)
    _bench_options(env)
    S, U, R, L, B = env.S, env.U, env.R, wl.trace_len, env.B
    n_ep, first, ep_len = 50, 0, 37
    ep_no = np.arange(first, first + n_ep)
    env.set_episode_table(scenario=(ep_no * 7) % tabs.n_scenarios, se_base=(ep_no % 64) * L, se_len=L, se_offset=(ep_no * 11) % L,
                          trf_base=((ep_no * 7) % tabs.n_scenarios) * L, trf_len=L, trf_offset=(ep_no * 3) % L, first_episode=first)
    env.set_max_steps(np.full(B, ep_len, dtype=np.int32))
    start = first + np.arange(B) % n_ep
    env.enable_autoreset(first, first + n_ep, random_episodes=False, seed=1, episode_numbers=start)
    tab = env.episode_table
    scen_of = lambda ep: int(tab[ep - first]["scenario"])
    members = np.array([(tabs.ue_slice[scen_of(int(ep))] >= 0).sum() for ep in start])
    if config == 5:
        sample, _ = _native_sample(B, 3)
    else:
        lo = _partition_bounds(B, 3, unit=2)
        sample = []
        for k in range(3):
            idx = np.arange(lo[k], lo[k + 1])
            sample += idx[members[idx] <= 64][:5].tolist() + idx[members[idx] > 64][:5].tolist() + [int(idx[-1])]
    cfg = pyoracle.make_cfg(S, U, R, env.G, env.Us, bandwidth_hz=env.bandwidth_hz, max_age_cap=env.max_age_cap, max_steps=10 ** 6)
    se_pool, trf_host = wl.se_pool, wl.traffic_pool.cpu().numpy().astype(np.float64)
    tile_cache = {}

    def tile(ep, t):
        r = tab[ep - first]
        i = int(r["se_base"] + (r["se_offset"] + t) % r["se_len"])
        if i not in tile_cache:
            tile_cache[i] = se_pool[i].transpose(0, 1).contiguous().cpu().numpy()
        return tile_cache[i]

    def trow(ep, t): r = tab[ep - first]; return int(r["trf_base"] + (r["trf_offset"] + t) % r["trf_len"])
    intra = np.full(S, wl.intra, dtype=np.int32)
    oenvs, cur, tstep = {}, {}, {}
    for b in sample:
        o = pyoracle.OracleEnv(cfg); o.set_scenario(tabs, scen_of(int(start[b]))); o.reset(tile(int(start[b]), 0))
        oenvs[b], cur[b], tstep[b] = o, int(start[b]), 0
    env.reset()
    total = 0
    for k in (10, 20, 9, 30, 12):                      # 81 TTIs: two episode ends per env (TTIs 37 and 74)
        env.rollout(k)
        torch.cuda.synchronize()
        total += k
        g = {n: x.cpu().numpy() for n, x in env.views().items()}
        oi, rw = env.obs_inter.cpu().numpy(), env.reward.cpu().numpy()
        for b, o in oenvs.items():
            for _ in range(k):
                o.step(o.policy_mapf(), intra, tile(cur[b], tstep[b]), trf_host[trow(cur[b], tstep[b])])
                tstep[b] += 1
                last = o.obs()
                if tstep[b] >= ep_len:
                    cur[b] = cur[b] + 1 if cur[b] + 1 < first + n_ep else first
                    tstep[b] = 0
                    o.set_scenario(tabs, scen_of(cur[b])); o.reset(tile(cur[b], 0))
            np.testing.assert_allclose(rw[b], last["reward"], rtol=0, atol=REW_TOL, err_msg=str((total, b)))
            assert int(g["episode_number"][b]) == cur[b] and int(g["step_number"][b]) == tstep[b], (total, b)
            np.testing.assert_allclose(oi[b], o.obs()["obs_inter"], rtol=0, atol=OBS_TOL, err_msg=str((total, b)))
            raw = o.raw()
            mp = np.asarray(tabs.ue_max_pkts)[scen_of(cur[b])].astype(np.float64)
            assert np.array_equal(g["queue_pkts"][b].astype(np.float64), np.rint(raw["buffer_occupancies"] * mp)), (total, b)
            if tstep[b] > 0:
                for name in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts"):
                    assert np.array_equal(g[name][b].astype(np.float64), raw[name]), (total, b, name)
    env.close()
