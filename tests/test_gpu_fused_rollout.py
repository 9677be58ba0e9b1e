"""ranenv_rollout's launches that take their envs through several TTIs (step_loop in csrc/ranenv_step_body.hpp; include/ranenv.h
"ranenv_rollout"): bit for bit what one launch per TTI leaves behind -- state, observations, rewards, done flags, episode
metrics -- with and without partitions, in both SE modes, across device auto-resets (launches end at the TTI at which an
episode of the batch ends), with per-env episode lengths, with the traffic drawn on the device.  (Against the oracle the
rollout is checked elsewhere, e.g. tests/test_gpu_rollout_and_partitions.py, which now runs fused by default.)
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from tests.test_gpu_se_gather_and_ranges import _bench_like, _short_episode_setup

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _same(a, b, what=""):
    for k, x in a.views().items():
        assert torch.equal(x, b.views()[k]), (what, k)
    assert torch.equal(a.obs_inter, b.obs_inter) and torch.equal(a.obs_intra, b.obs_intra), what
    assert torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done), what


@pytest.mark.parametrize("se_mode", ["stream", "gather"])
@pytest.mark.parametrize("parts", [1, 3])
def test_fused_rollouts_equal_one_launch_per_tti(monkeypatch, se_mode, parts):
    """BASELINE configs[2]'s shape, 512 envs: rollouts of 1, 2, 7, 23 and 45 TTIs (launches of up to 1, 1, 1, 5 and 10
    TTIs, the last one of a rollout shorter) against the same rollouts with RANENV_FUSE=1."""
    _need_gpu()
    monkeypatch.delenv("RANENV_FUSE", raising=False)          # (`a` runs the default policy whatever knob the suite runs under)
    a = _bench_like(512, se_mode == "gather")
    monkeypatch.setenv("RANENV_FUSE", "1")
    b = _bench_like(512, se_mode == "gather")
    monkeypatch.delenv("RANENV_FUSE")
    for wl in (a, b):
        wl.env.enable_metrics(0)
        wl.env.reset()
        wl.env.set_partitions(parts)
    for K in (1, 2, 7, 23, 45):
        a.env.rollout(K); b.env.rollout(K)
        torch.cuda.synchronize()
        _same(a.env, b.env, (se_mode, parts, K))
        assert torch.equal(a.env.episode_metrics()["running"], b.env.episode_metrics()["running"]), K
    assert int(a.env.views()["step_number"][0]) == 1 + 2 + 7 + 23 + 45
    a.env.close(); b.env.close()


def test_a_fused_launch_covers_the_ttis_it_says(monkeypatch):
    """The profile counters: a rollout of 40 TTIs over 3 partitions is launches of up to 10 TTIs, the partitions' first ones
    of different lengths; with RANENV_FUSE=1 it is 3 x 40 launches of one."""
    _need_gpu()
    monkeypatch.delenv("RANENV_FUSE", raising=False)          # (the default policy, whatever knob the suite runs under)
    monkeypatch.delenv("RANENV_FUSE_FIRST", raising=False)
    a = _bench_like(96, False)
    a.env.set_option("persist", 0)                            # (a batch this small would run as one persistent launch: below)
    a.env.reset(); a.env.set_partitions(3)
    # first launches of 10 / 6 / 1 TTIs (partition 0 / 1 / 2), then 10s: 10+10+10+10, 6+10+10+10+4, 1+10+10+10+9
    a.env.profile_begin(); a.env.rollout(40); pa = a.env.profile_end()
    assert (pa["n_launches"], pa["n_ttis"]) == (14, 120)
    a.env.profile_begin(); a.env.rollout(23); pa = a.env.profile_end()          # 5+5+5+5+3, 3+5+5+5+5, 1+5+5+5+5+2
    assert (pa["n_launches"], pa["n_ttis"]) == (16, 69)
    a.env.profile_begin(); a.env.step(); pa = a.env.profile_end()
    assert (pa["n_launches"], pa["n_ttis"]) == (3, 3)
    # the persistent rollout: one launch for all 40 TTIs of all 96 envs (a batch far below what the chip holds is one class)
    a.env.set_option("persist", -1); a.env.set_option("mix", 1)          # (mix = 2, a test setting, keeps two classes for small batches too)
    a.env.profile_begin(); a.env.rollout(40); pa = a.env.profile_end()
    if a.env.get_option("compact"):                           # (it needs compact steps: not under the RANENV_COMPACT=0 pass of the suite)
        assert (pa["n_launches"], pa["n_ttis"], pa["n_env_ttis"]) == (1, 40, 96 * 40)
    a.env.close()
    monkeypatch.setenv("RANENV_FUSE", "1")
    b = _bench_like(96, False)
    b.env.set_option("persist", 0)
    b.env.reset(); b.env.set_partitions(3)
    b.env.profile_begin(); b.env.rollout(40); pb = b.env.profile_end()
    assert (pb["n_launches"], pb["n_ttis"]) == (120, 120)
    b.env.close()


@pytest.mark.parametrize("se_mode", ["stream", "gather"])
@pytest.mark.parametrize("staggered", [False, True])
def test_fused_rollouts_across_device_autoresets(monkeypatch, se_mode, staggered):
    """Episodes of 13 TTIs with the scenario changing at every reset.  In lock step the launches run up to the TTI at which
    the episodes end (13 = 10 + 3 inside a rollout of 40 ...), the advance + reset launches follow, the next launch starts
    the new episodes; with per-env episode lengths between 5 and 13 some episode ends almost every TTI and the launches
    shrink accordingly.  Same state, observations, episode numbers and per-episode metric sums as one launch per TTI."""
    _need_gpu()
    monkeypatch.delenv("RANENV_FUSE", raising=False)
    envs = []
    for fuse in (None, "1"):
        if fuse:
            monkeypatch.setenv("RANENV_FUSE", fuse)
        env, tabs, se_pool, trf, start, dims = _short_episode_setup(48, 13, False, se_mode)
        if fuse:
            monkeypatch.delenv("RANENV_FUSE")
        if staggered:
            env.set_max_steps(5 + (np.arange(48) * 7) % 9)
        env.enable_metrics(8)
        env.reset()
        env.set_partitions(2)
        envs.append(env)
    a, b = envs
    n_launch = []
    for K in (40, 9, 31):
        a.profile_begin(); a.rollout(K); pa = a.profile_end()
        b.rollout(K)
        torch.cuda.synchronize()
        n_launch.append(pa["n_launches"])
        for name, x in a.views().items():
            if name != "se_mean":            # (compact steps: not kept up for UEs outside every slice)
                assert torch.equal(x, b.views()[name]), (K, name)
        assert torch.equal(a.obs_inter, b.obs_inter) and torch.equal(a.obs_intra, b.obs_intra) and torch.equal(a.reward, b.reward), K
        ma, mb = a.episode_metrics(), b.episode_metrics()
        for k in ("running", "episode_log", "episodes_done"):
            assert torch.equal(ma[k], mb[k]), (K, k)
    if not staggered:
        # step launches only (the reset launches are timed too): 40 TTIs from step 0 = 10 + 3 | 10 + 3 | 10 + 3 | 1
        assert n_launch[0] < 2 * 40
    assert int(a.views()["episode_number"].max()) >= 5
    a.close(); b.close()


def test_fused_rollout_with_the_traffic_drawn_on_the_device(monkeypatch):
    """The Philox / Poisson generator is keyed by (env, episode, TTI, UE): a launch that runs several TTIs draws what the
    same TTIs draw one launch at a time."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_bench_workload
    monkeypatch.delenv("RANENV_FUSE", raising=False)
    envs = []
    for fuse in (None, "1"):
        if fuse:
            monkeypatch.setenv("RANENV_FUSE", fuse)
        wl, _ = make_bench_workload(2, torch.device("cuda", 0), batch=256, n_traces=8, trace_len=30, traffic="philox")
        if fuse:
            monkeypatch.delenv("RANENV_FUSE")
        wl.env.reset(); wl.env.set_partitions(3)
        envs.append(wl.env)
    a, b = envs
    a.rollout(48); b.rollout(48)
    torch.cuda.synchronize()
    _same(a, b, "philox")
    assert int(a.views()["pkt_incoming"].sum()) > 0
    a.close(); b.close()


# ---------------------------------------------------------------------------------------------- persistent rollout
def _persist_pair(B, se_mode, setup, **opts):
    """Two identical workloads: `a` with the default rollout schedule, `b` with option persist (+ opts)."""
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    out = []
    for persist in (0, 1):
        wl = make_mult_slice_workload(B, torch.device("cuda", 0), n_scenarios=32, n_traces=16, trace_len=24, max_steps=1000)
        wl.env.set_se_mode(se_mode)
        setup(wl)
        wl.env.set_option("persist", persist)
        if persist:
            for k, v in opts.items():
                wl.env.set_option(k, v)
        out.append(wl)
    return out


def _same_state(a, b, where):
    va, vb = a.env.views(), b.env.views()
    scen = va["episodes"][:, 0].to(torch.int64)                 # as on the device: auto-reset may have moved on
    in_slice = torch.as_tensor(a.tables.ue_slice >= 0, device=a.env.device)[scen]
    for k in va:
        if k == "se_mean":                                      # of UEs outside every slice: not kept up by compact steps
            assert torch.equal(va[k][in_slice], vb[k][in_slice]), (where, k)
        else:
            assert torch.equal(va[k], vb[k]), (where, k)
    assert torch.equal(a.env.obs_inter, b.env.obs_inter) and torch.equal(a.env.obs_intra, b.env.obs_intra), where
    assert torch.equal(a.env.reward, b.env.reward) and torch.equal(a.env.done, b.env.done), where


@pytest.mark.parametrize("se_mode", ["stream", "gather"])
@pytest.mark.parametrize("B,opts", [(300, {}),                                           # everything resident: nobody ever waits
                                    (300, {"persist_grid": 96, "persist_chunk": 3}),      # ~4 envs per workgroup: hand-overs all the time
                                    (4096, {}),                                           # the headline batch on the real grid
                                    (4096, {"persist_grid": 1500, "persist_chunk": 2}),
                                    (1000, {"persist_grid": 8, "persist_chunk": 1})])     # a handful of workgroups, one per XCD or fewer
def test_persistent_rollout_equals_the_launch_per_chunk_rollout(se_mode, B, opts):
    """Option persist: one persistent work-queue launch per workgroup class (one wave / two waves per env) instead of launches of
    <= 10 TTIs per partition.  Same numbers bit for bit after every call, whatever the grid: with a grid far smaller than the
    batch every chunk of every env is handed from one workgroup to another through the per-XCD ready queues (stale-L1 and
    cross-XCD hazards would show up as wrong state here), with the default grid the queues are hardly touched."""
    _need_gpu()
    a, b = _persist_pair(B, se_mode, lambda wl: wl.env.set_partitions(3), **opts)
    a.env.reset(); b.env.reset()
    t = 0
    for k in (1, 7, 23, 10, 40, 3):
        a.env.rollout(k); b.env.rollout(k)
        torch.cuda.synchronize()
        t += k
        _same_state(a, b, (se_mode, B, t))
    a.env.step(); b.env.step()                              # a joined step behind a persistent rollout, then another rollout
    a.env.rollout(12); b.env.rollout(12)
    torch.cuda.synchronize()
    _same_state(a, b, (se_mode, B, "tail"))
    assert b.env.get_option("persist_errors") == 0
    a.env.close(); b.env.close()


@pytest.mark.parametrize("se_mode", ["stream", "gather"])
def test_persistent_rollout_through_episode_ends(se_mode):
    """With device auto-reset a persistent launch ends at the TTI at which the first episode of the batch ends; the advance +
    RESET launches follow and the envs are sorted into classes again (their scenarios changed).  Per-env episode lengths."""
    _need_gpu()
    B, n_ep = 600, 40

    def setup(wl):
        env = wl.env
        ep = np.arange(n_ep)
        env.set_episode_table(scenario=(ep * 5) % 32, se_base=(ep % 16) * 24, se_len=24, se_offset=ep % 24,
                              trf_base=((ep * 5) % 32) * 24, trf_len=24, trf_offset=(ep * 7) % 24)
        env.set_max_steps(17 + (np.arange(B) % 5) * 6)
        env.enable_autoreset(0, n_ep, episode_numbers=np.arange(B) % n_ep)
        env.enable_metrics(8)
    a, b = _persist_pair(B, se_mode, setup, persist_grid=256, persist_chunk=4)
    a.env.reset(); b.env.reset()
    for k in (30, 9, 50, 21):
        a.env.rollout(k); b.env.rollout(k)
        torch.cuda.synchronize()
        _same_state(a, b, (se_mode, k))
        ma, mb = a.env.episode_metrics(), b.env.episode_metrics()
        assert torch.equal(ma["episodes_done"], mb["episodes_done"]) and torch.equal(ma["episode_log"], mb["episode_log"])
    assert int(a.env.episode_metrics()["episodes_done"].min()) >= 2
    assert b.env.get_option("persist_errors") == 0
    a.env.close(); b.env.close()


# ---------------------------------------------------------------------------------------------- packed waves
@pytest.mark.parametrize("se_mode", ["stream", "gather"])
@pytest.mark.parametrize("B", [2, 64, 4098])
def test_two_envs_per_wave_equal_one_env_per_wave_at_the_reference_size(se_mode, B):
    """Option pack (default on): envs of at most 32 UEs / 8 slices -- S 5, U 25, 27 RBGs of 5: the reference's own size -- are
    stepped two per wave (lanes 0-31 / 32-63, ranenv_core_kernel_packed).  Same numbers bit for bit as one env per wave: rollouts
    over 3 partitions (even ranges), single steps, external scores + per-slice intra choice, device auto-resets that change the
    scenario, the device traffic generator."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    dev = torch.device("cuda", 0)
    n_ep = 24
    outs = []
    for pack in (1, 0):
        wl = make_mult_slice_workload(B, dev, policy=2, intra=1, n_scenarios=24, n_traces=12, trace_len=20, n_slices=5, n_ues=25,
                                      n_rbs=135, rbs_per_rbg=5, max_ues_slice=5, max_steps=17, min_slices=3, min_ues=2)
        env = wl.env
        env.set_option("pack", pack); env.set_option("persist", 0)
        env.set_se_mode(se_mode)
        ep = np.arange(n_ep)
        env.set_episode_table(scenario=(ep * 5) % 24, se_base=(ep % 12) * 20, se_len=20, se_offset=ep % 20,
                              trf_base=((ep * 5) % 24) * 20, trf_len=20, trf_offset=(ep * 7) % 20)
        env.enable_autoreset(0, n_ep, episode_numbers=np.arange(B) % n_ep)
        env.enable_metrics(4)
        if B > 2:
            env.set_partitions(3)
        env.reset()
        snaps = []
        env.rollout(9); snaps.append(_snap(env))
        env.step(); snaps.append(_snap(env))
        env.rollout(23); snaps.append(_snap(env))                      # across the episode end at TTI 17
        g = torch.Generator(device=dev); g.manual_seed(11)
        env.set_policy(0, 255)
        for _ in range(3):
            sc = torch.rand((B, env.S), generator=g, device=dev, dtype=torch.float64) * 2 - 1
            ic = torch.randint(0, 3, (B, env.S), generator=g, device=dev, dtype=torch.uint8)
            env.step(sc, ic); snaps.append(_snap(env))
        env.set_policy(1, 0)
        env.set_traffic_generator(seed=77)
        env.rollout(12); snaps.append(_snap(env))
        m = env.episode_metrics()
        snaps.append({"done_eps": m["episodes_done"].clone(), "log": m["episode_log"].clone(), "run": m["running"].clone()})
        outs.append(snaps)
        env.close()
    for i, (x, y) in enumerate(zip(*outs)):
        for k in x:
            assert torch.equal(x[k], y[k]), (se_mode, B, i, k)


def _snap(env):
    torch.cuda.synchronize()
    d = {k: v.clone() for k, v in env.views().items() if k != "se_mean"}
    d.update(obs_inter=env.obs_inter.clone(), obs_intra=env.obs_intra.clone(), reward=env.reward.clone(), done=env.done.clone())
    return d


# ---------------------------------------------------------------------------------------------- mixed blocks
@pytest.mark.parametrize("se_mode", ["stream", "gather"])
@pytest.mark.parametrize("B", [37, 4096])
def test_mixed_blocks_equal_one_workgroup_per_env(se_mode, B):
    """Option mix: a whole-batch step launch as one block per env of more than 64 slice members and one block per TWO envs of at
    most 64 (a wave each, no block barrier between them) -- ranenv_core_kernel_mixed -- against one two-wave workgroup per env:
    single steps and multi-TTI launches (a rollout on one stream), device policy and external scores, bit for bit.  B = 37: forced
    (mix = 2), an odd number of narrow envs (the last block's second wave has none)."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    dev = torch.device("cuda", 0)
    outs = []
    for mix in (2, 0):
        wl = make_mult_slice_workload(B, dev, n_scenarios=32, n_traces=16, trace_len=24, max_steps=1000)
        env = wl.env
        env.set_option("mix", mix); env.set_option("persist", 0)
        env.set_se_mode(se_mode)
        env.reset()
        snaps = []
        for _ in range(4):
            env.step()
        snaps.append(_snap(env))
        env.rollout(13); snaps.append(_snap(env))                      # one partition: launches of several TTIs over the whole batch
        g = torch.Generator(device=dev); g.manual_seed(5)
        env.set_policy(0, 255)
        for _ in range(3):
            sc = torch.rand((B, env.S), generator=g, device=dev, dtype=torch.float64) * 2 - 1
            ic = torch.randint(0, 3, (B, env.S), generator=g, device=dev, dtype=torch.uint8)
            env.step(sc, ic)
        snaps.append(_snap(env))
        if mix:
            env.profile_begin(); env.set_policy(2, 1); env.step(); k = env.profile_end()
            members = (wl.tables.ue_slice[wl.scenario] >= 0).sum(axis=1)
            assert k["n_launches"] == 1 and k["n_env_ttis"] == B and (members <= 64).sum() > 0 and (members > 64).sum() > 0
        else:
            env.set_policy(2, 1); env.step()
        snaps.append(_snap(env))
        outs.append(snaps)
        env.close()
    for i, (x, y) in enumerate(zip(*outs)):
        for k in x:
            assert torch.equal(x[k], y[k]), (se_mode, B, i, k)
