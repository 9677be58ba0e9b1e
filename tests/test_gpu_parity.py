"""GPU parity: the HIP path (through the C ABI) against the golden fixtures and the CPU oracle.

Integer outputs (packets in/out/dropped, queue length, age sum, RB ranges) must be bit-exact;
observations are float32 on the device (compared at 1e-5 as BASELINE.json's north_star states)
and rewards float64 (compared at 1e-9).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from tests.common import TRACE_CASES, load_golden, poisson_traffic_rows, tables_from
from tests.synth import se_tile

pytestmark = pytest.mark.gpu

OBS_TOL = 1e-5
REW_TOL = 1e-9


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.fixture(params=["lean", "small", "gather", "packed", "packed-gather", "mixed", "mixed-gather"])
def build(request, monkeypatch):
    """The streaming step kernel has two builds (96 VGPRs / 8 SE loads in flight for batches that fill the CUs, 128 VGPRs /
    32 in flight for small ones); ranenv_create picks by batch size.  Test batches are small, so the choice is forced here
    (RANENV_SMALL_BATCH is read at create) and every case runs against both -- and against the SE gather mode
    (RANENV_SE_MODE=gather: BatchedRanEnv.bind_se_pool switches it on, pooled tiles are then read through the sidecars).
    "packed" / "packed-gather": option pack on (the default) -- envs of at most 32 UEs and 8 slices are stepped two per wave
    (ranenv_core_kernel_packed) whenever a launch covers an even number of them; the other builds run with RANENV_PACK=0, so
    that the one-env-per-wave kernels keep their coverage at the reference's own size.
    "mixed" / "mixed-gather": whole-batch steps of two-wave workgroups (U > 64) as mixed blocks -- one block per env of more than 64
    slice members, one per two envs of at most 64 (ranenv_core_kernel_mixed) -- forced for these small batches (RANENV_MIX=2); the
    other builds run with RANENV_MIX=0."""
    monkeypatch.setenv("RANENV_SMALL_BATCH", "0" if request.param in ("lean", "packed", "packed-gather", "mixed", "mixed-gather") else "1")
    monkeypatch.setenv("RANENV_PACK", "1" if request.param.startswith("packed") else "0")
    monkeypatch.setenv("RANENV_MIX", "2" if request.param.startswith("mixed") else "0")
    if request.param.endswith("gather"):
        monkeypatch.setenv("RANENV_SE_MODE", "gather")
    return "gather" if request.param.endswith("gather") else request.param


def _env(**kw):
    from intent_radio_sched_multi_slice_amd.batched_env import BatchedRanEnv
    return BatchedRanEnv(**kw)


def _rb_major(se_ue_major):
    return np.ascontiguousarray(np.swapaxes(se_ue_major, -1, -2))


def _cmp_raw(env, b, raw, tag):
    v = env.views()
    for name, key in (("pkt_incoming", "pkt_incoming"), ("pkt_throughputs", "pkt_throughputs"),
                      ("pkt_effective_thr", "pkt_effective_thr"), ("dropped_pkts", "dropped_pkts")):
        got = v[name][b].cpu().numpy().astype(np.float64)
        assert np.array_equal(got, raw[key]), (tag, name, got, raw[key])


@pytest.mark.parametrize("case", TRACE_CASES)
def test_golden_traces(case, build):
    """Closed-loop traces whose agent side was produced by the reference's own code."""
    _need_gpu()
    fx = load_golden(case)
    cfg = fx["cfg"]
    S, U, R, G, Us = (int(x) for x in cfg[:5])
    seed, steps_per_ep, cap = int(cfg[5]), int(cfg[7]), int(cfg[8])
    plumbing = len(cfg) > 9 and int(cfg[9]) == 1
    tabs = tables_from(fx)
    B = 4                          # (even: the packed build steps envs 0+1 and 2+3 in one wave each)
    env = _env(batch=B, n_slices=S, n_ues=U, n_rbs=R, rbs_per_rbg=G, max_ues_slice=Us,
               n_scenarios=tabs.n_scenarios, bandwidth_hz=float(fx["bw"]), max_steps=steps_per_ep, max_age_cap=cap)
    env.load_scenarios(tabs)
    env.set_policy(0, 255)
    k = 0
    pooled = build == "gather"     # the gather mode reads pooled tiles: the episode's tiles become a trace of the pool
    for ep, idx in enumerate(fx["scen_ids"]):
        get_se = (lambda t: np.full((U, R), 2.0, dtype=np.float32)) if plumbing else \
            (lambda t: se_tile(seed + ep, t, U, R))
        if pooled:
            env.bind_se_pool(torch.as_tensor(np.stack([_rb_major(get_se(t)) for t in range(steps_per_ep)]), device=env.device))
            assert env.se_mode == "gather"
            env.set_episodes(scenario=int(idx), se_base=0, se_len=steps_per_ep)
        else:
            env.set_episodes(scenario=int(idx))
        se0 = np.broadcast_to(_rb_major(get_se(0)), (B, R, U))
        obs = env.reset(se_tiles=None if pooled else se0)
        got = np.concatenate([obs["obs_inter"][1].cpu().numpy(), obs["obs_intra"][1].cpu().numpy().ravel()])
        np.testing.assert_allclose(got, fx["reset_obs"][ep], rtol=0, atol=OBS_TOL)
        for t in range(steps_per_ep):
            se = None if pooled else np.broadcast_to(_rb_major(get_se(t)), (B, R, U))
            sc = np.broadcast_to(fx["scores"][k], (B, S))
            ic = np.broadcast_to(fx["intra"][k].astype(np.uint8), (B, S))
            tr = np.broadcast_to(fx["traffic"][k], (B, U))
            obs, rew, done = env.step(sc, ic, tr, se)
            v = env.views()
            for b in (0, 1, B - 1):
                tag = (case, ep, t, b)
                cnt = v["rb_count"][b].cpu().numpy()
                st = v["rb_start"][b].cpu().numpy()
                assert np.array_equal(cnt, fx["rb_count"][k]), tag
                used = cnt > 0
                assert np.array_equal(st[used], fx["rb_start"][k][used]), tag
                _cmp_raw(env, b, {n: fx[n][k] for n in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr",
                                                        "dropped_pkts")}, tag)
                ro = env.raw_observation()
                assert np.array_equal(ro["buffer_occupancies"][b].cpu().numpy(), fx["buffer_occupancies"][k]), tag
                assert np.array_equal(ro["buffer_latencies"][b].cpu().numpy(), fx["buffer_latencies"][k]), tag
                np.testing.assert_allclose(obs["obs_inter"][b].cpu().numpy(), fx["obs_inter"][k], rtol=0, atol=OBS_TOL)
                np.testing.assert_allclose(obs["obs_intra"][b].cpu().numpy(), fx["obs_intra"][k], rtol=0, atol=OBS_TOL)
                np.testing.assert_allclose(rew[b].cpu().numpy(), fx["reward"][k], rtol=0, atol=REW_TOL)
            assert int(done[0]) == (1 if t == steps_per_ep - 1 else 0)
            k += 1
        assert np.array_equal(v["mask_inter"][0].cpu().numpy(), fx["mask_inter"][k - 1])
        assert np.array_equal(v["mask_intra"][0].cpu().numpy(), fx["mask_intra"][k - 1])
    env.close()


def _oracle_batch(tabs, scen, S, U, R, G, Us, steps):
    from oracle import pyoracle
    cfg = pyoracle.make_cfg(S, U, R, G, Us, max_steps=steps)
    envs = []
    for b in range(len(scen)):
        e = pyoracle.OracleEnv(cfg)
        e.set_scenario(tabs, int(scen[b]))
        envs.append(e)
    return envs


@pytest.mark.parametrize("policy,intra", [(1, 0), (2, 1), (0, 255), (2, 2)])
@pytest.mark.parametrize("size", ["ref", "scaled"])
def test_batch_vs_oracle(policy, intra, size, build):
    """B envs on distinct scenarios/traces from HBM pools, device policies, against the oracle."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.scenario import generate_scaled_scenarios
    from oracle import pyoracle
    if size == "ref":
        S, U, R, G, Us, B = 5, 25, 135, 5, 5, 24
        tabs = generate_scaled_scenarios(6, seed=3, n_slices=S, n_ues=U, max_ues_slice=Us, min_slices=3, min_ues=2)
    else:
        S, U, R, G, Us, B = 10, 100, 135, 1, 10, 32
        tabs = generate_scaled_scenarios(6, seed=4)
    steps, trace_len = 36, 12
    rng = np.random.default_rng(100 + policy * 10 + intra)
    scen = rng.integers(0, tabs.n_scenarios, B)
    n_traces = 5
    se_pool = np.stack([se_tile(50 + i // trace_len, i % trace_len, U, R) for i in range(n_traces * trace_len)])
    trf_rows = np.concatenate([poisson_traffic_rows(tabs, int(scen[b]), rng, trace_len) for b in range(B)])
    trf_rows[::7] *= 5.0
    se_trace = rng.integers(0, n_traces, B)
    se_off = rng.integers(0, trace_len, B)
    trf_off = rng.integers(0, trace_len, B)
    env = _env(batch=B, n_slices=S, n_ues=U, n_rbs=R, rbs_per_rbg=G, max_ues_slice=Us,
               n_scenarios=tabs.n_scenarios, max_steps=steps)
    env.load_scenarios(tabs)
    env.bind_se_pool(torch.as_tensor(_rb_major(se_pool), device=env.device))
    env.bind_traffic_pool(torch.as_tensor(trf_rows.astype(np.int32), device=env.device))
    env.set_episodes(scenario=scen, se_base=se_trace * trace_len, se_len=trace_len, se_offset=se_off,
                     trf_base=np.arange(B) * trace_len, trf_len=trace_len, trf_offset=trf_off)
    env.set_policy(policy, intra)
    oenvs = _oracle_batch(tabs, scen, S, U, R, G, Us, steps)
    t0 = np.zeros(B, dtype=np.int64)   # global step at which env b's current episode started
    tile_of = lambda b, t: int(se_trace[b] * trace_len + (se_off[b] + t - t0[b]) % trace_len)
    row_of = lambda b, t: int(b * trace_len + (trf_off[b] + t - t0[b]) % trace_len)
    obs = env.reset()
    for b in range(B):
        oenvs[b].reset(se_pool[tile_of(b, 0)])
    for t in range(steps):
        if policy == 0:
            sc = rng.uniform(-1, 1, (B, S))
            sc[rng.random((B, S)) < 0.2] = 1.0
            ic = rng.integers(0, 3, (B, S)).astype(np.uint8)
            obs, rew, done = env.step(sc, ic)
        else:
            sc, ic = None, np.full((B, S), intra, dtype=np.uint8)
            obs, rew, done = env.step()
        v = env.views()
        ro = env.raw_observation()
        g = {k: x.cpu().numpy() for k, x in v.items()}
        gro = {k: x.cpu().numpy() for k, x in ro.items()}
        goi, goa, grw = obs["obs_inter"].cpu().numpy(), obs["obs_intra"].cpu().numpy(), rew.cpu().numpy()
        for b in range(B):
            oe = oenvs[b]
            if policy == 1:
                s_b = oe.policy_marr()
            elif policy == 2:
                s_b = oe.policy_mapf()
            else:
                s_b = sc[b]
            np.testing.assert_allclose(g["policy_scores"][b], s_b, rtol=0, atol=1e-12)
            start, count, _ = oe.action_format(s_b, ic[b], want_dense=False)
            oe.step(s_b, ic[b], se_pool[tile_of(b, t)], trf_rows[row_of(b, t)])
            tag = (size, policy, intra, t, b)
            assert np.array_equal(g["rb_count"][b], count), tag
            assert np.array_equal(g["rb_start"][b][count > 0], start[count > 0]), tag
            raw = oe.raw()
            for name in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts"):
                assert np.array_equal(g[name][b].astype(np.float64), raw[name]), (tag, name)
            assert np.array_equal(gro["buffer_occupancies"][b], raw["buffer_occupancies"]), tag
            assert np.array_equal(gro["buffer_latencies"][b], raw["buffer_latencies"]), tag
            o = oe.obs()
            np.testing.assert_allclose(goi[b], o["obs_inter"], rtol=0, atol=OBS_TOL, err_msg=str(tag))
            np.testing.assert_allclose(goa[b], o["obs_intra"], rtol=0, atol=OBS_TOL, err_msg=str(tag))
            np.testing.assert_allclose(grw[b], o["reward"], rtol=0, atol=REW_TOL, err_msg=str(tag))
        if t == 17:   # masked reset of every third env mid-episode (deque survives, buffers do not)
            mask = (np.arange(B) % 3 == 0).astype(np.uint8)
            before = obs["obs_inter"].cpu().numpy().copy()
            obs = env.reset(env_mask=mask)
            after = obs["obs_inter"].cpu().numpy()
            assert np.array_equal(after[mask == 0], before[mask == 0])
            t0[mask == 1] = t + 1
            for b in np.nonzero(mask)[0]:
                oenvs[b].reset(se_pool[tile_of(b, t + 1)])
                np.testing.assert_allclose(after[b], oenvs[b].obs()["obs_inter"], rtol=0, atol=OBS_TOL)
    env.close()


@pytest.mark.parametrize("shape", [
    dict(S=3, U=7, R=5, G=1, Us=4),        # R < 8: numpy's plain loop, single wave
    dict(S=4, U=37, R=100, G=5, Us=12),    # one leaf with a tail, U not a multiple of 4
    dict(S=16, U=128, R=300, G=3, Us=16),  # three leaves (150 -> 72+78 | 150), full slot grid
    dict(S=6, U=64, R=408, G=8, Us=11),    # four leaves, G does not divide R
    dict(S=16, U=256, R=64, G=1, Us=16),   # the largest UE count of this build: four waves per env
    dict(S=5, U=30, R=48, G=2, Us=8, D=3),  # a 3-deep observation window: the ring wraps every three TTIs
])
@pytest.mark.parametrize("variant", ["external", "device"])
def test_shapes_vs_oracle(shape, variant, build):
    """Other sizes than the BASELINE ones: every numpy pairwise-sum shape of the SE row reduction,
    a partial last wave of UEs, a full 16 x 16 slot grid; with the caller's scores / schedulers
    (allocation at the head of every step) and with MAPF + PF on the device (allocation at the head of the step or,
    for a hashed half of the envs, at the tail of the step before)."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.scenario import generate_scaled_scenarios
    from oracle import pyoracle
    S, U, R, G, Us = (shape[k] for k in ("S", "U", "R", "G", "Us"))
    D = shape.get("D", 10)
    tabs = generate_scaled_scenarios(3, seed=5, n_slices=S, n_ues=U, max_ues_slice=Us, min_slices=max(1, S // 2),
                                     min_ues=max(1, Us // 3))
    B, steps = 6, 14
    rng = np.random.default_rng(7)
    scen = rng.integers(0, tabs.n_scenarios, B)
    se_pool = np.stack([se_tile(90, t, U, R) for t in range(B * steps)])
    env = _env(batch=B, n_slices=S, n_ues=U, n_rbs=R, rbs_per_rbg=G, max_ues_slice=Us,
               n_scenarios=tabs.n_scenarios, max_steps=steps, hist_depth=D)
    env.load_scenarios(tabs)
    env.bind_se_pool(torch.as_tensor(_rb_major(se_pool), device=env.device))
    trf = np.concatenate([poisson_traffic_rows(tabs, int(scen[b]), rng, steps) for b in range(B)])
    env.bind_traffic_pool(torch.as_tensor(trf.astype(np.int32), device=env.device))
    env.set_episodes(scenario=scen, se_base=np.arange(B) * steps, se_len=steps, trf_base=np.arange(B) * steps, trf_len=steps)
    env.set_policy(0, 255) if variant == "external" else env.set_policy(2, 1)
    ocfg = pyoracle.make_cfg(S, U, R, G, Us, max_steps=steps, hist_depth=D)
    oenvs = []
    for b in range(B):
        o = pyoracle.OracleEnv(ocfg); o.set_scenario(tabs, int(scen[b])); o.reset(se_pool[b * steps]); oenvs.append(o)
    env.reset()
    for t in range(steps):
        if variant == "external":
            sc = rng.uniform(-1, 1, (B, S)); ic = rng.integers(0, 3, (B, S)).astype(np.uint8)
            obs, rew, done = env.step(sc, ic)
        else:
            sc = np.stack([o.policy_mapf() for o in oenvs]); ic = np.ones((B, S), dtype=np.uint8)
            obs, rew, done = env.step()
        g = {k: x.cpu().numpy() for k, x in env.views().items()}
        for b, o in enumerate(oenvs):
            _, count, _ = o.action_format(sc[b], ic[b], want_dense=False)
            assert np.array_equal(g["rb_count"][b], count), (shape, variant, t, b)
            o.step(sc[b], ic[b], se_pool[b * steps + t], trf[b * steps + t])
            raw = o.raw()
            for name in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts"):
                assert np.array_equal(g[name][b].astype(np.float64), raw[name]), (shape, variant, t, b, name)
            oo = o.obs()
            np.testing.assert_allclose(obs["obs_inter"][b].cpu().numpy(), oo["obs_inter"], rtol=0, atol=OBS_TOL)
            np.testing.assert_allclose(obs["obs_intra"][b].cpu().numpy(), oo["obs_intra"], rtol=0, atol=OBS_TOL)
            np.testing.assert_allclose(rew[b].cpu().numpy(), oo["reward"], rtol=0, atol=REW_TOL)
    env.close()


def test_policy_switching_vs_oracle(build):
    """The allocation of a TTI is made either at the head of its own step (caller's scores, or no valid stored
    allocation) or at the tail of the step before (device policy, half of the envs).  Walk through every
    hand-over: external -> MARR+RR -> MAPF+PF (set_policy in between) -> external -> dense -> MAPF+PF
    -> masked reset -> MT."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.scenario import generate_scaled_scenarios
    S, U, R, G, Us, B = 5, 25, 135, 1, 5, 8
    tabs = generate_scaled_scenarios(4, seed=11, n_slices=S, n_ues=U, max_ues_slice=Us, min_slices=3, min_ues=2)
    plan = (["ext"] * 3 + ["marr_rr"] * 3 + ["mapf_pf"] * 3 + ["ext"] * 2 + ["dense"] * 2 + ["mapf_pf"] * 2 +
            ["reset"] + ["mapf_mt"] * 3 + ["ext"] + ["marr_rr"] * 2)
    steps = len(plan)
    rng = np.random.default_rng(21)
    scen = rng.integers(0, tabs.n_scenarios, B)
    se_pool = np.stack([se_tile(70, t, U, R) for t in range(B * steps)])
    trf = np.concatenate([poisson_traffic_rows(tabs, int(scen[b]), rng, steps) for b in range(B)])
    env = _env(batch=B, n_slices=S, n_ues=U, n_rbs=R, rbs_per_rbg=G, max_ues_slice=Us,
               n_scenarios=tabs.n_scenarios, max_steps=steps)
    env.load_scenarios(tabs)
    env.bind_se_pool(torch.as_tensor(_rb_major(se_pool), device=env.device))
    env.bind_traffic_pool(torch.as_tensor(trf.astype(np.int32), device=env.device))
    env.set_episodes(scenario=scen, se_base=np.arange(B) * steps, se_len=steps, trf_base=np.arange(B) * steps, trf_len=steps)
    oenvs = _oracle_batch(tabs, scen, S, U, R, G, Us, steps)
    env.reset()
    for b in range(B):
        oenvs[b].reset(se_pool[b * steps])
    dev = {"marr_rr": (1, 0), "mapf_pf": (2, 1), "mapf_mt": (2, 2)}
    t = 0                                    # TTIs since the trace started (a reset does not advance it)
    t0 = np.zeros(B, dtype=np.int64)
    for what in plan:
        if what == "reset":
            mask = (np.arange(B) % 2 == 0).astype(np.uint8)
            env.reset(env_mask=mask)
            for b in np.nonzero(mask)[0]:
                t0[b] = t
                oenvs[b].reset(se_pool[b * steps + 0])   # a reset env replays its trace from the offset
            continue
        idx = [b * steps + int(t - t0[b]) for b in range(B)]
        if what == "ext":
            env.set_policy(0, 255)
            sc = rng.uniform(-1, 1, (B, S)); ic = rng.integers(0, 3, (B, S)).astype(np.uint8)
            obs, rew, done = env.step(sc, ic)
        elif what == "dense":
            sc = rng.uniform(-1, 1, (B, S)); ic = rng.integers(0, 3, (B, S)).astype(np.uint8)
            dense = np.zeros((B, U, R), dtype=np.uint8)
            for b in range(B):
                start, count, _ = oenvs[b].action_format(sc[b], ic[b], want_dense=False)
                for u in range(U):
                    dense[b, u, start[u]:start[u] + count[u]] = 1
            tiles = np.stack([_rb_major(se_pool[i][None])[0] for i in idx])
            obs, rew, done = env.step_dense(dense, trf[idx].astype(np.float64), tiles)
        else:
            pol, intra = dev[what]
            env.set_policy(pol, intra)
            sc = np.stack([o.policy_marr() if pol == 1 else o.policy_mapf() for o in oenvs])
            ic = np.full((B, S), intra, dtype=np.uint8)
            obs, rew, done = env.step()
        g = {k: x.cpu().numpy() for k, x in env.views().items()}
        for b, o in enumerate(oenvs):
            _, count, _ = o.action_format(sc[b], ic[b], want_dense=False)
            assert np.array_equal(g["rb_count"][b], count), (what, t, b)
            o.step(sc[b], ic[b], se_pool[idx[b]], trf[idx[b]])
            raw = o.raw()
            for name in ("pkt_incoming", "pkt_throughputs", "pkt_effective_thr", "dropped_pkts"):
                assert np.array_equal(g[name][b].astype(np.float64), raw[name]), (what, t, b, name)
            oo = o.obs()
            np.testing.assert_allclose(obs["obs_inter"][b].cpu().numpy(), oo["obs_inter"], rtol=0, atol=OBS_TOL)
            np.testing.assert_allclose(obs["obs_intra"][b].cpu().numpy(), oo["obs_intra"], rtol=0, atol=OBS_TOL)
            np.testing.assert_allclose(rew[b].cpu().numpy(), oo["reward"], rtol=0, atol=REW_TOL)
        t += 1
    env.close()
