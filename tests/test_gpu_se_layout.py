"""The two layouts of a bound SE pool (include/ranenv.h: ranenv_bind_se_pool / ranenv_bind_se_pool_quad): RB-major [R][U] -- the order of
the reference's .mat, channels/quadriga.py:70-72 -- and RB-quad-major [ceil(R/4)][U][4], which the host layer binds by default.  Same
values, same summation order: every number a step produces is identical bit for bit, for every path that replays pooled tiles."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from tests.common import comparable_views

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("U,R", [(100, 135), (25, 135), (7, 5), (256, 419), (33, 8), (64, 130)])
def test_retile_quad_is_an_index_transform(U, R):
    _need_gpu()
    import ctypes as C
    from intent_radio_sched_multi_slice_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n = 5
    g = torch.Generator(device=dev); g.manual_seed(U * 1000 + R)
    rb = torch.rand((n, R, U), generator=g, device=dev, dtype=torch.float32)
    Rq = (R + 3) // 4
    quad = torch.full((n, Rq, U, 4), -1.0, device=dev, dtype=torch.float32)
    st = lib.ranenv_se_retile_quad(C.c_void_p(rb.data_ptr()), C.c_void_p(quad.data_ptr()), n, U, R, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert st == 0
    torch.cuda.synchronize()
    ref = torch.zeros((n, Rq * 4, U), device=dev)
    ref[:, :R] = rb
    assert torch.equal(quad, ref.reshape(n, Rq, 4, U).permute(0, 1, 3, 2).contiguous())        # zeros behind RB R-1


@pytest.mark.parametrize("shape", [dict(), dict(n_slices=5, n_ues=25, n_rbs=135, rbs_per_rbg=5, max_ues_slice=5, min_slices=3, min_ues=2),
                                   dict(n_slices=3, n_ues=40, n_rbs=419, max_ues_slice=16, min_slices=2, min_ues=3)])
@pytest.mark.parametrize("se_mode", ["stream", "gather"])
def test_quad_pool_equals_rb_major_pool_bit_for_bit(shape, se_mode):
    """reset, step(), rollouts over partitions (launches of several TTIs / persistent launches), a step with the caller's scores, a
    dense step from the pool, auto-reset: the same state, observations and rewards from both layouts."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    dev = torch.device("cuda", 0)
    outs = []
    for layout in ("rb", "quad"):
        wl = make_mult_slice_workload(300, dev, n_scenarios=12, n_traces=6, trace_len=20, max_steps=1000, se_layout=layout, **shape)
        env = wl.env
        assert env.se_layout == layout
        env.set_se_mode(se_mode)
        env.set_partitions(3)
        snaps = []

        def snap():
            torch.cuda.synchronize()
            snaps.append((comparable_views(wl), env.obs_inter.clone(), env.obs_intra.clone(), env.reward.clone()))
        env.reset(); snap()
        for _ in range(3):
            env.step()
        snap()
        env.rollout(23); snap()
        g = torch.Generator(device=dev); g.manual_seed(3)
        sc = torch.rand((env.B, env.S), generator=g, device=dev, dtype=torch.float64) * 2 - 1
        ic = torch.randint(0, 3, (env.B, env.S), generator=g, device=dev, dtype=torch.uint8)
        env.step(sc, ic); snap()
        if se_mode == "stream":
            dense = torch.zeros((env.B, env.U, env.R), dtype=torch.uint8, device=dev)
            dense[:, :, ::3] = 1
            env.step_dense(dense, traffic_bits=torch.full((env.B, env.U), 5e5, dtype=torch.float64, device=dev)); snap()
        env.rollout(9); snap()
        outs.append(snaps)
        env.close()
    for i, (a, b) in enumerate(zip(*outs)):
        for k in a[0]:
            assert torch.equal(a[0][k], b[0][k]), (i, k)
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]), i


def test_quad_pool_sidecars_and_pooled_tiles():
    """The gather sidecars built from a quad pool are those built from the RB-major pool; pooled_tiles() hands tiles back RB-major."""
    _need_gpu()
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    dev = torch.device("cuda", 0)
    side = []
    for layout in ("rb", "quad"):
        wl = make_mult_slice_workload(16, dev, n_scenarios=4, n_traces=3, trace_len=7, se_layout=layout)
        wl.env.set_se_mode("gather")
        sc = wl.env.se_sidecars()
        side.append({k: v.clone() for k, v in sc.items()})
        idx = torch.as_tensor([0, 5, 20], device=dev)
        assert torch.equal(wl.env.pooled_tiles(idx), wl.se_pool[idx])
        wl.env.close()
    for k in side[0]:
        assert torch.equal(side[0][k], side[1][k]), k


def test_bad_quad_pool_arguments_are_refused():
    _need_gpu()
    import ctypes as C
    from intent_radio_sched_multi_slice_amd import _lib
    from intent_radio_sched_multi_slice_amd.batched_env import BatchedRanEnv
    env = BatchedRanEnv(batch=4, n_slices=2, n_ues=10, n_rbs=13, max_ues_slice=5, n_scenarios=1)
    lib = _lib.load()
    pool = torch.zeros((3, 4, 10, 4), dtype=torch.float32, device=env.device)
    need = 4 * 10 * 4
    assert lib.ranenv_bind_se_pool_quad(env._h, C.c_void_p(pool.data_ptr()), 3, need - 4) != 0          # stride too small
    assert lib.ranenv_bind_se_pool_quad(env._h, C.c_void_p(pool.data_ptr() + 4), 3, need) != 0           # not 16-byte aligned
    assert lib.ranenv_bind_se_pool_quad(env._h, C.c_void_p(pool.data_ptr()), 0, need) != 0               # no tiles
    assert lib.ranenv_bind_se_pool_quad(env._h, C.c_void_p(pool.data_ptr()), 3, need) == 0
    with pytest.raises(_lib.RanEnvError):
        env.bind_se_pool(torch.zeros((3, 5, 10, 4), dtype=torch.float32, device=env.device))
    env.close()


def test_one_pool_copy_is_shared_zero_copy_between_workloads(monkeypatch):
    """bench.py keeps ONE copy of the SE pool: a workload built with keep_rb_major=False hands out the RB-quad-major tensor that is
    bound (the RB-major tensor the generator wrote is dropped), a second workload binds that very tensor without copying, and both replay the
    same tiles as a workload that kept the RB-major pool."""
    _need_gpu()
    monkeypatch.delenv("RANENV_SE_LAYOUT", raising=False)        # (the default layout, whatever the suite's pass presets)
    from intent_radio_sched_multi_slice_amd.workloads import make_mult_slice_workload
    dev = torch.device("cuda", 0)
    ref = make_mult_slice_workload(32, dev, n_scenarios=4, n_traces=3, trace_len=7)                      # keeps the RB-major tensor
    a = make_mult_slice_workload(32, dev, n_scenarios=4, n_traces=3, trace_len=7, keep_rb_major=False)
    assert a.se_pool.dim() == 4 and a.se_pool is a.env.bound_se_pool and a.env.se_pool_rb_major is None
    b = make_mult_slice_workload(32, dev, n_scenarios=4, n_traces=3, trace_len=7, se_pool=a.se_pool)     # binds the quad tensor as it is
    assert b.env.bound_se_pool.data_ptr() == a.se_pool.data_ptr() and b.env.se_layout == "quad"
    idx = torch.arange(21, device=dev)
    assert torch.equal(a.env.pooled_tiles(idx), ref.se_pool) and torch.equal(b.env.pooled_tiles(idx), ref.se_pool)
    outs = []
    for wl in (ref, a, b):
        wl.env.reset(); wl.env.rollout(11); torch.cuda.synchronize()
        v = wl.env.views()
        outs.append((wl.env.reward.clone(), wl.env.obs_inter.clone(), v["queue_pkts"].clone(), v["pkt_throughputs"].clone()))
    for x, y, z in zip(*outs):
        assert torch.equal(x, y) and torch.equal(x, z)
    kept = make_mult_slice_workload(8, dev, n_scenarios=4, n_traces=3, trace_len=7)
    kept.env.bind_se_pool(kept.se_pool, keep_rb_major=True)
    assert kept.env.se_pool_rb_major is kept.se_pool
    for wl in (ref, a, b, kept):
        wl.env.close()
