"""Array arguments may be HOST or DEVICE pointers (include/rsrl_hip.h, conventions): device pointers are used in place and the call is asynchronous on the
ctx's stream.  Every entry point that takes arrays, with raw hipMalloc buffers (rsrl_amd/_devmem.py: the HIP runtime through ctypes, what a Rust caller's hip-sys binding hands over --
no torch: its first GPU use costs a cold box minutes), against the same call with host arrays on a twin ctx -- bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = {
    "reg": dict(domain=0, order=5, algo=0, policy=1, epsilon=0.2, gamma=0.9, lr=0.001, n_envs=96, seed=3, max_episode_steps=60),
    "tile_shared": dict(domain=1, basis=1, n_tilings=8, tiles_per_dim=8, algo=1, policy=1, epsilon=0.1, gamma=0.99, lr=0.1 / 8 / 96, weight_mode=1, n_envs=96, seed=3,
                        max_episode_steps=60),
    "wave_bf16": dict(domain=2, order=7, algo=2, policy=2, tau=1.0, gamma=0.99, lr=2.5e-4, alpha=1.0, weight_dtype=1, n_envs=8, seed=3, max_episode_steps=60),
    "generic_tdl": dict(domain=1, order=2, algo=8, policy=3, gamma=0.9, lam=0.5, trace=1, n_envs=96, seed=3, max_episode_steps=60),
    "lambda_tile": dict(domain=0, basis=1, n_tilings=4, tiles_per_dim=8, algo=3, policy=1, epsilon=0.2, gamma=0.99, alpha=0.02, lam=0.8, n_envs=16, seed=3),
}


@pytest.mark.parametrize("name", list(CASES))
def test_device_pointer_arguments_equal_host_arguments(name):
    import rsrl_amd as ra
    from rsrl_amd._devmem import DeviceBuffer
    kw = CASES[name]
    with ra.Context(**kw) as h, ra.Context(**kw) as d:
        L = d._L
        N, D, A, F, O = d.N, d.D, d.A, d.F, d.n_out
        ptr = lambda b: C.c_void_p(b.ptr)       # noqa: E731
        f32, i32, u8 = np.float32, np.int32, np.uint8

        def like(b, host):                         # a fresh device buffer holding `host`
            n = DeviceBuffer(b.count, b.dtype); n.from_host(host)
            return n
        ok = lambda rc: ra._abi.check(rc)               # noqa: E731
        for c in (h, d):
            c.reset()
            c.train(25, want_stats=False)
        # get_states / get_actions / episode steps into device memory
        ds = DeviceBuffer(D * N, f32); da = DeviceBuffer(N, i32)
        de = DeviceBuffer(N, i32)
        ok(L.rsrl_hip_get_states(d._h, ptr(ds))); ok(L.rsrl_hip_get_actions(d._h, ptr(da))); ok(L.rsrl_hip_get_episode_steps(d._h, ptr(de)))
        d.sync()
        assert np.array_equal(ds.to_host((D, N)), h.states) and np.array_equal(da.to_host(), h.actions)
        assert np.array_equal(de.to_host().view(np.uint32), h.episode_steps)
        # q_evaluate / policy_mode on device states, device outputs
        dq = DeviceBuffer(O * N, f32)
        ok(L.rsrl_hip_q_evaluate(d._h, ptr(ds), N, ptr(dq)))
        d.sync()
        assert np.array_equal(dq.to_host((O, N)), h.q_evaluate(h.states), equal_nan=True)
        if kw["algo"] not in (7, 8):
            dm = DeviceBuffer(N, i32)
            ok(L.rsrl_hip_policy_mode(d._h, ptr(ds), N, ptr(dm)))
            d.sync()
            assert np.array_equal(dm.to_host(), h.policy_mode(h.states))
        # domain_step with device outputs, then handle with device inputs (sparse / shared-trace agents aside, every agent has a handle)
        frm_h, nxt_h, rew_h, term_h = h.domain_step(h.actions)
        dfrm = DeviceBuffer(D * N, f32); dnxt = DeviceBuffer(D * N, f32)
        drew = DeviceBuffer(N, f32); dterm = DeviceBuffer(N, u8)
        ok(L.rsrl_hip_domain_step(d._h, ptr(da), ptr(dfrm), ptr(dnxt), ptr(drew), ptr(dterm)))
        d.sync()
        assert np.array_equal(dnxt.to_host((D, N)), nxt_h) and np.array_equal(drew.to_host(), rew_h) and np.array_equal(dterm.to_host(), term_h)
        td_h = h.handle(frm_h, h.actions, rew_h, nxt_h, term_h)
        dtd = DeviceBuffer(N, f32)
        ok(L.rsrl_hip_handle(d._h, ptr(dfrm), ptr(da), ptr(drew), ptr(dnxt), ptr(dterm), N, ptr(dtd)))
        d.sync()
        assert np.array_equal(dtd.to_host(), td_h, equal_nan=True)
        # get_weights / set_weights through device memory: learner 0 (or the shared approximator)
        dw = DeviceBuffer(F * O, f32)
        ok(L.rsrl_hip_get_weights(d._h, 0, ptr(dw)))
        d.sync()
        assert np.array_equal(dw.to_host((F, O)), h.get_weights(0))
        w2_h = dw.to_host((F, O)) * np.float32(0.5)
        w2 = like(dw, w2_h)
        ok(L.rsrl_hip_set_weights(d._h, 0, ptr(w2)))
        h.set_weights(w2_h, 0)
        assert np.array_equal(d.get_weights(0), h.get_weights(0))
        # set_states from device memory: the array is checked ON the device by the host path's rule (ABI 9; it used to be clamped silently, NaN passing) --
        # a non-finite / far-out-of-range component refuses the call and leaves the ctx's states untouched; set_actions clamps a device array (no host copy)
        before = d.states
        for bad in (float("inf"), float("nan"), 1e30):
            bad_h = ds.to_host((D, N)); bad_h[0, 0] = bad
            bad_s = like(ds, bad_h)
            assert L.rsrl_hip_set_states(d._h, ptr(bad_s)) == -1 and b"device array of states" in L.rsrl_hip_last_error()
            assert np.array_equal(d.states, before)
        bad_h = da.to_host(); bad_h[0] = 77
        bad_a = like(da, bad_h)
        ok(L.rsrl_hip_set_actions(d._h, ptr(bad_a)))
        d.sync()
        assert 0 <= d.actions[0] < A
        ok(L.rsrl_hip_set_states(d._h, ptr(dnxt))); ok(L.rsrl_hip_set_actions(d._h, ptr(da)))
        h.states, h.actions = nxt_h, h.actions
        for c in (h, d):
            c.train(20, want_stats=False)
        assert np.array_equal(d.states, h.states) and np.array_equal(d.get_weights(0), h.get_weights(0), equal_nan=True)
        # greedy rollout with device outputs
        if kw["algo"] not in (7, 8):
            dn = DeviceBuffer(N, i32); dt = DeviceBuffer(N, f32)
            ok(L.rsrl_hip_rollout_greedy(d._h, 40, ptr(dn), ptr(dt)))
            d.sync()
            n_h, t_h = h.rollout_greedy(40)
            assert np.array_equal(dn.to_host().view(np.uint32), n_h) and np.array_equal(dt.to_host(), t_h)
