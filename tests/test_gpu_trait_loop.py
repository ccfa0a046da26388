"""The TRAIT-GRANULAR loop -- one C-ABI call per trait method, the loop a drop-in caller writes (rsrl/examples/q_learning.rs:40-52):

    env.transition(a) -> agent.handle(&t) -> [terminal: new episode] -> policy.sample(rng, env.emit().state())

must reproduce the oracle's REFERENCE-ORDER driver loop (oracle/rsrl_oracle_impl.h orc_run_train, instantiation f32d: every Q evaluated afresh from the
weights, as the reference's call pattern does) BIT FOR BIT -- states, actions, weights -- whichever kernels serve the calls:

    fused     ctx-owned stream, device arrays: the four calls of a batch-step are deferred and launched as ONE kernel (k_trait_lm<TRAIT_STEP>)
    separate  the same calls, one kernel each (RSRL_NO_TRAIT_DEFER=1): k_domain_step, k_trait_lm<TRAIT_HANDLE>, k_domain_reset, k_trait_sample (hand-over cache)
    host      numpy arrays through rsrl_amd.Context (staged; the fast kernels)
    generic   a ctx in the feature-major layout (steps_per_launch = 0): k_handle / k_qop, no hand-over
    mixed     fused steps interleaved with calls that break the pattern (the deferred calls are flushed one by one)

rsrl_hip_train itself carries Q(s,.) between steps with a rank-1 correction (equal in exact arithmetic, not in the last bit): the trait loop is compared
with it at a tolerance (test_trait_loop_close_to_train)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = {
    # name: (device kwargs, oracle kwargs)
    "mc5_q_egreedy": (dict(domain=0, order=5, algo=0, policy=1, epsilon=0.1, gamma=0.9, lr=0.001),
                      dict(domain=0, order=5, algo=0, policy=1, epsilon=0.1, gamma=0.9, lr=0.001)),
    "cp1_sarsa_egreedy": (dict(domain=1, order=1, algo=1, policy=1, epsilon=0.2, gamma=0.99, lr=0.01),
                          dict(domain=1, order=1, algo=1, policy=1, epsilon=0.2, gamma=0.99, lr=0.01)),
    "ac1_esarsa_softmax": (dict(domain=2, order=1, algo=2, policy=2, tau=1.0, gamma=0.99, lr=0.005, alpha=1.0),
                           dict(domain=2, order=1, algo=2, policy=2, tau=1.0, gamma=0.99, lr=0.005, alpha=1.0)),
    "mc3_pal_greedy": (dict(domain=0, order=3, algo=5, policy=0, gamma=0.9, lr=0.002, alpha=0.5),
                       dict(domain=0, order=3, algo=5, policy=0, gamma=0.9, lr=0.002, alpha=0.5)),
}
N, K, SEED = 333, 70, 11          # ragged (not a multiple of 64 / 256)


def oracle_run(orc, okw, n=N, k=K):
    ag = orc.make_agent(seed=SEED, max_episode_steps=0, **okw)
    run = orc.Run(ag, n, "f32d")
    run.reset()
    st = run.train(k)
    return run, st


def device_ptr_loop(ctx, k, break_every=0):
    """the loop with device arrays through the raw C ABI; break_every > 0: every so many steps a call that is not part of the pattern is made between the calls"""
    from rsrl_amd import _abi
    from rsrl_amd._devmem import DeviceBuffer
    L, h, n, D = ctx._L, ctx._h, ctx.N, ctx.D
    frm, to = DeviceBuffer(D * n, "float32"), DeviceBuffer(D * n, "float32")
    rew, act, term, td = DeviceBuffer(n, "float32"), DeviceBuffer(n, "int32"), DeviceBuffer(n, "uint8"), DeviceBuffer(n, "float32")
    p = lambda b: C.c_void_p(b.ptr)      # noqa: E731
    ok = _abi.check
    ok(L.rsrl_hip_get_actions(h, p(act)))
    for j in range(k):
        ok(L.rsrl_hip_domain_step(h, p(act), p(frm), p(to), p(rew), p(term)))
        if break_every and j % break_every == 1:
            ctx.sync()                                               # flushes the accepted transition as k_domain_step
        ok(L.rsrl_hip_handle(h, p(frm), p(act), p(rew), p(to), p(term), n, p(td)))
        if break_every and j % break_every == 2:
            _ = ctx.checksum()                                       # flushes transition + handle
        ok(L.rsrl_hip_domain_reset(h, p(term)))
        if break_every and j % break_every == 3:
            _ = ctx.states                                           # flushes all three
        ok(L.rsrl_hip_policy_sample(h, None, n, p(act)))
    ctx.sync()
    last = dict(frm=frm.to_host((D, n)), to=to.to_host((D, n)), rew=rew.to_host(), term=term.to_host(), act=act.to_host(), td=td.to_host())
    for b in (frm, to, rew, act, term, td):
        b.free()
    return last


def host_loop(ctx, k):
    a = ctx.actions
    for _ in range(k):
        frm, nxt, rew, term = ctx.domain_step(a)
        ctx.handle(frm, a, rew, nxt, term)
        ctx.domain_reset(term)
        a = ctx.policy_sample()
    return a


def check(ctx, run, what):
    assert ctx.step_count == run.t, what
    assert np.array_equal(ctx.states.T, run.state), f"{what}: states differ from the reference-order oracle"
    assert np.array_equal(ctx.actions, run.action), f"{what}: actions differ"
    W = np.stack([ctx.get_weights(i) for i in range(ctx.N)])
    assert np.array_equal(W, run.weights), f"{what}: weights differ"


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("mode", ["fused", "separate", "host", "generic", "mixed"])
def test_trait_loop_is_the_reference_order_oracle_bit_for_bit(orc, name, mode, monkeypatch):
    import rsrl_amd as ra
    dkw, okw = CASES[name]
    run, st = oracle_run(orc, okw)
    if name.startswith("cp1"):
        assert st["episodes"] > 0, "the case is there for its terminal transitions"
    if mode == "separate":
        monkeypatch.setenv("RSRL_NO_TRAIT_DEFER", "1")
    spl = 0 if mode == "generic" else 1
    with ra.Context(n_envs=N, seed=SEED, max_episode_steps=0, steps_per_launch=spl, **dkw) as c:
        c.reset()
        if mode in ("host", "generic"):
            a = host_loop(c, K)
            assert np.array_equal(a, run.action)
        else:
            last = device_ptr_loop(c, K, break_every=5 if mode == "mixed" else 0)
            assert np.array_equal(last["act"], run.action)
        check(c, run, f"{name}/{mode}")
    run.close()


def test_fused_step_writes_every_output_of_the_separate_calls(monkeypatch):
    """the arrays a caller reads back after a fused batch-step hold what the separate kernels write: the transition (from, to -- the OBSERVED s', terminal
    or not --, reward, terminal flag), handle's TD errors and the sampled actions"""
    import rsrl_amd as ra
    kw = dict(domain=1, order=1, algo=1, policy=1, epsilon=0.2, gamma=0.99, lr=0.01, n_envs=200, seed=5, max_episode_steps=0, steps_per_launch=1)
    outs = {}
    for mode in ("fused", "separate"):
        if mode == "separate":
            monkeypatch.setenv("RSRL_NO_TRAIT_DEFER", "1")
        with ra.Context(**kw) as c:
            c.reset()
            outs[mode] = [device_ptr_loop(c, k) for k in (1, 9, 25)]         # (each call continues the run)
            outs[mode + "_w"] = c.checksum()
    for a, b in zip(outs["fused"], outs["separate"]):
        for key in a:
            assert np.array_equal(a[key], b[key], equal_nan=True), key
    assert outs["fused_w"] == outs["separate_w"]
    assert any(o["term"].any() for o in outs["fused"]), "no terminal transition in the sample: the case is there for them"


def test_sample_after_handle_hits_the_cache_with_the_bits_of_a_fresh_evaluation():
    """policy_sample(states) on the fast path: states the hand-over cache holds, states it does not, and the same call on a ctx without the fast path"""
    import rsrl_amd as ra
    kw = dict(domain=0, order=5, algo=0, policy=1, epsilon=0.3, gamma=0.9, lr=0.01, n_envs=130, seed=2, max_episode_steps=0)
    with ra.Context(steps_per_launch=1, **kw) as f, ra.Context(steps_per_launch=0, **kw) as g:
        rng = np.random.default_rng(0)
        for c in (f, g):
            c.reset()
        for _ in range(6):
            a = f.actions
            frm, nxt, rew, term = f.domain_step(a)
            g.domain_step(a)
            f.handle(frm, a, rew, nxt, term)
            g.handle(frm, a, rew, nxt, term)
            mixed = nxt.copy()
            other = rng.random(f.N) < 0.4                       # 40 % of the learners are asked about a state of the caller's own
            lo, hi = f.state_bounds()
            mixed[:, other] = (lo[:, None] + rng.random((f.D, int(other.sum()))) * (hi - lo)[:, None]).astype(np.float32)
            assert np.array_equal(f.policy_sample(mixed), g.policy_sample(mixed))
            assert np.array_equal(f.q_evaluate(mixed), g.q_evaluate(mixed))
            na = f.policy_sample()
            assert np.array_equal(na, g.policy_sample())
        assert f.checksum()[0] != 0


def test_trait_loop_close_to_train(orc):
    """against rsrl_hip_train on a twin ctx: the same loop, Q(s,.) carried with a rank-1 correction there -- the trajectories agree except where an argmax margin
    is below fp32 resolution, the weights to rounding"""
    import rsrl_amd as ra
    kw = dict(domain=0, order=5, algo=0, policy=1, epsilon=0.1, gamma=0.9, lr=0.001, n_envs=512, seed=4, max_episode_steps=0, steps_per_launch=1)
    with ra.Context(**kw) as a, ra.Context(**kw) as b:
        a.reset(); b.reset()
        device_ptr_loop(a, 200)
        b.train(200, want_stats=False)
        same = (a.actions == b.actions) & np.all(a.states == b.states, axis=0)
        assert same.mean() >= 0.98, same.mean()
        Wa = np.stack([a.get_weights(i) for i in np.flatnonzero(same)[:64]])
        Wb = np.stack([b.get_weights(i) for i in np.flatnonzero(same)[:64]])
        assert np.max(np.abs(Wa - Wb)) <= 1e-6 * max(1.0, float(np.max(np.abs(Wb))))


def test_deferred_calls_are_not_lost_and_errors_still_surface():
    import rsrl_amd as ra
    from rsrl_amd import _abi
    from rsrl_amd._devmem import DeviceBuffer
    kw = dict(domain=0, order=5, algo=0, policy=1, epsilon=0.1, n_envs=64, seed=1, steps_per_launch=1)
    c = ra.Context(**kw)
    c.reset()
    L, h, n, D = c._L, c._h, c.N, c.D
    frm, to = DeviceBuffer(D * n, "float32"), DeviceBuffer(D * n, "float32")
    rew, act, term = DeviceBuffer(n, "float32"), DeviceBuffer(n, "int32"), DeviceBuffer(n, "uint8")
    p = lambda b: C.c_void_p(b.ptr)      # noqa: E731
    _abi.check(L.rsrl_hip_get_actions(h, p(act)))
    s0 = c.states
    _abi.check(L.rsrl_hip_domain_step(h, p(act), p(frm), p(to), p(rew), p(term)))
    # a bad handle call in the middle of the pattern reports its error; the accepted transition is not lost
    assert L.rsrl_hip_handle(h, p(frm), p(act), p(rew), p(to), None, n, None) != 0
    assert c.step_count == 0
    s1 = c.states                                                # flushes: the transition has happened
    assert not np.array_equal(s0, s1)
    assert np.array_equal(frm.to_host((D, n)), s0) and np.array_equal(to.to_host((D, n)), s1)
    # destroy with an accepted transition pending: the caller's arrays are still written
    _abi.check(L.rsrl_hip_domain_step(h, p(act), p(frm), p(to), p(rew), p(term)))
    c.close()
    assert np.array_equal(frm.to_host((D, n)), s1)
    for b in (frm, to, rew, act, term):
        b.free()
