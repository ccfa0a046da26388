"""Property tests of the CPU oracle (hypothesis): invariants of the path that hold for every input, complementing the
known-answer pins of test_oracle_golden.py.  CPU only."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

fin = dict(allow_nan=False, allow_infinity=False)


@settings(max_examples=60, deadline=None)
@given(order=st.integers(1, 5), x=st.floats(-1.2, 0.6, **fin), v=st.floats(-0.07, 0.07, **fin))
def test_fourier_features_are_cosines(orc, order, x, v):
    # lfa Fourier (+with_bias): every feature is cos(pi * c.s~) for an integer vector c in {0..n}^2 \\ {0}, constant last
    phi = orc.fourier_project(orc.MOUNTAIN_CAR, order, np.array([x, v]))
    n1 = order + 1
    assert len(phi) == n1 * n1 and phi[-1] == 1.0 and np.all(np.abs(phi) <= 1.0 + 1e-15)
    sx, sv = (x + 1.2) / 1.8, (v + 0.07) / 0.14
    for k in range(1, n1 * n1):
        i, j = divmod(k, n1)
        assert abs(phi[k - 1] - np.cos(np.pi * (i * sx + j * sv))) < 1e-12
    # the f32 mirror (separable construction, device op order) stays within the stated tolerance of the f64 value
    phi32 = orc.fourier_project(orc.MOUNTAIN_CAR, order, np.array([x, v], dtype=np.float32), "f32")
    s32 = np.array([x, v], dtype=np.float32).astype(np.float64)
    phi_ref = orc.fourier_project(orc.MOUNTAIN_CAR, order, s32)
    assert np.max(np.abs(phi32 - phi_ref)) < 2e-6


@settings(max_examples=60, deadline=None)
@given(domain=st.integers(0, 2), T=st.sampled_from([4, 8, 16]), B=st.integers(2, 9), u=st.lists(st.floats(0.0, 1.0, **fin), min_size=4, max_size=4))
def test_tile_indices_are_in_range_one_per_tiling(orc, domain, T, B, u):
    lo, hi = orc.domain_bounds(domain)
    D = len(lo)
    s = (lo + (hi - lo) * np.array(u[:D])).astype(np.float32)
    ag = orc.make_agent(domain=domain, basis=orc.TILE, n_tilings=T, tiles_per_dim=B)
    idx = orc.tile_indices(ag, s)
    cells = B ** D
    assert len(idx) == T
    for t, k in enumerate(idx):
        assert t * cells <= k < (t + 1) * cells            # tiling t owns the index block [t*B^D, (t+1)*B^D)


@settings(max_examples=80, deadline=None)
@given(q=st.lists(st.floats(-5, 5, **fin), min_size=2, max_size=3), eps=st.floats(0.0, 1.0, **fin), tau=st.floats(0.05, 5.0, **fin))
def test_policy_probabilities_form_a_distribution(orc, q, eps, tau):
    q = np.array(q)
    for pol in (orc.GREEDY, orc.EGREEDY, orc.SOFTMAX, orc.RANDOM):
        p = orc.policy_probs(pol, q, eps=eps, tau=tau)
        assert np.all(p >= 0) and abs(p.sum() - 1.0) < 1e-9
    # greedy mass sits on the maxima; epsilon-greedy mixes it with the uniform distribution (epsilon_greedy.rs:38-45)
    pg, pe = orc.policy_probs(orc.GREEDY, q), orc.policy_probs(orc.EGREEDY, q, eps=eps)
    assert np.allclose(pe, (1 - eps) * pg + eps / len(q), atol=1e-12)
    assert pg[np.argmax(q)] > 0


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 2 ** 32 - 1), env=st.integers(0, 2 ** 31 - 1), t=st.integers(0, 2 ** 40), blk=st.integers(0, 4))
def test_draws_are_pure_functions_of_their_address(orc, seed, env, t, blk):
    a, b = orc.draw(seed, env, t, blk), orc.draw(seed, env, t, blk)
    assert list(a) == list(b)
    assert list(orc.draw(seed, env, t + 1, blk)) != list(a) and list(orc.draw(seed, env + 1, t, blk)) != list(a)


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 2 ** 64 - 1), env=st.integers(0, 2 ** 31 - 1), t=st.integers(0, 2 ** 40))
def test_per_step_draws_are_halves_of_a_shared_philox_block(orc, seed, env, t):
    # the stream contract: the per-step draws (STEP with its alias RESET, INNER) use two words, and the batch-steps
    # 2k and 2k + 1 share the Philox4x32-10 block at counter (k, env, block): words 0-1 for the even step, 2-3 for the odd one;
    # every other block index takes the whole block at counter t
    k = t >> 1
    key = [seed & 0xffffffff, seed >> 32]
    for blk, base in ((orc.BLK_STEP, orc.BLK_STEP), (orc.BLK_RESET, orc.BLK_STEP), (orc.BLK_INNER, orc.BLK_INNER)):
        p = list(orc.philox([k & 0xffffffff, k >> 32, env, base], key))
        even, odd = list(orc.draw(seed, env, 2 * k, blk)), list(orc.draw(seed, env, 2 * k + 1, blk))
        assert even == [p[0], p[1], p[1], 0] and odd == [p[2], p[3], p[3], 0]
    for blk in (orc.BLK_INIT, orc.BLK_API, 16, 79):
        assert list(orc.draw(seed, env, t, blk)) == list(orc.philox([t & 0xffffffff, t >> 32, env, blk], key))


@settings(max_examples=25, deadline=None)
@given(domain=st.integers(0, 2), a=st.integers(0, 1), u=st.lists(st.floats(0.05, 0.95, **fin), min_size=4, max_size=4))
def test_domain_steps_stay_inside_the_state_space(orc, domain, a, u):
    lo, hi = orc.domain_bounds(domain)
    D = len(lo)
    s = lo + (hi - lo) * np.array(u[:D])
    for _ in range(20):
        s, r, term = orc.domain_step(domain, s, a)
        assert np.all(s >= lo - 1e-12) and np.all(s <= hi + 1e-12)        # clip! / wrap! (macros.rs:3-24)
        assert r in (-1.0, 0.0, 1.0) or domain == 1
        if term:
            break
