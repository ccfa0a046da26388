"""Round-5 CPU tests of the oracle (no GPU): the teacher-forcing tape and what it is used to show."""
import numpy as np


def test_c5_lr_is_past_the_stability_limit(orc):
    # why the population statistics of configs[4] are taken at lr 2.5e-4: at the configuration's lr = 1e-3 the REFERENCE rule itself diverges
    # (f64 oracle, no device involved): lr * |phi|^2 ~ 2.05 > 2
    ag = orc.make_agent(seed=11, max_episode_steps=100, domain=2, order=7, algo=2, policy=2, tau=1.0, gamma=0.99, lr=1e-3, alpha=1.0)
    run = orc.Run(ag, 4, "f64")
    run.reset()
    a = run.train(400)["sum_abs_td_error"]
    b = run.train(400)["sum_abs_td_error"]
    phi = orc.fourier_project(2, 7, np.zeros(4), "f64")
    assert 2.0 < 1e-3 * float(phi @ phi) and b > 3 * a


def test_teacher_step_is_train_with_fp32_rounded_states(orc):
    # the tape records what the agents handled, successor states are fp32-representable, and with a domain whose successor states are
    # fp32-exact anyway (none is) the run would equal train(); here: the tape's transitions replayed through orc.handle reproduce the run's W
    ag = orc.make_agent(domain=0, order=3, algo=orc.SARSA, policy=orc.EGREEDY, seed=5, max_episode_steps=30, gamma=0.9, lr=0.01)
    run = orc.Run(ag, 6, "f64")
    run.reset()
    W = np.zeros((6, run.F, run.A))
    for k in range(80):
        t = run.teacher_step()
        assert np.array_equal(t["to"].astype(np.float32).astype(np.float64), t["to"])
        assert np.array_equal(t["frm"].astype(np.float32).astype(np.float64), t["frm"])
        for i in range(6):
            d = orc.handle(ag, W[i], t["frm"][i], t["action"][i], t["reward"][i], t["to"][i], t["terminal"][i], orc.draw(5, i, k, orc.BLK_INNER), "f64")
            assert d == t["td"][i]
    assert np.array_equal(W, run.weights) and np.abs(W).max() > 0
    assert t["stats"]["env_steps"] == 6


def test_teacher_step_shared_minibatch_rule(orc):
    # shared W: every error against W_t, one summed update
    ag = orc.make_agent(domain=1, basis=orc.TILE, algo=orc.SARSA, policy=orc.EGREEDY, shared_w=True, seed=2, max_episode_steps=20, gamma=0.99, lr=0.01)
    run = orc.Run(ag, 16, "f64")
    run.reset()
    W = np.zeros((run.F, run.A))
    for k in range(30):
        t = run.teacher_step()
        dW = np.zeros_like(W)
        for i in range(16):
            Wi = W.copy()
            orc.handle(ag, Wi, t["frm"][i], t["action"][i], t["reward"][i], t["to"][i], t["terminal"][i], orc.draw(2, i, k, orc.BLK_INNER), "f64")
            dW += Wi - W
        W += dW
    assert np.allclose(W, run.weights, rtol=0, atol=1e-15) and np.abs(W).max() > 0


def test_sparse_trace_rule_is_the_dense_rule_while_nothing_is_evicted(orc):
    # the sparse-trace mini-batch loop (f64) against a dense numpy restatement of SURVEY A.7 with eligibility traces: identical while no list is full
    N, K, T, B, A = 5, 40, 4, 5, 2
    kw = dict(domain=1, basis=orc.TILE, n_tilings=T, tiles_per_dim=B, algo=orc.SARSA_LAMBDA, policy=orc.EGREEDY, epsilon=0.2, shared_w=True, seed=11,
              gamma=0.97, lam=0.8, trace=orc.TRACE_ACCUMULATE, alpha=0.01, max_episode_steps=25)
    ag = orc.make_agent(**kw)
    run = orc.Run(ag, N, "f64")
    run.reset()
    F = run.F
    W = np.zeros((F, A)); Z = np.zeros((N, F, A))
    s = run.state.copy(); a = run.action.copy(); ep = np.zeros(N, int)
    rate = 0.97 * 0.8
    for k in range(K):
        dW = np.zeros_like(W); nss = []; flags = []
        for i in range(N):
            ns, r, term = orc.domain_step(1, s[i], a[i], "f64")
            ep[i] += 1
            idx, idn = orc.tile_indices(ag, s[i]), orc.tile_indices(ag, ns)
            qs, qn = W[idx].sum(0), W[idn].sum(0)
            Z[i] *= rate
            Z[i][idx, a[i]] += 1.0
            if term:
                delta = r - qs[a[i]]
            else:
                na = orc.policy_sample(orc.EGREEDY, qn, orc.draw(11, i, k, orc.BLK_INNER), eps=0.2)
                delta = r + 0.97 * qn[na] - qs[a[i]]
            dW += 0.01 * delta * Z[i]
            if term:
                Z[i][:] = 0
            nss.append(ns); flags.append(term or ep[i] >= 25)
        W += dW
        for i in range(N):
            ns = nss[i]
            if flags[i]:
                ns = orc.domain_reset(1, "f64"); ep[i] = 0
            q = W[orc.tile_indices(ag, ns)].sum(0)
            a[i] = orc.policy_sample(orc.EGREEDY, q, orc.draw(11, i, k, orc.BLK_STEP), eps=0.2)
            s[i] = ns
    run.train_sparse_lambda(K)
    assert np.allclose(run.weights, W, rtol=0, atol=1e-14) and np.abs(W).max() > 0
    assert np.array_equal(run.action, a) and np.allclose(run.state, s, atol=1e-14)
    for i in range(N):
        assert np.allclose(run.sparse_trace(i), Z[i], rtol=0, atol=1e-15)
