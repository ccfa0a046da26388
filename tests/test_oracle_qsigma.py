"""CPU tests of the oracle's QSigma restatement (control/td/q_sigma.rs:14-202) -- the reference has no test for it and cannot
run (it panics at its first full backup, see oracle/rsrl_oracle_impl.h): parity unpinned beyond an independent numpy
transcription of the same lines with the same one-line repair, and the special cases the algorithm must reduce to."""
import numpy as np
import pytest


def np_argmaxima(v):                       # utils.rs:6-21
    mx, ixs = -np.finfo(np.float64).max, []
    for i, x in enumerate(v):
        if abs(x - mx) < 1e-7:
            ixs.append(i)
        elif x > mx:
            mx, ixs = x, [i]
    return ixs, mx


class NpQSigma:
    """q_sigma.rs transcribed line by line in numpy; propagate() without the dead out-of-bounds z update of its last iteration"""

    def __init__(self, orc, ag, W, n_steps, sigma, alpha, gamma, lr, eps):
        self.orc, self.ag, self.W, self.n, self.sigma, self.alpha, self.gamma, self.lr, self.eps = orc, ag, W, n_steps, sigma, alpha, gamma, lr, eps
        self.entries = []

    def phi(self, s):
        return self.orc.fourier_project(0, 5, s, "f64")

    def handle(self, s, a, r, ns, term, na):
        qa = self.phi(s) @ self.W[:, a]
        if term:
            e = dict(s=s, a=a, q=qa, residual=r - qa, pi=0.0, mu=1.0)
        else:
            nqs = self.phi(ns) @ self.W
            mx_ix, exp_nqs = np_argmaxima(nqs)
            pi = 1.0 / len(mx_ix) if na in mx_ix else 0.0
            mu = self.eps / 3 + (1 - self.eps) * pi                      # epsilon_greedy.rs:49-63
            e = dict(s=s, a=a, q=qa, residual=r + self.gamma * (self.sigma * nqs[na] + (1 - self.sigma) * exp_nqs) - qa, pi=pi, mu=mu)
        self.entries.append(e)
        if len(self.entries) >= self.n:
            g, z, isr = self.entries[0]["q"], 1.0, 1.0
            for k in range(self.n):
                b1 = self.entries[k]
                g += z * b1["residual"]
                if k + 1 < self.n:
                    b2 = self.entries[k + 1]
                    z *= self.gamma * ((1 - self.sigma) * b2["pi"] + self.sigma)
                isr *= 1 - self.sigma + self.sigma * b1["pi"] / b1["mu"]
            anchor = self.entries.pop(0)
            qsa = self.phi(anchor["s"]) @ self.W[:, anchor["a"]]
            self.W[:, anchor["a"]] += self.lr * (self.alpha * isr * (g - qsa)) * self.phi(anchor["s"])
        if term:
            self.entries = []
        return e["residual"]


@pytest.mark.parametrize("n_steps,sigma", [(1, 0.0), (1, 1.0), (3, 0.5), (4, 0.0), (2, 1.0)])
def test_qsigma_matches_numpy_transcription(orc, n_steps, sigma):
    rng = np.random.default_rng(n_steps * 10 + int(sigma * 4))
    lo, hi = orc.domain_bounds(0)
    kw = dict(gamma=0.9, lr=0.05, alpha=0.6, epsilon=0.2)
    ag = orc.make_agent(algo=orc.Q_SIGMA, policy=orc.EGREEDY, sigma=sigma, n_steps=n_steps, **kw)
    W = rng.normal(size=(36, 3)) * 0.2
    Wn = W.copy()
    bk = orc.QSigmaBackup(n_steps, "f64")
    ref = NpQSigma(orc, ag, Wn, n_steps, sigma, 0.6, 0.9, 0.05, 0.2)
    s = lo + (hi - lo) * rng.random(2)
    for k in range(40):
        a = int(rng.integers(0, 3))
        ns = lo + (hi - lo) * rng.random(2)
        term = int(k % 11 == 10)
        x = orc.draw(7, 0, k, orc.BLK_INNER)
        # the agent's inner a' as the oracle draws it, so that the transcription sees the same action
        na = orc.policy_sample(orc.EGREEDY, orc.q_evaluate(ag, W, ns, "f64"), x, eps=0.2, prec="f64")
        d = bk.handle(ag, W, s, a, -1.0, ns, term, x)
        dn = ref.handle(s, a, -1.0, ns, bool(term), na)
        assert abs(d - dn) <= 1e-12 * (1 + abs(dn))
        assert np.max(np.abs(W - Wn)) <= 1e-12
        assert len(bk) == len(ref.entries)
        s = ns
    assert np.max(np.abs(W - (rng.normal(size=(36, 3)) * 0 + W))) == 0 and len(bk) <= max(0, n_steps - 1)


def test_qsigma_one_step_tree_backup_is_qlearning(orc):
    # n_steps = 1, sigma = 0: g = q + residual with the greedy expectation, isr = 1 => alpha * Q-learning's TD error (q_learning.rs:57-62)
    rng = np.random.default_rng(5)
    lo, hi = orc.domain_bounds(0)
    kw = dict(gamma=0.95, lr=0.05, epsilon=0.3)
    qs = orc.make_agent(algo=orc.Q_SIGMA, policy=orc.EGREEDY, sigma=0.0, n_steps=1, alpha=1.0, **kw)
    ql = orc.make_agent(algo=orc.QLEARNING, policy=orc.EGREEDY, **kw)
    bk = orc.QSigmaBackup(1, "f64")
    for k in range(30):
        W = rng.normal(size=(36, 3)) * 0.3
        W2 = W.copy()
        s, ns = lo + (hi - lo) * rng.random(2), lo + (hi - lo) * rng.random(2)
        a, term = int(rng.integers(0, 3)), int(k % 7 == 6)
        d1 = bk.handle(qs, W, s, a, -1.0, ns, term, orc.draw(1, 0, k, orc.BLK_INNER))
        d2 = orc.handle(ql, W2, s, a, -1.0, ns, term, (0, 0, 0, 0), "f64")
        assert abs(d1 - d2) <= 1e-12 and np.max(np.abs(W - W2)) <= 1e-12
        assert len(bk) == 0


def test_qsigma_backup_fills_then_slides_and_terminal_clears(orc):
    ag = orc.make_agent(algo=orc.Q_SIGMA, policy=orc.GREEDY, sigma=1.0, n_steps=4, alpha=0.5, lr=0.1)
    bk = orc.QSigmaBackup(4, "f64")
    W = np.zeros((36, 3))
    s = np.array([-0.5, 0.0])
    lens = []
    for k in range(9):
        bk.handle(ag, W, s, k % 3, -1.0, s + 0.01, int(k == 6), orc.draw(0, 0, k, orc.BLK_INNER))
        lens.append(len(bk))
    assert lens == [1, 2, 3, 3, 3, 3, 0, 1, 2]           # no update before n entries; pop after every update; terminal -> clear (q_sigma.rs:113-154)
    assert np.abs(W).max() > 0


def test_qsigma_learns_mountain_car(orc):
    # the repaired agent is a working learner: greedy rollouts get shorter than the step cap
    ag = orc.make_agent(algo=orc.Q_SIGMA, policy=orc.EGREEDY, sigma=0.5, n_steps=4, alpha=1.0, lr=0.002, gamma=0.99, epsilon=0.1, seed=2,
                        max_episode_steps=400)
    run = orc.Run(ag, 8, "f64")
    run.reset()
    run.train(12000)
    n, _ = run.rollout_greedy(400)
    assert n.mean() < 350
