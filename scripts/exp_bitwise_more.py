import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsrl_amd as ra
from oracle import oracle as orc
N, K = 128, 300
cases = [
 ("sarsa_lambda sat", dict(domain=0, order=5, algo=3, policy=1, trace=1, gamma=0.99, alpha=0.01, lam=0.7, epsilon=0.2)),
 ("q_lambda acc", dict(domain=0, order=5, algo=4, policy=1, trace=0, gamma=0.99, alpha=0.01, lam=0.7, epsilon=0.2)),
 ("sarsa_lambda dutch cp", dict(domain=1, order=1, algo=3, policy=1, trace=2, gamma=0.99, alpha=0.01, lam=0.7, epsilon=0.2)),
 ("greedy_gq", dict(domain=0, order=3, algo=6, policy=1, gamma=0.99, lr=0.1, lr_td=0.001, epsilon=0.1)),
 ("td", dict(domain=0, order=5, algo=7, policy=3, gamma=0.99, lr=0.01)),
 ("td_lambda", dict(domain=0, order=3, algo=8, policy=3, gamma=0.9, lam=0.3, trace=1)),
 ("pal", dict(domain=0, order=5, algo=5, policy=1, gamma=0.95, lr=0.01, alpha=0.5, epsilon=0.1)),
]
for name, kw in cases:
    ag = orc.make_agent(seed=9, max_episode_steps=40, **kw)
    run = orc.Run(ag, N, "f32d"); run.reset(); run.train(K)
    with ra.Context(n_envs=N, seed=9, max_episode_steps=40, **kw) as c:
        c.reset(); c.train(K)
        st = np.all(c.states.T == run.state, axis=1) & (c.actions == run.action)
        w = np.array([np.array_equal(c.get_weights(i), run.weights[i]) for i in range(N)])
        dw = max(np.abs(c.get_weights(i) - run.weights[i]).max() for i in range(N))
        print(f"{name:24s} states+actions identical {st.mean():.3f}  weights bit-identical {w.mean():.3f}  max|dW| {dw:.3g}", flush=True)
