import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsrl_amd as ra
mode = sys.argv[1] if len(sys.argv) > 1 else "fourier"
if mode == "fourier":
    c = ra.Context(n_envs=131072, policy=1, epsilon=0.1, lr=0.001/131072, weight_mode=ra.W_SHARED, max_episode_steps=1000)
else:
    c = ra.Context(domain=1, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=ra.SARSA, n_envs=262144, policy=1, epsilon=0.1,
                   gamma=0.99, lr=0.0125/262144, weight_mode=ra.W_SHARED, max_episode_steps=1000)
c.reset(); c.train(50, want_stats=False); c.sync()
t0 = time.perf_counter(); c.train(300, want_stats=False); c.sync(); dt = time.perf_counter() - t0
print(mode, "us/step", dt / 300 * 1e6, "env-steps/s %.3g" % (c.N * 300 / dt))
