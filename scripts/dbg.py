import sys, numpy as np
sys.path.insert(0, '.')
import rsrl_amd as ra
n = int(sys.argv[1]); spl = int(sys.argv[2]); stats = sys.argv[3] == '1'
c = ra.Context(n_envs=n, policy=1, epsilon=0.1, seed=9, max_episode_steps=100, steps_per_launch=spl)
c.reset(); c.sync(); print('reset ok', n, spl, stats, flush=True)
c.train(3, want_stats=stats); c.sync(); print('train 3 ok', flush=True)
c.train(300, want_stats=stats); c.sync(); print('train ok', flush=True)
