import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rsrl_amd as ra
with ra.Context(n_envs=65536, policy=1, epsilon=0.1, max_episode_steps=1000) as c:
    c.reset(); c.train(100, want_stats=False); c.sync()
    for K in (20, 5, 1):
        n = 20000
        t0 = time.perf_counter()
        for _ in range(n): c.train(K, want_stats=False)
        t1 = time.perf_counter()
        c.sync()
        t2 = time.perf_counter()
        print("K", K, "host us/call %.2f" % ((t1 - t0) / n * 1e6), "total us/call %.2f" % ((t2 - t0) / n * 1e6), "GPU us/call at 0.73 us/step %.1f" % (K * 0.73))
