import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rsrl_amd as ra
for n in (65536, 262144, 1048576, 4194304):
    with ra.Context(n_envs=n, policy=1, epsilon=0.1, max_episode_steps=1000, steps_per_launch=1) as c:
        c.reset(); c.train(200, want_stats=False); c.sync()
        t0 = time.perf_counter(); c.train(600, want_stats=False); c.sync(); dt = time.perf_counter() - t0
        print(json.dumps({"n_envs": n, "us_per_step": dt / 600 * 1e6, "GBps_608": n * 608 * 600 / dt / 1e9, "env_steps_per_s": n * 600 / dt}))
