import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rsrl_amd as ra
for n in (65536, 131072, 262144, 524288, 1048576):
    with ra.Context(n_envs=n, policy=1, epsilon=0.1, max_episode_steps=1000) as c:
        c.reset(); c.train(512, want_stats=False); c.sync()
        c.timing_enable(True); t0 = time.perf_counter(); c.train(2560, want_stats=False); c.sync(); dt = time.perf_counter() - t0
        ms, nl, kn = c.timing_read()
        print(json.dumps({"n_envs": n, "waves_per_simd": n / 65536, "env_steps_per_s": n * 2560 / dt, "us_per_launch": ms * 1e3 / nl}))
with ra.Context(n_envs=1048576, policy=1, epsilon=0.1, max_episode_steps=1000, weight_mode=ra.W_SHARED, lr=1e-9) as c:
    c.reset(); c.train(64, want_stats=False); c.sync(); t0 = time.perf_counter(); c.train(256, want_stats=False); c.sync(); dt = time.perf_counter() - t0
    print(json.dumps({"shared_1M_envs_one_gpu_env_steps_per_s": 1048576 * 256 / dt, "us_per_batch_step": dt / 256 * 1e6}))
