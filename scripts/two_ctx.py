import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rsrl_amd as ra
def run(ctxs, steps=2560):
    for c in ctxs:
        c.reset(); c.train(512, want_stats=False)
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    for k in range(steps // 256):
        for c in ctxs: c.train(256, want_stats=False)
    for c in ctxs: c.sync()
    dt = time.perf_counter() - t0
    return sum(c.N for c in ctxs) * steps / dt
kw = dict(policy=1, epsilon=0.1, max_episode_steps=1000)
with ra.Context(n_envs=65536, **kw) as a:
    print("one ctx 65536      ", "%.3g" % run([a]))
with ra.Context(n_envs=131072, **kw) as a:
    print("one ctx 131072     ", "%.3g" % run([a]))
with ra.Context(n_envs=65536, **kw) as a, ra.Context(n_envs=65536, env_offset=65536, **kw) as b:
    print("two ctxs 2 x 65536 ", "%.3g" % run([a, b]))
with ra.Context(n_envs=65536, **kw) as a, ra.Context(n_envs=65536, env_offset=65536, **kw) as b, ra.Context(n_envs=65536, env_offset=131072, **kw) as c, ra.Context(n_envs=65536, env_offset=196608, **kw) as d:
    print("four ctxs 4 x 65536", "%.3g" % run([a, b, c, d]))
with ra.Context(n_envs=262144, **kw) as a:
    print("one ctx 262144     ", "%.3g" % run([a]))
