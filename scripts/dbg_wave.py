import os, sys, json, subprocess
sys.path.insert(0, "/root/repo")
if len(sys.argv) > 1:
    import numpy as np
    import rsrl_amd as ra
    spl = int(sys.argv[1])
    kw = dict(domain=2, order=7, algo=2, policy=2, n_envs=9, seed=5, max_episode_steps=12, lr=0.01, gamma=0.99, weight_dtype=ra.W_BF16)
    out = []
    with ra.Context(steps_per_launch=spl, **kw) as c:
        c.reset()
        for k in range(0, 30, spl if spl <= 30 else 30):
            c.train(min(spl, 30) if spl > 1 else 1)
            out.append((c.states.copy(), c.actions.copy(), c.get_weights(0).copy(), c.get_weights(8).copy()))
    np.save(sys.argv[2], np.array([np.concatenate([o[0].ravel(), o[1].ravel().astype(np.float32), [np.abs(o[2]).sum(), np.abs(o[3]).sum()]]) for o in out]))
else:
    import numpy as np
    res = {}
    for pk in ("1", "0"):
        for spl in (1, 5, 30):
            f = f"/tmp/dbg_{pk}_{spl}.npy"
            subprocess.check_call([sys.executable, __file__, str(spl), f], env=dict(os.environ, RSRL_WAVE_PK=pk))
            res[(pk, spl)] = np.load(f)
    ref = res[("0", 1)]
    print("old fused30 == old step1 (last):", np.array_equal(res[("0", 30)][-1], ref[-1]))
    for spl in (1, 5, 30):
        x = res[("1", spl)]
        n = 30 // spl if spl <= 30 else 1
        for k in range(x.shape[0]):
            r = ref[(k + 1) * spl - 1]
            if not np.array_equal(x[k], r):
                bad = np.flatnonzero(x[k] != r)
                print("pk spl", spl, "first diff after", (k + 1) * spl, "steps at idx", bad[:12], x[k][bad[:6]], r[bad[:6]])
                break
        else:
            print("pk spl", spl, "all equal")
