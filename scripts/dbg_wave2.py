import os, sys, subprocess
sys.path.insert(0, "/root/repo")
import numpy as np
if len(sys.argv) > 1:
    import rsrl_amd as ra
    kw = dict(domain=2, order=7, algo=2, policy=2, n_envs=9, seed=5, max_episode_steps=12, lr=0.01, gamma=0.99, weight_dtype=ra.W_BF16)
    with ra.Context(steps_per_launch=1, **kw) as c:
        c.reset()
        c.train(1)
        np.save(sys.argv[1], c.get_weights(0))
else:
    W = {}
    for pk in ("1", "0"):
        f = f"/tmp/w_{pk}.npy"
        subprocess.check_call([sys.executable, __file__, f], env=dict(os.environ, RSRL_WAVE_PK=pk))
        W[pk] = np.load(f)
    a, b = W["1"], W["0"]          # (F, A) reference order f = (k - 1) mod F
    d = a != b
    print("mismatching entries", d.sum(), "of", a.size, "per action", d.sum(axis=0))
    f = np.flatnonzero(d.any(axis=1))
    k = (f + 1) % 4096
    print("first mismatching k:", k[:40])
    print("k mod 8 histogram", np.bincount(k % 8, minlength=8), "chunk j histogram", np.bincount(k // 512, minlength=8), "lane hist (first 16)", np.bincount((k % 512) // 8, minlength=64)[:16])
    col = int(np.argmax(d.sum(axis=0)))
    for kk in k[:8]:
        ff = (kk - 1) % 4096
        print(kk, a[ff, col], b[ff, col], hex(a[ff, col].view(np.uint32)), hex(b[ff, col].view(np.uint32)))
    # is pk's matrix a permutation of old's within pairs?
    ka = np.arange(4096); fa = (ka - 1) % 4096
    A_k, B_k = a[fa, col], b[fa, col]
    sw = B_k.reshape(-1, 2)[:, ::-1].reshape(-1)
    print("equal to pair-swapped old:", np.array_equal(A_k, sw), " equal count", (A_k == B_k).sum(), "swapped-equal count", (A_k == sw).sum())
