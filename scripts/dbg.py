import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsrl_amd as ra
M = 64
c = ra.Context(domain=1, basis=ra.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=ra.QLEARNING, n_envs=M, policy=0, gamma=0.99, lr=0.5, weight_mode=ra.W_SHARED)
s = np.zeros((4, M), dtype=np.float32)
s[0, 32:] = 1.0     # two groups of states
a = (np.arange(M) % 2).astype(np.int32)
r = np.full(M, -1.0, dtype=np.float32)
term = np.ones(M, dtype=np.uint8)
td = c.handle(s, a, r, s, term)
W = c.get_weights()
idx = c.tile_indices(s)
print("td", td[:4], "nonzero W", np.count_nonzero(W), "sum", W.sum())
print("expected sum", 8 * M * 0.5 * -1.0)
for t in range(2):
    print("tiling", t, "idx", idx[t, 0], idx[t, 40], "W", W[idx[t, 0]], W[idx[t, 40]])
