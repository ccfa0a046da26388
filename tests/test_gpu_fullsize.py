"""Size-independent properties at BASELINE.json's full sizes (the oracle cannot finish these in seconds):
determinism, sharding invariance, fusion invariance (checksums of checksums computed on the device), index
ranges, reward/episode bookkeeping identities."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rsrl_amd
    return rsrl_amd


C2 = dict(policy=1, epsilon=0.1, gamma=0.9, lr=0.001, seed=0, max_episode_steps=1000)


def test_c2_full_size_determinism_fusion_and_sharding(ra):
    N, K = 65536, 192
    sums = []
    for spl in (64, 64, 1):                               # same seed twice, then one step per launch
        with ra.Context(n_envs=N, steps_per_launch=spl, **C2) as c:
            c.reset()
            st = c.train(K)
            sums.append(c.checksum())
            assert st["env_steps"] == N * K
            # MountainCar: reward -1 on every non-terminal transition, 0 on the terminal one (discrete.rs:84-95)
            assert st["sum_reward"] == -(st["env_steps"] - (st["episodes"] - st["episodes_truncated"]))
            assert st["sum_episode_steps"] <= st["env_steps"]
    assert sums[0] == sums[1], "same seed must give bit-identical weights and states"
    assert sums[0] == sums[2], "K-step fusion must not change a single bit"
    # two half-size shards keyed by global env id == the full run (word-index-weighted checksums do not add up across
    # shards, so compare the per-shard slices of the full run instead)
    with ra.Context(n_envs=N, **C2) as full, ra.Context(n_envs=N // 2, env_offset=N // 2, **C2) as hi:
        full.reset(), hi.reset()
        full.train(K), hi.train(K)
        assert np.array_equal(full.states[:, N // 2:], hi.states)
        assert np.array_equal(full.actions[N // 2:], hi.actions)
        for i in (0, 12345, N // 2 - 1):
            assert np.array_equal(full.get_weights(N // 2 + i), hi.get_weights(i))


def test_c3_full_size_tiles(ra):
    N, T, B = 262144, 8, 8
    with ra.Context(domain=1, basis=ra.TILE_CODING, n_tilings=T, tiles_per_dim=B, algo=ra.SARSA, policy=1, epsilon=0.1, gamma=0.99,
                    lr=0.0125 / N, weight_mode=ra.W_SHARED, n_envs=N, seed=3, max_episode_steps=200) as c:
        c.reset()
        st = c.train(40)
        idx = c.tile_indices(c.states)
        W = c.get_weights()
    assert idx.shape == (T, N)
    for t in range(T):                                    # tiling t owns the index block [t*B^4, (t+1)*B^4)
        assert idx[t].min() >= t * B ** 4 and idx[t].max() < (t + 1) * B ** 4
    assert np.all(np.isfinite(W)) and np.count_nonzero(W) > 0
    assert st["env_steps"] == N * 40 and st["episodes"] > 0
    # CartPole: reward -1 exactly on terminal transitions (cart_pole.rs:99-110)
    assert st["sum_reward"] == -(st["episodes"] - st["episodes_truncated"])


def test_c4_full_size_shared_weights_reproducible(ra):
    N = 1048576                                            # the whole 8-GPU batch on one device: state is tiny
    kw = dict(C2, n_envs=N, weight_mode=ra.W_SHARED, lr=0.001 / N)
    out = []
    for _ in range(2):
        with ra.Context(**kw) as c:
            c.reset()
            st = c.train(25)
            out.append((c.checksum(), c.get_weights().copy()))
            assert st["env_steps"] == N * 25
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])     # fixed-order delta reduction
    assert np.max(np.abs(out[0][1])) > 0


def test_c5_full_size_bf16_wave_family(ra):
    N = 65536                                              # 65 536 x 3 x 4096 bf16 = 1.6 GB of weights
    kw = dict(domain=2, order=7, algo=ra.EXPECTED_SARSA, policy=ra.SOFTMAX, tau=1.0, gamma=0.99, lr=0.001, alpha=1.0, n_envs=N,
              weight_dtype=ra.W_BF16, seed=5, max_episode_steps=1000)
    sums = []
    for spl in (6, 1):
        with ra.Context(steps_per_launch=spl, **kw) as c:
            c.reset()
            st = c.train(6)
            sums.append(c.checksum())
            w = c.get_weights(N - 1)
            assert np.all((w.view(np.uint32) & 0xffff) == 0) and np.all(np.isfinite(w))
            assert st["env_steps"] == N * 6
    assert sums[0] == sums[1]


def test_lambda_full_size_determinism(ra):
    kw = dict(n_envs=65536, algo=ra.SARSA_LAMBDA, policy=1, epsilon=0.2, gamma=0.99, alpha=0.01, lam=0.7, trace=ra.TRACE_SATURATE,
              seed=0, max_episode_steps=1000)
    sums = []
    for spl in (128, 16):
        with ra.Context(steps_per_launch=spl, **kw) as c:
            c.reset()
            c.train(128)
            sums.append(c.checksum())
    assert sums[0] == sums[1]


def test_c2_full_size_sampled_learners_bitwise_vs_oracle(ra, orc):
    # BASELINE.json configs[1] at its FULL size (65 536 learners), 2 000 batch-steps: the learners are independent and their RNG
    # streams are keyed by the global env id, so the CPU oracle (device order, f32d) can replay any slice of the batch on its
    # own -- three slices of 192 learners (first wave, an interior block boundary, the last learners) must match the full-size
    # device run bit for bit: states, actions, every weight
    import threading
    N, K, M = 65536, 2000, 192
    offs = (0, 32768 - 96, N - M)
    runs, th = {}, []

    def cpu(off):
        r = orc.Run(orc.make_agent(policy=orc.EGREEDY, epsilon=0.1, gamma=0.9, lr=0.001, seed=0, max_episode_steps=1000, env_offset=off), M, "f32d")
        r.reset()
        r.train_dev(K)
        runs[off] = r
    for off in offs:
        th.append(threading.Thread(target=cpu, args=(off,)))
        th[-1].start()
    with ra.Context(n_envs=N, **C2) as c:
        c.reset()
        c.train(K, want_stats=False)
        S, A = c.states, c.actions
        [t.join() for t in th]
        for off in offs:
            r = runs[off]
            assert np.array_equal(S[:, off:off + M].T, r.state) and np.array_equal(A[off:off + M], r.action), off
            for i in range(0, M, 7):
                assert np.array_equal(c.get_weights(off + i), r.weights[i]), (off, i)
        n_dev, _ = c.rollout_greedy(400)
    for off in offs:
        n_cpu, _ = runs[off].rollout_greedy(400)
        assert np.array_equal(n_dev[off:off + M], n_cpu)


def test_c5_full_size_sampled_learners_bitwise_vs_oracle(ra, orc):
    # BASELINE.json configs[4]'s per-GPU share (32 768 Acrobot learners, ExpectedSARSA + Fourier(7) + Softmax, bf16 weights):
    # 24 sampled learners x 60 batch-steps of the full-size device run against the oracle's wave-order loop, bit for bit
    N, K, M = 32768, 60, 8
    kw = dict(domain=2, order=7, algo=2, policy=2, tau=1.0, gamma=0.99, lr=0.001, alpha=1.0, seed=5, max_episode_steps=1000)
    offs = (0, 16384 - 4, N - M)
    runs = {}
    for off in offs:
        r = orc.Run(orc.make_agent(env_offset=off, **kw), M, "f32d")
        r.reset_wave()
        r.train_wave(K, bf16=True)
        runs[off] = r
    with ra.Context(n_envs=N, weight_dtype=ra.W_BF16, **kw) as c:
        c.reset()
        c.train(K, want_stats=False)
        S, A = c.states, c.actions
        for off in offs:
            r = runs[off]
            assert np.array_equal(S[:, off:off + M].T, r.state) and np.array_equal(A[off:off + M], r.action), off
            for i in (0, M - 1):
                assert np.array_equal(c.get_weights(off + i), r.weights[i]), (off, i)


def test_streaming_kernel_beyond_the_infinity_cache_bitwise_vs_fused(ra):
    # k_step_reg_q4 writes the touched column back as whole 128-byte lines once the weights exceed 256 MiB (64-byte sectors below):
    # 700 003 learners x 432 B = 302 MB, a last wave of three learners -- against the fused loop (weights in registers), bit for bit
    N = 700003
    kw = dict(C2, n_envs=N)
    ref = None
    for spl in (8, 1):
        with ra.Context(steps_per_launch=spl, **kw) as c:
            c.reset()
            c.timing_enable(True)
            st = c.train(8)
            assert st["env_steps"] == N * 8
            got = (c.checksum(), c.get_weights(0).copy(), c.get_weights(N // 2 + 5).copy(), c.get_weights(N - 1).copy(), c.states.copy())
            if spl == 1:
                assert c.timing_read()[2] == "k_step_reg_q4", c.timing_read()
        if ref is None:
            ref = got
        else:
            assert got[0] == ref[0]
            assert all(np.array_equal(a, b) for a, b in zip(got[1:], ref[1:]))
    assert np.abs(ref[1]).max() > 0
