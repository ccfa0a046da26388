#!/usr/bin/env python3
"""Adversarial campaign against the C ABI: configurations with out-of-range, non-finite and contradictory fields, and calls with null pointers, learner
indices and batch sizes out of range, non-finite states, actions outside the action set.  Whatever comes in, every entry point must RETURN (an error
code with a message, or success) -- no crash, no hang, no NaN-poisoned neighbour: after every abuse a small healthy ctx on the same device must still
train to the same checksum.

    python tests/fuzz_abi.py [n_cases=300] [seed=0]        (GPU box; each case runs in this process: a crash ends the campaign with its number)"""
import ctypes as C
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rsrl_amd as ra  # noqa: E402
from rsrl_amd import _abi  # noqa: E402

BAD_F = [float("nan"), float("inf"), -float("inf"), -1.0, 0.0, 1e308, -1e-320, 2.0]
INT_FIELDS = {
    "domain": [-1, 3, 99, 0, 1, 2], "basis": [-1, 2, 0, 1], "order": [-3, 0, 8, 100, 1, 5, 7], "n_tilings": [-1, 0, 3, 5, 64, 4, 8, 16],
    "tiles_per_dim": [-1, 0, 1, 2, 1000, 65536, 8], "algo": [-1, 10, 255, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9], "policy": [-1, 4, 7, 0, 1, 2, 3],
    "weight_mode": [-1, 2, 0, 1], "weight_dtype": [-1, 2, 5, 0, 1], "trace": [-1, 3, 0, 1, 2], "agent_policy": [-2, 4, -1, 0, 1, 2],
    "exchange": [-1, 3, 9, 0, 1, 2], "n_steps": [-1, 0, 33, 1000, 1, 4, 32], "peer_timeout_ms": [-5, 0, 1, 10 ** 9], "device": [-1, 1, 99, 0],
}
F_FIELDS = ["gamma", "lr", "alpha", "epsilon", "tau", "lam", "lr_td", "agent_epsilon", "agent_tau", "sigma", "epsilon_decay", "epsilon_min"]


def healthy_checksum():
    with ra.Context(n_envs=64, policy=1, seed=3, max_episode_steps=50) as c:
        c.reset()
        c.train(40, want_stats=False)
        return c.checksum()


def poke(L, h, rng, log):
    """a created ctx: abuse every entry point that takes arguments"""
    N = int(L.rsrl_hip_n_envs(h)); D = L.rsrl_hip_state_dim(h); A = L.rsrl_hip_n_actions(h); F = L.rsrl_hip_n_features(h); O = L.rsrl_hip_n_outputs(h)
    big = np.zeros(max(1, F * max(O, A)) + 16, dtype=np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)       # noqa: E731
    st = _abi.Stats()
    calls = [
        lambda: L.rsrl_hip_reset(h),
        lambda: L.rsrl_hip_train(h, 3, C.byref(st)),
        lambda: L.rsrl_hip_train(h, 0, None),
        lambda: L.rsrl_hip_train(h, -5, None),
        lambda: L.rsrl_hip_get_weights(h, -1, p(big)), lambda: L.rsrl_hip_get_weights(h, N, p(big)), lambda: L.rsrl_hip_get_weights(h, 2 ** 40, p(big)),
        lambda: L.rsrl_hip_get_weights(h, 0, None), lambda: L.rsrl_hip_set_weights(h, 0, None), lambda: L.rsrl_hip_set_weights(h, N + 7, p(big)),
        lambda: L.rsrl_hip_get_traces(h, N, p(big)), lambda: L.rsrl_hip_set_traces(h, -3, p(big)), lambda: L.rsrl_hip_get_td_weights(h, N, p(big)),
        lambda: L.rsrl_hip_get_states(h, None), lambda: L.rsrl_hip_set_states(h, None), lambda: L.rsrl_hip_set_actions(h, None),
        lambda: L.rsrl_hip_set_episode_steps(h, None), lambda: L.rsrl_hip_set_q_carry(h, None),
        lambda: L.rsrl_hip_q_evaluate(h, None, 4, p(big)), lambda: L.rsrl_hip_q_evaluate(h, p(big), -1, p(big)), lambda: L.rsrl_hip_q_evaluate(h, p(big), 0, p(big)),
        lambda: L.rsrl_hip_save_weights(h, None), lambda: L.rsrl_hip_load_weights(h, None), lambda: L.rsrl_hip_load_weights(h, b"/nonexistent/dir/x"),
        lambda: L.rsrl_hip_save_weights(h, b"/nonexistent/dir/x"),
    ]
    # non-finite states and actions outside the action set through the setters, then a few steps: the ctx may learn garbage, it may not fault
    def bad_state():
        s = np.full((D, N), rng.choice([np.nan, np.inf, -np.inf, 1e30, -1e30]), dtype=np.float32)
        rc = L.rsrl_hip_set_states(h, p(s))
        a = np.full(N, int(rng.choice([-7, A, 2 ** 31 - 1, -2 ** 31])), dtype=np.int32)
        rc2 = L.rsrl_hip_set_actions(h, p(a))
        rc3 = L.rsrl_hip_train(h, 4, None)
        L.rsrl_hip_sync(h)
        return (rc, rc2, rc3)
    calls.append(bad_state)
    def bad_weights():
        w = np.full(F * O, rng.choice([np.nan, np.inf, 3e38]), dtype=np.float32)
        rc = L.rsrl_hip_set_weights(h, 0, p(w))
        rc2 = L.rsrl_hip_train(h, 3, None)
        L.rsrl_hip_sync(h)
        return (rc, rc2)
    calls.append(bad_weights)
    def handle_bad():
        M = int(rng.choice([0, 1, N, N + 5]))
        if M <= 0:
            return L.rsrl_hip_handle(h, p(big), p(big), p(big), p(big), p(big), M, p(big))
        s = np.zeros((D, M), dtype=np.float32); a = np.full(M, int(rng.choice([-1, A + 3, 0])), dtype=np.int32)
        r = np.full(M, rng.choice([np.nan, 1.0]), dtype=np.float32); t = np.full(M, 7, dtype=np.uint8); out = np.zeros(M, dtype=np.float32)
        return L.rsrl_hip_handle(h, p(s), p(a), p(r), p(s), p(t), M, p(out))
    calls.append(handle_bad)
    def eval_bad():                                    # the query entry points on non-finite / huge states
        M = int(rng.choice([1, 5, N]))
        s = np.full((D, M), rng.choice([np.nan, np.inf, -np.inf, 1e30, -3e38]), dtype=np.float32)
        out = np.zeros((max(A, O) + 1) * M + 64, dtype=np.float32); iout = np.zeros(64 * M + 64, dtype=np.int32)
        rcs = [L.rsrl_hip_q_evaluate(h, p(s), M, p(out)), L.rsrl_hip_q_find_max(h, p(s), M, p(iout), p(out)), L.rsrl_hip_policy_mode(h, p(s), M, p(iout))]
        if hasattr(L, "rsrl_hip_tile_indices"):
            rcs.append(L.rsrl_hip_tile_indices(h, p(s), M, p(iout)))
        L.rsrl_hip_sync(h)
        return rcs
    calls.append(eval_bad)
    def rollout_bad():
        n_st = np.zeros(N, dtype=np.uint32); tot = np.zeros(N, dtype=np.float32)
        return [L.rsrl_hip_rollout_greedy(h, int(v), p(n_st), p(tot)) for v in (-1, 1, 2, 30)]
    calls.append(rollout_bad)
    def ranks_bad():                                   # the multi-rank entry points: null lists, duplicates, counts and ranks out of range, garbage handles
        hbuf = (C.c_uint8 * (128 * 4))(*[int(x) for x in rng.integers(0, 256, 512)])
        two = (C.c_void_p * 2)(h, h); nul = (C.c_void_p * 2)(h, None)
        ws, rk, ex = C.c_int(0), C.c_int(0), C.c_int(0); ident = C.c_uint64(0)
        rcs = [L.rsrl_hip_peer_export(h, int(v), None) for v in (-1, 0, 1)]
        rcs += [L.rsrl_hip_peer_export(h, int(v), hbuf) for v in (-1, 0, 100000)]
        rcs += [L.rsrl_hip_peer_connect(h, None, 2, 0), L.rsrl_hip_peer_connect(h, hbuf, 2, 5), L.rsrl_hip_peer_connect(h, hbuf, 0, 0),
                L.rsrl_hip_peer_connect(h, hbuf, 2, int(rng.integers(0, 2)))]
        rcs += [L.rsrl_hip_comm_init(h, None, 1, 0), L.rsrl_hip_comm_init(h, hbuf, 0, 0), L.rsrl_hip_comm_init(h, hbuf, 2, 7), L.rsrl_hip_comm_init(h, hbuf, -1, -1)]
        rcs += [L.rsrl_hip_group_create(None, 2), L.rsrl_hip_group_create(two, 0), L.rsrl_hip_group_create(two, -3), L.rsrl_hip_group_create(two, 2),
                L.rsrl_hip_group_create(nul, 2), L.rsrl_hip_group_train(None, 1, 3), L.rsrl_hip_group_train(two, 2, 2), L.rsrl_hip_group_train(nul, 2, 2),
                L.rsrl_hip_group_train(two, 1, -4)]
        rcs += [L.rsrl_hip_can_access_peer(-1, 0), L.rsrl_hip_can_access_peer(0, 99), L.rsrl_hip_device_identity(99, C.byref(ident)), L.rsrl_hip_device_identity(0, None),
                L.rsrl_hip_comm_info(h, None, None, None), L.rsrl_hip_comm_info(h, C.byref(ws), C.byref(rk), C.byref(ex)), L.rsrl_hip_comm_unique_id(None)]
        rcs.append(L.rsrl_hip_train(h, 2, None))
        L.rsrl_hip_sync(h)
        return rcs
    calls.append(ranks_bad)
    order = rng.permutation(len(calls))[: int(rng.integers(3, 12))]
    for j in order:
        rc = calls[int(j)]()
        log.append((int(j), rc if not isinstance(rc, tuple) else list(rc)))
    L.rsrl_hip_sync(h)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    L = _abi.lib()
    ref = healthy_checksum()
    created = refused = 0
    for idx in range(n_cases):
        cfg = _abi.Config()
        L.rsrl_hip_config_init(C.byref(cfg))
        cfg.n_envs = int(rng.choice([1, 3, 64, 300, 2000]))
        cfg.seed = int(rng.integers(0, 1 << 30))
        # a plausible base, then 1-4 fields pushed out of range
        cfg.domain = int(rng.integers(0, 3)); cfg.basis = int(rng.integers(0, 2)); cfg.algo = int(rng.integers(0, 10)); cfg.policy = int(rng.integers(0, 4))
        cfg.order = int(rng.choice([1, 2, 3, 5, 7])); cfg.weight_mode = int(rng.random() < 0.25)
        for _ in range(int(rng.integers(0, 5))):
            k = rng.integers(0, 4)
            if k == 0:
                name = str(rng.choice(list(INT_FIELDS)))
                setattr(cfg, name, int(rng.choice(INT_FIELDS[name])))
            elif k == 1:
                setattr(cfg, str(rng.choice(F_FIELDS)), float(rng.choice(BAD_F)))
            elif k == 2:
                cfg.n_envs = int(rng.choice([0, -1, -2 ** 40, 2 ** 62, 1]))
            else:
                name = str(rng.choice(["env_offset", "max_episode_steps", "steps_per_launch", "struct_size"]))
                val = {"env_offset": [-1, 2 ** 40, 2 ** 32 - 1], "max_episode_steps": [0, 1, 2 ** 32 - 1], "steps_per_launch": [0, 1, 2 ** 32 - 1, 3],
                       "struct_size": [0, 4, 17, 10 ** 6, C.sizeof(_abi.Config) - 8]}[name]
                setattr(cfg, name, int(rng.choice(val)))
        # never ask for more than ~2 GB: a table of F*A floats (twice with an auxiliary matrix) per learner
        feats = (max(1, min(8, cfg.order)) + 1) ** (2 if cfg.domain == 0 else 4) if cfg.basis == 0 else max(1, min(16, cfg.n_tilings)) * max(1, min(64, cfg.tiles_per_dim)) ** (2 if cfg.domain == 0 else 4)
        if 0 < cfg.n_envs <= 10 ** 7 and cfg.weight_mode == 0 and feats * 3 * 4 * 2 * cfg.n_envs > 2e9:
            cfg.n_envs = max(1, int(2e9 / (feats * 24)))
        if cfg.peer_timeout_ms == 0 or cfg.peer_timeout_ms > 300:
            cfg.peer_timeout_ms = 200                               # (a rank whose peers never show up gives up after 0.2 s instead of the default 4 s)
        h = C.c_void_p()
        rc = L.rsrl_hip_create(C.byref(cfg), C.byref(h))
        log = []
        if rc == 0 and h:
            created += 1
            poke(L, h, rng, log)
            L.rsrl_hip_destroy(h)
        else:
            refused += 1
            msg = L.rsrl_hip_last_error()
            assert msg and len(msg) > 3, (idx, rc)
        now = healthy_checksum()
        ok = now == ref
        print(f"{idx:4d} create rc {rc:3d} {'poked ' + str(len(log)) if rc == 0 else 'refused'}  healthy {'ok' if ok else 'CHANGED'}", flush=True)
        if not ok:
            print("SUMMARY " + json.dumps({"cases": idx + 1, "failed_at": idx, "log": log}), flush=True)
            sys.exit(1)
    # null handles
    for fn in ("rsrl_hip_reset", "rsrl_hip_sync", "rsrl_hip_destroy"):
        getattr(L, fn)(None)
    print("SUMMARY " + json.dumps({"cases": n_cases, "seed": seed, "created": created, "refused": refused, "failures": []}), flush=True)


if __name__ == "__main__":
    main()
