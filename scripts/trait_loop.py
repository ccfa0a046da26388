#!/usr/bin/env python3
"""The TRAIT-GRANULAR loop a drop-in user writes (rsrl/examples/q_learning.rs:40-52), one C-ABI call per trait method, device pointers only:

    t  = env.transition(a)         rsrl_hip_domain_step(actions, from, to, reward, terminal)
         agent.handle(&t)          rsrl_hip_handle(from, actions, reward, to, terminal)
         terminal -> new episode   rsrl_hip_domain_reset(terminal)          (examples/q_learning.rs:37, :47-51)
    a' = policy.sample(rng, env.emit().state())    rsrl_hip_policy_sample(NULL = the ctx's own envs, actions)

measure(...) -> dict: wall-clock us per batch-step (everything enqueued, one synchronize at the end), host-side us per call.
Used by bench.py (`trait_loop` leg) and stand-alone:  python scripts/trait_loop.py [n_envs] [steps]
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# bytes the loop must move per env-step if every call streams what it needs exactly once (MountainCar D = 2, Fourier(5): F = 36, A = 3)
#   domain_step : state in 8 + action 4, state out 8, from 8, to 8, reward 4, terminal 1                          = 41
#   handle      : from 8 + action 4 + reward 4 + to 8 + terminal 1, W read 432, touched column written 144        = 601
#   domain_reset: terminal 1 (+ the few restarted states)                                                           = 1
#   sample      : state 8 + the handed-over Q(s',.) 12 and its key 8 (instead of W 432 again), action out 4 + the ctx's pending action 4  = 36
ALG_BYTES = {"domain_step": 41, "handle": 601, "domain_reset": 1, "policy_sample_handover": 8 + 8 + 12 + 4 + 4, "policy_sample_reeval": 8 + 432 + 4 + 4}
ALG_BYTES_LOOP = 41 + 601 + 1 + 36          # 679 B per env-step with the Q(s',.) hand-over (a second pass over W is an implementation choice, not algorithmic)


def measure(n_envs=65536, steps=400, warmup=50, device=0, steps_per_launch=0, per_call=False, **ctx_kw):
    import rsrl_amd
    from rsrl_amd import _abi
    from rsrl_amd._devmem import DeviceBuffer
    kw = dict(domain=rsrl_amd.MOUNTAIN_CAR, basis=rsrl_amd.FOURIER, order=5, algo=rsrl_amd.QLEARNING, policy=rsrl_amd.EPSILON_GREEDY, epsilon=0.1,
              gamma=0.9, lr=0.001, n_envs=n_envs, seed=0, max_episode_steps=0, steps_per_launch=steps_per_launch, device=device)
    kw.update(ctx_kw)
    ctx = rsrl_amd.Context(**kw)
    L, h, N, D = ctx._L, ctx._h, ctx.N, ctx.D
    frm, to = (DeviceBuffer(D * N, "float32", device) for _ in range(2))
    rew = DeviceBuffer(N, "float32", device)
    act = DeviceBuffer(N, "int32", device)
    term = DeviceBuffer(N, "uint8", device)
    p = lambda b: C.c_void_p(b.ptr)      # noqa: E731
    ctx.reset()
    _abi.check(L.rsrl_hip_get_actions(h, p(act)))
    calls = [("domain_step", lambda: L.rsrl_hip_domain_step(h, p(act), p(frm), p(to), p(rew), p(term))),
             ("handle", lambda: L.rsrl_hip_handle(h, p(frm), p(act), p(rew), p(to), p(term), N, None)),
             ("domain_reset", lambda: L.rsrl_hip_domain_reset(h, p(term))),
             ("policy_sample", lambda: L.rsrl_hip_policy_sample(h, None, N, p(act)))]

    def run(k, host_t=None):
        for _ in range(k):
            for j, (_, fn) in enumerate(calls):
                if host_t is not None:
                    t0 = time.perf_counter()
                    rc = fn()
                    host_t[j] += time.perf_counter() - t0
                else:
                    rc = fn()
                if rc != 0:
                    _abi.check(rc)
    run(warmup)
    ctx.sync()
    t0 = time.perf_counter()
    run(steps)
    ctx.sync()
    dt = time.perf_counter() - t0
    out = {"learners": N, "steps": steps, "us_per_batch_step": dt / steps * 1e6, "value": N * steps / dt, "unit": "env-steps/s",
           "calls_per_step": len(calls), "algorithmic_bytes_per_env_step": ALG_BYTES_LOOP,
           "frac_of_8TBps": ALG_BYTES_LOOP * N * steps / dt / 8.0e12}
    # HIP events around every launch of the trait kernels, in a run of its own (two event records per launch cost host time).  For these ~8 us kernels
    # the interval between the two events of a launch includes its dispatch latency (~2-3 us): an upper bound of the kernel's duration, which
    # rocprofv3's kernel trace gives exactly (profiles/r06_kernel_stats_trait.md)
    ctx.timing_enable(True)
    run(min(steps, 200))
    ms, n_l, kn = ctx.timing_read()
    ctx.timing_enable(False)
    out["kernel"] = kn
    out["hip_event_launches"] = n_l
    out["hip_event_us_per_batch_step"] = ms * 1e3 / max(1, min(steps, 200))
    out["hip_event_covers"] = ("the one fused launch per batch-step, dispatch latency included" if kn.endswith("<step>") else
                               "handle + sample only (k_domain_step / k_domain_reset are not bracketed), dispatch latency included")
    if per_call:
        host_t = [0.0] * len(calls)
        ctx.sync()
        run(steps, host_t)
        ctx.sync()
        out["host_us_per_call"] = {name: host_t[j] / steps * 1e6 for j, (name, _) in enumerate(calls)}
        # each call alone, synchronised: its kernel + launch latency
        alone = {}
        for name, fn in calls:
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(100):
                fn()
            ctx.sync()
            alone[name] = (time.perf_counter() - t0) / 100 * 1e6
        out["us_per_call_back_to_back"] = alone
    weights_sum = float(abs(ctx.get_weights(0)).sum())
    out["learner0_sum_abs_w"] = weights_sum
    ctx.close()
    for b in (frm, to, rew, act, term):
        b.free()
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    spl = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    print(json.dumps(measure(n, k, per_call=True, steps_per_launch=spl)))
