#!/usr/bin/env python3
"""bench.py -- env-steps/s of the TD-control hot path on MI355X.

Workload (BASELINE.json configs[1]): 65 536 vectorised MountainCar envs per GPU, QLearning + Fourier(5)
+ epsilon-greedy(0.1), gamma 0.9, SGD(0.001), per-env weights, max_episode_steps 1000, synthetic seeded
episodes (all envs start at MountainCar::default(); diversity comes from the per-env Philox streams).
A "step" = one pass of the hot path (transition -> handle -> sample) over the whole batch of envs.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (WORLD_SIZE unset): bench.py starts the N ranks itself (torch.distributed.run, one rank per
GPU, rendezvous on 127.0.0.1).  Under a launcher it is one of the ranks.  Envs are sharded by global id with NO
data-path collective (independent learners) => weak scaling.

Timed regions (warm-up, >= 2 000 batch-steps, median of 5): a K-step call is short (K = 20 is ~15 us of
arithmetic), so the call is repeated R times back to back between ONE pair of barrier + synchronize (R is chosen so that a
region lasts >= ~1 s; config.repeats); the region is measured FIVE times and `value` / `ms_per_step` are the MEDIAN region
(all ranks' env-steps / max-over-ranks region time); `regions_s` and `spread` carry all five.  rsrl_hip_train is asynchronous
and, on its own stream, coalesces calls that arrive while the stream is busy (same results bit for bit, include/rsrl_hip.h):
the R x K batch-steps run as launches of up to 4096 steps -- config.steps_per_launch and roofline.launches report what was
actually launched, and `value_no_coalesce` is the same K-step driver call with one launch per call (RSRL_NO_COALESCE=1).

roofline.frac is a fraction of a PUBLISHED peak (MI355X_MICROARCH.md): the fused kernel keeps W in the register file, so its
bound is the fp32 vector unit -- frac = flop per env-step (profiles/isa_mix.json, from rocprofv3 instruction-class counters) x
measured kernel rate / 157.3 TFLOP/s; `issue_slots` prices the same instruction mix at the guide's issue rates (2 cycles per
plain VALU, 4 per packed); the micro-benchmarked ceiling of round 2 survives as a clearly named secondary field.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ENVS = 65536
BYTES_PER_ENV_STEP = 608            # 2*D*4 + 8 + 8 + F*A*4 (W read) + F*4 (W column write)
HBM_PEAK = 8.0e12                   # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_VECTOR_PEAK = 157.3e12         # MI355X_MICROARCH.md: peak FP32 (vector), spec
N_SIMD, CLOCK_HZ = 1024, 2.4e9      # 256 CUs x 4 SIMDs, max clock
# HBM bytes one launch of the fused kernel must move per learner, whatever its depth: W in + W out (2 x 432), state in/out
# (2 x 8), action in/out (2 x 4), episode counter in/out (2 x 4), carried Q in/out (2 x 12)
FUSED_BYTES_PER_LEARNER_LAUNCH = 2 * (432 + 8 + 4 + 4 + 12)
TARGET_REGION_S = 1.0               # per timed region; the region is measured N_REGIONS times, `value` is the median
N_REGIONS = 5
SPEC_CYCLES = {"pk": 4.0, "other": 2.0}   # MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cycles on a SIMD-32; a packed fp32 op = two of them


def usable_cores():
    """Host cores this process may actually use (affinity mask and cgroup cpu quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def _profiles_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


_LIVE_DIGESTS = {}


def profile_digest(kname, tables=("isa_mix.json", "pmc_traffic.json")):
    """Do the committed per-kernel constants (profiles/isa_mix.json: flop per env-step; pmc_traffic.json: HBM bytes) belong to the binary this
    process runs?  scripts/summarize_profile.py stamps every entry with the sha256 of the profiled kernel's machine code; this recomputes it from
    the library bench.py loaded (rsrl_amd/_kdigest.py: every instantiation of the kernel template inside librsrl_hip.so's gfx950 code objects).
    -> {"profile_digest_matches": bool, "kernel_code_sha256": live, "profiled_code_sha256": {table: stamped}}.  A kernel edit without a
    re-profile flips the flag, and every fraction that multiplies a live rate by those constants is then printed as null."""
    from rsrl_amd import _build, _kdigest
    lib = os.environ.get("RSRL_HIP_LIB", _build.LIB_PATH)
    if kname not in _LIVE_DIGESTS:
        try:
            _LIVE_DIGESTS[kname] = _kdigest.kernel_digests(lib, [kname])[kname]
        except Exception:      # noqa: BLE001
            _LIVE_DIGESTS[kname] = None
    live, stamped = _LIVE_DIGESTS[kname], {}
    for t in tables:
        rec = (_profiles_json(t) or {}).get(kname)
        rec = rec[0] if isinstance(rec, list) and rec else rec
        stamped[t] = rec.get("code_sha256") if isinstance(rec, dict) else None
    ok = live is not None and all(v == live for v in stamped.values())
    return {"profile_digest_matches": bool(ok), "kernel_code_sha256": live, "profiled_code_sha256": stamped}


def gate_on_profile(rl, kernels):
    """stamp a roofline object with the digest check of its kernels; when the committed profile is not of this binary, every fraction derived from it
    is withheld (null) and survives only under a name that says so"""
    checks = {k: profile_digest(k) for k in kernels}
    ok = all(c["profile_digest_matches"] for c in checks.values())
    rl["profile_digest_matches"] = ok
    rl["profile_digest"] = checks if len(checks) > 1 else next(iter(checks.values()))
    if not ok:
        stale = {}
        for key in ("frac", "useful_frac"):
            if rl.get(key) is not None:
                stale[key] = rl[key]
                rl[key] = None
        for sub in ("hbm", "issue_slots"):
            if isinstance(rl.get(sub), dict):
                for key in ("frac", "frac_lower_bound"):
                    if rl[sub].get(key) is not None:
                        stale[f"{sub}.{key}"] = rl[sub][key]
                        rl[sub][key] = None
        if rl.get("traffic") is not None:
            stale["traffic"] = rl["traffic"]
            rl["traffic"] = None
        rl["from_stale_profile"] = dict(stale, note="profiles/isa_mix.json / pmc_traffic.json were taken from OTHER machine code than this library's: "
                                                    "re-run scripts/gpu_profile.sh + scripts/summarize_profile.py")
    return rl


def pmc_traffic(kernel, envs, steps_per_launch):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json: FETCH_SIZE and
    WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes), or None when no pass was collected for this
    kernel / configuration.  bench.py cannot run rocprofv3 on itself."""
    rec = (_profiles_json("pmc_traffic.json") or {}).get(kernel)
    recs = [r for r in (rec if isinstance(rec, list) else [rec] if rec else []) if r.get("envs") == envs]
    if steps_per_launch is None:                     # a kernel whose bytes per launch do not depend on the launch's depth, or that has one depth only
        return recs[0]["traffic_bytes_per_launch"] if recs else None
    for r in recs:
        if abs(r.get("steps_per_launch", -1) - steps_per_launch) < 1e-9:
            return r["traffic_bytes_per_launch"]
    # the fused loop moves the same bytes per launch whatever its depth (W in + out, state: measured equal at 256 and at 20
    # steps per launch): a launch of another depth takes the deepest measured one
    if kernel == "k_train_reg" and recs and steps_per_launch > 1:
        return max(recs, key=lambda r: r.get("steps_per_launch", 0))["traffic_bytes_per_launch"]
    return None


def cpu_baseline(seconds=9.0, seconds_optimised=5.0):
    """The CPU oracle timed on this box's host cores, one independent group of learners per core.
    value     : the reference-faithful f64 port -- the reference's call pattern (4 projections per step, a heap-allocated
                feature vector per call, one learner at a time): what rsrl's own loop does on these cores.
    optimised : the same computation (bit-identical results, tests/test_oracle_golden.py) with the repeated projections and
                the heap traffic removed (phi(s), Q(s,.) carried; 1 projection per step), so that the GPU/CPU ratio is not
                inflated by the reference's call pattern."""
    from oracle import oracle as orc
    cores = usable_cores()
    envs_per_thread, chunk = 16, 250

    def timed(fast, secs):
        counts = [0] * cores
        t_end = time.perf_counter() + secs

        def work(tid):
            ag = orc.make_agent(policy=orc.EGREEDY, epsilon=0.1, seed=0, env_offset=tid * envs_per_thread,
                                gamma=0.9, lr=0.001, max_episode_steps=1000)
            run = orc.Run(ag, envs_per_thread, "f64")
            run.reset()
            while time.perf_counter() < t_end:
                (run.train_fast if fast else run.train)(chunk)
                counts[tid] += envs_per_thread * chunk
            run.close()

        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
        [t.start() for t in th]
        [t.join() for t in th]
        return sum(counts), time.perf_counter() - t0

    total, dt = timed(False, seconds)
    total_o, dt_o = timed(True, seconds_optimised)
    return {"value": total / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "per_core": total / dt / cores,
            "optimised": {"value": total_o / dt_o, "per_core": total_o / dt_o / cores,
                          "what": "same results, 1 projection per step, no heap traffic (orc_run_train_fast)"},
            "sample": f"{cores} threads x {envs_per_thread} f64 learners, MountainCar QLearning Fourier(5) "
                      f"eps-greedy, {total} env-steps in {dt:.1f} s (reference call pattern) + {total_o} env-steps in "
                      f"{dt_o:.1f} s (optimised), oracle/rsrl_oracle.c, gcc -O2"}


def greedy_rollout_check(ctx, sample=256, limit=1000):
    """north_star: "the 1-GPU greedy rollout length matching the CPU reference".  The device's greedy rollout (Domain::rollout with
    policy.mode, lib.rs:448-479) over ALL learners; then the f64 CPU oracle's rollout from the very same weights for `sample` learners
    whose device rollout ENDED before the limit (a rollout that runs into the limit on both sides says nothing) -- fewer than `sample`
    terminated: the rest of the sample are the first learners that did not.  min_argmax_margin: the smallest gap between the best and
    the second best action value over every action selection of the compared f64 rollouts (the comparison is meaningful
    where it is above fp32 resolution, ~1e-7 of |Q|)."""
    import numpy as np
    from oracle import oracle as orc
    n_dev, _ = ctx.rollout_greedy(limit)
    m = min(sample, ctx.N)
    ended = np.flatnonzero(n_dev < limit)
    pick = ended[:m]
    if len(pick) < m:
        pick = np.concatenate([pick, np.flatnonzero(n_dev >= limit)[:m - len(pick)]])
    ag = orc.make_agent(policy=orc.EGREEDY, epsilon=0.1, seed=0, gamma=0.9, lr=0.001, max_episode_steps=1000)
    run = orc.Run(ag, len(pick), "f64")
    for j, i in enumerate(pick):
        run.weights[j] = ctx.get_weights(int(i)).astype(np.float64)
    n_cpu, _, margin = run.rollout_greedy_margin(limit)
    run.close()
    same = n_dev[pick] == n_cpu
    term = n_dev[pick] < limit
    return {"limit": limit, "all_learners": int(ctx.N), "terminated_frac_all_learners": float((n_dev < limit).mean()),
            "device_mean_n_states_all_learners": float(n_dev.mean()),
            "compared": int(len(pick)), "terminated_frac": float(term.mean()),
            "device_mean_n_states": float(n_dev[pick].mean()), "cpu_reference_mean_n_states": float(n_cpu.mean()),
            "identical_n_states_frac": float(same.mean()),
            "identical_n_states_frac_of_terminated": float(same[term].mean()) if term.any() else None,
            "min_argmax_margin": float(margin.min()), "median_min_argmax_margin": float(np.median(margin)),
            "min_argmax_margin_of_identical": float(margin[same].min()) if same.any() else None,
            "max_min_margin_of_differing": float(margin[~same].max()) if (~same).any() else None,
            "note": "CPU = f64 oracle (oracle/rsrl_oracle.c orc_run_rollout_greedy_margin) from the same weights; a learner can differ only "
                    "where an argmax margin is below fp32 resolution (max_min_margin_of_differing says how small the margin of the "
                    "differing learners was)"}


def parity_sample(m=2048):
    """Device vs the f64 CPU oracle (the reference's precision) on m random in-range states, single-step quantities of SURVEY 8(d):
    worst |dphi|, |dQ| / (1 + |Q|), |ddelta| / (1 + |delta|), |dW| after one update, and the three transitions (checker only:
    scripts/measure_parity.py; tests/test_gpu_parity_mc.py asserts the tolerances, the bitwise suites compare against the
    device-order oracle)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("measure_parity", os.path.join(ROOT, "scripts", "measure_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.measure(m)
    out["sample"] = f"{m} uniformly random in-range states per quantity, device (fp32) vs oracle f64"
    # every BASELINE.json configuration against the f64 oracle: teacher-forced learning (identical fp32-representable inputs), bf16 + SR vs fp32
    # weights vs f64, free-running population statistics -- a shortened live sample of what tests/test_gpu_parity_f64.py asserts at full
    # length (profiles/r05_parity_configs.json: the full-length worst cases)
    try:
        out["configs"] = mod.measure_configs(scale=0.25)
        out["configs"]["sample"] = ("a quarter of the asserted lengths: 250 teacher-forced steps (configs[4]: 50), 500 free-running steps; "
                                    "keys as in scripts/measure_parity.py")
    except Exception as e:      # noqa: BLE001
        out["configs"] = {"error": repr(e)}
    return out


def guarded(fn, timeout_s):
    """Run a secondary measurement in a daemon thread; {"error": "timeout"} if it does not come back in time."""
    box = {}

    def run():
        try:
            box["r"] = fn()
        except Exception as e:      # noqa: BLE001
            box["r"] = {"error": repr(e)}
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(timeout_s)
    return box.get("r", {"error": "timeout"})


def streaming_leg(rsrl_amd, envs, rank, device, steps=2000, warmup=200):
    """Secondary measurement: the SAME workload with one batch-step per launch (k_step_reg_lm / k_step_reg_q4), i.e. the 608 B/env-step
    streaming formulation the HBM roofline of SURVEY 8(d) is defined on -- every byte of it really moves.  Never part of `value`.
    At 65 536 learners the 28 MB of weights stay in L2 / the Infinity Cache between launches; the HBM-resident figure is the one at
    1 048 576 learners (453 MB of weights: beyond the 256 MiB Infinity Cache), reported beside it (roofline_streaming_hbm)."""
    try:
        ctx = rsrl_amd.Context(domain=rsrl_amd.MOUNTAIN_CAR, order=5, algo=rsrl_amd.QLEARNING, policy=rsrl_amd.EPSILON_GREEDY,
                               epsilon=0.1, gamma=0.9, lr=0.001, n_envs=envs, env_offset=rank * envs, seed=0,
                               max_episode_steps=1000, steps_per_launch=1, device=device)
        ctx.reset()
        ctx.train(warmup, want_stats=False)
        ctx.sync()
        ctx.timing_enable(True)
        t0 = time.perf_counter()
        ctx.train(steps, want_stats=False)
        ctx.sync()
        dt = time.perf_counter() - t0
        ms, n, kn = ctx.timing_read()
        ctx.close()
        avg = ms * 1e-3 / max(1, n)
        ach = BYTES_PER_ENV_STEP * envs / avg
        working_set = envs * (432 + 8 + 4 + 4 + 12)
        chk = profile_digest(kn, tables=("pmc_traffic.json",))
        return {"bound": "hbm", "kernel": kn, "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
                "frac_what": "algorithmic bytes (608 B per env-step, SURVEY 8(d)) x live HIP-event rate / 8 TB/s: no profiled constant in it",
                "profile_digest_matches": chk["profile_digest_matches"], "profile_digest": chk,
                "traffic": pmc_traffic(kn, envs, 1) if chk["profile_digest_matches"] else None, "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/pmc_traffic.json)",
                "algorithmic_bytes_per_launch": BYTES_PER_ENV_STEP * envs, "avg_launch_ms": avg * 1e3, "launches": n,
                "algorithmic_bytes_per_env_step": BYTES_PER_ENV_STEP, "learners": envs, "working_set_bytes": working_set,
                "resident_in": "HBM" if working_set > 256 * 2 ** 20 else ("Infinity Cache / L2 between launches" if working_set > 32 * 2 ** 20 else "L2 between launches"),
                "env_steps_per_s_this_rank": envs * steps / dt}
    except Exception as e:
        return {"error": repr(e)}


def trait_loop_leg(envs, device, defer=True, steps=400, warmup=50):
    """Secondary measurement: the TRAIT-GRANULAR loop a drop-in caller writes (examples/q_learning.rs:40-52) -- rsrl_hip_domain_step -> rsrl_hip_handle ->
    rsrl_hip_domain_reset -> rsrl_hip_policy_sample(NULL), one C-ABI call per trait method, device arrays, the same workload as `value` on a ctx in the
    learner-major layout (steps_per_launch = 1).  defer=True: a ctx-owned stream, where the library accepts the four calls and launches them as one kernel;
    defer=False (RSRL_NO_TRAIT_DEFER=1): one kernel per call, what a caller-supplied stream gets.  frac = the loop's algorithmic bytes (679 B per env-step:
    scripts/trait_loop.py) x env-steps/s / 8 TB/s, wall clock with one synchronize at the end (host time of the Python caller included)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("trait_loop", os.path.join(ROOT, "scripts", "trait_loop.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = os.environ.get("RSRL_NO_TRAIT_DEFER")
    try:
        if defer:
            os.environ.pop("RSRL_NO_TRAIT_DEFER", None)
        else:
            os.environ["RSRL_NO_TRAIT_DEFER"] = "1"
        out = mod.measure(envs, steps, warmup, device=device, steps_per_launch=1)
    except Exception as e:      # noqa: BLE001
        return {"error": repr(e)}
    finally:
        if old is None:
            os.environ.pop("RSRL_NO_TRAIT_DEFER", None)
        else:
            os.environ["RSRL_NO_TRAIT_DEFER"] = old
    out.update({"bound": "hbm", "peak": HBM_PEAK / 1e9, "unit_roofline": "GB/s", "achieved": out["algorithmic_bytes_per_env_step"] * out["value"] / 1e9,
                "frac": out["algorithmic_bytes_per_env_step"] * out["value"] / HBM_PEAK,
                "launches_per_batch_step": 1 if defer else 4,
                "what": "transition -> handle -> new episodes -> sample through the C ABI, one call per trait method, device arrays; "
                        + ("the four calls accepted and launched as ONE kernel (ctx-owned stream)" if defer else "one kernel per call (what a caller-supplied stream gets)")})
    return out


def hbm_copy_measured(device, gib=1.0, reps=10):
    """what this box's memory system delivers: a float4 device copy (rsrl_hip_measure_copy, HIP events), GB/s counting read + write -- SURVEY 8(d) asks for the
    HBM fractions against it as well as against the published 8 TB/s (MI355X_MICROARCH.md quotes 6.29 TB/s for the same kind of copy)"""
    import ctypes as C
    from rsrl_amd import _abi
    out = C.c_double()
    _abi.check(_abi.lib().rsrl_hip_measure_copy(int(device), int(gib * 2 ** 30), int(reps), C.byref(out)))
    return {"GBps": out.value, "bytes": int(gib * 2 ** 30), "reps": reps, "what": "float4 device-to-device copy kernel, read + write, HIP events (rsrl_amd/csrc/measure.hip)"}


def shared_w_leg(cp, rsrl_amd, make_sharded_context, exchange, envs_per_gpu=131072, steps=320, warmup=64):
    """Secondary measurement (BASELINE.json configs[3]): 131 072 MountainCar envs per GPU, ONE shared Fourier(5)
    approximator, per-batch-step exchange of the weight delta (exchange AUTO: the one-hop peer exchange whenever every device
    reaches every other's memory; RCCL: the any-topology fallback).  The exchange is attached for a single rank too (a group of
    size 1 runs the same sequence).  Never part of `value`.  The ranks FAIL TOGETHER: after every phase they tell each other
    whether it worked, so a rank that hit an error never leaves the others waiting in a collective."""
    import numpy as np

    def together(err, what):
        oks = cp.all_gather_bytes(err is None)
        if all(oks):
            return None
        return {"error": (repr(err) if err is not None else f"rank(s) {[r for r, ok in enumerate(oks) if not ok]} failed while {what}"), "failed_while": what}
    try:
        ctx = make_sharded_context(envs_per_gpu * cp.world, cp, force_exchange=True, domain=rsrl_amd.MOUNTAIN_CAR, order=5,
                                   algo=rsrl_amd.QLEARNING, policy=rsrl_amd.EPSILON_GREEDY, epsilon=0.1, gamma=0.9,
                                   lr=0.001 / (envs_per_gpu * cp.world), weight_mode=rsrl_amd.W_SHARED, seed=0,
                                   max_episode_steps=1000, exchange=exchange)
    except Exception as e:                       # (make_sharded_context fails on every rank together)
        return {"error": repr(e), "failed_while": "setting the exchange up"}
    err = None
    try:
        comm_world, comm_rank, comm_kind = ctx.comm_info()      # what the attached exchange itself reports (ncclCommCount for RCCL)
        ctx.reset()
        ctx.train(warmup, want_stats=False)
        ctx.sync()
    except Exception as e:      # noqa: BLE001
        err = e
    bad = together(err, "warming up")
    if bad:
        ctx.close()
        return bad
    cp.barrier()
    dt_own, ms, n_l, kn, w = 0.0, 0.0, 1, "", None
    try:
        ctx.timing_enable(True)
        t0 = time.perf_counter()
        ctx.train(steps, want_stats=False)
        ctx.sync()
        dt_own = time.perf_counter() - t0
        ms, n_l, kn = ctx.timing_read()
        w = ctx.get_weights()
    except Exception as e:      # noqa: BLE001
        err = e
    bad = together(err, "timing the batch-steps")
    ctx.close()
    if bad:
        return bad
    dt = cp.max_over_ranks(dt_own)
    chk = float(np.abs(w).sum())
    lo, hi = -cp.max_over_ranks(-chk), cp.max_over_ranks(chk)
    return {"workload": f"{envs_per_gpu} MountainCar envs per GPU, shared-W QLearning Fourier(5), per-step "
                        f"{ {0: 'RCCL all-reduce of the fixed-point delta table', 1: 'one-hop peer-write exchange of the delta'}.get(comm_kind, 'exchange') }", "ranks": cp.world, "steps": steps,
            "value": envs_per_gpu * cp.world * steps / dt, "unit": "env-steps/s", "us_per_batch_step": dt / steps * 1e6,
            "exchange_world_size": comm_world, "exchange_kind": {0: "rccl", 1: "peer"}.get(comm_kind, "none"),
            "per_rank_env_steps_per_s": [envs_per_gpu * steps / max(1e-12, float(x)) for x in cp.all_gather_bytes(dt_own)],
            "per_rank_kernel_us_per_batch_step": [float(x) for x in cp.all_gather_bytes(ms * 1e3 / max(1, steps))],
            "kernel": kn, "kernel_us_per_batch_step": ms * 1e3 / max(1, n_l),
            "roofline": leg_roofline(kn, envs_per_gpu * steps / max(1e-12, ms * 1e-3), envs_per_gpu,
                                     ms * 1e-3 if kn == "k_shared_persist" else ms * 1e-3 / max(1, n_l), 1 if kn == "k_shared_persist" else n_l, 32,
                                     "SURVEY 8(d): C4 is VALU / latency bound (W stays on the chip: 32 B/env-step of state stream); the batch-step = two fabric hops "
                                     "of the delta all-reduce (~3.2 us, profiles/r03_ubench_granule_allreduce.txt) + the learners' arithmetic at two waves per SIMD"),
            "replicas_consistent": bool(lo == hi), "sum_abs_w": hi}


def leg_roofline(kname, per_gpu_steps_per_s, envs, avg_launch_s, launches, alg_bytes_per_env_step, what, also=()):
    """roofline object of a secondary leg: every `frac` is a fraction of a PUBLISHED peak and follows from a file under profiles/.
    bound "valu": flop per env-step (profiles/isa_mix.json: rocprofv3 instruction-class counters of this kernel) x the kernel's env-steps/s by HIP
    events / 157.3 TFLOP/s; `issue_slots`: VALU instructions per env-step at the guide's 2 cycles each (a LOWER bound of the slots used: packed
    instructions take 4) against 1024 SIMDs x 2.4 GHz; `traffic` = HBM bytes per launch from the PMC passes (profiles/pmc_traffic.json), `hbm` = that
    traffic over the launch duration against 8 TB/s.  SURVEY 8(d)'s algorithmic bytes x rate are kept under a name that is not `frac`."""
    mixes = _profiles_json("isa_mix.json") or {}
    mix = mixes.get(kname) or {}
    flop = mix.get("flop_per_env_step")
    n_valu = mix.get("valu_wave_instr_per_env_step")
    for k in also:                                       # the batch-step's other kernels (`also`): their instructions and bytes count too
        if flop and mixes.get(k, {}).get("flop_per_env_step") is not None:
            flop += mixes[k]["flop_per_env_step"]
            n_valu += mixes[k].get("valu_wave_instr_per_env_step", 0.0)
    rl = {"bound": "valu", "kernel": kname, "unit": "TFLOP/s", "peak": FP32_VECTOR_PEAK / 1e12, "env_steps_per_s_kernel": per_gpu_steps_per_s,
          "avg_launch_ms": avg_launch_s * 1e3, "launches": launches, "what": what}
    if flop:
        rl.update({"achieved": flop * per_gpu_steps_per_s / 1e12, "frac": flop * per_gpu_steps_per_s / FP32_VECTOR_PEAK, "flop_per_env_step": flop,
                   "source": "profiles/isa_mix.json (rocprofv3 SQ_INSTS_VALU_* class counters of this kernel per env-step) x HIP-event kernel rate of this run"})
    else:
        rl.update({"achieved": None, "frac": None, "error": f"profiles/isa_mix.json has no flop count for {kname}"})
    if n_valu:
        cyc = n_valu * SPEC_CYCLES["other"]                          # SIMD cycles per env-step if every wave-instruction took 2
        peak = N_SIMD * CLOCK_HZ / cyc
        rl["issue_slots"] = {"valu_wave_instr_per_env_step": n_valu, "frac_lower_bound": per_gpu_steps_per_s / peak, "peak_env_steps_per_s": peak,
                             "what": "every VALU wave-instruction priced at 2 cycles (MI355X_MICROARCH.md), 1024 SIMDs x 2.4 GHz; packed fp32 instructions take 4, "
                                     "so the slots really used are more"}
    tfile = _profiles_json("pmc_traffic.json") or {}
    rec = tfile.get(kname)
    rec = rec[0] if isinstance(rec, list) and rec else rec
    if isinstance(rec, dict) and also:
        rec = dict(rec)
        for k in also:
            o = tfile.get(k)
            if isinstance(o, dict) and o.get("envs") == rec.get("envs"):
                rec["bytes_per_env_step"] += o["bytes_per_env_step"]
                rec["traffic_bytes_per_launch"] += o["traffic_bytes_per_launch"]
    rl["traffic"], rl["traffic_unit"] = None, "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes; profiles/pmc_traffic.json)"
    if isinstance(rec, dict) and rec.get("envs") == envs and avg_launch_s > 0:
        if rec.get("scales") == "per_launch":            # W in + out once per launch, whatever its depth: the profiled launch's bytes are this launch's
            tr = rec["traffic_bytes_per_launch"]
            bps = tr / avg_launch_s
        else:                                            # bytes follow the batch-steps: scale the profiled launch to this one's depth
            steps_here = per_gpu_steps_per_s * avg_launch_s / envs
            tr = rec["bytes_per_env_step"] * envs * steps_here
            bps = rec["bytes_per_env_step"] * per_gpu_steps_per_s
        rl["traffic"] = tr
        rl["traffic_profiled"] = {"bytes_per_launch": rec["traffic_bytes_per_launch"], "steps_per_launch": rec.get("steps_per_launch"), "scales": rec.get("scales")}
        rl["hbm"] = {"achieved": bps / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": bps / HBM_PEAK, "what": "REAL traffic (PMC) over the kernel's time"}
    rl["algorithmic_bytes_equivalent"] = {"bytes_per_env_step": alg_bytes_per_env_step, "equivalent_GBps": alg_bytes_per_env_step * per_gpu_steps_per_s / 1e9,
                                          "note": "SURVEY 8(d)'s algorithmic bytes x rate: NOT moved bytes and not a roofline fraction"}
    return gate_on_profile(rl, [kname] + list(also))


def valu_roofline(kname, per_gpu_steps_per_s, steps_per_wave_cycles=None):
    """The fused kernel keeps W in the register file (HBM idle, profiles/r0*_pmc*): its bound is the fp32 VECTOR unit.
    frac = flop per env-step x env-steps/s / 157.3 TFLOP/s (MI355X_MICROARCH.md, peak FP32 vector = the f32 matrix peak); the
    flop count is the kernel's own executed instruction mix (profiles/isa_mix.json: rocprofv3 SQ_INSTS_VALU_* class counters
    per env-step -- FMA = 2 flop per lane, packed = two lanes' worth, multiplies by zero of the masked column update included:
    they are issued).  Secondary: `issue_slots` = the mix priced at the guide's issue rates (2 cycles per VALU instruction on a
    SIMD-32, 4 per packed fp32 instruction) against 1024 SIMDs x 2.4 GHz; `ubench_ceiling` = round 2's ceiling from this
    machine's measured saturated issue costs (scripts/ubench/valu_issue.hip), which is NOT a published peak."""
    mix = (_profiles_json("isa_mix.json") or {}).get(kname)
    if not mix:
        return None
    pk_fma = mix.get("pk_fma", mix["pk"])
    flop = pk_fma * 4 + (mix["pk"] - pk_fma) * 2 + mix.get("fp_fma", 0) * 2 + mix.get("fp_other", 0)
    n_valu = mix["pk"] + mix["mad_u64"] + mix["cndmask"] + mix["other"]
    spec_cyc = mix["pk"] * SPEC_CYCLES["pk"] + (n_valu - mix["pk"]) * SPEC_CYCLES["other"]
    spec_peak = N_SIMD * 64 * CLOCK_HZ / spec_cyc
    ub = {"pk": 4.47, "mad_u64": 4.96, "cndmask": 4.13, "other": 2.46}
    ub_cyc = sum(mix[k] * ub[k] for k in ub)
    ub_peak = N_SIMD * 64 * CLOCK_HZ / ub_cyc
    # one wave per SIMD (65 536 learners / 1 024 SIMDs): a lone wave issues one VALU instruction per 4.1-4.5 cycles whatever its class and pays ~120
    # cycles per taken branch (scripts/ubench/valu_pair.hip: 1 024- against 64-instruction loop bodies); the fused loop takes one branch per two steps
    lone_cyc = n_valu * 4.2 + 60.0
    lone_peak = N_SIMD * 64 * CLOCK_HZ / lone_cyc
    tf = flop * per_gpu_steps_per_s / 1e12
    useful = mix.get("useful_flop_per_env_step", flop - 4.0 * mix.get("pk_fma_masked_zero", 36.0))
    return {"bound": "valu", "achieved": tf, "peak": FP32_VECTOR_PEAK / 1e12, "unit": "TFLOP/s", "frac": flop * per_gpu_steps_per_s / FP32_VECTOR_PEAK,
            "useful_frac": useful * per_gpu_steps_per_s / FP32_VECTOR_PEAK, "useful_flop_per_env_step": useful,
            "useful_what": "frac counts EXECUTED flop; useful_frac leaves out the masked column update's multiplies by zero (36 packed fmas per env-step: the action "
                           "is per lane, registers cannot be indexed by a lane value -- DESIGN 4.1)",
            "flop_per_env_step": flop, "env_steps_per_s_per_gpu": per_gpu_steps_per_s,
            "issue_slots": {"valu_instr_per_env_step": n_valu, "packed": mix["pk"], "cycles_per_env_step_at_spec_rates": spec_cyc,
                            "peak_env_steps_per_s": spec_peak, "frac": per_gpu_steps_per_s / spec_peak,
                            "what": "2 cycles per VALU instruction, 4 per packed fp32 instruction (MI355X_MICROARCH.md), 1024 SIMDs x 2.4 GHz"},
            "ubench_ceiling": {"peak_env_steps_per_s": ub_peak, "frac": per_gpu_steps_per_s / ub_peak,
                               "what": "NOT a published peak: the same mix at this machine's measured saturated issue costs (8 waves/SIMD: packed 4.47, "
                                       "mad_u64 4.96, cndmask 4.13, other 2.46 cycles; profiles/r01_ubench_valu_issue.txt)"},
            "lone_wave_ceiling": {"peak_env_steps_per_s": lone_peak, "frac": per_gpu_steps_per_s / lone_peak, "cycles_per_env_step": lone_cyc,
                                  "what": "NOT a published peak: ONE wave per SIMD is all 65 536 learners give 1 024 SIMDs, and a lone wave issues one VALU "
                                          "instruction per 4.1-4.5 cycles (4.2 used) whatever its class + ~120 cycles per taken branch, one per two steps "
                                          "(profiles/r05_ubench_valu_pair.txt): at this size the kernel's time is its instruction COUNT"},
            "source": "profiles/isa_mix.json (rocprofv3 SQ_INSTS_VALU_* class counters per env-step) x HIP-event kernel rate of this run"}


def config_leg(rsrl_amd, name, kw, steps, warmup, bytes_per_env_step, what, extra=None, also=()):
    """Secondary measurement of another BASELINE.json configuration's per-GPU share (a parity-test configuration, never part of
    `value`): env-steps/s, the dominant kernel's HIP-event time per batch-step, and a roofline object on SURVEY 8(d)'s
    algorithmic bytes per env-step against the 8 TB/s of MI355X_MICROARCH.md."""
    try:
        ctx = rsrl_amd.Context(**kw)
        ctx.reset()
        ctx.train(warmup, want_stats=False)
        ctx.sync()
        ctx.timing_enable(True)
        t0 = time.perf_counter()
        ctx.train(steps, want_stats=False)
        ctx.sync()
        dt = time.perf_counter() - t0
        ms, n, kn = ctx.timing_read()
        ctx.close()
        per_step = ms * 1e-3 / max(1, steps)              # seconds of kernel time per batch-step (all launches of the call / its steps)
        rate = kw["n_envs"] / per_step if per_step > 0 else 0.0
        rec = {"workload": name, "value": kw["n_envs"] * steps / dt, "unit": "env-steps/s", "us_per_batch_step": dt / steps * 1e6,
               "kernel_us_per_batch_step": per_step * 1e6,
               "roofline": leg_roofline(kn, rate, kw["n_envs"], ms * 1e-3 / max(1, n), n, bytes_per_env_step, what, also)}
        if also:
            rec["roofline"]["kernels"] = [kn] + list(also)
        if extra:
            rec["roofline"].update(extra)
        return rec
    except Exception as e:
        return {"workload": name, "error": repr(e)}


def hbm_leg(rsrl_amd, name, kw, steps, warmup, bytes_per_env_step, what, bytes_fn=None):
    """A leg whose kernels are memory sweeps: env-steps/s, HIP-event kernel time per batch-step, and SURVEY 8(d)-style algorithmic bytes per env-step x the kernel
    rate against 8 TB/s (no instruction-mix constants needed: nothing here depends on a committed profile)."""
    try:
        ctx = rsrl_amd.Context(**kw)
        ctx.reset()
        ctx.train(warmup, want_stats=False)
        ctx.sync()
        ctx.timing_enable(True)
        t0 = time.perf_counter()
        ctx.train(steps, want_stats=False)
        ctx.sync()
        dt = time.perf_counter() - t0
        ms, n, kn = ctx.timing_read()
        extra = {}
        if bytes_fn is not None:                  # bytes that depend on what the run left behind (the sparse traces' live entries)
            bytes_per_env_step, extra = bytes_fn(ctx)
        ctx.close()
        per_step = ms * 1e-3 / max(1, steps)
        ach = bytes_per_env_step * kw["n_envs"] / per_step if per_step > 0 else 0.0
        return {"workload": name, "value": kw["n_envs"] * steps / dt, "unit": "env-steps/s", "us_per_batch_step": dt / steps * 1e6,
                "kernel_us_per_batch_step": per_step * 1e6,
                "roofline": {"bound": "hbm", "kernel": kn, "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
                             "algorithmic_bytes_per_env_step": bytes_per_env_step, "what": what, **extra}}
    except Exception as e:
        return {"workload": name, "error": repr(e)}


def sparse_trace_bytes(ctx):
    """HBM bytes per learner-step of the sparse-trace lambda agents on what the lists really hold: every LIVE entry's 16-bit key and value read, its value written
    back (10 B: a key is written only where the step appended / evicted, one per tiling), + the 2 x T gathers, the T new keys, the T lengths both ways.  The per-entry
    8-byte term goes into LDS, not HBM.  (Earlier figures of this leg: 12 416 B = FULL 512-entry lists x 24 B, the one round 5's 0.10 was quoted on; then
    24 B x live entries.  Both counted the LDS term and a 32-bit key read AND written back.)"""
    import numpy as np
    n = ctx.N
    live = float(np.mean([int((ctx.get_traces(i) != 0).sum()) for i in range(0, n, max(1, n // 48))]))
    return 10 * live + 16 * 2 * 4 + 8 * (4 + 2 + 8), {"mean_live_entries": live, "bytes_by_the_previous_accounting": 3 * live * 8 + 16 * 2 * 4}


def _num(x, digits=5):
    """a number at `digits` significant digits (the compact line is for a parser and a reader, not for reproducing bits)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return x
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if x == x and abs(x) != float("inf") else None
    return x


def _leg_summary(rec, copy_gbps=None):
    """{value, us, bound, frac, ...}: the triple VERDICT r5 asks for per leg, from a leg's full record (whatever its shape)"""
    if not isinstance(rec, dict):
        return None
    if "error" in rec or "skipped" in rec:
        return {"error": str(rec.get("error") or rec.get("skipped"))[:160]}
    rl = rec.get("roofline") if isinstance(rec.get("roofline"), dict) else rec
    out = {"value": _num(rec.get("value", rec.get("env_steps_per_s_this_rank"))),
           "us": _num(rec.get("us_per_batch_step", (rec.get("avg_launch_ms") or 0) * 1e3 or None)),
           "kernel": rl.get("kernel") or rec.get("kernel"), "bound": rl.get("bound"), "frac": _num(rl.get("frac"))}
    if isinstance(rl.get("hbm"), dict):
        out["hbm_frac"] = _num(rl["hbm"].get("frac"))
    if rl.get("bound") == "hbm" and copy_gbps and rl.get("achieved"):
        out["frac_of_measured_copy"] = _num(rl["achieved"] / copy_gbps)
    if "us_per_batch_step" not in rec and "calls" in rec and "avg_launch_ms" in rec:      # value_no_coalesce: a launch is a whole K-step call
        out["us_per_launch"] = out.pop("us")
    if "kernel_us_per_batch_step" in rec:
        out["kernel_us"] = _num(rec["kernel_us_per_batch_step"])
    if "floor_us" in rec:
        out["floor_us"] = _num(rec["floor_us"]); out["floor_frac"] = _num(rec.get("floor_frac"))
    if rl.get("profile_digest_matches") is False:
        out["profile_digest_matches"] = False
    return {k: v for k, v in out.items() if v is not None}


# C3's floor in THIS formulation (one more formulation or a written floor; DESIGN 4.4 has the formulations that were built and lost):
# a batch-step is three DEPENDENT launches -- W_{t+1} needs every learner's term, every learner's action needs W_{t+1}.
C3_FLOOR = {"floor_us": 13.6,
            "floor": "k_shared_ca at the issue rate of its 4 waves per SIMD: 893 VALU per learner-step (PMC) x 4 waves x 2.55 cycles (the fast class at >= 2 waves per "
                     "SIMD, profiles/r05_ubench_valu_pair.txt) / 2.4 GHz = 3.8 us, + its two dependent gather round trips through L2 (2 x ~0.9 us) = 5.6 us; "
                     "k_tile_scatter and k_apply_rep are each one dependent launch of the captured graph, ~4.0 us whatever they do (k_apply_rep moves 2 MB in 4.4 us; "
                     "an 11.4 us grid barrier is the alternative, profiles/r05_ubench_grid_barrier.txt): 5.6 + 2 x 4.0"}
COMPACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline")
LEG_KEYS = ("trait_loop", "trait_loop_unfused", "trait_loop_1m", "value_no_coalesce", "roofline_streaming", "roofline_streaming_hbm", "c3_shared_tiles",
            "c5_wave_bf16", "lambda_shared_tiles", "lambda_generic_order", "shared_w", "shared_w_rccl")
COMPACT_LIMIT = 6000      # bytes; the driver's parser lost the 24 KB line of round 5 (BENCH_r05.json: parsed = null) -- tests/test_bench_line_cpu.py


def compact_line(full, detail_path="bench_detail.json"):
    """The ONE line rank 0 prints last on stdout: the driver's contract keys, the roofline and CPU-baseline objects, one {value, us, frac} triple per
    secondary leg -- and nothing else.  Everything (per-rank records, per-configuration parity, ceilings, digests, `what` strings) goes to the detail file
    and to stderr."""
    line = {k: full.get(k) for k in COMPACT_KEYS[:12]}
    for k in ("value", "ms_per_step"):
        line[k] = _num(line[k], 7)
    cfg = full.get("config") or {}
    line["config"] = {k: (_num(v) if not isinstance(v, str) else v) for k, v in cfg.items() if k in ("workload", "envs_per_gpu", "steps_per_launch", "repeats", "ranks", "parallelism")}
    rl = full.get("roofline") or {}
    crl = {k: _num(rl.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "useful_frac", "avg_launch_ms", "launches", "traffic", "profile_digest_matches")
           if k in rl}
    if isinstance(rl.get("hbm"), dict):
        crl["hbm_frac"] = _num(rl["hbm"].get("frac"))
    if isinstance(rl.get("issue_slots"), dict):
        crl["issue_slots_frac"] = _num(rl["issue_slots"].get("frac"))
    if isinstance(rl.get("lone_wave_ceiling"), dict):
        crl["lone_wave_frac"] = _num(rl["lone_wave_ceiling"].get("frac"))
    line["roofline"] = crl
    copy = (full.get("hbm_copy_measured") or {}).get("GBps")
    if copy:
        line["hbm_copy_measured_GBps"] = _num(copy)
        if isinstance(crl.get("hbm_frac"), float) and rl.get("hbm", {}).get("achieved"):
            crl["hbm_frac_of_measured_copy"] = _num(rl["hbm"]["achieved"] / copy)
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "per_core": _num(cb.get("per_core")), "optimised_value": _num((cb.get("optimised") or {}).get("value")),
                                "sample": (cb.get("sample") or "")[:200]}
    legs = {}
    for k in LEG_KEYS:
        sm = _leg_summary(full.get(k), copy)
        if sm:
            legs[{"roofline_streaming": "streaming", "roofline_streaming_hbm": "streaming_1m"}.get(k, k)] = sm
    line["legs"] = legs
    gr = full.get("greedy_rollout")
    if isinstance(gr, dict):
        line["greedy_rollout"] = {k: _num(gr.get(k)) for k in ("limit", "compared", "terminated_frac", "identical_n_states_frac", "min_argmax_margin",
                                                               "max_min_margin_of_differing", "error") if k in gr}
    par = full.get("parity")
    if isinstance(par, dict):
        line["parity"] = {k: _num(par.get(k), 3) for k in ("phi_max_abs", "q_max_rel", "delta_max_rel", "w_update_max_abs_rel", "error") if k in par}
    line["spread"] = _num(full.get("spread"), 3)
    if full.get("oversubscribed"):
        line["oversubscribed"] = full["oversubscribed"][:100]
    line["detail"] = detail_path
    return line


def _flush_c_stdio():
    """RCCL prints its version banner through C stdio, which is block-buffered when stdout is a file or a pipe: unflushed, the banner lands at process exit,
    AFTER the result line, and the line is no longer the last one of stdout"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def emit(full, detail_path=None):
    """detail -> bench_detail.json (--detail PATH, else next to bench.py and under gpurun_out/ when that exists) and stderr; the compact line -> stdout, LAST"""
    detail = json.dumps(full)
    wrote = None
    targets = [detail_path] if detail_path else [os.path.join(d, "bench_detail.json") for d in (ROOT, os.path.join(ROOT, "gpurun_out")) if os.path.isdir(d)]
    for path in targets:
        try:
            with open(path, "w") as f:
                f.write(detail + "\n")
            wrote = wrote or (path if detail_path else os.path.relpath(path, ROOT))
        except OSError:
            pass
    print(detail, file=sys.stderr, flush=True)
    text = json.dumps(compact_line(full, wrote or "stderr"), separators=(",", ":"))
    if len(text) > COMPACT_LIMIT:           # never again a line the driver cannot parse: drop the optional objects, largest first
        slim = compact_line(full, wrote or "stderr")
        for k in ("parity", "greedy_rollout", "legs"):
            slim.pop(k, None)
            text = json.dumps(slim, separators=(",", ":"))
            if len(text) <= COMPACT_LIMIT:
                break
    sys.stdout.flush()
    _flush_c_stdio()
    print(text, flush=True)


def spawn_ranks(n):
    """--gpus N without a launcher: start the N ranks ourselves, one per GPU, and pass rank 0's JSON line through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2560)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--repeats", type=int, default=0, help="back-to-back repetitions of the K-step call inside one timed region (0 = auto: >= --region-seconds)")
    ap.add_argument("--regions", type=int, default=N_REGIONS, help="how many times the timed region is measured (value = the median)")
    ap.add_argument("--region-seconds", type=float, default=TARGET_REGION_S)
    ap.add_argument("--no-nocoalesce-leg", action="store_true", help="skip the one-launch-per-call measurement")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the secondary C3 / C5 measurements")
    ap.add_argument("--steps-per-launch", type=int, default=0, help="fuse depth (0 = library default)")
    ap.add_argument("--envs", type=int, default=N_ENVS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shared-leg", action="store_true", help="skip the secondary shared-W (exchange) measurements")
    ap.add_argument("--no-streaming-leg", action="store_true", help="skip the secondary 1-step-per-launch measurement")
    ap.add_argument("--no-trait-leg", action="store_true", help="skip the secondary trait-granular-loop measurements")
    ap.add_argument("--detail", default=None, help="where the full result goes (default: bench_detail.json next to bench.py and under gpurun_out/)")
    ap.add_argument("--allow-oversubscribe", action="store_true", help="let several ranks share a device (test boxes only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    from rsrl_amd.distributed import ControlPlane, make_sharded_context
    cp = ControlPlane()          # gloo control plane (rendezvous / barrier / max over ranks); no-op for one rank
    rank, local_rank, world = cp.rank, cp.info.local_rank, cp.world
    import rsrl_amd
    ndev = rsrl_amd.device_count()
    if world > ndev and not args.allow_oversubscribe:
        if rank == 0:
            print(f"bench.py: {world} ranks but {ndev} visible GPU(s); one rank per GPU is the contract "
                  "(--allow-oversubscribe shares devices on a test box)", file=sys.stderr)
        sys.exit(2)
    device = local_rank % max(1, ndev)

    # one-off costs that are not steps (loading the kernels' code object on first use) are paid by a throw-away 64-env ctx
    # of the same configuration, so that --warmup 0 still times steps and nothing else
    with rsrl_amd.Context(domain=rsrl_amd.MOUNTAIN_CAR, basis=rsrl_amd.FOURIER, order=5, algo=rsrl_amd.QLEARNING,
                          policy=rsrl_amd.EPSILON_GREEDY, epsilon=0.1, gamma=0.9, lr=0.001, n_envs=64, seed=0,
                          max_episode_steps=1000, steps_per_launch=args.steps_per_launch, device=device) as prime:
        prime.reset()
        prime.train(2, want_stats=False)
        prime.sync()
    ctx = rsrl_amd.Context(domain=rsrl_amd.MOUNTAIN_CAR, basis=rsrl_amd.FOURIER, order=5,
                           algo=rsrl_amd.QLEARNING, policy=rsrl_amd.EPSILON_GREEDY, epsilon=0.1,
                           gamma=0.9, lr=0.001, n_envs=args.envs, env_offset=rank * args.envs, seed=0,
                           max_episode_steps=1000, steps_per_launch=args.steps_per_launch, device=device)
    ctx.reset()
    if args.warmup > 0:
        ctx.train(args.warmup, want_stats=False)
    ctx.sync()
    # R: how many K-step calls make a >= 1 s region (one untimed calibration burst; the same R on every rank)
    repeats = args.repeats
    if repeats <= 0:
        burst = 64                                   # back-to-back, as in the timed region (the library coalesces short calls)
        t0 = time.perf_counter()
        for _ in range(burst):
            ctx.train(args.steps, want_stats=False)
        ctx.sync()
        t_call = cp.max_over_ranks(time.perf_counter() - t0) / burst
        repeats = int(min(4000000, max(3, -(-1.15 * args.region_seconds // max(t_call, 1e-7)))))    # (the burst is a little slower per call than the steady state)

    def timed_region(n_calls):
        """n_calls back-to-back K-step calls between one barrier + synchronize pair -> (max over ranks, this rank's own) seconds"""
        cp.barrier()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(n_calls):
            ctx.train(args.steps, want_stats=False)
        ctx.sync()
        own = time.perf_counter() - t0
        worst = cp.max_over_ranks(own)
        cp.barrier()
        return worst, own

    ctx.timing_enable(True)
    regions = [timed_region(repeats) for _ in range(args.regions)]
    kernel_ms, launches, kname = ctx.timing_read()
    ctx.timing_enable(False)
    order = sorted(range(len(regions)), key=lambda j: regions[j][0])
    dt = regions[order[len(order) // 2]][0]                    # the MEDIAN region
    own_rates = cp.all_gather_bytes(args.envs * args.steps * repeats / regions[order[len(order) // 2]][1])
    # per rank, so that the first real multi-GPU run localises a slow rank: its own median region, its kernel time per batch-step by HIP events
    # on its own stream, its device and which physical device that is
    try:
        dev_id = "%x" % rsrl_amd.device_identity(device)
    except Exception:      # noqa: BLE001
        dev_id = None
    per_rank = cp.all_gather_bytes({"rank": rank, "local_rank": local_rank, "device": device, "device_identity": dev_id,
                                    "kernel_us_per_batch_step": kernel_ms * 1e3 / max(1, args.steps * repeats * len(regions)),
                                    "kernel_launches": launches, "regions_s": [r[1] for r in regions],
                                    "env_steps_per_s": args.envs * args.steps * repeats / regions[order[len(order) // 2]][1]})
    # the same K-step driver call WITHOUT launch coalescing (one launch per call): what a single train(K) costs
    no_coalesce = None
    if not args.no_nocoalesce_leg and args.steps_per_launch != 1:
        os.environ["RSRL_NO_COALESCE"] = "1"
        try:
            t0 = time.perf_counter()
            for _ in range(64):
                ctx.train(args.steps, want_stats=False)
            ctx.sync()
            t_call = cp.max_over_ranks(time.perf_counter() - t0) / 64
            n_nc = int(min(400000, max(3, -(-0.3 // max(t_call, 1e-7)))))
            ctx.timing_enable(True)
            dt_nc = timed_region(n_nc)[0]
            ms_nc, l_nc, _ = ctx.timing_read()
            ctx.timing_enable(False)
            no_coalesce = {"value": args.envs * world * args.steps * n_nc / dt_nc, "unit": "env-steps/s", "calls": n_nc, "region_s": dt_nc,
                           "launches": l_nc, "avg_launch_ms": ms_nc / max(1, l_nc),
                           "what": f"the same {args.steps}-step driver call, one launch per call (RSRL_NO_COALESCE=1): every call loads and stores every "
                                   "learner's weights around its steps; `value` is the coalesced figure"}
        finally:
            os.environ.pop("RSRL_NO_COALESCE", None)
    rollout = guarded(lambda: greedy_rollout_check(ctx), 120) if rank == 0 else None
    parity = guarded(parity_sample, 180) if (rank == 0 and not args.no_cpu_baseline) else None
    # secondary legs run under a watchdog: whatever happens to them, rank 0 still prints the headline line
    streaming = guarded(lambda: streaming_leg(rsrl_amd, args.envs, rank, device), 120) \
        if (args.steps_per_launch != 1 and not args.no_streaming_leg) else None
    # ... and at an HBM-resident size: 1 048 576 learners = 453 MB of weights (the 28 MB of 65 536 learners never leave L2 / the Infinity Cache)
    streaming_hbm = guarded(lambda: streaming_leg(rsrl_amd, 1048576, rank, device, steps=160, warmup=32), 120) \
        if (args.steps_per_launch != 1 and not args.no_streaming_leg and world <= ndev) else None
    # the trait-granular loop (one C-ABI call per trait method): fused by the library on its own stream, one kernel per call, and at an HBM-resident size
    trait = trait_unfused = trait_1m = copy_bw = None
    if not args.no_trait_leg and world <= ndev:
        trait = guarded(lambda: trait_loop_leg(args.envs, device, defer=True), 120)
        trait_unfused = guarded(lambda: trait_loop_leg(args.envs, device, defer=False), 120)
        trait_1m = guarded(lambda: trait_loop_leg(1048576, device, defer=True, steps=100, warmup=20), 120)
    if rank == 0:
        copy_bw = guarded(lambda: hbm_copy_measured(device), 60)
    c3 = c5 = None
    if not args.no_config_legs and world <= ndev:
        c3 = guarded(lambda: config_leg(
            rsrl_amd, "BASELINE.json configs[2]: 262144 CartPole envs, SARSA + tile coding (8 tilings x 8^4), eps-greedy, ONE shared table, 1 GPU",
            dict(domain=rsrl_amd.CART_POLE, basis=rsrl_amd.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=rsrl_amd.SARSA, n_envs=262144,
                 policy=rsrl_amd.EPSILON_GREEDY, epsilon=0.1, gamma=0.99, lr=0.0125 / 262144, weight_mode=rsrl_amd.W_SHARED, max_episode_steps=1000,
                 env_offset=rank * 262144, device=device), 960, 64, 208,
            "SURVEY 8(d): 208 B/env-step, of which 48 B are the HBM stream (state, action, counter) and 160 B are gathers / atomic "
            "read-modify-writes of the shared table served by L2; the batch-step is three DEPENDENT launches (step, scatter, apply): latency-bound, "
            "none of the three fractions binds", also=("k_tile_scatter", "k_apply_rep")), 120)
        if isinstance(c3, dict) and "error" not in c3:
            c3.update(C3_FLOOR)
            if c3.get("us_per_batch_step"):
                c3["floor_frac"] = C3_FLOOR["floor_us"] / c3["us_per_batch_step"]
        c5 = guarded(lambda: config_leg(
            rsrl_amd, "BASELINE.json configs[4], one GPU's share: 32768 Acrobot envs, ExpectedSARSA + Fourier(7) + Softmax, bf16 weights, per-env W",
            dict(domain=rsrl_amd.ACROBOT, order=7, algo=rsrl_amd.EXPECTED_SARSA, policy=rsrl_amd.SOFTMAX, tau=1.0, gamma=0.99, lr=0.001, alpha=1.0,
                 n_envs=32768, weight_dtype=rsrl_amd.W_BF16, max_episode_steps=1000, env_offset=rank * 32768, device=device), 512, 64, 32816,
            "SURVEY 8(d): 32 816 B/env-step if W (24 KiB bf16 per learner) were streamed every step; k_train_wave_pk keeps W PACKED in registers (two "
            "bf16 per VGPR, 96 of them) for the whole launch, so the figure is an equivalent, not moved bytes: two waves per SIMD, 4.4 cycles per "
            "VALU instruction against the measured 2.5 (fast class) / 4.9 (slow class: packed, shifts, bfe, perm, readlane ...) of "
            "profiles/r05_ubench_valu_pair.txt: VALU-issue bound (DESIGN 4.6)"), 180)
    # the eligibility-trace kernels rebuilt in round 6, at the BASELINE configurations' learner count
    lam_shared = lam_generic = None
    if not args.no_config_legs and world <= ndev:
        lam_shared = guarded(lambda: hbm_leg(
            rsrl_amd, "SARSALambda over ONE shared tile table (8 x 8^4), sparse per-learner traces, 65536 CartPole envs",
            dict(domain=rsrl_amd.CART_POLE, basis=rsrl_amd.TILE_CODING, n_tilings=8, tiles_per_dim=8, algo=rsrl_amd.SARSA_LAMBDA, n_envs=65536,
                 policy=rsrl_amd.EPSILON_GREEDY, epsilon=0.1, gamma=0.99, alpha=0.0125 / 65536, lam=0.9, weight_mode=rsrl_amd.W_SHARED, max_episode_steps=200,
                 env_offset=rank * 65536, device=device), 256, 64, 2 * 512 * 8 + 16 * 2 * 4 + 512 * 8,
            "per learner-step: every LIVE trace entry's 16-bit key + value read and its value written back (10 B) + 2 x 8 gathers + the new keys and lengths (mean_live_entries "
            "of 512 per learner, measured after the run); k_shared_ca -> k_sparse_trace_scatter -> k_apply_rep (kernels_sparse_lambda.hpp): a 16-lane group per "
            "(learner, tiling), four learners per wave, one block per CU; with short lists latency- and line-bound (a sub-list of ~11-20 live entries "
            "is one partly used 128-byte line each way); with full lists (a learned policy: episodes end at the step cap, nothing resets) the scatter kernel runs at the "
            "measured copy bandwidth (75 us per batch-step).  Round 5's form took 246 us per batch-step at 16 384 learners, this one 27", sparse_trace_bytes), 180)
        lam_generic = guarded(lambda: hbm_leg(
            rsrl_amd, "SARSALambda on a generic Fourier order (CartPole, order 3: F = 256), per-learner W and trace in memory, 65536 envs",
            dict(domain=rsrl_amd.CART_POLE, order=3, algo=rsrl_amd.SARSA_LAMBDA, n_envs=65536, policy=rsrl_amd.EPSILON_GREEDY, epsilon=0.1, gamma=0.99,
                 alpha=1e-3, lam=0.8, max_episode_steps=200, env_offset=rank * 65536, device=device), 128, 32, 4 * 256 * 2 * 4,
            "per learner-step: W and Z read and written once (4 x A F x 4 B); k_train_lambda_mem4, four threads per learner (kernels_lambda_mem.hpp); "
            "round 5's one-thread form: 0.03"), 120)
    # shared-W legs: `shared_w` = what a user gets (exchange AUTO: the one-hop peer exchange whenever every rank's device reaches every
    # other's, and then the persistent kernel); `shared_w_rccl` = the any-topology fallback asked for explicitly
    shared = shared_rccl = None
    if not args.no_shared_leg:
        shared = guarded(lambda: shared_w_leg(cp, rsrl_amd, make_sharded_context, rsrl_amd.EXCHANGE_AUTO,
                                              envs_per_gpu=131072 if world <= ndev else max(512, min(131072, args.envs))), 240)
        if world > ndev:
            # RCCL admits one rank per device (ncclCommInitRank: "invalid usage" for a duplicate GPU): on an oversubscribed test box the leg
            # has nothing to measure.  The peer exchange above runs: ranks that share a device decide together which kernels fit it.
            shared_rccl = {"skipped": f"{world} ranks on {ndev} device(s): RCCL refuses ranks that share a device (shared_w, the peer exchange, runs)"}
        elif not (isinstance(shared, dict) and shared.get("error") == "timeout"):
            shared_rccl = guarded(lambda: shared_w_leg(cp, rsrl_amd, make_sharded_context, rsrl_amd.EXCHANGE_RCCL), 240)
    hung = any(isinstance(x, dict) and x.get("error") == "timeout" for x in (streaming, streaming_hbm, shared, shared_rccl, c3, c5, trait, trait_unfused, trait_1m))

    # every rank's C stdio (RCCL's banner) goes out BEFORE rank 0 composes the result: the result line is the last line of the job's stdout
    _flush_c_stdio()
    if world > 1 and not hung:
        guarded(cp.barrier, 120)
    if rank == 0:
        total_steps = args.steps * repeats
        total_env_steps = total_steps * args.envs * world
        value = total_env_steps / dt
        avg_launch_s = kernel_ms * 1e-3 / max(1, launches)
        steps_per_launch = total_steps * len(regions) / max(1, launches)
        per_gpu_kernel_rate = args.envs * steps_per_launch / avg_launch_s if avg_launch_s > 0 else 0.0
        n_gpus = min(world, ndev)
        fused = kname == "k_train_reg"
        traffic = pmc_traffic(kname, args.envs, round(steps_per_launch))
        traffic_src = ("rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE, separate passes; profiles/pmc_traffic.json; the fused loop's bytes "
                       "per launch do not depend on its depth: passes at 256 and at 20 steps per launch)")
        if traffic is None and fused:
            traffic = float(FUSED_BYTES_PER_LEARNER_LAUNCH * args.envs)
            traffic_src = ("analytic: 920 B per learner per launch (W in + out, state, action, episode counter, carried Q), independent of the "
                           "depth; the PMC pass at 256 steps per launch measured 1.006x this figure (profiles/pmc_traffic.json)")
        out = {
            "metric": "env-steps/sec (whole node), MountainCar Q-learning Fourier-5",
            "value": value, "unit": "env-steps/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / total_steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "timed_region_s": dt, "regions_s": [r[0] for r in regions],
            "spread": (max(r[0] for r in regions) - min(r[0] for r in regions)) / dt,
            "timing": f"median of {len(regions)} regions, each {repeats} back-to-back calls of {args.steps} batch-steps between one barrier + synchronize pair",
            "ranks_seen": world, "per_rank_env_steps_per_s": [float(x) for x in own_rates], "per_rank": per_rank,
            "config": {"workload": f"{args.envs} vectorised MountainCar envs per GPU, QLearning + Fourier(5), "
                                   "eps-greedy(0.1), gamma 0.9, SGD(0.001), per-env W, 1xMI355X per rank "
                                   "(BASELINE.json configs[1])",
                       "envs_per_gpu": args.envs, "steps_per_launch": steps_per_launch, "repeats": repeats,
                       "timed": f"{repeats} back-to-back calls of {args.steps} batch-steps per region, {len(regions)} regions",
                       "ranks": world,
                       "parallelism": f"env-sharded x{world}, no data-path collective"},
        }
        if world > ndev:
            out["oversubscribed"] = f"{world} ranks share {ndev} device(s): not a scaling measurement"
        common = {"kernel": kname, "avg_launch_ms": avg_launch_s * 1e3, "launches": launches,
                  "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src}
        if fused:
            # the fused kernel keeps W in the register file for the whole launch: the bytes of the streaming formulation are not
            # moved, so the HBM "roofline" of SURVEY 8(d) does not bound it.  Its bound is VALU issue.
            rl = valu_roofline(kname, per_gpu_kernel_rate) or {"bound": "valu", "error": "profiles/isa_mix.json missing"}
            rl.update(common)
            if traffic and avg_launch_s > 0:
                rl["hbm"] = {"achieved": traffic / avg_launch_s / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                             "frac": traffic / avg_launch_s / HBM_PEAK, "what": "REAL traffic of the launch / its duration"}
            gate_on_profile(rl, [kname])
            rl["streaming_formulation_equivalent"] = {
                "algorithmic_bytes_per_env_step": BYTES_PER_ENV_STEP,
                "equivalent_GBps": BYTES_PER_ENV_STEP * per_gpu_kernel_rate / 1e9,
                "note": "608 B/env-step x rate: what an unfused implementation would have to stream; NOT a bandwidth this kernel "
                        "moves and not a roofline fraction (see roofline_streaming for the kernel that does move them)"}
            out["roofline"] = rl
        else:
            ach = BYTES_PER_ENV_STEP * args.envs * steps_per_launch / avg_launch_s if avg_launch_s > 0 else 0.0
            out["roofline"] = dict({"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
                                    "algorithmic_bytes_per_env_step": BYTES_PER_ENV_STEP}, **common)
        if rollout is not None:
            out["greedy_rollout"] = rollout
        if parity is not None:
            out["parity"] = parity
        if no_coalesce is not None:
            out["value_no_coalesce"] = no_coalesce
        if streaming is not None:
            out["roofline_streaming"] = streaming
        if streaming_hbm is not None:
            out["roofline_streaming_hbm"] = streaming_hbm
        if trait is not None:
            out["trait_loop"] = trait
        if trait_unfused is not None:
            out["trait_loop_unfused"] = trait_unfused
        if trait_1m is not None:
            out["trait_loop_1m"] = trait_1m
        if copy_bw is not None:
            out["hbm_copy_measured"] = copy_bw
            if isinstance(copy_bw, dict) and copy_bw.get("GBps"):
                for leg in (streaming, streaming_hbm, trait, trait_unfused, trait_1m):
                    if isinstance(leg, dict) and leg.get("achieved"):
                        leg["frac_of_measured_copy"] = leg["achieved"] / copy_bw["GBps"]
        if c3 is not None:
            out["c3_shared_tiles"] = c3
        if c5 is not None:
            out["c5_wave_bf16"] = c5
        if lam_shared is not None:
            out["lambda_shared_tiles"] = lam_shared
        if lam_generic is not None:
            out["lambda_generic_order"] = lam_generic
        if shared is not None:
            out["shared_w"] = shared
        if shared_rccl is not None:
            out["shared_w_rccl"] = shared_rccl
        if not args.no_cpu_baseline:
            # the other ranks are idle by now (they wait in the closing barrier below): the host cores are rank 0's
            out["cpu_baseline"] = cpu_baseline()
            if world > 1:
                out["cpu_baseline"]["measured_with_ranks"] = world
        emit(out, args.detail)
    if hung:
        os._exit(0)          # a secondary leg is stuck in a collective: do not wait for it in the destructors
    if world > 1:
        guarded(cp.barrier, 120)          # the ranks leave together (rank 0 was timing the CPU baseline on the host cores)
    ctx.close()
    cp.close()


if __name__ == "__main__":
    main()
