#!/usr/bin/env python3
"""bench.py -- env-steps/s of the TD-control hot path on MI355X.

Workload (BASELINE.json configs[1]): 65 536 vectorised MountainCar envs per GPU, QLearning + Fourier(5)
+ epsilon-greedy(0.1), gamma 0.9, SGD(0.001), per-env weights, max_episode_steps 1000, synthetic seeded
episodes (all envs start at MountainCar::default(); diversity comes from the per-env Philox streams).
A "step" = one pass of the hot path (transition -> handle -> sample) over the whole batch of envs.

    python bench.py --gpus N --steps K --warmup W

N > 1: launched by torch.distributed.run, one rank per GPU; envs are sharded by global id with NO
data-path collective (independent learners) => weak scaling; value = all ranks' env-steps / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ENVS = 65536
BYTES_PER_ENV_STEP = 608            # 2*D*4 + 8 + 8 + F*A*4 (W read) + F*4 (W column write)
HBM_PEAK = 8.0e12                   # MI355X_MICROARCH.md: 8.0 TB/s spec


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


def pmc_traffic(kernel, envs, steps_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), or None
    when no pass was collected for this kernel/configuration.  bench.py cannot run rocprofv3 on itself."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[kernel]
    except Exception:
        return None
    if rec["envs"] != envs or abs(rec["steps_per_launch"] - steps_per_launch) > 1e-9:
        return None
    return rec["traffic_bytes_per_launch"]


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
    """Secondary measurement: the SAME workload with one batch-step per launch (k_step_reg), i.e. the 608 B/env-step
    streaming formulation the HBM roofline is defined on.  Never part of `value`."""
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
        return {"bound": "hbm", "kernel": kn, "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK,
                "traffic": pmc_traffic(kn, envs, 1), "avg_launch_ms": avg * 1e3, "launches": n,
                "env_steps_per_s_this_rank": envs * steps / dt}
    except Exception as e:
        return {"error": repr(e)}


def shared_w_leg(cp, rsrl_amd, make_sharded_context, envs_per_gpu=131072, steps=300, warmup=50):
    """Secondary measurement (BASELINE.json configs[3]): 131 072 MountainCar envs per GPU, ONE shared Fourier(5)
    approximator, per-batch-step all-reduce of the 432 B weight delta over RCCL.  Never part of `value`."""
    try:
        ctx = make_sharded_context(envs_per_gpu * cp.world, cp, domain=rsrl_amd.MOUNTAIN_CAR, order=5,
                                   algo=rsrl_amd.QLEARNING, policy=rsrl_amd.EPSILON_GREEDY, epsilon=0.1, gamma=0.9,
                                   lr=0.001 / (envs_per_gpu * cp.world), weight_mode=rsrl_amd.W_SHARED, seed=0,
                                   max_episode_steps=1000)
        ctx.reset()
        ctx.train(warmup, want_stats=False)
        ctx.sync()
        cp.barrier()
        t0 = time.perf_counter()
        ctx.train(steps, want_stats=False)
        ctx.sync()
        dt = cp.max_over_ranks(time.perf_counter() - t0)
        w = ctx.get_weights()
        import numpy as np
        chk = np.array([float(np.abs(w).sum())])
        lo, hi = -cp.max_over_ranks(-chk[0]), cp.max_over_ranks(chk[0])
        ctx.close()
        return {"workload": f"{envs_per_gpu} MountainCar envs per GPU, shared-W QLearning Fourier(5), per-step RCCL "
                            f"all-reduce of the 432 B delta", "n_gpus": cp.world, "steps": steps,
                "value": envs_per_gpu * cp.world * steps / dt, "unit": "env-steps/s", "us_per_batch_step": dt / steps * 1e6,
                "replicas_consistent": bool(lo == hi), "sum_abs_w": hi}
    except Exception as e:                       # never let the secondary leg take the headline down
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20480)
    ap.add_argument("--warmup", type=int, default=2048)
    ap.add_argument("--steps-per-launch", type=int, default=0, help="fuse depth (0 = library default)")
    ap.add_argument("--envs", type=int, default=N_ENVS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shared-leg", action="store_true", help="skip the secondary shared-W (RCCL) measurement")
    ap.add_argument("--no-streaming-leg", action="store_true", help="skip the secondary 1-step-per-launch measurement")
    args = ap.parse_args()

    from rsrl_amd.distributed import ControlPlane, make_sharded_context
    cp = ControlPlane()          # gloo control plane (rendezvous / barrier / max over ranks); no-op for one rank
    rank, local_rank, world = cp.rank, cp.info.local_rank, cp.world
    import rsrl_amd
    device = local_rank % max(1, rsrl_amd.device_count())     # one rank per GPU; modulo only matters on under-sized test boxes

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
    cp.barrier()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    ctx.train(args.steps, want_stats=False)
    ctx.sync()
    dt = time.perf_counter() - t0
    dt = cp.max_over_ranks(dt)
    cp.barrier()
    kernel_ms, launches, kname = ctx.timing_read()
    ctx.timing_enable(False)
    n_states, _ = ctx.rollout_greedy(500)
    # secondary legs run under a watchdog: whatever happens to them, rank 0 still prints the headline line
    streaming = guarded(lambda: streaming_leg(rsrl_amd, args.envs, rank, device), 120) \
        if (args.steps_per_launch != 1 and not args.no_streaming_leg) else None
    shared = guarded(lambda: shared_w_leg(cp, rsrl_amd, make_sharded_context), 240) if not args.no_shared_leg else None
    hung = any(isinstance(x, dict) and x.get("error") == "timeout" for x in (streaming, shared))

    if rank == 0:
        total_env_steps = args.steps * args.envs * world
        value = total_env_steps / dt
        avg_launch_s = kernel_ms * 1e-3 / max(1, launches)
        steps_per_launch = args.steps / max(1, launches)   # exact: --steps is a multiple of the fuse depth by default
        algo_bytes_per_launch = BYTES_PER_ENV_STEP * args.envs * steps_per_launch
        achieved = algo_bytes_per_launch / avg_launch_s if avg_launch_s > 0 else 0.0
        out = {
            "metric": "env-steps/sec (whole node), MountainCar Q-learning Fourier-5",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.envs} vectorised MountainCar envs per GPU, QLearning + Fourier(5), "
                                   "eps-greedy(0.1), gamma 0.9, SGD(0.001), per-env W, 1xMI355X per rank "
                                   "(BASELINE.json configs[1])",
                       "envs_per_gpu": args.envs, "steps_per_launch": steps_per_launch,
                       "parallelism": f"env-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK,
                         "traffic": pmc_traffic(kname, args.envs, round(steps_per_launch)),
                         "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/pmc_traffic.json)",
                         "kernel": kname, "avg_launch_ms": avg_launch_s * 1e3, "launches": launches,
                         "algorithmic_bytes_per_env_step": BYTES_PER_ENV_STEP,
                         "note": "algorithmic bytes = 608 B/env-step (unfused streaming formulation) x env-steps "
                                 "per launch; the fused launch keeps W in VGPRs so real HBM traffic is far lower "
                                 "and frac may exceed 1 (see DESIGN.md)"},
            "greedy_rollout_mean_n_states": float(n_states.mean()),
        }
        if streaming is not None:
            out["roofline_streaming"] = streaming
        # the fused kernel is VALU-issue bound (profiles/r01_pmc_summary.md).  Issue cost of one env-step at SATURATED occupancy:
        # VALU instructions per env-step from the PMC pass (SQ_INSTS_VALU / wave / steps, profiles/pmc_traffic.json), of which the
        # static ISA mix has 180 v_pk_fma_f32, 31 v_mad_u64_u32 and 17 v_cndmask (scripts/isa_stats.py); cycles per instruction at
        # 8 waves/SIMD from scripts/ubench/valu_issue.hip (profiles/r01_ubench_valu_issue.txt): pk_fma 4.47, mad_u64 4.96,
        # cndmask 4.13, everything else 2.46.  The launch itself runs at ONE wave per SIMD (65 536 learners = 1024 waves).
        if kname == "k_train_reg":
            try:
                instr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["k_train_reg"]["valu_instr_per_env_step"]
            except Exception:
                instr = None
            if instr:
                n_pk, n_mad, n_cnd = 180.0, 31.0, 17.0
                cyc = n_pk * 4.47 + n_mad * 4.96 + n_cnd * 4.13 + max(0.0, instr - n_pk - n_mad - n_cnd) * 2.46
                simds, clk = 1024, 2.4e9
                peak_steps = simds * 64 * clk / cyc
                out["valu_roofline"] = {"valu_instr_per_env_step": instr, "saturated_issue_cycles_per_env_step": cyc,
                                        "peak_env_steps_per_s_per_gpu": peak_steps, "frac": (value / world) / peak_steps,
                                        "source": "profiles/pmc_traffic.json (SQ_INSTS_VALU / wave / steps), scripts/isa_stats.py (static mix), "
                                                  "profiles/r01_ubench_valu_issue.txt (cycles per instruction at 8 waves/SIMD)"}
        if shared is not None:
            out["shared_w"] = shared
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if hung:
        os._exit(0)          # a secondary leg is stuck in a collective: do not wait for it in the destructors
    ctx.close()
    cp.close()


if __name__ == "__main__":
    main()
