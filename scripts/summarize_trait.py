#!/usr/bin/env python3
"""gpurun_out/prof/trait (scripts/gpu_round6_profile.sh) -> profiles/<tag>_kernel_stats_trait.md: the trait-granular loop's kernels under rocprofv3
(kernel trace + FETCH_SIZE / WRITE_SIZE in passes of their own), fused (the four calls of a batch-step deferred into one launch) and unfused."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import summarize_profile as sp   # noqa: E402
import trait_loop                # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
G = os.path.join(sp.G, "trait")
N = 65536
md = [f"# The trait-granular loop under rocprofv3 ({tag}, 1 x MI355X, 65 536 MountainCar learners, QLearning + Fourier(5), device pointers)", "",
      "`scripts/trait_loop.py`: `rsrl_hip_domain_step -> rsrl_hip_handle -> rsrl_hip_domain_reset -> rsrl_hip_policy_sample(NULL)` per batch-step, one C-ABI call per",
      "trait method (`examples/q_learning.rs:40-52`).  Algorithmic bytes per env-step if every call streams what it needs once: "
      f"{trait_loop.ALG_BYTES_LOOP} B (`trait_loop.ALG_BYTES`).", ""]
for variant, what in (("fused", "learner-major W (`steps_per_launch = 1`), default: the four calls of a batch-step are deferred and run as ONE launch (`k_trait_lm<step>`)"),
                      ("unfused", "learner-major W, `RSRL_NO_TRAIT_DEFER=1`: one kernel per call, Q(s',.) handed from `handle` to `sample` through the ctx's cache"),
                      ("generic", "feature-major W (`steps_per_launch = 0`: the layout of the fused driver loop): the generic one-thread-per-transition kernels "
                                  "`k_handle` / `k_qop` (`models.hpp`), no hand-over -- `sample` evaluates Q(s',.) again")):
    d = os.path.join(G, variant)
    plain = sp.first_json(os.path.join(d, "plain.json")) or {}
    md += [f"## {variant} -- {what}", "",
           f"plain run: {plain.get('us_per_batch_step', float('nan')):.2f} us per batch-step wall, {plain.get('value', 0):.3e} env-steps/s, "
           f"{plain.get('frac_of_8TBps', 0):.3f} of 8 TB/s on the algorithmic bytes", "", sp.stats_table(os.path.join(d, "kt"), top=6), ""]
    rows = ["| kernel | avg us (trace) | HBM bytes fetched / launch | written / launch | per env-step | algorithmic per env-step | traffic / algorithmic |", "|---|---|---|---|---|---|---|"]
    for sub, alg in (("k_trait_lm", None), ("k_handle", trait_loop.ALG_BYTES["handle"]), ("k_qop", trait_loop.ALG_BYTES["policy_sample_reeval"]),
                     ("k_domain_step", trait_loop.ALG_BYTES["domain_step"]), ("k_domain_reset", trait_loop.ALG_BYTES["domain_reset"])):
        us = sp.kernel_avg_us(os.path.join(d, "kt"), sub)
        f = sp.counter_means(os.path.join(d, "p1"), sub)[0].get("FETCH_SIZE")
        w = sp.counter_means(os.path.join(d, "p2"), sub)[0].get("WRITE_SIZE")
        if us is None or f is None or w is None:
            continue
        us = us[0]
        fb, wb = 2 * f * 1024, w * 1024                      # KiB; gfx950 FETCH_SIZE reports half of a wide streaming read (MI355X_MICROARCH.md)
        if alg is None:
            alg = trait_loop.ALG_BYTES_LOOP if variant == "fused" else trait_loop.ALG_BYTES["handle"] + trait_loop.ALG_BYTES["policy_sample_handover"]
            note = "" if variant == "fused" else " (handle + sample launches averaged)"
        else:
            note = ""
        per = (fb + wb) / N
        rows.append(f"| {sub}{note} | {us:.2f} | {fb:.3e} | {wb:.3e} | {per:.0f} B | {alg} B | {per / alg:.2f} |")
    md += rows + [""]
open(os.path.join(sp.P, f"{tag}_kernel_stats_trait.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
