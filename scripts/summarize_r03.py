#!/usr/bin/env python3
"""gpurun_out/ of scripts/gpu_profile_r03.sh -> the tracked evidence under profiles/ (r03_*)."""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def first_json(path):
    for ln in open(path, errors="replace").read().splitlines():
        if ln.startswith("{"):
            return json.loads(ln)
    return None


def table(tag, top=8):
    rows = ["| kernel | calls | avg us | total ms | % | vgpr | lds |", "|---|---|---|---|---|---|---|"]
    for ln in open(os.path.join(G, "kprof", tag, "stats.txt")).read().splitlines()[:top]:
        m = re.match(r"(.*?)\s+n=\s*(\d+) avg\s+([\d.]+) us\s+total\s+([\d.]+) ms\s+([\d.]+)%\s+vgpr (\S+) agpr \S+ sgpr \S+ lds (\S+)", ln)
        if m:
            name = m.group(1).strip().replace("void ", "")
            name = name.split("(rsrl::")[0] if "<" in name else name.split("(")[0]
            rows.append(f"| `{name}` | {m.group(2)} | {m.group(3)} | {m.group(4)} | {m.group(5)} | {m.group(6)} | {m.group(7)} |")
    return "\n".join(rows)


def main():
    os.makedirs(P, exist_ok=True)
    plain = first_json(os.path.join(G, "r03", "bench_driver.json"))
    prof = first_json(os.path.join(G, "kprof", "bench_driver", "kt.log"))
    plain1k = first_json(os.path.join(G, "r03", "bench_1024.json"))
    prof1k = first_json(os.path.join(G, "kprof", "bench_1024", "kt.log"))
    for name, d in (("r03_bench_driver.json", plain), ("r03_bench_1024.json", plain1k)):
        json.dump(d, open(os.path.join(P, name), "w"), indent=1)
    shutil.copy(os.path.join(G, "r03", "bench_configs.jsonl"), os.path.join(P, "r03_bench_configs.jsonl"))

    def line(d, what):
        r = d["roofline"]
        return (f"* {what}: value {d['value']:.4g} env-steps/s, `{r['kernel']}` {r['avg_launch_ms'] * 1e3:.2f} us per launch by HIP events "
                f"({r['launches']} launches of {d['config']['steps_per_launch']:.1f} batch-steps = {r['avg_launch_ms'] * 1e3 / d['config']['steps_per_launch']:.4f} us per batch-step), "
                f"roofline.frac {r['frac']:.3f} (fp32 vector), issue slots {r['issue_slots']['frac']:.3f}")
    md = ["# rocprofv3 --kernel-trace of the bench command (round 3, 1 x MI355X)", "",
          "## `python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's invocation)", "",
          line(plain, "plain run"), line(prof, "the same command under rocprofv3 (`--no-cpu-baseline`, legs off)"), "",
          f"The profiler's `k_train_reg` row mixes two populations: the ~{prof['roofline']['launches']} coalesced launches of the five timed regions "
          f"({prof['roofline']['avg_launch_ms'] * 1e3:.0f} us each by HIP events) and the {prof['value_no_coalesce']['launches']} one-launch-per-call launches of the "
          f"`value_no_coalesce` leg ({prof['value_no_coalesce']['avg_launch_ms'] * 1e3:.1f} us each), plus warm-up and calibration: "
          f"({prof['roofline']['launches']} x {prof['roofline']['avg_launch_ms'] * 1e3:.0f} + {prof['value_no_coalesce']['launches']} x "
          f"{prof['value_no_coalesce']['avg_launch_ms'] * 1e3:.1f}) / {prof['roofline']['launches'] + prof['value_no_coalesce']['launches']} = "
          f"{(prof['roofline']['launches'] * prof['roofline']['avg_launch_ms'] * 1e3 + prof['value_no_coalesce']['launches'] * prof['value_no_coalesce']['avg_launch_ms'] * 1e3) / (prof['roofline']['launches'] + prof['value_no_coalesce']['launches']):.0f} us,",
          "which is the row's average.  The second command below has one population only: there the profiler's average and the HIP events agree.", "", table("bench_driver"), "",
          "## `python bench.py --gpus 1 --steps 1024 --warmup 5`: a call is exactly one 1024-step launch, profiled or not", "",
          line(plain1k, "plain run"), line(prof1k, "under rocprofv3"), "", table("bench_1024"), ""]
    open(os.path.join(P, "r03_kernel_stats.md"), "w").write("\n".join(md) + "\n")
    other = ["# rocprofv3 --kernel-trace of the other configurations (round 3, 1 x MI355X)", ""]
    for tag, what in (("legs", "`bench.py` with every secondary leg (C3 tile coding, C5 wave family, streaming kernel, shared-W legs)"),
                      ("shared_persist", "`scripts/prof_shared.py fourier none`: C4's share, 131 072 learners, ONE persistent launch per train call (64 + 320 batch-steps)"),
                      ("shared_persist_peer", "the same with a peer group of size 1 attached (system-scope hop-2 stores / loads)"),
                      ("shared_perstep", "the same with RSRL_NO_PERSIST=1: one launch per batch-step (round 2's path, with this round's four MFMA chains)"),
                      ("tile", "`scripts/prof_shared.py tile none`: C3, 262 144 learners, three launches per batch-step")):
        other += [f"## {what}", "", table(tag, 6), ""]
    pmc = [ln for ln in open(os.path.join(G, "kprof", "shared_persist", "pmc.txt")).read().splitlines() if "k_shared_persist" in ln]
    other += ["## PMC passes of `k_shared_persist` (mean per launch over the 64- and the 320-step launch = 192 batch-steps, 2 048 waves)", "", "```"] + pmc + ["```", ""]
    open(os.path.join(P, "r03_kernel_stats_other.md"), "w").write("\n".join(other) + "\n")
    print(open(os.path.join(P, "r03_kernel_stats.md")).read())


if __name__ == "__main__":
    main()
