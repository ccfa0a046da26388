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
    pmc_r03()


def pmc_r03():
    """gpurun_out/pmc_r03 (scripts/gpu_pmc_r03.sh) -> profiles/r03_pmc_raw.json, isa_mix.json, pmc_traffic.json.
    Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes
    of a streaming read => fetched bytes = 2 * FETCH_SIZE * 1024."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from summarize_r02 import counter_means
    import glob
    src = os.path.join(G, "pmc_r03")
    if not os.path.isdir(src):
        print("no gpurun_out/pmc_r03: PMC files left as they are")
        return
    raw = {}
    for key, sub, name in (("fu", "k_train_reg", "k_train_reg@256"), ("k1", "k_step_reg_lm", "k_step_reg_lm@1")):
        rec = {}
        for d in sorted(glob.glob(os.path.join(src, key + "_*"))):
            if os.path.isdir(d):
                m, us = counter_means(d, sub)
                rec.update(m)
                if us:
                    rec.setdefault("kernel_us_under_pmc", {})[os.path.basename(d)] = us
        raw[name] = rec
    json.dump(raw, open(os.path.join(P, "r03_pmc_raw.json"), "w"), indent=1)
    tp = os.path.join(P, "pmc_traffic.json")
    traffic = json.load(open(tp)) if os.path.exists(tp) else {}
    srcnote = "profiles/r03_pmc_raw.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 gfx950 correction, KiB units)"
    r = raw["k_train_reg@256"]
    if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
        fetch, write = 2.0 * r["FETCH_SIZE"] * 1024.0, r["WRITE_SIZE"] * 1024.0
        rec = {"envs": 65536, "steps_per_launch": 256, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write, "source": srcnote}
        if "SQ_INSTS_VALU" in r and r.get("SQ_WAVES", 0) > 0:
            rec["valu_instr_per_env_step"] = r["SQ_INSTS_VALU"] / r["SQ_WAVES"] / 256
            rec["wave_quad_cycles_per_env_step"] = r.get("SQ_WAVE_CYCLES", 0.0) / r["SQ_WAVES"] / 256
        old = [x for x in traffic.get("k_train_reg", []) if x.get("steps_per_launch") != 256]
        traffic["k_train_reg"] = [rec] + old
    k1 = raw["k_step_reg_lm@1"]
    if "FETCH_SIZE" in k1 and "WRITE_SIZE" in k1:
        fetch, write = 2.0 * k1["FETCH_SIZE"] * 1024.0, k1["WRITE_SIZE"] * 1024.0
        rec = {"envs": 65536, "steps_per_launch": 1, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write, "source": srcnote}
        if "SQ_INSTS_VALU" in k1 and k1.get("SQ_WAVES", 0) > 0:
            rec["valu_instr_per_env_step"] = k1["SQ_INSTS_VALU"] / k1["SQ_WAVES"]
            rec["wave_quad_cycles_per_env_step"] = k1.get("SQ_WAVE_CYCLES", 0.0) / k1["SQ_WAVES"]
        traffic["k_step_reg_lm"] = rec
    json.dump(traffic, open(tp, "w"), indent=1)
    if "SQ_INSTS_VALU_FMA_F32" in r and r.get("SQ_WAVES", 0) > 0:
        per = lambda k: r.get(k, 0.0) / r["SQ_WAVES"] / 256.0      # noqa: E731
        total, fma, mul, add = per("SQ_INSTS_VALU"), per("SQ_INSTS_VALU_FMA_F32"), per("SQ_INSTS_VALU_MUL_F32"), per("SQ_INSTS_VALU_ADD_F32")
        i64, i32, cvt = per("SQ_INSTS_VALU_INT64"), per("SQ_INSTS_VALU_INT32"), per("SQ_INSTS_VALU_CVT")
        # static counts of the executed path of the steady-state loop (scripts/isa_stats.py on train_reg_d0b.hip, round 3):
        pk_fma, pk_other, cnd = 157.0, 23.0, 17.0      # v_pk_fma_f32 314 / pair of steps; v_pk_mul_f32 42 + v_pk_add_f32 4; v_cndmask_b32 34 (60 before the action masks)
        pk = pk_fma + pk_other
        mix = {"pk": pk, "pk_fma": pk_fma, "mad_u64": i64, "cndmask": cnd, "other": total - pk - i64 - cnd, "fp_fma": fma - pk_fma, "fp_other": (mul + add) - pk_other,
               "counters_per_env_step": {"SQ_INSTS_VALU": total, "FMA_F32": fma, "MUL_F32": mul, "ADD_F32": add, "INT64": i64, "INT32": i32, "CVT": cvt,
                                         "SQ_INSTS_SALU": per("SQ_INSTS_SALU"), "SQ_WAVE_CYCLES_quad": per("SQ_WAVE_CYCLES"), "SQ_WAIT_ANY_quad": per("SQ_WAIT_ANY"),
                                         "SQ_ACTIVE_INST_ANY_quad": per("SQ_ACTIVE_INST_ANY")},
               "what": "VALU instructions per env-step of k_train_reg<MountainCar, Fourier 5, QLearning, EpsilonGreedy> (65 536 learners, 256 steps per launch): "
                       "dynamic counts from the rocprofv3 class counters (profiles/r03_pmc_raw.json: SQ_INSTS_VALU_* / SQ_WAVES / steps); pk = packed fp32 "
                       "instructions (v_pk_fma_f32 = pk_fma, counted inside FMA_F32; v_pk_mul_f32 / v_pk_add_f32 inside MUL_F32 / ADD_F32) and cndmask are static "
                       "counts of the executed path; mad_u64 = INT64 (Philox: 2 per round, one block per TWO steps); fp_fma = FMA_F32 - pk_fma; fp_other = MUL_F32 + ADD_F32 - the packed ones"}
        json.dump({"k_train_reg": mix}, open(os.path.join(P, "isa_mix.json"), "w"), indent=1)
    print(json.dumps({"traffic": traffic, "isa_total": raw["k_train_reg@256"].get("SQ_INSTS_VALU")}, indent=1)[:1800])


if __name__ == "__main__":
    main()
