#!/usr/bin/env python3
"""gpurun_out/prof (scripts/gpu_profile.sh) -> the tracked evidence:

    profiles/<tag>_bench_driver.json      the driver's invocation, plain
    profiles/<tag>_kernel_stats.md        rocprofv3 --kernel-trace --stats of that command (and of its 1024-step variant, one population)
    profiles/<tag>_kernel_stats_legs.md   ... of bench.py with every secondary leg, and of each leg run bare (scripts/profile_leg.py)
    profiles/<tag>_pmc_raw.json           per leg and kernel: mean per dispatch of every counter (one rocprofv3 run per counter group)
    profiles/isa_mix.json                 per kernel: VALU instructions, flop and class counters PER ENV-STEP   (bench.py: valu rooflines)
    profiles/pmc_traffic.json             per kernel: HBM bytes per launch                                          (bench.py: roofline.traffic)

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a
wide streaming read => fetched bytes = 2 * FETCH_SIZE * 1024.  Counters are summed over their instances (XCDs / SEs) per dispatch and
averaged over the dispatches of the kernel after its first two.

    python scripts/summarize_profile.py r04
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles")

# leg -> kernels of interest (substring of the kernel name, dominant first)
LEG_KERNELS = {
    "fused": ["k_train_reg"], "stream": ["k_step_reg_lm"], "stream1m": ["k_step_reg_q4"], "persist": ["k_shared_persist"], "perstep": ["k_shared_step"],
    "tile": ["k_shared_ca", "k_tile_scatter", "k_apply_rep"], "wave": ["k_train_wave_pk"],
}
# static counts of the executed path of k_train_reg's steady-state loop (scripts/isa_stats.py on train_reg_d0b.hip): v_pk_fma_f32 314 per
# pair of steps; v_pk_mul_f32 42 + v_pk_add_f32 4; v_cndmask_b32 30 (after the assembly post-pass of rsrl_amd/_asmfilter.py: 719 instructions per
# pair of steps, 26 of them scalar)
K_TRAIN_REG_STATIC = {"pk_fma": 157.0, "pk_other": 23.0, "cndmask": 15.0,
                      # of the 54 packed fmas of the column update only the taken action's 18 change a weight: the other 36 multiply by zero because
                      # the action is per lane and registers cannot be indexed by a lane value -- executed, not useful
                      "pk_fma_masked_zero": 36.0}


def first_json(path):
    try:
        for ln in open(path, errors="replace").read().splitlines():
            if ln.startswith("{"):
                return json.loads(ln)
    except OSError:
        pass
    return None


def short(name):
    full = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return (full.split("(rsrl::")[0] if "<" in full else full.split("(")[0]).strip()


def stats_table(d, top=8):
    ks = glob.glob(os.path.join(d, "*kernel_stats.csv"))
    rows = ["| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    if not ks:
        return "\n".join(rows + ["| (no kernel_stats.csv) | | | | |"])
    for i, row in enumerate(csv.DictReader(open(ks[0]))):
        if i >= top:
            break
        rows.append(f"| `{short(row['Name'])[:110]}` | {row['Calls']} | {float(row['AverageNs']) / 1e3:.2f} | {float(row['TotalDurationNs']) / 1e6:.2f} | {row['Percentage']} |")
    return "\n".join(rows)


def kernel_avg_us(d, sub):
    ks = glob.glob(os.path.join(d, "*kernel_stats.csv"))
    if not ks:
        return None
    for row in csv.DictReader(open(ks[0])):
        if sub in row["Name"]:
            return float(row["AverageNs"]) / 1e3, int(row["Calls"])
    return None


def counter_means(d, sub, skip_first=2):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))
    if not f:
        return {}, 0.0, 0
    per, dur = defaultdict(lambda: defaultdict(float)), {}
    for row in csv.DictReader(open(f[0])):
        if sub not in row["Kernel_Name"]:
            continue
        per[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
        dur[int(row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
    out = {}
    for name, by in per.items():
        ids = sorted(by)[skip_first:] or sorted(by)
        out[name] = sum(by[i] for i in ids) / len(ids)
    ids = sorted(dur)[skip_first:] or sorted(dur)
    return out, (sum(dur[i] for i in ids) / len(ids) if ids else 0.0), len(ids)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    os.makedirs(P, exist_ok=True)
    plain = first_json(os.path.join(G, "bench_driver.json"))
    prof = first_json(os.path.join(G, "bench_driver_profiled.json"))
    prof1k = first_json(os.path.join(G, "bench_1024_profiled.json"))
    if plain:
        json.dump(plain, open(os.path.join(P, f"{tag}_bench_driver.json"), "w"), indent=1)

    def line(d, what):
        if not d:
            return f"* {what}: (missing)"
        r = d["roofline"]
        return (f"* {what}: value {d['value']:.4g} env-steps/s, `{r['kernel']}` {r['avg_launch_ms'] * 1e3:.2f} us per launch by HIP events "
                f"({r['launches']} launches of {d['config']['steps_per_launch']:.1f} batch-steps = {r['avg_launch_ms'] * 1e3 / d['config']['steps_per_launch']:.4f} us per batch-step), "
                f"roofline.frac " + (f"{r['frac']:.3f}" if r.get('frac') is not None else
                                     f"withheld (this run preceded the stamping of the profile it would be computed from; from the stale constants: {r.get('from_stale_profile', {}).get('frac')})"))
    md = [f"# rocprofv3 --kernel-trace --stats of the bench command ({tag}, 1 x MI355X)", "",
          "## `python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's invocation)", "",
          line(plain, "plain run"), line(prof, "the same command under rocprofv3 (`--no-cpu-baseline`, legs off)"), "",
          "The profiler's `k_train_reg` row mixes the coalesced launches of the timed regions with the one-launch-per-call launches of the "
          "`value_no_coalesce` leg, the warm-up and the calibration burst; the 1024-step command below has ONE population (a call is a quarter of a "
          "4096-step launch): there the profiler's average and the HIP events agree.", "", stats_table(os.path.join(G, "kt_driver")), "",
          "## `python bench.py --gpus 1 --steps 1024 --warmup 5`", "", line(prof1k, "under rocprofv3"), "", stats_table(os.path.join(G, "kt_1024")), ""]
    open(os.path.join(P, f"{tag}_kernel_stats.md"), "w").write("\n".join(md) + "\n")

    legs_md = [f"# rocprofv3 --kernel-trace --stats of the other configurations ({tag}, 1 x MI355X)", "",
               "## `bench.py --steps 1024` with every secondary leg (C3 tile coding, C5 wave family, streaming kernels, shared-W legs)", "",
               stats_table(os.path.join(G, "kt_legs"), 14), ""]
    raw, mix, traffic = {}, {}, {}
    tp = os.path.join(P, "pmc_traffic.json")
    ip = os.path.join(P, "isa_mix.json")
    for leg, kernels in LEG_KERNELS.items():
        d = os.path.join(G, leg)
        if not os.path.isdir(d):
            continue
        info = first_json(os.path.join(d, "plain.json")) or {}
        legs_md += [f"## leg `{leg}` (scripts/profile_leg.py {leg}): " + json.dumps({k: info.get(k) for k in ("kernel", "learners", "steps_per_dispatch", "event_us_per_batch_step", "env_steps_per_s")}), "",
                    stats_table(os.path.join(d, "kt"), 6), ""]
        for sub in kernels:
            rec, durs = {}, {}
            for p in sorted(glob.glob(os.path.join(d, "p[0-9]"))):
                m, us, n = counter_means(p, sub)
                rec.update(m)
                if us:
                    durs[os.path.basename(p)] = us
            if not rec:
                continue
            ka = kernel_avg_us(os.path.join(d, "kt"), sub)
            rec["kernel_us_under_pmc"] = durs
            if ka:
                rec["kernel_us_kernel_trace"], rec["kernel_trace_calls"] = ka
            raw[f"{leg}:{sub}"] = rec
            learners = info.get("learners", 0)
            spd = info.get("steps_per_dispatch", 1) if sub in ("k_train_reg", "k_shared_persist", "k_train_wave", "k_train_wave_pk") else 1
            env_steps = float(learners) * spd                       # env-steps one dispatch of this kernel covers
            if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
                fetch, write = 2.0 * rec["FETCH_SIZE"] * 1024.0, rec["WRITE_SIZE"] * 1024.0
                traffic[sub] = {"envs": learners, "steps_per_launch": spd, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                                "traffic_bytes_per_launch": fetch + write, "bytes_per_env_step": (fetch + write) / env_steps if env_steps else None,
                                # what the bytes of a launch follow: its number of batch-steps (every step streams / exchanges), or nothing but
                                # the launch itself (W in + out once, whatever the depth)
                                "scales": "per_launch" if sub in ("k_train_reg", "k_train_wave", "k_train_wave_pk") else "per_step",
                                "source": f"profiles/{tag}_pmc_raw.json [{leg}:{sub}] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 gfx950 correction, KiB units)"}
            if "SQ_INSTS_VALU" in rec and env_steps:
                per = lambda k: rec.get(k, 0.0) / env_steps             # noqa: E731   wave-instructions (or counter units) per env-step
                lane = lambda k: rec.get(k, 0.0) * 64.0 / env_steps     # noqa: E731   ... per env-step of ONE lane's learner (one thread per learner)
                e = {"valu_wave_instr_per_env_step": per("SQ_INSTS_VALU"),
                     "flop_per_env_step_classes_unpacked": 64.0 * (2.0 * per("SQ_INSTS_VALU_FMA_F32") + per("SQ_INSTS_VALU_MUL_F32") + per("SQ_INSTS_VALU_ADD_F32")),
                     "flops_counter_raw_per_env_step": per("SQ_INSTS_VALU_FLOPS_FP32"), "mfma_mops_f32_per_env_step": per("SQ_INSTS_VALU_MFMA_MOPS_F32"),
                     "mfma_f32_instr_per_env_step": per("SQ_INSTS_VALU_MFMA_F32"),
                     "valu_instr": lane("SQ_INSTS_VALU"), "fma_f32": lane("SQ_INSTS_VALU_FMA_F32"), "mul_f32": lane("SQ_INSTS_VALU_MUL_F32"),
                     "add_f32": lane("SQ_INSTS_VALU_ADD_F32"), "trans_f32": lane("SQ_INSTS_VALU_TRANS_F32"), "int32": lane("SQ_INSTS_VALU_INT32"),
                     "int64": lane("SQ_INSTS_VALU_INT64"), "cvt": lane("SQ_INSTS_VALU_CVT"), "salu": lane("SQ_INSTS_SALU"), "lds_instr": lane("SQ_INSTS_LDS"),
                     "wave_quad_cycles": lane("SQ_WAVE_CYCLES"), "wait_any_quad": lane("SQ_WAIT_ANY"), "active_inst_any_quad": lane("SQ_ACTIVE_INST_ANY"),
                     "active_inst_valu_quad": lane("SQ_ACTIVE_INST_VALU"),
                     "unit": "valu_instr .. active_inst_valu_quad: per env-step as ONE LANE's learner sees them (wave-instructions x 64 / env-steps; k_train_wave runs one "
                             "learner per WAVE: divide by 64 for its per-wave counts); *_per_env_step: counter units / env-steps",
                     "env_steps_per_dispatch": env_steps, "waves_per_dispatch": rec.get("SQ_WAVES"), "source": f"profiles/{tag}_pmc_raw.json [{leg}:{sub}]"}
                mix[sub] = e
        # (k_train_reg keeps its established fields below)
    json.dump(raw, open(os.path.join(P, f"{tag}_pmc_raw.json"), "w"), indent=1)
    # the profile belongs to the BINARY it was taken from: the sha256 of every profiled kernel's machine code (all instantiations), computed on the
    # GPU box from the library the profiled processes loaded (scripts/gpu_profile.sh -> kernel_digests.json; rsrl_amd/_kdigest.py).  bench.py prints a
    # roofline fraction from these constants only while the library it loaded carries the same code (`profile_digest_matches`).
    digests = first_json(os.path.join(G, "kernel_digests.json")) or {}
    for table in (mix, traffic):
        for k, e in table.items():
            e["code_sha256"] = digests.get(k)
    old_traffic = json.load(open(tp)) if os.path.exists(tp) else {}
    if "k_train_reg" in traffic:                                    # keep the list form (entries per launch depth) bench.py reads
        keep = [x for x in old_traffic.get("k_train_reg", []) if isinstance(x, dict) and x.get("steps_per_launch") != traffic["k_train_reg"]["steps_per_launch"]]
        traffic["k_train_reg"] = [traffic["k_train_reg"]] + keep
    old_traffic.update(traffic)
    json.dump(old_traffic, open(tp, "w"), indent=1)
    old_mix = json.load(open(ip)) if os.path.exists(ip) else {}
    if "k_train_reg" in mix:
        e = mix["k_train_reg"]
        st = K_TRAIN_REG_STATIC
        pk = st["pk_fma"] + st["pk_other"]
        e.update({"pk": pk, "pk_fma": st["pk_fma"], "mad_u64": e["int64"], "cndmask": st["cndmask"], "other": e["valu_instr"] - pk - e["int64"] - st["cndmask"],
                  "fp_fma": e["fma_f32"] - st["pk_fma"], "fp_other": e["mul_f32"] + e["add_f32"] - st["pk_other"],
                  "what": "VALU instructions per env-step of k_train_reg<MountainCar, Fourier 5, QLearning, EpsilonGreedy> (65 536 learners, 256 steps per launch): dynamic counts "
                          "from the rocprofv3 class counters; pk / pk_fma / cndmask are static counts of the executed path (scripts/isa_stats.py); mad_u64 = INT64; "
                          "fp_fma = FMA_F32 - pk_fma; fp_other = MUL_F32 + ADD_F32 - the packed ones"})
    # flop per env-step.  k_train_reg's count is known from its static packed counts (pk_fma x 4 + pk_other x 2 + fp_fma x 2 + fp_other): it
    # calibrates what SQ_INSTS_VALU_FLOPS_FP32 counts per wave-instruction; the other kernels' flop then come from that counter (+ 512 per
    # MFMA op: v_mfma_f32_4x4x1 = 16 blocks x 4 x 4 x 2), or -- if the counter does not calibrate -- from the class counters with packed
    # instructions counted once (a LOWER bound)
    scale, how = None, "class counters, packed instructions counted once: a LOWER bound (SQ_INSTS_VALU_FLOPS_FP32 did not calibrate)"
    if "k_train_reg" in mix:
        e = mix["k_train_reg"]
        known = e["pk_fma"] * 4 + (e["pk"] - e["pk_fma"]) * 2 + e["fp_fma"] * 2 + e["fp_other"]
        e["flop_per_env_step"] = known
        e["useful_flop_per_env_step"] = known - 4.0 * K_TRAIN_REG_STATIC["pk_fma_masked_zero"]
        e["useful_flop_how"] = "flop_per_env_step minus the masked column update's multiplies by zero (36 packed fmas x 4 flop)"
        e["flop_how"] = "static packed counts + class counters (the established k_train_reg accounting)"
        rawc = e.get("flops_counter_raw_per_env_step", 0.0)
        for cand, label in ((64.0, "per-lane flops per wave-instruction (x 64 lanes)"), (1.0, "flops of all lanes")):
            if rawc > 0 and abs(rawc * cand / known - 1.0) < 0.05:
                scale, how = cand, f"SQ_INSTS_VALU_FLOPS_FP32 counts {label}: calibrated on k_train_reg ({rawc * cand:.1f} vs {known:.1f} flop per env-step from the static packed counts)"
        e["flops_counter_calibration"] = how
    for k, e in mix.items():
        if k == "k_train_reg":
            continue
        mf = 512.0 * e.get("mfma_mops_f32_per_env_step", 0.0)
        base = e["flops_counter_raw_per_env_step"] * scale if scale and e.get("flops_counter_raw_per_env_step") else e["flop_per_env_step_classes_unpacked"]
        e["flop_per_env_step"] = base + mf
        e["mfma_flop_per_env_step"] = mf
        e["flop_how"] = how + (" + 512 flop per MFMA op" if mf else "")
    old_mix.update(mix)
    json.dump(old_mix, open(ip, "w"), indent=1)
    legs_md += ["## PMC per kernel (mean per dispatch; profiles/%s_pmc_raw.json)" % tag, "", "```"]
    for k, rec in raw.items():
        legs_md.append(k + "  " + "  ".join(f"{c}={v:.5g}" for c, v in sorted(rec.items()) if isinstance(v, (int, float))))
    legs_md += ["```", ""]
    open(os.path.join(P, f"{tag}_kernel_stats_legs.md"), "w").write("\n".join(legs_md) + "\n")
    print(open(os.path.join(P, f"{tag}_kernel_stats.md")).read()[:3000])
    print(json.dumps({k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("valu_instr", "valu_wave_instr_per_env_step", "flop_per_env_step", "flops_counter_raw_per_env_step", "flop_per_env_step_classes_unpacked", "flop_how")} for k, v in mix.items()}, indent=1))
    print(json.dumps({k: (v[0] if isinstance(v, list) else v).get("traffic_bytes_per_launch") for k, v in old_traffic.items()}, indent=1))


if __name__ == "__main__":
    main()
