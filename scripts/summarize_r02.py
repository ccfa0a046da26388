#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of a GPU visit (gpurun_out/<visit>/) into the tracked files under profiles/:

    profiles/<tag>_kernel_stats.md   rocprofv3 --kernel-trace --stats of the driver's bench command (per-kernel averages)
    profiles/<tag>_pmc_raw.json      per-launch means of every PMC counter collected for k_train_reg (fu_* = 256 steps per launch,
                                     k20_* = 20 steps per launch with RSRL_NO_COALESCE=1)
    profiles/<tag>_bench*.json       the bench lines of the same visit
    profiles/pmc_traffic.json        what bench.py reads for roofline.traffic (k_train_reg at both depths; k_step_reg_lm kept)
    profiles/isa_mix.json            dynamic VALU instruction mix per env-step from the SQ_INSTS_VALU_* class counters

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of a streaming read => fetched bytes = 2 * FETCH_SIZE * 1024.

    python scripts/summarize_r02.py v3 r02
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENVS = 65536


def counter_means(d, kernel_sub, skip_first=2):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))
    if not f:
        return {}, 0.0
    per, dur = defaultdict(lambda: defaultdict(float)), {}
    for row in csv.DictReader(open(f[0])):
        if kernel_sub not in row["Kernel_Name"]:
            continue
        per[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
        dur[int(row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
    out = {}
    for name, by in per.items():
        ids = sorted(by)[skip_first:] or sorted(by)
        out[name] = sum(by[i] for i in ids) / len(ids)
    ids = sorted(dur)[skip_first:] or sorted(dur)
    return out, (sum(dur[i] for i in ids) / len(ids) if ids else 0.0)


def main():
    visit, tag = sys.argv[1], sys.argv[2]
    src, out = os.path.join(ROOT, "gpurun_out", visit), os.path.join(ROOT, "profiles")
    # ---- kernel stats of the driver's command
    ks = glob.glob(os.path.join(src, "stats_driver", "*kernel_stats.csv"))
    lines = [f"# rocprofv3 --kernel-trace --stats of `python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline` ({tag}, 1 x MI355X)", "",
             "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    if ks:
        for row in csv.DictReader(open(ks[0])):
            full = row["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            nm = (full.split("(rsrl::")[0] if "<" in full else full.split("(")[0]).strip()
            lines.append(f"| `{nm}` | {row['Calls']} | {float(row['AverageNs']) / 1e3:.2f} | {float(row['TotalDurationNs']) / 1e6:.2f} | {row['Percentage']} |")
    for name in ("bench_driver.json", "bench_k20_nocoalesce.json"):
        bp = os.path.join(src, name)
        if os.path.exists(bp):
            for ln in open(bp):
                if ln.startswith("{"):
                    b = json.loads(ln)
                    json.dump(b, open(os.path.join(out, f"{tag}_{name}"), "w"), indent=1)
                    r = b["roofline"]
                    lines += ["", f"`{name}` (same visit, no profiler): value {b['value']:.4g} env-steps/s, `{r['kernel']}` {r['avg_launch_ms'] * 1e3:.2f} us per launch by HIP "
                                  f"events ({r['launches']} launches of {b['config']['steps_per_launch']:.1f} batch-steps), timed region {b.get('timed_region_s', 0):.3f} s."]
    open(os.path.join(out, f"{tag}_kernel_stats.md"), "w").write("\n".join(lines) + "\n")
    # ---- PMC passes of the fused kernel
    raw = {}
    for key, spl in (("fu", 256), ("k20", 20)):
        rec = {}
        for d in sorted(glob.glob(os.path.join(src, key + "_*"))):
            if not os.path.isdir(d):
                continue
            m, us = counter_means(d, "k_train_reg")
            rec.update(m)
            if us:
                rec.setdefault("kernel_us_under_pmc", {})[os.path.basename(d)] = us
        raw[f"k_train_reg@{spl}"] = rec
    json.dump(raw, open(os.path.join(out, f"{tag}_pmc_raw.json"), "w"), indent=1)
    tp = os.path.join(out, "pmc_traffic.json")
    traffic = json.load(open(tp)) if os.path.exists(tp) else {}
    recs = []
    for spl in (256, 20):
        r = raw.get(f"k_train_reg@{spl}", {})
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            fetch, write = 2.0 * r["FETCH_SIZE"] * 1024.0, r["WRITE_SIZE"] * 1024.0
            rec = {"envs": ENVS, "steps_per_launch": spl, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
                   "source": f"profiles/{tag}_pmc_raw.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 gfx950 correction, KiB units)"}
            if "SQ_INSTS_VALU" in r and r.get("SQ_WAVES", 0) > 0:
                rec["valu_instr_per_env_step"] = r["SQ_INSTS_VALU"] / r["SQ_WAVES"] / spl
                rec["wave_quad_cycles_per_env_step"] = r.get("SQ_WAVE_CYCLES", 0.0) / r["SQ_WAVES"] / spl
            recs.append(rec)
    if recs:
        traffic["k_train_reg"] = recs
    json.dump(traffic, open(tp, "w"), indent=1)
    # ---- dynamic instruction mix (per wave per batch-step = per env-step of a lane)
    r = raw.get("k_train_reg@256", {})
    if "SQ_INSTS_VALU_FMA_F32" in r and r.get("SQ_WAVES", 0) > 0:
        per = lambda k: r.get(k, 0.0) / r["SQ_WAVES"] / 256.0      # noqa: E731
        total, fma, mul, add = per("SQ_INSTS_VALU"), per("SQ_INSTS_VALU_FMA_F32"), per("SQ_INSTS_VALU_MUL_F32"), per("SQ_INSTS_VALU_ADD_F32")
        i64, i32, cvt = per("SQ_INSTS_VALU_INT64"), per("SQ_INSTS_VALU_INT32"), per("SQ_INSTS_VALU_CVT")
        # static counts of the executed path of the steady-state loop (scripts/isa_stats.py on train_reg_d0b.hip):
        pk_fma = 157.0        # v_pk_fma_f32: Q(s',.) 54 + column update 54 + rank-1 dot 18 + projection (tables 16, sincos 10, products 5)
        pk_other = 23.0       # v_pk_mul_f32 21 (projection) + v_pk_add_f32 2
        pk = pk_fma + pk_other
        cnd = 30.0            # v_cndmask_b32 (60 per unrolled pair of steps)
        mix = {"pk": pk, "pk_fma": pk_fma, "mad_u64": i64, "cndmask": cnd, "other": total - pk - i64 - cnd, "fp_fma": fma - pk_fma,
               "fp_other": (mul + add) - pk_other,
               "counters_per_env_step": {"SQ_INSTS_VALU": total, "FMA_F32": fma, "MUL_F32": mul, "ADD_F32": add, "INT64": i64, "INT32": i32, "CVT": cvt,
                                         "SQ_INSTS_SALU": per("SQ_INSTS_SALU"), "SQ_WAVE_CYCLES_quad": per("SQ_WAVE_CYCLES"), "SQ_WAIT_ANY_quad": per("SQ_WAIT_ANY"),
                                         "SQ_ACTIVE_INST_ANY_quad": per("SQ_ACTIVE_INST_ANY")},
               "what": "VALU instructions per env-step of k_train_reg<MountainCar, Fourier 5, QLearning, EpsilonGreedy> (65 536 learners, 256 steps per launch): "
                       f"dynamic counts from the rocprofv3 class counters (profiles/{tag}_pmc_raw.json: SQ_INSTS_VALU_* / SQ_WAVES / steps); pk = packed fp32 "
                       "instructions (v_pk_fma_f32 = pk_fma, counted inside FMA_F32; v_pk_mul_f32 / v_pk_add_f32 inside MUL_F32 / ADD_F32) and cndmask are static "
                       "counts of the executed path; mad_u64 = INT64 (Philox: 2 per round, one block per TWO steps); fp_fma = FMA_F32 - pk_fma; fp_other = MUL_F32 + ADD_F32 - the packed ones"}
        json.dump({"k_train_reg": mix}, open(os.path.join(out, "isa_mix.json"), "w"), indent=1)
    # ---- the streaming kernel's PMC passes (k1_*) refresh its traffic record
    k1 = {}
    for d in sorted(glob.glob(os.path.join(src, "k1_*"))):
        if os.path.isdir(d):
            m, us = counter_means(d, "k_step_reg_lm")
            k1.update(m)
            if us:
                k1.setdefault("kernel_us_under_pmc", {})[os.path.basename(d)] = us
    if "FETCH_SIZE" in k1 and "WRITE_SIZE" in k1:
        raw["k_step_reg_lm@1"] = k1
        json.dump(raw, open(os.path.join(out, f"{tag}_pmc_raw.json"), "w"), indent=1)
        fetch, write = 2.0 * k1["FETCH_SIZE"] * 1024.0, k1["WRITE_SIZE"] * 1024.0
        rec = {"envs": ENVS, "steps_per_launch": 1, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
               "source": f"profiles/{tag}_pmc_raw.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 gfx950 correction, KiB units)"}
        if "SQ_INSTS_VALU" in k1 and k1.get("SQ_WAVES", 0) > 0:
            rec["valu_instr_per_env_step"] = k1["SQ_INSTS_VALU"] / k1["SQ_WAVES"]
            rec["wave_quad_cycles_per_env_step"] = k1.get("SQ_WAVE_CYCLES", 0.0) / k1["SQ_WAVES"]
        traffic["k_step_reg_lm"] = rec
        json.dump(traffic, open(tp, "w"), indent=1)
    # ---- per-kernel averages of the other configurations (rocprofv3 --kernel-trace --stats of scripts/prof_shared.py / bench_configs.py)
    other = [f"# rocprofv3 --kernel-trace --stats of the other configurations ({tag}, 1 x MI355X)", ""]
    titles = {"prof_fourier_none": "C4 share: 131 072 MountainCar envs, shared W, single rank (`scripts/prof_shared.py fourier none`)",
              "prof_fourier_rccl": "C4 share with an RCCL communicator of size 1: finalize -> ncclAllReduce -> apply (`... fourier rccl`); RCCL serves a 1-rank "
                                   "all-reduce with a device copy (`__amd_rocclr_copyBuffer`), it launches no collective kernel",
              "prof_fourier_peer": "C4 share with the peer-write exchange, group of size 1 (`... fourier peer`)",
              "prof_tile_none": "C3: 262 144 CartPole envs, SARSA, tiles 8 x 8^4, shared W (`scripts/prof_shared.py tile none`)",
              "prof_configs": "C5/2 (bf16) and C5' (f32) Acrobot Fourier(7) wave family, L1 SARSA(lambda), C3' per-learner tile tables (`scripts/bench_configs.py`)"}
    for key, title in titles.items():
        f = glob.glob(os.path.join(src, key, "*kernel_stats.csv"))
        if not f:
            continue
        other += [f"## {title}", "", "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
        for row in list(csv.DictReader(open(f[0])))[:9]:
            full = row["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            nm = (full.split("(rsrl::")[0] if "<" in full else full.split("(")[0]).strip()
            other.append(f"| `{nm}` | {row['Calls']} | {float(row['AverageNs']) / 1e3:.2f} | {float(row['TotalDurationNs']) / 1e6:.2f} | {row['Percentage']} |")
        other.append("")
    for name in ("shared_walls.txt", "bench_configs.jsonl", "scale_n.jsonl"):
        fp = os.path.join(src, name)
        if os.path.exists(fp):
            other += [f"## {name} (wall clock, no profiler)", "", "```"] + [ln.rstrip()[:400] for ln in open(fp) if ln.strip() and not ln.startswith(("RCCL", "HIP ver", "ROCm", "Hostname", "Librccl"))] + ["```", ""]
    open(os.path.join(out, f"{tag}_kernel_stats_other.md"), "w").write("\n".join(other) + "\n")
    print(json.dumps({"traffic": traffic, "raw_keys": {k: sorted(v)[:20] for k, v in raw.items()}}, indent=1)[:2500])


if __name__ == "__main__":
    main()
