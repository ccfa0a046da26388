#!/usr/bin/env python3
"""Turn the raw output of scripts/gpu_profile_round.sh (gpurun_out/round/) into the tracked files under profiles/:

    profiles/<tag>_kernel_stats.md   rocprofv3 --kernel-trace --stats of the default bench command (per-kernel averages)
    profiles/<tag>_pmc_raw.json      per-launch means of every PMC counter for k_step_reg / k_train_reg
    profiles/<tag>_bench.json        the bench line of the same visit
    profiles/pmc_traffic.json        what bench.py reads for roofline.traffic and valu_roofline

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of a streaming read => fetched bytes = 2 * FETCH_SIZE * 1024.

    python scripts/summarize_profile.py r01b
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "round")
OUT = os.path.join(ROOT, "profiles")
ENVS = 65536


def counter_means(d, kernel_sub, skip_first=2):
    """mean over dispatches (after the first `skip_first`) of each counter's per-dispatch SUM over its instances"""
    f = glob.glob(os.path.join(SRC, d, "*counter_collection.csv"))
    if not f:
        return {}, 0
    per = defaultdict(lambda: defaultdict(float))
    dur = {}
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
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(OUT, exist_ok=True)
    # ---- kernel stats of the bench command
    ks = glob.glob(os.path.join(SRC, "stats", "*kernel_stats.csv"))
    lines = [f"# rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline` ({tag}, 1 x MI355X)", "",
             "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    stats = {}
    if ks:
        for row in csv.DictReader(open(ks[0])):
            full = row["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            nm = (full.split("(rsrl::")[0] if "<" in full else full.split("(")[0]).strip()
            lines.append(f"| `{nm}` | {row['Calls']} | {float(row['AverageNs']) / 1e3:.2f} | {float(row['TotalDurationNs']) / 1e6:.2f} | {row['Percentage']} |")
            stats[nm] = float(row["AverageNs"]) / 1e3
    bench = None
    bp = os.path.join(SRC, "bench.json")
    if os.path.exists(bp):
        for ln in open(bp):
            if ln.startswith("{"):
                bench = json.loads(ln)
    if bench:
        r = bench["roofline"]
        rs = bench.get("roofline_streaming") or {}
        lines += ["", f"bench.py (same visit, no profiler): value {bench['value']:.4g} env-steps/s, `{r['kernel']}` {r['avg_launch_ms'] * 1e3:.2f} us per launch by HIP events "
                      f"({r['launches']} launches); roofline_streaming `{rs.get('kernel', 'k_step_reg')}` {rs.get('avg_launch_ms', 0) * 1e3:.2f} us per batch-step inside the 32-step graph."]
        json.dump(bench, open(os.path.join(OUT, f"{tag}_bench.json"), "w"), indent=1)
    open(os.path.join(OUT, f"{tag}_kernel_stats.md"), "w").write("\n".join(lines) + "\n")
    # ---- PMC passes
    raw = {}
    for key, sub, spl in (("k1", "k_step_reg_lm", 1), ("fu", "k_train_reg", 256)):
        rec = {}
        for i in (1, 2, 3, 4):
            m, us = counter_means(f"{key}_{i}", sub)
            rec.update(m)
            if us:
                rec.setdefault("kernel_us_under_pmc", {})[f"pass{i}"] = us
        raw[sub] = rec
    json.dump(raw, open(os.path.join(OUT, f"{tag}_pmc_raw.json"), "w"), indent=1)
    traffic = {}
    for sub, spl in (("k_step_reg_lm", 1), ("k_train_reg", 256)):
        r = raw.get(sub, {})
        if "FETCH_SIZE" not in r or "WRITE_SIZE" not in r:
            continue
        fetch, write = 2.0 * r["FETCH_SIZE"] * 1024.0, r["WRITE_SIZE"] * 1024.0
        rec = {"envs": ENVS, "steps_per_launch": spl, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
               "traffic_bytes_per_launch": fetch + write,
               "source": f"profiles/{tag}_pmc_raw.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 gfx950 correction, KiB units)"}
        if "SQ_INSTS_VALU" in r and "SQ_WAVES" in r and r["SQ_WAVES"] > 0:
            rec["valu_instr_per_env_step"] = r["SQ_INSTS_VALU"] / r["SQ_WAVES"] / spl
            rec["wave_quad_cycles_per_env_step"] = r.get("SQ_WAVE_CYCLES", 0.0) / r["SQ_WAVES"] / spl
        traffic[sub] = rec
    json.dump(traffic, open(os.path.join(OUT, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({"stats": stats, "traffic": traffic}, indent=1))


if __name__ == "__main__":
    main()
