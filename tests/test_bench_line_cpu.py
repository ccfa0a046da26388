"""bench.py's contract with the driver: the LAST stdout line is ONE JSON object the driver can parse.  Round 5's line had grown to ~24 KB (per-rank records,
per-configuration parity, ceilings, digests) and BENCH_r05.json recorded `parsed: null` -- the round's headline went unvalidated.  The line is
now built by bench.compact_line from the full result (which goes to bench_detail.json and stderr); this test holds its size, shape and position."""
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def canned():
    """the full result of a real run (the builder's stamped copy of round 5's 25 KB line) + every leg added since, with oversized free text everywhere"""
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_driver.json")))
    trait = {"learners": 65536, "steps": 400, "us_per_batch_step": 8.150820212904364, "value": 8040417809.270719, "unit": "env-steps/s", "calls_per_step": 4,
             "algorithmic_bytes_per_env_step": 679, "kernel": "k_trait_lm<step>", "hip_event_us_per_batch_step": 10.9, "bound": "hbm", "peak": 8000.0,
             "achieved": 5459.4, "frac": 0.6824304615618523, "what": "x" * 900, "host_us_per_call": {"domain_step": 1.17, "handle": 0.8}}
    full["trait_loop"] = trait
    full["trait_loop_unfused"] = dict(trait, us_per_batch_step=18.4, frac=0.30, kernel="k_trait_lm<handle>")
    full["trait_loop_1m"] = dict(trait, learners=1048576, us_per_batch_step=165.9, frac=0.536)
    lam = {"workload": "w" * 300, "value": 4.8e8, "unit": "env-steps/s", "us_per_batch_step": 135.7, "kernel_us_per_batch_step": 133.6,
           "roofline": {"bound": "hbm", "kernel": "k_sparse_trace_scatter", "achieved": 6092.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.7615, "what": "v" * 600}}
    full["lambda_shared_tiles"] = lam                                     # round 6's legs: the rebuilt eligibility-trace kernels, C3's written floor
    full["lambda_generic_order"] = dict(lam, value=4.3e8, us_per_batch_step=152.3, roofline=dict(lam["roofline"], kernel="k_train_lambda_mem", frac=0.441))
    if isinstance(full.get("c3_shared_tiles"), dict):
        full["c3_shared_tiles"].update({"floor_us": 13.6, "floor": "f" * 700, "floor_frac": 0.62})
    full["hbm_copy_measured"] = {"GBps": 5100.0, "bytes": 2 ** 30, "reps": 10, "what": "y" * 500}
    full["greedy_rollout"] = {"limit": 1000, "compared": 256, "terminated_frac": 1.0, "identical_n_states_frac": 0.996, "min_argmax_margin": 3.1e-9,
                              "max_min_margin_of_differing": 3.1e-9, "note": "z" * 700}
    full["per_rank"] = full.get("per_rank", []) * 8
    full["config"]["timed"] = "t" * 400
    return full


def test_compact_line_is_small_parsable_and_complete():
    import bench
    full = canned()
    assert len(json.dumps(full)) > 20000                                  # the input IS the kind of record that broke round 5
    line = bench.compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= bench.COMPACT_LIMIT < 8192, len(text)
    back = json.loads(text)
    assert "\n" not in text and back == json.loads(json.dumps(line))
    for k in CONTRACT:
        assert k in back, k
    assert back["value"] == float(f"{full['value']:.7g}") and back["n_gpus"] == 1 and back["steps"] == 20 and back["warmup"] == 5
    assert abs(back["ms_per_step"] - full["ms_per_step"]) <= 1e-6 * full["ms_per_step"]
    assert back["config"]["workload"].startswith("65536 vectorised MountainCar") and "model" not in back["config"]
    rl = back["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches", "profile_digest_matches"):
        assert k in rl, k
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    cb = back["cpu_baseline"]
    assert set(cb) >= {"value", "unit", "cores", "kind", "sample"} and cb["kind"] == "port" and cb["cores"] >= 1
    legs = back["legs"]
    for k in ("trait_loop", "trait_loop_unfused", "trait_loop_1m", "streaming", "streaming_1m", "c3_shared_tiles", "c5_wave_bf16", "lambda_shared_tiles",
              "lambda_generic_order", "shared_w", "shared_w_rccl"):
        assert {"value", "frac"} <= set(legs[k]) and ("us" in legs[k]), (k, legs[k])
    assert legs["c3_shared_tiles"]["floor_us"] == 13.6 and legs["lambda_shared_tiles"]["kernel"] == "k_sparse_trace_scatter"
    assert legs["trait_loop"]["frac_of_measured_copy"] > legs["trait_loop"]["frac"]        # 5.1 TB/s measured < 8 TB/s published
    gr = back["greedy_rollout"]
    assert gr["terminated_frac"] > 0 and "min_argmax_margin" in gr and gr["compared"] == 256
    assert back["hbm_copy_measured_GBps"] == 5100.0
    assert back["detail"].endswith("bench_detail.json")


def test_emit_prints_the_compact_line_last_on_stdout(monkeypatch, tmp_path, capsys):
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "gpurun_out").mkdir()
    full = canned()
    bench.emit(full)
    cap = capsys.readouterr()
    out_lines = [ln for ln in cap.out.splitlines() if ln.strip()]
    assert len(out_lines) == 1 and len(out_lines[0]) <= bench.COMPACT_LIMIT
    line = json.loads(out_lines[-1])
    assert line["value"] > 9e10 and line["roofline"]["frac"] > 0.4
    # the detail went to the side file(s) and to stderr, complete
    for d in (tmp_path, tmp_path / "gpurun_out"):
        assert json.load(open(d / "bench_detail.json"))["per_rank"] == full["per_rank"]
    assert json.loads(cap.err.strip().splitlines()[-1])["parity"]["configs"] == full["parity"]["configs"]


def test_an_oversized_line_is_cut_down_not_printed(monkeypatch, tmp_path, capsys):
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = canned()
    for k in list(bench.LEG_KEYS):                                        # every leg a failure with a long message
        full[k] = {"error": "E" * 5000}
    monkeypatch.setattr(bench, "COMPACT_LIMIT", 2400)
    bench.emit(full)
    text = capsys.readouterr().out.strip().splitlines()[-1]
    back = json.loads(text)
    assert len(text) <= 2400 and all(k in back for k in CONTRACT) and "roofline" in back and "cpu_baseline" in back
