"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol
include/rsrl_hip.h declares (no compute calls without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def abi():
    import __graft_entry__ as g
    g.build()
    from rsrl_amd import _abi
    return _abi


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "rsrl_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rsrl_hip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(abi):
    L = abi.lib()
    names = header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/rsrl_hip.h but not exported"
    assert sorted(abi.SYMBOLS) == names, "python binding table and header drifted apart"
    assert L.rsrl_hip_abi_version() == 9


def test_config_struct_layout_and_defaults(abi):
    cfg = abi.Config()
    assert abi.lib().rsrl_hip_config_init(C.byref(cfg)) == 0
    assert cfg.struct_size == C.sizeof(abi.Config)          # README example defaults (q_learning.rs:19-32)
    assert (cfg.domain, cfg.basis, cfg.order, cfg.algo, cfg.policy) == (0, 0, 5, 0, 0)
    assert (cfg.gamma, cfg.lr) == (0.9, 0.001)


def test_no_cpu_fallback_without_device(abi):
    import rsrl_amd
    try:
        import subprocess
        has_gpu = subprocess.run(["/opt/rocm/bin/rocminfo"], capture_output=True, text=True).stdout.count("gfx950") > 0
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(rsrl_amd.RsrlHipError) as ei:
        rsrl_amd.Context(n_envs=4)
    assert ei.value.code == -2            # RSRL_HIP_EHIP: no device -> loud failure, never an eager fallback


def test_bad_struct_size_rejected(abi):
    cfg = abi.Config()
    abi.lib().rsrl_hip_config_init(C.byref(cfg))
    cfg.struct_size = 12
    h = C.c_void_p()
    assert abi.lib().rsrl_hip_create(C.byref(cfg), C.byref(h)) == -1
    assert b"struct_size" in abi.lib().rsrl_hip_last_error()


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: nothing under rsrl_amd/ may import, include, link or load it
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b|#\s*include\s*[\"<][^\">]*oracle|liboracle|oracle\.py|"
                     r"oracle\.(lib|build)\(|dlopen[^\n]*oracle|-loracle)", re.M)
    for dp, _, fns in os.walk(os.path.join(ROOT, "rsrl_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert not bad.search(txt), f"{fn} uses the oracle"
    import subprocess
    out = subprocess.run(["ldd", os.path.join(ROOT, "rsrl_amd", "lib", "librsrl_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_cpp_host_mirror_compiles_against_the_abi(abi, tmp_path):
    # rsrl_amd/host/rsrl.hpp + the C++ counterpart of rsrl/examples/q_learning.rs build and link (no GPU needed)
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    lib_dir = os.path.join(ROOT, "rsrl_amd", "lib")
    for name in ("q_learning", "sarsa_lambda", "greedy_gq", "pal", "q_sigma"):       # the reference's examples of the same names
        exe = tmp_path / name
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "examples", name + ".cpp"),
                               "-L" + lib_dir, "-lrsrl_hip", "-L/opt/rocm/lib", "-Wl,-rpath," + lib_dir,
                               "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
        assert exe.exists()


def test_bench_cpu_baseline_leg_runs_without_a_gpu():
    # bench.py's cpu_baseline object (the oracle timed on the host cores: reference call pattern + optimised loop)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = bench.cpu_baseline(0.4, 0.3)
    assert out["unit"] == "env-steps/s" and out["kind"] == "port" and out["cores"] >= 1
    assert out["value"] > 0 and out["optimised"]["value"] > out["value"]           # fewer projections, no heap traffic
    assert "sample" in out and abs(out["per_core"] * out["cores"] - out["value"]) < 1e-6 * out["value"]
    assert bench.BYTES_PER_ENV_STEP == 2 * 2 * 4 + 8 + 8 + 36 * 3 * 4 + 36 * 4   # 608 B per env-step


def test_asm_filter_removes_only_the_packed_fp32_wait_states():
    # rsrl_amd/_asmfilter.py: `s_nop 0` goes only between a packed fp32 instruction and an ordinary VALU consumer of its result
    from rsrl_amd import _asmfilter
    src = "\n".join([
        "k:",
        "\tv_pk_fma_f32 v[10:11], v[10:11], v[4:5], s[2:3] op_sel_hi:[1,1,0]",
        "\ts_nop 0",                                               # 1: consumer reads v[10:11] -> removed
        "\tv_pk_fma_f32 v[10:11], v[10:11], v[4:5], s[6:7] op_sel_hi:[1,1,0]",
        "\ts_nop 0",                                               # 2: consumer reads v11 as a single register -> removed
        "\tv_mul_f32_e32 v1, v11, v2",
        "\tv_pk_mul_f32 v[20:21], v[4:5], v[6:7]",
        "\ts_nop 0",                                               # 3: next instruction does not read v[20:21] -> stays
        "\tv_add_f32_e32 v3, v1, v2",
        "\tv_pk_add_f32 v[30:31], v[4:5], v[6:7]",
        "\ts_nop 1",                                               # 4: two wait states: some other hazard -> stays
        "\tv_add_f32_e32 v3, v30, v2",
        "\tv_pk_add_f32 v[30:31], v[4:5], v[6:7]",
        "\ts_nop 0",                                               # 5: v_readlane is not an ordinary consumer -> stays
        "\tv_readlane_b32 s0, v30, 3",
        "\tv_cmp_gt_f32_e64 s[0:1], v1, v2",
        "\ts_nop 0",                                               # 6: not behind a packed instruction -> stays
        "\tv_cndmask_b32_e64 v1, v2, v3, s[0:1]",
        "\tv_pk_mul_f32 v[40:41], v[4:5], v[6:7]",
        "\ts_nop 0",                                               # 7: a label before the consumer -> stays
        ".LBB0_1:",
        "\tv_add_f32_e32 v3, v40, v2",
        "\tv_pk_mul_f32 v[50:51], v[4:5], v[6:7]",
        "\ts_nop 0",                                               # 8: the destination only WRITTEN by the next instruction -> stays
        "\tv_mov_b32_e32 v50, v2",
        "\tv_pk_fma_f32 v[60:61], v[4:5], v[6:7], v[8:9]",
        "\t; a comment line",
        "\ts_nop 0",                                               # 9: comments in between do not hide the producer -> removed
        "\tv_pk_mul_f32 v[62:63], v[0:1], v[60:61]",
        "\tv_pk_mul_f32 v[70:71], v[4:5], v[6:7]",
        "\t;;#ASMSTART",
        "\tv_pk_mov_b32 v[72:73], v[0:1], v[2:3] op_sel:[1,1]",     # inline asm: the recogniser does not count it ...
        "\t;;#ASMEND",
        "\ts_nop 0",                                              # 10: ... so this one is redundant a fortiori -> removed
        "\tv_pk_fma_f32 v[74:75], v[70:71], v[4:5], v[6:7]",
        "\tv_cmp_gt_f32_e32 vcc, v1, v2",
        "\t;;#ASMSTART",
        "\tv_ashrrev_i32_e32 v80, 31, v3",
        "\t;;#ASMEND",
        "\ts_nop 0",                                              # 11: not a packed producer -> stays
        "\tv_cndmask_b32_e32 v1, v2, v3, vcc",
    ])
    out, removed = _asmfilter.filter_asm(src)
    assert removed == 4
    kept = [ln for ln in out.split("\n")]
    assert len(kept) == len(src.split("\n")) - 4
    assert out.count("s_nop 0") == 6 and out.count("s_nop 1") == 1
    # every instruction but the removed wait states is still there, in order
    assert [ln for ln in src.split("\n") if ln.strip() != "s_nop 0"] == [ln for ln in out.split("\n") if ln.strip() != "s_nop 0"]
    assert _asmfilter.filter_asm(out) == (out, 0)                  # idempotent


def test_asm_filter_keeps_a_wait_state_that_may_belong_to_an_earlier_instruction():
    # one wait state can be the LAST slot of a longer hazard counted from an instruction before the packed producer (a wide store whose data
    # registers the consumer overwrites, a VALU that wrote an SGPR / VCC, a trans op, an MFMA): within four slots of such an instruction nothing is removed
    from rsrl_amd import _asmfilter

    def case(before, n_between=0):
        src = "\n".join(["k:"] + ["\t" + b for b in before] + ["\tv_mul_f32_e32 v9, v8, v8"] * n_between +
                        ["\tv_pk_fma_f32 v[10:11], v[0:1], v[2:3], v[4:5]", "\ts_nop 0", "\tv_mul_f32_e32 v0, v10, v6"])
        return _asmfilter.filter_asm(src)[1]
    assert case([]) == 1 and case(["v_fma_f32 v20, v21, v22, v23", "global_store_dwordx2 v30, v[0:1], s[0:1]"]) == 1
    for hazard in ("global_store_dwordx4 v30, v[0:3], s[0:1]", "buffer_store_dwordx3 v[0:2], v30, s[0:3], 0 offen", "flat_store_dwordx4 v[30:31], v[0:3]",
                   "v_cmp_lt_f32_e32 vcc, v1, v2", "v_cmp_gt_f32_e64 s[2:3], v1, v2", "v_readfirstlane_b32 s4, v1", "v_exp_f32_e32 v1, v2", "v_rcp_f32_e32 v1, v2",
                   "v_mad_u64_u32 v[12:13], s[4:5], v2, v3, v[14:15]", "v_mfma_f32_4x4x1_16b_f32 v[0:3], v4, v5, v[0:3]", "s_and_saveexec_b64 s[0:1], vcc",
                   "s_mov_b64 exec, s[0:1]", "v_add_co_u32_e32 v1, vcc, v2, v3", "v_div_scale_f32 v1, vcc, v2, v3, v4"):
        for n in range(_asmfilter.LOOKBACK):
            assert case([hazard], n) == 0, (hazard, n)
        assert case([hazard], _asmfilter.LOOKBACK) == 1, hazard             # ... and beyond the window the rule applies again
    # an inline-asm instruction counts as a possible source too
    src = "\n".join(["k:", "\t;;#ASMSTART", "\tv_cmp_lt_f32_e32 vcc, v1, v2", "\t;;#ASMEND", "\tv_pk_mul_f32 v[10:11], v[0:1], v[2:3]", "\ts_nop 0", "\tv_add_f32_e32 v0, v10, v6"])
    assert _asmfilter.filter_asm(src)[1] == 0


def test_asm_filter_scope_and_counts():
    # the pass runs on the translation units where it pays and only under the validated compiler; the per-unit removal counts of the shipped library
    # are pinned, so a toolchain or source change that moves them is noticed (and re-validated on the GPU) rather than shipped silently
    import json
    from rsrl_amd import _build
    assert set(_build.NOP_FILTER_SOURCES) == {"train_reg_d0a.hip", "train_reg_d0b.hip", "train_reg_d1.hip", "train_reg_d2.hip"} | set(_build.ABI_UNITS)
    assert _build.nop_filter_applies("/x/train_reg_d0b.hip") == _build.nop_filter_enabled()
    assert not _build.nop_filter_applies("/x/train_td.hip") and not _build.nop_filter_applies("/x/train_gq.hip")
    if _build.hipcc_version() and _build.NOP_FILTER_VALIDATED_HIPCC in _build.hipcc_version() and os.environ.get("RSRL_NOP_FILTER", "1") != "0":
        assert _build.nop_filter_enabled()
    counts = json.load(open(_build.NOP_COUNTS_PATH))
    assert set(counts) == {os.path.basename(s) for s in _build.sources()}
    for name, n in counts.items():
        assert (n is not None) == (name in _build.NOP_FILTER_SOURCES and _build.nop_filter_enabled()), name
    if _build.nop_filter_enabled():
        assert {k: v for k, v in counts.items() if v is not None} == EXPECTED_NOP_COUNTS, counts


def test_asm_filter_fails_closed_on_an_unknown_compiler(monkeypatch):
    from rsrl_amd import _build
    monkeypatch.setattr(_build, "_HIPCC_VERSION", "AMD clang version 23.0.0git (roc-8.0.0 1 deadbeef)")
    monkeypatch.setattr(_build, "_warned_version", True)
    assert not _build.nop_filter_enabled() and not _build.nop_filter_applies("/x/train_reg_d0b.hip")


# wait states removed per translation unit in the shipped build (rsrl_amd/lib/librsrl_hip.nop_filter.json, written by _build); re-validate on the GPU
# (tests -m gpu, scripts/ab_bits.py) before changing these
EXPECTED_NOP_COUNTS = {"abi_ctx.hip": 0, "abi_group.hip": 0, "abi_train.hip": 1144, "abi_trait.hip": 59, "abi_weights.hip": 0, "kernels_util.hip": 0, "launch_shared.hip": 88, "train_reg_d0a.hip": 528, "train_reg_d0b.hip": 315, "train_reg_d1.hip": 20, "train_reg_d2.hip": 21}


def test_campaign_scripts_compile():
    # tests/fuzz_*.py run on the GPU box only; here: they parse
    import glob
    import py_compile
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_*.py")))
    assert len(files) >= 4
    for f in files:
        py_compile.compile(f, doraise=True)


def test_rust_binding_block_is_the_header(abi):
    """INTEGRATION.md's `rsrl-hip-sys` block (it declared 45 of the 65 exports and claimed to mirror the header one to one): generated from the
    header by scripts/gen_rust_sys.py, and held here to (1) the header in the tree, (2) the symbols the shared library exports -- every one declared, nothing
    else --, (3) the arity of the ctypes binding every test drives, (4) the layout of the two structs."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("gen_rust_sys", os.path.join(ROOT, "scripts", "gen_rust_sys.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    block = gen.doc_block()
    assert block == gen.generate(), "INTEGRATION.md's Rust block is stale: python scripts/gen_rust_sys.py --write"
    decl = dict(gen.declared_functions(block))
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "rsrl_amd", "lib", "librsrl_hip.so")], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\b(rsrl_hip_[a-z0-9_]+)$", nm, flags=re.M)))
    assert sorted(decl) == exported == header_symbols()
    for name, (_, args) in abi.SYMBOLS.items():
        assert decl[name] == len(args), f"{name}: {decl[name]} parameters in the Rust block, {len(args)} in rsrl_amd/_abi.py"
    # struct fields: same names, same order, same widths as the ctypes mirror of the C struct
    width = {"u32": 4, "i32": 4, "i64": 8, "u64": 8, "f64": 8, "*mut c_void": 8}
    for sname, ct in (("rsrl_hip_config", abi.Config), ("rsrl_hip_stats", abi.Stats)):
        m = re.search(r"pub struct %s \{(.*?)\n\}" % sname, block, flags=re.S)
        fields = re.findall(r"pub (\w+): ([^,]+),", m.group(1))
        names = [{"lambda": "lam"}.get(n, n) for n, _ in fields]
        assert names == [f[0] for f in ct._fields_], sname
        assert [width[t.strip()] for _, t in fields] == [C.sizeof(f[1]) for f in ct._fields_], sname
