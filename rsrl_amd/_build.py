"""Build librsrl_hip.so for gfx950 in-tree (rsrl_amd/lib/).  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "librsrl_hip.so")

# -ffp-contract=off: fused multiply-adds are written out as fmaf in the sources, nothing else is fused,
# so the fp32 op order is explicit (bit-exact tile indices; tight parity with the f32 oracle).
# -fno-slp-vectorize: SLP packing of adjacent f32 FMAs into v_pk_fma_f32 costs ~250 v_mov per step to build
# register pairs and pushes the fused kernel into scratch; the W-sized loops are packed BY HAND instead (f2 storage in
# kernels_reg.hpp: no shuffles, +14 % on the fused kernel at one wave per SIMD).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
               "-shared", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


# per-source flags: the fused register-family loops run at ONE wave per SIMD (65 536 learners = 1024 waves), where
# single-wave ILP is all there is -- LLVM's max-ILP scheduling strategy is worth +2..3 % there (A/B on MI355X, round 1)
PER_SOURCE_FLAGS = {name: ["-mllvm", "-amdgpu-sched-strategy=" + os.environ.get("RSRL_SCHED_STRATEGY", "max-ilp")]
                    for name in ("train_reg_d0a.hip", "train_reg_d0b.hip", "train_reg_d1.hip", "train_reg_d2.hip")}


ABI_UNITS = ("abi_ctx.hip", "abi_trait.hip", "abi_weights.hip", "abi_train.hip", "abi_group.hip", "kernels_util.hip", "launch_shared.hip")      # the C ABI's translation units (ctx.hpp; one file until round 6)
# the ABI units (generic / shared-W kernels): the runtime dispatch over agents and policies leaves a few 3-float arrays
# (Q(s,.), Q(s',.)) as allocas; promoted to LDS next to the 37 KiB reduction tile of k_shared_ca they made the kernel 2-4x
# slower (measured 7.7 us with the arrays in scratch, 21-34 us promoted) -- keep them out of LDS in these translation units.
for _u in ABI_UNITS:
    PER_SOURCE_FLAGS[_u] = ["-mllvm", "-disable-promote-alloca-to-lds"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0)")


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]


def deps():
    inc = os.path.join(os.path.dirname(HERE), "include", "rsrl_hip.h")
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [inc, os.path.abspath(__file__), os.path.join(HERE, "_asmfilter.py")]


STAMP_PATH = LIB_PATH + ".sha256"


def source_digest(extra_flags=()):
    """sha256 over everything the library is built from: every file under csrc/, the ABI header, the flags (per source too)
    and hipcc's version -- content, not mtimes: a snapshot copied to another box (gpurun) keeps its digest."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(os.path.dirname(HERE), "include", "rsrl_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    h.update(repr((HIPCC_FLAGS, sorted(PER_SOURCE_FLAGS.items()), list(extra_flags), nop_filter_enabled(), NOP_FILTER_SOURCES)).encode())
    h.update(open(os.path.join(HERE, "_asmfilter.py"), "rb").read())
    h.update(hipcc_version().encode())
    return h.hexdigest()


_HIPCC_VERSION = None


def hipcc_version():
    """`hipcc --version` (cached): a compiler upgrade makes the stamped library stale.  A box without hipcc (it only runs the prebuilt
    library) contributes the empty string -- there is_stale() is never asked to rebuild anyway."""
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        try:
            _HIPCC_VERSION = subprocess.run([hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout.strip()
        except Exception:
            _HIPCC_VERSION = ""
    return _HIPCC_VERSION


def is_stale():
    """True unless librsrl_hip.so exists AND was built from exactly the sources in the tree (digest stamped next to it)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    return open(STAMP_PATH).read().strip() != source_digest()


# The assembly post-pass is applied ONLY where its gain was measured (k_train_reg +1.0-1.3 %, k_train_wave +6 %, k_shared_persist
# ~+10 %; the last two are launched from abi_train.hip -- the ABI's units keep the pass they were validated with as one file) and ONLY under the compiler whose hazard recogniser it was validated against: any other
# `hipcc --version` compiles every source in one plain hipcc call (fail closed -- a new recogniser may place wait states for other reasons).
NOP_FILTER_SOURCES = ("train_reg_d0a.hip", "train_reg_d0b.hip", "train_reg_d1.hip", "train_reg_d2.hip") + ABI_UNITS
NOP_FILTER_VALIDATED_HIPCC = "roc-7.2.0 26014 7b800a19466229b8479a78de19143dc33c3ab9b5"     # substring of `hipcc --version` (AMD clang 22.0.0git)
NOP_COUNTS_PATH = LIB_PATH.replace(".so", ".nop_filter.json")
_warned_version = False


def nop_filter_enabled():
    """the assembly post-pass of _asmfilter.py (drops the compiler's false-positive wait states behind packed fp32 instructions);
    RSRL_NOP_FILTER=0, or a compiler other than the validated one, compiles every source in one hipcc call instead"""
    global _warned_version
    if os.environ.get("RSRL_NOP_FILTER", "1") == "0":
        return False
    if NOP_FILTER_VALIDATED_HIPCC not in hipcc_version():
        if not _warned_version:
            import sys
            print("rsrl_amd._build: this hipcc is not the one the assembly post-pass was validated with (" + NOP_FILTER_VALIDATED_HIPCC +
                  "): building WITHOUT it", file=sys.stderr)
            _warned_version = True
        return False
    return True


def nop_filter_applies(src):
    return nop_filter_enabled() and os.path.basename(src) in NOP_FILTER_SOURCES


def _tool(name):
    return subprocess.run([hipcc(), "-print-prog-name=" + name], capture_output=True, text=True, check=True).stdout.strip()


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)


def _compile_one(args):
    src, obj, verbose, extra = args
    flags = [f for f in HIPCC_FLAGS if f != "-shared" and not (f == "-fno-slp-vectorize" and "-fslp-vectorize" in extra)]
    per_src = [] if any("amdgpu-sched-strategy" in e for e in extra) else PER_SOURCE_FLAGS.get(os.path.basename(src), [])
    base = [hipcc()] + flags + per_src + list(extra)
    if not nop_filter_applies(src):
        _run(base + ["-c", src, "-o", obj], verbose)
        return obj, None
    # hipcc's own steps, taken apart so that the device assembly can be filtered in between:
    #   device: .hip -> .s (clang) -> filtered .s -> .o (assembler) -> code object (lld) -> fat binary (clang-offload-bundler)
    #   host  : .hip -> .o with the fat binary embedded
    try:
        return _compile_filtered(base, src, obj, verbose)
    except (subprocess.CalledProcessError, OSError) as e:
        # a toolchain laid out differently (no clang / lld / clang-offload-bundler next to hipcc): the plain compile is the same program,
        # minus the post-pass -- say so and carry on
        import sys
        print(f"rsrl_amd._build: assembly post-pass unavailable for {os.path.basename(src)} ({e}); compiling it in one hipcc call", file=sys.stderr)
        _run(base + ["-c", src, "-o", obj], verbose)
        return obj, None


def _compile_filtered(base, src, obj, verbose):
    from . import _asmfilter
    stem = obj[:-2] if obj.endswith(".o") else obj
    asm, dev_o, code, fatbin = stem + ".gfx950.s", stem + ".gfx950.o", stem + ".gfx950.co", stem + ".hipfb"
    _run(base + ["-Wno-unused-command-line-argument", "--cuda-device-only", "-S", src, "-o", asm], verbose)
    text, removed = _asmfilter.filter_asm(open(asm).read())
    with open(asm, "w") as f:
        f.write(text)
    if verbose:
        print(f"# {os.path.basename(src)}: {removed} wait states behind packed fp32 instructions removed")
    _run([_tool("clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm, "-o", dev_o], verbose)
    _run([_tool("lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", dev_o, "-o", code], verbose)
    _run([_tool("clang-offload-bundler"), "-type=o", "-bundle-align=4096",
          "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + code,
          "-output=" + fatbin], verbose)
    _run(base + ["-Wno-unused-command-line-argument", "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fatbin,
                 "-c", src, "-o", obj], verbose)
    for tmp in (asm, dev_o, code, fatbin):
        os.remove(tmp)
    return obj, removed


def build(force=False, verbose=False, out=None, extra_flags=()):
    """Compile every HIP source (in parallel) and link rsrl_amd/lib/librsrl_hip.so.
    out / extra_flags build an A/B variant (e.g. -DRSRL_DOT_SPLIT=2) next to the product library."""
    if out is not None:
        return _build_to(out, list(extra_flags), verbose)
    if not force and not is_stale():
        return LIB_PATH
    return _build_to(LIB_PATH, [], verbose)


def _build_to(lib_path, extra_flags, verbose):
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj_" + os.path.basename(lib_path).replace(".so", ""))
    os.makedirs(obj_dir, exist_ok=True)
    jobs = [(s, os.path.join(obj_dir, os.path.basename(s) + ".o"), verbose, extra_flags) for s in sources()]
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        results = list(ex.map(_compile_one, jobs))
    objs = [r[0] for r in results]
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-L/opt/rocm/lib", "-lrccl", "-o", lib_path]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    if os.path.abspath(lib_path) == os.path.abspath(LIB_PATH):
        with open(STAMP_PATH, "w") as f:
            f.write(source_digest() + "\n")
        # how many wait states the post-pass removed per translation unit (null = compiled in one plain hipcc call): tests/test_abi_cpu.py
        # holds the expected counts, so a toolchain or source change that moves them is noticed
        import json
        with open(NOP_COUNTS_PATH, "w") as f:
            json.dump({os.path.basename(j[0]): r[1] for j, r in zip(jobs, results)}, f, indent=1, sort_keys=True)
            f.write("\n")
    else:
        import shutil
        shutil.rmtree(obj_dir, ignore_errors=True)      # an A/B variant's objects are of no further use (16 MB each in every gpurun snapshot)
    return lib_path


PK_FORWARD_SRC = os.path.join(os.path.dirname(HERE), "scripts", "ubench", "pk_forward.hip")
PK_FORWARD_BIN = PK_FORWARD_SRC[:-4]


def build_pk_forward(force=False):
    """scripts/ubench/pk_forward: the micro-test behind the assembly post-pass (does the hardware interlock a packed-fp32 producer and its
    next-slot consumer?), run by tests/test_gpu_round5.py.  Built in-tree so that it travels to the GPU box."""
    if not force and os.path.exists(PK_FORWARD_BIN) and os.path.getmtime(PK_FORWARD_BIN) >= os.path.getmtime(PK_FORWARD_SRC):
        return PK_FORWARD_BIN
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-Wno-unused-command-line-argument", "-o", PK_FORWARD_BIN, PK_FORWARD_SRC])
    return PK_FORWARD_BIN


RCCL_STUB_SRC = os.path.join(os.path.dirname(HERE), "tests", "stubs", "rccl_stub.cpp")
RCCL_STUB_LIB = os.path.join(os.path.dirname(RCCL_STUB_SRC), "librccl_stub.so")


def build_rccl_stub(force=False):
    """tests/stubs/librccl_stub.so: the TEST DOUBLE of the RCCL entry points (tests/test_gpu_rccl_stub.py LD_PRELOADs it to run the N > 1 RCCL
    path on one device).  Host-only C++; built in-tree so that it travels to the GPU box.  Never loaded by the product."""
    if not force and os.path.exists(RCCL_STUB_LIB) and os.path.getmtime(RCCL_STUB_LIB) >= os.path.getmtime(RCCL_STUB_SRC):
        return RCCL_STUB_LIB
    rocm = os.path.dirname(os.path.dirname(os.path.realpath(hipcc())))
    subprocess.check_call(["g++", "-shared", "-fPIC", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), RCCL_STUB_SRC,
                           "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", RCCL_STUB_LIB])
    return RCCL_STUB_LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
