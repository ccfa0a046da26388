"""The roofline's inputs are tied to the binary: profiles/isa_mix.json / pmc_traffic.json carry the sha256 of the machine code
they were profiled on, and bench.py withholds every fraction derived from them when the loaded library carries other code.  CPU only."""
import json
import os
import shutil
import struct

import bench
from rsrl_amd import _build, _kdigest

KERNELS = ["k_train_reg", "k_train_wave", "k_shared_ca"]


def _code_span(blob, kernel):
    """(offset, length) inside the library image of one instantiation's machine code"""
    for m in __import__("re").finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + 24)
        pos = base + 32
        if n > 64:
            continue
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, pos)
            pos += 24 + tlen
            elf = blob[base + off:base + off + size]
            if elf[:4] != b"\x7fELF":
                continue
            for name, code in _kdigest._functions(elf):
                if f"{len(kernel)}{kernel}" in name and len(code) > 64:
                    return base + off + elf.index(code), len(code)
    raise AssertionError(kernel)


def test_every_profiled_kernel_has_a_digest():
    d = _kdigest.kernel_digests(_build.LIB_PATH, KERNELS + ["k_no_such_kernel"])
    assert all(d[k] and len(d[k]) == 64 for k in KERNELS) and d["k_no_such_kernel"] is None
    assert len({d[k] for k in KERNELS}) == len(KERNELS)


def test_one_changed_instruction_flips_only_that_kernels_digest(tmp_path):
    blob = bytearray(open(_build.LIB_PATH, "rb").read())
    before = _kdigest.kernel_digests(_build.LIB_PATH, KERNELS)
    off, n = _code_span(bytes(blob), "k_train_reg")
    blob[off + n // 2] ^= 0x01                       # one bit of one instruction of one instantiation
    p = tmp_path / "edited.so"
    p.write_bytes(bytes(blob))
    after = _kdigest.kernel_digests(str(p), KERNELS)
    assert after["k_train_reg"] != before["k_train_reg"]
    assert after["k_train_wave"] == before["k_train_wave"] and after["k_shared_ca"] == before["k_shared_ca"]


def test_bench_withholds_fractions_from_a_stale_profile(tmp_path, monkeypatch):
    # a profiles/ directory stamped with THIS library's digests -> fractions printed; the same constants against an edited library -> null
    live = _kdigest.kernel_digests(_build.LIB_PATH, ["k_train_reg"])["k_train_reg"]
    prof = tmp_path / "profiles"
    prof.mkdir()
    mix = json.load(open(os.path.join(bench.ROOT, "profiles", "isa_mix.json")))
    tr = json.load(open(os.path.join(bench.ROOT, "profiles", "pmc_traffic.json")))
    mix["k_train_reg"]["code_sha256"] = live
    for e in tr["k_train_reg"]:
        e["code_sha256"] = live
    json.dump(mix, open(prof / "isa_mix.json", "w"))
    json.dump(tr, open(prof / "pmc_traffic.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "_LIVE_DIGESTS", {})
    rl = bench.gate_on_profile(bench.valu_roofline("k_train_reg", 9.6e10), ["k_train_reg"])
    assert rl["profile_digest_matches"] is True and 0.3 < rl["useful_frac"] < rl["frac"] < 0.6
    # ... the deliberate edit: one instruction of the kernel changes, the committed constants do not
    blob = bytearray(open(_build.LIB_PATH, "rb").read())
    off, n = _code_span(bytes(blob), "k_train_reg")
    blob[off + 8] ^= 0x80
    edited = tmp_path / "librsrl_hip_edited.so"
    edited.write_bytes(bytes(blob))
    monkeypatch.setenv("RSRL_HIP_LIB", str(edited))
    monkeypatch.setattr(bench, "_LIVE_DIGESTS", {})
    rl = bench.gate_on_profile(bench.valu_roofline("k_train_reg", 9.6e10), ["k_train_reg"])
    assert rl["profile_digest_matches"] is False and rl["frac"] is None and rl["useful_frac"] is None
    assert rl["issue_slots"]["frac"] is None and "frac" in rl["from_stale_profile"]
