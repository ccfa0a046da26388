"""sha256 of a kernel family's MACHINE CODE inside librsrl_hip.so -- what ties a committed profile to the binary it was taken from.

`profiles/isa_mix.json` / `pmc_traffic.json` hold per-kernel constants (flop and HBM bytes per env-step) that bench.py multiplies by a live
kernel rate.  They are only true of the code they were profiled on, so scripts/summarize_profile.py stamps every entry with
`code_sha256 = kernel_digests(lib)[kernel]` and bench.py prints a roofline fraction only while the loaded library's digest is the same one
(`profile_digest_matches`).  The digest covers the instruction bytes of EVERY instantiation of the kernel template (all FUNC symbols of the
gfx950 code objects whose demangled-agnostic name contains the kernel's identifier, sorted by name): host-side edits, comments and the
other kernels of a translation unit do not move it; one changed instruction in any instantiation does.

Pure Python (the GPU box of a bench run needs no binutils): .hip_fatbin holds one uncompressed clang offload bundle per translation unit
(`__CLANG_OFFLOAD_BUNDLE__`, u64 count, then {u64 offset, u64 size, u64 triple length, triple}), each gfx950 entry is an ELF64 code object."""
import hashlib
import re
import struct

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob):
    """every gfx950 ELF embedded in the library image"""
    for m in re.finditer(re.escape(_MAGIC), blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + len(_MAGIC))
        pos = base + len(_MAGIC) + 8
        if n > 64:
            continue                                    # the magic inside a string table, not a bundle header
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, pos)
            triple = blob[pos + 24:pos + 24 + tlen].decode("ascii", "replace")
            pos += 24 + tlen
            if "amdgcn" in triple and size and blob[base + off:base + off + 4] == b"\x7fELF":
                yield blob[base + off:base + off + size]


def _functions(elf):
    """(name, code bytes) of every FUNC symbol of an ELF64 little-endian code object"""
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    for name_off, typ, flags, addr, off, size, link, info, align, entsize in secs:
        if typ != 2:                                    # SHT_SYMTAB
            continue
        str_off = secs[link][4]
        for k in range(size // 24):
            st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, off + k * 24)
            if (st_info & 0xf) != 2 or st_size == 0 or st_shndx == 0 or st_shndx >= shnum:      # STT_FUNC, defined
                continue
            end = elf.index(b"\0", str_off + st_name)
            sec = secs[st_shndx]
            foff = sec[4] + (st_value - sec[3])
            yield elf[str_off + st_name:end].decode("ascii", "replace"), elf[foff:foff + st_size]


def kernel_digests(lib_path, kernels):
    """{kernel identifier: sha256 hex over (name, code) of every instantiation, or None when the library holds no such kernel}"""
    blob = open(lib_path, "rb").read()
    found = {k: [] for k in kernels}
    for elf in _code_objects(blob):
        for name, code in _functions(elf):
            for k in kernels:
                # the identifier as a whole word of the mangled name: <length><identifier> (Itanium), e.g. 11k_train_regILi0E...
                if f"{len(k)}{k}" in name:
                    found[k].append((name, code))
    out = {}
    for k, items in found.items():
        if not items:
            out[k] = None
            continue
        h = hashlib.sha256()
        for name, code in sorted(items):
            h.update(name.encode() + b"\0" + struct.pack("<Q", len(code)) + code)
        out[k] = h.hexdigest()
    return out


if __name__ == "__main__":
    import json
    import sys
    from . import _build
    print(json.dumps(kernel_digests(_build.LIB_PATH, sys.argv[1:] or ["k_train_reg"]), indent=1))
