#!/usr/bin/env python3
"""Compile one HIP source for gfx950 with -save-temps and print per-basic-block instruction histograms
plus register usage of the kernels whose mangled name contains a given substring.

    python scripts/isa_stats.py rsrl_amd/csrc/train_reg_d0b.hip k_train_regILi0ELi5ELi0ELi1E
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = os.path.abspath(sys.argv[1])
    pat = sys.argv[2]
    min_block = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    extra = sys.argv[4:]
    with tempfile.TemporaryDirectory() as td:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c",
               "-save-temps", src, "-o", os.path.join(td, "x.o")] + extra
        subprocess.run(cmd, cwd=td, check=True, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
        s = open(os.path.join(td, asm)).read()
    for name in re.findall(r"^(\S+):\s*; @\S+", s, flags=re.M):
        if pat not in name or "__device_stub__" in name:
            continue
        i = s.index(name + ":")
        j = s.index(".Lfunc_end", i)
        blocks, order, cur = {}, ["entry"], "entry"
        for ln in s[i:j].split("\n"):
            t = ln.strip()
            m = re.match(r"^(\.LBB\d+_\d+):", t)
            if m:
                cur = m.group(1)
                order.append(cur)
                continue
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            blocks.setdefault(cur, []).append(t.split()[0])
        print("==", name)
        tot = 0
        for b in order:
            c = Counter(blocks.get(b, []))
            n = sum(c.values())
            tot += n
            if n >= min_block:
                print(f"  {b:12s} {n:5d}  " + " ".join(f"{k}:{v}" for k, v in c.most_common(9)))
        k = s.index(".amdhsa_kernel " + name)
        meta = s[k:k + 4000]
        regs = {key: re.search(rf"\.amdhsa_{key} (\d+)", meta) for key in
                ("next_free_vgpr", "accum_offset", "next_free_sgpr", "private_segment_fixed_size", "group_segment_fixed_size")}
        print("  total static instrs", tot, {k: (int(v.group(1)) if v else None) for k, v in regs.items()})


if __name__ == "__main__":
    main()
