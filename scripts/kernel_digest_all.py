"""sha256 of the machine code of EVERY device function in librsrl_hip.so, one line per symbol: `python scripts/kernel_digest_all.py > a.txt`
before and after a refactoring, `diff` = which kernels' code moved (none, for a source-only reshuffle)."""
import hashlib
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rsrl_amd import _build, _kdigest

path = sys.argv[1] if len(sys.argv) > 1 else _build.LIB_PATH
blob = open(path, "rb").read()
rows = {}
for elf in _kdigest._code_objects(blob):
    for name, code in _kdigest._functions(elf):
        rows[name] = hashlib.sha256(code).hexdigest()[:16] + f" {len(code)}"
for name in sorted(rows):
    print(rows[name], name)
