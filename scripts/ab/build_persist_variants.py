#!/usr/bin/env python3
"""A/B variants of the library with pieces of k_shared_persist compiled out (for scripts/gpu_exp_persist.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsrl_amd import _build
for v in (sys.argv[1:] or ["0", "1", "2", "4", "7"]):
    out = os.path.join(_build.LIB_DIR, f"librsrl_hip_ab{v}.so")
    _build.build(out=out, extra_flags=[f"-DRSRL_PERSIST_ABLATE={v}"])
    print(out)
