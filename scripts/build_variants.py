#!/usr/bin/env python3
"""A/B variants of the library: `name=-DFLAG=V[,-DFLAG2=V]` ... -> rsrl_amd/lib/variants/name.so (used through RSRL_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsrl_amd import _build
os.makedirs(os.path.join(_build.LIB_DIR, "variants"), exist_ok=True)
for spec in sys.argv[1:]:
    name, flags = spec.split("=", 1)
    out = os.path.join(_build.LIB_DIR, "variants", name + ".so")
    _build.build(out=out, extra_flags=flags.split(","))
    print(out)
