"""Round-5 GPU tests: the hardware evidence behind the assembly post-pass, rollouts without a limit, ..."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rsrl_amd
    return rsrl_amd


def test_packed_fp32_producer_and_next_slot_consumer_are_interlocked():
    # rsrl_amd/_asmfilter.py removes the compiler's `s_nop 0` between v_pk_{fma,mul,add}_f32 and a VALU consumer of the result.  The micro-test
    # isolates that pair inside one inline-asm statement -- no wait state, `s_nop 0`, `s_nop 4` -- for a VOP2, a VOP3 and a VOP3P consumer over
    # > 10^6 random operand sets per variant, the destination poisoned with NaNs beforehand: every result must equal the compiler's own arithmetic
    # bit for bit, with AND without the wait state (a stale read would return the poison).
    from rsrl_amd import _build
    exe = _build.build_pk_forward()
    out = subprocess.run([exe, "1048576"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["lane_pairs_per_variant"] >= 1000000
    assert len(r["mismatches"]) == 9 and all(v == 0 for v in r["mismatches"].values()), r
