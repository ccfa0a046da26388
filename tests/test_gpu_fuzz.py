"""A bounded slice of the differential campaign tests/fuzz_parity.py in the GPU suite: 250 random configurations of the whole supported space
(domain x basis x order x agent x policy x weight mode x dtype x fuse depth x episode cap x learner count x env offset, random train() splits), the HIP
path through the C ABI against the oracle's device-order instantiation, bit for bit.  (Round 5: 9 000 cases over eight seeds ran clean after the
campaign's two findings -- the oracle's f32 trace rate, and SARSALambda / QLambda missing on the generic Fourier orders -- were fixed.)"""
import os
import subprocess
import sys
import json

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_configurations_bitwise_against_the_oracle():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "250", "20260929"], capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("SUMMARY ")]
    assert line, (p.stdout[-2000:], p.stderr[-2000:])
    d = json.loads(line[0][8:])
    assert p.returncode == 0 and not d["failures"], d["failures"][:3]
    assert d["counts"].get("ok", 0) >= 240 and d["counts"].get("refused", 0) == 0, d["counts"]      # every sampled configuration exists
