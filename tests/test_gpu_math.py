"""The shortened fp32 division of the path-tracing kernels (pt_math.h fdiv: rcp + one correction of the reciprocal + one of the
quotient, without the compiler's operand scaling) must give the bits of the IEEE quotient on normal operands.  Compiles the stand-alone
probe (tools/probes/lean_div_probe.hip: the same instruction sequence against `a / b`) with hipcc and runs it on the MI355X."""
import os
import re
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shortened_division_matches_the_ieee_quotient_on_normal_operands(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "lean_div_probe")
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-o", exe, os.path.join(ROOT, "tools", "probes", "lean_div_probe.hip")],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300).stdout
    rows = re.findall(r"exponents \[(-?\d+), (-?\d+)\], ([0-9.e+]+) operand pairs: division mismatches L1 (\d+)", out)
    assert len(rows) >= 4, out
    checked = 0.0
    for lo, hi, pairs, bad in rows:
        if int(lo) >= -60 and int(hi) <= 60:            # dividend, divisor and quotient all normal
            assert int(bad) == 0, (lo, hi, bad)
            checked += float(pairs)
    assert checked >= 8e9, checked
