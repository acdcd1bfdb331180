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
    # outside the contract, pinned so that the comment in pt_math.h stays true: an overflowing quotient is +-inf like a / b, zero /
    # infinite / NaN operands are repaired by v_div_fixup, and a subnormal divisor gives +-inf (or NaN) where IEEE gives a large finite number
    sp = {int(i): (l, e) for i, l, e in re.findall(r"special (\d+): .* -> fdiv (\S+) , ieee (\S+)", out)}
    assert len(sp) == 10, out
    as_f = lambda t: float(t.replace("-nan", "nan"))
    print("fdiv outside its contract (fdiv, ieee):", sp)
    for i in (0, 1, 2, 6, 8, 9):                           # overflow, x / 0, x / inf, inf / x: the same as IEEE
        assert as_f(sp[i][0]) == as_f(sp[i][1]), (i, sp[i])
    for i in (3, 4, 5):                                    # subnormal divisor
        l, e = as_f(sp[i][0]), as_f(sp[i][1])
        assert l != l or abs(l) == float("inf") or l == e, sp[i]
    assert as_f(sp[7][0]) != as_f(sp[7][0]) and as_f(sp[7][1]) != as_f(sp[7][1])          # 0 / 0
