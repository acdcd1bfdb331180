"""Sanitizer-instrumented mutation fuzzing of the plain-C++ scene side: image decoders (PNG, JPEG, RGBE, EXR), the glTF / GLB
loader and the animation sampler (tests/fuzz/host_fuzz.cpp).  CPU only.  A longer run: `python tools/fuzz_host.py 20000 <seed>`.
The bugs it has found so far are pinned by explicit cases in test_gltf_loader.py."""
import os, shutil, subprocess, sys, tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_decoders_and_loader_survive_mutated_files():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    import fuzz_host
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, ".host_fuzz")
        try:
            fuzz_host.build(exe)
        except subprocess.CalledProcessError:
            pytest.skip("g++ cannot link the sanitizer runtimes here")
        fuzz_host.write_seeds(d)
        for args in (("3000", "101"), ("1500", "102", "sink")):
            r = subprocess.run([exe, d] + list(args), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, (r.stdout[-500:], r.stderr[-4000:])
            assert "fuzz:" in r.stdout
