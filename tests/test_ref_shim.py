"""Self-test of the abseil / nlohmann stand-ins (oracle/ref_shim) that let the genuine reference
sources compile: the forms the hot-path files use, checked against the real libraries' documented
behaviour (oracle/ref_shim_selftest.cpp).  Test infrastructure testing test infrastructure; g++ only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stand_ins_behave_as_documented(tmp_path):
    exe = tmp_path / "ref_shim_selftest"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused-variable",
                           "-I", os.path.join(ROOT, "oracle", "ref_shim"),
                           os.path.join(ROOT, "oracle", "ref_shim_selftest.cpp"), "-o", str(exe), "-pthread"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "all checks passed" in r.stdout
