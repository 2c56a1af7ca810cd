"""Shared by tests/test_dropin.py (CPU) and tests/test_z2_gpu_dropin.py (GPU): building the two variants of
tests/dropin/user_program.cc and comparing transcripts."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCE = os.path.join(ROOT, "tests", "dropin", "user_program.cc")
TRANSCRIPT = os.path.join(ROOT, "tests", "golden", "dropin_transcript.txt")
REFERENCE_ROOT = os.environ.get("OSG_REFERENCE_ROOT", "/root/reference")


def build_reference_variant(out_path):
    """The user program against the genuine reference headers + oracle/_ref/libspiel_ref.so."""
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-pthread", "-I", os.path.join(ROOT, "oracle", "ref_shim"),
                           "-I", REFERENCE_ROOT, SOURCE, "-o", out_path, "-L", ref_dir, "-lspiel_ref",
                           f"-Wl,-rpath,{ref_dir}"])


def build_hip_variant(out_path):
    """The SAME source, not a character changed, with the product's drop-in headers on the include path
    (include/open_spiel/** -> the MI355X host mirror) + open_spiel_amd/libosg_hip.so."""
    lib_dir = os.path.join(ROOT, "open_spiel_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-pthread", "-I", os.path.join(ROOT, "include"), SOURCE,
                           "-o", out_path, "-L", lib_dir, "-losg_hip", f"-Wl,-rpath,{lib_dir}"])


_FLOAT = re.compile(r"-?\d+\.\d+(?:e[-+]?\d+)?")


def assert_same_transcript(got, want, float_atol=1e-9):
    """Line by line; on the solver-output lines (fp64 results of the CFR family: every line naming "CFR") numbers may
    differ by float_atol."""
    got_lines, want_lines = got.rstrip("\n").split("\n"), want.rstrip("\n").split("\n")
    assert len(got_lines) == len(want_lines), f"{len(got_lines)} lines, expected {len(want_lines)}"
    for i, (g, w) in enumerate(zip(got_lines, want_lines)):
        if g == w:
            continue
        assert "CFR" in w, f"line {i + 1}:\n  got  {g}\n  want {w}"
        assert _FLOAT.sub("#", g) == _FLOAT.sub("#", w), f"line {i + 1}:\n  got  {g}\n  want {w}"
        for a, b in zip(_FLOAT.findall(g), _FLOAT.findall(w)):
            assert abs(float(a) - float(b)) <= float_atol, f"line {i + 1}: {a} vs {b}"
