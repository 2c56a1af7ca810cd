"""Source-level drop-in: tests/dropin/user_program.cc — one user program in the style of
the reference's examples — compiles unchanged against the genuine reference API and against the MI355X
host mirror.  Here (CPU): both variants build; the reference variant runs and must reproduce the committed
transcript tests/golden/dropin_transcript.txt (regenerate with OSG_UPDATE_GOLDEN=1); the mirror variant
must refuse to run without a GPU.  tests/test_z2_gpu_dropin.py runs the mirror variant on the device and
compares it with the same transcript."""
import os
import subprocess

import pytest

import dropin_common as dc


def test_user_program_builds_against_the_mirror_and_fails_loudly_without_a_gpu(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    exe = str(tmp_path / "user_hip")
    dc.build_hip_variant(exe)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "no HIP device" in (r.stderr + r.stdout)


def test_reference_variant_reproduces_the_committed_transcript(reference, tmp_path):
    if not reference.sources_present():
        pytest.skip("needs the reference headers (/root/reference)")
    exe = str(tmp_path / "user_ref")
    dc.build_reference_variant(exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, check=True)
    assert r.stdout.endswith("done\n")
    if os.environ.get("OSG_UPDATE_GOLDEN") == "1" or not os.path.exists(dc.TRANSCRIPT):
        with open(dc.TRANSCRIPT, "w") as f:
            f.write(r.stdout)
    with open(dc.TRANSCRIPT) as f:
        dc.assert_same_transcript(r.stdout, f.read(), float_atol=0.0)
