"""The drop-in user program (tests/dropin/user_program.cc) built against the MI355X host mirror and run on
the device: its transcript — game descriptions, legal actions, tensors, strings, returns of fixed
playthroughs of the five games, MCTS-Solver proofs, CFR / CFR+ exploitabilities — equals the transcript the
SAME source produced when built against the genuine reference (tests/golden/dropin_transcript.txt)."""
import subprocess

import pytest

import dropin_common as dc

pytestmark = pytest.mark.gpu


def test_mirror_variant_reproduces_the_reference_transcript(tmp_path):
    exe = str(tmp_path / "user_hip")
    dc.build_hip_variant(exe)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    with open(dc.TRANSCRIPT) as f:
        dc.assert_same_transcript(r.stdout, f.read(), float_atol=1e-9)
