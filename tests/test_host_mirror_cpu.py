"""The host mirror's pure-host code WITHOUT a device (tests/native/host_mirror_cpu_test.cpp): the SearchNode tree
decoded from the device's flat arrays — a proven draw's outcome is {0, 0} and prints "0 0", never "-0" (the
round-1 GPU run went red on that) —, BestChild ordering, poker outcomes, dirichlet_noise, the observer piece
tables, ShardRange.  What the drop-in user program prints goes through this code."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_mirror_cpu_checks(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    exe = str(tmp_path / "host_mirror_cpu_test")
    lib_dir = os.path.join(ROOT, "open_spiel_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-I", ROOT,
                           os.path.join(ROOT, "tests", "native", "host_mirror_cpu_test.cpp"), "-o", exe,
                           "-L", lib_dir, "-losg_hip", f"-Wl,-rpath,{lib_dir}"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok: host mirror CPU checks", r.stdout + r.stderr
