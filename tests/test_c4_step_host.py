"""The body of the headline kernel (open_spiel_amd/csrc/osg_c4_step.h: c4_fused_step, host + device) driven
on the CPU over 200 000 random connect_four games — legal, illegal, out-of-range and "no action" inputs,
steps on finished games — against a plain array model written from the rules as connect_four.cc:130-209
states them.  Bit-exact planes (incl. the stored result flags), legal mask and status byte at every step."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fused_step_equals_the_array_model(tmp_path):
    exe = str(tmp_path / "c4_step_host_test")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--cuda-host-only", "-x", "hip", "-O2", "-w",
                           "-I", os.path.join(ROOT, "open_spiel_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "c4_step_host_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert r.stdout.startswith("ok: 200000 games")


def test_tic_tac_toe_fused_step_equals_the_array_model(tmp_path):
    """osg_ttt_step.h (the body of k_step_vec<Ttt>): random games with illegal / out-of-range / no-op actions and
    every pair of disjoint 9-bit boards as a status query, against an array model of tic_tac_toe.cc:109-148,215-227."""
    exe = str(tmp_path / "ttt_step_host_test")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--cuda-host-only", "-x", "hip", "-O2", "-w",
                           "-I", os.path.join(ROOT, "open_spiel_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "ttt_step_host_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert r.stdout.startswith("ok: 200000 games")


def test_counter_rng_and_keyed_orders_equal_the_oracles_restatement(tmp_path):
    """osg_common.h (Rng, order_key, path hash, fill_base / fill_key and the two-part fill_base the search kernel uses)
    against oracle/spiel_oracle_core.cpp over 200 000 random inputs: the functions every replay test relies on."""
    lib = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    exe = str(tmp_path / "keyed_order_host_test")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--cuda-host-only", "-x", "hip", "-O2", "-w",
                           "-I", os.path.join(ROOT, "open_spiel_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "native", "keyed_order_host_test.cpp"), "-x", "none", lib,
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert r.stdout.startswith("ok: 200000 random inputs")
