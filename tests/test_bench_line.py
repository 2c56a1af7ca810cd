"""The bench line the driver keeps (bench.compact_line): strict JSON, under bench.LINE_LIMIT characters, with every
field of the contract — on the full record of round 5's N = 1 run (tests/golden/bench_full_record_r05_n1.json: the
21 KB line the driver could not keep) and on the same record widened to 8 ranks with every N > 1 field present."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _strict(text):
    def no_constants(name):
        raise ValueError(f"non-JSON constant {name}")
    return json.loads(text, parse_constant=no_constants)


@pytest.fixture()
def full_n1():
    with open(os.path.join(ROOT, "tests", "golden", "bench_full_record_r05_n1.json")) as f:
        return json.load(f)


def _full_n8(full):
    d = copy.deepcopy(full)
    w = 8
    d.update(n_gpus=w, collective_backend="nccl", rccl_world=w)
    d["value"] = full["value"] * 7.9
    d["per_rank"] = {"env_steps_per_s": [2.0248850677e11 + i for i in range(w)],
                     "avg_launch_us": [5.144965171813965 + i * 1e-3 for i in range(w)], "note": "x" * 200}
    m = d["secondary"]["mcts"]
    m["per_rank_sims_per_s"] = [1.3781367488e8 + i for i in range(w)]
    m["single_rank_all_roots"] = {"value": 1.1025093990466063e9, "unit": "sims/s", "seconds": 0.0608, "what": "y" * 100}
    m["strong_scaling_efficiency"] = 0.8512345678
    x = d["secondary"]["mccfr"]
    x.update(allreduce_us=23.456789123, allreduce_bytes=44928, allreduce_backend="torch.distributed nccl",
             oneshot={"allreduce_us": 9.87654321, "trajectories_per_s": 2.123456789e10, "nash_conv_after": 1.48, "what": "z" * 300})
    return d


def test_compact_line_n1_is_small_strict_and_complete(full_n1):
    assert len(json.dumps(full_n1)) > 16000          # the record that was lost
    line = bench.compact_line(full_n1)
    text = json.dumps(line, allow_nan=False)
    assert len(text) < bench.LINE_LIMIT == 4096, len(text)
    back = _strict(text)
    assert "secondary_truncated" not in back
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_checked_states", "detail"):
        assert k in back, k
    assert back["value"] == pytest.approx(full_n1["value"], rel=1e-5) and back["n_gpus"] == 1 and back["dtype"] == "u64"
    assert back["config"]["states_per_gpu"] == 1 << 20 and "model" not in back["config"] and back["config"]["workload"]
    rf = back["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
              "avg_launch_us", "hbm_frac", "hbm_avg_launch_us", "hbm_states", "hbm_traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-4)
    # algorithmic bytes / avg launch time = achieved: the cross-check the driver could not run in round 5
    assert rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9 == pytest.approx(rf["achieved"], rel=1e-4)
    cb = back["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"] and cb["unit"]
    assert cb["single_thread_value"] > 0
    sec = back["secondary"]
    assert set(sec) >= {"mcts", "cfr", "cfr_leduc_3p", "mccfr", "env_step", "hex_step", "ttt_mcts"}
    for k, v in sec.items():
        for kk, vv in v.items():
            assert not isinstance(vv, (dict, list)) or k in ("mccfr", "mcts"), (k, kk)   # scalars beyond one level
        if k != "nash_conv_us":
            assert v["value"] > 0 and v["unit"]
    assert sec["mcts"]["parity_checked_roots"] == 1024 and sec["cfr"]["parity_checked_iterations"] == 20100
    assert sec["mccfr"]["parity_checked_trajectories"] == 1 << 20
    # no prose: nothing in the line is longer than a workload label
    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(s) for s in strings(back)) <= 100


def test_compact_line_n8_carries_the_first_contact_fields(full_n1):
    line = bench.compact_line(_full_n8(full_n1))
    text = json.dumps(line, allow_nan=False)
    assert len(text) < bench.LINE_LIMIT, len(text)
    back = _strict(text)
    assert "secondary_truncated" not in back
    assert back["n_gpus"] == 8 and back["collective_backend"] == "nccl" and back["rccl_world"] == 8
    assert len(back["per_rank"]["env_steps_per_s"]) == 8 and len(back["per_rank"]["avg_launch_us"]) == 8
    m, x = back["secondary"]["mcts"], back["secondary"]["mccfr"]
    assert len(m["per_rank_sims_per_s"]) == 8 and m["strong_scaling_efficiency"] == pytest.approx(0.851235)
    assert m["single_rank_all_roots"] > 0
    assert x["allreduce_us"] == {"rccl": pytest.approx(23.4568), "oneshot": pytest.approx(9.87654)}
    assert x["allreduce_bytes"] == 44928 and x["oneshot_trajectories_per_s"] > 0


def test_compact_line_survives_failed_legs_and_non_finite_numbers(full_n1):
    d = copy.deepcopy(full_n1)
    d["secondary"]["env_step"] = {"error": "RuntimeError: " + "e" * 5000}
    d["secondary"]["hex_step"] = {"error": "boom"}
    d["secondary"]["cfr"]["leduc"] = {"error": "nope"}
    d["secondary"]["mccfr"]["nash_conv_after"] = float("nan")
    d["roofline"].pop("dram_leg"); [d["roofline"].pop(k) for k in list(d["roofline"]) if k.startswith("hbm_")]
    d.pop("cpu_baseline")
    text = json.dumps(bench.compact_line(d), allow_nan=False)
    assert len(text) < bench.LINE_LIMIT
    back = _strict(text)
    assert back["secondary"]["env_step"]["error"].startswith("RuntimeError") and "cpu_baseline" not in back
    assert back["secondary"]["mccfr"]["nash_conv_after"] is None      # NaN never reaches the line
    d["secondary"] = {"error": "x" * 9000, "traceback": "t" * 9000}
    back = _strict(json.dumps(bench.compact_line(d), allow_nan=False))
    assert len(back["secondary"]["error"]) == 200


def test_emit_prints_the_compact_line_last_and_writes_the_detail_file(full_n1, tmp_path, monkeypatch, capsys):
    monkeypatch.setenv("OSG_BENCH_DETAIL_DIR", str(tmp_path))
    bench.emit(full_n1)
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert len(lines) == 2 and len(lines[-1]) < bench.LINE_LIMIT
    last = _strict(lines[-1])
    assert last["metric"].startswith("env-steps/sec") and not lines[0].startswith('{"metric"')
    assert _strict(lines[0])["bench_detail"]["secondary"]["mcts"]["parity"]["roots"] == 1024
    with open(os.path.join(str(tmp_path), bench.DETAIL_FILE)) as f:
        assert json.load(f)["roofline"]["dram_leg"]["states"] == 1 << 24
    assert last["detail"].endswith(bench.DETAIL_FILE)


def test_profile_source_stamps_detect_a_changed_kernel(tmp_path, monkeypatch):
    """tools/profile_sources.py: a profile stamped with the hashes of the kernel's sources is `current` until one of them
    changes (content, not time stamps), `unstamped` without its sidecar."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import profile_sources as ps
    csrc = tmp_path / "open_spiel_amd" / "csrc"
    csrc.mkdir(parents=True)
    (tmp_path / "profiles").mkdir()
    (csrc / "osg_cfr_small.hip").write_text("kernel v1")
    (csrc / "osg_cfr_mccfr.hip").write_text("the other kernel")
    (csrc / "osg_common.h").write_text("header v1")
    monkeypatch.setattr(ps, "ROOT", str(tmp_path))
    monkeypatch.setattr(ps, "CSRC", str(csrc))
    prof = tmp_path / "profiles" / "r09_pmc_solvers.json"
    prof.write_text("{}")
    assert ps.status("pmc_solvers") == (str(prof), "unstamped") and ps.status("pmc_k_mcts_wave") == (None, "missing")
    ps.stamp(str(prof), "pmc_solvers")
    assert ps.status("pmc_solvers")[1] == "current" and ps.is_current("profiles/r09_pmc_solvers.json") is True
    (csrc / "osg_cfr_small.hip").write_text("kernel v2")
    assert ps.status("pmc_solvers")[1] == "stale: open_spiel_amd/csrc/osg_cfr_small.hip"
    assert ps.is_current("profiles/r09_pmc_solvers.json") is False and ps.is_current("profiles/other.json") is None
