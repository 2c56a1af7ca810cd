"""The reference's OWN unit tests for the hot path, run on the genuine reference build.

oracle/Makefile.ref `reftests` compiles the reference's *_test.cc files (unmodified, from
/root/reference, each with its own main()) against the same abseil / nlohmann stand-ins and links
them with oracle/_ref/libspiel_ref.so.  Passing them says the recipe + stand-ins build a reference
that meets the reference's own expectations: game rules and playthrough invariants
(tests/basic_tests.cc RandomSimTest etc.), the JSON struct API of tic_tac_toe / connect_four, MCTS
solver answers and garbage collection, CFR / CFR+ convergence and serialization round trips,
ES-MCCFR bounds and serialization (including its RNG state), the exploitability known answers.

Needs the reference sources (build container only): skipped elsewhere.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEST_DIR = os.path.join(ROOT, "oracle", "_ref", "tests")

PASSING = ["tic_tac_toe_test", "connect_four_test", "hex_test", "kuhn_poker_test", "leduc_poker_test",
           "mcts_test", "cfr_test", "external_sampling_mccfr_test", "tabular_exploitability_test",
           "tensor_view_test", "action_view_test", "random_test", "nlohmann_json_test"]


@pytest.fixture(scope="module")
def reftests(reference):
    if not reference.sources_present():
        pytest.skip("needs the reference sources (/root/reference)")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref", "-j8",
                           "REF=" + reference.REFERENCE_ROOT, "reftests"])
    return TEST_DIR


def _run(reftests, name):
    return subprocess.run([os.path.join(reftests, name)], capture_output=True, text=True, timeout=1200)


@pytest.mark.parametrize("name", PASSING)
def test_reference_unit_test_passes_on_the_genuine_build(reftests, name):
    r = _run(reftests, name)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]


def test_outcome_sampling_mccfr_test_trips_only_its_stream_dependent_inequality(reftests):
    """outcome_sampling_mccfr_test.cc passes its convergence bounds and its serialization round trip
    here; the one check that can fail is :72, `exploitability2 > exploitability3` — 500 more
    iterations of a solver whose draws come from absl::discrete_distribution /
    absl::uniform_real_distribution with a fixed seed.  The stand-ins draw from libstdc++'s
    distributions instead (abseil's streams are the documented unpinned part), and on
    that stream the exploitability happens to go 0.113 -> 0.126.  Anything else failing is a bug."""
    r = _run(reftests, "outcome_sampling_mccfr_test")
    out = r.stdout + r.stderr
    assert "Game: kuhn_poker" in out and "Game: leduc_poker" in out   # the bound tests ran (and passed: no abort before)
    if r.returncode != 0:
        assert "outcome_sampling_mccfr_test.cc:72 exploitability2 > exploitability3" in out, out[-2000:]


def test_reference_command_line_tools_run_on_the_genuine_build(reference):
    """examples/benchmark_game.cc (the one throughput tool the reference ships for this path) and
    examples/mcts_example.cc (BASELINE.json configs[0]: tic_tac_toe MCTSBot with RandomRolloutEvaluator,
    CPU plumbing), built unmodified (`make -f oracle/Makefile.ref reftools`, with a small absl/flags
    stand-in).  1000-simulation MCTS self-play of tic_tac_toe never loses a game."""
    if not reference.sources_present():
        pytest.skip("needs the reference sources (/root/reference)")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref", "-j4",
                           "REF=" + reference.REFERENCE_ROOT, "reftools"])
    tools = os.path.join(ROOT, "oracle", "_ref", "tools")
    r = subprocess.run([os.path.join(tools, "benchmark_game"), "--game=connect_four", "--sims=2000", "--attempts=1"],
                       capture_output=True, text=True, timeout=300, check=True)
    assert "Benchmark: game: connect_four, num_sims: 2000." in r.stdout and "moves/s" in r.stdout
    r = subprocess.run([os.path.join(tools, "mcts_example"), "--game=tic_tac_toe", "--player1=mcts", "--player2=mcts",
                        "--max_simulations=1000", "--rollout_count=20", "--seed=42", "--num_games=5", "--quiet"],
                       capture_output=True, text=True, timeout=600, check=True)
    out = r.stdout + r.stderr
    assert "Number of games played: 5" in out and "Overall wins: 0,0" in out
