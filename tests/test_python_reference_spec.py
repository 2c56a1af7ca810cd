"""The reference's PYTHON algorithm files as a second, independent specification.

open_spiel/python/algorithms/{cfr,exploitability,best_response,get_all_states}.py and
python/policy.py are imported unmodified from /root/reference and run on top of the genuine C++
games (oracle/_ref) through a minimal `pyspiel` stand-in (oracle/pyspiel_over_capi.py: the real
pybind11 module cannot be built here).  Their results are compared with the reference's C++ solvers
(same library) and with the restatement, and the known answers of the reference's own Python tests
(python/algorithms/exploitability_test.py:40-100, cfr_test.py:195-229) are checked.

Needs /root/reference: skipped elsewhere (build container only).
"""
import os
import sys

import numpy as np
import pytest

REFERENCE_ROOT = os.environ.get("OSG_REFERENCE_ROOT", "/root/reference")


@pytest.fixture(scope="module")
def py(reference):
    if not reference.sources_present():
        pytest.skip("needs the reference sources (/root/reference)")
    import pyspiel_over_capi
    pyspiel = pyspiel_over_capi.install(reference)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from open_spiel.python import policy
    from open_spiel.python.algorithms import cfr, exploitability

    class Bundle:
        pass
    b = Bundle()
    b.pyspiel, b.policy, b.cfr, b.exploitability = pyspiel, policy, cfr, exploitability
    return b


def _python_tables(solver, keys):
    reg, cum = [], []
    for k in keys:
        node = solver._info_state_nodes[k]  # pylint: disable=protected-access
        reg.append([node.cumulative_regret[a] for a in node.legal_actions])
        cum.append([node.cumulative_policy[a] for a in node.legal_actions])
    return reg, cum


@pytest.mark.parametrize("game_string,kind,iters", [
    ("kuhn_poker", "cfr", 40), ("kuhn_poker", "cfr_plus", 25), ("kuhn_poker", "cfr_simultaneous", 12),
    ("kuhn_poker(players=3)", "cfr", 4), ("leduc_poker", "cfr", 2),
])
def test_python_cfr_equals_the_cpp_solvers_and_the_restatement(py, oracle, reference, game_string, kind, iters):
    """python/algorithms/cfr.py (its own recursion, numpy accumulators) against CFRSolver /
    CFRPlusSolver / CFRSolverBase(simultaneous) of cfr.cc and against the restatement: cumulative
    regrets and cumulative policy of every infostate.  (python/algorithms/cfr_test.py:246-272 makes
    the same comparison against the pybind solvers, to 1e-9.)"""
    game = py.pyspiel.load_game(game_string)
    if kind == "cfr":
        solver = py.cfr.CFRSolver(game)
    elif kind == "cfr_plus":
        solver = py.cfr.CFRPlusSolver(game)
    else:
        solver = py.cfr._CFRSolver(game, regret_matching_plus=False, linear_averaging=False,  # pylint: disable=protected-access
                                   alternating_updates=False)
    cpp = reference.Solver(reference.Game(game_string), kind)
    mine = oracle.Solver(oracle.Game(game_string), kind)
    for _ in range(iters):
        solver.evaluate_and_update_policy()
    cpp.iterate(iters)
    mine.iterate(iters)
    tc, tm = cpp.tables(), mine.tables()
    assert tc["keys"] == tm["keys"] == sorted(solver._info_state_nodes)  # pylint: disable=protected-access
    reg, cum = _python_tables(solver, tc["keys"])
    for j in range(len(tc["keys"])):
        n = int(tc["nact"][j])
        for table, rows in (("regrets", reg), ("cum_policy", cum)):
            np.testing.assert_allclose(rows[j], tc[table][j, :n], rtol=0, atol=1e-12)
            np.testing.assert_allclose(rows[j], tm[table][j, :n], rtol=0, atol=1e-12)
    # and the Python judge on the Python average policy agrees with the C++ judge on the C++ one
    if game_string == "kuhn_poker":
        nc = py.exploitability.nash_conv(game, solver.average_policy(), use_cpp_br=False)
        assert abs(nc - cpp.nash_conv()) <= 1e-12
        assert abs(nc - mine.nash_conv()) <= 1e-12


def test_python_exploitability_known_answers(py, oracle, reference):
    """python/algorithms/exploitability_test.py:40-100 on the Python judge, next to both C++ judges."""
    game = py.pyspiel.load_game("kuhn_poker")
    uniform = py.policy.UniformRandomPolicy(game)
    r0 = py.exploitability.best_response(game, uniform, player_id=0)
    assert r0["best_response_action"] == {"0": 1, "1": 1, "2": 0, "0pb": 0, "1pb": 1, "2pb": 1}
    r1 = py.exploitability.best_response(game, uniform, player_id=1)
    assert r1["best_response_action"] == {"0p": 1, "1p": 1, "2p": 1, "0b": 0, "1b": 1, "2b": 1}
    nc = py.exploitability.nash_conv(game, uniform, use_cpp_br=False)
    assert abs(nc - 11 / 12) < 1e-12
    for impl in (oracle, reference):
        assert abs(impl.Game("kuhn_poker").eval_named_policy(0, 0) - nc) < 1e-14
    assert abs(py.exploitability.nash_conv(game, py.policy.FirstActionPolicy(game), use_cpp_br=False) - 2) < 1e-12
    leduc = py.pyspiel.load_game("leduc_poker")
    nc = py.exploitability.nash_conv(leduc, py.policy.UniformRandomPolicy(leduc), use_cpp_br=False)
    assert abs(nc - 4.747222222222222) < 1e-12
    for impl in (oracle, reference):
        assert abs(impl.Game("leduc_poker").eval_named_policy(0, 0) - nc) < 1e-12


@pytest.mark.parametrize("regret_matching_plus", [False, True])
def test_python_simultaneous_two_step_average(py, regret_matching_plus):
    """python/algorithms/cfr_test.py:195-229: after two simultaneous-update iterations the average
    policy at "1b" is [0.5 / 2, 1.5 / 2]."""
    game = py.pyspiel.load_game("kuhn_poker", {"players": 2})
    solver = py.cfr._CFRSolver(game, regret_matching_plus=regret_matching_plus, linear_averaging=False,  # pylint: disable=protected-access
                               alternating_updates=False)
    solver.evaluate_and_update_policy()
    solver.evaluate_and_update_policy()
    np.testing.assert_allclose(solver.average_policy().policy_for_key("1b"), [0.25, 0.75])
    assert len(solver.current_policy().state_lookup) == 12   # cfr_test.py:231-240
