"""Pin the CPU oracle against the reference's known-answer tests.

Every constant below is quoted from a test in /root/reference/open_spiel (cited
per test); nothing here reads the reference at run time.

Every test runs twice: on the restatement (oracle/liboracle.so) and on the
GENUINE reference build (oracle/_ref/libspiel_ref.so, the reference's own .cc
files compiled by oracle/Makefile.ref) — the second run checks the build recipe
and its abseil / nlohmann stand-ins against the reference's own expectations.
"""
import math

import numpy as np
import pytest


@pytest.fixture(scope="module", params=["restatement", "genuine_reference"])
def oracle(request):
    if request.param == "restatement":
        import oracle_py
        oracle_py.build()
        return oracle_py
    return request.getfixturevalue("reference")


def _play(game, actions):
    s = game.new_initial_state()
    for a in actions:
        s.apply_action(a)
    return s


# games/connect_four/connect_four_test.cc:38-59
def test_connect_four_fast_loss(oracle):
    g = oracle.Game("connect_four")
    s = _play(g, [3, 3, 4, 4, 2, 2])
    assert not s.is_terminal()
    s.apply_action(1)
    assert s.is_terminal()
    assert s.returns() == [1.0, -1.0]
    assert str(s) == ".......\n.......\n.......\n.......\n..ooo..\n.xxxx..\n"
    assert s.current_player() == -4


# games/connect_four/connect_four_test.cc:68-87 (full board, no line -> draw)
def test_connect_four_full_board_draw(oracle):
    g = oracle.Game("connect_four")
    rows_top_first = ["ooxxxoo", "xxoooxx", "ooxxxoo", "xxoooxx", "ooxxxoo", "xxoooxx"]
    rows = rows_top_first[::-1]
    cols = {"x": [], "o": []}
    # Build a legal move order: fill column by column is impossible (turn
    # alternation), so schedule greedily: keep per-column heights and pick any
    # column whose next cell belongs to the mover.
    height = [0] * 7
    s = g.new_initial_state()
    mover = "x"
    for _ in range(42):
        for c in range(7):
            if height[c] < 6 and rows[height[c]][c] == mover:
                s.apply_action(c)
                height[c] += 1
                break
        else:
            pytest.fail("no schedulable move")
        mover = "o" if mover == "x" else "x"
    assert str(s) == "\n".join(rows_top_first) + "\n"
    assert s.is_terminal()
    assert s.returns() == [0.0, 0.0]
    del cols


# games/hex/hex_test.cc:31-48
def test_hex_board_orientation(oracle):
    g = oracle.Game("hex(num_cols=3,num_rows=4)")
    s = _play(g, [1, 2, 4, 5, 7, 8, 10])
    assert s.is_terminal()
    assert s.returns() == [1.0, -1.0]


# games/hex/hex_test.cc:50-67
def test_hex_swap_rule(oracle):
    g = oracle.Game("hex(board_size=3,swap=True)")
    s = _play(g, [1])
    assert 9 in s.legal_actions()
    s.apply_action(9)
    la = s.legal_actions()
    assert 1 in la and 3 not in la
    assert s.current_player() == 0


# games/leduc_poker/leduc_poker_test.cc:68-96
def test_leduc_starting_player(oracle):
    g = oracle.Game("leduc_poker(players=3,starting_player=1)")
    s = g.new_initial_state()
    assert s.is_chance_node()
    for c in (0, 2, 4):
        s.apply_action(c)
    assert s.current_player() == 1
    s.apply_action(0)
    assert s.current_player() == 2
    s.apply_action(2)
    assert s.current_player() == 0
    s.apply_action(1)
    assert s.is_chance_node()
    s.apply_action(3)
    assert s.current_player() == 2


# python/tests/observation_test.py:32-89
def test_leduc_observation_goldens(oracle):
    g = oracle.Game("leduc_poker")
    s = _play(g, [1, 2, 2, 1, 3])
    np.testing.assert_array_equal(
        s.observation_tensor(0), [1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 3, 3])
    assert (s.observation_string(0) ==
            "[Observer: 0][Private: 1][Round 2][Player: 0][Pot: 6][Money: 97 97][Public: 3][Ante: 3 3]")
    np.testing.assert_array_equal(
        s.information_state_tensor(0),
        [1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    assert (s.information_state_string(0) ==
            "[Observer: 0][Private: 1][Round 2][Player: 0][Pot: 6][Money: 97 97][Public: 3][Round1: 2 1][Round2: ]")


# integration_tests/api_test.py:75-101 (chance, decision, terminal; infostates)
@pytest.mark.parametrize("game,counts,infostates", [
    ("kuhn_poker", (4, 24, 30), 12),
    ("leduc_poker", (157, 3780, 5520), 936),
    ("kuhn_poker(players=3)", (17, 288, 312), 48),
])
def test_tree_census(oracle, game, counts, infostates):
    c = oracle.Game(game).tree_census()
    assert tuple(c[:3]) == counts
    assert c[3] == infostates


# algorithms/tabular_exploitability_test.cc:470-499
def test_exploitability_known_answers(oracle):
    kuhn = oracle.Game("kuhn_poker")
    leduc = oracle.Game("leduc_poker")
    assert kuhn.eval_named_policy(0, 1) == pytest.approx(0.4583333333333335, abs=1e-12)
    assert leduc.eval_named_policy(0, 1) == pytest.approx(2.373611111111111, abs=1e-12)
    assert kuhn.eval_named_policy(0, 0) == pytest.approx(0.916666666666667, abs=1e-12)
    assert leduc.eval_named_policy(0, 0) == pytest.approx(4.747222222222222, abs=1e-12)
    assert kuhn.eval_named_policy(1, 0) == pytest.approx(2.0, abs=1e-12)   # first-action
    assert kuhn.eval_named_policy(2, 1, alpha=0.2) == pytest.approx(0.0, abs=1e-12)
    assert kuhn.eval_named_policy(2, 0, alpha=0.0) == pytest.approx(0.0, abs=1e-12)


# algorithms/cfr_test.cc:36-62
def test_cfr_kuhn_300(oracle):
    g = oracle.Game("kuhn_poker")
    solver = oracle.Solver(g, "cfr")
    t0 = solver.tables()
    assert len(t0["keys"]) == 12  # python/algorithms/cfr_test.py:231-240
    np.testing.assert_array_equal(t0["cur_policy"], np.full((12, 2), 0.5))
    solver.iterate(300)
    v = solver.expected_returns()
    assert v[0] == pytest.approx(-1 / 18, abs=1e-3)
    assert v[1] == pytest.approx(1 / 18, abs=1e-3)
    assert solver.exploitability() <= 0.05


# algorithms/cfr_test.cc:94-103
def test_cfr_plus_kuhn_200(oracle):
    g = oracle.Game("kuhn_poker")
    solver = oracle.Solver(g, "cfr_plus")
    solver.iterate(200)
    v = solver.expected_returns()
    assert v[0] == pytest.approx(-1 / 18, abs=1e-3)
    assert solver.exploitability() <= 0.05


# algorithms/cfr_test.cc:288-301
@pytest.mark.parametrize("game,bound", [
    ("kuhn_poker(players=3)", 1.0), ("kuhn_poker(players=4)", 1.0), ("leduc_poker", 2.0)])
def test_cfr_multiplayer_bounds(oracle, game, bound):
    solver = oracle.Solver(oracle.Game(game), "cfr")
    solver.iterate(10)
    assert solver.nash_conv() <= bound


# python/algorithms/cfr_test.py:195-229: simultaneous updates, two steps,
# average policy at "1b" = [0.5/2, 1.5/2]; uniform before and after step one.
def test_cfr_simultaneous_two_step(oracle):
    solver = oracle.Solver(oracle.Game("kuhn_poker"), "cfr_simultaneous")
    np.testing.assert_allclose(solver.tables()["avg_policy"], 0.5)
    solver.iterate(1)
    np.testing.assert_allclose(solver.tables()["avg_policy"], 0.5)
    solver.iterate(1)
    t = solver.tables()
    row = t["keys"].index("1b")
    np.testing.assert_allclose(t["avg_policy"][row], [0.5 / 2, 1.5 / 2])


# algorithms/external_sampling_mccfr_test.cc:104-109: seed 230398247, kuhn 1000
# iterations NashConv <= 0.05, leduc 1000 iterations <= 2.5.  The reference's
# abseil/libstdc++ draw sequence is unpinned, so the exact bounds are asserted
# for the reference seed on OUR stream and slightly looser ones (the spread at
# 1000 iterations is ~0.04-0.075 / 2.2-2.8) for other seeds.
@pytest.mark.parametrize("seed,kb,lb", [(230398247, 0.05, 2.5), (1, 0.1, 3.0), (2, 0.1, 3.0)])
def test_mccfr_bounds(oracle, seed, kb, lb):
    k = oracle.Solver(oracle.Game("kuhn_poker"), "mccfr_simple", seed)
    k.iterate(1000)
    assert k.nash_conv() <= kb
    l = oracle.Solver(oracle.Game("leduc_poker"), "mccfr_simple", seed)
    l.iterate(1000)
    assert l.nash_conv() <= lb
    l.iterate(4000)
    assert l.nash_conv() <= 1.2


def test_mccfr_improves(oracle):
    g = oracle.Game("kuhn_poker")
    s = oracle.Solver(g, "mccfr_simple", 7)
    s.iterate(1)
    e0 = s.exploitability()
    s.iterate(400)
    assert s.exploitability() < e0
    f = oracle.Solver(oracle.Game("kuhn_poker(players=3)"), "mccfr_full", 39693847)
    f.iterate(100)
    assert math.isfinite(f.nash_conv())


# algorithms/mcts_test.cc:126-155 (UCT_C=2, 10000 sims, 10 MB, solve, seed 42,
# RandomRolloutEvaluator(20, 42)).  ttt action id = 3*row+col.
def _ttt(oracle, names):
    g = oracle.Game("tic_tac_toe")
    s = g.new_initial_state()
    for nm in names.split():
        cp = s.current_player()
        a = [a for a in s.legal_actions() if s.action_to_string(cp, a) == nm]
        assert len(a) == 1
        s.apply_action(a[0])
    return s


def test_mcts_solve_draw(oracle):
    s = _ttt(oracle, "x(1,1) o(0,0) x(2,2)")
    assert str(s) == "o..\n.x.\n..x"
    r = s.mcts_search(2.0, 10000, 20, 10, True, 42)
    assert r["root_outcome"] == 0
    assert all(c[3] <= 0 for c in r["children"])
    assert s.action_to_string(1, r["best_action"]) in ("o(2,0)", "o(0,2)")


def test_mcts_solve_loss(oracle):
    s = _ttt(oracle, "x(1,1) o(0,0) x(2,2) o(0,1) x(0,2)")
    assert str(s) == "oox\n.x.\n..x"
    r = s.mcts_search(2.0, 10000, 20, 10, True, 42)
    assert r["root_outcome"] == -1
    assert all(c[3] == -1 for c in r["children"])


def test_mcts_solve_win(oracle):
    s = _ttt(oracle, "x(0,1) o(2,2)")
    assert str(s) == ".x.\n...\n..o"
    r = s.mcts_search(2.0, 10000, 20, 10, True, 42)
    assert r["root_outcome"] == 1
    assert s.action_to_string(0, r["best_action"]) == "x(0,2)"


# algorithms/mcts_test.cc:45-77: self-play returns are zero-sum.
def test_mcts_selfplay_zero_sum(oracle):
    g = oracle.Game("tic_tac_toe")
    r = g.mcts_selfplay(2.0, 100, 20, 42)
    assert r[0] + r[1] == 0


# algorithms/mcts_test.cc:157-170: 1 MB budget forces garbage collection.
def test_mcts_garbage_collect(oracle):
    s = oracle.Game("tic_tac_toe").new_initial_state()
    r = s.mcts_search(2.0, 200000, 1, 1, True, 42)
    assert (not math.isnan(r["root_outcome"])) or r["root_visits"] == 200000


# tests/basic_tests.cc:321-562 RandomSimTest invariants, restated.
@pytest.mark.parametrize("game", [
    "tic_tac_toe", "connect_four", "connect_four(rows=5,columns=6,x_in_row=3)",
    "hex(num_cols=5,num_rows=5)", "hex", "hex(num_cols=2,num_rows=3)", "hex(num_cols=2,num_rows=2)",
    "hex(swap=True)", "hex(plain_obs_tensor=True,swap=True)", "hex(board_size=9)",
    "kuhn_poker", "kuhn_poker(players=3)", "kuhn_poker(players=4)", "kuhn_poker(players=5)",
    "leduc_poker", "leduc_poker(players=3)", "leduc_poker(action_mapping=True)",
    "leduc_poker(suit_isomorphism=True)", "leduc_poker(players=3,starting_player=2)",
])
def test_random_sim_invariants(oracle, game):
    g = oracle.Game(game)
    n = 40
    rec = g.random_playouts(12345, n, want_obs=True, want_info=True)
    L = g.max_plies
    assert rec["longest"] <= L
    for i in range(n):
        t_end = int(np.argmax(rec["terminal"][i]))
        assert rec["terminal"][i, t_end] == 1, "playout must reach a terminal state"
        r = rec["returns"][i, t_end]
        assert abs(r.sum()) < 1e-9                      # zero-sum (basic_tests.cc:547-561)
        assert (r >= g.min_utility - 1e-9).all() and (r <= g.max_utility + 1e-9).all()
        assert (rec["returns"][i, :t_end] == 0).all()   # terminal reward model
        assert rec["cur_player"][i, t_end] == -4
        assert not rec["mask"][i, t_end].any()          # no legal actions at terminal
        assert (rec["mask"][i, :t_end].reshape(t_end, -1).any(axis=1)).all()
    assert np.isfinite(rec["obs"]).all()
    if rec["info"] is not None:
        assert np.isfinite(rec["info"]).all()


def test_outcome_sampling_mccfr_bounds(oracle):
    """outcome_sampling_mccfr_test.cc:36-48,86-88: 10000 iterations, seed 230398247: NashConv kuhn <= 0.17,
    leduc <= 3.07 (the action draw uses a CDF scan instead of absl::discrete_distribution)."""
    for game, bound in (("kuhn_poker", 0.17), ("leduc_poker", 3.07)):
        s = oracle.Solver(oracle.Game(game), "mccfr_outcome", seed=230398247)
        s.iterate(10000)
        assert s.nash_conv() <= bound
