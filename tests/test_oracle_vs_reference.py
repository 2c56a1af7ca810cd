"""The restatement (oracle/liboracle.so) against the GENUINE reference build
(oracle/_ref/libspiel_ref.so), call for call.

oracle/_ref is the reference's own .cc files for the hot path compiled unmodified
from /root/reference by oracle/Makefile.ref; both libraries export the same
extern "C" driver (oracle/spiel_oracle_capi.cpp), so every deterministic result
the GPU tests take from the oracle is checked here against the real
implementation: bit for bit for legal sets, players, terminal flags, returns,
tensors and strings, and bit for bit for the fp64 CFR tables and judge values
too (same additions in the same order).  Not compared: anything that depends on
abseil's / libstdc++'s random streams (MCTS visit counts, sampled MCCFR tables)
— those are pinned at outcome level by tests/test_oracle_known_answers.py,
which runs on both builds.

Skipped when the library is not built (it needs /root/reference at build time;
the built file travels with the repo snapshot).
"""
import numpy as np
import pytest

GAME_CONFIGS = [
    ("tic_tac_toe", 400),
    ("connect_four", 300),
    ("connect_four(rows=5,columns=6,x_in_row=3)", 300),
    ("connect_four(egocentric_obs_tensor=True)", 200),
    ("hex(board_size=9)", 60),
    ("hex", 30),
    ("hex(board_size=5,swap=True)", 300),
    ("hex(num_rows=3,num_cols=4)", 300),
    ("hex(board_size=4,plain_obs_tensor=True)", 200),
    ("kuhn_poker", 400),
    ("kuhn_poker(players=3)", 400),
    ("kuhn_poker(players=4)", 300),
    ("leduc_poker", 400),
    ("leduc_poker(players=3)", 300),
    ("leduc_poker(suit_isomorphism=True)", 300),
    # the remaining configurations the GPU parity tests take from the oracle (tests/test_gpu_parity.py GAMES)
    ("hex(board_size=5)", 200),
    ("hex(num_cols=2,num_rows=3)", 200),
    ("hex(num_cols=2,num_rows=2)", 200),
    ("hex(board_size=5,plain_obs_tensor=True,swap=True)", 200),
    ("hex(board_size=4,string_rep=explicit)", 200),
    ("kuhn_poker(players=5)", 200),
    ("kuhn_poker(players=10)", 100),
    ("leduc_poker(action_mapping=True)", 300),
    ("leduc_poker(players=3,starting_player=2)", 200),
]


def _pair(oracle, reference, game_string):
    try:
        rg = reference.Game(game_string)
    except reference.OracleError as e:  # a parameter this reference version does not know
        pytest.skip(f"reference rejects {game_string}: {e}")
    return oracle.Game(game_string), rg


@pytest.mark.parametrize("game_string,n", GAME_CONFIGS)
def test_game_description_matches(oracle, reference, game_string, n):
    og, rg = _pair(oracle, reference, game_string)
    for attr in ("num_distinct_actions", "max_chance_outcomes", "num_players",
                 "observation_tensor_size", "information_state_tensor_size", "max_game_length",
                 "max_chance_nodes_in_history", "min_utility", "max_utility", "has_chance"):
        assert getattr(og, attr) == getattr(rg, attr), attr
    assert str(og) == str(rg)
    assert og.observation_tensor_shape() == rg.observation_tensor_shape()
    assert og.information_state_tensor_shape() == rg.information_state_tensor_shape()


@pytest.mark.parametrize("game_string,n", GAME_CONFIGS)
def test_seeded_playouts_are_bit_identical(oracle, reference, game_string, n):
    """Same seeded playouts through both implementations; every per-ply record equal:
    LegalActions (chance outcomes at chance nodes), CurrentPlayer, IsTerminal, Returns,
    ObservationTensor and InformationStateTensor of every player."""
    og, rg = _pair(oracle, reference, game_string)
    a = og.random_playouts(0xC0FFEE, n, want_obs=True, want_info=True)
    b = rg.random_playouts(0xC0FFEE, n, want_obs=True, want_info=True)
    assert a["longest"] == b["longest"] > 0
    for k in ("actions", "mask", "cur_player", "terminal", "returns", "obs", "info"):
        if a[k] is None:
            assert b[k] is None
            continue
        assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("game_string,n", GAME_CONFIGS)
def test_strings_and_offturn_queries_match_along_playouts(oracle, reference, game_string, n):
    """ToString / HistoryString / InformationStateString / ObservationString / ActionToString,
    ChanceOutcomes and LegalActions(player) for every player, at every state of seeded playouts."""
    og, rg = _pair(oracle, reference, game_string)
    count = max(4, min(25, n // 12))
    rec = og.random_playouts(0x51A7E, count)
    has_info = og.information_state_tensor_size > 0 or "poker" in game_string
    for i in range(count):
        so, sr = og.new_initial_state(), rg.new_initial_state()
        for t in range(rec["actions"].shape[1] + 1):
            assert str(so) == str(sr)
            assert so.history_str() == sr.history_str()
            assert so.history() == sr.history()
            assert so.current_player() == sr.current_player()
            assert so.is_terminal() == sr.is_terminal()
            assert so.returns() == sr.returns()
            for p in range(og.num_players):
                assert so.observation_string(p) == sr.observation_string(p)
                if has_info:
                    assert so.information_state_string(p) == sr.information_state_string(p)
                if not so.is_terminal():
                    assert so.legal_actions(p) == sr.legal_actions(p)
            if so.is_terminal():
                break
            if so.is_chance_node():
                assert so.chance_outcomes() == sr.chance_outcomes()
            legal = so.legal_actions()
            assert legal == sr.legal_actions()
            cp = so.current_player()
            for a in legal:
                assert so.action_to_string(cp, a) == sr.action_to_string(cp, a)
            a = int(rec["actions"][i, t]) if t < rec["actions"].shape[1] else -1
            if a < 0:
                break
            so.apply_action(a)
            sr.apply_action(a)


@pytest.mark.parametrize("game_string", ["kuhn_poker", "leduc_poker", "kuhn_poker(players=3)"])
def test_tree_census_matches(oracle, reference, game_string):
    og, rg = _pair(oracle, reference, game_string)
    assert og.tree_census() == rg.tree_census()


@pytest.mark.parametrize("game_string,kind,iters", [
    ("kuhn_poker", "cfr", 300),
    ("kuhn_poker", "cfr_plus", 200),
    ("kuhn_poker", "cfr_simultaneous", 60),
    ("kuhn_poker(players=3)", "cfr", 40),
    ("kuhn_poker(players=3)", "cfr_plus", 25),
    ("leduc_poker", "cfr", 12),
    ("leduc_poker", "cfr_plus", 12),
    ("leduc_poker", "cfr_simultaneous", 6),
])
def test_cfr_tables_are_bit_identical(oracle, reference, game_string, kind, iters):
    """CFRSolver / CFRPlusSolver / CFRSolverBase(simultaneous): cumulative regrets, cumulative
    policy, current policy and the normalised average policy after `iters` iterations, checked
    at several points on the way — equal to the last bit (cfr.cc:331-408 adds in DFS order;
    the restatement adds in the same order), and so are NashConv / exploitability / expected
    returns of the average policy."""
    og, rg = _pair(oracle, reference, game_string)
    so, sr = oracle.Solver(og, kind), reference.Solver(rg, kind)
    done = 0
    for stop in sorted({1, 2, iters // 2, iters}):
        so.iterate(stop - done)
        sr.iterate(stop - done)
        done = stop
        a, b = so.tables(), sr.tables()
        assert a["keys"] == b["keys"]
        for k in ("nact", "legal", "regrets", "cum_policy", "cur_policy", "avg_policy"):
            assert np.array_equal(a[k], b[k]), (k, stop)
    assert so.nash_conv() == sr.nash_conv()
    assert so.exploitability() == sr.exploitability()
    assert np.array_equal(so.expected_returns(), sr.expected_returns())


@pytest.mark.parametrize("game_string", ["kuhn_poker", "leduc_poker", "kuhn_poker(players=3)"])
def test_judge_of_named_and_supplied_policies_matches(oracle, reference, game_string):
    """tabular_exploitability.cc / best_response.cc / expected_returns.cc on the uniform and the
    first-action policy, the Kuhn optimal family, and a supplied table (a CFR average policy)."""
    og, rg = _pair(oracle, reference, game_string)
    for which_policy in (0, 1):
        for which in (0, 1):
            assert og.eval_named_policy(which_policy, which) == rg.eval_named_policy(which_policy, which)
    if game_string == "kuhn_poker":
        for alpha in (0.0, 0.1, 0.25, 1.0 / 3.0):
            assert og.eval_named_policy(2, 0, alpha) == rg.eval_named_policy(2, 0, alpha)
    so = oracle.Solver(og, "cfr")
    so.iterate(8)
    t = so.tables()
    for which in (0, 1):
        vo, evo = og.eval_policy(t["keys"], t["nact"], t["legal"], t["avg_policy"], which)
        vr, evr = rg.eval_policy(t["keys"], t["nact"], t["legal"], t["avg_policy"], which)
        assert vo == vr
        assert np.array_equal(evo, evr)


def test_replay_hooks_are_restatement_only(reference):
    """The genuine build has no counter-stream hooks: asking for them is an error, not a silent
    fallback (so a parity test can never believe it replayed a device search on the reference)."""
    g = reference.Game("tic_tac_toe")
    with pytest.raises(reference.OracleError):
        g.new_initial_state().mcts_search(2.0, 10, 1, 5, False, 1, counter_root=0, counter_seed=1)
    s = reference.Solver(reference.Game("kuhn_poker"), "mccfr_simple", 1)
    with pytest.raises(reference.OracleError):
        s.mccfr_minibatch(1, 0, 4)


@pytest.mark.parametrize("game_string,kind,iters,seed", [
    ("kuhn_poker", "mccfr_simple", 500, 7),
    ("leduc_poker", "mccfr_simple", 300, 3),
    ("kuhn_poker(players=3)", "mccfr_simple", 200, 11),
    ("kuhn_poker(players=3)", "mccfr_full", 300, 39693847),
    ("leduc_poker", "mccfr_full", 100, 1),
    ("kuhn_poker", "mccfr_outcome", 500, 5),
    ("leduc_poker", "mccfr_outcome", 300, 5),
])
def test_sampled_mccfr_tables_are_bit_identical(oracle, reference, game_string, kind, iters, seed):
    """ExternalSamplingMCCFRSolver draws from std::mt19937 + std::uniform_real_distribution
    (external_sampling_mccfr.h:110-111) — libstdc++ on both sides, no abseil — so its whole table is
    reproducible: same infostates discovered, same regrets / average policy to the last bit after
    hundreds of iterations.  OutcomeSamplingMCCFRSolver uses absl's distributions (std's in the
    stand-in): equality there checks the traversal logic, not abseil's stream."""
    og, rg = _pair(oracle, reference, game_string)
    so, sr = oracle.Solver(og, kind, seed), reference.Solver(rg, kind, seed)
    done = 0
    for stop in (1, iters // 3, iters):
        so.iterate(stop - done)
        sr.iterate(stop - done)
        done = stop
        a, b = so.tables(), sr.tables()
        assert a["keys"] == b["keys"]
        for k in ("nact", "legal", "regrets", "cum_policy", "cur_policy", "avg_policy"):
            assert np.array_equal(a[k], b[k]), (k, stop)
    assert so.nash_conv() == sr.nash_conv()


@pytest.mark.parametrize("game_string,depth", [
    ("tic_tac_toe", 0), ("tic_tac_toe", 3), ("tic_tac_toe", 5),
    ("connect_four", 0), ("connect_four", 9),
    ("hex(board_size=5)", 0), ("hex(board_size=5)", 6), ("hex(board_size=4,swap=True)", 1),
    ("kuhn_poker", 2), ("kuhn_poker", 3), ("leduc_poker", 2), ("leduc_poker", 4),
    ("kuhn_poker(players=3)", 3),
])
@pytest.mark.parametrize("solve", [False, True])
def test_mcts_search_trees_are_identical(oracle, reference, game_string, depth, solve):
    """MCTSBot::MCTSearch (mcts.cc:273-467) with RandomRolloutEvaluator: the root's children —
    action order after the shuffle, visit counts, total rewards, proven outcomes — and the chosen
    action, for UCT and PUCT.  Both builds draw from std::mt19937 through the same conventions
    (the stand-in's absl::Uniform; see oracle/ref_shim), so select / expand / rollout / backup /
    MCTS-Solver are compared decision for decision."""
    og, rg = _pair(oracle, reference, game_string)
    rec = og.random_playouts(0xBEEF + depth, 3, stop=[depth] * 3)
    for i in range(3):
        so, sr = og.new_initial_state(), rg.new_initial_state()
        for t in range(depth):
            a = int(rec["actions"][i, t])
            if a < 0:
                break
            so.apply_action(a)
            sr.apply_action(a)
        if so.is_terminal() or so.is_chance_node():  # a bot is never asked to move at a chance node
            continue
        for puct in (False, True):
            for sims, n_rollouts in ((60, 1), (250, 3)):
                a = so.mcts_search(2.0, sims, n_rollouts, 100, solve, 42 + i, puct=puct)
                b = sr.mcts_search(2.0, sims, n_rollouts, 100, solve, 42 + i, puct=puct)
                assert a["best_action"] == b["best_action"]
                assert a["root_visits"] == b["root_visits"]
                assert np.array_equal(a["root_outcome"], b["root_outcome"], equal_nan=True)
                assert np.array_equal(a["children"], b["children"], equal_nan=True)


def test_mcts_selfplay_is_identical(oracle, reference):
    """Two MCTS bots playing each other (mcts_test.cc:45-77): same game, move for move."""
    for game_string in ("tic_tac_toe", "kuhn_poker", "leduc_poker"):
        og, rg = _pair(oracle, reference, game_string)
        for seed in (1, 2, 3):
            assert np.array_equal(og.mcts_selfplay(2.0, 100, 5, seed), rg.mcts_selfplay(2.0, 100, 5, seed))


@pytest.mark.parametrize("game_string", ["tic_tac_toe", "connect_four", "hex(board_size=3)", "kuhn_poker", "leduc_poker"])
def test_error_behaviour_matches(oracle, reference, game_string):
    """What the reference treats as fatal (SpielFatalError / SPIEL_CHECK -> an exception under the pybind
    handler) the restatement rejects too: unknown games and parameters, an action applied to a terminal
    state, a tensor for a player that does not exist.  Neither side may silently accept."""
    for impl in (oracle, reference):
        with pytest.raises(impl.OracleError):
            impl.Game("no_such_game")
        with pytest.raises(impl.OracleError):
            impl.Game(game_string.split("(")[0] + "(no_such_parameter=1)")
    og, rg = _pair(oracle, reference, game_string)
    rec = og.random_playouts(3, 1)
    for impl, g in ((oracle, og), (reference, rg)):
        s = g.new_initial_state()
        for a in rec["actions"][0]:
            if a < 0:
                break
            s.apply_action(int(a))
        assert s.is_terminal()
        assert s.legal_actions() == []
        if "poker" not in game_string:
            # the board games refuse (PlayerToState of the terminal player id, tic_tac_toe.cc:66-76,
            # connect_four.cc:60-69; hex.cc:229 checks the cell).  The poker games' DoApplyAction has no such
            # check in a Release build (only SPIEL_DCHECKs), so nothing is asserted for them; the MI355X engine
            # itself rejects every action on a terminal state (OSG_ERR_ILLEGAL), the stricter of the two.
            with pytest.raises(impl.OracleError):
                s.apply_action(0)
        with pytest.raises(impl.OracleError):
            s.observation_tensor(g.num_players)
        with pytest.raises(impl.OracleError):
            s.observation_tensor(-1)


@pytest.mark.parametrize("alt", [0, 1])
@pytest.mark.parametrize("lin", [0, 1])
@pytest.mark.parametrize("rmp", [0, 1])
def test_every_cfr_switch_combination_is_bit_identical(oracle, reference, alt, lin, rmp):
    """CFRSolverBase(game, alternating_updates, linear_averaging, regret_matching_plus) (cfr.h:188-196) for all
    eight switch combinations, kuhn 30 iterations and leduc 3."""
    kind = f"cfr_base_alt{alt}_lin{lin}_rmp{rmp}"
    for game_string, iters in (("kuhn_poker", 30), ("leduc_poker", 3)):
        og, rg = _pair(oracle, reference, game_string)
        so, sr = oracle.Solver(og, kind), reference.Solver(rg, kind)
        so.iterate(iters)
        sr.iterate(iters)
        a, b = so.tables(), sr.tables()
        assert a["keys"] == b["keys"]
        for k in ("regrets", "cum_policy", "cur_policy", "avg_policy"):
            assert np.array_equal(a[k], b[k]), (game_string, kind, k)


def test_mcts_with_garbage_collection_is_identical(oracle, reference):
    """mcts.cc:441-482: a 1 MB node budget forces MCTSBot::GarbageCollect during the search (connect_four,
    unsolved, 60 000 simulations): the surviving root statistics are the same in both builds."""
    og, rg = _pair(oracle, reference, "connect_four")
    so, sr = og.new_initial_state(), rg.new_initial_state()
    for a in (3, 3, 2):
        so.apply_action(a)
        sr.apply_action(a)
    a = so.mcts_search(2.0, 60000, 1, 1, False, 7)
    b = sr.mcts_search(2.0, 60000, 1, 1, False, 7)
    assert a["root_visits"] == b["root_visits"] == 60000
    assert a["best_action"] == b["best_action"]
    assert np.array_equal(a["children"], b["children"], equal_nan=True)


@pytest.mark.parametrize("game_string", ["tic_tac_toe", "connect_four", "hex(board_size=9)", "kuhn_poker", "leduc_poker"])
def test_rollout_replay_sums_are_identical(oracle, reference, game_string):
    """RandomRolloutEvaluator-style playouts from mid-game positions on the device's counter streams
    (osgo_replay_rollouts: what the GPU rollout parity tests take from the oracle): summed returns and the
    number of steps played are the same in both builds."""
    og, rg = _pair(oracle, reference, game_string)
    rec = og.random_playouts(0xA11CE, 6)
    for i in range(6):
        hist = [int(a) for a in rec["actions"][i] if a >= 0]
        hist = hist[:len(hist) // 2]
        a_sum, a_steps = og.replay_rollouts(hist, 99, 1000 + i, 16)
        b_sum, b_steps = rg.replay_rollouts(hist, 99, 1000 + i, 16)
        assert a_steps == b_steps > 0
        assert np.array_equal(a_sum, b_sum)


@pytest.mark.parametrize("game_string,warm,count", [
    ("kuhn_poker", 50, 400),
    ("leduc_poker", 40, 300),
    ("kuhn_poker(players=3)", 30, 200),
])
def test_frozen_table_replay_pins(oracle, reference, game_string, warm, count):
    """osgo_mccfr_frozen_replay (the full-size checker of the device's ES-MCCFR mini-batch: config 5's parity
    record in tests/test_gpu_timed_batch.py and bench.py) against (a) the slow per-trajectory hook
    osgo_mccfr_minibatch of the restatement — same streams, same frozen table, sums in the same trajectory order
    with one thread: equal to the last few ulps of the table-minus-table differences the slow hook forms — and
    (b) itself on the GENUINE reference build, whose State / SampleAction / regret matching are the reference's
    own code: bit for bit, one thread and several (per-thread sums are added in thread order)."""
    og = oracle.Game(game_string)
    slow = oracle.Solver(og, "mccfr_simple", seed=0)
    slow.iterate(warm)                      # a table that is not all 1e-6 (and does not hold every infostate yet)
    t0 = slow.tables()
    seed, first = 0xFEED5, 1000
    slow.mccfr_minibatch(seed, first, count)
    t1 = slow.tables()
    # rows the warm-up never created are discovered by the mini-batch: list every row of t1, frozen values from t0
    idx0 = {k: i for i, k in enumerate(t0["keys"])}
    keys = list(t1["keys"])
    amax = t1["regrets"].shape[1]
    frozen = np.full((len(keys), amax), 1e-6)
    cum0 = np.full((len(keys), amax), 1e-6)
    for j, k in enumerate(keys):
        if k in idx0:
            frozen[j] = t0["regrets"][idx0[k]]
            cum0[j] = t0["cum_policy"][idx0[k]]
    fast = oracle.mccfr_frozen_replay(og, keys, frozen, seed, first, count, threads=1)
    for j, k in enumerate(keys):
        n = int(t1["nact"][j])
        np.testing.assert_allclose(frozen[j, :n] + fast["d_regrets"][j, :n], t1["regrets"][j, :n], rtol=1e-12, atol=1e-12,
                                   err_msg=f"{game_string} regrets at {k!r}")
        np.testing.assert_allclose(cum0[j, :n] + fast["d_cum_policy"][j, :n], t1["cum_policy"][j, :n], rtol=1e-12,
                                   atol=1e-12, err_msg=f"{game_string} cum_policy at {k!r}")
    assert int(fast["visits"].sum()) > count      # every trajectory visits several infostates
    assert (np.abs(fast["d_regrets"]) <= fast["mass"] + 1e-300).all()
    rg = reference.Game(game_string)
    for threads in (1, 3):
        a = oracle.mccfr_frozen_replay(og, keys, frozen, seed, first, count, threads=threads)
        b = reference.mccfr_frozen_replay(rg, keys, frozen, seed, first, count, threads=threads)
        for name in ("d_regrets", "d_cum_policy", "mass", "visits"):
            assert np.array_equal(a[name], b[name]), f"{game_string} {name} threads={threads}"
    # thread count only changes the order of the final per-thread adds
    np.testing.assert_allclose(a["d_regrets"], fast["d_regrets"], rtol=0, atol=1e-11 * max(1.0, fast["mass"].max()))
    assert np.array_equal(a["visits"], fast["visits"])
