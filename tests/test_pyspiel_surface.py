"""`open_spiel_amd.pyspiel_hip`: scripts written against pyspiel's State / Game / MCTSBot /
CFRSolver API (open_spiel/python/pybind11/{pyspiel,bots,policy}.cc) run unchanged for the
hot-path games.  The GPU tests mirror the reference's Python tests of this surface."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def pyspiel():
    import __graft_entry__ as ge
    ge.build()
    from open_spiel_amd import pyspiel_hip
    return pyspiel_hip


def test_module_imports_and_describes_games_without_a_gpu(pyspiel):
    g = pyspiel.load_game("hex(board_size=9)")
    assert (g.num_distinct_actions(), g.num_players(), g.observation_tensor_shape()) == (81, 2, [9, 9, 9])
    assert str(pyspiel.load_game("kuhn_poker")) == "kuhn_poker()"
    with pytest.raises(pyspiel.SpielError):
        pyspiel.load_game("chess")
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(pyspiel.SpielError):   # no CPU fallback: creating states needs the device
            g.new_initial_state()


def test_game_headers_match_the_reference_playthroughs(pyspiel, goldens):
    """The header block of every reference playthrough (GetParameters with defaults, sizes, utilities,
    tensor piece names and shapes) through Game's methods and open_spiel_amd.observation; no GPU needed."""
    from open_spiel_amd.observation import _pieces
    for name, play in goldens.items():
        hdr = play["header"]
        game = pyspiel.load_game(play["game"])
        params = game.get_parameters()
        text = "{" + ",".join(f"{k}={params[k]}" for k in sorted(params)) + "}"
        assert text == hdr["GetParameters"], name
        assert game.num_distinct_actions() == int(hdr["NumDistinctActions"])
        assert game.max_chance_outcomes() == int(hdr["MaxChanceOutcomes"])
        assert game.num_players() == int(hdr["NumPlayers"])
        assert (game.min_utility(), game.max_utility(), game.utility_sum()) == (
            float(hdr["MinUtility"]), float(hdr["MaxUtility"]), float(hdr["UtilitySum"]))
        assert game.max_game_length() == int(hdr["MaxGameLength"])
        assert game.observation_tensor_size() == int(hdr["ObservationTensorSize"])
        assert str(game) == hdr["ToString"].strip('"')
        t = game.get_type()
        assert t.short_name == play["game"].split("(")[0] and t.utility == "Utility.ZERO_SUM"
        def shape_text(pieces):
            if len(pieces) == 1 and pieces[0][0] == "observation":
                return str(list(pieces[0][1]))
            return ", ".join(f"{n}: {list(sh)}" for n, sh in pieces)
        assert shape_text(_pieces(game, False)) == hdr["ObservationTensorShape"], name
        if "InformationStateTensorShape" in hdr:
            assert t.provides_information_state_tensor
            assert shape_text(_pieces(game, True)) == hdr["InformationStateTensorShape"], name
            assert game.information_state_tensor_size() == int(hdr["InformationStateTensorSize"])


@pytest.mark.gpu
def test_play_a_game_like_a_pyspiel_script(pyspiel):
    game = pyspiel.load_game("tic_tac_toe")
    state = game.new_initial_state()
    assert state.current_player() == 0 and not state.is_terminal()
    assert state.legal_actions() == list(range(9))
    assert state.legal_actions(1) == []                      # not the acting player (api_test.py:358)
    for a in [4, 0, 8, 1, 2, 6, 3, 5, 7]:                    # a drawn game
        assert a in state.legal_actions()
        state.apply_action(a)
    assert state.is_terminal() and state.returns() == [0.0, 0.0]
    assert state.current_player() == -4 and state.legal_actions() == []
    assert state.history() == [4, 0, 8, 1, 2, 6, 3, 5, 7]
    obs = state.observation_tensor(0)
    assert len(obs) == 27 and sum(obs) == 9
    assert str(state) == "oox\nxxo\noxx" and state.history_str() == "4, 0, 8, 1, 2, 6, 3, 5, 7"
    clone = game.new_initial_state().child(4)
    assert clone.history() == [4] and clone.current_player() == 1
    assert str(clone) == "...\n.x.\n..." and clone.observation_string(0) == str(clone)
    assert clone.action_to_string(1, 0) == "o(0,0)" and clone.action_to_string(8) == "o(2,2)"
    leduc = pyspiel.load_game("leduc_poker").new_initial_state()
    assert leduc.action_to_string(3) == "Chance outcome:3"   # chance to move
    leduc.apply_action(3)
    leduc.apply_action(0)
    assert leduc.action_to_string(0, 2) == "Raise" and "Round 1 sequence: " in str(leduc)
    with pytest.raises(pyspiel.SpielError):
        clone.apply_action(4)                                # occupied cell
    with pytest.raises(pyspiel.SpielError):
        clone.observation_tensor(2)                          # player out of range


@pytest.mark.gpu
def test_leduc_observation_goldens(pyspiel):
    """python/tests/observation_test.py:32-89: tensors after actions 1, 2, 2, 1, 3."""
    game = pyspiel.load_game("leduc_poker")
    state = game.new_initial_state()
    for a in [1, 2]:
        assert state.is_chance_node()
        state.apply_action(a)
    for a in [2, 1, 3]:
        state.apply_action(a)
    assert state.observation_tensor(0) == [1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 3, 3]
    info = state.information_state_tensor(0)
    assert len(info) == 30 and info[:14] == [1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0]
    assert state.information_state_string(0).startswith("[Observer: 0][Private: 1][Round 2]")
    probs = dict(game.new_initial_state().chance_outcomes())
    assert len(probs) == 6 and abs(sum(probs.values()) - 1) < 1e-12


@pytest.mark.gpu
def test_mcts_bot_solver_answers(pyspiel):
    """python/algorithms/mcts_test.py / mcts_test.cc:126-155 through MCTSBot.mcts_search."""
    game = pyspiel.load_game("tic_tac_toe")
    evaluator = pyspiel.RandomRolloutEvaluator(n_rollouts=20, seed=42)
    bot = pyspiel.MCTSBot(game, evaluator, 2.0, 10000, 10, True, 42, False)
    state = game.new_initial_state()
    for a in [1, 8]:                                          # "x(0,1) o(2,2)": x wins with x(0,2)
        state.apply_action(a)
    root = bot.mcts_search(state)
    assert root.outcome[root.player] == 1
    best = root.best_child()
    assert best.action == 2 and best.outcome[best.player] == 1
    assert bot.step(state) == 2
    assert sum(c.explore_count for c in root.children) == root.explore_count - 1
    # batch form: one search per state
    batch = game.new_initial_states(64)
    batch.apply_actions(np.full(64, 1, np.int32))
    batch.apply_actions(np.full(64, 8, np.int32))
    assert bot.step_batch(batch) == [2] * 64


@pytest.mark.gpu
def test_mcts_bot_step_with_policy(pyspiel):
    """Bot::StepWithPolicy (mcts.cc:268-271): the chosen action with probability one."""
    game = pyspiel.load_game("tic_tac_toe")
    bot = pyspiel.MCTSBot(game, pyspiel.RandomRolloutEvaluator(8, 3), 2.0, 200, 64, True, 11, False)
    state = game.new_initial_state()
    for a in (4, 0, 8):
        state.apply_action(a)
    policy, action = bot.step_with_policy(state)
    assert policy == [(action, 1.0)] and action in state.legal_actions()


@pytest.mark.gpu
def test_cfr_solver_like_cfr_example(pyspiel, oracle):
    """examples/cfr_example.cc:33-45 / python cfr_test.py: CFRSolver on kuhn_poker."""
    game = pyspiel.load_game("kuhn_poker")
    solver = pyspiel.CFRSolver(game)
    for _ in range(10):
        solver.evaluate_and_update_policy()
    solver.evaluate_and_update_policy(290)
    policy = solver.average_policy()
    table = policy.policy_table()
    assert len(table) == 12
    # judged by the oracle's exploitability (the reference's expl <= 0.05 after 300 iterations)
    keys = sorted(table)
    nact = np.array([len(table[k]) for k in keys], np.int32)
    acts = np.array([[a for a, _ in table[k]] for k in keys], np.int64)
    probs = np.array([[p for _, p in table[k]] for k in keys], np.float64)
    expl, ev = oracle.Game("kuhn_poker").eval_policy(keys, nact, acts, probs, which=1)
    assert expl <= 0.05 and abs(ev[0] + 1 / 18) < 1e-3
    assert abs(pyspiel.exploitability(game, policy) - expl) < 1e-12          # device judge == oracle judge
    assert abs(pyspiel.expected_returns(game, policy)[0] - ev[0]) < 1e-12
    assert pyspiel.nash_conv(game, policy) >= 0
    state = game.new_initial_state()
    state.apply_action(2)
    state.apply_action(0)
    assert state.information_state_string() == "2"
    ap = policy.action_probabilities(state)
    assert set(ap) == {0, 1} and abs(sum(ap.values()) - 1) < 1e-12
    values = solver.info_state_values_table()["2"]
    assert values.legal_actions == [0, 1] and len(values.cumulative_regrets) == 2
    mccfr = pyspiel.ExternalSamplingMCCFRSolver(game, 7)
    for _ in range(50):
        mccfr.run_iteration()
    mccfr.run_mini_batch(20000)
    assert len(mccfr.average_policy().policy_table()) == 12


@pytest.mark.gpu
def test_kuhn_optimal_policy_is_unexploitable(pyspiel):
    """pyspiel.kuhn_poker.get_optimal_policy (games_kuhn_poker.cc:23-24) judged on the device:
    exploitability 0 and value -1/18 for every alpha in [0, 1/3] (tabular_exploitability_test.cc)."""
    game = pyspiel.load_game("kuhn_poker")
    for alpha in (0.0, 0.2, 1.0 / 3):
        policy = pyspiel.kuhn_poker.get_optimal_policy(alpha)
        assert len(policy.policy_table()) == 12
        assert abs(pyspiel.exploitability(game, policy)) < 1e-12
        assert abs(pyspiel.nash_conv(game, policy)) < 1e-12
        ev = pyspiel.expected_returns(game, policy)
        assert abs(ev[0] + 1 / 18) < 1e-12 and abs(ev[1] - 1 / 18) < 1e-12
    with pytest.raises(pyspiel.SpielError):
        pyspiel.kuhn_poker.get_optimal_policy(0.5)


@pytest.mark.gpu
def test_cfr_solver_pickle_round_trip(pyspiel):
    """policy.cc:237-241: CFR solvers pickle through Serialize / DeserializeCFRSolver
    (cfr.cc:284-307,699-781); hex floats make the round trip lossless (cfr_test.cc:191-256)."""
    import pickle
    game = pyspiel.load_game("leduc_poker")
    solver = pyspiel.CFRPlusSolver(game)
    solver.evaluate_and_update_policy(3)
    text = solver.serialize()
    assert text.startswith("# Automatically generated by OpenSpiel CFRSolverBase::Serialize\n[Meta]\nVersion: 1\n")
    assert "[Game]\nleduc_poker()\n[SolverType]\nCFRPlusSolver\n[SolverSpecificState]\n3\n[SolverValuesTable]\n" in text
    clone = pickle.loads(pickle.dumps(solver))
    solver.evaluate_and_update_policy()
    clone.evaluate_and_update_policy()
    a, b = solver.info_state_values_table(), clone.info_state_values_table()
    assert len(a) == len(b) == 936
    for k, v in a.items():
        assert v.cumulative_regrets == b[k].cumulative_regrets
        assert v.cumulative_policy == b[k].cumulative_policy
        assert v.current_policy == b[k].current_policy
    with pytest.raises(pyspiel.SpielError):
        pyspiel.deserialize_cfr_solver(text)  # it is a CFRPlusSolver checkpoint


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["external", "outcome"])
def test_mccfr_solver_pickle_round_trip(pyspiel, kind):
    """external_sampling_mccfr_test.cc:112-171 / outcome_sampling_mccfr_test.cc: serialize, restore, and
    both solvers stay in lock step (same counter-RNG position, same tables)."""
    import pickle
    game = pyspiel.load_game("leduc_poker")
    solver = (pyspiel.ExternalSamplingMCCFRSolver(game, seed=7) if kind == "external"
              else pyspiel.OutcomeSamplingMCCFRSolver(game, epsilon=0.5, seed=7))
    for _ in range(10):
        solver.run_iteration()
    solver.run_mini_batch(5000)
    text = solver.serialize()
    name = "ExternalSamplingMCCFRSolver" if kind == "external" else "OutcomeSamplingMCCFRSolver"
    assert f"[SolverType]\n{name}\n[SolverSpecificState]\n[SolverRNG]\ncounter 7 5020\n" in text
    assert ("[SolverAverageType]\nSimpleAverageType\n" if kind == "external" else "[SolverEpsilon]\n") in text
    assert "[SolverDefaultPolicy]\nUniformPolicy:\n[SolverValuesTable]\n" in text
    restored = pickle.loads(pickle.dumps(solver))
    assert restored.info_state_values_table().keys() == solver.info_state_values_table().keys()
    for s in (solver, restored):
        s.run_iteration()
        s.run_mini_batch(3000)
    a, b = solver.info_state_values_table(), restored.info_state_values_table()
    for key in a:
        np.testing.assert_allclose(a[key].cumulative_regrets, b[key].cumulative_regrets, rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(a[key].cumulative_policy, b[key].cumulative_policy, rtol=1e-11, atol=1e-12)
    # a checkpoint of the other solver type is refused
    other = (pyspiel.deserialize_outcome_sampling_mccfr_solver if kind == "external"
             else pyspiel.deserialize_external_sampling_mccfr_solver)
    with pytest.raises(pyspiel.SpielError):
        other(text)


@pytest.mark.gpu
def test_serialize_game_and_state_round_trip(pyspiel):
    """spiel_test.cc / python/tests/pyspiel_test.py: serialize_game_and_state text, deserialize, pickle."""
    import pickle
    game = pyspiel.load_game("leduc_poker")
    state = game.new_initial_state()
    for a in (1, 3, 1, 2, 1):  # deals J / Q, call, raise, call
        state.apply_action(a)
    text = pyspiel.serialize_game_and_state(game, state)
    assert text == ("# Automatically generated by OpenSpiel SerializeGameAndState\n[Meta]\nVersion: 1\n\n"
                    "[Game]\nleduc_poker()\n[State]\n1\n3\n1\n2\n1\n\n")
    game2, state2 = pyspiel.deserialize_game_and_state(text)
    assert str(game2) == str(game) and state2.history() == state.history()
    assert state2.information_state_string(0) == state.information_state_string(0)
    assert state2.current_player() == state.current_player() and state2.legal_actions() == state.legal_actions()
    state3 = pickle.loads(pickle.dumps(state))
    assert state3.history() == state.history() and state3.observation_tensor(0) == state.observation_tensor(0)
    assert str(pickle.loads(pickle.dumps(game))) == str(game)
    assert game.deserialize_state(state.serialize()).history() == state.history()
    c4 = pyspiel.load_game("connect_four")
    s = c4.new_initial_state()
    assert pyspiel.deserialize_game_and_state(pyspiel.serialize_game_and_state(c4, s))[1].history() == []
    with pytest.raises(pyspiel.SpielError):
        c4.deserialize_state("9\n")  # not a column


@pytest.mark.gpu
def test_observation_wrapper_like_observation_test(pyspiel):
    """python/tests/observation_test.py:32-89 through open_spiel_amd.observation: tensor, named pieces in
    the observers' order, views that follow set_from, the information-state string."""
    from open_spiel_amd.observation import INFO_STATE_OBS_TYPE, IIGObservationType, PrivateInfoType, make_observation
    game = pyspiel.load_game("leduc_poker")
    state = game.new_initial_state()
    for a in (1, 2, 2, 1, 3):  # deal 1, deal 2, bet, call, deal 3
        state.apply_action(a)
    obs = make_observation(game)
    obs.set_from(state, player=0)
    np.testing.assert_array_equal(obs.tensor, [1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 3, 3])
    assert list(obs.dict) == ["player", "private_card", "community_card", "pot_contribution"]
    np.testing.assert_array_equal(obs.dict["player"], [1, 0])
    np.testing.assert_array_equal(obs.dict["private_card"], [0, 1, 0, 0, 0, 0])
    np.testing.assert_array_equal(obs.dict["community_card"], [0, 0, 0, 1, 0, 0])
    np.testing.assert_array_equal(obs.dict["pot_contribution"], [3, 3])
    info = make_observation(game, INFO_STATE_OBS_TYPE)
    info.set_from(state, player=0)
    assert list(info.dict) == ["player", "private_card", "community_card", "betting"]
    np.testing.assert_array_equal(info.dict["betting"], [[[0, 1], [1, 0], [0, 0], [0, 0]],
                                                         [[0, 0], [0, 0], [0, 0], [0, 0]]])
    assert info.string_from(state, 0) == ("[Observer: 0][Private: 1][Round 2][Player: 0][Pot: 6]"
                                          "[Money: 97 97][Public: 3][Round1: 2 1][Round2: ]")
    info.set_from(state, player=1)  # the dict entries are views into the tensor
    np.testing.assert_array_equal(info.dict["player"], [0, 1])
    np.testing.assert_array_equal(info.dict["private_card"], [0, 0, 1, 0, 0, 0])
    assert make_observation(game, IIGObservationType(perfect_recall=True, private_info=PrivateInfoType.ALL_PLAYERS)) is None
    kuhn = pyspiel.load_game("kuhn_poker")
    ks = kuhn.new_initial_state()
    for a in (2, 0, 1):  # deals 2 / 0, player 0 bets
        ks.apply_action(a)
    ko = make_observation(kuhn, INFO_STATE_OBS_TYPE)
    ko.set_from(ks, 1)
    assert list(ko.dict) == ["player", "private_card", "betting"] and ko.dict["betting"].shape == (3, 2)
    np.testing.assert_array_equal(ko.dict["private_card"], [1, 0, 0])
    np.testing.assert_array_equal(ko.dict["betting"][0], [0, 1])
    c4 = pyspiel.load_game("connect_four")
    co = make_observation(c4)
    cs = c4.new_initial_state()
    cs.apply_action(3)
    co.set_from(cs, 0)
    assert list(co.dict) == ["observation"] and co.dict["observation"].shape == (3, 6, 7)
    assert co.dict["observation"][0, 0, 3] == 1 and co.dict["observation"][2].sum() == 41
    assert make_observation(c4, INFO_STATE_OBS_TYPE) is None
