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
        assert t.short_name == play["game"].split("(")[0] and t.utility == pyspiel.GameType.Utility.ZERO_SUM
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
    assert f"[SolverType]\n{name}\n[SolverSpecificState]\n[SolverRNG]\n" in text
    # the solver's std::mt19937 as the reference dumps it (RunIteration() draws from it), then the position of the
    # counter streams (external: the mini-batch drew 5000 trajectories from them; outcome: 10 iterations x 2 episodes
    # took a stream index each, then 5000)
    rng = text.split("[SolverRNG]\n")[1].split("[SolverAverageType]" if kind == "external" else "[SolverEpsilon]")[0].split("\n")
    assert len(rng[0].split()) == 625 and rng[1] == ("counter 7 5000" if kind == "external" else "counter 7 5020")
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
    every = make_observation(game, IIGObservationType(perfect_recall=True, private_info=PrivateInfoType.ALL_PLAYERS))
    every.set_from(state, player=0)       # leduc_poker.cc:119-129: every player's card, [players, cards]
    assert list(every.dict) == ["player", "private_cards", "community_card", "betting"]
    np.testing.assert_array_equal(every.dict["private_cards"], [[0, 1, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0]])
    assert every.string_from(state, 0) == "[Privates: 12][Round 2][Player: 0][Pot: 6][Money: 97 97][Public: 3][Round1: 2 1][Round2: ]"
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
    info_c4 = make_observation(c4, INFO_STATE_OBS_TYPE)   # observer.cc:158-159: the information-state string, no tensor
    assert info_c4.tensor is None and info_c4.string_from(cs, 0) == "3"


# ---- round 2: Bot base class, the whole SearchNode tree, any Evaluator, observer bindings, game submodules ----
@pytest.mark.gpu
def test_mcts_bot_is_a_bot_and_returns_the_whole_tree(pyspiel):
    game = pyspiel.load_game("connect_four")
    ev = pyspiel.RandomRolloutEvaluator(2, 7)
    bot = pyspiel.MCTSBot(game, ev, 2.0, 300, 100, True, 11, False, pyspiel.ChildSelectionPolicy.UCT,
                          max_wall_clock_time=-1.0, dirichlet_alpha=0.0, dirichlet_epsilon=0.0,
                          dont_return_chance_node=False)
    assert isinstance(bot, pyspiel.Bot) and bot.provides_policy() and not bot.is_clonable()
    state = game.new_initial_state()
    root = bot.mcts_search(state)
    assert root.explore_count == 300 and root.player == 0 and root.action == pyspiel.INVALID_ACTION
    assert sorted(c.action for c in root.children) == list(range(7))
    assert sum(c.explore_count for c in root.children) == 299
    grand = [g for c in root.children for g in c.children]
    assert grand and all(g.player == 1 for g in grand), "MCTSearch returns the tree below the root's children too"
    def check(node):  # a node's children were visited once less than the node (its first visit evaluated it)
        if node.children:
            assert sum(c.explore_count for c in node.children) in (node.explore_count - 1, node.explore_count)
            np.testing.assert_allclose(sum(c.prior for c in node.children), 1.0, atol=1e-12)
            for c in node.children:
                check(c)
    check(root)
    text = root.to_string(state)
    assert "sims:   300" in text and "7 children" in text
    assert len(root.children_str(state).strip().split("\n")) == 7
    action = bot.step(state)
    policy, action2 = bot.step_with_policy(state)
    assert policy == [(action2, 1.0)] and action in range(7)


@pytest.mark.gpu
def test_mcts_bot_with_a_python_evaluator_equals_the_oracle_replay(pyspiel, oracle):
    """Any Evaluator drives the device search: a Python subclass implementing the stub network of
    oracle/spiel_oracle_capi.cpp (integer arithmetic) — requests for prior(state) / evaluate(state) come back to
    the host with the State (history included) — and the result equals the oracle's MCTSBot(StubNetEvaluator)
    replayed on the same tree-policy streams, node for node at the root."""
    class Stub(pyspiel.Evaluator):
        def __init__(self):
            pyspiel.Evaluator.__init__(self)
            self.histories = []

        def _x(self, state):
            cur = state.current_player()
            return np.asarray(state.observation_tensor(cur if cur >= 0 else 0)).astype(np.int64)

        def evaluate(self, state):
            x = self._x(state)
            self.histories.append(state.history())
            i = np.arange(x.size)
            v = float(int((x * ((7 * i + 3) % 1009)).sum()) % 2001 - 1000) / 1024.0
            return [v, -v]

        def prior(self, state):
            if state.is_chance_node():
                return state.chance_outcomes()
            x = self._x(state)
            i = np.arange(x.size)
            legal = state.legal_actions()
            k = [1 + int((x * ((31 * i + 17 * a + 5) % 13)).sum()) % 7 for a in legal]
            return [(a, kk / sum(k)) for a, kk in zip(legal, k)]

    for game_name, moves, puct in [("tic_tac_toe", [4, 0], True), ("connect_four", [3, 3, 2], True), ("leduc_poker", [1, 4, 1], False)]:
        game = pyspiel.load_game(game_name)
        state = game.new_initial_state()
        ostate = oracle.Game(game_name).new_initial_state()
        for a in moves:
            state.apply_action(a)
            ostate.apply_action(a)
        ev = Stub()
        policy = pyspiel.ChildSelectionPolicy.PUCT if puct else pyspiel.ChildSelectionPolicy.UCT
        bot = pyspiel.MCTSBot(game, ev, 1.3, 120, 1000, False, 0x51, False, policy)
        root = bot.mcts_search(state)
        want = ostate.mcts_search_stub(1.3, 120, 0, 0x51, puct=puct)
        assert root.explore_count == want["root_visits"]
        got = {c.action: (c.explore_count, c.total_reward, c.prior) for c in root.children}
        assert sorted(got) == sorted(int(a) for a in want["children"][:, 0])
        for a, cnt, tot, pr in want["children"]:
            assert got[int(a)] == (int(cnt), tot, pr), (game_name, int(a), got[int(a)], (cnt, tot, pr))
        assert root.best_child().action == want["best_action"]
        assert all(h[:len(moves)] == moves for h in ev.histories), "the evaluator sees states with their history"


@pytest.mark.gpu
def test_mcts_bot_root_noise_and_wall_clock(pyspiel):
    game = pyspiel.load_game("connect_four")
    state = game.new_initial_state()
    ev = pyspiel.RandomRolloutEvaluator(1, 3)
    plain = pyspiel.MCTSBot(game, ev, 2.0, 64, 100, False, 5, False, pyspiel.ChildSelectionPolicy.PUCT).mcts_search(state)
    noisy = pyspiel.MCTSBot(game, ev, 2.0, 64, 100, False, 5, False, pyspiel.ChildSelectionPolicy.PUCT,
                            dirichlet_alpha=0.3, dirichlet_epsilon=0.25).mcts_search(state)
    p0 = {c.action: c.prior for c in plain.children}
    p1 = {c.action: c.prior for c in noisy.children}
    assert all(abs(p0[a] - 1 / 7) < 1e-15 for a in p0)
    assert abs(sum(p1.values()) - 1.0) < 1e-12 and max(abs(p1[a] - p0[a]) for a in p0) > 1e-3
    assert all(p1[a] >= 0.75 * p0[a] - 1e-12 for a in p0)
    timed = pyspiel.MCTSBot(game, ev, 2.0, 10 ** 7, 1, False, 5, False, pyspiel.ChildSelectionPolicy.UCT,
                            max_wall_clock_time=0.2)
    root = timed.mcts_search(state)
    assert 1 <= root.explore_count < 10 ** 7


@pytest.mark.gpu
def test_observer_bindings_like_python_observation(pyspiel):
    """python/pybind11/observer.cc:30-97: Game.make_observer, _Observation (buffer protocol, tensors(), set_from,
    string_from), SpanTensor views — driven the way python/observation.py:63-125 drives them."""
    game = pyspiel.load_game("kuhn_poker")
    state = game.new_initial_state()
    for a in (2, 1, 0):
        state.apply_action(a)
    info_type = pyspiel.IIGObservationType(perfect_recall=True)
    observer = game.make_observer(info_type, {})
    assert str(observer) == "Observer()"
    obs = pyspiel._Observation(game, observer)
    assert obs.has_tensor() and obs.has_string()
    tensor = np.frombuffer(obs, np.float32)
    assert tensor.shape == (game.information_state_tensor_size(),)
    views = {t.name: t.data for t in obs.tensors()}
    assert [(i.name, i.shape) for i in obs.tensors_info()] == [("player", [2]), ("private_card", [3]), ("betting", [3, 2])]
    obs.set_from(state, 1)
    np.testing.assert_array_equal(tensor, np.asarray(state.information_state_tensor(1), np.float32))
    np.testing.assert_array_equal(views["player"], [0, 1])
    np.testing.assert_array_equal(views["private_card"], [0, 1, 0])
    np.testing.assert_array_equal(views["betting"], [[1, 0], [0, 0], [0, 0]])
    assert obs.string_from(state, 1) == state.information_state_string(1)
    default = pyspiel._Observation(game, game.make_observer())
    default.set_from(state, 0)
    np.testing.assert_array_equal(np.frombuffer(default, np.float32), np.asarray(state.observation_tensor(0), np.float32))
    every = game.make_observer(pyspiel.IIGObservationType(private_info=pyspiel.PrivateInfoType.ALL_PLAYERS))
    assert [(i.name, i.shape) for i in pyspiel._Observation(game, every).tensors_info()] == [("pot_contribution", [2])]  # kuhn_poker.cc:82-105
    board = pyspiel.load_game("tic_tac_toe")
    assert [(i.name, i.shape) for i in pyspiel._Observation(board, board.make_observer()).tensors_info()] == [("observation", [3, 3, 3])]
    assert not pyspiel._Observation(board, board.make_observer(info_type)).has_tensor()   # observer.cc:158-159: the string only


@pytest.mark.gpu
def test_game_submodules(pyspiel):
    """games_tic_tac_toe.cc:37-100 and games_leduc_poker.cc:28-60."""
    ttt = pyspiel.tic_tac_toe
    assert (ttt.NUM_ROWS, ttt.NUM_COLS, ttt.NUM_CELLS) == (3, 3, 9)
    assert ttt.player_to_cellstate(0) == ttt.CellState.CROSS and ttt.player_to_cellstate(1) == ttt.CellState.NOUGHT
    assert [ttt.cellstate_to_string(c) for c in (ttt.CellState.EMPTY, ttt.CellState.NOUGHT, ttt.CellState.CROSS)] == [".", "o", "x"]
    s = pyspiel.load_game("tic_tac_toe").new_initial_state()
    for a in (4, 0, 8):
        s.apply_action(a)
    assert s.board_at(1, 1) == ttt.CROSS and s.board_at(0, 0) == ttt.NOUGHT and s.board_at(2, 2) == ttt.CROSS
    assert s.board_at(0, 1) == ttt.EMPTY and len(s.board()) == 9
    leduc = pyspiel.leduc_poker
    assert leduc.INVALID_CARD == -10000 and int(leduc.ActionType.RAISE) == 2 and leduc.FOLD == leduc.ActionType.FOLD
    s = pyspiel.load_game("leduc_poker").new_initial_state()
    assert s.get_private_cards() == [leduc.INVALID_CARD] * 2 and s.public_card() == leduc.INVALID_CARD
    for a in (0, 3, 2, 1, 4, 1):   # deal J0 to p0, Q1 to p1; raise, call; public card K0; call
        s.apply_action(a)
    assert s.get_private_cards() == [0, 3] and s.private_card(1) == 3 and s.public_card() == 4
    assert (s.round(), s.pot(), s.money()) == (2, 6, [97, 97])
    assert s.round1() == [2, 1] and s.round2() == [1]
    with pytest.raises(pyspiel.SpielError):
        pyspiel.load_game("kuhn_poker").new_initial_state().public_card()


@pytest.mark.gpu
def test_policy_hierarchy_and_judges(pyspiel):
    """python/pybind11/policy.cc:90-222: Policy / TabularPolicy / UniformPolicy / PreferredActionPolicy, the
    factories, and exploitability / nash_conv / expected_returns of ANY Policy — the reference's known answers
    (tabular_exploitability_test.cc: uniform kuhn 0.4583..., leduc 2.3736...; first-action NashConv 2;
    Kuhn optimal policy 0; policy_test.py / exploitability_test.py)."""
    kuhn = pyspiel.load_game("kuhn_poker")
    uniform = pyspiel.UniformPolicy()
    assert isinstance(uniform, pyspiel.Policy)
    assert abs(pyspiel.exploitability(kuhn, uniform) - 0.4583333333333335) < 1e-14
    tab = pyspiel.UniformRandomPolicy(kuhn)
    assert isinstance(tab, pyspiel.TabularPolicy) and isinstance(tab, pyspiel.Policy) and len(tab) == 12
    assert tab.get_state_policy("0pb") == [(0, 0.5), (1, 0.5)]
    assert abs(pyspiel.exploitability(kuhn, tab) - 0.4583333333333335) < 1e-14
    assert abs(pyspiel.nash_conv(kuhn, pyspiel.GetFirstActionPolicy(kuhn)) - 2.0) < 1e-14
    assert abs(pyspiel.nash_conv(kuhn, pyspiel.PreferredActionPolicy([0, 1])) - 2.0) < 1e-14   # always pass == first action
    assert abs(pyspiel.exploitability(kuhn, pyspiel.kuhn_poker.get_optimal_policy(0.2))) < 1e-14
    ev = pyspiel.expected_returns(kuhn, pyspiel.kuhn_poker.get_optimal_policy(0.1))
    assert abs(ev[0] + 1 / 18) < 1e-14 and abs(ev[1] - 1 / 18) < 1e-14
    leduc = pyspiel.load_game("leduc_poker")
    assert abs(pyspiel.exploitability(leduc, pyspiel.UniformRandomPolicy(leduc)) - 2.373611111111111) < 1e-12
    # a state's policy through every accessor
    s = kuhn.new_initial_state()
    for a in (1, 0, 1):
        s.apply_action(a)
    assert tab.action_probabilities(s) == {0: 0.5, 1: 0.5} == uniform.action_probabilities(s, 1)
    assert tab.get_state_policy_as_parallel_vectors(s) == ([0, 1], [0.5, 0.5])
    tab.set_prob(s.information_state_string(), 1, 1.0)
    tab.set_prob(s.information_state_string(), 0, 0.0)
    assert tab.action_probabilities(s) == {0: 0.0, 1: 1.0}
    assert "0b:  0=0 1=1" in str(tab)          # policy.cc:210-229: "key: " then " action=prob" per action
    # a Python Policy (python/policy.py's interface: action_probabilities(state, player_id)) judged on the device
    class AlwaysBet(pyspiel.Policy):
        def __init__(self):
            pyspiel.Policy.__init__(self)

        def action_probabilities(self, state, player_id=None):
            legal = state.legal_actions()
            return {a: (1.0 if a == max(legal) else 0.0) for a in legal}

    py_nc = pyspiel.nash_conv(kuhn, AlwaysBet())
    assert abs(py_nc - pyspiel.nash_conv(kuhn, pyspiel.PreferredActionPolicy([1, 0]))) < 1e-14
    assert len(pyspiel.ToTabularPolicy(kuhn, AlwaysBet())) == 12
    # solver policies are Policy objects too
    solver = pyspiel.CFRSolver(kuhn)
    solver.evaluate_and_update_policy(50)
    avg = solver.average_policy()
    assert isinstance(avg, pyspiel.TabularPolicy)
    assert pyspiel.exploitability(kuhn, avg) < 0.03


@pytest.mark.gpu
def test_tabular_best_response(pyspiel, oracle):
    """best_response.h:38-130 / python/pybind11/policy.cc:138-162 on the device: values and actions of the best
    response to the uniform policy and to a CFR average policy — the reference's known answers
    (exploitability_test.py: best response to uniform kuhn is worth 11/24 + ... ; NashConv = sum of BR values
    in a zero-sum game) and the oracle's judge."""
    kuhn = pyspiel.load_game("kuhn_poker")
    uniform = pyspiel.UniformRandomPolicy(kuhn)
    values = []
    for p in (0, 1):
        br = pyspiel.TabularBestResponse(kuhn, p, uniform)
        values.append(br.value(""))
        actions = br.get_best_response_actions()
        policy = br.get_best_response_policy()
        assert set(actions) == {k for k in uniform.policy_table() if (len(k) % 2 == 1) == (p == 0)}
        for k, a in actions.items():
            assert dict(policy.get_state_policy(k))[a] == 1.0 and sum(pr for _, pr in policy.get_state_policy(k)) == 1.0
    assert abs(sum(values) - pyspiel.nash_conv(kuhn, uniform)) < 1e-14          # zero-sum: NashConv = sum of BR values
    assert abs(sum(values) / 2 - 0.4583333333333335) < 1e-14
    # against the uniform policy: with a king facing a bet the responder calls, with a jack it folds; opening with a
    # king, betting and passing are worth the same 1.5 (bet: called half the time; pass: the opponent bets half the
    # time and is called) — a tie, which goes to the first action (best_response.cc:207-211 strict >)
    a0 = pyspiel.TabularBestResponse(kuhn, 0, uniform).get_best_response_actions()
    assert a0["2pb"] == 1 and a0["0pb"] == 0 and a0["2"] == 0
    # against the oracle's judge on a trained policy, then set_policy re-uses the object
    solver = pyspiel.CFRSolver(kuhn)
    solver.evaluate_and_update_policy(30)
    avg = solver.average_policy()
    br = pyspiel.TabularBestResponse(kuhn, 1, uniform)
    br.set_policy(avg)
    o = oracle.Solver(oracle.Game("kuhn_poker"), "cfr")
    o.iterate(30)
    want = (o.nash_conv() + sum(o.expected_returns())) / 1.0  # NashConv = sum_p (BR_p - EV_p); EVs sum to 0
    got = br.value("") + pyspiel.TabularBestResponse(kuhn, 0, avg).value("")
    assert abs(got - want) < 1e-12
    leduc = pyspiel.load_game("leduc_poker")
    vals = [pyspiel.TabularBestResponse(leduc, p, pyspiel.UniformPolicy()).value("") for p in (0, 1)]
    assert abs(sum(vals) / 2 - 2.373611111111111) < 1e-12


def test_pyspiel_alias_package_without_a_gpu():
    """`import pyspiel` (repository root on sys.path) is the reference's module name over pyspiel_hip: same objects,
    the game submodules importable as pyspiel.<game>, per-game virtual Game classes."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    had = sys.modules.pop("pyspiel", None)      # (other tests install a stand-in module under this name)
    try:
        pyspiel = importlib.import_module("pyspiel")
        from open_spiel_amd import pyspiel_hip
        assert pyspiel.load_game is pyspiel_hip.load_game and pyspiel.MCTSBot is pyspiel_hip.MCTSBot
        assert pyspiel.CFRSolver is pyspiel_hip.CFRSolver and pyspiel.State is pyspiel_hip.State
        ttt = importlib.import_module("pyspiel.tic_tac_toe")
        assert ttt is pyspiel.tic_tac_toe and ttt.CellState.CROSS == pyspiel_hip.tic_tac_toe.CellState.CROSS
        from pyspiel import leduc_poker, connect_four, kuhn_poker  # noqa: F401
        game = pyspiel.load_game("tic_tac_toe")
        assert isinstance(game, pyspiel.tic_tac_toe.TicTacToeGame)
        assert not isinstance(game, pyspiel.connect_four.ConnectFourGame)
        assert isinstance(pyspiel.load_game("leduc_poker(players=3)"), pyspiel.leduc_poker.LeducGame)
        assert not isinstance(3, pyspiel.tic_tac_toe.TicTacToeState)
        with pytest.raises(TypeError):
            pyspiel.tic_tac_toe.TicTacToeState()
    finally:
        sys.modules.pop("pyspiel", None)
        for k in [k for k in sys.modules if k.startswith("pyspiel.")]:
            sys.modules.pop(k)
        if had is not None:
            sys.modules["pyspiel"] = had


@pytest.mark.gpu
def test_per_game_state_classes_through_the_alias(pyspiel):
    """isinstance(state, pyspiel.<game>.<Game>State) as user code of the reference writes it
    (games_tic_tac_toe.cc:75, games_connect_four.cc:75, games_leduc_poker.cc:39)."""
    import importlib
    import sys
    had = sys.modules.pop("pyspiel", None)
    try:
        alias = importlib.import_module("pyspiel")
        s = alias.load_game("tic_tac_toe").new_initial_state()
        assert isinstance(s, alias.State) and isinstance(s, alias.tic_tac_toe.TicTacToeState)
        assert not isinstance(s, alias.leduc_poker.LeducState)
        s.apply_action(4)
        assert isinstance(s.clone(), alias.tic_tac_toe.TicTacToeState) and s.board()[4] == alias.tic_tac_toe.CellState.CROSS
        ls = alias.load_game("leduc_poker").new_initial_state()
        assert isinstance(ls, alias.leduc_poker.LeducState) and not isinstance(ls, alias.connect_four.ConnectFourState)
        assert isinstance(alias.load_game("connect_four").new_initial_state(), alias.connect_four.ConnectFourState)
    finally:
        sys.modules.pop("pyspiel", None)
        for k in [k for k in sys.modules if k.startswith("pyspiel.")]:
            sys.modules.pop(k)
        if had is not None:
            sys.modules["pyspiel"] = had


@pytest.mark.gpu
def test_python_and_cpp_bots_through_evaluate_bots(pyspiel):
    """python/tests/bot_test.py:33-44 (a C++ stock bot against a Python subclass of pyspiel.Bot through evaluate_bots) with
    fewer episodes, and the two other stock bots of bots.cc:184-196."""

    class UniformRandomBot(pyspiel.Bot):   # the shape of open_spiel/python/bots/uniform_random.py
        def __init__(self, player_id, rng):
            pyspiel.Bot.__init__(self)
            self._player_id, self._rng = player_id, rng

        def restart_at(self, state):
            pass

        def provides_policy(self):
            return True

        def step_with_policy(self, state):
            legal = state.legal_actions(self._player_id)
            if not legal:
                return [], pyspiel.INVALID_ACTION
            return [(a, 1 / len(legal)) for a in legal], int(self._rng.choice(legal))

        def step(self, state):
            return self.step_with_policy(state)[1]

    game = pyspiel.load_game("kuhn_poker")
    bots = [pyspiel.make_uniform_random_bot(0, 1234), UniformRandomBot(1, np.random.RandomState(4321))]
    results = np.array([pyspiel.evaluate_bots(game.new_initial_state(), bots, it) for it in range(1500)])
    np.testing.assert_allclose(results.mean(axis=0), [0.125, -0.125], atol=0.1)
    assert np.all(results.sum(axis=1) == 0)
    # the stateful bot checks on every inform_action that the run loop's state equals its own copy
    bots = [pyspiel.make_stateful_random_bot(game, 0, 7), pyspiel.make_policy_bot(game, 1, 8, pyspiel.UniformPolicy())]
    results = np.array([pyspiel.evaluate_bots(game.new_initial_state(), bots, it) for it in range(300)])
    assert np.all(results.sum(axis=1) == 0) and set(np.unique(results)) <= {-2.0, -1.0, 1.0, 2.0}


@pytest.mark.gpu
def test_cpp_mcts_bot_through_evaluate_bots(pyspiel):
    """python/tests/bot_test.py:52-69: one MCTSBot object in both seats, then a search inspected by hand."""
    game = pyspiel.load_game("tic_tac_toe")
    bots = [pyspiel.MCTSBot(game, pyspiel.RandomRolloutEvaluator(1, 0), 2.0, 100, 100, False, 42, False)] * 2
    results = np.array([pyspiel.evaluate_bots(game.new_initial_state(), bots, it) for it in range(10)])
    assert results.shape == (10, 2) and np.all(results.sum(axis=1) == 0)
    state = game.new_initial_state()
    node = bots[0].mcts_search(state)
    assert sum(c.explore_count for c in node.children) == 100 - 1 or sum(c.explore_count for c in node.children) == 100
    assert "explored" in node.children_str(state) or len(node.children_str(state)) > 0
    assert node.best_child().to_string(state)


@pytest.mark.gpu
def test_history_and_string_helpers_of_state(pyspiel):
    """pyspiel.cc:413-470: full_history (PlayerAction records), string_to_action (spiel.cc:432-439),
    is_initial_non_chance_state (spiel.cc:947-964), player_reward, and the Game-side sizes next to them."""
    game = pyspiel.load_game("tic_tac_toe")
    state = game.new_initial_state()
    assert state.string_to_action("x(1,1)") == 4 and state.string_to_action(0, "x(2,2)") == 8
    with pytest.raises(pyspiel.SpielError):
        state.string_to_action("no such move")
    for a in (4, 0, 8):
        state.apply_action(a)
    assert [(pa.player, pa.action) for pa in state.full_history()] == [(0, 4), (1, 0), (0, 8)]
    assert state.player_reward(0) == 0.0 and not state.is_mean_field_node()
    assert game.max_move_number() == 9 and game.max_history_length() == 9 and game.policy_tensor_shape() == [9]
    assert game.observation_tensor_layout() == pyspiel.TensorLayout.CHW
    with pytest.raises(pyspiel.SpielError):
        state.apply_actions([1, 2])               # a sequential game (spiel.h: ApplyActions is for simultaneous nodes)
    kuhn = pyspiel.load_game("kuhn_poker")
    s = kuhn.new_initial_state()
    assert not s.is_initial_non_chance_state()    # a chance node
    s.apply_action(0); s.apply_action(1)
    assert s.is_initial_non_chance_state()        # only chance outcomes so far, a player to move
    assert [pa.player for pa in s.full_history()] == [pyspiel.PlayerId.CHANCE] * 2
    s.apply_action(0)
    assert not s.is_initial_non_chance_state()
    assert kuhn.max_move_number() == kuhn.max_game_length() + kuhn.max_chance_nodes_in_history()


def test_load_game_with_a_parameter_dict_and_the_module_level_helpers(pyspiel):
    """pyspiel.cc:720-741 load_game(name, params), :168-172 game_parameters_from / to_string, :768-775 registered names,
    :811-815 sample_action — host-only, no GPU needed."""
    g = pyspiel.load_game("kuhn_poker", {"players": 3})
    assert str(g) == "kuhn_poker(players=3)" and g.num_players() == 3
    h = pyspiel.load_game("hex", {"board_size": 5, "swap": True})
    assert h.num_distinct_actions() == 26 and h.get_parameters()["swap"] is True
    assert str(pyspiel.load_game("tic_tac_toe", {})) == "tic_tac_toe()"
    with pytest.raises(pyspiel.SpielError):
        pyspiel.load_game("kuhn_poker", {"players": 1})
    params = pyspiel.game_parameters_from_string("leduc_poker(players=3,action_mapping=True)")
    assert params == {"name": "leduc_poker", "players": 3, "action_mapping": True}
    assert pyspiel.game_parameters_to_string(params) == "leduc_poker(action_mapping=True,players=3)"
    assert str(pyspiel.load_game(pyspiel.game_parameters_to_string(params))) == "leduc_poker(action_mapping=True,players=3)"
    assert set(pyspiel.registered_names()) == {"tic_tac_toe", "connect_four", "hex", "kuhn_poker", "leduc_poker"}
    assert (pyspiel.PlayerId.CHANCE, pyspiel.PlayerId.TERMINAL, pyspiel.PlayerId.SIMULTANEOUS, pyspiel.PlayerId.INVALID,
            pyspiel.PlayerId.MEAN_FIELD, pyspiel.PlayerId.DEFAULT_PLAYER_ID) == (-1, -4, -2, -3, -5, 0)   # spiel_globals.h:44-60
    outcomes = [(0, 0.25), (3, 0.5), (7, 0.25)]
    assert [pyspiel.sample_action(outcomes, z)[0] for z in (0.0, 0.2499, 0.25, 0.74, 0.75, 0.999)] == [0, 0, 3, 3, 7, 7]
    with pytest.raises(pyspiel.SpielError):
        pyspiel.sample_action(outcomes, 1.0)


@pytest.mark.gpu
def test_state_get_type(pyspiel):
    s = pyspiel.load_game("kuhn_poker").new_initial_state()
    assert s.get_type() == pyspiel.StateType.CHANCE
    s.apply_action(0); s.apply_action(1)
    assert s.get_type() == pyspiel.StateType.DECISION
    s.apply_action(0); s.apply_action(0)
    assert s.get_type() == pyspiel.StateType.TERMINAL


def test_reference_python_modules_import_over_the_pyspiel_alias(tmp_path):
    """Where the reference tree exists (the build container): its Python algorithm files for this path, and the Python-game
    modules they pull in at import (which build pyspiel.GameType / GameInfo objects and call register_game), import with
    `pyspiel` resolving to this repository's alias package — i.e. every module-level pyspiel name they touch exists.
    (absl-py and ml_collections are not installed here: a stand-in for absl and a namespace for the games package.)"""
    import os
    import subprocess
    import sys
    ref = os.environ.get("OSG_REFERENCE_ROOT", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "open_spiel", "python")):
        pytest.skip("no reference tree here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    absl = tmp_path / "absl"
    (absl / "testing").mkdir(parents=True)
    (absl / "__init__.py").write_text("")
    (absl / "logging.py").write_text("import logging as _l\ninfo = _l.info; warning = _l.warning; error = _l.error; debug = _l.debug\n")
    (absl / "flags.py").write_text("class _F:\n    pass\nFLAGS = _F()\ndef DEFINE_string(*a, **k): pass\n"
                                   "DEFINE_integer = DEFINE_float = DEFINE_bool = DEFINE_boolean = DEFINE_enum = DEFINE_list = DEFINE_string\n")
    (absl / "app.py").write_text("def run(main): main([])\n")
    (absl / "testing" / "__init__.py").write_text("")
    code = f"""
import sys, types, importlib
sys.path[:0] = [{root!r}, {ref!r}, {str(tmp_path)!r}]
import pyspiel
assert pyspiel.__name__ == "pyspiel" and "open_spiel_amd" in pyspiel.load_game.__module__
import open_spiel.python
games = types.ModuleType("open_spiel.python.games")          # (its __init__ imports chat_game -> ml_collections)
games.__path__ = [{os.path.join(ref, "open_spiel", "python", "games")!r}]
sys.modules["open_spiel.python.games"] = games
for name in ["games.kuhn_poker", "games.liars_poker", "games.iterated_prisoners_dilemma", "policy", "rl_environment",
             "vector_env", "observation", "bots.uniform_random", "algorithms.cfr", "algorithms.cfr_br",
             "algorithms.exploitability", "algorithms.best_response", "algorithms.get_all_states", "algorithms.mcts",
             "algorithms.external_sampling_mccfr", "algorithms.outcome_sampling_mccfr", "algorithms.expected_game_score",
             "algorithms.evaluate_bots", "algorithms.generate_playthrough", "algorithms.fictitious_play",
             "algorithms.policy_utils"]:
    importlib.import_module("open_spiel.python." + name)
registered = sys.modules["open_spiel_amd.pyspiel_hip"]._python_games
assert "python_kuhn_poker" in registered
print("IMPORTED")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0 and "IMPORTED" in r.stdout, (r.stdout + r.stderr)[-3000:]
