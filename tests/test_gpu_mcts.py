"""MCTS on the device vs the oracle's MCTSBot.

RNG-stream parity with the reference is impossible by construction (abseil's
distributions are unpinned), so parity is established in two ways:
  * replay: the oracle's MCTSBot (pinned by the reference's known-answer tests,
    tests/test_oracle_known_answers.py) runs with every draw taken from the
    device's counter streams; visit counts, total rewards, proven outcomes and
    the chosen action must then be IDENTICAL, root by root;
  * the reference's own behavioural tests (mcts_test.cc:126-155 solver answers).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import open_spiel_amd as osa
    return osa.Context(0)


def _roots(oracle, ctx, game, n, seed, max_stop, min_stop=0):
    import torch
    import open_spiel_amd as osa
    og = oracle.Game(game)
    rng = np.random.default_rng(seed)
    stop = rng.integers(min_stop, max_stop + 1, n).astype(np.int32)
    rec = og.random_playouts(seed, n, stop=stop)
    roots = osa.StateBatch(ctx, game, n)
    for t in range(og.max_plies):
        if (rec["actions"][:, t] < 0).all():
            break
        roots.apply_actions(torch.from_numpy(rec["actions"][:, t].astype(np.int32)))
    hists = [rec["actions"][i][rec["actions"][i] >= 0].tolist() for i in range(n)]
    return og, roots, hists


def _oracle_state(og, hist):
    s = og.new_initial_state()
    for a in hist:
        s.apply_action(int(a))
    return s


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("game,n,sims,n_rollouts,solve,max_stop", [
    ("tic_tac_toe", 96, 200, 3, True, 7),
    ("tic_tac_toe", 64, 64, 1, False, 5),
    ("connect_four", 64, 150, 2, True, 30),
    ("hex(board_size=5)", 48, 120, 1, True, 18),
    ("hex(board_size=9)", 24, 200, 1, False, 40),
    ("kuhn_poker", 48, 100, 2, False, 4),
    ("leduc_poker", 48, 150, 1, False, 8),
    ("hex(board_size=5)", 32, 300, 3, False, 24),
    ("hex(board_size=4,swap=True)", 32, 100, 2, True, 6),
    ("hex(num_cols=3,num_rows=4)", 32, 150, 1, True, 6),
    ("hex", 8, 300, 1, False, 60),
    ("tic_tac_toe", 32, 150, 70, True, 4),
    ("tic_tac_toe", 12, 1000, 20, True, 0),   # BASELINE configs[0]: MCTSBot(RandomRolloutEvaluator(20), 1000 sims) from the start
    ("kuhn_poker(players=3)", 48, 120, 1, False, 6),
    # small hex boards: the tree reaches finished games all the time, so the wave kernel's IsTerminal-at-first-visit
    # (a flood of the last stone's group, no edge labels on the way down) is exercised on every simulation
    ("hex(board_size=3)", 32, 400, 1, True, 4),
    ("hex(board_size=3)", 32, 300, 2, False, 5),
    ("hex(board_size=2)", 16, 60, 1, True, 2),
    ("hex(num_cols=4,num_rows=2)", 24, 200, 1, True, 3),
    # (boards with a single row or column go to the generic instantiation; they are not searched here: with the
    # reference's `else if` between the two edges black can never win on them, so a filled board is a state that is
    # not terminal and has no legal action — a random playout never ends, in the reference as well)
])
def test_mcts_replay_parity(oracle, ctx, game, n, sims, n_rollouts, solve, max_stop, layout):
    min_stop = (3 if "players=3" in game else 2) if "poker" in game else 0  # past the private deals
    og, roots, hists = _roots(oracle, ctx, game, n, 17, max_stop, min_stop)
    seed, offset = 0xFEED5EED, 12345
    res = roots.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=n_rollouts, solve=solve, seed=seed,
                            index_offset=offset, layout=layout)
    best = res["best_action"].cpu().numpy()
    visits = res["child_visits"].cpu().numpy()
    reward = res["child_reward"].cpu().numpy()
    outcome = res["child_outcome"].cpu().numpy()
    stats = res["root_stats"].cpu().numpy()
    checked = 0
    for i in range(n):
        st = _oracle_state(og, hists[i])
        if st.is_chance_node():
            continue  # MCTSBot is never asked to move at a chance node
        want = st.mcts_search(2.0, sims, n_rollouts, 4096, solve, 0, counter_root=offset + i, counter_seed=seed,
                              counter_layout=layout)
        assert stats[i, 0] == want["root_visits"], f"{game} root {i}: root visits"
        acts = want["children"][:, 0].astype(int)
        got_children = np.nonzero(outcome[i] != 3)[0]
        assert sorted(acts.tolist()) == got_children.tolist(), f"{game} root {i}: children"
        for a, cnt, tot, out in want["children"]:
            a = int(a)
            assert visits[i, a] == cnt, f"{game} root {i} action {a}: visits {visits[i, a]} vs {cnt}"
            assert reward[i, a] == tot, f"{game} root {i} action {a}: reward {reward[i, a]} vs {tot}"
            if solve:
                assert (outcome[i, a] == 2) == np.isnan(out)
                if not np.isnan(out):
                    assert outcome[i, a] == out
        if len(acts):
            assert best[i] == want["best_action"], f"{game} root {i}: best action"
        if solve:
            assert np.isnan(stats[i, 2]) == np.isnan(want["root_outcome"])
            if not np.isnan(want["root_outcome"]):
                assert stats[i, 2] == want["root_outcome"]
        checked += 1
    assert checked >= n // 3


@pytest.mark.parametrize("game,n,sims,n_rollouts,solve,max_stop", [
    ("hex(board_size=13)", 12, 120, 1, False, 60),            # 169 actions: six-word planes and masks
    ("hex(board_size=13)", 8, 150, 1, True, 160),             # late positions: the solver proves wins
    ("hex(board_size=15)", 6, 80, 1, False, 100),             # 225 actions: the widest that fits eight bits
    ("hex(board_size=19)", 6, 60, 1, False, 120),             # 361 actions: the nine-bit action / child-count fields
    ("hex(num_cols=17,num_rows=19,swap=True)", 6, 60, 1, False, 3),   # the swap action (id 323) as a child
    ("connect_four(rows=9,columns=12)", 16, 100, 2, True, 40),        # 120 board bits: two plane words per colour
    ("leduc_poker(players=4)", 24, 100, 1, False, 10),        # the five-word record
    ("leduc_poker(players=6)", 16, 80, 2, False, 14),
])
def test_mcts_replay_parity_on_the_wide_games(oracle, ctx, game, n, sims, n_rollouts, solve, max_stop):
    """The games beyond the four-word mask / the two-word records are searched by the lane-per-root kernel (layout 1, what
    layout 0 picks; layout 2 serves of them the hex boards without the swap rule — the test below — and refuses the others):
    the same replay parity as above — visits, rewards, outcomes, best action, root by root."""
    import open_spiel_amd as osa
    players = int(game.split("players=")[1].rstrip(")")) if "players=" in game else 0
    og, roots, hists = _roots(oracle, ctx, game, n, 23, max_stop, players)   # (poker: past the private deals)
    if not game.startswith("hex") or "swap=True" in game:
        with pytest.raises(osa.OsgError):
            roots.mcts_search(uct_c=2.0, max_simulations=4, n_rollouts=1, seed=1, layout=2)
    seed, offset = 0xFEED5EED, 777
    # (layout 0 = automatic picks the lane-per-root kernel for all of these but the hex boards without the swap rule)
    res = roots.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=n_rollouts, solve=solve, seed=seed,
                            index_offset=offset, layout=0 if (not game.startswith("hex") or "swap=True" in game) else 1)
    best = res["best_action"].cpu().numpy()
    visits = res["child_visits"].cpu().numpy()
    reward = res["child_reward"].cpu().numpy()
    outcome = res["child_outcome"].cpu().numpy()
    stats = res["root_stats"].cpu().numpy()
    checked = 0
    for i in range(n):
        st = _oracle_state(og, hists[i])
        if st.is_chance_node():
            continue
        want = st.mcts_search(2.0, sims, n_rollouts, 4096, solve, 0, counter_root=offset + i, counter_seed=seed,
                              counter_layout=1)
        assert stats[i, 0] == want["root_visits"], f"{game} root {i}: root visits"
        acts = want["children"][:, 0].astype(int)
        assert sorted(acts.tolist()) == np.nonzero(outcome[i] != 3)[0].tolist(), f"{game} root {i}: children"
        for a, cnt, tot, out in want["children"]:
            a = int(a)
            assert visits[i, a] == cnt and reward[i, a] == tot, f"{game} root {i} action {a}"
            if solve and not np.isnan(out):
                assert outcome[i, a] == out
        if len(acts):
            assert best[i] == want["best_action"], f"{game} root {i}: best action"
        checked += 1
    assert checked >= n // 3


@pytest.mark.parametrize("game,n,sims,n_rollouts,solve,max_stop,puct,max_nodes", [
    ("hex(board_size=12)", 10, 150, 1, False, 50, False, 0),     # 144 cells: three cell sets, six plane words
    ("hex(board_size=13)", 10, 200, 1, False, 60, False, 0),     # 169 cells
    ("hex(board_size=13)", 8, 200, 1, True, 165, False, 0),      # late positions: finished games in the tree, the solver
    ("hex(board_size=13)", 6, 150, 2, True, 120, True, 0),       # PUCT
    ("hex(board_size=13)", 6, 400, 1, False, 150, False, 60),    # few empty cells, deep trees, garbage collection
    ("hex(board_size=14)", 8, 120, 1, False, 80, False, 0),      # 196 cells: four cell sets
    ("hex(board_size=15)", 8, 120, 1, False, 100, False, 0),     # 225 cells
    ("hex(board_size=16)", 6, 100, 1, True, 200, False, 0),      # 256 cells: a child count that needs the ninth bit
    ("hex(board_size=16)", 6, 300, 1, False, 0, False, 0),       # ... from the empty board: 256 children of the root
    ("hex(board_size=17)", 6, 80, 1, False, 140, False, 0),      # 289 cells: six cell sets, five of them used
    ("hex(board_size=19)", 6, 80, 1, False, 120, False, 0),      # 361 cells: nine-bit actions, 41-bit fill keys
    ("hex(board_size=19)", 4, 400, 1, False, 0, False, 0),       # 361 children of the root, all visited once and more
    ("hex(board_size=19)", 6, 150, 3, True, 340, True, 0),       # nearly full boards
    ("hex(num_cols=19,num_rows=7)", 8, 150, 1, True, 60, False, 0),     # 133 cells, long rows
    ("hex(num_cols=5,num_rows=30)", 8, 150, 1, True, 80, False, 0),     # 150 cells, long columns
    ("hex(num_cols=31,num_rows=5)", 6, 200, 1, True, 100, False, 0),    # 155 cells: the widest row with a device layout
])
def test_mcts_wave_layout_on_the_boards_above_128_cells(oracle, ctx, game, n, sims, n_rollouts, solve, max_stop, puct,
                                                        max_nodes):
    """Round 6: the wavefront-per-root search (layout 2) on hex boards of more than 128 cells — the position as 3 / 4 / 6
    64-cell sets per colour in scalar registers, as many child slots per lane, the playout as one random fill ordered by
    (fill_key, cell): every root's children, visits, rewards, proven outcomes and best action against the oracle's replay
    of the same streams (counter_layout=2)."""
    og, roots, hists = _roots(oracle, ctx, game, n, 29, max_stop, 0)
    seed, offset = 0xFEED5EED, 4242
    kw = dict(uct_c=2.0, max_simulations=sims, n_rollouts=n_rollouts, solve=solve, seed=seed, index_offset=offset, layout=2)
    if puct:
        kw["puct"] = True
    if max_nodes:
        kw["max_nodes"] = max_nodes
    res = roots.mcts_search(**kw)
    auto = roots.mcts_search(**dict(kw, layout=0))       # what "automatic" picks on these boards
    assert bool((auto["child_visits"] == res["child_visits"]).all()) and bool((auto["child_reward"] == res["child_reward"]).all())
    best = res["best_action"].cpu().numpy()
    visits = res["child_visits"].cpu().numpy()
    reward = res["child_reward"].cpu().numpy()
    outcome = res["child_outcome"].cpu().numpy()
    stats = res["root_stats"].cpu().numpy()
    for i in range(n):
        st = _oracle_state(og, hists[i])
        want = st.mcts_search(2.0, sims, n_rollouts, -max_nodes if max_nodes else 4096, solve, 0, counter_root=offset + i,
                              counter_seed=seed, counter_layout=2, puct=puct)
        assert stats[i, 0] == want["root_visits"], f"{game} root {i}: root visits"
        acts = want["children"][:, 0].astype(int)
        assert sorted(acts.tolist()) == np.nonzero(outcome[i] != 3)[0].tolist(), f"{game} root {i}: children"
        for a, cnt, tot, out in want["children"]:
            a = int(a)
            assert visits[i, a] == cnt and reward[i, a] == tot, f"{game} root {i} action {a}: {visits[i, a]} / {cnt}, {reward[i, a]} / {tot}"
            if solve:
                assert (outcome[i, a] == 2) == np.isnan(out)
                if not np.isnan(out):
                    assert outcome[i, a] == out
        if len(acts):
            assert best[i] == want["best_action"], f"{game} root {i}: best action"
        if solve:
            assert np.isnan(stats[i, 2]) == np.isnan(want["root_outcome"])


@pytest.mark.parametrize("uct_c", [0.0, 1e-9, 1e-4, 0.37, 1e30])
@pytest.mark.parametrize("game,n,sims,n_rollouts,solve,max_stop", [
    ("hex(board_size=5)", 24, 400, 2, True, 10),
    ("hex(board_size=9)", 12, 500, 1, False, 30),
    ("hex(board_size=13)", 6, 300, 1, False, 120),
])
def test_wave_search_arg_max_is_exact_where_single_precision_cannot_decide(oracle, ctx, game, n, sims, n_rollouts, solve,
                                                                           max_stop, uct_c):
    """Round 6: the wave-per-root search forms every child's UCT value in fp32 first and takes the fp64 divisions and
    square root only where that cannot name the maximum.  Exploration constants that put the children's values closer
    than single precision resolves (1e-9, 1e-4), remove the exploration term (0: values are the mean returns, full of
    near ties between different statistics), overflow fp32 (1e30) or are ordinary (0.37): the trees must still be the
    oracle's, visit for visit."""
    og, roots, hists = _roots(oracle, ctx, game, n, 41, max_stop, 0)
    seed, offset = 0xABCDEF, 99
    res = roots.mcts_search(uct_c=uct_c, max_simulations=sims, n_rollouts=n_rollouts, solve=solve, seed=seed,
                            index_offset=offset, layout=2)
    visits = res["child_visits"].cpu().numpy()
    reward = res["child_reward"].cpu().numpy()
    best = res["best_action"].cpu().numpy()
    stats = res["root_stats"].cpu().numpy()
    for i in range(n):
        st = _oracle_state(og, hists[i])
        want = st.mcts_search(uct_c, sims, n_rollouts, 4096, solve, 0, counter_root=offset + i, counter_seed=seed,
                              counter_layout=2)
        assert stats[i, 0] == want["root_visits"], f"{game} c={uct_c} root {i}: root visits"
        for a, cnt, tot, out in want["children"]:
            a = int(a)
            assert visits[i, a] == cnt and reward[i, a] == tot, f"{game} c={uct_c} root {i} action {a}: {visits[i, a]} / {cnt}"
        if len(want["children"]):
            assert best[i] == want["best_action"], f"{game} c={uct_c} root {i}: best action"


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("game,n,sims,n_rollouts,solve,max_stop", [
    ("tic_tac_toe", 48, 150, 2, True, 5), ("connect_four", 32, 120, 1, False, 20),
    ("hex(board_size=5)", 32, 200, 1, True, 14), ("leduc_poker", 32, 100, 1, False, 7),
])
def test_mcts_puct_replay_parity(oracle, ctx, game, n, sims, n_rollouts, solve, max_stop, layout):
    """ChildSelectionPolicy::PUCT (mcts.cc:103-112) with the rollout evaluator's uniform prior."""
    min_stop = 2 if "poker" in game else 0
    og, roots, hists = _roots(oracle, ctx, game, n, 23, max_stop, min_stop)
    seed, offset = 0xABCDEF, 500
    res = roots.mcts_search(uct_c=1.5, max_simulations=sims, n_rollouts=n_rollouts, solve=solve, seed=seed,
                            index_offset=offset, layout=layout, puct=True)
    best = res["best_action"].cpu().numpy()
    visits = res["child_visits"].cpu().numpy()
    reward = res["child_reward"].cpu().numpy()
    checked = 0
    for i in range(n):
        st = _oracle_state(og, hists[i])
        if st.is_chance_node():
            continue
        want = st.mcts_search(1.5, sims, n_rollouts, 4096, solve, 0, counter_root=offset + i, counter_seed=seed,
                              counter_layout=layout, puct=True)
        for a, cnt, tot, _ in want["children"]:
            assert visits[i, int(a)] == cnt and reward[i, int(a)] == tot, f"{game} root {i} action {int(a)}"
        if len(want["children"]):
            assert best[i] == want["best_action"]
        checked += 1
    assert checked >= n // 3


def _ttt_batch(ctx, moves, n=4):
    import torch
    import open_spiel_amd as osa
    b = osa.StateBatch(ctx, "tic_tac_toe", n)
    for a in moves:
        b.apply_actions(torch.full((n,), a, dtype=torch.int32))
    return b


@pytest.mark.parametrize("layout", [1, 2])
def test_solver_known_answers(ctx, layout):
    """mcts_test.cc:126-155 (UCT_C=2, RandomRolloutEvaluator(20, 42), 10000 simulations,
    solve=true): the three MCTS-Solver positions of the reference's own test."""
    # MCTSTest_SolveDraw: "x(1,1) o(0,0) x(2,2)" -> "o..\n.x.\n..x", o to move, proven draw
    b = _ttt_batch(ctx, [4, 0, 8])
    r = b.mcts_search(uct_c=2.0, max_simulations=10000, n_rollouts=20, solve=True, seed=42, layout=layout)
    stats = r["root_stats"].cpu().numpy()
    outcome = r["child_outcome"].cpu().numpy()
    best = r["best_action"].cpu().numpy()
    assert (stats[:, 2] == 0).all()                       # root->outcome[root->player] == 0
    for i in range(len(best)):
        kids = outcome[i][outcome[i] != 3]
        assert (kids <= 0).all() or (kids == 2).any()     # no winning moves among proven children
        assert not ((kids == 1).any())
        assert outcome[i, best[i]] == 0                   # best.outcome[best.player] == 0
        assert best[i] in (6, 2)                          # o(2,0) or o(0,2); all others lose
    # MCTSTest_SolveLoss: "... o(0,1) x(0,2)" -> "oox\n.x.\n..x": every move loses
    b = _ttt_batch(ctx, [4, 0, 8, 1, 2])
    r = b.mcts_search(uct_c=2.0, max_simulations=10000, n_rollouts=20, solve=True, seed=42, layout=layout)
    stats = r["root_stats"].cpu().numpy()
    outcome = r["child_outcome"].cpu().numpy()
    assert (stats[:, 2] == -1).all()
    for i in range(outcome.shape[0]):
        kids = outcome[i][outcome[i] != 3]
        assert len(kids) == 4 and (kids == -1).all()
    # MCTSTest_SolveWin: "x(0,1) o(2,2)" -> ".x.\n...\n..o": x wins, best move x(0,2)
    b = _ttt_batch(ctx, [1, 8])
    r = b.mcts_search(uct_c=2.0, max_simulations=10000, n_rollouts=20, solve=True, seed=42, layout=layout)
    stats = r["root_stats"].cpu().numpy()
    outcome = r["child_outcome"].cpu().numpy()
    best = r["best_action"].cpu().numpy()
    assert (stats[:, 2] == 1).all()
    assert (best == 2).all()
    assert (outcome[np.arange(len(best)), best] == 1).all()


@pytest.mark.parametrize("layout", [1, 2])
def test_search_is_independent_of_batch_position_and_sharding(ctx, oracle, layout):
    """Root i's search depends only on (seed, index_offset + i): the same roots searched as
    one batch or as two shards give identical statistics (multi-GPU sharding by root index)."""
    import torch
    og, roots, _ = _roots(oracle, ctx, "hex(board_size=5)", 64, 5, 10)
    whole = roots.mcts_search(max_simulations=80, seed=9, index_offset=0, layout=layout)
    lo = roots.gather(torch.arange(0, 32)).mcts_search(max_simulations=80, seed=9, index_offset=0, layout=layout)
    hi = roots.gather(torch.arange(32, 64)).mcts_search(max_simulations=80, seed=9, index_offset=32, layout=layout)
    for key in ("best_action", "child_visits", "child_reward"):
        got = torch.cat([lo[key], hi[key]]).cpu().numpy()
        np.testing.assert_array_equal(got, whole[key].cpu().numpy())
    del og


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("game,n,sims,max_nodes,solve,max_stop", [
    ("connect_four", 48, 400, 40, False, 20),        # collects every few simulations, the limit climbs
    ("connect_four", 32, 600, 200, True, 24),
    ("tic_tac_toe", 48, 300, 25, True, 4),
    ("hex(board_size=5)", 32, 500, 120, False, 12),
    ("hex(board_size=9)", 12, 400, 600, False, 30),
    ("leduc_poker", 32, 400, 30, False, 7),           # chance nodes in the tree
    ("kuhn_poker", 32, 200, 8, False, 3),
])
def test_mcts_garbage_collection_replay_parity(oracle, ctx, game, n, sims, max_nodes, solve, max_stop, layout):
    """MCTSBot's node budget (mcts.cc:441-482): when nodes_ >= max_nodes_ every node visited fewer than
    gc_limit_ times loses its children, gc_limit_ adapts (x1.25 / x0.9, at least 5), cleared nodes are
    expanded again when the search returns to them.  The oracle's MCTSBot at the same max_nodes_ with every
    draw from the device's counter streams must produce IDENTICAL root statistics — any difference in when
    a collection happens, what it removes or how the limit moves changes the visit counts."""
    min_stop = 2 if "poker" in game else 0
    og, roots, hists = _roots(oracle, ctx, game, n, 31, max_stop, min_stop)
    seed, offset = 0x6C6C6563, 777
    res = roots.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=1, solve=solve, seed=seed,
                            index_offset=offset, layout=layout, max_nodes=max_nodes)
    free = roots.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=1, solve=solve, seed=seed,
                             index_offset=offset, layout=layout)
    best = res["best_action"].cpu().numpy()
    visits = res["child_visits"].cpu().numpy()
    reward = res["child_reward"].cpu().numpy()
    outcome = res["child_outcome"].cpu().numpy()
    stats = res["root_stats"].cpu().numpy()
    checked = differs = 0
    for i in range(n):
        st = _oracle_state(og, hists[i])
        if st.is_chance_node() or st.is_terminal():
            continue
        want = st.mcts_search(2.0, sims, 1, -max_nodes, solve, 0, counter_root=offset + i, counter_seed=seed,
                              counter_layout=layout)
        assert stats[i, 0] == want["root_visits"], f"{game} root {i}: root visits"
        got_children = np.nonzero(outcome[i] != 3)[0]
        assert sorted(want["children"][:, 0].astype(int).tolist()) == got_children.tolist(), f"{game} root {i}: children"
        for a, cnt, tot, out in want["children"]:
            a = int(a)
            assert visits[i, a] == cnt, f"{game} root {i} action {a}: visits {visits[i, a]} vs {cnt}"
            assert reward[i, a] == tot, f"{game} root {i} action {a}: reward {reward[i, a]} vs {tot}"
        if len(want["children"]):
            assert best[i] == want["best_action"], f"{game} root {i}: best action"
        differs += int((visits[i] != free["child_visits"][i].cpu().numpy()).any())
        checked += 1
    assert checked >= n // 3
    # the budget really bit: with it the searches differ from the unconstrained ones
    assert differs >= checked // 4, f"{game}: garbage collection changed only {differs} of {checked} searches"


def test_hex_fill_playout_matches_sequential_random_play(ctx):
    """The wave kernel's hex playout (one random fill of the board, keyed order) and the generic
    sequential random playout (k_rollout) estimate the same quantity: black's expected return under
    uniformly random play from the empty board.  262 144 playouts each; 5 sigma ~ 0.015."""
    import open_spiel_amd as osa
    n, r = 4096, 64
    roots = osa.StateBatch(ctx, "hex(board_size=5)", n)
    seq = roots.rollout(2024, r)                      # [n, 2] sums over r playouts
    est_seq = float(seq[:, 0].sum()) / (n * r)
    # two simulations: the first evaluates the root, the second a uniformly random child (all tie)
    res = roots.mcts_search(uct_c=2.0, max_simulations=2, n_rollouts=r, seed=99, layout=2)
    visits = res["child_visits"].cpu().numpy()
    reward = res["child_reward"].cpu().numpy()
    assert (visits.sum(1) == 1).all()
    first_move = visits.argmax(1)
    counts = np.bincount(first_move, minlength=25)
    assert counts.min() > n / 25 * 0.6 and counts.max() < n / 25 * 1.4, "the tie-break is not uniform over the cells"
    est_fill = reward.sum() / n                      # child total = mean of its r playouts (black's return)
    assert abs(est_fill - est_seq) < 0.015, (est_fill, est_seq)
    assert 0.0 < est_seq < 0.5, "black moves first and gets 13 of 25 cells: a modest edge"


def test_one_row_hex_boards_are_refused_instead_of_never_ending(ctx):
    """hex with a single row or column: the reference's `else if` between a colour's two edges (hex.cc:122-126,
    146-150) lets one colour never win there, so a filled board is not terminal and has no legal action — a random
    playout would never end.  The entry points that play out say so instead of hanging the device."""
    import open_spiel_amd as osa
    for game in ("hex(num_cols=5,num_rows=1)", "hex(num_cols=1,num_rows=4)"):
        roots = osa.StateBatch(ctx, game, 4)
        assert roots.legal_actions_mask().any()  # the rules themselves are served
        with pytest.raises(osa.OsgError, match="single row or column"):
            roots.mcts_search(uct_c=2.0, max_simulations=8, n_rollouts=1, seed=1)
        with pytest.raises(osa.OsgError, match="single row or column"):
            roots.rollout(seed=1, n_rollouts=2)


def test_a_playout_from_a_record_the_rules_cannot_finish_is_cut_off(ctx):
    """An uploaded connect_four record with all 42 cells taken but the result flags cleared is neither terminal nor
    has a legal column: a random playout from it would spin for ever.  The playout loops stop after
    kMaxPlayoutPlies moves (Returns() of a running game: zeros), so rollouts and both search layouts come back."""
    import numpy as np
    import open_spiel_amd as osa
    n = 8
    b = osa.StateBatch(ctx, "connect_four", n)
    w = b.raw_words()
    full = sum(0x3F << (7 * c) for c in range(7))          # six cells in each of the seven columns
    x = sum(1 << (7 * c + r) for c in range(7) for r in range(6) if (r // 2 + c) % 2 == 0)  # some split of them
    w[0, :] = x                                              # plane 0: x stones, flags byte left at zero
    w[1, :] = full & ~x
    b.load_raw_words(w)
    assert not b.is_terminal().any() and not b.legal_actions_mask().any()
    total, steps = b.rollout(seed=3, n_rollouts=2, want_steps=True)
    assert float(total.abs().sum()) == 0.0
    for layout in (1, 2):
        res = b.mcts_search(uct_c=2.0, max_simulations=4, n_rollouts=1, seed=1, layout=layout)
        assert res["root_stats"].shape[0] == n
