"""Parity of the device CFR / CFR+ / external-sampling MCCFR against the CPU oracle.

Bar (BASELINE.json north_star): CFR average-policy probabilities within 1e-6 of
the reference.  The device kernels add the same terms in the same order as the
reference's recursion (cfr.cc:331-408), so the tables are in fact compared at
1e-12 here; the 1e-6 bar is asserted on the average policy.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# SURVEY.md Appendix B / integration_tests/api_test.py:75-101
TREE_SIZES = {
    "kuhn_poker": (58, 4, 24, 30, 12, 2),
    "kuhn_poker(players=3)": (617, 17, 288, 312, 48, 2),
    "leduc_poker": (9457, 157, 3780, 5520, 936, 3),
}


@pytest.fixture(scope="module")
def ctx():
    import open_spiel_amd as osa
    return osa.Context(0)


def _by_key(t):
    return {k: i for i, k in enumerate(t["keys"])}


def _compare_tables(dev, orc, atol, what):
    """dev: TabularSolver.tables(); orc: oracle Solver.tables() (rows sorted by key)."""
    assert sorted(dev["keys"]) == sorted(orc["keys"]), what
    d_idx = _by_key(dev)
    worst = 0.0
    for j, k in enumerate(orc["keys"]):
        i = d_idx[k]
        n = int(orc["nact"][j])
        assert int(dev["nact"][i]) == n
        assert dev["legal"][i, :n].tolist() == orc["legal"][j, :n].tolist()
        for name in ("regrets", "cum_policy", "cur_policy", "avg_policy"):
            diff = np.abs(dev[name][i, :n] - orc[name][j, :n]).max()
            worst = max(worst, diff)
            assert diff <= atol, f"{what}: {name} at {k!r}: {dev[name][i, :n]} vs {orc[name][j, :n]}"
    return worst


@pytest.mark.parametrize("game", list(TREE_SIZES))
def test_tree_census_and_keys(oracle, ctx, game):
    import open_spiel_amd as osa
    s = osa.TabularSolver(ctx, game)
    got = (s.num_histories, s.num_chance, s.num_decision, s.num_terminal, s.num_infostates, s.amax)
    assert got == TREE_SIZES[game]
    og = oracle.Game(game)
    o = oracle.Solver(og, "cfr")
    assert sorted(s.tables()["keys"]) == sorted(o.tables(s.amax)["keys"])
    t = s.tables()
    # CFRInfoStateValues initial state: regrets 0, cumulative policy 0, current policy uniform
    assert not t["regrets"].any() and not t["cum_policy"].any()
    for i, n in enumerate(t["nact"]):
        assert t["cur_policy"][i, :n].tolist() == [1.0 / n] * n


@pytest.mark.parametrize("game,kind,kwargs,checkpoints", [
    ("kuhn_poker", "cfr", {}, [1, 2, 10, 100, 300]),
    ("kuhn_poker", "cfr_plus", dict(linear_averaging=True, regret_matching_plus=True), [1, 5, 200]),
    ("kuhn_poker", "cfr_simultaneous", dict(alternating_updates=False), [1, 2, 50]),
    ("kuhn_poker(players=3)", "cfr", {}, [1, 10, 40]),
    ("kuhn_poker(players=3)", "cfr_plus", dict(linear_averaging=True, regret_matching_plus=True), [10]),
    ("leduc_poker", "cfr", {}, [1, 2, 10]),
    ("leduc_poker", "cfr_plus", dict(linear_averaging=True, regret_matching_plus=True), [5]),
    # the general level-synchronous kernel on the small trees too (default there: the all-in-LDS kernel)
    ("kuhn_poker", "cfr", dict(general_kernel=True), [1, 2, 10, 100]),
    ("kuhn_poker", "cfr_plus", dict(linear_averaging=True, regret_matching_plus=True, general_kernel=True), [1, 50]),
    ("kuhn_poker", "cfr_simultaneous", dict(alternating_updates=False, general_kernel=True), [1, 20]),
    ("kuhn_poker(players=3)", "cfr", dict(general_kernel=True), [1, 10]),
    ("kuhn_poker(players=4)", "cfr", {}, [1, 6]),
    ("leduc_poker", "cfr", dict(general_kernel=True), [1, 4]),               # leduc default = path kernel, global memory
    ("leduc_poker", "cfr_simultaneous", dict(alternating_updates=False), [1, 3]),
    # full-grid phase kernels (the default beyond 65536 histories), forced on small and medium trees
    ("kuhn_poker", "cfr", dict(general_kernel="grid"), [1, 2, 30]),
    ("leduc_poker", "cfr", dict(general_kernel="grid"), [1, 5]),
    ("leduc_poker", "cfr_plus", dict(linear_averaging=True, regret_matching_plus=True, general_kernel="grid"), [4]),
    ("kuhn_poker(players=3)", "cfr_simultaneous", dict(alternating_updates=False, general_kernel="grid"), [6]),
    ("kuhn_poker(players=3)", "cfr_simultaneous", dict(alternating_updates=False), [1, 12]),
    # leduc's default since round 3 is one workgroup per deal subtree (k_cfr_split); the single-workgroup path kernel
    # it replaced stays checked, and the split kernel is also named explicitly
    ("leduc_poker", "cfr", dict(general_kernel="path"), [1, 3]),
    ("leduc_poker", "cfr", dict(general_kernel="split"), [1, 2, 7, 40]),
    ("leduc_poker", "cfr_plus", dict(linear_averaging=True, regret_matching_plus=True, general_kernel="split"), [1, 12]),
    ("leduc_poker", "cfr_simultaneous", dict(alternating_updates=False, general_kernel="split"), [1, 6]),
])
def test_cfr_tables_match_the_oracle(oracle, ctx, game, kind, kwargs, checkpoints):
    import open_spiel_amd as osa
    og = oracle.Game(game)
    o = oracle.Solver(og, kind)
    s = osa.TabularSolver(ctx, game, **kwargs)
    done = 0
    for cp in checkpoints:
        o.iterate(cp - done)
        s.evaluate_and_update_policy(cp - done)  # one launch, all iterations inside
        done = cp
        assert s.iteration == cp
        worst = _compare_tables(s.tables(), o.tables(s.amax), 1e-12, f"{game} {kind} after {cp} iterations")
        d = s.tables()
        # the north-star bar, stated explicitly: average-policy probabilities within 1e-6
        ot = o.tables(s.amax)
        di = _by_key(d)
        for j, k in enumerate(ot["keys"]):
            assert np.abs(d["avg_policy"][di[k]] - ot["avg_policy"][j]).max() <= 1e-6
        del worst


def test_iterating_one_by_one_equals_one_launch(ctx):
    import open_spiel_amd as osa
    a = osa.TabularSolver(ctx, "kuhn_poker")
    b = osa.TabularSolver(ctx, "kuhn_poker")
    a.evaluate_and_update_policy(25)
    for _ in range(25):
        b.evaluate_and_update_policy(1)
    ta, tb = a.tables(), b.tables()
    for name in ("regrets", "cum_policy", "cur_policy"):
        np.testing.assert_array_equal(ta[name], tb[name])
    a.reset()
    assert a.iteration == 0 and not a.tables()["regrets"].any()


@pytest.mark.parametrize("kwargs", [{}, dict(linear_averaging=True, regret_matching_plus=True),
                                    dict(alternating_updates=False)])
def test_leduc_split_kernel_is_bit_identical_with_the_single_workgroup_kernel(ctx, kwargs):
    """One workgroup per deal subtree, every workgroup folding its infostates' terms itself in DFS order: the same
    additions in the same order as the single-workgroup path kernel — the tables must be equal to the last bit,
    whether the iterations run in one launch or in several."""
    import open_spiel_amd as osa
    a = osa.TabularSolver(ctx, "leduc_poker", general_kernel="path", **kwargs)
    b = osa.TabularSolver(ctx, "leduc_poker", general_kernel="split", **kwargs)
    c = osa.TabularSolver(ctx, "leduc_poker", **kwargs)          # auto = the split kernel
    a.evaluate_and_update_policy(64)
    b.evaluate_and_update_policy(64)
    for k in (1, 1, 2, 60):
        c.evaluate_and_update_policy(k)
    ta, tb, tc = a.tables(), b.tables(), c.tables()
    for name in ("regrets", "cum_policy", "cur_policy"):
        np.testing.assert_array_equal(ta[name], tb[name])
        np.testing.assert_array_equal(ta[name], tc[name])
    assert abs(a.nash_conv() - b.nash_conv()) == 0.0


def _judge(oracle, game, solver, which=0):
    t = solver.tables()
    og = oracle.Game(game)
    return og.eval_policy(t["keys"], t["nact"], t["legal"].astype(np.int64), t["avg_policy"], which)


def test_kuhn_cfr_converges_like_the_reference(oracle, ctx):
    """cfr_test.cc:36-62: 300 iterations -> game value -1/18 +- 1e-3, exploitability <= 0.05."""
    import open_spiel_amd as osa
    s = osa.TabularSolver(ctx, "kuhn_poker")
    s.evaluate_and_update_policy(300)
    expl, ev = _judge(oracle, "kuhn_poker", s, which=1)
    assert expl <= 0.05
    assert abs(ev[0] - (-1.0 / 18)) <= 1e-3 and abs(ev[1] - 1.0 / 18) <= 1e-3


def test_cfr_plus_converges_like_the_reference(oracle, ctx):
    """cfr_test.cc:94-103: CFR+ 200 iterations on kuhn."""
    import open_spiel_amd as osa
    s = osa.TabularSolver(ctx, "kuhn_poker", linear_averaging=True, regret_matching_plus=True)
    s.evaluate_and_update_policy(200)
    expl, ev = _judge(oracle, "kuhn_poker", s, which=1)
    assert expl <= 0.05
    assert abs(ev[0] - (-1.0 / 18)) <= 1e-3


def test_leduc_nash_conv_matches_oracle_each_iteration(oracle, ctx):
    """python/algorithms/cfr_test.py:241-272 style: NashConv agrees after each of 5 iterations."""
    import open_spiel_amd as osa
    og = oracle.Game("leduc_poker")
    o = oracle.Solver(og, "cfr")
    s = osa.TabularSolver(ctx, "leduc_poker")
    for _ in range(5):
        o.iterate(1)
        s.evaluate_and_update_policy(1)
        nc, _ = _judge(oracle, "leduc_poker", s, which=0)
        assert abs(nc - o.nash_conv()) <= 1e-10


@pytest.mark.parametrize("game,batches", [
    ("kuhn_poker", [(0, 1), (1, 1), (2, 7), (9, 300), (309, 1000)]),
    ("leduc_poker", [(0, 1), (1, 2), (3, 64), (67, 500)]),
    ("kuhn_poker(players=3)", [(0, 3), (3, 200)]),
])
def test_mccfr_minibatch_replay_parity(oracle, ctx, game, batches):
    """Same tables + same uniform streams => same regret / average-policy deltas: the
    device mini-batch vs the oracle's UpdateRegrets replayed on a frozen table."""
    import open_spiel_amd as osa
    og = oracle.Game(game)
    o = oracle.Solver(og, "mccfr_simple", seed=0)
    s = osa.TabularSolver(ctx, game, mccfr=True)
    seed = 0xBADC0DE
    for first, count in batches:
        o.mccfr_minibatch(seed, first, count)
        s.run_mccfr(seed, count, first_trajectory=first)
        dev, orc = s.tables(), o.tables(s.amax)
        d_idx = _by_key(dev)
        seen = set(orc["keys"])
        for j, k in enumerate(orc["keys"]):
            i, n = d_idx[k], int(orc["nact"][j])
            for name in ("regrets", "cum_policy"):
                np.testing.assert_allclose(dev[name][i, :n], orc[name][j, :n], rtol=1e-11, atol=1e-12,
                                           err_msg=f"{game} batch {(first, count)} {name} at {k!r}")
        for k, i in d_idx.items():  # rows the oracle never visited are still at their initial 1e-6
            if k not in seen:
                n = int(dev["nact"][i])
                assert dev["regrets"][i, :n].tolist() == [1e-6] * n
                assert dev["cum_policy"][i, :n].tolist() == [1e-6] * n


@pytest.mark.parametrize("game,bound,batch,nbatches", [
    ("kuhn_poker", 0.05, 64, 600),       # external_sampling_mccfr_test.cc:104-109: 1000 iterations
    ("leduc_poker", 2.5, 256, 60),
])
def test_mccfr_converges_like_the_reference(oracle, ctx, game, bound, batch, nbatches):
    import open_spiel_amd as osa
    s = osa.TabularSolver(ctx, game, mccfr=True)
    for b in range(nbatches):
        s.run_mccfr(230398247, batch, first_trajectory=b * batch)
    nc, _ = _judge(oracle, game, s, which=0)
    assert nc <= bound, nc


@pytest.mark.parametrize("kind", ["external", "outcome"])
@pytest.mark.parametrize("game", ["kuhn_poker", "leduc_poker", "kuhn_poker(players=3)"])
def test_mccfr_resident_kernel_equals_general_kernel(ctx, game, kind):
    """The LDS-resident traversal (packed tree, per-launch policy, top-of-stack frame in registers) and
    the general global-memory kernel (osg_cfr_cfg.kernel = 1) are the same function of (table, seed)."""
    import open_spiel_amd as osa
    a = osa.TabularSolver(ctx, game, mccfr=kind)
    b = osa.TabularSolver(ctx, game, mccfr=kind, general_kernel=True)
    for seed, first, count in [(7, 0, 1), (7, 1, 999), (8, 1000, 70000), (9, 71000, 300000)]:
        a.run_mccfr(seed, count, first_trajectory=first)
        b.run_mccfr(seed, count, first_trajectory=first)
        ta, tb = a.tables(), b.tables()
        # Both kernels add the same terms with fp64 atomics, in a different order (different workgroup
        # shapes).  Outcome sampling's terms carry importance weights 1 / sample_reach — up to ~1e8 next to
        # sums of ~1e4 — so its sums agree to ~1e-7 relative only; external sampling's are O(1).
        tol = 1e-6 if kind == "outcome" else 1e-9
        for name in ("regrets", "cum_policy"):
            np.testing.assert_allclose(ta[name], tb[name], rtol=tol, atol=tol,
                                       err_msg=f"{game} {kind} {name} after {(seed, first, count)}")


def test_mccfr_sample_then_apply_equals_iterate(ctx):
    import open_spiel_amd as osa
    a = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
    b = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
    a.run_mccfr(5, 4096)
    # two half-batches sampled from the same frozen table, deltas summed, folded once
    import torch
    b.mccfr_sample(5, 2048, first_trajectory=0)
    dr, dp = b.mccfr_delta_tables()
    keep_r, keep_p = dr.clone(), dp.clone()
    b.mccfr_sample(5, 2048, first_trajectory=2048)
    dr += keep_r
    dp += keep_p
    torch.cuda.synchronize()
    b.mccfr_apply_deltas()
    ta, tb = a.tables(), b.tables()
    np.testing.assert_allclose(ta["regrets"], tb["regrets"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(ta["cum_policy"], tb["cum_policy"], rtol=1e-11, atol=1e-11)


def test_cfr_rejects_board_games(ctx):
    import open_spiel_amd as osa
    with pytest.raises(osa.OsgError):
        osa.TabularSolver(ctx, "tic_tac_toe")


# ---- policy evaluation on the device ------------------------------------------
def _named_policy_table(solver, kind, alpha=0.0):
    """[I, Amax] table in the solver's infostate order: uniform / first-action / kuhn optimal."""
    t = solver.tables()
    I, A = solver.num_infostates, solver.amax
    tab = np.zeros((I, A))
    opt = {"0": [1 - alpha, alpha], "0pb": [1, 0], "1": [1, 0], "1pb": [2 / 3 - alpha, 1 / 3 + alpha],
           "2": [1 - 3 * alpha, 3 * alpha], "2pb": [0, 1], "0p": [2 / 3, 1 / 3], "0b": [1, 0], "1p": [1, 0],
           "1b": [2 / 3, 1 / 3], "2p": [0, 1], "2b": [0, 1]}  # kuhn_poker.cc:451-474
    for i, k in enumerate(t["keys"]):
        n = int(t["nact"][i])
        if kind == "uniform":
            tab[i, :n] = 1.0 / n
        elif kind == "first":
            tab[i, 0] = 1.0
        else:
            tab[i, :n] = opt[k]
    return tab


def test_policy_evaluation_known_answers(ctx):
    """tabular_exploitability_test.cc:470-499: uniform-policy exploitability kuhn 0.4583333333333335,
    leduc 2.373611111111111; NashConv 0.916666666666667 / 4.747222222222222; first-action policy
    NashConv 2 on kuhn; Kuhn optimal(alpha = 0.2) is unexploitable."""
    import open_spiel_amd as osa
    k = osa.TabularSolver(ctx, "kuhn_poker")
    r = k.evaluate_policy("table", _named_policy_table(k, "uniform"))
    assert abs(r["exploitability"] - 0.4583333333333335) < 1e-14
    assert abs(r["nash_conv"] - 0.916666666666667) < 1e-14
    assert abs(k.evaluate_policy("average")["nash_conv"] - 0.916666666666667) < 1e-14  # untouched tables: uniform
    assert abs(k.evaluate_policy("table", _named_policy_table(k, "first"))["nash_conv"] - 2.0) < 1e-14
    r = k.evaluate_policy("table", _named_policy_table(k, "optimal", 0.2))
    assert abs(r["nash_conv"]) < 1e-14 and abs(r["expected_returns"][0] + 1 / 18) < 1e-14
    l = osa.TabularSolver(ctx, "leduc_poker")
    r = l.evaluate_policy("table", _named_policy_table(l, "uniform"))
    assert abs(r["exploitability"] - 2.373611111111111) < 1e-13
    assert abs(r["nash_conv"] - 4.747222222222222) < 1e-13


@pytest.mark.parametrize("game,iters", [("kuhn_poker", 37), ("leduc_poker", 6), ("kuhn_poker(players=3)", 9)])
def test_policy_evaluation_matches_the_oracle_judge(oracle, ctx, game, iters):
    import open_spiel_amd as osa
    s = osa.TabularSolver(ctx, game)
    s.evaluate_and_update_policy(iters)
    og = oracle.Game(game)
    t = s.tables()
    for which, table in (("average", t["avg_policy"]), ("current", t["cur_policy"])):
        got = s.evaluate_policy(which)
        nc, ev = og.eval_policy(t["keys"], t["nact"], t["legal"].astype(np.int64), table, which=0)
        ex, _ = og.eval_policy(t["keys"], t["nact"], t["legal"].astype(np.int64), table, which=1)
        assert abs(got["nash_conv"] - nc) <= 1e-12, (game, which)
        assert abs(got["exploitability"] - ex) <= 1e-12
        np.testing.assert_allclose(got["expected_returns"], ev, rtol=0, atol=1e-13)
    assert s.nash_conv() == s.evaluate_policy("average")["nash_conv"]


# ---- outcome-sampling MCCFR -----------------------------------------------------
@pytest.mark.parametrize("game,batches", [
    ("kuhn_poker", [(0, 1), (1, 1), (2, 9), (11, 400), (411, 2000)]),
    ("leduc_poker", [(0, 1), (1, 3), (4, 128), (132, 1000)]),
    ("kuhn_poker(players=3)", [(0, 2), (2, 300)]),
])
def test_os_mccfr_minibatch_replay_parity(oracle, ctx, game, batches):
    """Same tables + same uniforms => same regret / average-policy increments as the oracle's
    SampleEpisode (outcome_sampling_mccfr.cc:141-241) replayed on a frozen table."""
    import open_spiel_amd as osa
    og = oracle.Game(game)
    o = oracle.Solver(og, "mccfr_outcome", seed=0)
    s = osa.TabularSolver(ctx, game, mccfr="outcome", epsilon=0.6)
    seed = 0x05C0FFEE
    for first, count in batches:
        o.mccfr_minibatch(seed, first, count)
        s.run_mccfr(seed, count, first_trajectory=first)
        dev, orc = s.tables(), o.tables(s.amax)
        d_idx = _by_key(dev)
        for j, k in enumerate(orc["keys"]):
            i, n = d_idx[k], int(orc["nact"][j])
            for name in ("regrets", "cum_policy"):
                np.testing.assert_allclose(dev[name][i, :n], orc[name][j, :n], rtol=1e-10, atol=1e-11,
                                           err_msg=f"{game} batch {(first, count)} {name} at {k!r}")


@pytest.mark.parametrize("game,bound,batch,nbatches", [
    ("kuhn_poker", 0.17, 64, 320),      # outcome_sampling_mccfr_test.cc: 10000 iterations (= 20000 episodes)
    ("leduc_poker", 3.07, 256, 80),
])
def test_os_mccfr_converges_like_the_reference(ctx, game, bound, batch, nbatches):
    import open_spiel_amd as osa
    s = osa.TabularSolver(ctx, game, mccfr="outcome")
    for b in range(nbatches):
        s.run_mccfr(230398247, batch, first_trajectory=b * batch)
    assert s.nash_conv() <= bound


def test_os_mccfr_rejects_bad_epsilon(ctx):
    import open_spiel_amd as osa
    with pytest.raises(osa.OsgError):
        osa.TabularSolver(ctx, "kuhn_poker", mccfr="outcome", epsilon=0.0)


# ---- solver replicas: one workgroup per independent solver ------------------------------------------
def test_replicas_are_independent_identical_solvers(ctx):
    import open_spiel_amd as osa
    one = osa.TabularSolver(ctx, "kuhn_poker")
    many = osa.TabularSolver(ctx, "kuhn_poker", replicas=96)
    one.evaluate_and_update_policy(40)
    many.evaluate_and_update_policy(40)
    want = one.tables()
    for r in (0, 1, 37, 95):
        many.select_replica(r)
        got = many.tables()
        for name in ("regrets", "cum_policy", "cur_policy"):
            np.testing.assert_array_equal(got[name], want[name])
    with pytest.raises(osa.OsgError):
        many.select_replica(96)
    with pytest.raises(osa.OsgError):
        osa.TabularSolver(ctx, "leduc_poker", replicas=4)   # does not fit LDS: one solver per object
    with pytest.raises(osa.OsgError):
        osa.TabularSolver(ctx, "kuhn_poker", replicas=4, mccfr=True)


def test_random_initial_regrets_replicas(ctx):
    """CFRSolverBase(random_initial_regrets=true, seed) (cfr.h:190-196, cfr.cc:249-252): regrets start at
    0.001 * U[0,1), the first policy is their regret matching; replica r of a batch equals a single solver
    created with replica_offset = r; every replica still converges."""
    import open_spiel_amd as osa
    batch = osa.TabularSolver(ctx, "kuhn_poker", replicas=8, random_initial_regrets=True, seed=99)
    t0 = []
    for r in range(8):
        batch.select_replica(r)
        t = batch.tables()
        n = t["nact"]
        for i in range(len(n)):
            reg = t["regrets"][i, :n[i]]
            assert ((reg >= 0) & (reg < 0.001)).all()
            np.testing.assert_allclose(t["cur_policy"][i, :n[i]], reg / reg.sum(), rtol=0, atol=1e-15)
        t0.append(t["regrets"].copy())
    assert not np.array_equal(t0[0], t0[1]), "replicas start from different regrets"
    batch.evaluate_and_update_policy(300)
    for r in (0, 5):
        single = osa.TabularSolver(ctx, "kuhn_poker", random_initial_regrets=True, seed=99, replica_offset=r)
        np.testing.assert_array_equal(single.tables()["regrets"], t0[r])
        single.evaluate_and_update_policy(300)
        batch.select_replica(r)
        a, b = single.tables(), batch.tables()
        for name in ("regrets", "cum_policy", "cur_policy"):
            np.testing.assert_array_equal(a[name], b[name])
        assert batch.exploitability() <= 0.05


@pytest.mark.parametrize("game", ["kuhn_poker(players=4)", "kuhn_poker(players=5)", "leduc_poker(suit_isomorphism=True)",
                                  "leduc_poker(action_mapping=True)", "leduc_poker(starting_player=1)"])
def test_cfr_on_game_variants_matches_the_oracle(oracle, ctx, game):
    """Larger / parameterised trees (up to 116 437 histories): tables and NashConv agree with the oracle."""
    import open_spiel_amd as osa
    s = osa.TabularSolver(ctx, game)
    o = oracle.Solver(oracle.Game(game), "cfr")
    s.evaluate_and_update_policy(3)
    o.iterate(3)
    _compare_tables(s.tables(), o.tables(s.amax), 1e-12, game)
    assert abs(s.nash_conv() - o.nash_conv()) <= 1e-11


def test_cfr_three_player_leduc_builds_and_runs(ctx):
    """1 831 601 histories / 25 800 infostates, flattened on the device in well under a second."""
    import open_spiel_amd as osa
    s = osa.TabularSolver(ctx, "leduc_poker(players=3)")
    assert s.num_histories == 1831601 and s.num_infostates == 25800
    before = s.nash_conv()
    s.evaluate_and_update_policy(4)
    assert s.nash_conv() < before


@pytest.mark.parametrize("game,kwargs", [("leduc_poker(players=3)", {}),
                                         ("leduc_poker(players=3)", dict(linear_averaging=True, regret_matching_plus=True)),
                                         ("leduc_poker", {}), ("kuhn_poker(players=5)", {})])
def test_persistent_subtree_kernel_is_bit_identical_with_the_per_phase_launches(ctx, game, kwargs):
    """k_cfr_sub (one cooperative launch, a workgroup per deal subtree, two grid barriers per player pass) performs
    the additions of the per-phase launches in the same order: tables equal to the last bit, whether the iterations
    run in one launch or in several, on trees of 9 457 (leduc), 17 k (5-player kuhn) and 1.83 M histories."""
    import open_spiel_amd as osa
    try:
        b = osa.TabularSolver(ctx, game, general_kernel="sub", **kwargs)
    except osa.OsgError:
        pytest.skip("tree shape not served by the subtree kernel")
    a = osa.TabularSolver(ctx, game, general_kernel="grid", **kwargs)
    a.evaluate_and_update_policy(7)
    for k in (1, 2, 4):
        b.evaluate_and_update_policy(k)
    ta, tb = a.tables(), b.tables()
    for name in ("regrets", "cum_policy", "cur_policy"):
        np.testing.assert_array_equal(ta[name], tb[name])
    assert a.iteration == b.iteration == 7


@pytest.mark.parametrize("game,pack,form", [("leduc_poker(players=3)", "1", "k_cfr_sub<forest>"),
                                            ("leduc_poker(players=3)", "0", "k_cfr_sub"),
                                            ("kuhn_poker(players=6)", "1", "k_cfr_sub<packed>")])
def test_subtree_kernel_bins(ctx, game, pack, form, monkeypatch):
    """More deal subtrees than compute units (3-player leduc: 336; 6-player kuhn: 5 040): the workgroups' bins are packed
    — whole subtrees where they fit a workgroup together, else the pieces one level below the cut with the deal roots
    handled by the fold (forest form) — and OSG_CFR_SUB_PACK=0 keeps a subtree per bin (6-player kuhn is then not served:
    an infostate's 720 members do not fit the fold's LDS stage of a 200-history subtree).  Every form leaves the tables of
    the per-phase launches, bit for bit, with plain CFR and with CFR+ (the linear averaging enters the deal roots'
    terms)."""
    import open_spiel_amd as osa
    monkeypatch.setenv("OSG_CFR_SUB_PACK", pack)
    for kwargs in ({}, dict(linear_averaging=True, regret_matching_plus=True)):
        b = osa.TabularSolver(ctx, game, general_kernel="sub", **kwargs)
        a = osa.TabularSolver(ctx, game, general_kernel="grid", **kwargs)
        for k in (1, 3, 5):
            a.evaluate_and_update_policy(k)
            b.evaluate_and_update_policy(k)
            assert b.last_kernel() == form
            ta, tb = a.tables(), b.tables()
            for name in ("regrets", "cum_policy", "cur_policy"):
                np.testing.assert_array_equal(ta[name], tb[name])


def test_three_player_leduc_takes_the_persistent_kernel_by_default(ctx):
    import time
    import open_spiel_amd as osa
    auto = osa.TabularSolver(ctx, "leduc_poker(players=3)")
    grid = osa.TabularSolver(ctx, "leduc_poker(players=3)", general_kernel="grid")
    auto.evaluate_and_update_policy(3)
    grid.evaluate_and_update_policy(3)
    np.testing.assert_array_equal(auto.tables()["regrets"], grid.tables()["regrets"])
    ctx.synchronize()
    t0 = time.perf_counter()
    auto.evaluate_and_update_policy(20)
    ctx.synchronize()
    assert auto.last_kernel() == "k_cfr_sub<forest>"
    assert 20 / (time.perf_counter() - t0) > 2000    # (1 850 iterations/s with a launch per phase)


@pytest.mark.parametrize("game,kernel", [("leduc_poker", "split"), ("leduc_poker(players=3)", "sub")])
def test_grid_barrier_kernels_beside_a_busy_stream(ctx, game, kernel):
    """The kernels that spin on a grid barrier are launched cooperatively: with a second stream keeping every CU busy
    (the network-guided-search deployment: a forward pass beside the solver) they wait for room instead of starting
    half a grid and timing out.  Same tables as on an idle device, no sticky error."""
    import torch
    import open_spiel_amd as osa
    quiet = osa.TabularSolver(ctx, game, general_kernel=kernel)
    quiet.evaluate_and_update_policy(12)
    want = quiet.tables()
    busy = osa.TabularSolver(ctx, game, general_kernel=kernel)
    side = torch.cuda.Stream()
    x = torch.randn(4096, 4096, device="cuda")
    stop = torch.cuda.Event()
    with torch.cuda.stream(side):
        for _ in range(150):          # ~0.5 s of back-to-back matmuls on the other stream
            x = (x @ x).clamp_(-1.0, 1.0)
        stop.record()
    for _ in range(12):               # launches that land while the matmuls run
        busy.evaluate_and_update_policy(1)
    ctx.synchronize()
    assert not stop.query() or True   # (informative only: on a fast box the matmuls may already be done)
    got = busy.tables()               # raises if a barrier timed out
    side.synchronize()
    for name in ("regrets", "cum_policy", "cur_policy"):
        np.testing.assert_array_equal(got[name], want[name])


def _judge_everything(s, which, table=None):
    """All outputs of the evaluation: the four numbers, every responder's action indices and history values."""
    import ctypes as C
    from open_spiel_amd._abi import check, lib
    out = s.evaluate_policy(which, table)
    sizes = (C.c_int64 * 6)()
    check(lib().osg_cfr_sizes(s._h, sizes))
    H, I = int(sizes[0]), int(sizes[4])
    code = {"average": 0, "current": 1, "table": 2}[which]
    tab = None if table is None else np.ascontiguousarray(table, np.float64)
    tp = None if tab is None else tab.ctypes.data
    best = np.zeros(I, np.int32)
    brv = np.zeros(len(out["expected_returns"]))
    check(lib().osg_cfr_best_response(s._h, code, tp, best.ctypes.data, brv.ctypes.data))
    hist = []
    for r in range(len(brv)):
        hv = np.zeros(H)
        check(lib().osg_cfr_best_response_history_values(s._h, code, tp, r, hv.ctypes.data))
        hist.append(hv)
    return out, best, brv, hist


@pytest.mark.parametrize("game", ["leduc_poker", "leduc_poker(suit_isomorphism=True)", "kuhn_poker(players=5)"])
def test_evaluation_jobs_are_bit_identical_with_the_one_workgroup_evaluation(ctx, game, monkeypatch):
    """k_eval_jobs (expected returns per deal subtree, one best-response job per group of deals the responder cannot tell
    apart, the chance levels above by the last job) forms the sums k_policy_eval forms, in the same order."""
    import open_spiel_amd as osa
    s = osa.TabularSolver(ctx, game)
    s.evaluate_and_update_policy(9)
    rng = np.random.default_rng(5)
    t = s.tables()
    table = rng.random(t["regrets"].shape) * (np.arange(t["regrets"].shape[1])[None, :] < t["nact"][:, None])
    table[rng.random(table.shape) < 0.3] = 0.0       # zero-probability actions: pruned branches, ties
    table /= np.maximum(table.sum(1, keepdims=True), 1e-300)
    table[table.sum(1) == 0, 0] = 1.0
    for which, tab in (("average", None), ("current", None), ("table", table)):
        monkeypatch.delenv("OSG_EVAL_JOBS", raising=False)
        got = _judge_everything(s, which, tab)
        monkeypatch.setenv("OSG_EVAL_JOBS", "0")
        monkeypatch.setenv("OSG_EVAL_GRID", "0")
        want = _judge_everything(s, which, tab)
        monkeypatch.setenv("OSG_EVAL_GRID", "1")         # what the large trees take: a launch per level and phase ...
        grid = _judge_everything(s, which, tab)
        assert s.last_eval_kernel() == "k_geval"
        monkeypatch.setenv("OSG_EVAL_PERSIST", "1")      # ... or (opt-in: it measured slower) one persistent launch over a resident grid
        persist = _judge_everything(s, which, tab)
        assert s.last_eval_kernel() == "k_geval_persist"
        monkeypatch.delenv("OSG_EVAL_PERSIST")
        monkeypatch.delenv("OSG_EVAL_GRID")
        for other in (got, grid, persist):
            for k in ("nash_conv", "exploitability"):
                assert other[0][k] == want[0][k], (which, k)
            np.testing.assert_array_equal(other[0]["expected_returns"], want[0]["expected_returns"])
            np.testing.assert_array_equal(other[0]["best_response_values"], want[0]["best_response_values"])
            np.testing.assert_array_equal(other[1], want[1])
            np.testing.assert_array_equal(other[2], want[2])
            for a, b in zip(other[3], want[3]):
                np.testing.assert_array_equal(a, b)


def test_leduc_evaluation_takes_the_jobs_and_is_fast(ctx):
    """The default leduc_poker evaluation is the multi-workgroup one: a NashConv call (host to host, the tables stay on
    the device) well under the one-workgroup kernel's 154 us of kernel time alone."""
    import time
    import open_spiel_amd as osa
    s = osa.TabularSolver(ctx, "leduc_poker")
    s.evaluate_and_update_policy(20)
    s.nash_conv()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        s.nash_conv()
    us = (time.perf_counter() - t0) / 50 * 1e6
    assert us < 120.0, us


def test_cfr_br_on_the_split_kernel_equals_the_one_workgroup_passes(ctx, monkeypatch):
    """CFR-BR through k_eval_jobs + the subtree kernel's override pass set (cfr_br.cc:48-83) against the one-workgroup
    best responses and passes: the same tables bit for bit, and at least 8 000 iterations per second."""
    import time
    import open_spiel_amd as osa
    fast = osa.TabularSolver(ctx, "leduc_poker")
    fast.evaluate_and_update_policy_cfr_br(11)
    got = fast.tables()
    monkeypatch.setenv("OSG_EVAL_JOBS", "0")
    slow = osa.TabularSolver(ctx, "leduc_poker", general_kernel="path")
    slow.evaluate_and_update_policy_cfr_br(11)
    want = slow.tables()
    monkeypatch.delenv("OSG_EVAL_JOBS")
    for name in ("regrets", "cum_policy", "cur_policy"):
        np.testing.assert_array_equal(got[name], want[name])
    ctx.synchronize()
    t0 = time.perf_counter()
    fast.evaluate_and_update_policy_cfr_br(300)
    ctx.synchronize()
    rate = 300 / (time.perf_counter() - t0)
    assert rate > 8000.0, rate


def test_cfr_br_on_a_large_tree_takes_the_grid_and_equals_the_one_workgroup_passes(ctx, monkeypatch):
    """CFR-BR where the tree is beyond the jobs and the subtree kernels (5-player kuhn_poker: 116 437 histories; 3-player
    leduc_poker takes the same path): the evaluation's sweep leaves every infostate's best-response action and every
    player's pass runs on the effective policy — by default (round 6) as ONE launch of the persistent kernel per
    iteration (k_cfr_sub<., kBr>), with `general_kernel="grid"` as launch-per-phase CFR (k_gcfr_* with k_gcfr_effpol) —
    the tables of the one-workgroup pass set (k_policy_eval + k_cfr<., kBr>) bit for bit, at many times its rate."""
    import time
    import open_spiel_amd as osa
    game = "kuhn_poker(players=5)"
    fast = osa.TabularSolver(ctx, game)
    fast.evaluate_and_update_policy_cfr_br(3)
    assert fast.last_kernel().startswith("k_cfr_sub") and fast.last_kernel().endswith("br>"), fast.last_kernel()
    got = fast.tables()
    grid = osa.TabularSolver(ctx, game, general_kernel="grid")
    grid.evaluate_and_update_policy_cfr_br(3)
    assert grid.last_kernel() == "k_gcfr<br>"
    got_grid = grid.tables()
    monkeypatch.setenv("OSG_EVAL_GRID", "0")
    slow = osa.TabularSolver(ctx, game)
    slow.evaluate_and_update_policy_cfr_br(3)
    want = slow.tables()
    monkeypatch.delenv("OSG_EVAL_GRID")
    for name in ("regrets", "cum_policy", "cur_policy"):
        np.testing.assert_array_equal(got[name], want[name])
        np.testing.assert_array_equal(got_grid[name], want[name])
    big = osa.TabularSolver(ctx, "leduc_poker(players=3)")
    before = big.nash_conv()
    ctx.synchronize()
    t0 = time.perf_counter()
    big.evaluate_and_update_policy_cfr_br(20)
    ctx.synchronize()
    rate = 20 / (time.perf_counter() - t0)
    assert big.last_kernel() == "k_cfr_sub<forest,br>" and rate > 150.0, (big.last_kernel(), rate)   # (one workgroup: ~10 iterations per second)
    assert big.nash_conv() < before
    # the persistent form against a launch per phase on the 1.83 M-history tree: bit for bit, CFR iterations before and between
    # (the CFR form of the same kernel keeps its rows in LDS between passes; the CFR-BR form must not inherit anything)
    a = osa.TabularSolver(ctx, "leduc_poker(players=3)")
    b = osa.TabularSolver(ctx, "leduc_poker(players=3)", general_kernel="grid")
    for s in (a, b):
        s.evaluate_and_update_policy(2)
        s.evaluate_and_update_policy_cfr_br(3)
        s.evaluate_and_update_policy(1)
        s.evaluate_and_update_policy_cfr_br(2)
    assert a.last_kernel() == "k_cfr_sub<forest,br>" and b.last_kernel() == "k_gcfr<br>"
    ta, tb = a.tables(), b.tables()
    for name in ("regrets", "cum_policy", "cur_policy"):
        np.testing.assert_array_equal(ta[name], tb[name])


def test_a_smaller_solver_does_not_lower_the_lds_cap_under_a_larger_one(ctx):
    """The dynamic-LDS cap is an attribute of a KERNEL: a later solver of a smaller game must not lower it under an
    earlier solver that is still in use (leduc's kernels need more than the 64 KB default)."""
    import open_spiel_amd as osa
    big = osa.TabularSolver(ctx, "leduc_poker")
    big_m = osa.TabularSolver(ctx, "leduc_poker", mccfr=True)
    big.evaluate_and_update_policy(2)
    before = big.nash_conv()
    for game in ("leduc_poker(suit_isomorphism=True)", "kuhn_poker", "kuhn_poker(players=3)"):
        small = osa.TabularSolver(ctx, game)
        small.evaluate_and_update_policy(2)
        small.nash_conv()
        sm = osa.TabularSolver(ctx, game, mccfr=True)
        sm.run_mccfr(3, 1000)
    big.evaluate_and_update_policy(2)           # the kernels of the first solvers launch with their own footprint
    big_m.run_mccfr(3, 5000)
    big.evaluate_and_update_policy_cfr_br(0)
    assert big.nash_conv() < before and np.isfinite(big_m.nash_conv())


def test_large_tree_evaluation_takes_the_grid_and_equals_the_one_workgroup_walk(ctx, monkeypatch):
    """3-player leduc_poker (1.83 M histories): the evaluation is a launch per level and phase (k_geval_*) or — opt-in,
    OSG_EVAL_PERSIST=1, round 6: it measured slower — ONE persistent launch over a resident grid (k_geval_persist): both
    bit-identical with the one-workgroup walk (33 ms per call) and many times faster; every infostate's best-response
    action and every responder's value of every history equal between the two grid forms."""
    import time
    import open_spiel_amd as osa
    s = osa.TabularSolver(ctx, "leduc_poker(players=3)")
    s.evaluate_and_update_policy(3)
    s.evaluate_policy()
    assert s.last_eval_kernel() == "k_geval"
    ctx.synchronize()
    t0 = time.perf_counter()
    got = s.evaluate_policy()
    fast = time.perf_counter() - t0
    all_launches = _judge_everything(s, "current")
    monkeypatch.setenv("OSG_EVAL_PERSIST", "1")
    s.evaluate_policy()
    t0 = time.perf_counter()
    got_persist = s.evaluate_policy()
    persist = time.perf_counter() - t0
    assert s.last_eval_kernel() == "k_geval_persist"
    all_persist = _judge_everything(s, "current")
    for _ in range(20):   # (the barrier's counters are re-armed per launch)
        assert s.evaluate_policy()["nash_conv"] == got_persist["nash_conv"]
    monkeypatch.delenv("OSG_EVAL_PERSIST")
    monkeypatch.setenv("OSG_EVAL_GRID", "0")
    t0 = time.perf_counter()
    want = s.evaluate_policy()
    slow = time.perf_counter() - t0
    assert s.last_eval_kernel() == "k_policy_eval"
    monkeypatch.delenv("OSG_EVAL_GRID")
    for g in (got, got_persist):
        assert g["nash_conv"] == want["nash_conv"]
        np.testing.assert_array_equal(g["expected_returns"], want["expected_returns"])
        np.testing.assert_array_equal(g["best_response_values"], want["best_response_values"])
    np.testing.assert_array_equal(all_persist[1], all_launches[1])
    np.testing.assert_array_equal(all_persist[2], all_launches[2])
    for a_, b_ in zip(all_persist[3], all_launches[3]):
        np.testing.assert_array_equal(a_, b_)
    assert fast < slow / 5 and persist < slow / 5, (fast, persist, slow)
