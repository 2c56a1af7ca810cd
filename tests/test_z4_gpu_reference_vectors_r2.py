"""Round-2 solvers of the HIP engine against outputs of the GENUINE reference
(tests/golden/reference_vectors_r2.npz, generator tests/golden/make_reference_vectors_r2.py):

  * CFRBRSolver (cfr_br.cc:48-83) — device best responses + per-player passes against them: tables at 1e-12;
  * ExternalSamplingMCCFRSolver, AverageType kSimple and kFull, driven by std::mt19937(seed) exactly as the
    reference drives it (RunIteration(std::mt19937*), external_sampling_mccfr.h:63-100): the device traversal
    consumes the generator's uniform_real_distribution sequence in visiting order, so the tables follow the
    reference's iteration by iteration — hundreds of data-dependent sampled traversals, tables at 1e-9
    (regret matching divides; the bar of north_star is 1e-6 on the average policy);
  * FullUpdateAverage (external_sampling_mccfr.cc:188-231) against the oracle's on a shared table (replay).

Neither /root/reference nor the genuine build is touched at run time.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def vectors():
    with np.load(os.path.join(HERE, "golden", "reference_vectors_r2.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def ctx():
    import open_spiel_amd as osa
    return osa.Context(0)


def _want(vectors, prefix):
    keys = bytes(vectors[prefix + "keys"]).decode().split("\n")
    return keys, {k: vectors[prefix + k] for k in ("nact", "legal", "regrets", "cum_policy", "cur_policy", "avg_policy")}


def _compare(dev, keys, want, names, atol, what, initial=None):
    idx = {k: i for i, k in enumerate(dev["keys"])}
    assert set(keys) <= set(idx), f"{what}: infostates the device does not know"
    worst = 0.0
    for j, k in enumerate(keys):
        i, n = idx[k], int(want["nact"][j])
        assert int(dev["nact"][i]) == n
        np.testing.assert_array_equal(dev["legal"][i, :n], want["legal"][j, :n])
        for name in names:
            d = np.abs(dev[name][i, :n] - want[name][j, :n]).max()
            worst = max(worst, float(d))
            assert d <= atol, f"{what}: {name} at {k!r}: {dev[name][i, :n]} vs {want[name][j, :n]}"
    if initial is not None:  # rows the reference never created are still at the solver's initial values
        seen = set(keys)
        for k, i in idx.items():
            if k not in seen:
                n = int(dev["nact"][i])
                assert dev["regrets"][i, :n].tolist() == [initial] * n, f"{what}: unvisited row {k!r} changed"
                assert dev["cum_policy"][i, :n].tolist() == [initial] * n, f"{what}: unvisited row {k!r} changed"
    return worst


@pytest.mark.parametrize("game,checkpoints", [("kuhn_poker", [1, 2, 10, 60]), ("leduc_poker", [1, 2, 5]),
                                              ("kuhn_poker(players=3)", [1, 8])])
def test_cfr_br_tables_equal_the_reference(ctx, vectors, game, checkpoints):
    import open_spiel_amd as osa
    s = osa.TabularSolver(ctx, game, alternating_updates=False)
    done = 0
    for cp in checkpoints:
        s.evaluate_and_update_policy_cfr_br(cp - done)
        done = cp
        assert s.iteration == cp
        keys, want = _want(vectors, f"cfr_br/{game}/{cp}/")
        dev = s.tables()
        assert sorted(keys) == sorted(dev["keys"])
        _compare(dev, keys, want, ("regrets", "cum_policy", "cur_policy"), 1e-12, f"{game} CFR-BR after {cp}")
        _compare(dev, keys, want, ("avg_policy",), 1e-6, f"{game} CFR-BR average policy after {cp}")
        assert abs(s.nash_conv() - float(vectors[f"cfr_br/{game}/{cp}/nash_conv"])) <= 1e-9


def test_cfr_br_refuses_other_solver_families(ctx):
    import open_spiel_amd as osa
    with pytest.raises(osa.OsgError):
        osa.TabularSolver(ctx, "kuhn_poker", linear_averaging=True, regret_matching_plus=True).evaluate_and_update_policy_cfr_br(1)
    with pytest.raises(osa.OsgError):
        osa.TabularSolver(ctx, "kuhn_poker", mccfr=True).evaluate_and_update_policy_cfr_br(1)


def _mt19937_uniforms(seed, n):
    """n draws of std::uniform_real_distribution<double>(0, 1) on std::mt19937(seed) (libstdc++:
    generate_canonical<double, 53> = two 32-bit outputs, low word first, over 2^64)."""
    rs = np.random.RandomState(seed)  # init_genrand(seed): the seeding std::mt19937(seed) uses
    raw = rs.randint(0, 2 ** 32, size=2 * n, dtype=np.uint64)
    lo, hi = raw[0::2].astype(np.float64), raw[1::2].astype(np.float64)
    u = (lo + hi * 4294967296.0) / 18446744073709551616.0
    return np.where(u >= 1.0, np.nextafter(1.0, 0.0), u)


@pytest.mark.parametrize("game,kind,seed,checkpoints", [
    ("kuhn_poker", "mccfr_simple", 7, [1, 40, 400]),
    ("kuhn_poker", "mccfr_full", 11, [1, 40, 400]),
    ("leduc_poker", "mccfr_simple", 3, [1, 25, 120]),
    ("leduc_poker", "mccfr_full", 5, [1, 25, 120]),
    ("kuhn_poker(players=3)", "mccfr_simple", 9, [60]),
    ("kuhn_poker(players=3)", "mccfr_full", 13, [60]),
])
def test_es_mccfr_on_the_reference_stream_equals_the_reference(ctx, vectors, game, kind, seed, checkpoints):
    """The engine's ES-MCCFR fed with the reference's own random stream, through the C-ABI."""
    import open_spiel_amd as osa
    full = kind == "mccfr_full"
    P = osa.describe(game).num_players
    for cp in checkpoints:
        s = osa.TabularSolver(ctx, game, mccfr=True)
        s.set_average_type(full)
        stream = _mt19937_uniforms(seed, 400_000)
        at = 0
        for _ in range(cp):
            for p in range(P):
                used = s.mccfr_sample_uniforms(p, stream[at:at + s.num_histories])
                s.mccfr_apply_deltas()
                at += used
            if full:
                s.mccfr_full_average()
        keys, want = _want(vectors, f"mccfr/{game}/{kind}/{seed}/{cp}/")
        _compare(s.tables(), keys, want, ("regrets", "cum_policy"), 1e-9, f"{game} {kind} seed {seed} after {cp}",
                 initial=1e-6)
        _compare(s.tables(), keys, want, ("avg_policy",), 1e-6, f"{game} {kind} average policy after {cp}")
        assert abs(s.nash_conv() - float(vectors[f"mccfr/{game}/{kind}/{seed}/{cp}/nash_conv"])) <= 1e-7


@pytest.mark.parametrize("game,kind,seed,cp", [("kuhn_poker", "mccfr_full", 11, 400), ("leduc_poker", "mccfr_simple", 3, 120),
                                               ("leduc_poker", "mccfr_full", 5, 25)])
def test_host_mirror_run_iteration_with_a_generator_equals_the_reference(vectors, game, kind, seed, cp):
    """The same through the drop-in class: ExternalSamplingMCCFRSolver::RunIteration(std::mt19937*) of the host
    mirror (pyspiel_hip), std::mt19937 and std::uniform_real_distribution from libstdc++ itself."""
    import open_spiel_amd.pyspiel_hip as ps
    g = ps.load_game(game)
    avg = ps.MCCFRAverageType.FULL if kind == "mccfr_full" else ps.MCCFRAverageType.SIMPLE
    s = ps.ExternalSamplingMCCFRSolver(g, seed, avg)
    s.run_iterations_mt19937(seed, cp)
    table = s.info_state_values_table()
    keys, want = _want(vectors, f"mccfr/{game}/{kind}/{seed}/{cp}/")
    for j, k in enumerate(keys):
        n = int(want["nact"][j])
        v = table[k]
        assert list(v.legal_actions) == want["legal"][j, :n].tolist()
        np.testing.assert_allclose(v.cumulative_regrets, want["regrets"][j, :n], rtol=0, atol=1e-9, err_msg=k)
        np.testing.assert_allclose(v.cumulative_policy, want["cum_policy"][j, :n], rtol=0, atol=1e-9, err_msg=k)


@pytest.mark.parametrize("game,kind,seed,cp", [("kuhn_poker", "mccfr_simple", 7, 400), ("leduc_poker", "mccfr_full", 5, 25)])
def test_host_mirror_default_run_iteration_and_checkpoint_follow_the_reference(vectors, game, kind, seed, cp):
    """RunIteration() with no argument draws from the solver's own std::mt19937(seed), as the reference's does
    (external_sampling_mccfr.h:66): the solver constructed like the reference's follows it — also across a
    Serialize / Deserialize in the middle, whose [SolverRNG] section carries the generator's state in the
    reference's own format (external_sampling_mccfr.cc:100-102, :262-264)."""
    import pickle
    import open_spiel_amd.pyspiel_hip as ps
    g = ps.load_game(game)
    avg = ps.MCCFRAverageType.FULL if kind == "mccfr_full" else ps.MCCFRAverageType.SIMPLE
    s = ps.ExternalSamplingMCCFRSolver(g, seed, avg)
    for _ in range(cp // 2):
        s.run_iteration()
    text = s.serialize()
    rng_section = text.split("[SolverRNG]\n")[1].split("\n")[0].split()
    assert len(rng_section) == 625  # 624 state words + the position: operator<< of std::mt19937
    s = pickle.loads(pickle.dumps(s))
    for _ in range(cp - cp // 2):
        s.run_iteration()
    table = s.info_state_values_table()
    keys, want = _want(vectors, f"mccfr/{game}/{kind}/{seed}/{cp}/")
    for j, k in enumerate(keys):
        n = int(want["nact"][j])
        v = table[k]
        np.testing.assert_allclose(v.cumulative_regrets, want["regrets"][j, :n], rtol=0, atol=1e-9, err_msg=k)
        np.testing.assert_allclose(v.cumulative_policy, want["cum_policy"][j, :n], rtol=0, atol=1e-9, err_msg=k)


@pytest.mark.parametrize("game", ["kuhn_poker", "leduc_poker", "kuhn_poker(players=3)"])
def test_full_update_average_replay_parity(oracle, ctx, game):
    """kFull on the engine's own counter streams: trajectory by trajectory the device and the oracle's
    UpdateRegrets see the same table and the same uniforms, each iteration ends with FullUpdateAverage on both
    sides; tables stay equal (the oracle creates rows lazily: rows it has not seen are at 1e-6 on the device)."""
    import open_spiel_amd as osa
    P = osa.describe(game).num_players
    o = oracle.Solver(oracle.Game(game), "mccfr_full", seed=0)
    s = osa.TabularSolver(ctx, game, mccfr=True)
    s.set_average_type(True)
    seed, g = 0xF0115EED, 0
    for it in range(12 if "leduc" in game else 40):
        for p in range(P):
            o.mccfr_minibatch(seed, g, 1)
            s.run_mccfr(seed, 1, first_trajectory=g)   # fewer than P trajectories: no full-average pass inside
            g += 1
        o.mccfr_full_average()
        s.mccfr_full_average()
    ot = o.tables(s.amax)
    _compare(s.tables(), ot["keys"], ot, ("regrets", "cum_policy"), 1e-11, f"{game} kFull replay", initial=1e-6)
    # a mini-batch of T trajectories ends with ONE pass weighted T / P
    before = s.tables()["cum_policy"].copy()
    s2 = osa.TabularSolver(ctx, game, mccfr=True)
    s2.set_average_type(True)
    s2.load_tables(regrets=s.tables()["regrets"], cum_policy=before)
    s2.mccfr_full_average(1.0)
    one = s2.tables()["cum_policy"] - before
    s2.load_tables(regrets=s.tables()["regrets"], cum_policy=before)
    s2.mccfr_full_average(8.0)
    # (both sides are differences of sums near 40: one ulp of the sums is 7e-15)
    np.testing.assert_allclose(s2.tables()["cum_policy"] - before, 8.0 * one, rtol=1e-12, atol=5e-14)


def _rng_line(text):
    return text.split("[SolverRNG]\n")[1].split("\n")[0]


@pytest.mark.parametrize("game,which", [("kuhn_poker", "outcome"), ("leduc_poker", "outcome"), ("kuhn_poker", "external")])
def test_mccfr_checkpoints_travel_both_ways_with_the_genuine_reference(game, which):
    """[SolverRNG] of the MCCFR checkpoints is the std::mt19937 dump the reference writes (outcome_sampling_mccfr.cc:
    94-98, 281-283; external_sampling_mccfr.cc:100-102, 262-264), so a checkpoint written by the host mirror loads in
    the running reference (oracle/_ref) and one written by the reference loads in the host mirror: same tables to the
    last bit, same generator state; the mirror's extra "counter ..." line is invisible to the reference's reader."""
    import pickle
    import reference_py
    if not reference_py.available():
        pytest.skip("oracle/_ref/libspiel_ref.so not present")
    import open_spiel_amd.pyspiel_hip as ps
    g = ps.load_game(game)
    rg = reference_py.Game(game)
    kind = "mccfr_outcome" if which == "outcome" else "mccfr_simple"
    mine = ps.OutcomeSamplingMCCFRSolver(g, 0.6, 11) if which == "outcome" else ps.ExternalSamplingMCCFRSolver(g, 11)
    load = ps.deserialize_outcome_sampling_mccfr_solver if which == "outcome" else ps.deserialize_external_sampling_mccfr_solver
    # ---- mirror -> reference ----
    for _ in range(40):
        mine.run_iteration()
    text = mine.serialize()
    words = _rng_line(text).split()
    assert len(words) == 625, "operator<< of std::mt19937: 624 state words and the position"
    assert text.split("[SolverRNG]\n")[1].split("\n")[1].startswith("counter ")
    restored = reference_py.Solver.deserialize(rg, text, kind)
    ref_t = restored.tables()
    dev = mine.info_state_values_table()
    assert sorted(dev) == ref_t["keys"]
    for j, k in enumerate(ref_t["keys"]):
        na = int(ref_t["nact"][j])
        assert list(dev[k].cumulative_regrets) == ref_t["regrets"][j, :na].tolist()
        assert list(dev[k].cumulative_policy) == ref_t["cum_policy"][j, :na].tolist()
    assert _rng_line(restored.serialize()) == _rng_line(text), "the reference read the generator's state"
    # ---- reference -> mirror ----
    theirs = reference_py.Solver(rg, kind, seed=5)
    theirs.iterate(30)
    ref_text = theirs.serialize()
    back = load(ref_text)
    ref_t = theirs.tables()
    dev = back.info_state_values_table()
    for j, k in enumerate(ref_t["keys"]):    # (the reference creates rows lazily: the mirror holds every infostate)
        na = int(ref_t["nact"][j])
        assert list(dev[k].cumulative_regrets) == ref_t["regrets"][j, :na].tolist()
        assert list(dev[k].cumulative_policy) == ref_t["cum_policy"][j, :na].tolist()
    assert _rng_line(back.serialize()) == _rng_line(ref_text)
    # the generator advances with RunIteration() and survives pickle
    before = _rng_line(back.serialize())
    back.run_iteration()
    after = _rng_line(back.serialize())
    assert after != before
    assert _rng_line(pickle.loads(pickle.dumps(back)).serialize()) == after
