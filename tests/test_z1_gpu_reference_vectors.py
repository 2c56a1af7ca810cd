"""The HIP engine against outputs of the GENUINE reference implementation.

tests/golden/reference_vectors.npz was produced by running the reference's own .cc files
(oracle/_ref, built by oracle/Makefile.ref; generator: tests/golden/make_reference_vectors.py) — not
by the restatement.  These tests replay the recorded playouts through the C-ABI and rebuild the
recorded CFR tables on the device; neither the oracle nor /root/reference is touched at run time.

Bar (BASELINE.json north_star): bit-exact legal-action sets, players, terminal flags, returns and
tensors; CFR average-policy probabilities within 1e-6 (the tables are compared at 1e-12).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def vectors():
    with np.load(os.path.join(HERE, "golden", "reference_vectors.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def ctx():
    import open_spiel_amd as osa
    return osa.Context(0)


PLAY_GAMES = ["tic_tac_toe", "connect_four", "hex(board_size=9)", "kuhn_poker", "leduc_poker"]
# other geometries / more players: run by the two tests at the END of this module (under `pytest -x` a surprise
# in a variant must not hide the solver and checkpoint tests in between)
VARIANT_GAMES = ["connect_four(rows=5,columns=6,x_in_row=3)", "hex(num_cols=3,num_rows=4)", "kuhn_poker(players=3)",
                 "leduc_poker(players=3)"]


def _replay_on_the_device(ctx, vectors, game):
    """At every ply of every recorded playout: LegalActions (chance outcomes at chance nodes),
    CurrentPlayer, IsTerminal, Returns, and ObservationTensor / InformationStateTensor of every
    player equal what the reference produced."""
    import torch
    import open_spiel_amd as osa
    p = f"play/{game}/"
    acts = vectors[p + "actions"]
    n, L = acts.shape
    batch = osa.StateBatch(ctx, game, n)
    P = batch.num_players
    has_info = (p + "info") in vectors
    for t in range(L + 1):
        bits = batch.legal_actions_mask_bits().cpu().numpy().view(np.uint32)
        np.testing.assert_array_equal(bits, vectors[p + "mask"][:, t], err_msg=f"{game}: legal mask at ply {t}")
        cur, term, rets = batch.status()
        np.testing.assert_array_equal(cur.cpu().numpy(), vectors[p + "cur_player"][:, t], err_msg=f"{game}: player at ply {t}")
        np.testing.assert_array_equal(term.cpu().numpy(), vectors[p + "terminal"][:, t], err_msg=f"{game}: terminal at ply {t}")
        np.testing.assert_array_equal(rets.cpu().numpy(), vectors[p + "returns"][:, t], err_msg=f"{game}: returns at ply {t}")
        for pl in range(P):
            got = batch.observation_tensor(pl).cpu().numpy()
            np.testing.assert_array_equal(got, vectors[p + "obs"][:, t, pl].astype(np.float32),
                                          err_msg=f"{game}: observation tensor p{pl} ply {t}")
            if has_info:
                got = batch.information_state_tensor(pl).cpu().numpy()
                np.testing.assert_array_equal(got, vectors[p + "info"][:, t, pl].astype(np.float32),
                                              err_msg=f"{game}: information state tensor p{pl} ply {t}")
        if t == L:
            break
        batch.apply_actions(torch.from_numpy(acts[:, t].astype(np.int32)))
    assert vectors[p + "terminal"][:, L].all()


@pytest.mark.parametrize("game", PLAY_GAMES)
def test_reference_playouts_replayed_on_the_device(ctx, vectors, game):
    _replay_on_the_device(ctx, vectors, game)


def _through_the_fused_step(ctx, vectors, game):
    """The same recorded playouts through the fused kernel (legality + apply + status + successor
    mask in one launch): terminal flag, player to move, no illegal flag, final returns."""
    import torch
    import open_spiel_amd as osa
    p = f"play/{game}/"
    acts = vectors[p + "actions"]
    n, L = acts.shape
    a, b = osa.StateBatch(ctx, game, n), osa.StateBatch(ctx, game, n)
    for t in range(L):
        col = acts[:, t]
        a8 = torch.from_numpy(np.where(col < 0, 255, col).astype(np.uint8)).cuda()
        _, status = a.step(a8, dst=b)
        st = status.cpu().numpy()
        term = (st & 0x80) != 0
        np.testing.assert_array_equal(term, vectors[p + "terminal"][:, t + 1] != 0)
        assert not (st & 0x40).any(), "no action of a reference playout is illegal"
        live = ~term
        np.testing.assert_array_equal((st[live] & 15).astype(np.int64) - 1, vectors[p + "cur_player"][:, t + 1][live])
        a, b = b, a
    np.testing.assert_array_equal(a.returns().cpu().numpy(), vectors[p + "returns"][:, L])


@pytest.mark.parametrize("game", PLAY_GAMES)
def test_reference_playouts_through_the_fused_step(ctx, vectors, game):
    _through_the_fused_step(ctx, vectors, game)


SOLVER_KWARGS = {
    "cfr": {},
    "cfr_plus": dict(linear_averaging=True, regret_matching_plus=True),
    "cfr_simultaneous": dict(alternating_updates=False),
}


def _cfr_runs(vectors):
    runs = {}
    for k in vectors:
        if k.startswith("cfr/") and k.endswith("/keys"):
            _, game, kind, iters, _ = k.split("/")
            runs.setdefault((game, kind), []).append(int(iters))
    return sorted((g, k, sorted(v)) for (g, k), v in runs.items())


def test_fixture_has_cfr_runs(vectors):
    assert len(_cfr_runs(vectors)) >= 5


@pytest.mark.parametrize("game,kind", [("kuhn_poker", "cfr"), ("kuhn_poker", "cfr_plus"),
                                       ("kuhn_poker", "cfr_simultaneous"), ("leduc_poker", "cfr"),
                                       ("leduc_poker", "cfr_plus")])
def test_cfr_tables_match_the_reference_outputs(ctx, vectors, game, kind):
    """EvaluateAndUpdatePolicy on the device vs the tables the reference's CFRSolver produced:
    cumulative regrets, cumulative policy, current policy and average policy at every recorded
    checkpoint; NashConv / exploitability / expected returns from the device judge."""
    import open_spiel_amd as osa
    checkpoints = dict((g + "/" + k, v) for g, k, v in _cfr_runs(vectors))[game + "/" + kind]
    s = osa.TabularSolver(ctx, game, **SOLVER_KWARGS[kind])
    done = 0
    for cp in checkpoints:
        s.evaluate_and_update_policy(cp - done)
        done = cp
        grp = f"cfr/{game}/{kind}/{cp}/"
        keys = vectors[grp + "keys"].tobytes().decode().split("\n")
        dev = s.tables()
        assert sorted(dev["keys"]) == sorted(keys)
        row = {k: i for i, k in enumerate(dev["keys"])}
        for j, k in enumerate(keys):
            i = row[k]
            na = int(vectors[grp + "nact"][j])
            assert int(dev["nact"][i]) == na
            assert dev["legal"][i, :na].tolist() == vectors[grp + "legal"][j, :na].tolist()
            for name in ("regrets", "cum_policy", "cur_policy", "avg_policy"):
                diff = np.abs(dev[name][i, :na] - vectors[grp + name][j, :na]).max()
                assert diff <= 1e-12, f"{game} {kind} after {cp}: {name} at {k!r}"
            # the north-star bar, stated explicitly
            assert np.abs(dev["avg_policy"][i, :na] - vectors[grp + "avg_policy"][j, :na]).max() <= 1e-6
        got = s.evaluate_policy("average")
        assert abs(got["nash_conv"] - float(vectors[grp + "nash_conv"])) <= 1e-10
        assert abs(got["exploitability"] - float(vectors[grp + "exploitability"])) <= 1e-10
        np.testing.assert_allclose(got["expected_returns"], vectors[grp + "expected_returns"], rtol=0, atol=1e-10)


def test_tree_census_matches_the_reference_outputs(ctx, vectors):
    import open_spiel_amd as osa
    for key in [k for k in vectors if k.startswith("census/")]:
        game = key.split("/")[1]
        s = osa.TabularSolver(ctx, game)
        chance, decision, terminal, infostates = (int(x) for x in vectors[key])
        assert (s.num_chance, s.num_decision, s.num_terminal, s.num_infostates) == (chance, decision, terminal, infostates)


# ---- wire format: the reference's solver checkpoints -------------------------------
@pytest.fixture(scope="module")
def pyspiel():
    from open_spiel_amd import pyspiel_hip
    return pyspiel_hip


def _table_rows(vectors, grp):
    keys = vectors[grp + "keys"].tobytes().decode().split("\n")
    return [(k, int(vectors[grp + "nact"][j]), j) for j, k in enumerate(keys)]


@pytest.mark.parametrize("game,kind,iters,more", [("kuhn_poker", "cfr", 10, 5), ("kuhn_poker", "cfr_plus", 7, 0),
                                                  ("leduc_poker", "cfr", 5, 0)])
def test_reference_checkpoint_loads_on_the_device(pyspiel, vectors, game, kind, iters, more):
    """A checkpoint written by the reference's CFRSolverBase::Serialize (lossless hex floats) is read by
    the host mirror's DeserializeCFRSolver / DeserializeCFRPlusSolver: the device tables then hold exactly
    the reference's values, and (CFRSolver) iterating on from it reproduces the reference's later tables."""
    text = vectors[f"ckpt/{game}/{kind}/{iters}/text"].tobytes().decode()
    load = pyspiel.deserialize_cfr_solver if kind == "cfr" else pyspiel.deserialize_cfr_plus_solver
    solver = load(text)
    grp = f"cfr/{game}/{kind}/{iters}/"
    table = solver.info_state_values_table()
    rows = _table_rows(vectors, grp)
    assert sorted(table) == sorted(k for k, _, _ in rows)
    for k, na, j in rows:
        v = table[k]
        assert list(v.legal_actions) == vectors[grp + "legal"][j, :na].tolist()
        assert list(v.cumulative_regrets) == vectors[grp + "regrets"][j, :na].tolist()
        assert list(v.cumulative_policy) == vectors[grp + "cum_policy"][j, :na].tolist()
        assert list(v.current_policy) == vectors[grp + "cur_policy"][j, :na].tolist()
    if more:
        solver.evaluate_and_update_policy(more)
        grp = f"cfr/{game}/{kind}/{iters + more}/"
        table = solver.info_state_values_table()
        for k, na, j in _table_rows(vectors, grp):
            v = table[k]
            np.testing.assert_allclose(v.cumulative_regrets, vectors[grp + "regrets"][j, :na], rtol=0, atol=1e-12)
            np.testing.assert_allclose(v.cumulative_policy, vectors[grp + "cum_policy"][j, :na], rtol=0, atol=1e-12)
            np.testing.assert_allclose(v.current_policy, vectors[grp + "cur_policy"][j, :na], rtol=0, atol=1e-12)
    # the 6-decimal form parses too (values then agree to the printed precision)
    coarse = load(vectors[f"ckpt/{game}/{kind}/{iters}/text6"].tobytes().decode()).info_state_values_table()
    grp = f"cfr/{game}/{kind}/{iters}/"
    for k, na, j in rows:
        np.testing.assert_allclose(coarse[k].cumulative_regrets, vectors[grp + "regrets"][j, :na], rtol=0, atol=1e-6)


@pytest.mark.parametrize("game,cls_name,kind,iters", [("kuhn_poker", "CFRSolver", "cfr", 12),
                                                     ("leduc_poker", "CFRSolver", "cfr", 3),
                                                     ("kuhn_poker", "CFRPlusSolver", "cfr_plus", 9)])
def test_device_checkpoint_loads_in_the_genuine_reference(pyspiel, game, cls_name, kind, iters):
    """The other direction, against the running reference (oracle/_ref/libspiel_ref.so, prebuilt in the
    build container; skipped when the file did not travel): a checkpoint written by the host mirror's
    Serialize() is accepted by the reference's DeserializeCFRSolver / DeserializeCFRPlusSolver and yields
    the device's tables exactly; for CFRSolver, both sides iterating on agree to 1e-12."""
    import reference_py
    if not reference_py.available():
        pytest.skip("oracle/_ref/libspiel_ref.so not present")
    solver = getattr(pyspiel, cls_name)(pyspiel.load_game(game))
    solver.evaluate_and_update_policy(iters)
    text = solver.serialize()
    rg = reference_py.Game(game)
    restored = reference_py.Solver.deserialize(rg, text, kind)
    ref_t = restored.tables()
    dev = solver.info_state_values_table()
    assert sorted(dev) == ref_t["keys"]
    for j, k in enumerate(ref_t["keys"]):
        na = int(ref_t["nact"][j])
        assert list(dev[k].legal_actions) == ref_t["legal"][j, :na].tolist()
        assert list(dev[k].cumulative_regrets) == ref_t["regrets"][j, :na].tolist()
        assert list(dev[k].cumulative_policy) == ref_t["cum_policy"][j, :na].tolist()
        assert list(dev[k].current_policy) == ref_t["cur_policy"][j, :na].tolist()
    if kind == "cfr":  # (the reference restores a CFRPlusSolver without its CFR+ switches, cfr.h:349-353)
        solver.evaluate_and_update_policy(4)
        restored.iterate(4)
        ref_t = restored.tables()
        dev = solver.info_state_values_table()
        for j, k in enumerate(ref_t["keys"]):
            na = int(ref_t["nact"][j])
            np.testing.assert_allclose(dev[k].cumulative_regrets, ref_t["regrets"][j, :na], rtol=0, atol=1e-12)
            np.testing.assert_allclose(dev[k].cumulative_policy, ref_t["cum_policy"][j, :na], rtol=0, atol=1e-12)


@pytest.mark.parametrize("game", PLAY_GAMES)
def test_serialize_game_and_state_equals_the_reference_text(pyspiel, vectors, game):
    """SerializeGameAndState (spiel.cc:582-603) of a mid-game state: the host mirror writes the very text the
    reference wrote, and reads it back to the same state."""
    text = vectors[f"state/{game}/text"].tobytes().decode()
    history = [int(a) for a in vectors[f"state/{game}/history"]]
    g = pyspiel.load_game(game)
    s = g.new_initial_state()
    for a in history:
        s.apply_action(a)
    assert pyspiel.serialize_game_and_state(g, s) == text
    g2, s2 = pyspiel.deserialize_game_and_state(text)
    assert str(g2) == str(g)
    assert list(s2.history()) == history
    assert str(s2) == str(s)


# ---- the variants, last (see VARIANT_GAMES) ---------------------------------------------------------------
@pytest.mark.parametrize("game", VARIANT_GAMES)
def test_reference_playouts_of_variants_replayed_on_the_device(ctx, vectors, game):
    _replay_on_the_device(ctx, vectors, game)


@pytest.mark.parametrize("game", VARIANT_GAMES)
def test_reference_playouts_of_variants_through_the_fused_step(ctx, vectors, game):
    _through_the_fused_step(ctx, vectors, game)
