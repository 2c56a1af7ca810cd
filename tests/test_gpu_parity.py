"""Parity of the HIP path against the CPU oracle, through the C-ABI.

Bar (BASELINE.json north_star): bit-exact legal-action sets, terminal flags,
current players and returns; observation tensors bit-exact (0/1 and small
integers in fp32).  Everything here needs a real MI355X.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GAMES = [
    "tic_tac_toe",
    "connect_four",
    "connect_four(rows=5,columns=6,x_in_row=3)",
    "connect_four(egocentric_obs_tensor=True)",
    "connect_four(rows=8,columns=8)",                    # 72 board bits: two plane words per colour
    "connect_four(rows=9,columns=10,x_in_row=5)",        # 100 bits
    "connect_four(rows=7,columns=15,egocentric_obs_tensor=True)",   # 120 bits
    "hex(board_size=9)",
    "hex(board_size=5)",
    "hex",                                   # 11x11, 4 words per bit plane
    "hex(num_cols=3,num_rows=4)",
    "hex(num_cols=2,num_rows=3)",
    "hex(num_cols=2,num_rows=2)",
    "hex(board_size=4,swap=True)",
    "hex(board_size=5,plain_obs_tensor=True,swap=True)",
    "hex(board_size=13)",                    # the boards above 128 actions: 6, 8 and 12 words per bit plane
    "hex(board_size=14,swap=True)",          # 197 action ids: 7 mask words in 8-word planes
    "hex(board_size=19)",
    "hex(num_cols=17,num_rows=19,swap=True)",
    "kuhn_poker",
    "kuhn_poker(players=3)",
    "kuhn_poker(players=5)",
    "kuhn_poker(players=10)",                # the largest the reference allows (kuhn_poker.h kMaxPlayers)
    "leduc_poker",
    "leduc_poker(players=3)",
    "leduc_poker(action_mapping=True)",
    "leduc_poker(suit_isomorphism=True)",
    "leduc_poker(players=3,starting_player=2)",
    "leduc_poker(players=4)",                            # 4 to 10 players: the five-plane record
    "leduc_poker(players=6,suit_isomorphism=True)",
    "leduc_poker(players=10)",                           # the largest the reference allows (leduc_poker.cc:49-50)
    "leduc_poker(players=7,action_mapping=True,starting_player=5)",
]


@pytest.fixture(scope="module")
def ctx():
    import open_spiel_amd as osa
    return osa.Context(0)


def _mask_words(bits_i32):
    return bits_i32.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("game", GAMES)
def test_step_by_step_parity(oracle, ctx, game):
    """Replay seeded oracle playouts ply by ply through osg_apply and compare
    LegalActions / CurrentPlayer / IsTerminal / Returns at EVERY position."""
    import torch
    import open_spiel_amd as osa
    og = oracle.Game(game)
    n = 1500  # deliberately not a multiple of the 256-thread block
    rec = og.random_playouts(20240921, n)
    L, W = og.max_plies, og.mask_words
    batch = osa.StateBatch(ctx, game, n)
    assert batch.desc.mask_words == W
    assert batch.desc.num_distinct_actions == og.num_distinct_actions
    assert batch.desc.max_chance_outcomes == og.max_chance_outcomes
    assert batch.desc.obs_size == og.observation_tensor_size
    assert batch.desc.info_size == og.information_state_tensor_size
    assert batch.desc.max_game_length == og.max_game_length
    for t in range(L + 1):
        bits = _mask_words(batch.legal_actions_mask_bits())
        np.testing.assert_array_equal(bits, rec["mask"][:, t], err_msg=f"{game}: legal mask at ply {t}")
        cur, term, rets = batch.status()
        np.testing.assert_array_equal(cur.cpu().numpy(), rec["cur_player"][:, t], err_msg=f"{game}: player at ply {t}")
        np.testing.assert_array_equal(term.cpu().numpy(), rec["terminal"][:, t], err_msg=f"{game}: terminal at ply {t}")
        np.testing.assert_array_equal(rets.cpu().numpy(), rec["returns"][:, t], err_msg=f"{game}: returns at ply {t}")
        if t == L:
            break
        batch.apply_actions(torch.from_numpy(rec["actions"][:, t].astype(np.int32)))
    # with swap=True a game can outlast MaxGameLength() by the swap move (hex.h:136)
    assert rec["terminal"][:, L].all() or "swap=True" in game


@pytest.mark.parametrize("game", ["tic_tac_toe", "connect_four", "hex(board_size=9)", "kuhn_poker", "leduc_poker",
                                  "hex(board_size=4,swap=True)", "leduc_poker(players=3)", "hex(board_size=13)",
                                  "hex(board_size=14,swap=True)", "connect_four(rows=8,columns=8)", "leduc_poker(players=5)",
                                  "connect_four(rows=5,columns=6,x_in_row=3)"])
@pytest.mark.parametrize("n", [4096, 4097, 4098])
def test_fused_step_parity(oracle, ctx, game, n):
    """The fused kernel (legality + apply + status + successor mask), out of place.  4096 states take the
    vectorised kernels (tic_tac_toe four, kuhn / leduc and the non-standard connect_four boards two states per
    thread with 16-byte plane accesses), 4097 the one-state-per-thread kernels, 4098 the two-state kernels but not
    tic_tac_toe's four-state one; the standard connect_four board takes its own kernels (the headline: k_step_c4std2
    for even batches, k_step_c4std for odd ones)."""
    import torch
    import open_spiel_amd as osa
    og = oracle.Game(game)
    rec = og.random_playouts(7, n)
    L = og.max_plies
    a, b = osa.StateBatch(ctx, game, n), osa.StateBatch(ctx, game, n)
    cmb = a.desc.compact_mask_bytes
    for t in range(L):
        acts = rec["actions"][:, t]
        a8 = torch.from_numpy(np.where(acts < 0, 255, acts).astype(np.uint8)).cuda()
        mask, status = a.step(a8, dst=b)
        st = status.cpu().numpy()
        term = (st & 0x80) != 0
        np.testing.assert_array_equal(term, rec["terminal"][:, t + 1] != 0)
        assert not (st & 0x40).any(), "no action of an oracle playout is illegal"
        cur = rec["cur_player"][:, t + 1]
        live = ~term
        np.testing.assert_array_equal((st[live] & 15).astype(np.int64) - 1, cur[live])
        if a.desc.game_kind <= 2:  # board games: outcome in bits 0-2
            r0 = rec["returns"][:, t + 1, 0]
            want = np.where(r0 > 0, 0, np.where(r0 < 0, 1, 2))
            np.testing.assert_array_equal((st[term] & 7), want[term])
        m = mask.cpu().numpy()
        gold = rec["mask"][:, t + 1]
        if cmb < 4:
            got = m.view(np.uint8 if cmb == 1 else np.uint16).reshape(n).astype(np.uint32)
            np.testing.assert_array_equal(got, gold[:, 0])
        else:
            np.testing.assert_array_equal(m.view(np.uint32).reshape(n, -1), gold)
        a, b = b, a
    np.testing.assert_array_equal(a.returns().cpu().numpy(), rec["returns"][:, L])


@pytest.mark.parametrize("game", ["tic_tac_toe", "kuhn_poker", "leduc_poker", "connect_four", "hex(board_size=9)",
                                  "connect_four(rows=5,columns=6,x_in_row=3)"])
@pytest.mark.parametrize("n", [4096, 4097, 1 << 17])
def test_fused_step_in_place_equals_out_of_place(ctx, game, n):
    """step(dst=None) hands the kernels src == dst (the state planes carry no __restrict__ for that reason): the
    in-place step must leave the successor records, masks and status bytes of the out-of-place one, for the
    several-states-per-thread kernels (even n) and the one-state ones (odd n), over a whole game."""
    import torch
    import open_spiel_amd as osa
    a = osa.StateBatch(ctx, game, n)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(n)
    for _ in range(a.desc.max_game_length + 2):
        lm = a.legal_actions_mask()
        noise = torch.rand(lm.shape, device="cuda", generator=gen)
        acts = torch.where(lm.any(1), (lm.to(torch.float32) * (noise + 0.01)).argmax(1),
                           torch.full((n,), 255, device="cuda")).to(torch.uint8)
        acts[::53] = 251   # illegal ones are refused the same way in both forms
        b = osa.StateBatch(ctx, game, n)
        m_out, s_out = a.step(acts, dst=b)
        m_in, s_in = a.step(acts)          # in place
        assert torch.equal(m_in, m_out) and torch.equal(s_in, s_out)
        assert (a.raw_words() == b.raw_words()).all()
        if bool(((s_in & 0x80) != 0).all()):
            break


@pytest.mark.parametrize("game", ["hex(board_size=9)", "hex(board_size=5)", "hex(board_size=11)"])
@pytest.mark.parametrize("n", [4096, 1 << 22])
def test_hex_step_without_the_mask_row(ctx, game, n):
    """osg_step with d_mask == NULL (hex.cc:280-293: the successor's legal actions are its empty cells, so the mask
    row is redundant with the record): same successor records and status bytes as the step that writes the mask —
    both store forms (ordinary below 2^22 states, non-temporal from there) — and ~occupied of the record IS the
    mask the other call wrote; every other game refuses a NULL mask."""
    import torch
    import open_spiel_amd as osa
    a = osa.StateBatch(ctx, game, n)
    a.random_steps(11, 7)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    for _ in range(3):
        lm = a.legal_actions_mask()
        noise = torch.rand(lm.shape, device="cuda", generator=gen)
        acts = torch.where(lm.any(1), (lm.to(torch.float32) * (noise + 0.01)).argmax(1),
                           torch.full((n,), 255, device="cuda")).to(torch.uint8)
        acts[::97] = 251
        b, c = osa.StateBatch(ctx, game, n), osa.StateBatch(ctx, game, n)
        m_full, s_full = a.step(acts, dst=b)
        m_none, s_none = a.step(acts, dst=c, want_mask=False)
        assert m_none is None and torch.equal(s_none, s_full)
        assert (b.raw_words() == c.raw_words()).all()
        # the mask the full step wrote is the set of empty cells of the successor (none once the game is over)
        assert torch.equal(c.legal_actions_mask_bits().view(torch.uint8).reshape(n, -1), m_full.reshape(n, -1))
        a = b
        del lm, noise
    with pytest.raises(osa.OsgError):
        c4 = osa.StateBatch(ctx, "connect_four", 64)
        c4.step(torch.zeros(64, dtype=torch.uint8, device="cuda"), want_mask=False)


@pytest.mark.parametrize("game", ["connect_four", "hex(board_size=9)", "hex(board_size=5)", "leduc_poker", "tic_tac_toe"])
@pytest.mark.parametrize("n", [1, 3, 37, 64, 257, 1000])
def test_observation_ragged_sizes_and_unaligned_output(ctx, game, n):
    """Batch sizes that end mid-wavefront / mid-chunk, and an output pointer that is only 4-byte
    aligned (the C-ABI takes any float*): same tensor as the aligned call, nothing written outside."""
    import torch
    import open_spiel_amd as osa
    b = osa.StateBatch(ctx, game, n)
    b.random_steps(4242 + n, 5)
    size = b.desc.obs_size
    want = b.observation_tensor(0)
    for shift in (1, 2, 3):
        flat = torch.full((n * size + 8,), -7.0, dtype=torch.float32, device="cuda")
        view = flat[shift:shift + n * size].view(n, size)
        b.observation_tensor(0, out=view)
        torch.cuda.synchronize()
        assert torch.equal(view, want), (game, n, shift)
        assert bool((flat[:shift] == -7.0).all()) and bool((flat[shift + n * size:] == -7.0).all())
    guard = torch.full((n * size + 64,), -7.0, dtype=torch.float32, device="cuda")
    b.observation_tensor(0, out=guard[:n * size].view(n, size))
    assert bool((guard[n * size:] == -7.0).all()), "wrote past the end of the tensor"
    assert torch.equal(guard[:n * size].view(n, size), want)


@pytest.mark.parametrize("game", GAMES)
def test_observation_parity(oracle, ctx, game):
    """ObservationTensor / InformationStateTensor for every player at every ply."""
    import torch
    import open_spiel_amd as osa
    og = oracle.Game(game)
    n = 96
    rec = og.random_playouts(99, n, want_obs=True, want_info=True)
    batch = osa.StateBatch(ctx, game, n)
    for t in range(og.max_plies + 1):
        for p in range(og.num_players):
            got = batch.observation_tensor(p).cpu().numpy()
            np.testing.assert_array_equal(got, rec["obs"][:, t, p], err_msg=f"{game} obs p{p} ply {t}")
            if rec["info"] is not None:
                got = batch.information_state_tensor(p).cpu().numpy()
                np.testing.assert_array_equal(got, rec["info"][:, t, p], err_msg=f"{game} info p{p} ply {t}")
        if t < og.max_plies:
            batch.apply_actions(torch.from_numpy(rec["actions"][:, t].astype(np.int32)))


def test_golden_playthroughs_on_device(ctx, goldens):
    """The reference's own playthrough goldens, replayed through the HIP path."""
    import torch
    import open_spiel_amd as osa
    for fname, rec in goldens.items():
        game = rec["game"]
        batch = osa.StateBatch(ctx, game, 3)
        P = batch.num_players
        for blk in rec["states"]:
            if not blk.get("skipped"):
                cur, term, rets = batch.status()
                assert bool(term[0]) == blk["is_terminal"], fname
                assert int(cur[0]) == blk["current_player"], fname
                if "returns" in blk:
                    assert rets[0].tolist() == blk["returns"], fname
                legal = np.nonzero(batch.legal_actions_mask().cpu().numpy()[0])[0].tolist()
                assert legal == blk["legal_actions"], (fname, blk["history"])
                if "chance_outcomes" in blk:
                    probs = batch.chance_outcome_probs().cpu().numpy()[0]
                    for a, pr in blk["chance_outcomes"]:
                        assert "{:.6g}".format(probs[a]) == "{:.6g}".format(pr)
                    assert abs(probs.sum() - 1.0) < 1e-12
                for key, gold in blk["tensors"].items():
                    p = int(key[-1])
                    want = np.array([float(c) for c in gold] if isinstance(gold, str) else gold, np.float32)
                    got = (batch.observation_tensor(p) if key.startswith("obs")
                           else batch.information_state_tensor(p)).cpu().numpy()[0]
                    np.testing.assert_array_equal(got, want, err_msg=f"{fname} {key} {blk['history']}")
            if "action" in blk:
                batch.apply_actions(torch.full((3,), blk["action"], dtype=torch.int32))
        assert bool(batch.is_terminal().all())
        del P


def test_illegal_and_terminal_actions_are_rejected(ctx):
    import torch
    import open_spiel_amd as osa
    b = osa.StateBatch(ctx, "connect_four", 8)
    for _ in range(6):
        b.apply_actions(torch.zeros(8, dtype=torch.int32))       # fill column 0
    before = b.raw_words().copy()
    with pytest.raises(osa.OsgError):
        b.apply_actions(torch.zeros(8, dtype=torch.int32))       # column full -> illegal
    np.testing.assert_array_equal(b.raw_words(), before)         # state untouched
    with pytest.raises(osa.OsgError):
        b.apply_actions(torch.full((8,), 7, dtype=torch.int32))  # out of range
    b.apply_actions(torch.full((8,), -1, dtype=torch.int32))     # -1 = skip, fine
    np.testing.assert_array_equal(b.raw_words(), before)
    # fused kernel flags instead of raising
    a8 = torch.zeros(8, dtype=torch.uint8, device="cuda")
    _, status = b.step(a8)
    assert ((status.cpu().numpy() & 0x40) != 0).all()
    # terminal states accept nothing (FastLoss, connect_four_test.cc:38-59)
    t = osa.StateBatch(ctx, "connect_four", 2)
    for a in [3, 3, 4, 4, 2, 2, 1]:
        t.apply_actions(torch.full((2,), a, dtype=torch.int32))
    assert bool(t.is_terminal().all())
    assert t.returns().tolist() == [[1.0, -1.0]] * 2
    assert not t.legal_actions_mask().any()
    with pytest.raises(osa.OsgError):
        t.apply_actions(torch.full((2,), 0, dtype=torch.int32))


def test_bad_game_strings(ctx):
    import open_spiel_amd as osa
    for bad in ["chess", "connect_four(rows=12,columns=12)", "hex(board_size=20)", "hex(num_cols=32,num_rows=3)", "hex(num_cols=5,num_rows=4,swap=True)", "kuhn_poker(players=11)",
                "leduc_poker(players=11)", "connect_four(foo=1)", "hex(swap=3)"]:
        with pytest.raises(osa.OsgError):
            osa.StateBatch(ctx, bad, 4)
    with pytest.raises(osa.OsgError):
        osa.StateBatch(ctx, "kuhn_poker", 4).observation_tensor(5)  # player out of range


def test_hex_above_128_actions_where_the_boundary_stops(ctx):
    """hex(13) ... hex(19) are served by the batch entry points (the parity tests above run them through every one) and
    by both search layouts (tests/test_gpu_mcts.py; the wave-per-root one without the swap rule only); the fused step keeps
    its one-byte action ids and the solvers the games with a tree: each says so."""
    import torch
    import open_spiel_amd as osa
    b = osa.StateBatch(ctx, "hex(board_size=19)", 64)
    assert b.desc.num_distinct_actions == 361 and b.desc.mask_words == 12 and b.desc.state_words == 48   # (4 x 12 plane words; the meta word folded in)
    b.random_steps(5, 30)
    assert int(b.legal_actions_mask().sum()) == 64 * (361 - 30)
    for layout in (0, 1, 2):                                   # round 6: both layouts take the board (0 picks the wave-per-root one) ...
        res = b.mcts_search(max_simulations=8, layout=layout)
        assert bool((res["child_visits"].sum(1) == 7).all())
    with pytest.raises(osa.OsgError, match="swap"):            # ... but not with the swap rule (its playout is a random fill)
        osa.StateBatch(ctx, "hex(board_size=13,swap=True)", 8).mcts_search(max_simulations=8, layout=2)
    with pytest.raises(osa.OsgError):
        osa.TabularSolver(ctx, "leduc_poker(players=4)")
    with pytest.raises(osa.OsgError, match="one byte"):
        b.step(torch.zeros(64, dtype=torch.uint8, device="cuda"))
    small = osa.StateBatch(ctx, "hex(board_size=15)", 64)      # 225 actions: the fused step serves it
    mask, status = small.step(torch.full((64,), 224, dtype=torch.uint8, device="cuda"))
    assert int((status & 0x40).sum()) == 0 and mask.shape == (64, 32)
    total = small.rollout(3, 3)
    assert bool((total[:, 0] == -total[:, 1]).all()) and bool((total[:, 0].abs() % 2 == 1).all())   # 3 playouts, no draws in hex


def test_clone_and_gather(ctx):
    import torch
    import open_spiel_amd as osa
    b = osa.StateBatch(ctx, "hex(board_size=9)", 300)
    b.random_steps(5, 7)
    c = b.clone()
    np.testing.assert_array_equal(b.raw_words(), c.raw_words())
    idx = torch.arange(299, -1, -3)
    g = b.gather(idx)
    np.testing.assert_array_equal(g.raw_words(), b.raw_words()[:, idx.numpy()])


@pytest.mark.parametrize("game,n_rollouts", [("tic_tac_toe", 20), ("connect_four", 8), ("hex(board_size=9)", 4),
                                             ("hex(board_size=5)", 8), ("kuhn_poker", 16), ("leduc_poker", 16),
                                             ("leduc_poker(players=3)", 8), ("kuhn_poker(players=4)", 8),
                                             ("hex(board_size=13)", 2), ("hex(board_size=19)", 2),
                                             ("connect_four(rows=8,columns=8)", 4), ("leduc_poker(players=5)", 6),
                                             ("leduc_poker(players=10)", 4)])
def test_rollout_replay_parity(oracle, ctx, game, n_rollouts):
    """RandomRolloutEvaluator on the device == the oracle replaying the same
    counter-RNG stream: identical summed returns AND identical ply counts."""
    import torch
    import open_spiel_amd as osa
    og = oracle.Game(game)
    n = 200
    rng = np.random.default_rng(3)
    stop = rng.integers(0, max(og.max_plies // 2, 1), n).astype(np.int32)
    rec = og.random_playouts(11, n, stop=stop)
    roots = osa.StateBatch(ctx, game, n)
    for t in range(og.max_plies):
        if (rec["actions"][:, t] < 0).all():
            break
        roots.apply_actions(torch.from_numpy(rec["actions"][:, t].astype(np.int32)))
    seed, offset = 0xC0FFEE, 1000
    total, steps = roots.rollout(seed, n_rollouts, index_offset=offset, want_steps=True)
    total, steps = total.cpu().numpy(), steps.cpu().numpy()
    for i in range(n):
        hist = rec["actions"][i]
        hist = hist[hist >= 0]
        want, want_steps = og.replay_rollouts(hist, seed, offset + i, n_rollouts)
        np.testing.assert_allclose(total[i], want, rtol=0, atol=1e-12, err_msg=f"{game} root {i}")
        assert steps[i] == want_steps


@pytest.mark.parametrize("game,n_rollouts", [("hex(board_size=9)", 6), ("hex", 3), ("hex(board_size=5,swap=True)", 9),
                                             ("hex(num_cols=3,num_rows=4)", 8), ("hex(board_size=13)", 3),
                                             ("hex(num_cols=17,num_rows=19,swap=True)", 2), ("hex(board_size=2)", 5)])
def test_hex_rollouts_without_ply_counts_take_the_fill_kernel_and_sum_the_same(ctx, game, n_rollouts):
    """osg_rollout with steps == NULL (what the mirror's RandomRolloutEvaluator asks for) runs hex playouts as
    HexT::fill_playout_winner — the same draws and moves without edge labels, the winner read off the filled board —
    and must hand back exactly the sums of the move-by-move kernel (which the test above replays on the oracle): roots at
    every depth incl. the empty board (the swap rule's plies), finished games, boards of six-word planes."""
    import torch
    import open_spiel_amd as osa
    n = 3000
    roots = osa.StateBatch(ctx, game, n)
    idx = torch.arange(n, device="cuda")
    depth_cap = roots.desc.num_distinct_actions + 2
    for t in range(depth_cap):                      # root i is advanced by (i mod depth_cap) random moves: some games end
        m = roots.legal_actions_mask().to(torch.float32)
        live = (m.sum(1) > 0) & (idx % depth_cap > t)
        m[m.sum(1) == 0, 0] = 1.0
        a = torch.multinomial(m, 1).squeeze(1).to(torch.int32)
        roots.apply_actions(torch.where(live, a, torch.full_like(a, -1)))
    assert bool(roots.is_terminal().any()) and not bool(roots.is_terminal().all())
    seed, offset = 0xF111, 77
    fill = roots.rollout(seed, n_rollouts, index_offset=offset)
    stepwise, steps = roots.rollout(seed, n_rollouts, index_offset=offset, want_steps=True)
    assert torch.equal(fill, stepwise), game
    assert int(steps.sum()) > 0


def test_random_steps_counters(ctx):
    import open_spiel_amd as osa
    b = osa.StateBatch(ctx, "connect_four", 1 << 12)
    counters = b.random_steps(1, 64)
    ctx.synchronize()
    steps, episodes = counters.tolist()
    assert steps == 64 * (1 << 12)
    assert episodes > 0  # 64 plies always finishes at least one 42-ply game


@pytest.mark.parametrize("game", ["connect_four", "hex(board_size=9)", "leduc_poker", "tic_tac_toe"])
@pytest.mark.parametrize("n", [1, 2, 63, 65])
def test_tiny_and_ragged_batches(oracle, ctx, game, n):
    """Batch sizes around the wave / pair boundaries (the connect_four fast path handles pairs; odd
    sizes take the generic kernel): fused step and tensor pack still agree with the oracle."""
    import torch
    import open_spiel_amd as osa
    og = oracle.Game(game)
    rec = og.random_playouts(5, n, want_obs=True)
    a, b = osa.StateBatch(ctx, game, n), osa.StateBatch(ctx, game, n)
    for t in range(min(og.max_plies, 12)):
        acts = rec["actions"][:, t]
        a8 = torch.from_numpy(np.where(acts < 0, 255, acts).astype(np.uint8)).cuda()
        _, status = a.step(a8, dst=b)
        st = status.cpu().numpy()
        np.testing.assert_array_equal((st & 0x80) != 0, rec["terminal"][:, t + 1] != 0)
        got = b.observation_tensor(0).cpu().numpy()
        np.testing.assert_array_equal(got, rec["obs"][:, t + 1, 0])
        a, b = b, a


def test_observation_and_information_state_strings_match_the_reference_playthroughs(goldens, ctx):
    """ObservationString / InformationStateString of every state of the reference's playthroughs
    (open_spiel/integration_tests/playthroughs/*.txt via tests/golden/playthroughs.json), formatted by
    the library from the packed device words."""
    import torch
    import open_spiel_amd as osa
    checked = 0
    for name, play in goldens.items():
        batch = osa.StateBatch(ctx, play["game"], 1)
        for blk in play["states"]:
            if not blk.get("skipped"):
                for p_str, want in blk.get("obs_str", {}).items():
                    assert batch.observation_string(0, int(p_str)) == want, (name, blk["history"], p_str)
                    checked += 1
                if batch.desc.info_size:
                    for p_str, want in blk.get("info_str", {}).items():
                        assert batch.information_state_string(0, int(p_str)) == want, (name, blk["history"], p_str)
                # the dump strips trailing blanks per line
                got = [l.rstrip() for l in batch.state_string(0).rstrip("\n").split("\n")]
                assert got == [l.rstrip() for l in blk["to_string"].rstrip("\n").split("\n")], (name, blk["history"])
                cp = blk["current_player"]
                names = [batch.action_string(0, -1 if cp == -1 else cp, a) for a in blk["legal_actions"]]
                assert names == blk["string_legal_actions"], (name, blk["history"])
            if "action" in blk:
                batch.apply_actions(torch.tensor([blk["action"]], dtype=torch.int32))
    assert checked > 100


@pytest.mark.parametrize("game", ["hex(board_size=4,string_rep=explicit)", "hex", "hex(board_size=19)",
                                  "hex(board_size=13,string_rep=explicit)", "connect_four(rows=5,columns=6,x_in_row=3)",
                                  "connect_four(rows=9,columns=10,x_in_row=5)", "leduc_poker(players=4)", "leduc_poker(players=10)",
                                  "kuhn_poker(players=3)", "leduc_poker(players=3)", "tic_tac_toe"])
def test_observation_strings_match_the_oracle(oracle, ctx, game):
    """The same strings on random trajectories of the variants the playthroughs do not cover."""
    import torch
    import open_spiel_amd as osa
    og = oracle.Game(game)
    n = 24
    rec = og.random_playouts(7, n)
    batch = osa.StateBatch(ctx, game, n)
    states = [og.new_initial_state() for _ in range(n)]
    for t in range(og.max_plies + 1):
        for i in range(0, n, 5):
            for p in range(og.num_players):
                assert batch.observation_string(i, p) == states[i].observation_string(p), (game, i, t, p)
            assert batch.state_string(i) == str(states[i]), (game, i, t)
            cp = states[i].current_player()
            for a in states[i].legal_actions():
                assert batch.action_string(i, cp, a) == states[i].action_to_string(cp, a), (game, i, t, a)
        if t == og.max_plies:
            break
        acts = rec["actions"][:, t].astype(np.int32)
        batch.apply_actions(torch.from_numpy(acts))
        for i in range(n):
            if acts[i] >= 0:
                states[i].apply_action(int(acts[i]))


def test_gather_index_bounds_and_buffer_checks(ctx):
    """Out-of-range gather indices never read out of bounds: a host index list is refused, a device index
    list yields the initial state for the bad entries; caller-supplied step / tensor buffers of the wrong dtype,
    size or device are refused instead of being handed to a kernel."""
    import ctypes as C
    import torch
    import open_spiel_amd as osa
    from open_spiel_amd._abi import lib
    b = osa.StateBatch(ctx, "connect_four", 64)
    b.random_steps(3, 10)
    words = b.raw_words()
    idx = torch.tensor([0, 5, 63, 64, -1, 1 << 40], dtype=torch.int64)
    g = b.gather(idx)                                     # device index path: bad entries -> initial state,
    with pytest.raises(osa.OsgError, match="3 illegal"):  # counted, and reported at the next synchronisation point
        ctx.synchronize()
    ctx.synchronize()                                     # (reported once)
    got = g.raw_words()
    np.testing.assert_array_equal(got[:, :3], words[:, [0, 5, 63]])
    assert (got[:, 3:] == 0).all()
    dst = osa.StateBatch(ctx, "connect_four", 3)
    bad = np.array([0, 64, 2], np.int64)
    assert lib().osg_batch_gather(dst._h, b._h, bad.ctypes.data, 1) != 0      # host index path: refused
    assert b"out of range" in lib().osg_last_error()
    ok = np.array([0, 63, 2], np.int64)
    assert lib().osg_batch_gather(dst._h, b._h, ok.ctypes.data, 1) == 0
    a8 = torch.zeros(64, dtype=torch.uint8, device="cuda")
    for wrong in (torch.zeros(64, dtype=torch.int32, device="cuda"), torch.zeros(63, dtype=torch.uint8, device="cuda"),
                  torch.zeros(64, dtype=torch.uint8), torch.zeros(128, dtype=torch.uint8, device="cuda")[::2]):
        with pytest.raises(osa.OsgError):
            b.step(wrong)
    with pytest.raises(osa.OsgError):
        b.step(a8, dst=osa.StateBatch(ctx, "connect_four", 32))
    with pytest.raises(osa.OsgError):
        b.observation_tensor(0, out=torch.zeros(64 * 126, dtype=torch.float64, device="cuda"))
    with pytest.raises(osa.OsgError):
        b.observation_tensor(0, out=torch.zeros(64 * 125, dtype=torch.float32, device="cuda"))
    b.observation_tensor(0, out=torch.zeros((64, 126), dtype=torch.float32, device="cuda"))


def test_objects_may_outlive_their_context():
    """Batches, solvers and trees hold a reference on the engine context: destroying them after the context
    (garbage-collection order in a binding) is safe, creating new ones on a destroyed context is refused."""
    import gc
    import torch
    import open_spiel_amd as osa
    c = osa.Context(0)
    b = osa.StateBatch(c, "tic_tac_toe", 16)
    s = osa.TabularSolver(c, "kuhn_poker")
    s.evaluate_and_update_policy(3)
    h = c._h
    c.close()
    with pytest.raises(osa.OsgError):
        osa.lib()  # keep the name used
        import ctypes as C
        out = C.c_void_p()
        from open_spiel_amd._abi import check
        check(osa.lib().osg_batch_create(h, b"tic_tac_toe", 4, C.byref(out)))
    assert b.legal_actions_mask().shape == (16, 9)      # still usable: the context lives until its last object goes
    del b, s
    gc.collect()
    torch.cuda.synchronize()
