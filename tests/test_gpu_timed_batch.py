"""The batch bench.py times IS a batch the CPU reference can check — at BASELINE.json's full sizes.

SURVEY.md 8(d) fixes the synthetic inputs on a counter stream so that "CPU oracle and GPU generate
identical batches".  `osg_synth_batch` (device) and `osgo_synth_batch` (oracle/spiel_oracle_capi.cpp, bound to
the GENUINE reference build oracle/_ref/libspiel_ref.so where it exists, else to the restatement) implement
the same recipe; these tests regenerate the whole batch on the host threads and compare EVERY state:

  config 2  connect_four, 2^20 states, seed 0x5EED, depth mod 36: depth, action, the state itself (as its
            ObservationTensor), legal mask, player to move; then ONE launch of the fused step kernel exactly
            as bench.py issues it (src -> dst, out of place): successor state, successor legal mask, status byte
            (terminal / player to move / outcome) and Returns() of all 2^20 states.
            Reference: spiel.cc:441-451, connect_four.cc:122-156, 277-285, 312-328.
  config 4  hex(board_size=9), 2^16 roots, depth mod 40: all roots state for state; then the wave-per-root
            search of ALL 2^16 roots at a reduced simulation count against the oracle's replay-mode MCTSBot:
            best action, visit vector, reward vector.  Reference: mcts.cc:273-467, hex.cc:229-293.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x5EED


def _threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


@pytest.fixture(scope="module")
def ctx():
    import open_spiel_amd as osa
    return osa.Context(0)


@pytest.fixture(scope="module")
def checker(oracle):
    """The genuine reference build where it is present (it is on the GPU box: the prebuilt file travels),
    else the restatement (itself equal to the genuine build call for call: tests/test_oracle_vs_reference.py)."""
    import reference_py
    if reference_py.available():
        return reference_py, "reference"
    return oracle, "port"


def _check_states(torch, batch, rec, suffix, n):
    """Every state of `batch` against the checker's record `suffix` ("0" before / "1" after the action):
    the state itself through ObservationTensor(0), the legal mask, the player to move, IsTerminal."""
    obs = batch.observation_tensor(0).to(torch.uint8).cpu().numpy()
    np.testing.assert_array_equal(obs, rec["obs" + suffix])
    del obs
    bits = batch.legal_actions_mask_bits().cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(bits, rec["mask" + suffix])
    cur, term, rets = [t.cpu().numpy() for t in batch.status()]
    np.testing.assert_array_equal(term, rec["term" + suffix])
    np.testing.assert_array_equal(cur, rec["cur" + suffix])
    return rets


def test_config2_timed_batch_every_state_against_the_reference(ctx, checker):
    import torch
    import open_spiel_amd as osa
    impl, kind = checker
    n, depth_mod = 1 << 20, 36
    src = osa.StateBatch(ctx, "connect_four", n)
    actions, depth = src.synth(SEED, depth_mod)
    rec = impl.Game("connect_four").synth_batch(SEED, n, depth_mod, threads=_threads())
    # the generator: same depths, same actions, same states
    np.testing.assert_array_equal(depth.cpu().numpy(), rec["depth"])
    np.testing.assert_array_equal(actions.cpu().numpy().astype(np.int16), rec["action"])
    assert rec["term0"].sum() == 0 and np.bincount(rec["depth"], minlength=depth_mod).min() > n // depth_mod // 2
    _check_states(torch, src, rec, "0", n)
    # ONE launch as bench.py times it: out of place, mask + status buffers of the batch
    dst = osa.StateBatch(ctx, "connect_four", n)
    mask, status = src.step_buffers()
    src.step(actions, dst=dst, mask=mask, status=status)
    st = status.cpu().numpy()
    assert ((st & 0x40) == 0).all(), "every synthetic action is legal"
    term1 = rec["term1"] != 0
    np.testing.assert_array_equal((st & 0x80) != 0, term1)
    np.testing.assert_array_equal((st[~term1] & 15).astype(np.int64) - 1, rec["cur1"][~term1])
    # outcome of the finished games: 0 player 0 wins, 1 player 1 wins, 2 draw  <->  Returns()
    want_outcome = np.where(rec["rets1"][:, 0] > 0, 0, np.where(rec["rets1"][:, 0] < 0, 1, 2))
    np.testing.assert_array_equal((st[term1] & 7), want_outcome[term1])
    np.testing.assert_array_equal(mask.cpu().numpy().reshape(n), (rec["mask1"][:, 0] & 0xFF).astype(np.uint8))
    rets = _check_states(torch, dst, rec, "1", n)
    np.testing.assert_array_equal(rets, rec["rets1"])
    # the source batch is untouched by the out-of-place launch (every timed launch does identical work)
    _check_states(torch, src, rec, "0", n)
    assert term1.sum() > 0, "the batch contains moves that end the game"
    print(f"config 2: {n} states checked against the {kind}")


def test_config2_shards_are_slices_of_one_stream(ctx):
    """Rank r of an N-GPU run generates indices [r * n, (r + 1) * n): the same states as that slice of one big batch."""
    import torch
    import open_spiel_amd as osa
    n = 1 << 16
    whole = osa.StateBatch(ctx, "connect_four", 2 * n)
    a_whole, d_whole = whole.synth(SEED, 36)
    part = osa.StateBatch(ctx, "connect_four", n)
    a_part, d_part = part.synth(SEED, 36, index_offset=n)
    assert torch.equal(a_part, a_whole[n:]) and torch.equal(d_part, d_whole[n:])
    np.testing.assert_array_equal(part.raw_words(), whole.raw_words()[:, n:])


@pytest.mark.parametrize("game,depth_mod,n", [("tic_tac_toe", 5, 1 << 16), ("hex(board_size=5)", 12, 1 << 14),
                                              ("connect_four(rows=5,columns=6,x_in_row=3)", 10, 1 << 14),
                                              ("kuhn_poker", 2, 1 << 14), ("leduc_poker", 3, 1 << 14),
                                              ("leduc_poker(players=3)", 4, 1 << 12),
                                              # sizes at which the tensors take the piece-form kernels (>= 2^24 floats),
                                              # with odd state counts so that the tensor ends inside a 16-byte piece
                                              ("tic_tac_toe", 5, (1 << 20) + 1), ("connect_four", 20, (1 << 18) + 1),
                                              ("connect_four(egocentric_obs_tensor=True)", 20, (1 << 18) + 3),
                                              ("hex(board_size=6)", 20, (1 << 16) + 3), ("hex(board_size=11)", 60, (1 << 15) + 1),
                                              ("hex(num_cols=7,num_rows=5)", 12, (1 << 16) + 2)])
def test_synth_batch_other_games(ctx, checker, game, depth_mod, n):
    """The generator is one template over the games: chance nodes drawn by their distribution, ragged depths."""
    import torch
    import open_spiel_amd as osa
    impl, _ = checker
    b = osa.StateBatch(ctx, game, n)
    actions, depth = b.synth(77, depth_mod, index_offset=123456789)
    rec = impl.Game(game).synth_batch(77, n, depth_mod, first=123456789, threads=_threads())
    np.testing.assert_array_equal(depth.cpu().numpy(), rec["depth"])
    np.testing.assert_array_equal(actions.cpu().numpy().astype(np.int16), rec["action"])
    _check_states(torch, b, rec, "0", n)
    b.apply_actions(actions.to(torch.int32))
    rets = _check_states(torch, b, rec, "1", n)
    np.testing.assert_array_equal(rets, rec["rets1"])


@pytest.mark.parametrize("game,depth_mod,n", [("kuhn_poker", 3, (1 << 22) + 1), ("leduc_poker", 8, (1 << 21) + 3),
                                              ("leduc_poker(suit_isomorphism=True)", 8, (1 << 21) + 1),
                                              ("connect_four(egocentric_obs_tensor=True)", 30, (1 << 18) + 1),
                                              ("tic_tac_toe", 7, (1 << 20) + 2), ("hex(board_size=9)", 60, (1 << 15) + 3)])
def test_piece_form_tensors_equal_the_span_kernels(ctx, game, depth_mod, n):
    """From 2^24 floats on the tensors are packed by the piece-form kernels (one aligned 16-byte piece per thread);
    below, by the span-per-wavefront kernels that the small-size tests compare with the oracle entry by entry.  The
    same states packed whole (piece form) and in chunks below the threshold (span kernels) must give the same bytes:
    observation and information-state tensors, for every player and for "the player to move" (-1)."""
    import torch
    import open_spiel_amd as osa
    b = osa.StateBatch(ctx, game, n)
    b.synth(99, depth_mod)
    kinds = [("observation_tensor", b.desc.obs_size)]
    if b.desc.info_size:
        kinds.append(("information_state_tensor", b.desc.info_size))
    for name, size in kinds:
        assert n * size >= 1 << 24
        chunk = ((1 << 24) - 1) // size
        for player in [-1] + list(range(b.num_players)):
            whole = getattr(b, name)(player)
            for first in range(0, n, chunk):
                idx = torch.arange(first, min(n, first + chunk), device="cuda")
                part = getattr(b.gather(idx), name)(player)
                assert torch.equal(whole[first:first + idx.numel()], part), (game, name, player, first)
            del whole


@pytest.mark.parametrize("game,depth_mod,n", [("kuhn_poker", 3, (1 << 22) + 1), ("leduc_poker", 8, (1 << 21) + 3),
                                              ("kuhn_poker(players=3)", 4, (1 << 21) + 1)])
def test_piece_form_poker_tensors_against_the_checker(ctx, checker, game, depth_mod, n):
    """The poker games' tensors at the sizes that take the piece-form kernels (>= 2^24 floats, odd state counts) DIRECTLY
    against the CPU reference — kuhn_poker.cc:72-107, leduc_poker.cc:103-192 through State::ObservationTensor /
    InformationStateTensor (spiel.cc:908-945) — entry by entry: both tensor kinds, every player, all states of the
    synthetic stream and of their successors (chance nodes, decision nodes and terminal states among them)."""
    import torch
    import open_spiel_amd as osa
    impl, kind = checker
    b = osa.StateBatch(ctx, game, n)
    actions, _ = b.synth(4242, depth_mod, index_offset=987654321)
    og = impl.Game(game)
    checked = 0
    for after in (False, True):       # the synthetic states, then their successors (terminal states among them)
        if after:
            b.apply_actions(actions.to(torch.int32))
            # (kuhn's stream stops at depth MaxGameLength() - 1 = 2 deals: its successors are decision nodes; leduc's end games)
            assert game != "leduc_poker" or int(b.is_terminal().sum().item()) > 0
        for which, name, size in ((0, "observation_tensor", b.desc.obs_size), (1, "information_state_tensor", b.desc.info_size)):
            assert n * size >= 1 << 24, "below the piece-form threshold"
            for player in range(b.num_players):
                want = og.synth_tensors(4242, n, depth_mod, which, player, first=987654321, threads=_threads(), after=after)
                got = getattr(b, name)(player)
                assert got.dtype == torch.float32 and tuple(got.shape) == (n, size)
                got8 = got.to(torch.uint8)
                assert torch.equal(got8.to(torch.float32), got), "entries are small integers"
                np.testing.assert_array_equal(got8.cpu().numpy(), want, err_msg=f"{game} {name} player {player} after={after}")
                checked += n * size
                del got, got8, want
    print(f"{game}: {checked} tensor entries of {n} states against the {kind}")


def test_synth_batch_argument_checks(ctx):
    import open_spiel_amd as osa
    b = osa.StateBatch(ctx, "connect_four", 64)
    with pytest.raises(osa.OsgError):
        b.synth(1, 0)
    with pytest.raises(osa.OsgError):
        b.synth(1, 43)
    one_row = osa.StateBatch(ctx, "hex(num_rows=1,num_cols=4)", 8)
    with pytest.raises(osa.OsgError):
        one_row.synth(1, 2)


def test_config4_all_roots_against_the_oracle_replay(ctx, oracle, checker):
    """All 2^16 hex(9) roots of config 4: the roots state for state against the reference rules, then the search of
    every root (wave-per-root layout, reduced simulation count) against the replay-mode MCTSBot of the oracle
    (restatement: the reference has no replay hook; the restatement's MCTSBot equals the genuine one tree for tree
    under shared draws, tests/test_oracle_vs_reference.py)."""
    import torch
    import open_spiel_amd as osa
    impl, kind = checker
    n, depth_mod, sims = 1 << 16, 40, 64
    game = "hex(board_size=9)"
    roots = osa.StateBatch(ctx, game, n)
    _, depth = roots.synth(SEED, depth_mod)
    rec = impl.Game(game).synth_batch(SEED, n, depth_mod, threads=_threads(), want_after=False)
    np.testing.assert_array_equal(depth.cpu().numpy(), rec["depth"])
    _check_states(torch, roots, rec, "0", n)
    got = roots.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=1, seed=SEED, index_offset=0, layout=2)
    want = oracle.Game(game).synth_mcts_replay(SEED, n, depth_mod, 2.0, sims, 1, SEED, 2, threads=_threads())
    np.testing.assert_array_equal(got["best_action"].cpu().numpy(), want["best_action"])
    np.testing.assert_array_equal(got["child_visits"].cpu().numpy(), want["child_visits"])
    np.testing.assert_array_equal(got["child_reward"].cpu().numpy(), want["child_reward"])
    # a shard of the same roots (rank 3 of 8) searches the same trees: global root index = index_offset + i
    first, count = 3 * (n // 8), n // 8
    shard = osa.StateBatch(ctx, game, count)
    shard.synth(SEED, depth_mod, index_offset=first)
    part = shard.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=1, seed=SEED, index_offset=first, layout=2)
    assert torch.equal(part["child_visits"], got["child_visits"][first:first + count])
    assert torch.equal(part["child_reward"], got["child_reward"][first:first + count])
    print(f"config 4: {n} roots checked against the {kind}, {n} searches x {sims} simulations against the oracle replay")


def test_config3_kuhn_cfr_1000_iterations_against_the_reference(ctx, checker):
    """Config 3 at its stated size: 1 000 EvaluateAndUpdatePolicy of kuhn_poker in ONE launch of
    k_cfr_small<lds, owner> — the kernel bench.py times — against the reference's CFRSolver run for 1 000 iterations
    (cfr.cc:263-469): cumulative regrets and cumulative policy of all 12 infostates to 1e-9 relative, average policy
    to 1e-6 (north_star); then the launch split 100 + 900 gives the same tables bit for bit."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import parity
    import open_spiel_amd as osa
    impl, kind = checker
    s = osa.TabularSolver(ctx, "kuhn_poker")
    s.evaluate_and_update_policy(1000)
    assert s.last_kernel() == "k_cfr_small<lds, owner>"
    t = s.tables()
    rec = parity.cfr_tables(impl, "kuhn_poker", "cfr", 1000, t["keys"], t["nact"], t["regrets"], t["cum_policy"], t["avg_policy"])
    assert rec["max_table_rel_error"] <= 1e-9 and rec["max_average_policy_abs_error"] <= 1e-6
    s2 = osa.TabularSolver(ctx, "kuhn_poker")
    s2.evaluate_and_update_policy(100)
    s2.evaluate_and_update_policy(900)
    t2 = s2.tables()
    for name in ("regrets", "cum_policy", "cur_policy"):
        assert np.array_equal(t[name], t2[name]), name
    print(f"config 3: 1000 iterations, 12 infostates against the {kind}: tables within {rec['max_table_rel_error']:.2e} relative, "
          f"average policy within {rec['max_average_policy_abs_error']:.2e}")


def test_three_player_leduc_cfr_at_size_against_the_reference(ctx, checker):
    """The largest tree served — leduc_poker(players=3): 1 831 601 histories, 25 800 infostates — on the kernel bench.py
    times for it (k_cfr_sub<forest>, the default) against the reference's own solvers, at size: 2 iterations of CFRSolver and
    of CFRPlusSolver (cfr.cc:263-469; 6.7 s per iteration on one host core, the two run in parallel threads), every regret and
    cumulative-policy cell of all 25 800 infostates to 1e-12 relative, the average policy to 1e-6; then the device judge
    (k_geval_*) against the reference's ExpectedReturns (expected_returns.cc) of the average policy and against
    TabularBestResponse's value for player 0 (best_response.cc:194-227; the full three-player NashConv costs the reference
    68 s: tests/test_gpu_cfr.py checks NashConv whole on smaller trees)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import parity
    import open_spiel_amd as osa
    impl, kind = checker
    game, iters = "leduc_poker(players=3)", 2
    device = {}
    for name, kwargs in (("cfr", {}), ("cfr_plus", dict(linear_averaging=True, regret_matching_plus=True))):
        s = osa.TabularSolver(ctx, game, **kwargs)
        assert s.num_histories == 1831601 and s.num_infostates == 25800
        s.evaluate_and_update_policy(iters)
        assert s.last_kernel() == "k_cfr_sub<forest>", s.last_kernel()
        ev = s.evaluate_policy()
        device[name] = {"tables": s.tables(), "expected_returns": ev["expected_returns"],
                        "best_response_values": ev["best_response_values"], "nash_conv": ev["nash_conv"]}
        assert abs(ev["nash_conv"] - (ev["best_response_values"] - ev["expected_returns"]).sum()) <= 1e-11
        del s
    rec = parity.cfr_large_tree(impl, game, ("cfr", "cfr_plus"), iters, device, threads=2, best_response_player=0)
    assert rec["infostates"] == 25800 and rec["max_table_rel_error"] <= 1e-12 and rec["max_average_policy_abs_error"] <= 1e-6
    print(f"3-player leduc at size against the {kind}: {iters} iterations of CFR and CFR+, 25 800 infostates: tables within "
          f"{rec['max_table_rel_error']:.2e} relative, expected returns within {rec['max_expected_returns_abs_error']:.2e}, "
          f"best-response value of player 0 within {rec['max_best_response_value_abs_error']:.2e} "
          f"(CPU {rec['cpu_seconds_cfr']:.0f} s + {rec['cpu_seconds_cfr_plus']:.0f} s in parallel)")


@pytest.mark.slow
def test_three_player_leduc_nash_conv_whole_against_the_reference(ctx, checker):
    """The full NashConv of 3-player leduc (three best responses + the on-policy values: ~70 s of reference CPU) —
    deselect with -m 'not slow' when in a hurry."""
    import open_spiel_amd as osa
    impl, kind = checker
    s = osa.TabularSolver(ctx, "leduc_poker(players=3)")
    s.evaluate_and_update_policy(1)
    o = impl.Solver(impl.Game("leduc_poker(players=3)"), "cfr")
    o.iterate(1)
    assert abs(s.nash_conv() - o.nash_conv()) <= 1e-10


@pytest.mark.parametrize("warm_batches", [0, 3])
def test_config5_full_minibatch_against_the_frozen_table_replay(ctx, checker, warm_batches):
    """Config 5 at its timed size: ONE WHOLE 2^20-trajectory leduc_poker mini-batch of k_mccfr_resident_flat — the
    kernel and the mini-batch size bench.py times — against the CPU's UpdateRegrets replay of the same 2^20
    trajectories on the frozen table (external_sampling_mccfr.cc:122-186 on the device's counter streams;
    oracle/spiel_oracle_capi.cpp osgo_mccfr_frozen_replay, bound to the genuine reference build where it exists):
    every regret and average-policy increment of all 936 infostates.  Tolerance 1e-11 x the cell's |increment|
    mass (summation order only; oracle/parity.py says why).  From the initial table (warm_batches = 0: the uniform
    policy) and from a table three mini-batches old (a non-uniform policy, like the bench's later mini-batches)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import parity
    import open_spiel_amd as osa
    impl, kind = checker
    game, n = "leduc_poker", 1 << 20
    s = osa.TabularSolver(ctx, game, mccfr=True)
    first = 0
    for _ in range(warm_batches):
        s.run_mccfr(SEED, 1 << 16, first_trajectory=first)
        first += 1 << 16
    before = s.tables()
    s.mccfr_sample(SEED, n, first_trajectory=first)
    # the form bench.py times: tables in LDS, the read-only tree records read through L2 (two workgroups per CU)
    assert s.last_kernel() == os.environ.get("OSG_EXPECT_MCCFR_KERNEL", "k_mccfr_resident_flat<tree in L2>")
    dreg, dcum = [t.cpu().numpy().copy() for t in s.mccfr_delta_tables()]
    rec = parity.mccfr_minibatch(impl, game, before["keys"], before["nact"], before["regrets"], dreg, dcum, SEED, first, n,
                                 _threads())
    # from the uniform policy 2^20 trajectories reach every infostate; a trained policy gives some actions probability 0
    assert rec["trajectories"] == n and (rec["infostates_visited"] == 936 if warm_batches == 0 else rec["infostates_visited"] > 600)
    # the fold adds exactly these deltas: table after == table before + delta, cell by cell
    s.mccfr_apply_deltas()
    after = s.tables()
    np.testing.assert_array_equal(after["regrets"], before["regrets"] + dreg)
    np.testing.assert_array_equal(after["cum_policy"], before["cum_policy"] + dcum)
    print(f"config 5: {n} trajectories ({rec['infostate_visits']} infostate visits, {rec['cells']} cells) against the {kind}: "
          f"worst cell {rec['max_error_over_mass']:.2e} of its mass")
