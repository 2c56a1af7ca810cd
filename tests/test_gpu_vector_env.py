"""The batched RL environment (open_spiel_amd/vector_env.py, osg_env_step) against a restatement
of the reference's `rl_environment.Environment` (python/rl_environment.py:257-452) built on the
CPU oracle's State, fed with the same chance draws."""
import numpy as np
import pytest

from test_gpu_fullsize import CounterRng

pytestmark = pytest.mark.gpu

FIRST, MID, LAST = 0, 1, 2


@pytest.fixture(scope="module")
def ctx():
    import open_spiel_amd as osa
    return osa.Context(0)


class OracleEnvironment:
    """rl_environment.Environment over the oracle (one environment), chance events sampled like the
    device does: stream (seed, env index, step index), SampleAction's CDF scan."""

    def __init__(self, og, seed, index, discount, use_observation):
        self.og, self.seed, self.index, self.discount = og, seed, index, discount
        self.use_observation = use_observation
        self.state = None
        self.should_reset = True

    def _sample_external_events(self, t):
        rng = CounterRng(self.seed, self.index, t)
        while self.state.is_chance_node():
            z, acc, pick = rng.unit(), 0.0, None
            outcomes = self.state.chance_outcomes()
            for a, pr in outcomes:
                if acc <= z < acc + pr:
                    pick = a
                    break
                acc += pr
            self.state.apply_action(outcomes[-1][0] if pick is None else pick)

    def _time_step(self, first):
        s, P = self.state, self.og.num_players
        step_type = FIRST if first else (LAST if s.is_terminal() else MID)
        self.should_reset = step_type == LAST
        obs = [s.observation_tensor(p) if self.use_observation else s.information_state_tensor(p) for p in range(P)]
        legal = [s.legal_actions(p) for p in range(P)]
        rewards = None if first else s.returns() if s.is_terminal() else [0.0] * P
        discounts = None if first else [0.0 if step_type == LAST else self.discount] * P
        return dict(info_state=obs, legal_actions=legal, current_player=s.current_player(), rewards=rewards,
                    discounts=discounts, step_type=step_type)

    def reset(self, t):
        self.should_reset = False
        self.state = self.og.new_initial_state()
        self._sample_external_events(t)
        return self._time_step(first=True)

    def step(self, action, t):
        if self.should_reset:  # rl_environment.py:405-406
            return self.reset(t)
        self.state.apply_action(int(action))
        self._sample_external_events(t)
        return self._time_step(first=False)


def _compare(ts, want, i, P, A, what):
    assert int(ts.step_type[i]) == want["step_type"], what
    assert int(ts.observations["current_player"][i]) == want["current_player"], what
    for p in range(P):
        np.testing.assert_array_equal(ts.observations["info_state"][p][i].cpu().numpy(), want["info_state"][p], what)
        legal = np.nonzero(ts.observations["legal_actions"][p][i].cpu().numpy())[0].tolist()
        assert legal == want["legal_actions"][p], what
    if want["rewards"] is None:
        assert ts.rewards[i].tolist() == [0.0] * P and ts.discounts[i].tolist() == [0.0] * P, what
    else:
        assert ts.rewards[i].tolist() == want["rewards"], what
        assert ts.discounts[i].tolist() == want["discounts"], what


@pytest.mark.parametrize("game,obs_type,steps,compact", [
    ("kuhn_poker", None, 14, False), ("leduc_poker", None, 30, False), ("leduc_poker", "observation", 20, False),
    ("tic_tac_toe", None, 25, False), ("connect_four", None, 60, False), ("hex(board_size=5)", None, 40, False),
    ("kuhn_poker(players=3)", None, 16, False),
    # the compact side arrays (osg_env_step_compact: action bytes, one flag byte, rewards as doubled signed bytes) hand
    # out the same TimeSteps
    ("connect_four", None, 60, True), ("leduc_poker", None, 30, True), ("kuhn_poker(players=3)", None, 16, True),
    ("tic_tac_toe", None, 25, True), ("hex(board_size=5)", None, 40, True),
])
def test_batched_environment_matches_rl_environment(oracle, ctx, game, obs_type, steps, compact):
    import torch
    import open_spiel_amd as osa
    n, seed, offset, discount = 48, 0xE27, 7000, 0.99
    og = oracle.Game(game)
    P, A = og.num_players, og.num_distinct_actions
    ot = osa.ObservationType.OBSERVATION if obs_type == "observation" else None
    env = osa.BatchedEnvironment(ctx, game, n, discount=discount, observation_type=ot, seed=seed, index_offset=offset,
                                 compact=compact)
    use_obs = obs_type == "observation" or og.information_state_tensor_size == 0
    refs = [OracleEnvironment(og, seed, offset + i, discount, use_obs) for i in range(n)]
    ts = env.reset()
    want = [r.reset(0) for r in refs]
    agent = np.random.default_rng(5)
    finished = 0
    for t in range(1, steps + 1):
        for i in range(n):
            _compare(ts, want[i], i, P, A, f"{game} env {i} step {t - 1}")
        actions = np.zeros(n, np.int32)
        for i, w in enumerate(want):
            legal = w["legal_actions"][w["current_player"]] if w["current_player"] >= 0 else []
            actions[i] = agent.choice(legal) if legal else 0  # ignored: the environment restarts
            finished += w["step_type"] == LAST
        ts = env.step(torch.from_numpy(actions))
        want = [r.step(actions[i], t) for i, r in enumerate(refs)]
    assert finished > 0, "the run must cover episode ends and restarts"


def test_reset_if_done_like_sync_vector_env(ctx):
    """vector_env.py:36-62: finished environments restart at once, unreset steps are returned too."""
    import torch
    import open_spiel_amd as osa
    n = 32
    env = osa.BatchedEnvironment(ctx, "tic_tac_toe", n, seed=1)
    ts = env.reset()
    assert bool((ts.step_type == FIRST).all())
    seen_done = False
    for _ in range(12):
        legal = ts.observations["legal_actions"][0] | ts.observations["legal_actions"][1]
        actions = legal.to(torch.float32).argmax(1).to(torch.int32)
        ts, rewards, done, unreset = env.step(actions, reset_if_done=True)
        assert bool((unreset.step_type[done] == LAST).all())
        assert bool((ts.step_type[done] == FIRST).all()) and bool((ts.step_type[~done] == MID).all())
        assert bool((rewards[~done] == 0).all())
        assert bool((ts.observations["current_player"][done] == 0).all())
        seen_done |= bool(done.any())
    assert seen_done


def test_illegal_action_is_reported(ctx):
    import torch
    import open_spiel_amd as osa
    env = osa.BatchedEnvironment(ctx, "connect_four", 4)
    env.reset()
    with pytest.raises(osa.OsgError):
        env.step(torch.full((4,), 9, dtype=torch.int32))  # IllegalActionError in the reference
    with pytest.raises(ValueError):
        osa.BatchedEnvironment(ctx, "tic_tac_toe", 4, observation_type=osa.ObservationType.INFORMATION_STATE)
    with pytest.raises(ValueError):
        osa.BatchedEnvironment(ctx, "tic_tac_toe", 4, discount=1.5)


@pytest.mark.parametrize("game", ["connect_four", "connect_four(rows=5,columns=6,x_in_row=3)"])
def test_env_step_two_per_thread_equals_one_per_thread(ctx, game):
    """k_env_step_x2 (two environments per thread, 16-byte accesses; taken for even batches of two-plane two-player games
    with aligned side arrays) against k_env_step (an odd batch takes it): the same states, step types, players, rewards,
    masks and reset flags step by step over several episodes, illegal actions refused and counted the same way."""
    import torch
    import open_spiel_amd as osa
    from open_spiel_amd._abi import check, lib
    n = 1 << 12

    def make(count):
        b = osa.StateBatch(ctx, game, count)
        return dict(b=b, reset=torch.ones(count, dtype=torch.uint8, device="cuda"),
                    cur=torch.empty(count, dtype=torch.int8, device="cuda"), typ=torch.empty(count, dtype=torch.uint8, device="cuda"),
                    rew=torch.empty((count, 2), dtype=torch.float64, device="cuda"),
                    msk=torch.empty((count, 1), dtype=torch.int32, device="cuda"))

    even, odd = make(n), make(n + 1)        # the odd batch keeps the one-environment kernel; its first n environments are compared
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    acts = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda")
    for t in range(60):
        for e, count in ((even, n), (odd, n + 1)):
            check(lib().osg_env_step(e["b"]._h, acts[:count].contiguous().data_ptr(), e["reset"].data_ptr(), 77, 0, t,
                                     e["cur"].data_ptr(), e["typ"].data_ptr(), e["rew"].data_ptr(), e["msk"].data_ptr()))
        torch.cuda.synchronize()   # (ctx.synchronize() would report the illegal actions of the previous step)
        for k in ("reset", "cur", "typ", "rew", "msk"):
            assert torch.equal(even[k], odd[k][:n]), (game, t, k)
        assert (even["b"].raw_words() == odd["b"].raw_words()[:, :n]).all(), (game, t)
        # next actions: a random legal column, every 37th environment an illegal one (counted, state unchanged)
        m = even["msk"][:, 0]
        cols = even["b"].desc.num_distinct_actions
        bits = ((m.unsqueeze(1) >> torch.arange(cols, device="cuda", dtype=torch.int32)) & 1).to(torch.float32)
        pick = (bits * (torch.rand(bits.shape, device="cuda", generator=gen) + 0.01)).argmax(1).to(torch.int32)
        pick = torch.where(bits.sum(1) > 0, pick, torch.full_like(pick, -1))
        pick[::37] = cols + 3
        acts[:n] = pick
        acts[n] = pick[0]
    with pytest.raises(osa.OsgError, match="illegal"):   # both forms counted their refused actions on the context
        ctx.synchronize()


@pytest.mark.parametrize("game,P", [("connect_four", 2), ("connect_four(rows=5,columns=6,x_in_row=3)", 2), ("leduc_poker", 2),
                                    ("leduc_poker(players=3)", 3), ("hex(board_size=13)", 2)])
def test_env_step_compact_equals_env_step(ctx, game, P):
    """osg_env_step_compact (u8 actions, one in/out flag byte, i8 rewards holding twice the return: 41 instead of 60 bytes per
    connect_four environment) against osg_env_step, side by side over several episodes: the same states, step types, players,
    rewards, masks and restarts step by step — the two-per-thread kernel (even batch) and the generic one (odd batch, wide
    boards, poker with its chance sampling) —, illegal actions refused and counted the same way."""
    import torch
    import open_spiel_amd as osa
    from open_spiel_amd._abi import check, lib
    for n in (1 << 11, (1 << 11) + 1):
        std = osa.StateBatch(ctx, game, n)
        cmp_ = osa.StateBatch(ctx, game, n)
        W = std.desc.mask_words
        reset = torch.ones(n, dtype=torch.uint8, device="cuda")
        cur = torch.empty(n, dtype=torch.int8, device="cuda"); typ = torch.empty(n, dtype=torch.uint8, device="cuda")
        rew = torch.empty((n, P), dtype=torch.float64, device="cuda")
        msk = torch.empty((n, W), dtype=torch.int32, device="cuda")
        flags = torch.full((n,), 2, dtype=torch.uint8, device="cuda")
        rew2 = torch.empty((n, P), dtype=torch.int8, device="cuda")
        msk2 = torch.empty((n, W), dtype=torch.int32, device="cuda")
        gen = torch.Generator(device="cuda"); gen.manual_seed(9)
        acts = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        lasts = 0
        for t in range(190 if game.startswith("hex") else 70):
            a8 = torch.where(acts < 0, torch.full_like(acts, 255), acts).to(torch.uint8)
            check(lib().osg_env_step(std._h, acts.data_ptr(), reset.data_ptr(), 77, 5, t, cur.data_ptr(), typ.data_ptr(),
                                     rew.data_ptr(), msk.data_ptr()))
            check(lib().osg_env_step_compact(cmp_._h, a8.data_ptr(), flags.data_ptr(), 77, 5, t, rew2.data_ptr(), msk2.data_ptr()))
            torch.cuda.synchronize()   # (ctx.synchronize() would report the illegal actions of the previous step)
            assert torch.equal(flags & 3, typ), (game, n, t)
            assert torch.equal((flags >> 2).to(torch.int8) - 4, cur), (game, n, t)
            assert torch.equal(rew2.to(torch.float64) * 0.5, rew), (game, n, t)
            assert torch.equal(msk2, msk), (game, n, t)
            assert torch.equal((flags & 3) == 2, reset.bool()), (game, n, t)
            assert (std.raw_words() == cmp_.raw_words()).all(), (game, n, t)
            lasts += int((typ == 2).sum())
            # next actions: a random legal action, every 41st environment an illegal one, every 29th left as it is
            A = std.desc.num_distinct_actions
            shifts = torch.arange(32, device="cuda", dtype=torch.int32)
            bits = ((msk.unsqueeze(-1) >> shifts) & 1).reshape(n, -1)[:, :A].to(torch.float32)
            pick = (bits * (torch.rand(bits.shape, device="cuda", generator=gen) + 0.01)).argmax(1).to(torch.int32)
            pick = torch.where(bits.sum(1) > 0, pick, torch.full_like(pick, -1))
            if A + 3 < 255:
                pick[::41] = A + 3
            pick[::29] = -1
            acts = pick
        assert lasts > (0 if game.startswith("hex") else n), "the run must cover episode ends and restarts"
        with pytest.raises(osa.OsgError, match="illegal"):
            ctx.synchronize()


def test_env_step_compact_refuses_returns_beyond_a_byte(ctx):
    import torch
    import open_spiel_amd as osa
    from open_spiel_amd._abi import lib
    b = osa.StateBatch(ctx, "leduc_poker(players=6)", 8)       # wins up to 5 x 13 = 65: 130 does not fit
    z = torch.zeros(64, dtype=torch.uint8, device="cuda")
    rc = lib().osg_env_step_compact(b._h, z.data_ptr(), z.data_ptr(), 0, 0, 0, z.data_ptr(), z.data_ptr())
    assert rc != 0 and "signed byte" in lib().osg_last_error().decode()
    b = osa.StateBatch(ctx, "hex(board_size=16)", 8)           # 256 actions: one byte cannot name them beside 0xFF
    rc = lib().osg_env_step_compact(b._h, z.data_ptr(), z.data_ptr(), 0, 0, 0, z.data_ptr(), z.data_ptr())
    assert rc != 0 and "one byte" in lib().osg_last_error().decode()
