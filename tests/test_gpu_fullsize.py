"""Parity at BASELINE.json's full sizes (2^20 states, 2^16 roots) through properties that
do not depend on the size, plus an oracle replay of a strided sample of the batch.

The small-size tests compare every state with the oracle; here the batch is too big for
that, so the checks are: invariants every reachable state satisfies, conservation sums
("checksum of checksums"), consistency between the fused kernel and the individual
queries on the same batch, and exact replay of ~250 sampled trajectories in the oracle.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M64 = (1 << 64) - 1


def _mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


class CounterRng:
    """The device's counter RNG (open_spiel_amd/csrc/osg_common.h), restated for replay."""

    def __init__(self, seed, stream, sub=0):
        a = _mix64((seed + 0x9E3779B97F4A7C15) & M64)
        b = _mix64(a ^ ((stream * 0xD1342543DE82EF95 + 0x632BE59BD9B4E019) & M64))
        self.s = _mix64(b ^ ((sub * 0xA0761D6478BD642F + 0xE7037ED1A0B428DB) & M64))

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & M64
        return _mix64(self.s)

    def below(self, n):
        return ((self.next() >> 32) * n) >> 32

    def unit(self):
        return (self.next() >> 11) * (1.0 / 9007199254740992.0)


@pytest.fixture(scope="module")
def ctx():
    import open_spiel_amd as osa
    return osa.Context(0)


def _popcount64(a):
    a = a.astype(np.uint64)
    c = np.zeros(a.shape, np.int64)
    for _ in range(64):
        c += (a & np.uint64(1)).astype(np.int64)
        a = a >> np.uint64(1)
    return c


def _replay_random_steps(oracle, game, seed, index, steps):
    """Oracle replay of osg_random_steps for one state index (auto-reset included)."""
    og = oracle.Game(game)
    s = og.new_initial_state()
    rng = CounterRng(seed, index, 0)
    episodes = 0
    for _ in range(steps):
        if s.is_terminal():
            s = og.new_initial_state()
            episodes += 1
        if s.is_chance_node():
            z = rng.unit()
            acc, pick = 0.0, None
            outcomes = s.chance_outcomes()
            for a, pr in outcomes:
                if acc <= z < acc + pr:
                    pick = a
                    break
                acc += pr
            s.apply_action(outcomes[-1][0] if pick is None else pick)
        else:
            legal = s.legal_actions()
            s.apply_action(legal[rng.below(len(legal))])
    return s, episodes


def test_connect_four_2pow20_states(oracle, ctx):
    import torch
    import open_spiel_amd as osa
    n, seed, steps = 1 << 20, 0x5EED, 29
    b = osa.StateBatch(ctx, "connect_four", n)
    counters = b.random_steps(seed, steps)
    ctx.synchronize()
    assert counters.tolist()[0] == n * steps
    words = b.raw_words()
    x, o = words[0] & np.uint64((1 << 56) - 1), words[1]
    flags = (words[0] >> np.uint64(56)).astype(np.int64)
    board = np.uint64(sum(((1 << 6) - 1) << (7 * c) for c in range(7)))
    # structural invariants of every reachable position
    assert not (x & o).any(), "a cell cannot hold both colours"
    assert not ((x | o) & ~board).any(), "stones only on playable cells"
    px, po = _popcount64(x), _popcount64(o)
    assert ((px == po) | (px == po + 1)).all(), "players alternate, x starts"
    col = (x | o)
    for c in range(7):  # gravity: every column is filled from the bottom without holes
        colbits = ((col >> np.uint64(7 * c)) & np.uint64(0x3F)).astype(np.int64)
        assert ((colbits & (colbits + 1)) == 0).all()
    # individual queries on the same batch
    cur, term, rets = [t.cpu().numpy() for t in b.status()]
    assert ((flags & 1) == term).all()
    assert (cur[term == 0] == ((px + po)[term == 0] & 1)).all()
    assert (cur[term == 1] == -4).all()
    assert (rets.sum(1) == 0).all() and (np.abs(rets) <= 1).all(), "zero-sum, utilities in [-1, 1]"
    assert (rets[term == 0] == 0).all()
    mask = b.legal_actions_mask().cpu().numpy()
    assert (mask[term == 1] == 0).all(), "no legal actions at terminal states"
    assert (mask[term == 0].sum(1) >= 1).all()
    top_free = np.stack([((col >> np.uint64(7 * c + 5)) & np.uint64(1)) == 0 for c in range(7)], 1)
    assert (mask[term == 0] == top_free[term == 0]).all(), "legal = columns whose top cell is empty"
    # the fused kernel agrees with the individual calls, at full size, out of place
    dst = osa.StateBatch(ctx, "connect_four", n)
    first_legal = torch.from_numpy(np.where(mask.any(1), mask.argmax(1), 255).astype(np.uint8)).cuda()
    m8, st = b.step(first_legal, dst=dst)
    st = st.cpu().numpy()
    assert ((st & 0x40) == 0).all()
    ref = b.clone()
    ref.apply_actions(torch.from_numpy(np.where(mask.any(1), mask.argmax(1), -1).astype(np.int32)))
    np.testing.assert_array_equal(dst.raw_words(), ref.raw_words())
    cur2, term2, _ = [t.cpu().numpy() for t in ref.status(False)[:2]] + [None]
    assert (((st & 0x80) != 0) == (term2 != 0)).all()
    live = term2 == 0
    assert (((st[live] & 15).astype(np.int64) - 1) == cur2[live]).all()
    bits = ref.legal_actions_mask_bits().cpu().numpy().view(np.uint32)[:, 0]
    assert (m8.cpu().numpy().reshape(n) == (bits & 0xFF)).all()
    # oracle replay of a strided sample of the 2^20 trajectories
    og_cols = 7
    for i in range(0, n, n // 256 + 1):
        s, _ = _replay_random_steps(oracle, "connect_four", seed, i, steps)
        want_mask = np.zeros(og_cols, np.uint8)
        for a in s.legal_actions():
            want_mask[a] = 1
        assert term[i] == s.is_terminal(), i
        assert cur[i] == s.current_player(), i
        assert rets[i].tolist() == s.returns(), i
        assert (mask[i] == want_mask).all(), i


def test_observation_tensor_2pow20_states(ctx):
    import open_spiel_amd as osa
    n = 1 << 20
    b = osa.StateBatch(ctx, "connect_four", n)
    b.random_steps(11, 17)
    obs = b.observation_tensor(0)
    assert obs.shape == (n, 126)
    planes = obs.view(n, 3, 42)
    assert bool((planes.sum(1) == 1).all()), "every cell is exactly one of x / o / empty"
    words = b.raw_words()
    x = words[0] & np.uint64((1 << 56) - 1)
    assert (planes[:, 0].sum(1).cpu().numpy() == _popcount64(x)).all()
    assert (planes[:, 1].sum(1).cpu().numpy() == _popcount64(words[1])).all()
    import torch
    assert float(obs.sum(dtype=torch.float64)) == n * 42


def test_hex9_observation_tensor_2pow18_states(oracle, ctx):
    """2^18 hex(9) states, [n, 9, 81]: every cell lies on exactly one plane, the empty plane is the
    complement of the stones, black / white planes count the stones, and a strided sample equals the
    oracle's ObservationTensor of the replayed trajectory."""
    import torch
    import open_spiel_amd as osa
    n, seed, steps = 1 << 18, 31, 33
    b = osa.StateBatch(ctx, "hex(board_size=9)", n)
    b.random_steps(seed, steps)
    obs = b.observation_tensor(0)
    assert obs.shape == (n, 729)
    planes = obs.view(n, 9, 81)
    assert bool((planes.sum(1) == 1).all()), "every cell is on exactly one plane"
    assert float(obs.sum(dtype=torch.float64)) == n * 81
    _, term, _ = [t.cpu().numpy() for t in b.status()]
    stones = (81 - planes[:, 4].sum(1)).cpu().numpy()  # plane 4 = empty (hex.cc:392-396)
    white = planes[:, :4].sum((1, 2)).cpu().numpy()
    black = planes[:, 5:].sum((1, 2)).cpu().numpy()
    assert (white + black == stones).all()
    assert ((black == white) | (black == white + 1)).all(), "black moves first, players alternate"
    win = (planes[:, 0].sum(1) + planes[:, 8].sum(1)).cpu().numpy() > 0  # a Win-labelled stone
    assert (win == (term != 0)).all(), "terminal <=> some stone carries a Win label"
    got = obs.cpu().numpy()
    for i in range(0, n, n // 64 + 1):
        s, _ = _replay_random_steps(oracle, "hex(board_size=9)", seed, i, steps)
        np.testing.assert_array_equal(got[i], np.asarray(s.observation_tensor(0), np.float32), err_msg=str(i))


@pytest.mark.parametrize("game,steps", [("hex(board_size=9)", 40), ("leduc_poker", 7), ("tic_tac_toe", 6),
                                        ("kuhn_poker", 4)])
def test_other_games_2pow20_random_steps_replay(oracle, ctx, game, steps):
    """2^20 states of every other game: counters, zero-sum checksum, sampled oracle replay."""
    import open_spiel_amd as osa
    n, seed = 1 << 20, 77
    b = osa.StateBatch(ctx, game, n)
    counters = b.random_steps(seed, steps)
    ctx.synchronize()
    assert counters.tolist()[0] == n * steps
    cur, term, rets = [t.cpu().numpy() for t in b.status()]
    assert abs(rets.sum()) < 1e-9, "zero-sum games: the returns of the whole batch sum to 0"
    assert (rets[term == 0] == 0).all()
    bits = b.legal_actions_mask_bits().cpu().numpy().view(np.uint32)
    assert (bits[term == 1] == 0).all() and (bits[term == 0].any(1)).all()
    for i in range(0, n, n // 128 + 1):
        s, _ = _replay_random_steps(oracle, game, seed, i, steps)
        assert term[i] == s.is_terminal(), (game, i)
        assert cur[i] == s.current_player(), (game, i)
        assert rets[i].tolist() == s.returns(), (game, i)
        want = np.zeros(bits.shape[1], np.uint32)
        for a in s.legal_actions():
            want[a // 32] |= np.uint32(1 << (a % 32))
        assert (bits[i] == want).all(), (game, i)


def test_hex9_2pow16_roots_rollouts(ctx):
    """Config 4's root count: 2^16 hex(9) roots, 4 rollouts each: every playout ends, hex has no
    draws (|sum of returns| == n_rollouts is impossible otherwise), batch sums are antisymmetric."""
    import open_spiel_amd as osa
    n, n_rollouts = 1 << 16, 4
    roots = osa.StateBatch(ctx, "hex(board_size=9)", n)
    roots.random_steps(5, 12)
    total, steps = roots.rollout(123, n_rollouts, want_steps=True)
    total, steps = total.cpu().numpy(), steps.cpu().numpy()
    assert (total[:, 0] == -total[:, 1]).all()
    assert (np.abs(total[:, 0]) % 2 == n_rollouts % 2).all(), "no draws in hex"
    assert (steps >= n_rollouts * 1).all() and (steps <= n_rollouts * 81).all()


def test_bench_two_ranks_code_path(tmp_path):
    """`python bench.py --gpus 2` — no torchrun on the command line: bench.py starts its two ranks itself — end to
    end (sharded roots / trajectories, delta all-reduce, max-over-ranks timing, per-rank rates, the strong-scaling
    efficiency against the same run's one-rank search, the roofline legs and the overlapped ES-MCCFR schedule), with
    the two ranks sharing this box's GPU over gloo — the RCCL run itself needs two GPUs
    (tests/test_z7_gpu_exchange_steps.py::test_bench_gpus_2_over_rccl)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OSG_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-pmc", "--states", str(1 << 16)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    sys.path.insert(0, root)
    import bench
    line, full = bench.read_lines(r.stdout)          # the < 4 KB line the driver keeps, and the full record printed before it
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["collective_backend"] == "gloo" and line["rccl_world"] is None
    assert len(line["per_rank"]["env_steps_per_s"]) == 2 and min(line["per_rank"]["env_steps_per_s"]) > 0
    assert len(line["per_rank"]["avg_launch_us"]) == 2
    assert line["roofline"]["copy_gbs"] > 0 and line["persistent"]["value"] > 0
    sec = line["secondary"]
    assert "error" not in sec and "secondary_truncated" not in line, sec
    assert sec["mcts"]["value"] > 0 and len(sec["mcts"]["per_rank_sims_per_s"]) == 2
    assert sec["mcts"]["single_rank_all_roots"] > 0 and sec["mcts"]["strong_scaling_efficiency"] > 0
    assert sec["mccfr"]["tables_finite"] and sec["mccfr"]["allreduce_us"]["gloo"] > 0 and sec["mccfr"]["allreduce_bytes"] == 44928
    # the one-shot all-reduce carries the same exchange step (two ranks on this one device)
    # (two bench processes time-share this ONE device: a spinning one-shot kernel can wait a scheduling quantum for
    # its peer's queue — 5.5 us in tests/test_z10_gpu_oneshot_allreduce.py, up to ~16 ms here; one rank per GPU has no such wait)
    assert "oneshot_error" not in sec["mccfr"], sec["mccfr"]
    assert 0 < sec["mccfr"]["allreduce_us"]["oneshot"] < 2e5 and sec["mccfr"]["oneshot_trajectories_per_s"] > 0
    assert line["parity_checked_states"] == 1 << 16 and line["parity_against"] in ("reference", "port")
    assert sec["ttt_mcts"]["value"] > 0
    fsec = full["secondary"]
    assert "copy_ceiling" in full["roofline"] and fsec["mccfr"]["oneshot"]["nash_conv_after"] < 4.7
    q = fsec["mccfr"]["quality"]
    assert q["world"] == 2 and q["nash_conv"] < 4.7 and q["overlapped"]["nash_conv"] < 4.7
    with open(os.path.join(root, line["detail"])) as f:
        assert json.load(f)["value"] == full["value"]


def test_hex9_mcts_2pow16_roots_under_full_load(oracle, ctx):
    """Config 4's root count with the chip saturated (64 wavefronts queued per SIMD): the wave-per-root
    kernel relies on one wavefront's memory operations being issued in order (no waits between a store
    and a later load of the same node).  Any violation would show as run-to-run differences or broken
    tree invariants; a strided sample of roots is also replayed exactly by the oracle's MCTSBot."""
    import torch
    import open_spiel_amd as osa
    n, sims, seed = 1 << 16, 128, 0xC0FFEE
    og = oracle.Game("hex(board_size=9)")
    roots = osa.StateBatch(ctx, "hex(board_size=9)", n)
    plies = 20
    roots.random_steps(77, plies)
    keep = (roots.is_terminal() == 0).cpu().numpy()
    a = roots.mcts_search(uct_c=2.0, max_simulations=sims, seed=seed, layout=2)
    b = roots.mcts_search(uct_c=2.0, max_simulations=sims, seed=seed, layout=2)
    for key in ("best_action", "child_visits", "child_reward"):
        assert torch.equal(a[key], b[key]), f"{key}: two runs of the same search differ"
    visits = a["child_visits"].cpu().numpy()
    reward = a["child_reward"].cpu().numpy()
    stats = a["root_stats"].cpu().numpy()
    live = keep & (stats[:, 3] == sims)
    assert live.sum() > n * 0.9
    assert (visits[live].sum(1) == sims - 1).all(), "every simulation but the first (it evaluates the root) visits one root child"
    assert (np.abs(reward[live]) <= visits[live]).all(), "|total reward| <= visits (returns are +-1)"
    assert (stats[live, 0] == sims).all()
    mask = roots.legal_actions_mask().cpu().numpy()
    assert (visits[live][mask[live] == 0] == 0).all(), "no visits on occupied cells"
    best = a["best_action"].cpu().numpy()
    assert (mask[live, :][np.arange(live.sum()), best[live]] == 1).all()
    # exact oracle replay of a strided sample (the root position is rebuilt from the random_steps stream)
    for i in range(0, n, n // 24 + 1):
        if not live[i]:
            continue
        s, _ = _replay_random_steps(oracle, "hex(board_size=9)", 77, i, plies)
        want = s.mcts_search(2.0, sims, 1, 4096, False, 0, counter_root=i, counter_seed=seed, counter_layout=2)
        for act, cnt, tot, _ in want["children"]:
            assert visits[i, int(act)] == cnt and reward[i, int(act)] == tot, (i, int(act))
        assert best[i] == want["best_action"], i
    del og


def test_rollout_sums_do_not_depend_on_how_rollouts_are_shared_out(ctx):
    """Rollout r of root i always plays from stream (seed, i, r): the sums of a root inside a 2^17-root
    batch (4 lanes per root, 4 rollouts each) equal those of the same root evaluated alone (16 lanes)."""
    import torch
    import open_spiel_amd as osa
    n, r, seed = 1 << 17, 16, 4321
    big = osa.StateBatch(ctx, "connect_four", n)
    big.random_steps(5, 9)
    total, steps = big.rollout(seed, r, want_steps=True)
    total, steps = total.cpu().numpy(), steps.cpu().numpy()
    assert (np.abs(total) <= r).all() and (total[:, 0] == -total[:, 1]).all()
    for i in (0, 1, 777, 65535, 65536, n - 1):
        one = big.gather(torch.tensor([i]))
        t1, s1 = one.rollout(seed, r, index_offset=i, want_steps=True)
        assert t1.cpu().numpy()[0].tolist() == total[i].tolist(), i
        assert int(s1.cpu().numpy()[0]) == int(steps[i]), i


@pytest.mark.parametrize("game,depth", [("leduc_poker", 5), ("leduc_poker(players=3)", 6), ("kuhn_poker", 3),
                                        ("tic_tac_toe", 4), ("connect_four(rows=5,columns=6,x_in_row=3)", 12)])
def test_fused_step_kernels_agree_at_2pow22_states(ctx, game, depth):
    """Which fused-step kernel a batch takes depends on its size and shape (several states per thread with 16-byte
    plane accesses for even batches — leduc_poker only from 2^22 states on — one state per thread otherwise).  The
    same 2^22 positions stepped as an even batch and as the first 2^22 states of an odd one must give the same
    successor records, masks and status bytes, bit for bit (the kernels themselves are checked against the oracle
    at small sizes in test_gpu_parity.py)."""
    import torch
    import open_spiel_amd as osa
    n = 1 << 22
    odd = osa.StateBatch(ctx, game, n + 1)
    odd.random_steps(11, depth)
    even = odd.gather(torch.arange(n, device="cuda"))
    lm = odd.legal_actions_mask()
    acts = torch.where(lm.any(1), (lm.to(torch.float32) * torch.rand(lm.shape, device="cuda")).argmax(1),
                       torch.full((n + 1,), 255, device="cuda")).to(torch.uint8)
    acts[::97] = 250  # some illegal ones
    m_odd, s_odd = odd.step(acts)
    m_even, s_even = even.step(acts[:n].contiguous())
    assert torch.equal(m_even, m_odd[:n])
    assert torch.equal(s_even, s_odd[:n])
    w_odd, w_even = odd.raw_words(), even.raw_words()
    assert (w_even == w_odd[:, :n]).all()
    assert int(((s_even & 0x40) != 0).sum()) >= n // 97  # the illegal ones were refused


def test_connect_four_tensor_kernels_agree_at_2pow22_states(ctx):
    """From 2^22 states on the connect_four tensor kernel writes with non-temporal stores (same arithmetic, another
    store instruction): every fourth row of a 2^22-state tensor must equal the tensor of those 2^20 states packed
    by the small-batch instantiation."""
    import torch
    import open_spiel_amd as osa
    n = 1 << 22
    big = osa.StateBatch(ctx, "connect_four", n)
    big.random_steps(5, 17)
    idx = torch.arange(0, n, 4, device="cuda")
    small = big.gather(idx)
    for player in (0, 1):
        t_big = big.observation_tensor(player)
        t_small = small.observation_tensor(player)
        assert torch.equal(t_big[idx], t_small)
        assert float(t_big.sum()) == float(n * 42)  # every cell is in exactly one of the three planes
        del t_big, t_small
