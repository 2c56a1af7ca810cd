"""The CPU side of the synthetic benchmark inputs (osgo_synth_batch in oracle/spiel_oracle_capi.cpp):
the restatement and the genuine reference build draw the same batch, the result does not depend on the number of
worker threads or on how an index range is cut into shards, and a pure-Python walk of the recipe over the oracle's
State API reproduces single states."""
import os

import numpy as np
import pytest

GAMES = [("connect_four", 36), ("hex(board_size=9)", 40), ("tic_tac_toe", 5), ("kuhn_poker", 2), ("leduc_poker", 3),
         ("leduc_poker(players=3)", 4), ("hex(board_size=4,swap=True)", 9)]
M64 = (1 << 64) - 1


def _mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


class CounterRng:
    """open_spiel_amd/csrc/osg_common.h Rng, restated."""

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


def _draw(s, rng):
    if s.is_chance_node():
        z, acc = rng.unit(), 0.0
        outcomes = s.chance_outcomes()
        for a, pr in outcomes:
            if acc <= z < acc + pr:
                return a
            acc += pr
        return outcomes[-1][0]
    legal = s.legal_actions()
    return legal[rng.below(len(legal))]


@pytest.mark.parametrize("game,depth_mod", GAMES)
def test_restatement_and_reference_draw_the_same_batch(oracle, reference, game, depth_mod):
    a = oracle.Game(game).synth_batch(0x5EED, 3000, depth_mod, first=11, threads=3)
    b = reference.Game(game).synth_batch(0x5EED, 3000, depth_mod, first=11, threads=2)
    for k, v in a.items():
        if v is not None:
            np.testing.assert_array_equal(v, b[k], err_msg=k)
    assert a["term0"].sum() == 0
    assert set(np.unique(a["depth"])) == set(range(depth_mod))


def test_threads_and_shards_do_not_change_the_batch(oracle):
    g = oracle.Game("connect_four")
    whole = g.synth_batch(7, 2048, 36, threads=1)
    again = g.synth_batch(7, 2048, 36, threads=5)
    tail = g.synth_batch(7, 1024, 36, first=1024, threads=2)
    for k, v in whole.items():
        np.testing.assert_array_equal(v, again[k], err_msg=k)
        np.testing.assert_array_equal(v[1024:], tail[k], err_msg=k)


@pytest.mark.parametrize("game,depth_mod", [("connect_four", 36), ("leduc_poker", 3)])
def test_python_walk_of_the_recipe(oracle, game, depth_mod):
    og = oracle.Game(game)
    rec = og.synth_batch(0x5EED, 64, depth_mod, first=1000)
    for i in range(0, 64, 7):
        rng = CounterRng(0x5EED, 1000 + i, 0x53594E5448)
        depth = rng.below(depth_mod)
        while True:
            s = og.new_initial_state()
            for _ in range(depth):
                if s.is_terminal():
                    break
                s.apply_action(_draw(s, rng))
            if not s.is_terminal():
                break
        action = _draw(s, rng)
        assert depth == rec["depth"][i] and action == rec["action"][i], i
        assert s.current_player() == rec["cur0"][i]
        np.testing.assert_array_equal(np.asarray(s.observation_tensor(0)).astype(np.uint8), rec["obs0"][i])
        s.apply_action(action)
        assert s.is_terminal() == bool(rec["term1"][i]) and s.returns() == rec["rets1"][i].tolist()


def test_synth_mcts_replay_equals_single_root_replay(oracle):
    """osgo_synth_mcts_replay = osgo_synth_batch's root + the replay-mode search the GPU tests already use root by root."""
    og = oracle.Game("hex(board_size=5)")
    out = og.synth_mcts_replay(5, 12, 10, 2.0, 40, 1, 99, 2, first=3, threads=3)
    for i in (0, 5, 11):
        rng = CounterRng(5, 3 + i, 0x53594E5448)
        depth = rng.below(10)
        while True:
            s = og.new_initial_state()
            for _ in range(depth):
                if s.is_terminal():
                    break
                s.apply_action(_draw(s, rng))
            if not s.is_terminal():
                break
        want = s.mcts_search(2.0, 40, 1, 4096, False, 0, counter_root=3 + i, counter_seed=99, counter_layout=2)
        assert out["best_action"][i] == want["best_action"]
        for act, cnt, tot, _ in want["children"]:
            assert out["child_visits"][i, int(act)] == cnt and out["child_reward"][i, int(act)] == tot


def test_sub_stream_jumps_never_meet():
    """Rng::jump_to (csrc/osg_common.h; restated in oracle/spiel_oracle_capi.cpp): sub-stream id starts at s0 + id * kJump
    and advances by kStep per draw, so two sub-streams of one trajectory would draw the same counter only after
    (id1 - id2) * kJump / kStep steps modulo 2^64: astronomically many for every pair of ids a traversal opens."""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "open_spiel_amd", "csrc", "osg_common.h")).read()
    k_jump = int(re.search(r"kJump = (0x[0-9A-Fa-f]+)ULL", src).group(1), 16)
    k_step = int(re.search(r"s \+= (0x[0-9A-Fa-f]+)ULL;", src).group(1), 16)
    m = 1 << 64
    inv = pow(k_step, -1, m)
    for d in range(1, 129):
        steps = (d * k_jump * inv) % m
        assert min(steps, m - steps) > 1 << 56, d
