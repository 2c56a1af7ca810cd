"""Pin the CPU oracle against the reference's playthrough goldens.

The fixtures in tests/golden/playthroughs.json were extracted by
tests/golden/make_golden.py from the reference's
integration_tests/playthroughs/*.txt (the files playthrough_test.py:74-95
regenerates byte-exactly).  Replay needs no RNG: each block carries the action
that leads to the next one.
"""
import numpy as np
import pytest

FILES = [
    "tic_tac_toe.txt", "connect_four.txt", "hex(board_size=5).txt", "kuhn_poker_2p.txt",
    "kuhn_poker_3p.txt", "leduc_poker_773740114.txt", "leduc_poker_1540482260.txt",
    "leduc_poker_3977671846.txt", "leduc_poker_3p.txt",
]


def _tensor(gold):
    if isinstance(gold, str):
        return np.array([float(c) for c in gold], np.float32)
    return np.array(gold, np.float32)


def _fmt6(x):
    return "{:.6g}".format(x)


@pytest.mark.parametrize("fname", FILES)
def test_playthrough(oracle, goldens, fname):
    rec = goldens[fname]
    game = oracle.Game(rec["game"])
    hdr = rec["header"]
    assert game.num_distinct_actions == int(hdr["NumDistinctActions"])
    assert game.max_chance_outcomes == int(hdr["MaxChanceOutcomes"])
    assert game.num_players == int(hdr["NumPlayers"])
    assert game.min_utility == float(hdr["MinUtility"])
    assert game.max_utility == float(hdr["MaxUtility"])
    assert game.max_game_length == int(hdr["MaxGameLength"])
    assert game.observation_tensor_size == int(hdr["ObservationTensorSize"])
    if "InformationStateTensorSize" in hdr:
        assert game.information_state_tensor_size == int(hdr["InformationStateTensorSize"])
    # The playthrough prints str(game) with every parameter incl. defaults.
    assert str(game) == hdr["ToString"].strip('"')
    assert game.parameters_string() == hdr["GetParameters"]

    state = game.new_initial_state()
    checked = 0
    for blk in rec["states"]:
        if not blk.get("skipped"):
            checked += 1
            assert state.is_terminal() == blk["is_terminal"]
            assert state.history() == blk["history"]
            assert state.current_player() == blk["current_player"]
            # the dump strips trailing blanks per line (exact strings are still
            # pinned through ObservationString below)
            assert ([l.rstrip() for l in str(state).rstrip("\n").split("\n")] ==
                    [l.rstrip() for l in blk["to_string"].rstrip("\n").split("\n")])
            assert state.legal_actions() == blk["legal_actions"]
            if "returns" in blk:
                # hex prints [0, -0] for a running game: compare values.
                assert state.returns() == blk["returns"]
            cp = state.current_player()
            if blk["legal_actions"]:
                names = [state.action_to_string(cp, a) for a in blk["legal_actions"]]
                assert names == blk["string_legal_actions"]
            # every non-acting player has no legal actions (basic_tests.cc:90-112)
            for p in range(game.num_players):
                if p != cp:
                    assert state.legal_actions(p) == []
            if "chance_outcomes" in blk:
                got = state.chance_outcomes()
                assert [a for a, _ in got] == [a for a, _ in blk["chance_outcomes"]]
                for (_, pg), (_, pe) in zip(got, blk["chance_outcomes"]):
                    assert _fmt6(pg) == _fmt6(pe)
                assert abs(sum(p for _, p in got) - 1.0) < 1e-12
            for p_str, s in blk["info_str"].items():
                assert state.information_state_string(int(p_str)) == s
            for p_str, s in blk["obs_str"].items():
                assert state.observation_string(int(p_str)) == s
            for key, gold in blk["tensors"].items():
                p = int(key[-1]) if key[-1].isdigit() else 0
                want = _tensor(gold)
                got = (state.observation_tensor(p) if key.startswith("obs")
                       else state.information_state_tensor(p))
                np.testing.assert_array_equal(got, want, err_msg=f"{fname} {key} h={blk['history']}")
        if "action" in blk:
            state.apply_action(blk["action"])
    assert checked >= 3
    assert state.is_terminal()
