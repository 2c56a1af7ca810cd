"""The struct API (open_spiel/spiel.h:235-299, 340-473, 967-971, 1332-1340; tic_tac_toe.cc:178-213, 273-336;
connect_four.cc:224-275, 352-562) through pyspiel on the device: states to structs / JSON with the reference's exact
text (the literals of tic_tac_toe_test.cc and connect_four_test.cc), states FROM structs / JSON / dicts
(osg_batch_set_cells builds the position on the device with the game's own rules), action structs, game parameters
as a struct and as JSON.  The reference's own tic_tac_toe_test.cc / connect_four_test.cc run against the C++ mirror in
tests/test_z6_*; the JSON value and the struct types are checked without a device in tests/test_host_mirror_cpu.py."""
import json
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pyspiel():
    import pyspiel as module
    return module


def test_reference_json_literals(pyspiel):
    ttt = pyspiel.load_game("tic_tac_toe")
    s = ttt.new_initial_state()
    assert s.to_json() == '{"board":[".",".",".",".",".",".",".",".","."],"current_player":"x"}'
    assert s.to_struct().to_json() == s.to_json() and s.to_dict() == json.loads(s.to_json())
    s.apply_action(4)
    assert s.to_observation_struct(0).to_json() == '{"board":[".",".",".",".","x",".",".",".","."],"current_player":"o"}'
    a = s.action_to_struct(0, 5)
    assert a.to_json() == '{"col":2,"row":1}' and (a.row, a.col) == (1, 2) and s.struct_to_actions(a) == [5]
    assert s.validate_action_struct(a).ok() and not s.validate_action_struct(s.action_to_struct(0, 4)).ok()
    assert s.apply_action_struct(a).ok() and str(s) == "...\n.xo\n..."
    c4 = pyspiel.load_game("connect_four")
    s = c4.new_initial_state()
    for col in (3, 4):
        s.apply_action(col)
    empty = '[".",".",".",".",".",".","."]'
    assert s.to_json() == ('{"board":[[".",".",".","x","o",".","."],' + ",".join([empty] * 5) +
                           '],"current_player":"x","is_terminal":false,"winner":""}')
    for col in (3, 4, 3, 4, 3):
        s.apply_action(col)
    d = s.to_dict()
    assert d["current_player"] == "Terminal" and d["is_terminal"] is True and d["winner"] == "x"
    assert [row[3] for row in d["board"]] == ["x", "x", "x", "x", ".", "."]


@pytest.mark.parametrize("game_string", ["tic_tac_toe", "connect_four", "connect_four(rows=5,columns=6,x_in_row=3)",
                                         "connect_four(rows=9,columns=12)"])
def test_states_from_json_equal_the_played_states(pyspiel, game_string):
    """Every position of 40 random games: new_initial_state(state.to_json()) is the same position — string, player to
    move, legal actions, terminal flag, returns, observation tensor — and plays on identically."""
    game = pyspiel.load_game(game_string)
    rng = random.Random(5)
    for _ in range(40):
        state = game.new_initial_state()
        while True:
            twin = game.new_initial_state(state.to_json())
            assert str(twin) == str(state) and twin.current_player() == state.current_player()
            assert twin.is_terminal() == state.is_terminal() and twin.returns() == state.returns()
            assert twin.legal_actions() == state.legal_actions()
            np.testing.assert_array_equal(twin.observation_tensor(0), state.observation_tensor(0))
            assert twin.history() == [] and twin.to_json() == state.to_json()
            by_dict = game.new_initial_state(state.to_dict())
            by_struct = game.new_initial_state(state.to_struct())
            assert str(by_dict) == str(state) == str(by_struct)
            if state.is_terminal():
                break
            a = rng.choice(state.legal_actions())
            state.apply_action(a)
            twin.apply_action(a)
            assert str(twin) == str(state) and twin.returns() == state.returns()


def test_invalid_structs_are_refused(pyspiel):
    ttt = pyspiel.load_game("tic_tac_toe")
    ok = {"board": ["x", "o", ".", ".", ".", ".", ".", ".", "."], "current_player": "x"}
    assert ttt.new_initial_state(ok).current_player() == 0
    for bad in (dict(ok, current_player="o"),                                   # the stone count says x
                dict(ok, board=["x", "x", "x", "o", "o", "o", ".", ".", "."]),      # both players have a line
                dict(ok, board=["o", "o", ".", ".", ".", ".", ".", ".", "."]),      # more o than x
                dict(ok, board=["x", "?", ".", ".", ".", ".", ".", ".", "."]),
                dict(ok, board=["x", "o"])):
        with pytest.raises(Exception):
            ttt.new_initial_state(bad)
    c4 = pyspiel.load_game("connect_four")
    rows = [["."] * 7 for _ in range(6)]
    rows[1][3] = "x"                                                                # floating: a gap below it
    with pytest.raises(Exception, match="gap"):
        c4.new_initial_state({"board": rows, "current_player": "o", "is_terminal": False, "winner": ""})
    rows = [["."] * 7 for _ in range(6)]
    rows[0][3] = "x"
    with pytest.raises(Exception):
        c4.new_initial_state({"board": rows, "current_player": "o", "is_terminal": True, "winner": ""})
    assert c4.new_initial_state({"board": rows, "current_player": "o", "is_terminal": False, "winner": ""}).current_player() == 1
    kuhn = pyspiel.load_game("kuhn_poker")
    with pytest.raises(Exception):
        kuhn.new_initial_state().to_json()                                          # ToStruct is not implemented (spiel.h:465-467)


def test_game_parameters_struct_and_json(pyspiel):
    params = pyspiel.connect_four.ConnectFourGameParams()
    assert (params.game_name, params.rows, params.columns, params.x_in_row, params.egocentric_obs_tensor) == ("connect_four", 6, 7, 4, False)
    params.rows, params.columns, params.x_in_row = 8, 9, 5
    game = pyspiel.load_game(params)
    assert game.observation_tensor_shape() == [3, 8, 9] and game.get_parameters()["x_in_row"] == 5
    assert json.loads(params.to_json()) == {"game_name": "connect_four", "rows": 8, "columns": 9, "x_in_row": 5,
                                            "egocentric_obs_tensor": False}
    small = pyspiel.load_game_from_json('{"game_name":"connect_four","rows":4,"columns":5,"x_in_row":3}')
    assert small.observation_tensor_shape() == [3, 4, 5] and small.max_game_length() == 20
    again = pyspiel.connect_four.ConnectFourGameParams(params.to_json())
    assert again.columns == 9 and pyspiel.connect_four.ConnectFourGameParams(json.loads(params.to_json())).rows == 8
    hexg = pyspiel.load_game_from_json('{"game_name":"hex","board_size":5,"swap":true}')
    assert hexg.num_distinct_actions() == 26 and "swap=True" in str(hexg)          # 25 cells + the swap move


def test_set_cells_touches_one_state_of_a_batch():
    """osg_batch_set_cells on a batch: only the addressed state changes, in every connect_four layout (the 6 x 7 board with
    its result byte, a generic board, a board above 64 bits) and tic_tac_toe; the refusals come back as errors."""
    import torch
    import open_spiel_amd as osa
    ctx = osa.Context(0)
    for game, cells, player, terminal in (
            ("tic_tac_toe", "xo.......", 0, False),
            ("tic_tac_toe", "xxxoo....", -4, True),
            ("connect_four", "xo....." + "." * 35, 0, False),
            ("connect_four", "xxxx..." + "ooo...." + "." * 28, -4, True),
            ("connect_four(rows=4,columns=5,x_in_row=3)", "xo..." + "." * 15, 0, False),
            ("connect_four(rows=9,columns=12)", "x" + "." * 107, 1, False)):
        b = osa.StateBatch(ctx, game, 8)
        b.random_steps(11, 3)
        before = b.raw_words().copy()
        b.set_cells(5, cells)
        after = b.raw_words()
        others = [i for i in range(8) if i != 5]
        assert (after[:, others] == before[:, others]).all(), game
        cur, term, _ = b.status()
        assert int(cur[5]) == player and bool(term[5]) == terminal, (game, cells)
        fresh = osa.StateBatch(ctx, game, 1)
        fresh.set_cells(0, cells)
        assert (fresh.raw_words()[:, 0] == after[:, 5]).all()
    b = osa.StateBatch(ctx, "connect_four", 2)
    for bad in ("." * 7 + "x" + "." * 34, "x" * 41, "q" + "." * 41, "xxxx..." + "oooo..." + "." * 28):
        with pytest.raises(osa.OsgError):
            b.set_cells(1, bad)
    with pytest.raises(osa.OsgError):
        osa.StateBatch(ctx, "kuhn_poker", 2).set_cells(0, "..")


@pytest.mark.parametrize("game_string,opening", [("tic_tac_toe", (4, 0, 8)), ("connect_four", (3, 3, 4, 2, 4)),
                                                 ("connect_four(rows=5,columns=6,x_in_row=3)", (0, 1, 1))])
def test_struct_built_states_keep_their_starting_position(pyspiel, game_string, opening):
    """A state built from a struct / JSON starts at that position (tic_tac_toe.cc:336, connect_four.cc:512:
    starting_state_str_ = ToJson()): State::Serialize emits it as the "starting_state=" first line and
    Game::DeserializeState parses it (spiel.cc:411-430, 540-560), so serialize -> deserialize, pickle, copy.deepcopy and
    ApplyAction + UndoAction all come back to the same position instead of replaying onto an empty board."""
    import copy
    import pickle
    game = pyspiel.load_game(game_string)
    played = game.new_initial_state()
    for a in opening:
        played.apply_action(a)
    assert played.starting_state_str() == "" and played.starting_state() is None
    assert played.serialize() == "".join(f"{a}\n" for a in opening)
    built = game.new_initial_state(played.to_json())
    assert built.starting_state_str() == played.to_json() and built.history() == []
    assert built.serialize() == "starting_state=" + played.to_json() + "\n\n"
    assert str(built.starting_state()) == str(played)

    def same(a, b):
        assert str(a) == str(b) and a.current_player() == b.current_player() and a.legal_actions() == b.legal_actions()
        assert a.is_terminal() == b.is_terminal() and a.returns() == b.returns()
        assert a.observation_tensor(0) == b.observation_tensor(0)

    moves = []
    for _ in range(2):
        a = built.legal_actions()[-1]
        moves.append((built.current_player(), a))
        built.apply_action(a)
        played.apply_action(a)
    same(built, played)
    text = built.serialize()
    assert text.splitlines()[0].startswith("starting_state={") and text.splitlines()[1:] == [str(a) for _, a in moves]
    for again in (game.deserialize_state(text), pickle.loads(pickle.dumps(built)), copy.deepcopy(built),
                  pyspiel.deserialize_game_and_state(pyspiel.serialize_game_and_state(game, built))[1], built.clone()):
        same(again, played)
        assert again.history() == [a for _, a in moves] and again.starting_state_str() == built.starting_state_str()
        assert again.serialize() == text
    # undo back to the starting position and replay
    for player, a in reversed(moves):
        built.undo_action(player, a)
    same(built, built.starting_state())
    assert built.history() == [] and str(built) != str(game.new_initial_state())
    for _, a in moves:
        built.apply_action(a)
    same(built, played)
