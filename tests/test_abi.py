"""CPU-side checks of the C-ABI library: it loads, exports every symbol
include/osg_abi.h declares, describes games without a device, and refuses to run
without one (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    import open_spiel_amd
    return open_spiel_amd


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "osg_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(osg_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    from open_spiel_amd import _abi
    declared = _declared_symbols()
    assert len(declared) >= 30
    assert sorted(_abi.SIGNATURES) == declared, "include/osg_abi.h and _abi.SIGNATURES disagree"
    handle = C.CDLL(_abi.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), f"libosg_hip.so does not export {name}"
    built.lib()


@pytest.mark.parametrize("game,A,C_,P,obs,info,L", [
    ("tic_tac_toe", 9, 0, 2, 27, 0, 9),
    ("connect_four", 7, 0, 2, 126, 0, 42),
    ("hex(board_size=9)", 81, 0, 2, 729, 0, 81),
    ("hex", 121, 0, 2, 1089, 0, 121),
    ("hex(board_size=5,swap=True)", 26, 0, 2, 225, 0, 25),
    # one board per Bits width NW = 1 .. 4 (cells <= 32, 64, 96, 128): parse_game fills only the variant that holds it
    ("hex(board_size=4)", 16, 0, 2, 144, 0, 16),
    ("hex(board_size=6)", 36, 0, 2, 324, 0, 36),
    ("hex(num_cols=9,num_rows=10)", 90, 0, 2, 810, 0, 90),
    ("hex(num_cols=11,num_rows=10)", 110, 0, 2, 990, 0, 110),
    ("kuhn_poker", 2, 3, 2, 7, 11, 3),
    ("kuhn_poker(players=3)", 2, 4, 3, 10, 17, 5),
    ("leduc_poker", 3, 6, 2, 16, 30, 8),
    ("leduc_poker(players=3)", 3, 8, 3, 22, 47, 14),
])
def test_game_descriptions_match_the_oracle(built, oracle, game, A, C_, P, obs, info, L):
    d = built.describe(game)
    og = oracle.Game(game)
    assert (d.num_distinct_actions, d.max_chance_outcomes, d.num_players) == (A, C_, P)
    assert (d.obs_size, d.info_size, d.max_game_length) == (obs, info, L)
    assert d.num_distinct_actions == og.num_distinct_actions
    assert d.max_chance_outcomes == og.max_chance_outcomes
    assert d.obs_size == og.observation_tensor_size
    assert d.info_size == og.information_state_tensor_size
    assert d.max_game_length == og.max_game_length
    assert d.max_chance_nodes == og.max_chance_nodes_in_history
    assert (d.min_utility, d.max_utility) == (og.min_utility, og.max_utility)
    assert [d.obs_shape[i] for i in range(d.obs_rank)] == og.observation_tensor_shape()
    assert d.canonical.decode() == str(og)


def test_bad_game_strings_are_rejected(built):
    for bad in ["chess", "connect_four(rows=12,columns=12)", "hex(board_size=20)", "connect_four(foo=1)",
                "hex(swap=3)", "kuhn_poker(players=1)", "leduc_poker(players=11)", "hex(board_size=3"]:
        with pytest.raises(built.OsgError):
            built.describe(bad)


def test_no_cpu_fallback(built):
    """Without a GPU the engine must fail loudly, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from open_spiel_amd import _abi
    h = C.c_void_p()
    rc = built.lib().osg_ctx_create(0, None, 1, C.byref(h))
    assert rc != 0
    assert b"no HIP device" in built.lib().osg_last_error() or b"hip" in built.lib().osg_last_error().lower()
    with pytest.raises(built.OsgError):
        built.Context(0)
    del _abi
