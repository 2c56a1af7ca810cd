"""`import pyspiel` for the MI355X hot path: the names of the reference's pybind11 module
(open_spiel/python/pybind11/pyspiel.cc:356-731, bots.cc:106-149, policy.cc:90-333, observer.cc:30-97 and the game
submodules games_{tic_tac_toe,connect_four,kuhn_poker,leduc_poker}.cc) served by open_spiel_amd.pyspiel_hip — the
pybind11 module over the C++ host mirror over the C-ABI (libosg_hip.so, hand-written HIP for gfx950).  Put the
repository root on sys.path and code written against `pyspiel` for the five games of the path runs unchanged; there
is no CPU fallback (without the HIP library or a GPU the first call raises).

The reference gives every game its own State subclass (`pyspiel.tic_tac_toe.TicTacToeState`, `.connect_four.
ConnectFourState`, `.leduc_poker.LeducState`; games_tic_tac_toe.cc:75, games_connect_four.cc:75,
games_leduc_poker.cc:39) and user code tests for them with isinstance.  The device keeps ONE State class for every
game (a state is a record in a struct-of-arrays batch), so those names are virtual classes here: isinstance(state,
pyspiel.tic_tac_toe.TicTacToeState) is true exactly for the states of that game, and the game-specific accessors
(board(), private_card(), ...) are methods of State that refuse other games.
"""
import sys as _sys

from open_spiel_amd import pyspiel_hip as _hip
from open_spiel_amd.pyspiel_hip import *  # noqa: F401,F403 - the alias IS the point

for _name in dir(_hip):
    if not _name.startswith("__"):
        globals()[_name] = getattr(_hip, _name)


def _short_name_of(obj):
    try:
        game = obj.get_game() if isinstance(obj, _hip.State) else obj
        return game.get_type().short_name
    except Exception:  # noqa: BLE001 - anything that is not one of our states / games is simply not an instance
        return None


def _virtual_class(name, base, short_name, doc):
    class _Meta(type):
        def __instancecheck__(cls, obj):
            return isinstance(obj, base) and _short_name_of(obj) == short_name

        def __subclasscheck__(cls, sub):
            return sub is cls

        def __call__(cls, *args, **kwargs):
            raise TypeError(f"{name} objects come from pyspiel.load_game('{short_name}'), as in the reference")

    return _Meta(name, (), {"__doc__": doc, "__module__": f"pyspiel.{short_name}"})


for _mod, _state_name in ((_hip.tic_tac_toe, "TicTacToeState"), (_hip.connect_four, "ConnectFourState"),
                          (_hip.leduc_poker, "LeducState"), (_hip.kuhn_poker, "KuhnState")):
    _short = _mod.__name__.rsplit(".", 1)[-1]
    if not hasattr(_mod, _state_name):
        setattr(_mod, _state_name, _virtual_class(_state_name, _hip.State, _short,
                                                  f"isinstance(state, {_state_name}): the states of {_short}"))
    _game_name = _state_name.replace("State", "Game")
    if not hasattr(_mod, _game_name):
        setattr(_mod, _game_name, _virtual_class(_game_name, _hip.Game, _short,
                                                 f"isinstance(game, {_game_name}): {_short} games"))
    _sys.modules[f"pyspiel.{_short}"] = _mod   # `import pyspiel.tic_tac_toe` / `from pyspiel import leduc_poker`
