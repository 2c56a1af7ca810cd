"""A minimal `pyspiel` module over the oracle's extern-"C" driver — TEST INFRASTRUCTURE ONLY.

Purpose: run the reference's *Python* algorithm files (open_spiel/python/algorithms/cfr.py,
exploitability.py, best_response.py, python/policy.py — imported unmodified from /root/reference)
as a second, independent specification of CFR and of the exploitability judge.  The
real `pyspiel` is a pybind11 extension that cannot be built here (pybind11_abseil / abseil are not
vendored); those Python files only need the State / Game query methods, which this module forwards
to `reference_py` (the genuine C++ games, oracle/_ref) or to `oracle_py` (the restatement).

    import pyspiel_over_capi
    pyspiel = pyspiel_over_capi.install(binding)      # puts a module named "pyspiel" in sys.modules
    from open_spiel.python.algorithms import cfr     # needs /root/reference on sys.path

Only what those files call is provided; anything else raises AttributeError (never a silent default).
"""
import enum
import sys
import types


class PlayerId(enum.IntEnum):
    DEFAULT_PLAYER_ID = 0
    CHANCE = -1
    SIMULTANEOUS = -2
    INVALID = -3
    TERMINAL = -4
    MEAN_FIELD = -5


class _GameType:
    class Dynamics(enum.Enum):
        SIMULTANEOUS = 0
        SEQUENTIAL = 1
        MEAN_FIELD = 2

    class ChanceMode(enum.Enum):
        DETERMINISTIC = 0
        EXPLICIT_STOCHASTIC = 1
        SAMPLED_STOCHASTIC = 2

    class Information(enum.Enum):
        ONE_SHOT = 0
        PERFECT_INFORMATION = 1
        IMPERFECT_INFORMATION = 2

    class Utility(enum.Enum):
        ZERO_SUM = 0
        CONSTANT_SUM = 1
        GENERAL_SUM = 2
        IDENTICAL = 3

    class RewardModel(enum.Enum):
        REWARDS = 0
        TERMINAL = 1

    def __init__(self, short_name, has_chance, imperfect):
        self.short_name = short_name
        self.long_name = short_name
        self.dynamics = _GameType.Dynamics.SEQUENTIAL
        self.chance_mode = (_GameType.ChanceMode.EXPLICIT_STOCHASTIC if has_chance
                            else _GameType.ChanceMode.DETERMINISTIC)
        self.information = (_GameType.Information.IMPERFECT_INFORMATION if imperfect
                            else _GameType.Information.PERFECT_INFORMATION)
        self.utility = _GameType.Utility.ZERO_SUM          # all five games (spiel.h GameType of each)
        self.reward_model = _GameType.RewardModel.TERMINAL
        self.provides_information_state_string = True
        self.provides_information_state_tensor = imperfect
        self.provides_observation_string = True
        self.provides_observation_tensor = True


class Game:
    def __init__(self, binding, game_string):
        self._b = binding
        self._g = binding.Game(game_string)
        name = str(self._g).split("(")[0]
        self._type = _GameType(name, self._g.has_chance, name in ("kuhn_poker", "leduc_poker"))

    def __str__(self):
        return str(self._g)

    def get_type(self):
        return self._type

    def num_players(self):
        return self._g.num_players

    def num_distinct_actions(self):
        return self._g.num_distinct_actions

    def max_chance_outcomes(self):
        return self._g.max_chance_outcomes

    def max_game_length(self):
        return self._g.max_game_length

    def observation_tensor_size(self):
        return self._g.observation_tensor_size

    def information_state_tensor_size(self):
        return self._g.information_state_tensor_size

    def observation_tensor_shape(self):
        return self._g.observation_tensor_shape()

    def information_state_tensor_shape(self):
        return self._g.information_state_tensor_shape()

    def min_utility(self):
        return self._g.min_utility

    def max_utility(self):
        return self._g.max_utility

    def new_initial_state(self):
        return State(self, self._g.new_initial_state())

    def new_initial_states(self):  # spiel.h:970-979: one state unless the game is a multi-population mean-field game
        return [self.new_initial_state()]


class State:
    def __init__(self, game, s):
        self._game = game
        self._s = s

    def get_game(self):
        return self._game

    def num_players(self):
        return self._game.num_players()

    def current_player(self):
        return self._s.current_player()

    def is_terminal(self):
        return self._s.is_terminal()

    def is_chance_node(self):
        return self._s.current_player() == PlayerId.CHANCE

    def is_simultaneous_node(self):
        return False

    def is_player_node(self):
        return self._s.current_player() >= 0

    def legal_actions(self, player=None):
        return self._s.legal_actions(player)

    def legal_actions_mask(self, player=None):
        n = (self._game.max_chance_outcomes() if self.is_chance_node() else self._game.num_distinct_actions())
        mask = [0] * n
        for a in self._s.legal_actions(player):
            mask[a] = 1
        return mask

    def chance_outcomes(self):
        return self._s.chance_outcomes()

    def apply_action(self, action):
        self._s.apply_action(int(action))

    def child(self, action):
        return State(self._game, self._s.child(int(action)))

    def clone(self):
        return State(self._game, self._s.clone())

    def returns(self):
        return self._s.returns()

    def player_return(self, player):
        return self._s.returns()[player]

    def rewards(self):
        return self._s.returns() if self._s.is_terminal() else [0.0] * self._game.num_players()

    def information_state_string(self, player=None):
        return self._s.information_state_string(self._s.current_player() if player is None else player)

    def observation_string(self, player=None):
        return self._s.observation_string(self._s.current_player() if player is None else player)

    def information_state_tensor(self, player=None):
        return list(self._s.information_state_tensor(self._s.current_player() if player is None else player))

    def observation_tensor(self, player=None):
        return list(self._s.observation_tensor(self._s.current_player() if player is None else player))

    def history(self):
        return self._s.history()

    def history_str(self):
        return self._s.history_str()

    def action_to_string(self, player, action=None):
        if action is None:
            player, action = self._s.current_player(), player
        return self._s.action_to_string(player, int(action))

    def __str__(self):
        return str(self._s)


def install(binding):
    """Create the stand-in module, register it as `pyspiel` and neutralise the imports of
    open_spiel.python that would pull in Python-implemented games; returns the module."""
    mod = types.ModuleType("pyspiel")
    mod.__doc__ = __doc__
    mod.PlayerId = PlayerId
    mod.GameType = _GameType
    mod.Game = Game
    mod.State = State
    mod.SpielError = binding.OracleError
    mod.load_game = lambda game_string, params=None: Game(
        binding, game_string if not params else
        game_string + "(" + ",".join(f"{k}={v}" for k, v in sorted(params.items())) + ")")
    mod.INVALID_ACTION = -1
    sys.modules["pyspiel"] = mod
    # python/algorithms/get_all_states.py imports open_spiel.python.games only for its side effect
    # (registering Python-implemented games, none of which is on this path).
    sys.modules.setdefault("open_spiel.python.games", types.ModuleType("open_spiel.python.games"))
    # python/rl_environment.py does `from absl import logging` (absl-py is not installed here).
    if "absl" not in sys.modules:
        import logging as _logging
        absl = types.ModuleType("absl")
        absl.logging = _logging
        sys.modules["absl"] = absl
        sys.modules["absl.logging"] = _logging
    return mod
