"""`open_spiel.python.observation` for the five games of this engine (python/observation.py:63-140,
python/pybind11/observer.cc:30-97): `make_observation(game, iig_obs_type)` returns an object with
`.tensor` (flat float32), `.dict` (named views into the tensor, pieces in the order the reference's
observers write them) and `set_from(state, player)` / `string_from(state, player)`.

Two observation types exist on the device — the ones `State::ObservationTensor` and
`State::InformationStateTensor` produce: the game's default observer (`make_observation(game)`) and the
perfect-recall, single-player-private one (`INFO_STATE_OBS_TYPE`).  Piece layouts:
  tic_tac_toe / connect_four / hex   "observation" [planes, rows, cols]          (DefaultObserver)
  kuhn_poker    "player" [P], "private_card" [P + 1], then "pot_contribution" [P] or "betting" [2P - 1, 2]
                (kuhn_poker.cc:72-107)
  leduc_poker   "player" [P], "private_card" [cards], "community_card" [cards], then "pot_contribution" [P]
                or "betting" [2, 3P - 2, 2]  (leduc_poker.cc:103-192)
Works with `open_spiel_amd.pyspiel_hip` games and states.
"""
import dataclasses
import enum

import numpy as np


class PrivateInfoType(enum.Enum):  # observer.h:60-67
    NONE = 0
    SINGLE_PLAYER = 1
    ALL_PLAYERS = 2


@dataclasses.dataclass(frozen=True)
class IIGObservationType:  # observer.h:75-104
    public_info: bool = True
    perfect_recall: bool = False
    private_info: PrivateInfoType = PrivateInfoType.SINGLE_PLAYER


# Corresponds to the old information_state_XXX methods (python/observation.py:58-59).
INFO_STATE_OBS_TYPE = IIGObservationType(perfect_recall=True)


def _pieces(game, info_state):
    """[(name, shape)] in tensor order."""
    name = str(game).split("(")[0]
    players = game.num_players()
    if name == "kuhn_poker":
        head = [("player", (players,)), ("private_card", (players + 1,))]
        return head + ([("betting", (2 * players - 1, 2))] if info_state else [("pot_contribution", (players,))])
    if name == "leduc_poker":
        size = game.observation_tensor_size()            # P + 2 cards + P
        cards = (size - 2 * players) // 2
        head = [("player", (players,)), ("private_card", (cards,)), ("community_card", (cards,))]
        return head + ([("betting", (2, 3 * players - 2, 2))] if info_state else [("pot_contribution", (players,))])
    if info_state:
        return None                                      # perfect-information games: no info-state tensor here
    return [("observation", tuple(game.observation_tensor_shape()))]


class _Observation:
    """Contains an observation from a game (python/observation.py:63-96)."""

    def __init__(self, game, info_state, pieces):
        self._game = game
        self._info_state = info_state
        total = int(sum(int(np.prod(shape)) for _, shape in pieces))
        expect = game.information_state_tensor_size() if info_state else game.observation_tensor_size()
        if total != expect:
            raise ValueError(f"piece layout ({total} floats) does not match the game's tensor ({expect})")
        self.tensor = np.zeros(total, np.float32)
        self.dict = {}
        offset = 0
        for name, shape in pieces:
            size = int(np.prod(shape))
            self.dict[name] = self.tensor[offset:offset + size].reshape(shape)  # a view: set_from updates it
            offset += size

    def set_from(self, state, player):
        values = (state.information_state_tensor(player) if self._info_state
                  else state.observation_tensor(player))
        self.tensor[:] = np.asarray(values, np.float32)

    def string_from(self, state, player):
        if self._info_state:
            return state.information_state_string(player)
        return state.observation_string(player)


def make_observation(game, imperfect_information_observation_type=None, params=None):
    """python/observation.py:99-125.  None when the requested observation type is not supported."""
    if params:
        raise ValueError(f"Observation parameters not supported; passed {params}")
    t = imperfect_information_observation_type
    if t is None:
        info_state = False
    elif t == INFO_STATE_OBS_TYPE:
        info_state = True
    elif t == IIGObservationType():  # the default imperfect-recall type is what ObservationTensor packs
        info_state = False
    else:
        return None
    pieces = _pieces(game, info_state)
    if pieces is None:
        return None
    return _Observation(game, info_state, pieces)
