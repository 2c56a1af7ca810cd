"""`open_spiel.python.observation` for the five games of this engine (python/observation.py:63-140,
python/pybind11/observer.cc:30-97): `make_observation(game, iig_obs_type)` returns an object with
`.tensor` (flat float32), `.dict` (named views into the tensor, pieces in the order the reference's
observers write them) and `set_from(state, player)` / `string_from(state, player)`.

Two observation types are packed on the device — the ones `State::ObservationTensor` and
`State::InformationStateTensor` produce: the game's default observer (`make_observation(game)`) and the
perfect-recall, single-player-private one (`INFO_STATE_OBS_TYPE`).  Every other `IIGObservationType` of the two
poker games (public_info x perfect_recall x PrivateInfoType NONE / SINGLE_PLAYER / ALL_PLAYERS) is the same pieces
chosen and arranged as the type asks, composed here from those two tensors; `compress()` / `decompress()` are
observer.cc:246-321.  Piece layouts of the two device tensors:
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


def _general_pieces(game, t):
    """Any IIGObservationType of the two poker games (kuhn_poker.cc:72-107, leduc_poker.cc:166-186)."""
    name = str(game).split("(")[0]
    players = game.num_players()
    out = []
    if name == "kuhn_poker":
        if t.private_info == PrivateInfoType.SINGLE_PLAYER:
            out += [("player", (players,)), ("private_card", (players + 1,))]
        if t.public_info:
            out.append(("betting", (2 * players - 1, 2)) if t.perfect_recall else ("pot_contribution", (players,)))
    else:
        cards = (game.observation_tensor_size() - 2 * players) // 2
        out.append(("player", (players,)))
        if t.private_info == PrivateInfoType.SINGLE_PLAYER:
            out.append(("private_card", (cards,)))
        elif t.private_info == PrivateInfoType.ALL_PLAYERS:
            out.append(("private_cards", (players, cards)))
        if t.public_info:
            out.append(("community_card", (cards,)))
            out.append(("betting", (2, 3 * players - 2, 2)) if t.perfect_recall else ("pot_contribution", (players,)))
    return out


def _chunks(text):
    out, pos = [], 0
    while pos < len(text):
        end = text.find("]", pos)
        if end < 0:
            break
        out.append(text[pos:end + 1])
        pos = end + 1
    return out


def _chunk(chunks, prefix):
    return next((c for c in chunks if c.startswith(prefix)), "")


class _Observation:
    """Contains an observation from a game (python/observation.py:63-96).

    kind: "default" / "info_state" (the two tensors the device packs), "general" (any other type of the poker games,
    composed from their pieces), "no_private" (a board game asked for private information only: empty), "info_string"
    (a board game's perfect-recall observer: the information-state string, no tensor)."""

    def __init__(self, game, kind, pieces, obs_type=None):
        self._game = game
        self._kind = kind
        self._type = obs_type
        self._info_state = kind == "info_state"
        self._pieces = pieces
        total = int(sum(int(np.prod(shape)) for _, shape in pieces))
        if kind in ("default", "info_state"):
            expect = game.information_state_tensor_size() if self._info_state else game.observation_tensor_size()
            if total != expect:
                raise ValueError(f"piece layout ({total} floats) does not match the game's tensor ({expect})")
        self.tensor = np.zeros(total, np.float32) if kind != "info_string" else None
        self.dict = {}
        offset = 0
        for name, shape in pieces:
            size = int(np.prod(shape))
            self.dict[name] = self.tensor[offset:offset + size].reshape(shape)  # a view: set_from updates it
            offset += size

    def _piece(self, state, player, name):
        info = name == "betting"
        values = np.asarray(state.information_state_tensor(player) if info else state.observation_tensor(player), np.float32)
        offset = 0
        for n, shape in _pieces(self._game, info):
            size = int(np.prod(shape))
            if n == name:
                return values[offset:offset + size]
            offset += size
        raise KeyError(name)

    def set_from(self, state, player):
        if self._kind in ("no_private", "info_string"):
            return
        if self._kind == "general":
            for name, shape in self._pieces:
                if name == "private_cards":
                    rows = [self._piece(state, q, "private_card") for q in range(self._game.num_players())]
                    self.dict[name][...] = np.stack(rows).reshape(shape)
                else:
                    self.dict[name][...] = self._piece(state, player, name).reshape(shape)
            return
        values = (state.information_state_tensor(player) if self._info_state
                  else state.observation_tensor(player))
        self.tensor[:] = np.asarray(values, np.float32)

    def string_from(self, state, player):
        if self._kind == "no_private":
            return ""
        if self._kind in ("info_state", "info_string"):
            return state.information_state_string(player)
        if self._kind == "default":
            return state.observation_string(player)
        return self._general_string(state, player)

    def _general_string(self, state, player):
        t, players = self._type, self._game.num_players()
        single, none = t.private_info == PrivateInfoType.SINGLE_PLAYER, t.private_info == PrivateInfoType.NONE
        if str(self._game).split("(")[0] == "kuhn_poker":           # kuhn_poker.cc:109-165
            h = list(state.history())
            out = ""
            if single:
                if t.perfect_recall or t.public_info:
                    if len(h) > player:
                        out += str(h[player])
                elif len(h) == 1 + player:
                    out += f"Received card {h[player]}"
            if t.public_info:
                if t.perfect_recall:
                    out += "".join("b" if a else "p" for a in h[players:])
                elif none:
                    if not h:
                        out += "start game"
                    elif len(h) > players:
                        out += "Bet" if h[-1] else "Pass"
                elif len(h) > player:
                    out += "".join(str(int(a)) for a in self._piece(state, player, "pot_contribution"))
            if t.public_info and none and h and len(h) <= players:
                out += f"Deal to player {len(h) - 1}"
            return out
        obs = _chunks(state.observation_string(player))                # leduc_poker.cc:192-238
        out = ""
        if single:
            out += _chunk(obs, "[Observer: ") + _chunk(obs, "[Private: ")
        elif t.private_info == PrivateInfoType.ALL_PLAYERS:
            cards = [_chunk(_chunks(state.observation_string(q)), "[Private: ")[10:-1] for q in range(players)]
            out += "[Privates: " + "".join(cards) + "]"
        if t.public_info:
            out += "".join(_chunk(obs, p) for p in ("[Round ", "[Player: ", "[Pot: ", "[Money: ", "[Public: "))
            if t.perfect_recall:
                info = _chunks(state.information_state_string(player))
                out += _chunk(info, "[Round1: ") + _chunk(info, "[Round2: ")
            else:
                out += _chunk(obs, "[Ante: ")
        return out

    def compress(self):
        """observer.cc:246-309: a header byte 1 + one bit per element when every element is 0 or 1, else 0 + the floats."""
        flat = self.tensor if self.tensor is not None else np.zeros(0, np.float32)
        if np.all((flat == 0) | (flat == 1)):
            return b"\x01" + np.packbits(flat != 0, bitorder="little").tobytes()
        return b"\x00" + flat.astype(np.float32).tobytes()

    def decompress(self, compressed):
        flat = self.tensor if self.tensor is not None else np.zeros(0, np.float32)
        if not compressed:
            raise ValueError("decompress: empty string")
        if compressed[0] == 1:
            if len(compressed) != 1 + (flat.size + 7) // 8:
                raise ValueError("decompress: size does not match the observation")
            bits = np.unpackbits(np.frombuffer(compressed[1:], np.uint8), bitorder="little")[:flat.size]
            flat[:] = bits.astype(np.float32)
        elif compressed[0] == 0:
            if len(compressed) != 1 + 4 * flat.size:
                raise ValueError("decompress: size does not match the observation")
            flat[:] = np.frombuffer(compressed[1:], np.float32)
        else:
            raise ValueError("Unrecognized compression scheme")


def make_observation(game, imperfect_information_observation_type=None, params=None):
    """python/observation.py:99-125.  None when the requested observation type is not supported."""
    if params:
        raise ValueError(f"Observation parameters not supported; passed {params}")
    t = imperfect_information_observation_type
    poker = game.max_chance_outcomes() > 0
    if t is not None and not poker:                      # perfect-information games: observer.cc:150-161
        if not t.public_info:
            return _Observation(game, "no_private", [], t)
        if t.perfect_recall:
            return _Observation(game, "info_string", [], t)
        return _Observation(game, "default", _pieces(game, False), t)
    if t is None or t == IIGObservationType():           # the default imperfect-recall type is what ObservationTensor packs
        kind = "default"
    elif t == INFO_STATE_OBS_TYPE:
        kind = "info_state"
    else:
        return _Observation(game, "general", _general_pieces(game, t), t)
    pieces = _pieces(game, kind == "info_state")
    if pieces is None:
        return None
    return _Observation(game, kind, pieces, t)
