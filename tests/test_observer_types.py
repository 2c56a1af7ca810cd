"""Every IIGObservationType (public_info x perfect_recall x PrivateInfoType NONE / SINGLE_PLAYER / ALL_PLAYERS), the
default observer and Observation.compress / decompress against outputs of the GENUINE reference
(tests/golden/observer_vectors.json, written by tests/golden/make_observer_vectors.py from oracle/_ref: observer.cc,
kuhn_poker.cc:65-165, leduc_poker.cc:92-242).

The two tensors the device packs are the default and the information-state observer's; every other type is the same
pieces chosen and arranged as the type asks, composed on the host — by open_spiel_amd/observation.py (checked here on
the CPU over the restatement's states, and on the GPU over pyspiel states) and by the C++ mirror's Observation
(checked on the GPU through pyspiel._Observation)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


@pytest.fixture(scope="module")
def vectors():
    with open(os.path.join(ROOT, "tests", "golden", "observer_vectors.json")) as f:
        return json.load(f)


def _unpack(packed):
    from make_observer_vectors import unpack
    return unpack(packed)


def _type(obs_module, t):
    if t is None:
        return None
    prv = {0: obs_module.PrivateInfoType.NONE, 1: obs_module.PrivateInfoType.SINGLE_PLAYER, 2: obs_module.PrivateInfoType.ALL_PLAYERS}[t[2]]
    return obs_module.IIGObservationType(public_info=bool(t[0]), perfect_recall=bool(t[1]), private_info=prv)


def _check(make, game, new_state, apply, golden, types):
    """make(t) -> an object with tensor / pieces() / set_from / string_from / compress / decompress."""
    checked = 0
    for record in golden["records"]:
        state = new_state()
        for a in record["history"]:
            apply(state, a)
        for ti, t in enumerate(types):
            want = record["observers"][ti]
            if want == "skipped":
                continue
            obs = make(t)
            if want is None:
                assert obs is None, (t, "the reference offers no such observer")
                continue
            assert obs is not None, t
            pieces = golden["pieces"][str(ti)]
            for player, w in enumerate(want):
                tensor = _unpack(w["tensor"])
                if tensor is None:
                    assert obs["tensor"]() is None or obs["tensor"]().size == 0
                else:
                    obs["set_from"](state, player)
                    np.testing.assert_array_equal(obs["tensor"](), tensor, err_msg=f"{t} player {player} after {record['history']}")
                    assert [(n, tuple(s)) for n, s in obs["pieces"]()] == [(n, tuple(s)) for n, s in pieces]
                assert obs["string_from"](state, player) == w["string"], (t, player, record["history"])
                if "compressed" in w and tensor is not None:
                    packed = obs["compress"]()
                    assert packed.hex() == w["compressed"]
                    obs["tensor"]()[...] = 7.0
                    obs["decompress"](packed)
                    np.testing.assert_array_equal(obs["tensor"](), tensor)
                checked += 1
    return checked


class _OracleGame:
    """The methods open_spiel_amd.observation asks a game for, over the CPU restatement (oracle_py)."""

    def __init__(self, game_string):
        import oracle_py
        self.g = oracle_py.Game(game_string)
        self._s = game_string

    def __str__(self):
        return self._s

    def num_players(self): return self.g.num_players
    def max_chance_outcomes(self): return self.g.max_chance_outcomes
    def observation_tensor_size(self): return self.g.observation_tensor_size
    def information_state_tensor_size(self): return self.g.information_state_tensor_size
    def observation_tensor_shape(self): return list(self.g.observation_tensor_shape())


class _OracleState:
    def __init__(self, s): self.s = s
    def observation_tensor(self, p): return self.s.observation_tensor(p)
    def information_state_tensor(self, p): return self.s.information_state_tensor(p)
    def observation_string(self, p): return self.s.observation_string(p)
    def information_state_string(self, p): return self.s.information_state_string(p)
    def history(self): return self.s.history()


def _python_observation(obs_module, game, t):
    o = obs_module.make_observation(game, _type(obs_module, t))
    if o is None:
        return None
    return {"tensor": lambda: o.tensor, "pieces": lambda: [(n, v.shape) for n, v in o.dict.items()], "set_from": o.set_from,
            "string_from": o.string_from, "compress": o.compress, "decompress": o.decompress}


def test_observation_py_serves_every_type_over_the_restatement(vectors):
    """CPU: the composition logic of open_spiel_amd/observation.py against the genuine reference's outputs, fed from the
    restatement's two tensors and two strings (which tests/test_oracle_vs_reference.py pins on the genuine build)."""
    from open_spiel_amd import observation as obs_module
    types = [tuple(t) if t else None for t in vectors["types"]]
    total = 0
    for game_string, golden in vectors["games"].items():
        game = _OracleGame(game_string)
        total += _check(lambda t: _python_observation(obs_module, game, t), game,
                        lambda: _OracleState(game.g.new_initial_state()), lambda st, a: st.s.apply_action(a), golden, types)
    assert total > 1500


@pytest.mark.gpu
def test_observation_py_serves_every_type_over_the_device_states(vectors):
    import pyspiel
    from open_spiel_amd import observation as obs_module
    types = [tuple(t) if t else None for t in vectors["types"]]
    for game_string, golden in vectors["games"].items():
        game = pyspiel.load_game(game_string)
        _check(lambda t: _python_observation(obs_module, game, t), game, game.new_initial_state,
               lambda st, a: st.apply_action(a), golden, types)


@pytest.mark.gpu
def test_the_mirror_observation_serves_every_type(vectors):
    """GPU: Game::MakeObserver(type) + Observation of the C++ host mirror through pyspiel (make_observer, _Observation:
    set_from, string_from, tensors_info, compress, decompress)."""
    import pyspiel
    types = [tuple(t) if t else None for t in vectors["types"]]

    def make(game, t):
        observer = game.make_observer(_type(pyspiel, t)) if t is not None else game.make_observer()
        if observer is None:
            return None
        o = pyspiel._Observation(game, observer)
        view = lambda: np.asarray(o) if o.has_tensor() else None   # noqa: E731 - the buffer protocol: a view
        return {"tensor": view, "pieces": lambda: [(i.name, tuple(i.shape)) for i in o.tensors_info()],
                "set_from": o.set_from, "string_from": o.string_from, "compress": o.compress, "decompress": o.decompress}

    total = 0
    for game_string, golden in vectors["games"].items():
        game = pyspiel.load_game(game_string)
        total += _check(lambda t: make(game, t), game, game.new_initial_state, lambda st, a: st.apply_action(a), golden, types)
    assert total > 1500
