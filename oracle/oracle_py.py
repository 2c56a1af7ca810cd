"""ctypes binding for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  The product package (open_spiel_amd/) never imports
this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force=False):
    """Compile oracle/liboracle.so with the committed Makefile (g++ only)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE)
            if f.endswith((".cpp", ".h")) or f == "Makefile"]
    stale = (not os.path.exists(_LIB_PATH) or
             any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-j8"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.osgo_last_error.restype = C.c_char_p
        for name in ("osgo_load_game", "osgo_new_state", "osgo_clone_state", "osgo_cfr_create",
                     "osgo_cfr_deserialize"):
            getattr(_lib, name).restype = C.c_void_p
    return _lib


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise OracleError(lib().osgo_last_error().decode())
    return rc


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else None


class Game:
    def __init__(self, game_string):
        h = lib().osgo_load_game(game_string.encode())
        if not h:
            raise OracleError(lib().osgo_last_error().decode())
        self._h = C.c_void_p(h)
        info = (C.c_double * 10)()
        _check(lib().osgo_game_info(self._h, info))
        self.num_distinct_actions = int(info[0])
        self.max_chance_outcomes = int(info[1])
        self.num_players = int(info[2])
        self.observation_tensor_size = int(info[3])
        self.information_state_tensor_size = int(info[4])
        self.max_game_length = int(info[5])
        self.max_chance_nodes_in_history = int(info[6])
        self.min_utility = info[7]
        self.max_utility = info[8]
        self.has_chance = bool(info[9])

    def __del__(self):
        try:
            lib().osgo_free_game(self._h)
        except Exception:
            pass

    def __str__(self):
        buf = C.create_string_buffer(512)
        lib().osgo_game_string(self._h, 0, buf, 512)
        return buf.value.decode()

    def parameters_string(self):
        buf = C.create_string_buffer(512)
        lib().osgo_game_string(self._h, 1, buf, 512)
        return buf.value.decode()

    def shape(self, which):
        out = (C.c_int * 8)()
        n = lib().osgo_game_shape(self._h, which, out, 8)
        return [out[i] for i in range(n)]

    def observation_tensor_shape(self):
        return self.shape(0)

    def information_state_tensor_shape(self):
        return self.shape(1)

    def new_initial_state(self):
        return State(self, C.c_void_p(lib().osgo_new_state(self._h)))

    @property
    def mask_words(self):
        return (max(self.num_distinct_actions, self.max_chance_outcomes) + 31) // 32

    @property
    def max_plies(self):
        return self.max_game_length + self.max_chance_nodes_in_history

    def random_playouts(self, seed, n, stop=None, want_obs=False, want_info=False):
        """Seeded playouts with a per-ply record (see spiel_oracle_capi.cpp)."""
        L, W, P = self.max_plies, self.mask_words, self.num_players
        acts = np.full((n, L), -1, np.int16)
        mask = np.zeros((n, L + 1, W), np.uint32)
        cur = np.zeros((n, L + 1), np.int8)
        term = np.zeros((n, L + 1), np.uint8)
        rets = np.zeros((n, L + 1, P), np.float64)
        obs = np.zeros((n, L + 1, P, self.observation_tensor_size), np.float32) if want_obs else None
        info = (np.zeros((n, L + 1, P, self.information_state_tensor_size), np.float32)
                if want_info and self.information_state_tensor_size else None)
        stop_a = None if stop is None else np.ascontiguousarray(stop, np.int32)
        rc = lib().osgo_random_playouts(
            self._h, C.c_uint64(seed), C.c_int64(n), L, W, _ptr(stop_a, C.c_int32),
            _ptr(acts, C.c_int16), _ptr(mask, C.c_uint32), _ptr(cur, C.c_int8),
            _ptr(term, C.c_uint8), _ptr(rets, C.c_double), _ptr(obs, C.c_float),
            _ptr(info, C.c_float))
        _check(rc)
        return dict(actions=acts, mask=mask, cur_player=cur, terminal=term, returns=rets,
                    obs=obs, info=info, longest=rc)

    def synth_batch(self, seed, n, depth_mod, first=0, threads=1, want_obs=True, want_after=True):
        """The CPU side of osg_synth_batch (SURVEY.md 8(d) synthetic inputs; see spiel_oracle_capi.cpp): states
        first .. first + n - 1 of the stream `seed`, each with its depth, its random legal action, and the
        record of the state before / after that action."""
        W, P, osz = self.mask_words, self.num_players, self.observation_tensor_size
        out = dict(depth=np.zeros(n, np.int32), action=np.zeros(n, np.int16),
                   mask0=np.zeros((n, W), np.uint32), cur0=np.zeros(n, np.int8), term0=np.zeros(n, np.uint8),
                   obs0=np.zeros((n, osz), np.uint8) if want_obs else None)
        if want_after:
            out.update(mask1=np.zeros((n, W), np.uint32), cur1=np.zeros(n, np.int8), term1=np.zeros(n, np.uint8),
                       rets1=np.zeros((n, P), np.float64), obs1=np.zeros((n, osz), np.uint8) if want_obs else None)
        else:
            out.update(mask1=None, cur1=None, term1=None, rets1=None, obs1=None)
        _check(lib().osgo_synth_batch(
            self._h, C.c_uint64(seed), C.c_int64(first), C.c_int64(n), int(depth_mod), int(threads), W,
            _ptr(out["depth"], C.c_int32), _ptr(out["action"], C.c_int16),
            _ptr(out["mask0"], C.c_uint32), _ptr(out["cur0"], C.c_int8), _ptr(out["term0"], C.c_uint8),
            _ptr(out["obs0"], C.c_uint8),
            _ptr(out["mask1"], C.c_uint32), _ptr(out["cur1"], C.c_int8), _ptr(out["term1"], C.c_uint8),
            _ptr(out["rets1"], C.c_double), _ptr(out["obs1"], C.c_uint8)))
        return out

    def synth_tensors(self, seed, n, depth_mod, which, player, first=0, threads=1, after=False):
        """ObservationTensor (which=0) / InformationStateTensor (1) of `player` for states first .. first + n - 1 of the
        synthetic stream, before their action or (after=True) after it, as uint8 [n, size]."""
        size = self.observation_tensor_size if which == 0 else self.information_state_tensor_size
        out = np.zeros((n, size), np.uint8)
        _check(lib().osgo_synth_tensors(self._h, C.c_uint64(seed), C.c_int64(first), C.c_int64(n), int(depth_mod),
                                        int(threads), int(which), int(player), int(bool(after)), _ptr(out, C.c_uint8)))
        return out

    def synth_mcts_replay(self, seed_roots, n, depth_mod, uct_c, max_simulations, n_rollouts, counter_seed,
                          counter_layout, first=0, threads=1):
        """Replay-mode MCTSBot searches of the synthetic roots first .. first + n - 1 (restatement only)."""
        A = self.num_distinct_actions
        best = np.zeros(n, np.int32)
        visits = np.zeros((n, A), np.int32)
        reward = np.zeros((n, A), np.float64)
        _check(lib().osgo_synth_mcts_replay(
            self._h, C.c_uint64(seed_roots), C.c_int64(first), C.c_int64(n), int(depth_mod), C.c_double(uct_c),
            int(max_simulations), int(n_rollouts), C.c_uint64(counter_seed), int(counter_layout), int(threads),
            _ptr(best, C.c_int32), _ptr(visits, C.c_int32), _ptr(reward, C.c_double)))
        return dict(best_action=best, child_visits=visits, child_reward=reward)

    def replay_rollouts(self, history, seed, root_index, n_rollouts):
        h = np.ascontiguousarray(history, np.int16)
        out = np.zeros(self.num_players, np.float64)
        steps = C.c_int64(0)
        _check(lib().osgo_replay_rollouts(self._h, _ptr(h, C.c_int16), len(h), C.c_uint64(seed),
                                          C.c_uint64(root_index), n_rollouts,
                                          _ptr(out, C.c_double), C.byref(steps)))
        return out, steps.value

    def tree_census(self):
        out = (C.c_int64 * 4)()
        _check(lib().osgo_tree_census(self._h, out))
        return tuple(out)

    def eval_named_policy(self, which_policy, which, alpha=0.0):
        out = C.c_double(0)
        _check(lib().osgo_eval_named_policy(self._h, which_policy, C.c_double(alpha), which,
                                            C.byref(out)))
        return out.value

    def eval_policy(self, keys, nact, actions, probs, which=0):
        """NashConv (which=0) / exploitability (1) + expected returns of a tabular policy."""
        amax = actions.shape[1]
        nact = np.ascontiguousarray(nact, np.int32)
        actions = np.ascontiguousarray(actions, np.int64)
        probs = np.ascontiguousarray(probs, np.float64)
        out = C.c_double(0)
        ev = np.zeros(self.num_players, np.float64)
        _check(lib().osgo_eval_policy(self._h, "\n".join(keys).encode(), amax,
                                      _ptr(nact, C.c_int), _ptr(actions, C.c_int64),
                                      _ptr(probs, C.c_double), which, C.byref(out),
                                      _ptr(ev, C.c_double)))
        return out.value, ev

    def mcts_selfplay(self, uct_c, max_simulations, n_rollouts, seed):
        out = np.zeros(self.num_players, np.float64)
        _check(lib().osgo_mcts_selfplay(self._h, C.c_double(uct_c), max_simulations, n_rollouts,
                                        seed, _ptr(out, C.c_double)))
        return out

    # ---- cpu_baseline timing legs ----
    def bench_env_steps(self, seed, pool, total_steps, threads):
        secs, units = C.c_double(0), C.c_int64(0)
        _check(lib().osgo_bench_env_steps(self._h, C.c_uint64(seed), C.c_int64(pool),
                                          C.c_int64(total_steps), threads, C.byref(secs),
                                          C.byref(units)))
        return secs.value, units.value

    def bench_observation(self, seed, pool, total, threads):
        secs, units = C.c_double(0), C.c_int64(0)
        _check(lib().osgo_bench_observation(self._h, C.c_uint64(seed), C.c_int64(pool), C.c_int64(total),
                                            threads, C.byref(secs), C.byref(units)))
        return secs.value, units.value

    def bench_playouts(self, seed, sims, threads):
        secs, moves = C.c_double(0), C.c_int64(0)
        _check(lib().osgo_bench_playouts(self._h, C.c_uint64(seed), C.c_int64(sims), threads,
                                         C.byref(secs), C.byref(moves)))
        return secs.value, moves.value

    def bench_mcts(self, seed, roots, depth_mod, max_simulations, n_rollouts, uct_c, threads):
        secs, sims = C.c_double(0), C.c_int64(0)
        _check(lib().osgo_bench_mcts(self._h, C.c_uint64(seed), roots, depth_mod, max_simulations,
                                     n_rollouts, C.c_double(uct_c), threads, C.byref(secs),
                                     C.byref(sims)))
        return secs.value, sims.value

    def bench_mcts_config1(self, max_simulations, repeats):
        """BASELINE.json configs[0]: the mcts_test.cc:35-49 bot on tic_tac_toe, 4 positions x repeats searches."""
        secs, sims = C.c_double(0), C.c_int64(0)
        _check(lib().osgo_bench_mcts_config1(self._h, max_simulations, repeats, C.byref(secs), C.byref(sims)))
        return secs.value, sims.value

    def bench_cfr(self, kind, iters, threads):
        secs = C.c_double(0)
        _check(lib().osgo_bench_cfr(self._h, kind, iters, threads, C.byref(secs)))
        return secs.value


class State:
    def __init__(self, game, handle):
        self.game = game
        self._h = handle

    def __del__(self):
        try:
            lib().osgo_free_state(self._h)
        except Exception:
            pass

    def clone(self):
        return State(self.game, C.c_void_p(lib().osgo_clone_state(self._h)))

    def apply_action(self, a):
        _check(lib().osgo_apply(self._h, C.c_int64(a)))

    def child(self, a):
        c = self.clone()
        c.apply_action(a)
        return c

    def current_player(self):
        return lib().osgo_current_player(self._h)

    def is_terminal(self):
        return bool(lib().osgo_is_terminal(self._h))

    def is_chance_node(self):
        return self.current_player() == -1

    def legal_actions(self, player=None):
        out = (C.c_int64 * 512)()
        if player is None:
            n = _check(lib().osgo_legal_actions(self._h, out, 512))
        else:
            n = _check(lib().osgo_legal_actions_for(self._h, player, out, 512))
        return [out[i] for i in range(n)]

    def returns(self):
        out = (C.c_double * 16)()
        n = _check(lib().osgo_returns(self._h, out))
        return [out[i] for i in range(n)]

    def chance_outcomes(self):
        a = (C.c_int64 * 64)()
        p = (C.c_double * 64)()
        n = _check(lib().osgo_chance_outcomes(self._h, a, p, 64))
        return [(a[i], p[i]) for i in range(n)]

    def _tensor(self, which, player, size):
        out = np.zeros(size, np.float32)
        _check(lib().osgo_tensor(self._h, which, player, _ptr(out, C.c_float), size))
        return out

    def observation_tensor(self, player):
        return self._tensor(0, player, self.game.observation_tensor_size)

    def information_state_tensor(self, player):
        return self._tensor(1, player, self.game.information_state_tensor_size)

    def _string(self, which, player=0, action=0):
        buf = C.create_string_buffer(4096)
        n = lib().osgo_string(self._h, which, player, C.c_int64(action), buf, 4096)
        if n < 0:
            raise OracleError(lib().osgo_last_error().decode())
        return buf.value.decode()

    def __str__(self):
        return self._string(0)

    def information_state_string(self, player):
        return self._string(1, player)

    def observation_string(self, player):
        return self._string(2, player)

    def history_str(self):
        return self._string(3)

    def action_to_string(self, player, action):
        return self._string(4, player, action)

    def serialize_game_and_state(self):
        """SerializeGameAndState(game, state) (genuine reference build only)."""
        buf = C.create_string_buffer(1 << 16)
        n = lib().osgo_serialize_game_and_state(self._h, buf, 1 << 16)
        if n < 0:
            raise OracleError(lib().osgo_last_error().decode())
        return buf.value.decode()

    def history(self):
        out = (C.c_int64 * 512)()
        n = lib().osgo_history(self._h, out, 512)
        return [out[i] for i in range(n)]

    def observer(self, player, obs_type=None):
        """Game::MakeObserver(obs_type) on this state (genuine reference build only).  obs_type: None (the default observer)
        or (public_info, perfect_recall, private_info) with private_info 0 kNone / 1 kSinglePlayer / 2 kAllPlayers.
        Returns None when the game offers no such observer, else a dict: tensor (float32 or None), pieces
        [(name, shape)], string, compressed (bytes)."""
        pub, rec, prv = (-1, 0, 0) if obs_type is None else (int(obs_type[0]), int(obs_type[1]), int(obs_type[2]))
        tensor = np.zeros(4096, np.float32)
        spec, text = C.create_string_buffer(2048), C.create_string_buffer(4096)
        comp = (C.c_ubyte * 20000)()
        comp_len = C.c_int(0)
        n = lib().osgo_observer(self._h, pub, rec, prv, player, _ptr(tensor, C.c_float), 4096, spec, 2048, text, 4096,
                                comp, 20000, C.byref(comp_len))
        if n == -2:
            return None
        if n < 0:
            raise OracleError(lib().osgo_last_error().decode())
        pieces = None
        if spec.value != b"-":
            pieces = [(item.split(":")[0], tuple(int(d) for d in item.split(":")[1].split("x") if d))
                      for item in spec.value.decode().split(";") if item]
        return {"tensor": tensor[:n].copy() if pieces is not None else None, "pieces": pieces,
                "string": text.value.decode(), "compressed": bytes(comp[:comp_len.value])}

    def mcts_search(self, uct_c, max_simulations, n_rollouts, max_memory_mb, solve, seed,
                    counter_root=-1, counter_seed=0, counter_layout=1, puct=False):
        best = C.c_int64(0)
        root_outcome = C.c_double(0)
        visits = C.c_int(0)
        cap = 512
        ch = np.zeros((cap, 4), np.float64)
        n = _check(lib().osgo_mcts_search(self._h, C.c_double(uct_c), max_simulations, n_rollouts,
                                          C.c_int64(max_memory_mb), int(solve), seed,
                                          C.byref(best), C.byref(root_outcome),
                                          _ptr(ch, C.c_double), cap, C.byref(visits),
                                          C.c_int64(counter_root), C.c_uint64(counter_seed),
                                          int(counter_layout), int(puct)))
        return dict(best_action=best.value, root_outcome=root_outcome.value,
                    root_visits=visits.value, children=ch[:n])


def _mcts_search_stub(self, uct_c, max_simulations, counter_root, counter_seed, max_nodes=0, solve=False, puct=False,
                      dont_return_chance_node=False):
    """MCTSBot with the deterministic stub network (see spiel_oracle_capi.cpp StubNetEvaluator) on the device's
    layout-1 tree-policy streams: the replay of open_spiel_amd.mcts.search with the same network in torch."""
    best, visits, nodes = C.c_int64(0), C.c_int(0), C.c_int(0)
    cap = 512
    ch = np.zeros((cap, 4), np.float64)
    n = _check(lib().osgo_mcts_search_stub(self._h, C.c_double(uct_c), max_simulations, int(max_nodes), int(solve),
                                           int(puct), int(dont_return_chance_node), C.c_int64(counter_root),
                                           C.c_uint64(counter_seed), C.byref(best), _ptr(ch, C.c_double), cap,
                                           C.byref(visits), C.byref(nodes)))
    return dict(best_action=best.value, root_visits=visits.value, nodes=nodes.value, children=ch[:n])


State.mcts_search_stub = _mcts_search_stub


SOLVER_KINDS = {"cfr": 0, "cfr_plus": 1, "mccfr_simple": 2, "mccfr_full": 3, "cfr_simultaneous": 4,
                "mccfr_outcome": 5, "cfr_br": 6}
# CFRSolverBase(game, alternating_updates, linear_averaging, regret_matching_plus) with any switch combination
for _alt in (0, 1):
    for _lin in (0, 1):
        for _rmp in (0, 1):
            SOLVER_KINDS[f"cfr_base_alt{_alt}_lin{_lin}_rmp{_rmp}"] = 16 + _alt + 2 * _lin + 4 * _rmp


def mccfr_frozen_replay(game, keys, regrets, seed, first, count, threads=1):
    """Summed ES-MCCFR (kSimple) increments of trajectories [first, first + count) against ONE frozen table
    (osgo_mccfr_frozen_replay: the device's mini-batch at its real size, `threads` trajectories at a time).
    keys: infostate strings, regrets: [len(keys), amax] float64 (rows not listed count as 1e-6).
    Returns dict(d_regrets, d_cum_policy, mass [rows, amax], visits [rows])."""
    regrets = np.ascontiguousarray(regrets, np.float64)
    n, amax = regrets.shape
    assert n == len(keys)
    d_reg = np.zeros((n, amax), np.float64)
    d_cum = np.zeros((n, amax), np.float64)
    mass = np.zeros((n, amax), np.float64)
    visits = np.zeros(n, np.int64)
    _check(lib().osgo_mccfr_frozen_replay(game._h, C.c_uint64(seed), C.c_int64(first), C.c_int64(count), int(threads),
                                          n, amax, "\n".join(keys).encode(), _ptr(regrets, C.c_double),
                                          _ptr(d_reg, C.c_double), _ptr(d_cum, C.c_double), _ptr(mass, C.c_double),
                                          _ptr(visits, C.c_int64)))
    return dict(d_regrets=d_reg, d_cum_policy=d_cum, mass=mass, visits=visits)


class Solver:
    """CFRSolver / CFRPlusSolver / ExternalSamplingMCCFRSolver of the oracle."""

    def __init__(self, game, kind="cfr", seed=0):
        self.game = game
        h = lib().osgo_cfr_create(game._h, SOLVER_KINDS[kind], seed)
        if not h:
            raise OracleError(lib().osgo_last_error().decode())
        self._h = C.c_void_p(h)

    def __del__(self):
        try:
            lib().osgo_cfr_free(self._h)
        except Exception:
            pass

    def iterate(self, iters=1):
        _check(lib().osgo_cfr_iterate(self._h, iters))

    def serialize(self, double_precision=-1):
        """CFRSolverBase::Serialize (genuine reference build only)."""
        n = _check(lib().osgo_cfr_serialize(self._h, double_precision, None, 0))
        buf = C.create_string_buffer(n + 1)
        _check(lib().osgo_cfr_serialize(self._h, double_precision, buf, n + 1))
        return buf.value.decode()

    @classmethod
    def deserialize(cls, game, text, kind="cfr"):
        """DeserializeCFRSolver / DeserializeCFRPlusSolver (genuine reference build only).
        `game` must be the binding's Game for the checkpoint's game (used for sizes only)."""
        h = lib().osgo_cfr_deserialize(text.encode(), SOLVER_KINDS[kind])
        if not h:
            raise OracleError(lib().osgo_last_error().decode())
        self = cls.__new__(cls)
        self.game = game
        self._h = C.c_void_p(h)
        return self

    def mccfr_minibatch(self, seed, first, count):
        """The device's mini-batch ES-MCCFR schedule on the oracle (frozen table per call)."""
        _check(lib().osgo_mccfr_minibatch(self._h, C.c_uint64(seed), C.c_int64(first), C.c_int64(count)))

    def mccfr_full_average(self):
        """FullUpdateAverage on the table as it is (the second half of a kFull RunIteration)."""
        _check(lib().osgo_mccfr_full_average(self._h))

    def tables(self, amax=None):
        amax = amax or self.game.num_distinct_actions
        n = lib().osgo_cfr_num_infostates(self._h)
        cap = 256 * max(n, 1)
        keys = C.create_string_buffer(cap)
        nact = np.zeros(n, np.int32)
        legal = np.zeros((n, amax), np.int64)
        reg = np.zeros((n, amax), np.float64)
        cum = np.zeros((n, amax), np.float64)
        curp = np.zeros((n, amax), np.float64)
        avg = np.zeros((n, amax), np.float64)
        _check(lib().osgo_cfr_tables(self._h, amax, keys, cap, _ptr(nact, C.c_int),
                                     _ptr(legal, C.c_int64), _ptr(reg, C.c_double),
                                     _ptr(cum, C.c_double), _ptr(curp, C.c_double),
                                     _ptr(avg, C.c_double)))
        ks = keys.value.decode().split("\n") if n else []
        return dict(keys=ks, nact=nact, legal=legal, regrets=reg, cum_policy=cum,
                    cur_policy=curp, avg_policy=avg)

    def _eval(self, which):
        out = C.c_double(0)
        _check(lib().osgo_cfr_eval(self._h, which, C.byref(out)))
        return out.value

    def nash_conv(self):
        return self._eval(0)

    def exploitability(self):
        return self._eval(1)

    def expected_returns(self):
        out = np.zeros(self.game.num_players, np.float64)
        _check(lib().osgo_cfr_expected_returns(self._h, _ptr(out, C.c_double)))
        return out

    def best_response_value(self, player):
        """TabularBestResponse(game, player, average policy).Value(root): one term of NashConv."""
        out = C.c_double(0)
        _check(lib().osgo_cfr_best_response_value(self._h, int(player), C.byref(out)))
        return out.value
