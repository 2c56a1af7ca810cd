"""Batched State API over the HIP engine (host-side mirror of pyspiel.State).

The reference exposes one state at a time (`state.legal_actions_mask()`,
`state.apply_action(a)`, `state.is_terminal()`, `state.returns()`,
`state.observation_tensor(p)`; open_spiel/python/pybind11/pyspiel.cc:356-474).
`StateBatch` keeps those names and meanings, vectorised over N states that live
struct-of-arrays in HBM.  PyTorch is plumbing only (device buffers, streams,
torch.distributed); every rule evaluation runs in libosg_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _abi
from ._abi import OsgError, check, lib

TERMINAL_PLAYER = -4
CHANCE_PLAYER = -1


def _ptr(t):
    """Device / host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        return t.ctypes.data
    return t.data_ptr()


class Context:
    """One per (process, device): binds the engine to a HIP stream."""

    def __init__(self, device=0, stream=None):
        if not torch.cuda.is_available():
            raise OsgError("no MI355X visible (torch.cuda.is_available() is False): the engine "
                           "has no CPU fallback")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        self.torch_stream = stream
        h = C.c_void_p()
        check(lib().osg_ctx_create(device, C.c_void_p(stream.cuda_stream), 0, C.byref(h)))
        self._h = h

    def synchronize(self):
        check(lib().osg_ctx_synchronize(self._h))

    def trim(self):
        """Free what the context caches between calls (the MCTS node pool, the staging buffer)."""
        check(lib().osg_ctx_trim(self._h))

    def set_stream(self, stream):
        """Issue every later call of this context's objects on `stream` (a torch.cuda.Stream of the same device),
        e.g. the stream a graph is being captured on; returns the stream that was bound before."""
        before = self.torch_stream
        check(lib().osg_ctx_set_stream(self._h, C.c_void_p(stream.cuda_stream)))
        self.torch_stream = stream
        return before

    def close(self):
        if self._h:
            lib().osg_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Game:
    """Static description of a game string (pyspiel.Game subset)."""

    def __init__(self, game_string):
        self.game_string = game_string
        self.desc = _abi.describe(game_string)

    def num_distinct_actions(self):
        return self.desc.num_distinct_actions

    def max_chance_outcomes(self):
        return self.desc.max_chance_outcomes

    def num_players(self):
        return self.desc.num_players

    def max_game_length(self):
        return self.desc.max_game_length

    def min_utility(self):
        return self.desc.min_utility

    def max_utility(self):
        return self.desc.max_utility

    def observation_tensor_shape(self):
        return [self.desc.obs_shape[i] for i in range(self.desc.obs_rank)]

    def observation_tensor_size(self):
        return self.desc.obs_size

    def information_state_tensor_shape(self):
        return [self.desc.info_shape[i] for i in range(self.desc.info_rank)]

    def information_state_tensor_size(self):
        return self.desc.info_size

    def __str__(self):
        return self.desc.canonical.decode()

    def new_initial_states(self, ctx, n):
        return StateBatch(ctx, self.game_string, n)


class StateBatch:
    """N states of one game in HBM (all start at Game::NewInitialState())."""

    def __init__(self, ctx, game_string, n):
        self.ctx = ctx
        self.game_string = game_string
        self.n = int(n)
        h = C.c_void_p()
        check(lib().osg_batch_create(ctx._h, game_string.encode(), self.n, C.byref(h)))
        self._h = h
        self.desc = _abi.GameDesc()
        check(lib().osg_batch_describe(self._h, C.byref(self.desc)))
        self.num_players = self.desc.num_players
        self.num_distinct_actions = self.desc.num_distinct_actions

    def __len__(self):
        return self.n

    def __del__(self):
        try:
            if self._h:
                lib().osg_batch_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _dev(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.ctx.device)

    def _checked(self, t, dtype, numel, what):
        """A caller-supplied device buffer handed to a kernel by raw pointer: right dtype, on this
        context's device, contiguous, exactly `numel` elements — anything else would read or write
        out of bounds silently."""
        if not isinstance(t, torch.Tensor):
            raise OsgError(f"{what}: expected a torch tensor on {self.ctx.device}")
        if t.dtype != dtype or not t.is_cuda or t.device != self.ctx.device or not t.is_contiguous() \
                or t.numel() != numel:
            raise OsgError(f"{what}: need a contiguous {dtype} tensor of {numel} elements on {self.ctx.device}, "
                           f"got {t.dtype} {tuple(t.shape)} on {t.device}"
                           f"{'' if t.is_contiguous() else ' (non-contiguous)'}")
        return t

    # -- State::Clone / reset ------------------------------------------------
    def reset(self):
        check(lib().osg_batch_reset(self._h))

    def clone(self):
        other = StateBatch(self.ctx, self.game_string, self.n)
        check(lib().osg_batch_copy(other._h, self._h))
        return other

    def gather(self, index):
        """New batch with state i = self[index[i]] (Clone() of chosen states)."""
        idx = torch.as_tensor(index, dtype=torch.int64, device=self.ctx.device).contiguous()
        other = StateBatch(self.ctx, self.game_string, idx.numel())
        check(lib().osg_batch_gather(other._h, self._h, _ptr(idx), 0))
        return other

    def raw_words(self):
        """The SoA image [state_words, n] as a numpy array (debug / fixtures)."""
        dt = np.uint64 if self.desc.state_word_bytes == 8 else np.uint32
        out = np.empty((self.desc.state_words, self.n), dt)
        check(lib().osg_batch_download(self._h, out.ctypes.data))
        return out

    def load_raw_words(self, words):
        dt = np.uint64 if self.desc.state_word_bytes == 8 else np.uint32
        w = np.ascontiguousarray(words, dt)
        assert w.shape == (self.desc.state_words, self.n)
        check(lib().osg_batch_upload(self._h, w.ctypes.data))

    def set_cells(self, index, cells):
        """State `index` := the position with these cells ('.', 'x', 'o'; tic_tac_toe: cell a = action a; connect_four:
        cell r * cols + c with row 0 the bottom row) — TicTacToeState(game, struct) / ConnectFourState(game, struct | string)
        (osg_batch_set_cells: built on the device with the game's own rules)."""
        text = cells.encode() if isinstance(cells, str) else bytes(cells)
        check(lib().osg_batch_set_cells(self._h, int(index), text, len(text)))

    # -- State::LegalActionsMask ---------------------------------------------
    def legal_actions_mask_bits(self):
        """[n, mask_words] int32 bit-packed legal mask (chance outcomes at chance nodes)."""
        out = self._dev((self.n, self.desc.mask_words), torch.int32)
        check(lib().osg_legal_mask(self._h, _ptr(out), 0))
        return out

    def legal_actions_bool(self):
        """[n, A] bool, True where action a is legal for the player to move: the bit mask against a row of powers of
        two (two small launches; legal_actions_mask() builds the reference's 0 / 1 layout with six)."""
        bits = self.legal_actions_mask_bits()
        A = self.desc.num_distinct_actions
        if getattr(self, "_pow2", None) is None:
            a = torch.arange(A, device=self.ctx.device)
            self._pow2 = (torch.ones_like(a, dtype=torch.int64) << (a % 32)).to(torch.int32)   # bit 31 wraps to the sign bit
            self._word = (a // 32).to(torch.int64)
        words = bits if bits.shape[1] == 1 else bits.index_select(1, self._word)
        return (words & self._pow2) != 0

    def legal_actions_mask(self):
        """[n, max(A, C)] uint8, one entry per action id (State::LegalActionsMask)."""
        bits = self.legal_actions_mask_bits()
        width = max(self.desc.num_distinct_actions, self.desc.max_chance_outcomes)
        shifts = torch.arange(32, device=bits.device, dtype=torch.int32)
        expanded = ((bits.unsqueeze(-1) >> shifts) & 1).reshape(self.n, -1)
        return expanded[:, :width].to(torch.uint8)

    # -- State::ApplyAction ----------------------------------------------------
    def apply_actions(self, actions, check_legal=True):
        """Apply actions[i] to state i (-1 = leave untouched).  Illegal actions raise."""
        a = torch.as_tensor(actions, dtype=torch.int32, device=self.ctx.device).contiguous()
        if a.numel() != self.n:
            raise OsgError("apply_actions: need one action per state")
        illegal = C.c_int64(0)
        check(lib().osg_apply(self._h, _ptr(a), 0, C.byref(illegal) if check_legal else None))
        if check_legal and illegal.value:
            raise OsgError(f"{illegal.value} illegal action(s) in apply_actions")

    # -- IsTerminal / CurrentPlayer / Returns ------------------------------------
    def status(self, want_returns=True):
        cur = self._dev((self.n,), torch.int8)
        term = self._dev((self.n,), torch.uint8)
        rets = self._dev((self.n, self.num_players), torch.float64) if want_returns else None
        check(lib().osg_status_query(self._h, _ptr(cur), _ptr(term), _ptr(rets), 0))
        return cur, term, rets

    def current_player(self):
        return self.status(False)[0]

    def is_terminal(self):
        return self.status(False)[1].bool()

    def returns(self):
        return self.status(True)[2]

    def chance_outcome_probs(self):
        """[n, max_chance_outcomes] float64: ChanceOutcomes() probabilities by outcome id."""
        out = self._dev((self.n, max(self.desc.max_chance_outcomes, 1)), torch.float64)
        check(lib().osg_chance_probs(self._h, _ptr(out), 0))
        return out

    # -- tensors -----------------------------------------------------------------
    def observation_tensor(self, player=-1, out=None):
        if out is None:
            out = self._dev((self.n, self.desc.obs_size), torch.float32)
        else:
            self._checked(out, torch.float32, self.n * self.desc.obs_size, "observation_tensor(out=)")
        check(lib().osg_observation(self._h, int(player), 0, _ptr(out), 0))
        return out

    def information_state_tensor(self, player=-1, out=None):
        if out is None:
            out = self._dev((self.n, self.desc.info_size), torch.float32)
        else:
            self._checked(out, torch.float32, self.n * self.desc.info_size, "information_state_tensor(out=)")
        check(lib().osg_observation(self._h, int(player), 1, _ptr(out), 0))
        return out

    def observation_string(self, index, player):
        """State::ObservationString(player) of state `index` (host formatter over the packed words)."""
        buf = C.create_string_buffer(1024)
        rc = lib().osg_observation_string(self._h, int(index), int(player), buf, 1024)
        if rc < 0:
            check(rc)
        return buf.value.decode()

    def state_string(self, index):
        """State::ToString() of state `index`."""
        buf = C.create_string_buffer(1024)
        rc = lib().osg_state_string(self._h, int(index), buf, 1024)
        if rc < 0:
            check(rc)
        return buf.value.decode()

    def action_string(self, index, player, action):
        """State::ActionToString(player, action) for state `index` (player -1 = chance)."""
        buf = C.create_string_buffer(64)
        rc = lib().osg_action_string(self._h, int(index), int(player), int(action), buf, 64)
        if rc < 0:
            check(rc)
        return buf.value.decode()

    def information_state_string(self, index, player):
        """State::InformationStateString(player) of state `index` (kuhn_poker / leduc_poker)."""
        buf = C.create_string_buffer(1024)
        rc = lib().osg_information_state_string(self._h, int(index), int(player), buf, 1024)
        if rc < 0:
            check(rc)
        return buf.value.decode()

    # -- the fused step -------------------------------------------------------------
    def step_buffers(self):
        mask = self._dev((self.n, self.desc.compact_mask_bytes), torch.uint8)
        status = self._dev((self.n,), torch.uint8)
        return mask, status

    def step(self, actions_u8, dst=None, mask=None, status=None, want_mask=True):
        """Fused legality check + ApplyAction + status + successor legal mask.

        actions_u8: [n] uint8 device tensor (0xFF = skip).  dst: destination batch
        (default: in place).  Returns (mask bytes [n, compact_mask_bytes], status [n]).
        want_mask=False (hex boards of up to 128 cells): the successor's mask row is not written — on a hex
        board it is ~occupied of the successor record; returns (None, status).
        """
        dst = dst or self
        if not want_mask:
            mask = None
            if status is None:
                status = self._dev((self.n,), torch.uint8)
            else:
                self._checked(status, torch.uint8, self.n, "step(status=)")
        elif mask is None or status is None:
            mask, status = self.step_buffers()
        else:
            self._checked(mask, torch.uint8, self.n * self.desc.compact_mask_bytes, "step(mask=)")
            self._checked(status, torch.uint8, self.n, "step(status=)")
        self._checked(actions_u8, torch.uint8, self.n, "step(actions_u8)")
        if dst.n != self.n or dst.game_string != self.game_string:
            raise OsgError("step(dst=): destination batch of a different game or size")
        check(lib().osg_step(self._h, dst._h, _ptr(actions_u8), None if mask is None else _ptr(mask), _ptr(status)))
        return mask, status

    # -- random play ------------------------------------------------------------------
    def random_steps(self, seed, steps, counters=None, index_offset=0):
        """`steps` uniformly random env steps per state with auto-reset; returns the
        [2] uint64-as-int64 device counters (steps applied, episodes finished)."""
        if counters is None:
            counters = torch.zeros(2, dtype=torch.int64, device=self.ctx.device)
        check(lib().osg_random_steps(self._h, int(seed), int(index_offset), int(steps), _ptr(counters)))
        return counters

    def synth(self, seed, depth_mod, index_offset=0):
        """SURVEY.md 8(d) synthetic inputs (osg_synth_batch): state i becomes the initial state advanced by
        depth_i = draw mod depth_mod random legal moves on the counter stream (seed, index_offset + i), never
        terminal; returns (actions [n] uint8 — one random legal action per state —, depth [n] int32).  The CPU
        oracle regenerates the same batch from (seed, index range, depth_mod)."""
        actions = self._dev((self.n,), torch.uint8)
        depth = self._dev((self.n,), torch.int32)
        check(lib().osg_synth_batch(self._h, int(seed), int(index_offset), int(depth_mod), _ptr(actions), _ptr(depth)))
        return actions, depth

    def rollout(self, seed, n_rollouts, index_offset=0, want_steps=False):
        """RandomRolloutEvaluator: SUM of Returns() over n_rollouts playouts per root."""
        total = self._dev((self.n, self.num_players), torch.float64)
        steps = self._dev((self.n,), torch.int32) if want_steps else None
        check(lib().osg_rollout(self._h, int(seed), int(index_offset), int(n_rollouts), _ptr(total),
                                _ptr(steps), 0))
        return (total, steps) if want_steps else total

    def mcts_search(self, uct_c=2.0, max_simulations=1024, n_rollouts=1, solve=False, max_nodes=0,
                    seed=0, index_offset=0, layout=0, puct=False):
        """MCTSBot.mcts_search for every root.  layout: 0 auto, 1 lane per root, 2 wave per root;
        puct: ChildSelectionPolicy.PUCT instead of UCT."""
        A = self.num_distinct_actions
        cfg = _abi.MctsCfg(uct_c, max_simulations, n_rollouts, int(solve), max_nodes, seed, index_offset,
                           int(layout), 1 if puct else 0)
        best = self._dev((self.n,), torch.int32)
        visits = self._dev((self.n, A), torch.int32)
        reward = self._dev((self.n, A), torch.float64)
        outcome = self._dev((self.n, A), torch.int8)
        stats = self._dev((self.n, 4), torch.float64)
        check(lib().osg_mcts_search(self._h, C.byref(cfg), _ptr(best), _ptr(visits), _ptr(reward),
                                    _ptr(outcome), _ptr(stats), 0))
        return dict(best_action=best, child_visits=visits, child_reward=reward, child_outcome=outcome,
                    root_stats=stats)


class TabularSolver:
    """CFRSolver / CFRPlusSolver / external-sampling MCCFR on the device.

    Mirrors pyspiel.CFRSolver(game).evaluate_and_update_policy() /
    average_policy() (open_spiel/python/pybind11/policy.cc:224-333).
    """

    def __init__(self, ctx, game_string, alternating_updates=True, linear_averaging=False,
                 regret_matching_plus=False, mccfr=False, general_kernel=False, epsilon=0.6, replicas=1,
                 random_initial_regrets=False, seed=0, replica_offset=0):
        """mccfr: False (CFR family), True / "external" (ES-MCCFR) or "outcome" (OS-MCCFR, `epsilon`)."""
        self.ctx = ctx
        self.game_string = game_string
        solver = {False: 0, True: 1, "external": 1, "outcome": 2}[mccfr]
        cfg = _abi.CfrCfg(int(alternating_updates), int(linear_averaging), int(regret_matching_plus),
                          solver, float(epsilon), {False: 0, True: 1, "grid": 2, "path": 3, "split": 4, "sub": 5}[general_kernel], int(replicas),
                          int(random_initial_regrets), int(seed), int(replica_offset))
        self.replicas = int(replicas)
        h = C.c_void_p()
        check(lib().osg_cfr_create(ctx._h, game_string.encode(), C.byref(cfg), C.byref(h)))
        self._h = h
        sizes = (C.c_int64 * 6)()
        check(lib().osg_cfr_sizes(self._h, sizes))
        (self.num_histories, self.num_chance, self.num_decision, self.num_terminal,
         self.num_infostates, self.amax) = [int(v) for v in sizes]

    def __del__(self):
        try:
            if self._h:
                lib().osg_cfr_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def evaluate_and_update_policy(self, iters=1):
        check(lib().osg_cfr_iterate(self._h, int(iters)))

    def evaluate_and_update_policy_cfr_br(self, iters=1):
        """CFRBRSolver.evaluate_and_update_policy (cfr_br.cc:48-83): every player's regret / average-policy pass
        runs against the other players' best responses to the current policy.  The solver must have been
        created as plain CFR (no linear averaging, no RM+)."""
        check(lib().osg_cfr_br_iterate(self._h, int(iters)))

    def set_average_type(self, full):
        """ES-MCCFR AverageType (external_sampling_mccfr.h:48): False kSimple, True kFull."""
        check(lib().osg_mccfr_set_average_type(self._h, 1 if full else 0))

    def mccfr_full_average(self, weight=1.0):
        """FullUpdateAverage (external_sampling_mccfr.cc:188-231) on the tables as they are."""
        check(lib().osg_mccfr_full_average(self._h, C.c_double(weight)))

    def mccfr_sample_uniforms(self, player, uniforms):
        """One UpdateRegrets(root, player, rng) whose draws are `uniforms` in visiting order (the sequence the
        reference's std::mt19937 + uniform_real_distribution would give); returns how many were used.  The
        deltas are left in mccfr_delta_tables(): fold with mccfr_apply_deltas()."""
        u = np.ascontiguousarray(uniforms, np.float64)
        used = C.c_int32(0)
        check(lib().osg_mccfr_sample_uniforms(self._h, int(player), u.ctypes.data, int(u.size), C.byref(used)))
        return used.value

    def reset(self):
        check(lib().osg_cfr_reset(self._h))

    def select_replica(self, r):
        """Which of the `replicas` independent solvers tables() / evaluate_policy() look at."""
        check(lib().osg_cfr_select_replica(self._h, int(r)))

    @property
    def iteration(self):
        return lib().osg_cfr_iteration(self._h)

    def last_kernel(self):
        """The kernel family the last iterate / sample call launched (diagnostic, osg_cfr_last_kernel)."""
        return lib().osg_cfr_last_kernel(self._h).decode()

    def last_eval_kernel(self):
        """The form the last policy evaluation took (diagnostic, osg_cfr_last_eval_kernel)."""
        return lib().osg_cfr_last_eval_kernel(self._h).decode()

    def run_mccfr(self, seed, trajectories, first_trajectory=0):
        """One mini-batch of external-sampling traversals, folded into the tables."""
        check(lib().osg_mccfr_iterate(self._h, int(seed), int(first_trajectory), int(trajectories)))

    def mccfr_sample(self, seed, trajectories, first_trajectory=0):
        """Traversals only: deltas stay in mccfr_delta_tables() (all-reduce them, then
        mccfr_apply_deltas())."""
        check(lib().osg_mccfr_sample(self._h, int(seed), int(first_trajectory), int(trajectories)))

    def load_tables(self, regrets=None, cum_policy=None, cur_policy=None):
        arrs = [None if a is None else np.ascontiguousarray(a, np.float64) for a in (regrets, cum_policy, cur_policy)]
        for a in arrs:
            assert a is None or a.shape == (self.num_infostates, self.amax)
        check(lib().osg_cfr_upload_tables(self._h, *[None if a is None else a.ctypes.data for a in arrs]))

    def _wrap(self, ptr, shape):
        """Zero-copy torch view of device fp64 memory owned by the solver."""
        holder = type("Holder", (), {})()
        holder.__cuda_array_interface__ = {
            "shape": tuple(shape), "typestr": "<f8", "data": (ptr, False), "version": 2, "strides": None}
        return torch.as_tensor(holder, device=self.ctx.device)

    def device_tables(self):
        """(regrets, cumulative policy, current policy) as [I, Amax] device views."""
        r, c, p = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().osg_cfr_table_ptrs(self._h, C.byref(r), C.byref(c), C.byref(p)))
        shape = (self.num_infostates, self.amax)
        return self._wrap(r.value, shape), self._wrap(c.value, shape), self._wrap(p.value, shape)

    def mccfr_delta_flat(self):
        """Both MCCFR delta tables as ONE [2, I, Amax] device view (they are adjacent in the
        solver's allocation), so a multi-GPU job needs a single all-reduce per mini-batch."""
        r, c = C.c_void_p(), C.c_void_p()
        check(lib().osg_mccfr_delta_ptrs(self._h, C.byref(r), C.byref(c)))
        n = self.num_infostates * self.amax
        if c.value != r.value + 8 * n:
            raise OsgError("delta tables are not adjacent")
        return self._wrap(r.value, (2, self.num_infostates, self.amax))

    def mccfr_delta_tables(self):
        flat = self.mccfr_delta_flat()
        return flat[0], flat[1]

    def mccfr_apply_deltas(self):
        check(lib().osg_mccfr_apply_deltas(self._h))

    def mccfr_new_delta_buffer(self):
        """A caller-owned [2, I, Amax] fp64 device buffer (regret deltas | average-policy deltas) for
        mccfr_sample_into / mccfr_apply_deltas_from: two of them double-buffer the exchange step."""
        return torch.zeros((2, self.num_infostates, self.amax), dtype=torch.float64, device=self.ctx.device)

    def _delta_buffer_ptr(self, buf, what):
        want = (2, self.num_infostates, self.amax)
        if (not isinstance(buf, torch.Tensor) or buf.dtype != torch.float64 or tuple(buf.shape) != want
                or not buf.is_contiguous() or buf.device != self.ctx.device):
            raise OsgError(f"{what}: expected a contiguous float64 tensor of shape {want} on {self.ctx.device}")
        return buf.data_ptr()

    def mccfr_sample_into(self, buf, seed, trajectories, first_trajectory=0):
        """Traversals only, deltas left in the caller's buffer (osg_mccfr_sample_into)."""
        check(lib().osg_mccfr_sample_into(self._h, int(seed), int(first_trajectory), int(trajectories),
                                          self._delta_buffer_ptr(buf, "mccfr_sample_into")))

    def mccfr_apply_deltas_from(self, buf):
        check(lib().osg_mccfr_apply_deltas_from(self._h, self._delta_buffer_ptr(buf, "mccfr_apply_deltas_from")))

    def tables(self):
        I, A = self.num_infostates, self.amax
        nact = np.zeros(I, np.int32)
        legal = np.zeros((I, A), np.int32)
        out = {k: np.zeros((I, A), np.float64) for k in ("regrets", "cum_policy", "cur_policy", "avg_policy")}
        check(lib().osg_cfr_tables(self._h, nact.ctypes.data, legal.ctypes.data, out["regrets"].ctypes.data,
                                   out["cum_policy"].ctypes.data, out["cur_policy"].ctypes.data,
                                   out["avg_policy"].ctypes.data))
        keys = []
        buf = C.create_string_buffer(512)
        for i in range(I):
            check(min(lib().osg_cfr_infostate_key(self._h, i, buf, 512), 0))
            keys.append(buf.value.decode())
        out.update(keys=keys, nact=nact, legal=legal)
        return out

    def evaluate_policy(self, which="average", table=None):
        """NashConv / exploitability / expected returns / best-response values of a policy on the
        device (algorithms::NashConv, Exploitability, ExpectedReturns, TabularBestResponse).
        which: "average" | "current" | "table" (then `table` is an [I, Amax] array in this
        solver's infostate order)."""
        code = {"average": 0, "current": 1, "table": 2}[which]
        P = _abi.describe(self.game_string).num_players
        ev, br = np.zeros(P), np.zeros(P)
        nc, ex = C.c_double(0), C.c_double(0)
        tab = None
        if code == 2:
            tab = np.ascontiguousarray(table, np.float64)
            assert tab.shape == (self.num_infostates, self.amax)
        check(lib().osg_cfr_evaluate_policy(self._h, code, None if tab is None else tab.ctypes.data,
                                            ev.ctypes.data, br.ctypes.data, C.byref(nc), C.byref(ex)))
        return dict(nash_conv=nc.value, exploitability=ex.value, expected_returns=ev, best_response_values=br)

    def nash_conv(self):
        return self.evaluate_policy()["nash_conv"]

    def exploitability(self):
        return self.evaluate_policy()["exploitability"]

    def average_policy(self):
        t = self.tables()
        return {k: [(int(t["legal"][i, a]), float(t["avg_policy"][i, a])) for a in range(t["nact"][i])]
                for i, k in enumerate(t["keys"])}
