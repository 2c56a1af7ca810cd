"""Batched reinforcement-learning environment on the device.

The reference steps one `rl_environment.Environment` at a time and vectorises with a Python
loop (`open_spiel/python/vector_env.py:51-54`, `open_spiel/python/rl_environment.py:379-418`).
Here N environments of one game are one `StateBatch` in HBM and a step is one kernel launch
(`osg_env_step`) plus one tensor-pack launch per player.  Names and semantics follow
`rl_environment.Environment` / `TimeStep` / `StepType` and `vector_env.SyncVectorEnv`:

  * `step(actions)` applies actions[i] to environment i; an environment whose previous time
    step was LAST starts a new episode instead and ignores its action;
  * chance events are sampled on the device right after every action and reset;
  * rewards are the terminal returns at LAST and 0 before (every game here has
    RewardModel::kTerminal); discounts are `discount`, 0 at LAST; both are None at FIRST in
    the reference — here the rows of FIRST steps are 0 and `step_type` tells them apart.

Observations are torch tensors living in HBM ([n, size] per player), ready for a model.

`compact=True` keeps the side arrays in the device's compact form (`osg_env_step_compact`: one action byte, one flag
byte, rewards as signed bytes holding twice the return — 41 instead of 60 bytes per connect_four environment and step)
and converts to the reference's TimeStep types only in `get_time_step`; the time steps are the same.
"""
import collections
import ctypes as C
import enum

import torch

from . import _abi
from ._abi import check, lib
from .engine import StateBatch


class StepType(enum.IntEnum):  # rl_environment.py:119-135
    FIRST = 0
    MID = 1
    LAST = 2


class ObservationType(enum.Enum):  # rl_environment.py:149-152
    OBSERVATION = 0
    INFORMATION_STATE = 1


class TimeStep(collections.namedtuple("TimeStep", ["observations", "rewards", "discounts", "step_type"])):
    """Batched `rl_environment.TimeStep`: every field has a leading environment dimension."""
    __slots__ = ()

    def first(self):
        return self.step_type == StepType.FIRST

    def mid(self):
        return self.step_type == StepType.MID

    def last(self):
        return self.step_type == StepType.LAST

    def current_player(self):
        return self.observations["current_player"]


class BatchedEnvironment:
    """N environments of one game; the batched counterpart of `rl_environment.Environment`."""

    def __init__(self, ctx, game_string, num_envs, discount=1.0, observation_type=None, seed=0, index_offset=0,
                 compact=False):
        if not 0.0 <= float(discount) <= 1.0:
            raise ValueError(f"discount must be in [0.0, 1.0], got {discount}")  # InvalidParameterError
        self.ctx = ctx
        self.batch = StateBatch(ctx, game_string, num_envs)
        d = self.batch.desc
        self.num_envs = int(num_envs)
        self.num_players = d.num_players
        self._discount = float(discount)
        if observation_type is None:  # default: information state when the game provides it (:231-235)
            observation_type = ObservationType.INFORMATION_STATE if d.info_size else ObservationType.OBSERVATION
        if observation_type == ObservationType.INFORMATION_STATE and not d.info_size:
            raise ValueError(f"information_state_tensor not supported by {game_string}")
        self._use_observation = observation_type == ObservationType.OBSERVATION
        self._seed, self._index_offset, self._t = int(seed), int(index_offset), 0
        dev = ctx.device
        n, P = self.num_envs, self.num_players
        self._should_reset = torch.ones(n, dtype=torch.uint8, device=dev)
        self._cur = torch.empty(n, dtype=torch.int8, device=dev)
        self._type = torch.empty(n, dtype=torch.uint8, device=dev)
        self._rewards = torch.empty((n, P), dtype=torch.float64, device=dev)
        self._mask = torch.empty((n, d.mask_words), dtype=torch.int32, device=dev)
        self._no_actions = torch.full((n,), -1, dtype=torch.int32, device=dev)
        self._compact = bool(compact)
        if self._compact:
            self._flags = torch.full((n,), 2, dtype=torch.uint8, device=dev)      # LAST: every environment starts an episode
            self._rewards_x2 = torch.empty((n, P), dtype=torch.int8, device=dev)

    def __len__(self):
        return self.num_envs

    # -- specs (rl_environment.py:463-520) --------------------------------------------------
    def observation_spec(self):
        d = self.batch.desc
        size = d.obs_size if self._use_observation else d.info_size
        return dict(info_state=(size,), legal_actions=(d.num_distinct_actions,), current_player=())

    def action_spec(self):
        d = self.batch.desc
        return dict(num_actions=d.num_distinct_actions, min=0, max=d.num_distinct_actions - 1, dtype=int)

    # -- stepping -----------------------------------------------------------------------------
    def _launch(self, actions):
        if self._compact:
            a8 = torch.where(actions < 0, torch.full_like(actions, 255), actions).to(torch.uint8)
            check(lib().osg_env_step_compact(self.batch._h, a8.data_ptr(), self._flags.data_ptr(), self._seed, self._index_offset,
                                             self._t, self._rewards_x2.data_ptr(), self._mask.data_ptr()))
            # the reference's TimeStep types, formed from the compact arrays (what get_time_step hands out)
            self._type = self._flags & 3
            self._cur = ((self._flags >> 2).to(torch.int8) - 4)
            self._rewards = self._rewards_x2.to(torch.float64) * 0.5
            self._t += 1
            return
        check(lib().osg_env_step(self.batch._h, actions.data_ptr(), self._should_reset.data_ptr(), self._seed,
                                 self._index_offset, self._t, self._cur.data_ptr(), self._type.data_ptr(),
                                 self._rewards.data_ptr(), self._mask.data_ptr()))
        self._t += 1

    def get_time_step(self):
        """The current `TimeStep` of every environment (rl_environment.py:257-318)."""
        d = self.batch.desc
        width = d.num_distinct_actions
        shifts = torch.arange(32, device=self._mask.device, dtype=torch.int32)
        legal = ((self._mask.unsqueeze(-1) >> shifts) & 1).reshape(self.num_envs, -1)[:, :width].to(torch.bool)
        decision = (self._cur >= 0).unsqueeze(1)
        info_state, legal_actions = [], []
        for p in range(self.num_players):
            t = (self.batch.observation_tensor(p) if self._use_observation else self.batch.information_state_tensor(p))
            info_state.append(t)
            # State::LegalActions(player) is empty unless `player` is to act (spiel.h:366-372)
            legal_actions.append(legal & decision & (self._cur == p).unsqueeze(1))
        running = (self._type == StepType.MID).to(torch.float64)  # 0 at LAST (terminal) and at FIRST (None there)
        discounts = (running * self._discount).unsqueeze(1).expand(-1, self.num_players)
        obs = dict(info_state=info_state, legal_actions=legal_actions, current_player=self._cur.clone())
        return TimeStep(observations=obs, rewards=self._rewards.clone(), discounts=discounts,
                        step_type=self._type.clone())

    def reset(self):
        """Start a new episode in every environment (rl_environment.py:420-452)."""
        self._should_reset.fill_(1)
        if self._compact:
            self._flags.fill_(2)
        self._launch(self._no_actions)
        return self.get_time_step()

    def step(self, actions, reset_if_done=False):
        """`Environment.step` for every environment.  With reset_if_done (vector_env.py:36-62) the
        environments that just finished are reset at once and a 4-tuple
        (time_steps, rewards, done, unreset_time_steps) is returned like SyncVectorEnv.step."""
        a = torch.as_tensor(actions, dtype=torch.int32, device=self.ctx.device).contiguous()
        if a.numel() != self.num_envs:
            raise ValueError(f"Invalid number of actions. Expected {self.num_envs}, got {a.numel()}")
        self._launch(a)
        self.ctx.synchronize()  # surfaces illegal actions (IllegalActionError in the reference)
        ts = self.get_time_step()
        if not reset_if_done:
            return ts
        done = ts.last()
        # SyncVectorEnv.step(reset_if_done=True): finished environments restart at once, the others
        # report get_time_step() of their unchanged state (action -1 = leave as it is).
        self._launch(torch.full_like(a, -1))
        return self.get_time_step(), ts.rewards, done, ts
