"""The reference's python/rl_environment.py (imported unmodified, running on the genuine C++ games through
oracle/pyspiel_over_capi.py) against the restated environment that tests/test_gpu_vector_env.py compares the
device's BatchedEnvironment with.  Same chance draws on both sides: rl_environment.Environment takes a
`chance_event_sampler`, here one that draws like the device (counter stream (seed, env index, step index),
SampleAction's CDF scan).  Pins the 8(f) "batched RL environment" row to the reference's own file.

Needs /root/reference: skipped elsewhere."""
import os
import sys

import numpy as np
import pytest

from test_gpu_fullsize import CounterRng
from test_gpu_vector_env import FIRST, LAST, MID, OracleEnvironment

REFERENCE_ROOT = os.environ.get("OSG_REFERENCE_ROOT", "/root/reference")


class DeviceLikeChanceSampler:
    """chance_event_sampler for rl_environment.Environment: the device's draws."""

    def __init__(self, seed, index):
        self.seed_, self.index = seed, index
        self.rng = None

    def begin(self, t):
        self.rng = CounterRng(self.seed_, self.index, t)

    def seed(self, seed=None):  # rl_environment.Environment.seed() forwards here
        del seed

    def __call__(self, state):
        z, acc = self.rng.unit(), 0.0
        outcomes = state.chance_outcomes()
        for a, pr in outcomes:
            if acc <= z < acc + pr:
                return a
            acc += pr
        return outcomes[-1][0]


@pytest.fixture(scope="module")
def rl_environment(reference):
    if not reference.sources_present():
        pytest.skip("needs the reference sources (/root/reference)")
    import pyspiel_over_capi
    pyspiel = pyspiel_over_capi.install(reference)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from open_spiel.python import rl_environment as rl
    return pyspiel, rl


@pytest.mark.parametrize("game,obs_type,steps", [
    ("kuhn_poker", None, 40), ("leduc_poker", None, 60), ("leduc_poker", "observation", 40),
    ("tic_tac_toe", None, 40), ("connect_four", None, 90), ("hex(board_size=5)", None, 60),
    ("kuhn_poker(players=3)", None, 40),
])
def test_restated_environment_equals_rl_environment_py(oracle, rl_environment, game, obs_type, steps):
    pyspiel, rl = rl_environment
    seed, discount = 0xE27, 0.99
    og = oracle.Game(game)
    P = og.num_players
    use_obs = obs_type == "observation" or og.information_state_tensor_size == 0
    agent = np.random.default_rng(11)
    for index in (7000, 7001, 7013):
        sampler = DeviceLikeChanceSampler(seed, index)
        env = rl.Environment(pyspiel.load_game(game), discount=discount, chance_event_sampler=sampler,
                             observation_type=(rl.ObservationType.OBSERVATION if use_obs
                                               else rl.ObservationType.INFORMATION_STATE))
        mine = OracleEnvironment(og, seed, index, discount, use_obs)
        sampler.begin(0)
        ts, want = env.reset(), mine.reset(0)
        finished = 0
        for t in range(1, steps + 1):
            what = f"{game} env {index} step {t - 1}"
            assert {rl.StepType.FIRST: FIRST, rl.StepType.MID: MID, rl.StepType.LAST: LAST}[ts.step_type] == want["step_type"], what
            assert ts.observations["current_player"] == want["current_player"], what
            for p in range(P):
                np.testing.assert_array_equal(np.asarray(ts.observations["info_state"][p], np.float32),
                                              want["info_state"][p], what)
                assert list(ts.observations["legal_actions"][p]) == want["legal_actions"][p], what
            if want["rewards"] is None:
                assert ts.rewards is None and ts.discounts is None, what
            else:
                assert list(ts.rewards) == want["rewards"], what
                assert list(ts.discounts) == want["discounts"], what
            cp = want["current_player"]
            legal = want["legal_actions"][cp] if cp >= 0 else []
            action = int(agent.choice(legal)) if legal else 0   # ignored: the environment restarts
            finished += want["step_type"] == LAST
            sampler.begin(t)
            ts, want = env.step([action]), mine.step(action, t)
        assert finished > 0, "the run must cover episode ends and restarts"
