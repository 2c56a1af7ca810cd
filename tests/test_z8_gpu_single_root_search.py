"""The one-root search of MCTSBot (osg_mcts_tree_* with RandomRolloutEvaluator in the launch): the two-wavefront form
(lane 0 walks the tree, a second wavefront plays each leaf's playouts in parallel) must build exactly the tree of the
one-lane form — same counter streams, integer returns summed in another order (the lane form is what the oracle's
MCTSBot replays draw for draw in tests/test_gpu_mcts.py and tests/test_z5_gpu_mcts_evaluator.py; mcts.cc:353-467)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, sys
sys.path.insert(0, ROOT)
from open_spiel_amd import pyspiel_hip as pyspiel
out = {}
for game_string, moves, sims, n_rollouts, solve, uct_c in CASES:
    game = pyspiel.load_game(game_string)
    state = game.new_initial_state()
    for a in moves:
        state.apply_action(a)
    bot = pyspiel.MCTSBot(game, pyspiel.RandomRolloutEvaluator(n_rollouts, 7), uct_c, sims, 50, solve, 0x5EED, False)
    rows = []
    def walk(node, depth):
        rows.append([depth, int(node.action), int(node.player), int(node.explore_count), float(node.total_reward),
                     [float(x) for x in node.outcome]])
        for c in node.children:
            walk(c, depth + 1)
    for _ in range(2):                       # two searches with one bot: the second one's streams start at search 1
        walk(bot.mcts_search(state), 0)
    out[game_string + str(moves) + str(uct_c)] = rows
print("TREES " + json.dumps(out))
'''

CASES = [
    ("tic_tac_toe", [], 400, 20, True, 2.0),
    ("tic_tac_toe", [4, 0, 8], 300, 7, True, 2.0),
    ("connect_four", [3, 3, 2], 300, 70, False, 2.0),        # more playouts than one round of lanes... (70 > 64)
    ("hex(board_size=5)", [12, 6], 200, 5, True, 2.0),
    ("kuhn_poker", [0, 1], 150, 9, False, 2.0),              # chance inside the playouts
    ("leduc_poker", [0, 3, 1], 150, 4, False, 2.0),
    ("hex(board_size=13)", [84, 70], 60, 3, False, 2.0),     # 167 children per node: the sequential expansion, nine-bit fields unused
    ("hex(board_size=19)", [180], 30, 2, False, 2.0),        # 360 children: the nine-bit action / child-count fields
    # round 6: the lockstep form's arg-max goes through an fp32 filter with an exact fallback; exploration constants at
    # which single precision cannot separate the children (values apart by less than its resolution), overflows
    # (1e30 * sqrt: +inf in fp32) or carries nothing (0) must still build the one-lane form's tree
    ("tic_tac_toe", [], 400, 20, True, 1e-9),
    ("tic_tac_toe", [4], 400, 3, False, 0.0),
    ("tic_tac_toe", [], 300, 20, True, 1e30),
    ("connect_four", [3, 3], 400, 5, True, 1e-5),
    ("connect_four", [], 300, 2, False, 0.37),
]


def _trees(coop):
    env = dict(os.environ, OSG_MCTS_COOP="1" if coop else "0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    code = f"ROOT={ROOT!r}\nCASES={CASES!r}\n" + CHILD
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("TREES ")][-1]
    return json.loads(line[6:])


def test_two_wavefront_one_root_search_builds_the_tree_of_the_one_lane_form():
    import __graft_entry__ as ge
    ge.build()
    coop, plain = _trees(True), _trees(False)
    assert coop.keys() == plain.keys()
    for key in coop:
        a, b = coop[key], plain[key]
        assert len(a) == len(b) and len(a) > 10, key
        for ra, rb in zip(a, b):
            assert ra[:4] == rb[:4], (key, ra, rb)                       # depth, action, player, visits
            assert ra[4] == rb[4], (key, ra, rb)                         # total reward: bit for bit
            assert np.array_equal(np.array(ra[5]), np.array(rb[5]), equal_nan=True), (key, ra, rb)
