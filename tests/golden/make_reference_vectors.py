#!/usr/bin/env python3
"""Golden vectors produced by RUNNING the genuine reference implementation.

Run in the build container (needs /root/reference; it does not exist on the GPU box):

    python tests/golden/make_reference_vectors.py

The producer is oracle/_ref/libspiel_ref.so: the reference's own .cc files for the hot path
(spiel.cc, the five games, mcts.cc, cfr.cc, external_sampling_mccfr.cc, tabular_exploitability.cc,
best_response.cc, ...) compiled unmodified from /root/reference by oracle/Makefile.ref and driven
through the extern "C" entry points of oracle/spiel_oracle_capi.cpp (-DOSGO_GENUINE_REFERENCE).
Nothing in this file is computed by the restatement or by the HIP engine.

Output: tests/golden/reference_vectors.npz (compressed; < 1 MB).  Contents, per BASELINE.json game (plus four variants):

  play/<game>/...   seeded playouts (playout i draws from CounterRng(seed, i), see the capi file):
                    actions [n,L] i16, mask [n,L+1,W] u32 (LegalActions, chance outcomes at chance
                    nodes), cur_player [n,L+1] i8, terminal [n,L+1] u8, returns [n,L+1,P] f64,
                    obs / info [n,L+1,P,size] u8 (ObservationTensor / InformationStateTensor of
                    every player at every ply; all values are small non-negative integers)
  cfr/<game>/<kind>/<iters>/...   CFRSolver / CFRPlusSolver / CFRSolverBase(simultaneous) tables
                    after <iters> EvaluateAndUpdatePolicy: keys, nact, legal, regrets, cum_policy,
                    cur_policy, avg_policy (fp64, exact), nash_conv, exploitability, expected_returns
  mccfr/<game>/<kind>/<seed>/<iters>/...   ExternalSamplingMCCFRSolver tables (std::mt19937 +
                    std::uniform_real_distribution: libstdc++ streams, reproducible)
  ckpt/<game>/<kind>/<iters>/text[6]   CFRSolverBase::Serialize() of the solver at that checkpoint
                    (cfr.cc:284-307; lossless hex floats, and 6 decimals)
  state/<game>/history, text   SerializeGameAndState() of a mid-game state (spiel.cc:582-603)
  judge/<game>/...  NashConv / exploitability of the uniform and first-action policies
  census/<game>     (chance, decision, terminal, infostates)

Consumers: tests/test_reference_vectors.py (the restatement reproduces every array, CPU) and
tests/test_z1_gpu_reference_vectors.py (the HIP engine reproduces them through the C-ABI).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

PLAYOUTS = [  # game, seed, n
    ("tic_tac_toe", 0x601D, 64),
    ("connect_four", 0x601D, 48),
    ("hex(board_size=9)", 0x601D, 12),
    ("kuhn_poker", 0x601D, 64),
    ("leduc_poker", 0x601D, 64),
    # variants (other geometries, more players)
    ("connect_four(rows=5,columns=6,x_in_row=3)", 0x601D, 48),
    ("hex(num_cols=3,num_rows=4)", 0x601D, 48),
    ("kuhn_poker(players=3)", 0x601D, 48),
    ("leduc_poker(players=3)", 0x601D, 32),
]
CFR = [  # game, kind, checkpoints
    ("kuhn_poker", "cfr", [1, 10, 15, 100]),
    ("kuhn_poker", "cfr_plus", [7, 50]),
    ("kuhn_poker", "cfr_simultaneous", [2, 20]),
    ("leduc_poker", "cfr", [1, 5]),
    ("leduc_poker", "cfr_plus", [3]),
]
MCCFR = [  # game, kind, seed, iters
    ("kuhn_poker", "mccfr_simple", 7, 200),
    ("leduc_poker", "mccfr_simple", 3, 60),
]
CHECKPOINTS = [("kuhn_poker", "cfr", 10), ("kuhn_poker", "cfr_plus", 7), ("leduc_poker", "cfr", 5)]  # must be CFR checkpoints above
JUDGE = ["kuhn_poker", "leduc_poker", "kuhn_poker(players=3)"]


def table_arrays(prefix, solver, out):
    t = solver.tables()
    out[prefix + "keys"] = np.frombuffer("\n".join(t["keys"]).encode(), np.uint8)
    for k in ("nact", "legal", "regrets", "cum_policy", "cur_policy", "avg_policy"):
        out[prefix + k] = t[k]
    out[prefix + "nash_conv"] = np.float64(solver.nash_conv())
    out[prefix + "exploitability"] = np.float64(solver.exploitability())
    out[prefix + "expected_returns"] = solver.expected_returns()


def main():
    import reference_py as ref
    if not ref.sources_present():
        raise SystemExit("needs /root/reference (the genuine reference sources)")
    ref.build()
    out = {}
    for game, seed, n in PLAYOUTS:
        g = ref.Game(game)
        rec = g.random_playouts(seed, n, want_obs=True, want_info=True)
        p = f"play/{game}/"
        out[p + "seed"] = np.uint64(seed)
        for k in ("actions", "mask", "cur_player", "terminal", "returns"):
            out[p + k] = rec[k]
        for k in ("obs", "info"):
            if rec[k] is not None:
                a = rec[k]
                assert (a >= 0).all() and (a <= 255).all() and (a == np.round(a)).all(), (game, k)
                out[p + k] = a.astype(np.uint8)
    # SerializeGameAndState (spiel.cc:582-603) of mid-game states: the history of playout 0 cut at half its length
    for game, seed, n in PLAYOUTS:
        g = ref.Game(game)
        acts = [int(a) for a in out[f"play/{game}/actions"][0] if a >= 0]
        acts = acts[:max(1, len(acts) // 2)]
        s = g.new_initial_state()
        for a in acts:
            s.apply_action(a)
        out[f"state/{game}/history"] = np.array(acts, np.int64)
        out[f"state/{game}/text"] = np.frombuffer(s.serialize_game_and_state().encode(), np.uint8)
    for game, kind, checkpoints in CFR:
        g = ref.Game(game)
        s = ref.Solver(g, kind)
        done = 0
        for cp in checkpoints:
            s.iterate(cp - done)
            done = cp
            table_arrays(f"cfr/{game}/{kind}/{cp}/", s, out)
            if (game, kind, cp) in CHECKPOINTS:  # CFRSolverBase::Serialize, lossless and 6-digit
                out[f"ckpt/{game}/{kind}/{cp}/text"] = np.frombuffer(s.serialize(-1).encode(), np.uint8)
                out[f"ckpt/{game}/{kind}/{cp}/text6"] = np.frombuffer(s.serialize(6).encode(), np.uint8)
    for game, kind, seed, iters in MCCFR:
        g = ref.Game(game)
        s = ref.Solver(g, kind, seed)
        s.iterate(iters)
        table_arrays(f"mccfr/{game}/{kind}/{seed}/{iters}/", s, out)
    for game in JUDGE:
        g = ref.Game(game)
        out[f"judge/{game}/uniform_nash_conv"] = np.float64(g.eval_named_policy(0, 0))
        out[f"judge/{game}/uniform_exploitability"] = np.float64(g.eval_named_policy(0, 1))
        out[f"judge/{game}/first_action_nash_conv"] = np.float64(g.eval_named_policy(1, 0))
        out[f"census/{game}"] = np.array(g.tree_census(), np.int64)
    path = os.path.join(ROOT, "tests", "golden", "reference_vectors.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
