#!/usr/bin/env python3
"""Round-2 golden vectors, produced by RUNNING the genuine reference (oracle/_ref/libspiel_ref.so: the
reference's own .cc files compiled unmodified by oracle/Makefile.ref, now including algorithms/cfr_br.cc).
Run in the build container (needs /root/reference):

    python tests/golden/make_reference_vectors_r2.py

Output: tests/golden/reference_vectors_r2.npz

  cfr_br/<game>/<iters>/...            CFRBRSolver tables after <iters> EvaluateAndUpdatePolicy (cfr_br.cc:48-83):
                                       keys, nact, legal, regrets, cum_policy, cur_policy, avg_policy, nash_conv
  mccfr/<game>/<kind>/<seed>/<iters>/...  ExternalSamplingMCCFRSolver(game, seed, kSimple | kFull) tables after
                                       <iters> RunIteration() — std::mt19937(seed) + std::uniform_real_distribution,
                                       the stream the device reproduces draw for draw through
                                       osg_mccfr_sample_uniforms / RunIteration(std::mt19937*)

Consumers: tests/test_z4_gpu_reference_vectors_r2.py (the HIP engine, through the C-ABI and the host mirror)
and tests/test_reference_vectors.py (the restatement, CPU).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

CFR_BR = [("kuhn_poker", [1, 2, 10, 60]), ("leduc_poker", [1, 2, 5]), ("kuhn_poker(players=3)", [1, 8])]
MCCFR = [  # game, kind, seed, checkpoints (each run from scratch)
    ("kuhn_poker", "mccfr_simple", 7, [1, 40, 400]),
    ("kuhn_poker", "mccfr_full", 11, [1, 40, 400]),
    ("leduc_poker", "mccfr_simple", 3, [1, 25, 120]),
    ("leduc_poker", "mccfr_full", 5, [1, 25, 120]),
    ("kuhn_poker(players=3)", "mccfr_simple", 9, [60]),
    ("kuhn_poker(players=3)", "mccfr_full", 13, [60]),
]


def table_arrays(prefix, solver, out):
    t = solver.tables()
    out[prefix + "keys"] = np.frombuffer("\n".join(t["keys"]).encode(), np.uint8)
    for k in ("nact", "legal", "regrets", "cum_policy", "cur_policy", "avg_policy"):
        out[prefix + k] = t[k]
    out[prefix + "nash_conv"] = np.float64(solver.nash_conv())


def main():
    import reference_py as ref
    if not ref.sources_present():
        raise SystemExit("needs /root/reference (the genuine reference sources)")
    ref.build()
    out = {}
    for game, checkpoints in CFR_BR:
        s = ref.Solver(ref.Game(game), "cfr_br")
        done = 0
        for cp in checkpoints:
            s.iterate(cp - done)
            done = cp
            table_arrays(f"cfr_br/{game}/{cp}/", s, out)
    for game, kind, seed, checkpoints in MCCFR:
        for cp in checkpoints:
            s = ref.Solver(ref.Game(game), kind, seed)
            s.iterate(cp)
            table_arrays(f"mccfr/{game}/{kind}/{seed}/{cp}/", s, out)
    path = os.path.join(ROOT, "tests", "golden", "reference_vectors_r2.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
