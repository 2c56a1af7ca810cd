#!/usr/bin/env python3
"""A larger one-off differential sweep than the test suite runs: seeded playouts of the restatement against the
genuine reference build (oracle/_ref), ~6e5 playouts over ten game configurations, every per-ply record compared
(tensors too for the smaller runs).  Test infrastructure only; needs oracle/_ref/libspiel_ref.so.
    python tools/differential_sweep.py      # round 1: all identical (profiles/r01_differential_sweep_container.log)"""
import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import oracle_py as O, reference_py as R
cfgs = [("tic_tac_toe", 60000), ("connect_four", 30000), ("connect_four(rows=4,columns=5,x_in_row=3)", 30000), ("hex(board_size=9)", 3000), ("hex(board_size=6,swap=True)", 10000), ("hex(num_rows=5,num_cols=7)", 6000), ("kuhn_poker(players=4)", 50000), ("leduc_poker", 60000), ("leduc_poker(players=3)", 30000), ("leduc_poker(players=3,suit_isomorphism=True,action_mapping=True)", 20000)]
for g, n in cfgs:
    t = time.time()
    try:
        og, rg = O.Game(g), R.Game(g)
    except Exception as e:
        print(g, "load:", e); continue
    bad = []
    for seed in (11, 12):
        a = og.random_playouts(seed, n, want_obs=(n <= 30000), want_info=(n <= 30000))
        b = rg.random_playouts(seed, n, want_obs=(n <= 30000), want_info=(n <= 30000))
        for k in a:
            if a[k] is None or isinstance(a[k], int): continue
            if not np.array_equal(a[k], b[k]): bad.append((seed, k))
    print(g, 2 * n, "playouts", "MISMATCH %s" % bad if bad else "identical", "%.1fs" % (time.time() - t), flush=True)
