"""A larger one-off replay sweep of the wave-per-root search on the hex boards above 128 cells than the test suite runs
(tests/test_gpu_mcts.py::test_mcts_wave_layout_on_the_boards_above_128_cells checks 4-10 roots per configuration): a few
hundred roots per board, every root's visits / rewards / best action against the oracle's replay (counter_layout=2).
Checker-side script (it calls oracle/): `python tools/sweep_wave_wide.py [scale]` on a GPU box; one line per configuration."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import open_spiel_amd as osa
import oracle_py as oracle
ctx = osa.Context(0)
CASES = [("hex(board_size=12)", 256, 128, 1, False, 70), ("hex(board_size=13)", 384, 128, 1, True, 150),
         ("hex(board_size=14)", 192, 100, 2, False, 90), ("hex(board_size=16)", 192, 100, 1, True, 230),
         ("hex(board_size=17)", 128, 100, 1, False, 200), ("hex(board_size=19)", 192, 128, 1, False, 300),
         ("hex(num_cols=31,num_rows=9)", 128, 100, 1, True, 200)]
bad_total = 0
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1      # roots x scale, simulations x min(scale, 3)
for game, n, sims, n_rollouts, solve, max_stop in CASES:
    n, sims = n * scale, sims * min(scale, 3)
    t0 = time.time()
    og = oracle.Game(game)
    rng = np.random.default_rng(5)
    stop = rng.integers(0, max_stop + 1, n).astype(np.int32)
    rec = og.random_playouts(5, n, stop=stop)
    roots = osa.StateBatch(ctx, game, n)
    for t in range(og.max_plies):
        if (rec["actions"][:, t] < 0).all():
            break
        roots.apply_actions(torch.from_numpy(rec["actions"][:, t].astype(np.int32)))
    seed, offset = 0x5EEDF00D, 31337
    res = roots.mcts_search(uct_c=2.0, max_simulations=sims, n_rollouts=n_rollouts, solve=solve, seed=seed, index_offset=offset, layout=2)
    visits = res["child_visits"].cpu().numpy(); reward = res["child_reward"].cpu().numpy(); best = res["best_action"].cpu().numpy()
    stats = res["root_stats"].cpu().numpy()
    bad = 0
    for i in range(n):
        st = og.new_initial_state()
        for a in rec["actions"][i][rec["actions"][i] >= 0]:
            st.apply_action(int(a))
        want = st.mcts_search(2.0, sims, n_rollouts, 4096, solve, 0, counter_root=offset + i, counter_seed=seed, counter_layout=2)
        ok = stats[i, 0] == want["root_visits"]
        for a, cnt, tot, out in want["children"]:
            ok = ok and visits[i, int(a)] == cnt and reward[i, int(a)] == tot
        if len(want["children"]):
            ok = ok and best[i] == want["best_action"]
        bad += 0 if ok else 1
    bad_total += bad
    print(f"{game}: {n} roots x {sims} simulations ({n_rollouts} playouts, solve={solve}): {'MISMATCH in ' + str(bad) + ' roots' if bad else 'identical'}  {time.time() - t0:.1f} s", flush=True)
sys.exit(1 if bad_total else 0)
