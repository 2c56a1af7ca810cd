#!/usr/bin/env python3
"""A larger one-off sweep of the HIP path against the oracle than the test suite runs: seeded playouts of ~40 game
configurations (every parameter the five games take, boards from the smallest to the largest the engine serves),
20 000 playouts each on two seeds, replayed ply by ply through the C-ABI — legal masks, player, terminal flag, returns
at EVERY position, the fused step's outputs and both tensors where the batch is small enough.  Test infrastructure
(it calls oracle/): `python tools/device_sweep.py [playouts]` on a GPU box; prints one line per configuration."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import open_spiel_amd as osa
import oracle_py as O

import faulthandler; faulthandler.enable()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
GAMES = [
    "tic_tac_toe", "connect_four", "connect_four(rows=4,columns=5,x_in_row=3)", "connect_four(rows=5,columns=6,x_in_row=3)",
    "connect_four(rows=7,columns=8)", "connect_four(x_in_row=5)", "connect_four(egocentric_obs_tensor=True)",
    "connect_four(rows=4,columns=4)", "connect_four(rows=8,columns=7,x_in_row=4)",
    "hex(board_size=2)", "hex(board_size=3)", "hex(board_size=5)", "hex(board_size=7)", "hex(board_size=9)", "hex(board_size=11)",
    "hex(num_rows=3,num_cols=4)", "hex(num_rows=5,num_cols=7)", "hex(num_rows=9,num_cols=4)", "hex(board_size=6,swap=True)",
    "hex(board_size=9,swap=True)", "hex(board_size=5,plain_obs_tensor=True)", "hex(board_size=4,plain_obs_tensor=True,swap=True)",
    "kuhn_poker", "kuhn_poker(players=3)", "kuhn_poker(players=4)", "kuhn_poker(players=7)", "kuhn_poker(players=10)",
    "leduc_poker", "leduc_poker(players=3)", "leduc_poker(action_mapping=True)", "leduc_poker(suit_isomorphism=True)",
    "leduc_poker(players=3,starting_player=2)", "leduc_poker(players=3,suit_isomorphism=True,action_mapping=True)",
    "leduc_poker(starting_player=1)", "leduc_poker(players=3,starting_player=1,action_mapping=True)",
    # the parameter ranges opened in round 4: hex above 128 actions, connect_four above 64 board bits, leduc 4-10 players
    "hex(board_size=12)", "hex(board_size=13)", "hex(board_size=14,swap=True)", "hex(board_size=15)", "hex(board_size=16)",
    "hex(board_size=19)", "hex(num_rows=19,num_cols=17,swap=True)", "hex(board_size=13,plain_obs_tensor=True)",
    "connect_four(rows=8,columns=8)", "connect_four(rows=9,columns=9,x_in_row=5)", "connect_four(rows=9,columns=10)",
    "connect_four(rows=7,columns=15,x_in_row=4)", "connect_four(rows=15,columns=8,egocentric_obs_tensor=True)",
    "leduc_poker(players=4)", "leduc_poker(players=5,suit_isomorphism=True)", "leduc_poker(players=7,action_mapping=True)",
    "leduc_poker(players=10)", "leduc_poker(players=8,starting_player=6)",
    # round 5: the boards of the 12-word hex record (three-word planes with spare bits: 65 ... 91 cells), at its edges
    "hex(board_size=8,swap=True)", "hex(num_rows=13,num_cols=7)", "hex(num_rows=10,num_cols=9,swap=True)", "hex(num_rows=9,num_cols=8)",
    "hex(num_rows=11,num_cols=6)", "hex(num_rows=12,num_cols=8)",
]
if os.environ.get("SWEEP_ONLY"): GAMES = [g for g in GAMES if g in os.environ["SWEEP_ONLY"].split(";")]
ctx = osa.Context(0)
bad_total = 0
for game in GAMES:
    t0 = time.time()
    try:
        og = O.Game(game)
    except Exception as e:  # noqa: BLE001
        print(f"{game}: oracle refuses: {e}", flush=True); continue
    try:
        probe = osa.StateBatch(ctx, game, 1)
    except Exception as e:  # noqa: BLE001
        print(f"{game}: engine refuses: {str(e)[:100]}", flush=True); continue
    del probe
    bad = []
    positions = 0
    for seed in (31, 32):
        want_t = og.observation_tensor_size * N * (og.max_plies + 1) * 4 < (3 << 30)
        rec = og.random_playouts(seed, N, want_obs=want_t, want_info=want_t and og.information_state_tensor_size > 0)
        L = og.max_plies
        a, b = osa.StateBatch(ctx, game, N), osa.StateBatch(ctx, game, N)
        for t in range(L + 1):
            bits = a.legal_actions_mask_bits().cpu().numpy().view(np.uint32)
            cur, term, rets = a.status()
            if not np.array_equal(bits, rec["mask"][:, t]): bad.append((seed, t, "mask"))
            if not np.array_equal(cur.cpu().numpy(), rec["cur_player"][:, t]): bad.append((seed, t, "player"))
            if not np.array_equal(term.cpu().numpy(), rec["terminal"][:, t]): bad.append((seed, t, "terminal"))
            if not np.array_equal(rets.cpu().numpy(), rec["returns"][:, t]): bad.append((seed, t, "returns"))
            if want_t:
                for p in range(min(og.num_players, 2)):
                    if not np.array_equal(a.observation_tensor(p).cpu().numpy(), rec["obs"][:, t, p]): bad.append((seed, t, f"obs{p}"))
                    if rec.get("info") is not None and not np.array_equal(a.information_state_tensor(p).cpu().numpy(), rec["info"][:, t, p]):
                        bad.append((seed, t, f"info{p}"))
            positions += N
            if t == L: break
            acts = rec["actions"][:, t]
            if og.num_distinct_actions > 255:         # (the fused step carries one-byte action ids: hex above 15 x 15 applies)
                a.apply_actions(torch.from_numpy(acts.astype(np.int32)))
                continue
            a8 = torch.from_numpy(np.where(acts < 0, 255, acts).astype(np.uint8)).cuda()
            mask, status = a.step(a8, dst=b)          # the fused kernel, out of place
            st = status.cpu().numpy()
            if not np.array_equal((st & 0x80) != 0, rec["terminal"][:, t + 1] != 0): bad.append((seed, t, "fused terminal"))
            if (st & 0x40).any(): bad.append((seed, t, "fused step called an oracle action illegal"))
            a, b = b, a
        del a, b, rec
    bad_total += len(bad)
    print(f"{game}: {2 * N} playouts, {positions} positions: {'MISMATCH ' + str(bad[:6]) if bad else 'identical'}  {time.time() - t0:.1f} s", flush=True)
print("TOTAL MISMATCHES", bad_total, flush=True)
sys.exit(1 if bad_total else 0)
