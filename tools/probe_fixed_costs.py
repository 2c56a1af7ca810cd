"""Fixed cost of each entry point: the same call on a large batch with the smallest possible amount of work."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, open_spiel_amd as osa
ctx = osa.Context(0)
def t(f, iters=20, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters): f()
    torch.cuda.synchronize(); return (time.time() - t0) / iters * 1e6
for game, n in (("hex(board_size=9)", 1 << 16), ("connect_four", 1 << 16)):
    roots = osa.StateBatch(ctx, game, n)
    for layout in (1, 2):
        for sims in (1, 2, 16):
            us = t(lambda: roots.mcts_search(max_simulations=sims, seed=1, layout=layout), iters=5, warm=2)
            print(f"mcts_search {game} n={n} layout={layout} sims={sims}: {us:.1f} us", flush=True)
for game in ("kuhn_poker", "leduc_poker"):
    s = osa.TabularSolver(ctx, game)
    print(f"cfr iterate(1) {game}: {t(lambda: s.evaluate_and_update_policy(1)):.1f} us; iterate(100): {t(lambda: s.evaluate_and_update_policy(100), 5, 1):.1f} us", flush=True)
    m = osa.TabularSolver(ctx, game, mccfr=True)
    print(f"mccfr run(1) {game}: {t(lambda: m.run_mccfr(1, 1)):.1f} us; run(2^16): {t(lambda: m.run_mccfr(1, 1 << 16)):.1f} us", flush=True)
    print(f"evaluate_policy {game}: {t(lambda: s.nash_conv(), 10, 2):.1f} us", flush=True)
b = osa.StateBatch(ctx, "connect_four", 1 << 20)
print(f"rollout c4 2^20 roots x 1: {t(lambda: b.rollout(1, 1), 5, 1):.1f} us", flush=True)
