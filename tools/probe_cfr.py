import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, torch, open_spiel_amd as osa
ctx = osa.Context(0)
for game, iters in [("kuhn_poker", 20000), ("kuhn_poker(players=3)", 2000), ("leduc_poker", 500)]:
    s = osa.TabularSolver(ctx, game)
    s.evaluate_and_update_policy(10); torch.cuda.synchronize()
    t = time.time(); s.evaluate_and_update_policy(iters); torch.cuda.synchronize(); dt = time.time() - t
    print(game, "CFR iters/s", iters / dt, "us/iter", dt / iters * 1e6, flush=True)
for game, n in [("kuhn_poker", 1 << 20), ("leduc_poker", 1 << 20), ("leduc_poker", 1 << 22)]:
    s = osa.TabularSolver(ctx, game, mccfr=True)
    s.run_mccfr(1, 4096); torch.cuda.synchronize()
    t = time.time(); s.run_mccfr(1, n, first_trajectory=4096); torch.cuda.synchronize(); dt = time.time() - t
    print(game, "MCCFR traj/s", n / dt, "ms", dt * 1e3, flush=True)
