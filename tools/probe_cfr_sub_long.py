import sys, time
sys.path.insert(0, ".")
import numpy as np, open_spiel_amd as osa
ctx = osa.Context(0)
a = osa.TabularSolver(ctx, "leduc_poker(players=3)", general_kernel="grid")
b = osa.TabularSolver(ctx, "leduc_poker(players=3)")
a.evaluate_and_update_policy(150); b.evaluate_and_update_policy(37); b.evaluate_and_update_policy(113)
ta, tb = a.tables(), b.tables()
print("150 iterations, sub (37 + 113) vs grid identical:", all(np.array_equal(ta[k], tb[k]) for k in ("regrets", "cum_policy", "cur_policy")), flush=True)
t0 = time.perf_counter(); b.evaluate_and_update_policy(30000); ctx.synchronize(); dt = time.perf_counter() - t0
print(f"30000 iterations in one call: {30000 / dt:.0f} it/s; nash_conv {b.nash_conv():.6f}; tables finite: {bool(np.isfinite(b.tables()['regrets']).all())}", flush=True)
