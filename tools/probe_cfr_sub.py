"""3-player leduc_poker CFR: the persistent cooperative kernel (k_cfr_sub, kernel "sub") against the per-phase
launches ("grid"): tables bit for bit, then iterations/s of both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, open_spiel_amd as osa
ctx = osa.Context(0)
game = os.environ.get("PROBE_GAME", "leduc_poker(players=3)")
for plus in (() if os.environ.get("PROBE_ONLY_SUB") else (False, True)):
    a = osa.TabularSolver(ctx, game, general_kernel="grid", regret_matching_plus=plus, linear_averaging=plus)
    b = osa.TabularSolver(ctx, game, general_kernel="sub", regret_matching_plus=plus, linear_averaging=plus)
    print("histories", a.num_histories, "infostates", a.num_infostates, flush=True)
    for n in (1, 2, 5):
        a.evaluate_and_update_policy(n); b.evaluate_and_update_policy(n)
        ta, tb = a.tables(), b.tables()
        same = all(np.array_equal(ta[k], tb[k]) for k in ("regrets", "cum_policy", "cur_policy"))
        print(f"plus={plus} after {a.iteration} iterations: tables identical = {same}", flush=True)
        if not same:
            for k in ("regrets", "cum_policy", "cur_policy"):
                print(k, np.abs(ta[k] - tb[k]).max(), int((ta[k] != tb[k]).sum()))
for name in (("sub",) if os.environ.get("PROBE_ONLY_SUB") else ("grid", "sub")):
    s = osa.TabularSolver(ctx, game, general_kernel=name)
    s.evaluate_and_update_policy(3); ctx.synchronize()
    iters = 20 if name == "grid" else 200
    t0 = time.perf_counter(); s.evaluate_and_update_policy(iters); ctx.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {iters / dt:.1f} iterations/s, {dt / iters * 1e6:.1f} us per iteration", flush=True)
    if name == "sub":   # the reference's calling pattern: one iteration per call
        t0 = time.perf_counter()
        for _ in range(100): s.evaluate_and_update_policy(1)
        ctx.synchronize(); dt = time.perf_counter() - t0
        print(f"sub, one iteration per call: {100 / dt:.1f} iterations/s", flush=True)
auto = osa.TabularSolver(ctx, game)
auto.evaluate_and_update_policy(2); ctx.synchronize()
t0 = time.perf_counter(); auto.evaluate_and_update_policy(50); ctx.synchronize(); dt = time.perf_counter() - t0
print(f"auto: {50 / dt:.1f} iterations/s; nash_conv {auto.nash_conv():.6f}")
