"""Policy evaluation and CFR-BR rates on leduc_poker: the jobs kernel against the one-workgroup kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, open_spiel_amd as osa
ctx = osa.Context(0)
for game in ("leduc_poker", "kuhn_poker"):
    for jobs in ("1", "0"):
        os.environ["OSG_EVAL_JOBS"] = jobs
        s = osa.TabularSolver(ctx, game, general_kernel=False if jobs == "1" else "path")
        s.evaluate_and_update_policy(20)
        s.nash_conv(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): nc = s.nash_conv()
        t1 = time.perf_counter()
        s.evaluate_and_update_policy_cfr_br(5); ctx.synchronize()
        t2 = time.perf_counter()
        s.evaluate_and_update_policy_cfr_br(400); ctx.synchronize()
        t3 = time.perf_counter()
        print(f"{game} jobs={jobs}: nash_conv {(t1 - t0) / 100 * 1e6:.1f} us/call ({nc:.6f}), cfr-br {400 / (t3 - t2):.0f} it/s", flush=True)

# the largest tree served: one evaluation of 3-player leduc_poker (1.83 M histories; its deal components do not fit a
# workgroup's LDS, so the one-workgroup kernel runs)
os.environ["OSG_EVAL_JOBS"] = "1"
s3 = osa.TabularSolver(ctx, "leduc_poker(players=3)")
s3.evaluate_and_update_policy(2); s3.nash_conv(); ctx.synchronize()
t0 = time.perf_counter(); nc = s3.nash_conv(); dt = time.perf_counter() - t0
print(f"leduc_poker(players=3): nash_conv {dt * 1e3:.2f} ms/call ({nc:.6f})", flush=True)
t0 = time.perf_counter(); s3.evaluate_and_update_policy_cfr_br(20); ctx.synchronize(); dt = time.perf_counter() - t0
print(f"leduc_poker(players=3): cfr-br {20 / dt:.0f} it/s ({dt / 20 * 1e3:.2f} ms per iteration; {s3.last_kernel()})", flush=True)
